// Issue rate of v_mfma_f32_16x16x4_f32 (the exact-fp32 MFMA of the ring kernels' F32 forms and of k_attn_ctx) on gfx950: cycles per
// MFMA for NACC independent accumulators, operands from NOPS distinct register pairs, 1 or 2 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_f32_rate.hip -o /tmp/mfma_f32_rate && /tmp/mfma_f32_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int NOPS>
__global__ void k(const float* in, float* out, unsigned long long* cyc, int iters) {
  float a[NOPS], b[NOPS];
  for (int e = 0; e < NOPS; ++e) { a[e] = in[threadIdx.x + e]; b[e] = in[threadIdx.x + 64 + e]; }
  f32x4 acc[NACC];
  for (int n = 0; n < NACC; ++n) acc[n] = f32x4{0, 0, 0, 0};
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 64 / NACC; ++r)
#pragma unroll
      for (int n = 0; n < NACC; ++n)
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(r * NACC + n) % NOPS], b[(r * NACC + n) % NOPS], acc[n], 0, 0, 0);
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  f32x4 s = acc[0];
  for (int n = 1; n < NACC; ++n) s += acc[n];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC, int NOPS>
void run(const float* in, float* out, unsigned long long* cyc, int threads, int grid, const char* what) {
  const int iters = 200;
  hipLaunchKernelGGL((k<NACC, NOPS>), dim3(grid), dim3(threads), 0, 0, in, out, cyc, iters);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<NACC, NOPS>), dim3(grid), dim3(threads), 0, 0, in, out, cyc, iters);
  (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n = 64.0 * iters;
  printf("%-44s NACC %d NOPS %2d: %6.1f s_memtime ticks (100 MHz) -> %.3f us per wave, %.1f ns per MFMA; launch %.1f us\n", what, NACC, NOPS,
         (double)c, c / 100.0, c * 10.0 / n, ms * 1000);
}

int main() {
  float *in, *out; unsigned long long* cyc;
  (void)hipMalloc(&in, 4096 * 4); (void)hipMalloc(&out, 1024 * 1024 * 4); (void)hipMalloc(&cyc, 8);
  (void)hipMemset(in, 0, 4096 * 4);
  // one wave per SIMD on every CU (256 threads x 256 workgroups), then two (512 threads)
  run<2, 8>(in, out, cyc, 256, 256, "1 wave/SIMD, all CUs");
  run<4, 8>(in, out, cyc, 256, 256, "1 wave/SIMD, all CUs");
  run<8, 8>(in, out, cyc, 256, 256, "1 wave/SIMD, all CUs");
  run<4, 32>(in, out, cyc, 256, 256, "1 wave/SIMD, all CUs");
  run<4, 32>(in, out, cyc, 512, 256, "2 waves/SIMD, all CUs");
  run<4, 32>(in, out, cyc, 256, 8, "1 wave/SIMD, 8 CUs");
  run<4, 1>(in, out, cyc, 256, 256, "1 wave/SIMD, all CUs, one operand pair");
  return 0;
}
