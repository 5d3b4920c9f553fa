// What does the shader clock do under load?  Every workgroup runs split-bf16 MFMAs on register operands (4 waves, one per
// SIMD) for `iters` iterations and reports the shader-clock cycles (s_memtime) and the 100 MHz reference ticks
// (s_memrealtime) it took: cycles / ticks x 100 MHz = the clock the CU actually ran at.  1 workgroup against 256, short
// (~50 us, one launch of the fused kernels) against long (~50 ms, a sample() call) runs, and MFMA duty cycles (`gap`:
// s_sleep units between MFMA groups) between "all MFMA" and what the fused kernels reach.
// Build & run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/ubench/clock_probe.hip -o /tmp/cp && /tmp/cp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void kclock(float* out, unsigned long long* res, int iters, int gap) {
  const int lane = threadIdx.x & 63;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane - e); }
  f32x4 acc[4];
  for (int n = 0; n < 4; ++n) acc[n] = f32x4{0, 0, 0, 0};
  unsigned long long c0, c1, r0, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k & 3], 0, 0, 0);
    for (int s = 0; s < gap; ++s) __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
  f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0) { res[2 * blockIdx.x] = c1 - c0; res[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
  float* out; unsigned long long* res;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&res, 1024 * 16);
  for (int gap : {0, 2, 6}) {
    for (int blocks : {1, 64, 256}) {
      for (int iters : {600, 600000}) {
        hipLaunchKernelGGL(kclock, dim3(blocks), dim3(256), 0, 0, out, res, iters, gap);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kclock, dim3(blocks), dim3(256), 0, 0, out, res, iters, gap);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[512]; hipMemcpy(h, res, 16 * blocks, hipMemcpyDeviceToHost);
        double c = 0, r = 0; for (int b = 0; b < blocks; ++b) { c += h[2 * b]; r += h[2 * b + 1]; }
        c /= blocks; r /= blocks;
        printf("gap %d blocks %3d iters %6d: %9.1f us; %10.0f shader cycles, %9.0f ref ticks -> %6.0f MHz; %5.1f cycles / MFMA, MFMA duty %.2f\n",
               gap, blocks, iters, ms * 1e3, c, r, c / r * 100.0, c / (12.0 * iters), 12.0 * iters * 16 / c);
      }
    }
  }
  return 0;
}
