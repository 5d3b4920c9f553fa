// Micro-benchmark: issue rate of bf16 MFMA shapes on one wave per SIMD (gfx950).  Build & run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDS>
__global__ void k16(const float* in, float* out, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[16384];
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)in[threadIdx.x + e]; b[e] = (__bf16)in[threadIdx.x + 8 + e]; }
  for (int t = threadIdx.x; t < 4096; t += blockDim.x) ((float*)sm)[t] = in[t & 255];
  __syncthreads();
  f32x4 acc[NACC];
  for (int n = 0; n < NACC; ++n) acc[n] = f32x4{0, 0, 0, 0};
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
    if (LDS) {
      a = *reinterpret_cast<const bf16x8*>(sm + ((threadIdx.x & 63) * 16 + (it & 7) * 1024));
      b = *reinterpret_cast<const bf16x8*>(sm + ((threadIdx.x & 63) * 16 + ((it + 3) & 7) * 1024 + 8192));
    }
#pragma unroll
    for (int r = 0; r < 12 / NACC; ++r)
#pragma unroll
      for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[n], 0, 0, 0);
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  f32x4 s = acc[0];
  for (int n = 1; n < NACC; ++n) s += acc[n];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void k32(const float* in, float* out, unsigned long long* cyc, int iters) {
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)in[threadIdx.x + e]; b[e] = (__bf16)in[threadIdx.x + 8 + e]; }
  f32x16 acc[2];
  for (int n = 0; n < 2; ++n) for (int e = 0; e < 16; ++e) acc[n][e] = 0;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int n = 0; n < 2; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0;
  for (int n = 0; n < 2; ++n) for (int e = 0; e < 16; ++e) s += acc[n][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  float *in, *out; unsigned long long* cyc;
  hipMalloc(&in, 1 << 20); hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 64);
  hipMemset(in, 0, 1 << 20);
  const int iters = 2000;
  auto report = [&](const char* name, int threads, int blocks, auto kern) {
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, in, out, cyc, iters);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, in, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s threads/blk %4d blocks %4d : %6.2f cycles per MFMA\n", name, threads, blocks, (double)c / (iters * 12.0));
  };
  report("16x16x32 bf16, 1 acc, regs", 256, 1, k16<1, false>);
  report("16x16x32 bf16, 4 acc, regs", 256, 1, k16<4, false>);
  report("16x16x32 bf16, 4 acc, regs, 1 wave/CU", 64, 1, k16<4, false>);
  report("16x16x32 bf16, 4 acc, regs, 8 waves/CU", 512, 1, k16<4, false>);
  report("16x16x32 bf16, 4 acc, regs, 256 blocks", 256, 256, k16<4, false>);
  report("16x16x32 bf16, 4 acc, 2 ds_read per 12", 256, 1, k16<4, true>);
  report("32x32x16 bf16, 2 acc, regs", 256, 1, k32);
  report("32x32x16 bf16, 2 acc, regs, 256 blocks", 256, 256, k32);
  return 0;
}
