// GroupNorm statistics (nn.GroupNorm, modules.py:99-103 and :485) on token-major activations.
// One workgroup per (sample, group): two passes over the group's rows x gsize channels (the second
// pass re-reads lines that are still in L2), wave-shuffle + LDS reduction, biased variance,
// rstd = 1/sqrt(var + eps).  The normalisation itself is applied by the consuming GEMM's prologue.
#include "mdt_kernels.h"

namespace mdt {

__device__ __forceinline__ float wave_sum_n(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum_n(v);
  if constexpr (NT == 64) return v;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();  // protect red[] reuse between the two passes
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) t += red[w];
  return t;
}

template <int NT>
__global__ __launch_bounds__(NT) void k_gn_stats(GnStatsArgs g) {
  __shared__ float red[4];
  const int b = blockIdx.x / g.groups, grp = blockIdx.x % g.groups;
  const float* base = g.x + (int64_t)b * g.rows * g.ld + grp * g.gsize;
  const int n = g.rows * g.gsize;
  float s = 0.f;
  for (int idx = threadIdx.x; idx < n; idx += NT) {
    const int r = idx / g.gsize, c = idx - r * g.gsize;
    s += base[(int64_t)r * g.ld + c];
  }
  const float mean = block_sum<NT>(s, red) / (float)n;
  float ss = 0.f;
  for (int idx = threadIdx.x; idx < n; idx += NT) {
    const int r = idx / g.gsize, c = idx - r * g.gsize;
    const float d = base[(int64_t)r * g.ld + c] - mean;
    ss += d * d;
  }
  const float var = block_sum<NT>(ss, red) / (float)n;
  if (threadIdx.x == 0) {
    g.stats[(int64_t)blockIdx.x * 2] = mean;
    g.stats[(int64_t)blockIdx.x * 2 + 1] = 1.0f / sqrtf(var + g.eps);
  }
}

hipError_t launch_gn_stats(const GnStatsArgs& g, hipStream_t s) {
  if (g.batch <= 0) return hipSuccess;
  const unsigned nblk = (unsigned)(g.batch * g.groups);
  if (g.rows * g.gsize <= 256)
    hipLaunchKernelGGL((k_gn_stats<64>), dim3(nblk), dim3(64), 0, s, g);
  else
    hipLaunchKernelGGL((k_gn_stats<256>), dim3(nblk), dim3(256), 0, s, g);
  return hipGetLastError();
}

}  // namespace mdt
