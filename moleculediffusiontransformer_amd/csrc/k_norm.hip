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

// ------------------------------------------------------------------------------------------------
// Fused GroupNorm apply: statistics + normalise + affine + FiLM + SiLU in ONE pass over a sample
// (ConvBlock1d.forward up to the convolution, modules.py:117-121).  As a GEMM prologue the same
// transform is recomputed for every tap and every N-tile (12x per element for a k=3 conv with 4 column
// tiles) and made those GEMMs VALU-bound; here each element is read once, transformed once and written
// once, and the convolution becomes a plain GEMM on the activated tensor (zero padding is then exact).
// One workgroup per sample; group g is owned by 256/G consecutive threads, which hold the group's
// elements in registers (two-pass variance, shuffle reduction in a fixed order: deterministic).
// TB threads per workgroup: 256, or 1024 for the large samples of the deep U-Net (16 K - 32 K elements: a quarter of the
// registers per thread, four times the loads in flight per sample)
// SPLIT (round 5): one workgroup per (sample, GROUP) instead of per sample -- the deep U-Net's samples are 64-128 KB, a
// workgroup of 1024 threads loaded all of it, reduced, then stored (two such workgroups per CU: 2.6-3.1 TB/s, the phases of a
// workgroup do not overlap); with a group per 256-thread workgroup eight of them share a CU and one's stores run under another's loads.
template <int NF4, int TB = 256, bool SPLIT = false>
__global__ __launch_bounds__(TB) void k_gn_act(GnActArgs a) {
  __shared__ float red[TB / 64];
  const int b = SPLIT ? blockIdx.x / a.groups : blockIdx.x, tid = threadIdx.x;
  const int tpg = SPLIT ? TB : TB / a.groups;      // threads per group (power of two)
  const int grp = SPLIT ? blockIdx.x % a.groups : tid / tpg, u = SPLIT ? tid : tid % tpg;
  const int q4 = a.gsize / 4;                      // float4 per row of the group's channel span
  const int nf4 = a.rows * q4;                     // float4 per group
  // two sources (round 6): a group's channels lie in x (the first ca) or in x2 (scaled), never in both (ca % gsize == 0)
  const bool second = a.x2 != nullptr && grp * a.gsize >= a.ca;
  const int pitch = a.x2 ? (second ? a.ld - a.ca : a.ca) : a.ld;
  const float* xb = second ? a.x2 + (int64_t)b * a.rows * pitch + (grp * a.gsize - a.ca) : a.x + (int64_t)b * a.rows * pitch + grp * a.gsize;
  const float in_scale = second ? a.scale2 : 1.0f;
  float* yb = a.y + (int64_t)b * a.rows * a.ld + grp * a.gsize;
  float4 v[NF4];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NF4; ++k) {
    const int e = u + k * tpg;
    v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < nf4) {
      const int r = e / q4, q = e - r * q4;
      v[k] = *reinterpret_cast<const float4*>(xb + (int64_t)r * pitch + 4 * q);
      if (second) { v[k].x *= in_scale; v[k].y *= in_scale; v[k].z *= in_scale; v[k].w *= in_scale; }
      s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
      if (a.raw16) {                                 // the raw input as bf16 (the A operand of the block's to_out convolution)
        const unsigned short h0 = __builtin_bit_cast(unsigned short, (__bf16)v[k].x), h1 = __builtin_bit_cast(unsigned short, (__bf16)v[k].y);
        const unsigned short h2 = __builtin_bit_cast(unsigned short, (__bf16)v[k].z), h3 = __builtin_bit_cast(unsigned short, (__bf16)v[k].w);
        *reinterpret_cast<uint2*>(a.raw16 + ((int64_t)b * a.rows + r) * a.ld + grp * a.gsize + 4 * q) =
            make_uint2(h0 | ((unsigned)h1 << 16), h2 | ((unsigned)h3 << 16));
      }
    }
  }
  auto group_sum = [&](float val) -> float {
    if (tpg <= 64) {
      for (int off = tpg >> 1; off >= 1; off >>= 1) val += __shfl_xor(val, off, 64);
      return val;
    }
    // a group spans several waves (G = 1 at 256 threads; G <= 8 at 1024): wave shuffle + the waves' partials through LDS,
    // summed in a fixed order
    for (int off = 32; off >= 1; off >>= 1) val += __shfl_xor(val, off, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = val;
    __syncthreads();
    const int nw = tpg / 64, w0 = (tid >> 6) / nw * nw;      // pairwise, in a fixed order: (r0 + r1) + (r2 + r3) ...
    float v[TB / 64];
#pragma unroll
    for (int k = 0; k < TB / 64; ++k) v[k] = k < nw ? red[w0 + k] : 0.f;
#pragma unroll
    for (int st = 1; st < TB / 64; st *= 2)
#pragma unroll
      for (int k = 0; k + st < TB / 64; k += 2 * st) v[k] += v[k + st];
    return v[0];
  };
  const float n = (float)(a.rows * a.gsize);
  const float mean = group_sum(s) / n;
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < NF4; ++k) {
    const int e = u + k * tpg;
    if (e < nf4) {
      const float d0 = v[k].x - mean, d1 = v[k].y - mean, d2 = v[k].z - mean, d3 = v[k].w - mean;
      ss += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
  }
  const float rstd = 1.0f / sqrtf(group_sum(ss) / n + a.eps);
#pragma unroll
  for (int k = 0; k < NF4; ++k) {
    const int e = u + k * tpg;
    if (e < nf4) {
      const int r = e / q4, q = e - r * q4;
      const int c = grp * a.gsize + 4 * q;
      const float4 ga = *reinterpret_cast<const float4*>(a.gamma + c);
      const float4 be = *reinterpret_cast<const float4*>(a.beta + c);
      float x[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
      const float g4[4] = {ga.x, ga.y, ga.z, ga.w}, b4[4] = {be.x, be.y, be.z, be.w};
      float f4[4] = {1.f, 1.f, 1.f, 1.f}, h4[4] = {0.f, 0.f, 0.f, 0.f};
      if (a.film) {
        const float4 fs = *reinterpret_cast<const float4*>(a.film + c);
        const float4 fh = *reinterpret_cast<const float4*>(a.film + a.ld + c);
        f4[0] = fs.x + 1.0f; f4[1] = fs.y + 1.0f; f4[2] = fs.z + 1.0f; f4[3] = fs.w + 1.0f;
        h4[0] = fh.x; h4[1] = fh.y; h4[2] = fh.z; h4[3] = fh.w;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float sc = rstd * g4[j];
        float t = x[j] * sc + (b4[j] - sc * mean);
        t = t * f4[j] + h4[j];
        if (a.silu) t = t / (1.0f + expf(-t));
        x[j] = t;
      }
      if (a.out16) {
        unsigned short hh[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) hh[j] = __builtin_bit_cast(unsigned short, (__bf16)x[j]);
        unsigned short* y16 = reinterpret_cast<unsigned short*>(a.y) + (int64_t)b * a.rows * a.ld + grp * a.gsize;
        *reinterpret_cast<uint2*>(y16 + (int64_t)r * a.ld + 4 * q) =
            make_uint2(hh[0] | ((unsigned)hh[1] << 16), hh[2] | ((unsigned)hh[3] << 16));
      } else {
        *reinterpret_cast<float4*>(yb + (int64_t)r * a.ld + 4 * q) = make_float4(x[0], x[1], x[2], x[3]);
      }
    }
  }
}

bool gn_act_eligible(int rows, int ld, int groups, int gsize) {
  if (groups <= 0 || 256 % groups || gsize % 4 || groups * gsize != ld) return false;
  const int tpg = 256 / groups, nf4 = rows * (gsize / 4);
  return (nf4 + tpg - 1) / tpg <= 32;      // <= 128 registers of elements per thread
}

hipError_t launch_gn_act(const GnActArgs& a, hipStream_t s) {
  if (a.batch <= 0) return hipSuccess;
  if (!gn_act_eligible(a.rows, a.ld, a.groups, a.gsize)) return hipErrorInvalidValue;
  if (a.x2 && (a.ca <= 0 || a.ca >= a.ld || a.ca % a.gsize || a.ca % 4 || (a.ld - a.ca) % 4)) return hipErrorInvalidValue;
  const int tpg = 256 / a.groups, per = (a.rows * (a.gsize / 4) + tpg - 1) / tpg;
  const int nf4 = a.rows * (a.gsize / 4);
  if (a.groups > 1 && nf4 >= 256 && nf4 <= 2048 && (int64_t)a.batch * a.groups < 0x7fffffffLL) {
    const dim3 grid((unsigned)(a.batch * a.groups));
    if (nf4 <= 256) hipLaunchKernelGGL((k_gn_act<1, 256, true>), grid, dim3(256), 0, s, a);
    else if (nf4 <= 512) hipLaunchKernelGGL((k_gn_act<2, 256, true>), grid, dim3(256), 0, s, a);
    else if (nf4 <= 1024) hipLaunchKernelGGL((k_gn_act<4, 256, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_gn_act<8, 256, true>), grid, dim3(256), 0, s, a);
    return hipGetLastError();
  }
  if (per <= 1) hipLaunchKernelGGL((k_gn_act<1>), dim3(a.batch), dim3(256), 0, s, a);
  else if (per <= 2) hipLaunchKernelGGL((k_gn_act<2>), dim3(a.batch), dim3(256), 0, s, a);
  else if (per <= 4) hipLaunchKernelGGL((k_gn_act<4>), dim3(a.batch), dim3(256), 0, s, a);
  else if (per <= 8) hipLaunchKernelGGL((k_gn_act<8>), dim3(a.batch), dim3(256), 0, s, a);
  else if (1024 % a.groups == 0 && per <= 16) hipLaunchKernelGGL((k_gn_act<4, 1024>), dim3(a.batch), dim3(1024), 0, s, a);
  else if (1024 % a.groups == 0) hipLaunchKernelGGL((k_gn_act<8, 1024>), dim3(a.batch), dim3(1024), 0, s, a);
  else if (per <= 16) hipLaunchKernelGGL((k_gn_act<16>), dim3(a.batch), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((k_gn_act<32>), dim3(a.batch), dim3(256), 0, s, a);
  return hipGetLastError();
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
