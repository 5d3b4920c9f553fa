// A-stationary split-bf16 GEMM for the wide-N / small-K layers of the transformer blocks
// (to_q / to_kv / feed_forward.0 / Transformer1d.to_in: K = C in {128, 256}, N up to 1536).
//
// In the generic tiled kernel (k_gemm_bf16x3.hip) every (M, N) tile re-pays the same serial latency chain --
// fetch the A rows, LayerNorm statistics, normalise, split to bf16, stage -- for ~0.4 us of MFMA work, and
// with N/64 = 8..16 column tiles that chain dominates.  Here one workgroup owns 64 rows: it fetches them
// ONCE (16 lanes per row, 256-B coalesced), computes the row statistics in registers (two-pass, exact),
// applies the prologue, splits to bf16 hi/lo and parks the tile in LDS for the whole kernel.  It then walks
// its range of 64-column steps, streaming the pre-split weight tiles through a two-stage LDS ring (the next
// tile's global loads are in flight during the current tile's MFMAs) and writing each finished 64x64
// output tile from the accumulators (bias / GELU / residual epilogue as k_gemm_bf16x3.hip).
//
// Grid = (M/64) x nsplit: when M/64 < #CUs (level-2 layers at B = 1024) the column range is split over
// nsplit workgroups per row tile so that the chip still fills; each of them re-stages the (cheap) A tile.
#include <cstdlib>

#include "mdt_kernels.h"

namespace mdt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float silu_as(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float gelu_as(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

constexpr int AS_BM = 64;     // rows per workgroup
constexpr int AS_BN = 64;     // columns per step
constexpr int AS_BKW = 128;   // K depth of one streamed weight chunk
constexpr int AS_WROWB = 4 * AS_BKW + 16;

template <int PRO, int KC>    // K = 128 * KC
__global__ __launch_bounds__(256) void k_gemm_as(GemmArgs g, int nsplit, int dbg) {
  constexpr int K = AS_BKW * KC;
  constexpr int AROWB = 4 * K + 16;              // A row: hi plane (2K B) | lo plane (2K B) | pad
  constexpr int WSTAGE = AS_BN * AS_WROWB;
  constexpr int F4 = K / 64;                     // float4 per (thread, row): 16 lanes cover one row
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As = smem;
  unsigned char* Ws = smem + AS_BM * AROWB;
  // epilogue staging [64][68] fp32; K = 128 leaves room for TWO of them: the tile of step j is written out while step j + 1 parks
  // into the other, and the barrier that only protected the staging buffer goes away
  constexpr int NCS = KC == 1 ? 2 : 1;
  constexpr int CSF = AS_BM * (AS_BN + 4);
  float* Cs0 = reinterpret_cast<float*>(smem + AS_BM * AROWB + 2 * WSTAGE);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mtile = blockIdx.x / nsplit, split = blockIdx.x % nsplit;
  const int m0 = mtile * AS_BM;
  const int nsteps_all = (g.N + AS_BN - 1) / AS_BN;
  const int per = (nsteps_all + nsplit - 1) / nsplit;
  const int step0 = split * per;
  const int nsteps = min(per, nsteps_all - step0);
  if (nsteps <= 0) return;
  const __bf16* Whi = reinterpret_cast<const __bf16*>(g.W);
  const __bf16* Wlo = reinterpret_cast<const __bf16*>(g.W_lo);

  // ---- W chunk (step j, k-chunk c) -> registers: 64 rows x (128 hi + 128 lo) bf16 = 2048 16-B segments ----
  uint4 rw[8];
  auto load_w = [&](int it) {
    const int j = it / KC, c = it - j * KC;
    const int nb = (step0 + j) * AS_BN;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + i * 256;
      const int row = idx >> 5, seg = idx & 31;               // seg < 16: hi plane, else lo plane
      const int n = min(nb + row, g.N - 1);
      rw[i] = *reinterpret_cast<const uint4*>((seg < 16 ? Whi : Wlo) + (int64_t)n * K + c * AS_BKW + (seg & 15) * 8);
    }
  };
  auto store_w = [&](int it, unsigned char* stage) {
    const int j = it / KC;
    const int nb = (step0 + j) * AS_BN;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + i * 256;
      const int row = idx >> 5, seg = idx & 31;
      const uint4 w = (nb + row < g.N) ? rw[i] : make_uint4(0u, 0u, 0u, 0u);
      *reinterpret_cast<uint4*>(stage + row * AS_WROWB + seg * 16) = w;
    }
  };

  load_w(0);   // first weight chunk in flight while the A tile is prepared

  // ---- A tile: 4 passes of 16 rows; 16 lanes per row, each lane F4 float4 (unconditional, clamped loads) ----
  {
    const int sub = tid & 15, rr = tid >> 4;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int row = pass * 16 + rr;
      const int m = m0 + row;
      const bool ok = m < g.M;
      const int mc = ok ? m : g.M - 1;
      const int b = mc / g.r_out;
      const int src = (mc - b * g.r_out) * g.t_stride + g.t_off;
      const float4* p = reinterpret_cast<const float4*>(g.A + ((int64_t)b * g.r_in + src) * g.lda + g.a_col);
      float4 v[F4];
#pragma unroll
      for (int i = 0; i < F4; ++i) v[i] = p[sub + 16 * i];
      float mean = 0.f, rstd = 1.f;
      if constexpr (PRO == 1) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < F4; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) s += __shfl_xor(s, off, 16);
        mean = s / (float)K;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < F4; ++i) {
          const float d0 = v[i].x - mean, d1 = v[i].y - mean, d2 = v[i].z - mean, d3 = v[i].w - mean;
          ss += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 16);
        rstd = 1.0f / sqrtf(ss / (float)K + g.eps);
      }
#pragma unroll
      for (int i = 0; i < F4; ++i) {
        const int ci = (sub + 16 * i) * 4;
        float x[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
        if constexpr (PRO == 1) {
          const float4 gam = *reinterpret_cast<const float4*>(g.p0 + ci);
          const float4 bet = *reinterpret_cast<const float4*>(g.p1 + ci);
          x[0] = (x[0] - mean) * rstd * gam.x + bet.x;
          x[1] = (x[1] - mean) * rstd * gam.y + bet.y;
          x[2] = (x[2] - mean) * rstd * gam.z + bet.z;
          x[3] = (x[3] - mean) * rstd * gam.w + bet.w;
        } else if constexpr (PRO == 2) {
          const float4 gam = *reinterpret_cast<const float4*>(g.p0 + ci);
          const float4 bet = *reinterpret_cast<const float4*>(g.p1 + ci);
          const float4 fsc = *reinterpret_cast<const float4*>(g.p3 + ci);
          const float4 fsh = *reinterpret_cast<const float4*>(g.p3 + K + ci);
          const float ga[4] = {gam.x, gam.y, gam.z, gam.w}, be[4] = {bet.x, bet.y, bet.z, bet.w};
          const float fa[4] = {fsc.x, fsc.y, fsc.z, fsc.w}, fs[4] = {fsh.x, fsh.y, fsh.z, fsh.w};
          const float* st = g.p2 + (int64_t)b * g.groups * 2;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int grp = min((ci + e) / g.gsize, g.groups - 1);
            const float sc = st[grp * 2 + 1] * ga[e];
            x[e] = x[e] * sc + (be[e] - sc * st[grp * 2]);
            x[e] = x[e] * (fa[e] + 1.0f) + fs[e];
            if (g.pro_silu) x[e] = silu_as(x[e]);
          }
        } else if constexpr (PRO == 3) {
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = silu_as(x[e]);
        }
        u16x4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xv = ok ? x[e] : 0.f;
          const __bf16 h = (__bf16)xv;
          const __bf16 l = (__bf16)(xv - (float)h);
          hi[e] = __builtin_bit_cast(unsigned short, h);
          lo[e] = __builtin_bit_cast(unsigned short, l);
        }
        unsigned char* rp = As + row * AROWB + ci * 2;
        *reinterpret_cast<u16x4*>(rp) = hi;
        *reinterpret_cast<u16x4*>(rp + 2 * K) = lo;
      }
    }
  }
  store_w(0, Ws);
  __syncthreads();

  const int li = lane & 31, lh = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  const int total = nsteps * KC;
  f32x16 acc;
  for (int it = 0; it < total; ++it) {
    const int j = it / KC, c = it - j * KC;
    if (it + 1 < total && !(dbg & 2)) load_w(it + 1);
    if (c == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    }
    const unsigned char* ws = Ws + (it & 1) * WSTAGE;
    const unsigned char* ap = As + (wr * 32 + li) * AROWB + (c * AS_BKW) * 2 + lh * 16;
    const unsigned char* bp = ws + (wc * 32 + li) * AS_WROWB + lh * 16;
    if (!(dbg & 4))
#pragma unroll
    for (int ks = 0; ks < AS_BKW / 16; ++ks) {
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ap + ks * 32);
      const bf16x8 al = *reinterpret_cast<const bf16x8*>(ap + ks * 32 + 2 * K);
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(bp + ks * 32);
      const bf16x8 bl = *reinterpret_cast<const bf16x8*>(bp + ks * 32 + 2 * AS_BKW);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    }
    const bool last_k = c == KC - 1;
    float* Cs = Cs0 + (NCS == 2 ? (j & 1) * CSF : 0);
    if (last_k && !(dbg & 1)) {
      // park the finished 64x64 tile in LDS (32x32 C/D layout: col = lane&31, row = (reg&3)+8*(reg>>2)+4*(lane>>5))
#pragma unroll
      for (int r = 0; r < 16; ++r)
        Cs[(wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * (AS_BN + 4) + wc * 32 + li] = acc[r];
    }
    if (it + 1 < total && !(dbg & 2)) store_w(it + 1, Ws + ((it + 1) & 1) * WSTAGE);
    __syncthreads();
    if (last_k && !(dbg & 1)) {
      store_tile_coalesced<AS_BM, AS_BN>(Cs, g, m0, (step0 + j) * AS_BN);
      if constexpr (NCS == 1) __syncthreads();      // Cs is rewritten by the next step (two buffers: by the step after next, behind its barrier)
    }
  }
}

template <int PRO, int KC>
static hipError_t launch_as(const GemmArgs& g, hipStream_t s) {
  constexpr int K = AS_BKW * KC;
  const size_t smem = (size_t)AS_BM * (4 * K + 16) + 2 * (size_t)AS_BN * AS_WROWB + (size_t)(KC == 1 ? 2 : 1) * AS_BM * (AS_BN + 4) * 4;
  static DevOnce attr_once;                          // per device (mdt_kernels.h)
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_as<PRO, KC>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  const int mt = (g.M + AS_BM - 1) / AS_BM;
  const int nsteps = (g.N + AS_BN - 1) / AS_BN;
  int nsplit = 1;
  while (mt * nsplit < 256 && nsplit * 2 <= nsteps) nsplit *= 2;    // fill the 256 CUs when M is small
  static const int dbg = mdt_tuning_env("MDT_DBG") ? atoi(mdt_tuning_env("MDT_DBG")) : 0;   // ablation switches (tuning aid)
  hipLaunchKernelGGL((k_gemm_as<PRO, KC>), dim3((unsigned)(mt * nsplit)), dim3(256), smem, s, g, nsplit, dbg);
  return hipGetLastError();
}

bool gemm_as_eligible(const GemmArgs& g) {
  return g.phases <= 1 && g.W_lo && g.taps == 1 && g.t_stride == 1 && g.t_off == 0 && (g.cin == 128 || g.cin == 256) && g.N >= 128 &&
         g.pro >= 0 && g.pro <= 3 && (g.pro != 2 || g.p3);
}

hipError_t launch_gemm_as(const GemmArgs& g, hipStream_t s) {
  if (g.M <= 0) return hipSuccess;
  const int kc = g.cin / AS_BKW;
#define MDT_AS_CASE(P)                                               \
  case P:                                                            \
    return kc == 1 ? launch_as<P, 1>(g, s) : launch_as<P, 2>(g, s);
  switch (g.pro) {
    MDT_AS_CASE(0)
    MDT_AS_CASE(1)
    MDT_AS_CASE(2)
    MDT_AS_CASE(3)
    default: return hipErrorInvalidValue;
  }
#undef MDT_AS_CASE
}

}  // namespace mdt
