// Split-bf16 ("bf16x3") implicit GEMM: fp32 operands, fp32 accumulation, three bf16 MFMAs per product.
//
//   a = a_hi + a_lo,  w = w_hi + w_lo   (bf16 round-to-nearest of the value and of its residual)
//   a*w ~= a_hi*w_hi + a_hi*w_lo + a_lo*w_hi          (dropped a_lo*w_lo term ~ 2^-18 |a w|)
//
// gfx950 has no TF32; its fp32 MFMA runs at the vector rate (157 TFLOP/s) while v_mfma_f32_32x32x16_bf16
// runs 16x faster, so three of them give fp32-class products at ~5x the fp32-MFMA rate.  Measured effect
// on the path's contract: a 64-step sample deviates 2e-6 (max-abs) from the fp32 CPU reference, against a
// 1e-4 budget and 1.5e-3 for plain bf16 ("Numerics of the split-bf16 GEMM").
//
// Same operator semantics as k_gemm.hip (taps, fused LayerNorm/GroupNorm/FiLM/SiLU prologue on A,
// bias/GELU/residual epilogue).  Weights arrive pre-split as two bf16 planes [N][K]; activations stay fp32
// in HBM and are split while they are staged (after the prologue), so no extra tensor is materialised.
//
// Workgroup = 4 waves as 2x2; block tile (64*TM) x (64*TN), wave tile (32*TM) x (32*TN), BK = 32.
// LDS row = [hi: 32 bf16 | lo: 32 bf16 | 16 B pad] = 144 B: MFMA fragments are conflict-free ds_read_b128.
// Two LDS stages: chunk k+1 is fetched to registers before, and written to LDS after, the MFMAs of chunk k,
// with one barrier per chunk.
#include "mdt_kernels.h"

namespace mdt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

constexpr int BK3 = 32;
constexpr int ROWB = 144;  // bytes per LDS row

__device__ __forceinline__ float silu3(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float gelu3(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

__device__ __forceinline__ void split4(const float x[4], u16x4& hi, u16x4& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const __bf16 h = (__bf16)x[e];
    const __bf16 l = (__bf16)(x[e] - (float)h);
    hi[e] = __builtin_bit_cast(unsigned short, h);
    lo[e] = __builtin_bit_cast(unsigned short, l);
  }
}

template <int PRO, int TM, int TN>
__global__ __launch_bounds__(256) void k_gemm3(GemmArgs g) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int RPT = BM / 32;          // A rows staged per thread (8 float4 per 32-float row)
  constexpr int WPT = BN * 8 / 256;     // 16-byte W segments staged per thread (4 hi + 4 lo per row)
  constexpr int STAGE = (BM + BN) * ROWB;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* rstat = reinterpret_cast<float*>(smem + 2 * STAGE);   // [BM][2], LayerNorm prologue only

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nt = (g.N + BN - 1) / BN;
  int id;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int m0 = (id / nt) * BM, n0 = (id % nt) * BN;
  const int K = g.taps * g.cin;
  const __bf16* Whi = reinterpret_cast<const __bf16*>(g.W);
  const __bf16* Wlo = reinterpret_cast<const __bf16*>(g.W_lo);

  const int c4 = tid & 7, r0 = tid >> 3;
  int rb[RPT], rs[RPT];
  bool rv[RPT];
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int m = m0 + r0 + i * 32;
    rv[i] = m < g.M;
    const int b = rv[i] ? m / g.r_out : 0;
    rb[i] = b;
    rs[i] = m - b * g.r_out;
  }

  if constexpr (PRO == 1) {
    // LayerNorm row statistics: 16 lanes per row, 4 rows per wave pass, two passes over L1-hot lines.
    const int sub = lane & 15;
    for (int rr = 0; rr < BM / 4; rr += 4) {
      const int row = wave * (BM / 4) + rr + (lane >> 4);
      const int m = m0 + row;
      float mean = 0.f, rstd = 0.f;
      const bool ok = m < g.M;
      const int b = ok ? m / g.r_out : 0;
      const int src = ok ? (m - b * g.r_out) * g.t_stride + g.t_off : 0;
      const float4* p = reinterpret_cast<const float4*>(g.A + ((int64_t)b * g.r_in + src) * g.lda + g.a_col);
      float s = 0.f;
      if (ok)
        for (int e = sub; e < g.cin / 4; e += 16) {
          const float4 v = p[e];
          s += (v.x + v.y) + (v.z + v.w);
        }
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) s += __shfl_xor(s, off, 16);
      mean = s / (float)g.cin;
      float ss = 0.f;
      if (ok)
        for (int e = sub; e < g.cin / 4; e += 16) {
          const float4 v = p[e];
          const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
          ss += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 16);
      rstd = 1.0f / sqrtf(ss / (float)g.cin + g.eps);
      if (sub == 0) {
        rstat[row * 2] = mean;
        rstat[row * 2 + 1] = rstd;
      }
    }
    __syncthreads();
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  float4 ra[RPT];
  uint4 rw[WPT];
  bool va[RPT];

  auto load_chunk = [&](int kc) {
    const int k0 = kc * BK3;
    const int tap = k0 / g.cin;
    const int ci = k0 - tap * g.cin + c4 * 4;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int src = rs[i] * g.t_stride + tap * g.t_dj + g.t_off;
      va[i] = rv[i] && src >= 0 && src < g.r_in;
      ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (va[i])
        ra[i] = *reinterpret_cast<const float4*>(g.A + ((int64_t)rb[i] * g.r_in + src) * g.lda + g.a_col + ci);
    }
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int idx = tid + i * 256;
      const int row = idx >> 3, seg = idx & 7;
      const int n = n0 + row;
      rw[i] = make_uint4(0u, 0u, 0u, 0u);
      if (n < g.N)
        rw[i] = *reinterpret_cast<const uint4*>((seg < 4 ? Whi : Wlo) + (int64_t)n * K + k0 + (seg & 3) * 8);
    }
  };

  auto store_chunk = [&](int kc, unsigned char* stage) {
    const int k0 = kc * BK3;
    const int tap = k0 / g.cin;
    const int ci = k0 - tap * g.cin + c4 * 4;
    float4 gam, bet, fsc, fsh;
    if constexpr (PRO == 1 || PRO == 2) {
      gam = *reinterpret_cast<const float4*>(g.p0 + ci);
      bet = *reinterpret_cast<const float4*>(g.p1 + ci);
    }
    int grp[4] = {0, 0, 0, 0};
    if constexpr (PRO == 2) {
      if (g.p3) {
        fsc = *reinterpret_cast<const float4*>(g.p3 + ci);
        fsh = *reinterpret_cast<const float4*>(g.p3 + g.cin + ci);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) grp[e] = min((ci + e) / g.gsize, g.groups - 1);
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      float x[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
      if (va[i]) {
        if constexpr (PRO == 1) {
          const int row = r0 + i * 32;
          const float mean = rstat[row * 2], rstd = rstat[row * 2 + 1];
          const float ga[4] = {gam.x, gam.y, gam.z, gam.w}, be[4] = {bet.x, bet.y, bet.z, bet.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = (x[e] - mean) * rstd * ga[e] + be[e];
        } else if constexpr (PRO == 2) {
          const float ga[4] = {gam.x, gam.y, gam.z, gam.w}, be[4] = {bet.x, bet.y, bet.z, bet.w};
          const float* st = g.p2 + (int64_t)rb[i] * g.groups * 2;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float mean = st[grp[e] * 2], rstd = st[grp[e] * 2 + 1];
            const float sc = rstd * ga[e];
            x[e] = x[e] * sc + (be[e] - sc * mean);
          }
          if (g.p3) {
            const float a[4] = {fsc.x, fsc.y, fsc.z, fsc.w}, s[4] = {fsh.x, fsh.y, fsh.z, fsh.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = x[e] * (a[e] + 1.0f) + s[e];
          }
          if (g.pro_silu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = silu3(x[e]);
          }
        } else if constexpr (PRO == 3) {
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = silu3(x[e]);
        }
      }
      u16x4 hi, lo;
      split4(x, hi, lo);
      unsigned char* rowp = stage + (r0 + i * 32) * ROWB + c4 * 8;
      *reinterpret_cast<u16x4*>(rowp) = hi;
      *reinterpret_cast<u16x4*>(rowp + 64) = lo;
    }
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int idx = tid + i * 256;
      const int row = idx >> 3, seg = idx & 7;
      *reinterpret_cast<uint4*>(stage + (BM + row) * ROWB + (seg >> 2) * 64 + (seg & 3) * 16) = rw[i];
    }
  };

  const int nk = K / BK3;
  const int li = lane & 31, lh = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  load_chunk(0);
  store_chunk(0, smem);
  __syncthreads();
  for (int kc = 0; kc < nk; ++kc) {
    unsigned char* cur = smem + (kc & 1) * STAGE;
    if (kc + 1 < nk) load_chunk(kc + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        const unsigned char* p = cur + (wr * 32 * TM + a * 32 + li) * ROWB + ks * 32 + lh * 16;
        ah[a] = *reinterpret_cast<const bf16x8*>(p);
        al[a] = *reinterpret_cast<const bf16x8*>(p + 64);
      }
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const unsigned char* p = cur + (BM + wc * 32 * TN + b * 32 + li) * ROWB + ks * 32 + lh * 16;
        bh[b] = *reinterpret_cast<const bf16x8*>(p);
        bl[b] = *reinterpret_cast<const bf16x8*>(p + 64);
      }
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[b], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[b], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
        }
    }
    if (kc + 1 < nk) store_chunk(kc + 1, smem + ((kc + 1) & 1) * STAGE);
    __syncthreads();
  }

  // ---- epilogue (32x32 C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) ----
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int n = n0 + wc * 32 * TN + b * 32 + li;
    const bool nok = n < g.N;
    const float bias = (g.bias && nok) ? g.bias[n] : 0.f;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wr * 32 * TM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int m = m0 + row;
        if (m >= g.M || !nok) continue;
        const int bb = m / g.r_out;
        const int64_t orow = (int64_t)bb * g.o_rows + (int64_t)(m - bb * g.r_out) * g.o_stride + g.o_off;
        float v = acc[a][b][r] + bias;
        if (g.act == 1) v = gelu3(v);
        if (g.res) v += g.res[orow * g.ldr + n];
        g.out[orow * g.ldc + g.o_col + n] = v;
      }
    }
  }
}

template <int PRO, int TM, int TN>
static hipError_t launch3(const GemmArgs& g, hipStream_t s) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  const int mt = (g.M + BM - 1) / BM, nt = (g.N + BN - 1) / BN;
  const size_t smem = 2 * (size_t)(BM + BN) * ROWB + (PRO == 1 ? BM * 2 * sizeof(float) : 0);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm3<PRO, TM, TN>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  hipLaunchKernelGGL((k_gemm3<PRO, TM, TN>), dim3((unsigned)(mt * nt)), dim3(256), smem, s, g);
  return hipGetLastError();
}

template <int PRO>
static hipError_t launch3_pro(const GemmArgs& g, hipStream_t s) {
  // Tile choice: the largest tile that still yields >= 2 workgroups per CU (256 CUs); small problems use
  // 64x64 so that the chip fills at all.
  auto tiles = [&](int bm, int bn) { return (int64_t)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn); };
  if (g.N > 64 && tiles(128, 128) >= 512) return launch3<PRO, 2, 2>(g, s);
  if (tiles(128, 64) >= 512) return launch3<PRO, 2, 1>(g, s);
  return launch3<PRO, 1, 1>(g, s);
}

hipError_t launch_gemm_bf16x3(const GemmArgs& g, hipStream_t s) {
  if (g.M <= 0) return hipSuccess;
  if (g.cin % 32) return hipErrorInvalidValue;   // caller falls back to the fp32-MFMA kernel for cin % 32 != 0
  switch (g.pro) {
    case 0: return launch3_pro<0>(g, s);
    case 1: return launch3_pro<1>(g, s);
    case 2: return launch3_pro<2>(g, s);
    case 3: return launch3_pro<3>(g, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace mdt
