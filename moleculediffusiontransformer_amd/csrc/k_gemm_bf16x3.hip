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
// Workgroup = 4 waves as 2x2; block tile (64*TM) x (64*TN), wave tile (32*TM) x (32*TN), BK = 32*BKT.
// LDS row = [hi: BK bf16 | lo: BK bf16 | 16 B pad]: the row pitch is an odd multiple of 16 B, so the MFMA
// fragment reads (ds_read_b128, one row per lane) are bank-conflict free.
// The layers of this network are small (K <= 1536, often a single 64x64 tile per CU), so the kernel is
// latency- not throughput-bound: K is consumed in few, fat chunks (up to 128 deep, 64 KB of loads in flight
// per workgroup) instead of many thin ones.  Two LDS stages: chunk k+1 is fetched to registers before, and
// written to LDS after, the MFMAs of chunk k, with one barrier per chunk.
#include <cstdio>
#include <cstdlib>

#include "mdt_kernels.h"

namespace mdt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));


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

// NPROD = 3: split products (above).  NPROD = 1: PLAIN bf16 products -- operands rounded to bf16 once, one MFMA per product,
// fp32 accumulation; only the hi plane exists (in LDS and in the weight buffer).  This is the reduced-precision mode of
// BASELINE configs[4] ("bf16"), selected per op by MDT_G_WFMT = 1; its error budget is stated in DESIGN.md / the bf16 tests.
template <int PRO, int TM, int TN, int BKT, int STAGES, int NPROD = 3>
__global__ __launch_bounds__(256) void k_gemm3(GemmArgs g) {
  constexpr int BM = 64 * TM, BN = 64 * TN, BK3 = 32 * BKT;
  constexpr int PLANES = NPROD == 3 ? 2 : 1;
  constexpr int ROWB = 2 * PLANES * BK3 + 16;    // bytes per LDS row: hi plane | [lo plane] | pad
  constexpr int C4 = BK3 / 4;           // float4 per A row chunk
  constexpr int AROWS = 256 / C4;       // A rows covered per pass of the 256 threads
  constexpr int RPT = BM / AROWS;       // A rows staged per thread
  constexpr int HSEG = BK3 / 8;         // 16-byte W segments per row and plane
  constexpr int WSEG = PLANES * HSEG;   // 16-byte W segments per row (hi [and lo] planes)
  constexpr int WPT = BN * WSEG / 256;  // W segments staged per thread
  static_assert(WPT >= 1, "tile too small for 256 staging threads");
  constexpr int STAGE = (BM + BN) * ROWB;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* rstat = reinterpret_cast<float*>(smem + STAGES * STAGE);   // [BM][2], LayerNorm prologue only
  gemm_select_phase(g);

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

  const int c4 = tid % C4, r0 = tid / C4;
  int rb[RPT], rs[RPT];
  bool rv[RPT];
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int m = m0 + r0 + i * AROWS;
    rv[i] = m < g.M;
    const int b = rv[i] ? m / g.r_out : 0;
    rb[i] = b;
    rs[i] = m - b * g.r_out;
  }

  auto ln_prepass = [&]() {
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
      for (int e = sub; e < g.cin / 4; e += 16) {
        const float4 v = p[e];
        s += (v.x + v.y) + (v.z + v.w);
      }
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) s += __shfl_xor(s, off, 16);
      mean = s / (float)g.cin;
      float ss = 0.f;
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
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  struct Regs {
    float4 ra[RPT];
    uint4 rw[WPT];
    bool va[RPT];
  };

  // Every global load below is UNCONDITIONAL (addresses clamped into the tensor, results zeroed by select):
  // loads inside exec-masked branches make hipcc fall back to s_waitcnt vmcnt(0) and serialise the pipeline.
  auto load_chunk = [&](int kc, Regs& R) {
    const int k0 = kc * BK3;
    const int tap = k0 / g.cin;
    const int ci = k0 - tap * g.cin + c4 * 4;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int src = rs[i] * g.t_stride + tap * g.t_dj + g.t_off;
      const int srcc = min(max(src, 0), g.r_in - 1);
      R.va[i] = rv[i] && src == srcc;
      R.ra[i] = *reinterpret_cast<const float4*>(g.A + ((int64_t)rb[i] * g.r_in + srcc) * g.lda + g.a_col + ci);
    }
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / WSEG, seg = idx % WSEG;            // seg < WSEG/2: hi plane, else lo plane
      const int n = min(n0 + row, g.N - 1);
      R.rw[i] = *reinterpret_cast<const uint4*>((seg < HSEG ? Whi : Wlo) + (int64_t)n * K + k0 + (seg % HSEG) * 8);
    }
  };

  auto store_chunk = [&](int kc, const Regs& R, unsigned char* stage) {
    const int k0 = kc * BK3;
    const int tap = k0 / g.cin;
    const int ci = k0 - tap * g.cin + c4 * 4;
    float ga[4], be[4], fa[4], fs[4];
    int grp[4] = {0, 0, 0, 0};
    if constexpr (PRO == 1 || PRO == 2) {
      const float4 gam = *reinterpret_cast<const float4*>(g.p0 + ci);
      const float4 bet = *reinterpret_cast<const float4*>(g.p1 + ci);
      ga[0] = gam.x; ga[1] = gam.y; ga[2] = gam.z; ga[3] = gam.w;
      be[0] = bet.x; be[1] = bet.y; be[2] = bet.z; be[3] = bet.w;
    }
    if constexpr (PRO == 2) {
      // p3 is always bound for the GroupNorm prologue (a zero vector when the block has no FiLM)
      const float4 fsc = *reinterpret_cast<const float4*>(g.p3 + ci);
      const float4 fsh = *reinterpret_cast<const float4*>(g.p3 + g.cin + ci);
      fa[0] = fsc.x + 1.0f; fa[1] = fsc.y + 1.0f; fa[2] = fsc.z + 1.0f; fa[3] = fsc.w + 1.0f;
      fs[0] = fsh.x; fs[1] = fsh.y; fs[2] = fsh.z; fs[3] = fsh.w;
#pragma unroll
      for (int e = 0; e < 4; ++e) grp[e] = min((ci + e) / g.gsize, g.groups - 1);
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      float x[4] = {R.ra[i].x, R.ra[i].y, R.ra[i].z, R.ra[i].w};
      if constexpr (PRO == 1) {
        const int row = r0 + i * AROWS;
        const float mean = rstat[row * 2], rstd = rstat[row * 2 + 1];
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = (x[e] - mean) * rstd * ga[e] + be[e];
      } else if constexpr (PRO == 2) {
        const float* st = g.p2 + (int64_t)rb[i] * g.groups * 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float mean = st[grp[e] * 2], rstd = st[grp[e] * 2 + 1];
          const float sc = rstd * ga[e];
          x[e] = x[e] * sc + (be[e] - sc * mean);
          x[e] = x[e] * fa[e] + fs[e];
        }
        if (g.pro_silu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = silu3(x[e]);
        }
      } else if constexpr (PRO == 3) {
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = silu3(x[e]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = R.va[i] ? x[e] : 0.f;     // conv zero padding / rows past M
      unsigned char* rowp = stage + (r0 + i * AROWS) * ROWB + c4 * 8;
      if constexpr (NPROD == 3) {
        u16x4 hi, lo;
        split4(x, hi, lo);
        *reinterpret_cast<u16x4*>(rowp) = hi;
        *reinterpret_cast<u16x4*>(rowp + 2 * BK3) = lo;
      } else {
        u16x4 hi;
#pragma unroll
        for (int e = 0; e < 4; ++e) hi[e] = __builtin_bit_cast(unsigned short, (__bf16)x[e]);
        *reinterpret_cast<u16x4*>(rowp) = hi;
      }
    }
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / WSEG, seg = idx % WSEG;
      const uint4 w = (n0 + row < g.N) ? R.rw[i] : make_uint4(0u, 0u, 0u, 0u);
      *reinterpret_cast<uint4*>(stage + (BM + row) * ROWB + seg * 16) = w;   // hi segments then lo segments
    }
  };

  const int li = lane & 31, lh = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  auto mfma_chunk = [&](const unsigned char* cur) {
#pragma unroll
    for (int ks = 0; ks < BK3 / 16; ++ks) {
      bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        const unsigned char* p = cur + (wr * 32 * TM + a * 32 + li) * ROWB + ks * 32 + lh * 16;
        ah[a] = *reinterpret_cast<const bf16x8*>(p);
        if constexpr (NPROD == 3) al[a] = *reinterpret_cast<const bf16x8*>(p + 2 * BK3);
      }
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const unsigned char* p = cur + (BM + wc * 32 * TN + b * 32 + li) * ROWB + ks * 32 + lh * 16;
        bh[b] = *reinterpret_cast<const bf16x8*>(p);
        if constexpr (NPROD == 3) bl[b] = *reinterpret_cast<const bf16x8*>(p + 2 * BK3);
      }
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          if constexpr (NPROD == 3) {
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[b], acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[b], acc[a][b], 0, 0, 0);
          }
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
        }
    }
  };

  const int nk = K / BK3;
  Regs R;
  load_chunk(0, R);
  if constexpr (PRO == 1) ln_prepass();     // row statistics while the first chunk is in flight
  store_chunk(0, R, smem);
  __syncthreads();
  if constexpr (STAGES == 2) {
    for (int kc = 0; kc < nk; ++kc) {
      if (kc + 1 < nk) load_chunk(kc + 1, R);
      mfma_chunk(smem + (kc & 1) * STAGE);
      if (kc + 1 < nk) store_chunk(kc + 1, R, smem + ((kc + 1) & 1) * STAGE);
      __syncthreads();
    }
  } else {
    // One LDS stage: less LDS per workgroup, so 2-4 workgroups share a CU and cover each other's
    // load latency (used when the grid has several workgroups per CU).
    for (int kc = 0; kc < nk; ++kc) {
      if (kc + 1 < nk) load_chunk(kc + 1, R);
      mfma_chunk(smem);
      if (kc + 1 < nk) {
        __syncthreads();
        store_chunk(kc + 1, R, smem);
        __syncthreads();
      }
    }
  }

  // ---- epilogue: accumulators -> LDS (fp32 [BM][BN+4]) -> coalesced 16-byte stores (see mdt_kernels.h).
  // 32x32 C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
  __syncthreads();            // every wave is done reading the operand stages that Cs overlays
  float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wr * 32 * TM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        Cs[row * (BN + 4) + wc * 32 * TN + b * 32 + li] = acc[a][b][r];
      }
  __syncthreads();
  store_tile_coalesced<BM, BN>(Cs, g, m0, n0);
}

template <int PRO, int TM, int TN, int BKT, int STAGES, int NPROD = 3>
static hipError_t launch3(const GemmArgs& g, hipStream_t s) {
  constexpr int BM = 64 * TM, BN = 64 * TN, ROWB = (NPROD == 3 ? 128 : 64) * BKT + 16;
  const int mt = (g.M + BM - 1) / BM, nt = (g.N + BN - 1) / BN;
  size_t smem = STAGES * (size_t)(BM + BN) * ROWB + (PRO == 1 ? BM * 2 * sizeof(float) : 0);
  const size_t ctile = (size_t)BM * (BN + 4) * sizeof(float);     // epilogue staging reuses the same memory
  if (smem < ctile) smem = ctile;
  static DevOnce attr_once;                          // per device (mdt_kernels.h)
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm3<PRO, TM, TN, BKT, STAGES, NPROD>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  hipLaunchKernelGGL((k_gemm3<PRO, TM, TN, BKT, STAGES, NPROD>), dim3((unsigned)(mt * nt), 1, (unsigned)(g.phases > 1 ? g.phases : 1)), dim3(256), smem, s, g);
  return hipGetLastError();
}

template <int PRO, int STAGES>
static hipError_t launch3_cfg(const GemmArgs& g, hipStream_t s, int cfg) {
  switch (cfg) {
    case 0: return launch3<PRO, 2, 2, 1, STAGES>(g, s);
    case 1: return launch3<PRO, 2, 1, 2, STAGES>(g, s);
    case 2: return launch3<PRO, 1, 1, 4, STAGES>(g, s);
    case 3: return launch3<PRO, 1, 1, 2, STAGES>(g, s);
    default: return hipErrorInvalidValue;
  }
}

// MDT_TILE="<cfg>,<stages>" forces a configuration (tuning aid): cfg 0 = 128x128xBK32, 1 = 128x64xBK64,
// 2 = 64x64xBK128, 3 = 64x64xBK64.
static int g_force_cfg = -2, g_force_stages = 0;

template <int PRO>
static hipError_t launch3_pro(const GemmArgs& g, hipStream_t s) {
  if (g_force_cfg == -2) {
    g_force_cfg = -1;
    if (const char* e = mdt_tuning_env("MDT_TILE")) sscanf(e, "%d,%d", &g_force_cfg, &g_force_stages);
  }
  // Tile choice: the largest tile that still yields >= 2 workgroups per CU (256 CUs); otherwise 64x64 tiles
  // with the deepest K chunk the channel count allows (fewest iterations for the latency-bound small layers).
  auto tiles = [&](int bm, int bn) {
    return (int64_t)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn) * (g.phases > 1 ? g.phases : 1);
  };
  int cfg, stages;
  if (g.N > 64 && tiles(128, 128) >= 512) cfg = 0;
  else if (g.cin % 64 == 0 && tiles(128, 64) >= 512) cfg = 1;
  else if (g.cin % 128 == 0) cfg = 2;
  else if (g.cin % 64 == 0) cfg = 3;
  else cfg = 0;
  if (g_force_cfg >= 0) {
    const int need = g_force_cfg == 2 ? 128 : (g_force_cfg == 0 ? 32 : 64);
    if (g.cin % need == 0) cfg = g_force_cfg;
  }
  const int bm = (cfg <= 1) ? 128 : 64, bn = (cfg == 0) ? 128 : 64;
  stages = tiles(bm, bn) >= 512 ? 1 : 2;
  if (g_force_stages) stages = g_force_stages;
  return stages == 1 ? launch3_cfg<PRO, 1>(g, s, cfg) : launch3_cfg<PRO, 2>(g, s, cfg);
}

// Plain-bf16 products (NPROD = 1).  Tile choice: 128x128 tiles with a 64-deep chunk when they fill the chip, else as above.
// MDT_TILE1="<cfg>,<stages>": cfg 0 = 128x128xBK64, 1 = 128x128xBK32, 2 = 128x64xBK64, 3 = 64x64xBK128, 4 = 64x64xBK64
template <int PRO, int STAGES>
static hipError_t launch1_cfg(const GemmArgs& g, hipStream_t s, int cfg) {
  switch (cfg) {
    case 0: return launch3<PRO, 2, 2, 2, STAGES, 1>(g, s);
    case 1: return launch3<PRO, 2, 2, 1, STAGES, 1>(g, s);
    case 2: return launch3<PRO, 2, 1, 2, STAGES, 1>(g, s);
    case 3: return launch3<PRO, 1, 1, 4, STAGES, 1>(g, s);
    case 4: return launch3<PRO, 1, 1, 2, STAGES, 1>(g, s);
    default: return hipErrorInvalidValue;
  }
}

template <int PRO>
static hipError_t launch1_pro(const GemmArgs& g, hipStream_t s) {
  static int force_cfg = -2, force_stages = 0;
  if (force_cfg == -2) {
    force_cfg = -1;
    if (const char* e = mdt_tuning_env("MDT_TILE1")) sscanf(e, "%d,%d", &force_cfg, &force_stages);
  }
  auto tiles = [&](int bm, int bn) {
    return (int64_t)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn) * (g.phases > 1 ? g.phases : 1);
  };
  int cfg;
  if (g.N > 64 && tiles(128, 128) >= 256) cfg = g.cin % 64 == 0 ? 0 : 1;
  else if (g.cin % 64 == 0 && tiles(128, 64) >= 256) cfg = 2;
  else if (g.cin % 128 == 0) cfg = 3;
  else if (g.cin % 64 == 0) cfg = 4;
  else cfg = 1;
  if (force_cfg >= 0) {
    const int need = force_cfg == 3 ? 128 : (force_cfg == 1 ? 32 : 64);
    if (g.cin % need == 0) cfg = force_cfg;
  }
  int stages = 2;
  if (force_stages) stages = force_stages;
  return stages == 1 ? launch1_cfg<PRO, 1>(g, s, cfg) : launch1_cfg<PRO, 2>(g, s, cfg);
}

hipError_t launch_gemm_bf16(const GemmArgs& g, hipStream_t s) {
  if (g.M <= 0) return hipSuccess;
  if (g.cin % 32) return hipErrorInvalidValue;
  switch (g.pro) {
    case 0: return launch1_pro<0>(g, s);
    case 1: return launch1_pro<1>(g, s);
    case 2: return launch1_pro<2>(g, s);
    case 3: return launch1_pro<3>(g, s);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_gemm_bf16x3(const GemmArgs& g, hipStream_t s) {
  if (g.M <= 0) return hipSuccess;
  if (g.cin % 32) return hipErrorInvalidValue;   // the compiler packs fp32 weights (k_gemm.hip) for such layers
  switch (g.pro) {
    case 0: return launch3_pro<0>(g, s);
    case 1: return launch3_pro<1>(g, s);
    case 2: return launch3_pro<2>(g, s);
    case 3: return launch3_pro<3>(g, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace mdt
