// Plain-bf16 GEMM of the deep-UNet configuration (BASELINE configs[4]: channels 256 -> levels of 512 / 1024 channels, "bf16"):
//   out[M][N] (fp32) = A16[M][K] (bf16) x W16[N][K]^T (bf16)  (+ bias, exact-erf GELU, + residual),   fp32 accumulation,
// with the Conv1d taps of the GEMM operator (k_gemm.hip) kept as row offsets inside a sample (zero padding = a zero source).
//
// Why a second kernel next to k_gemm3<NPROD = 1>: there the A operand is fp32 and normalised / activated while it is staged,
// once per column tile and per tap (GroupNorm + FiLM + SiLU of a 512-channel row block 3 taps x 4 column tiles = 12 times),
// and the staging goes through registers -- the launch is bound by that VALU work, not by the MFMAs (199 TFLOP/s at
// B = 512).  Here the prologue runs ONCE per element in a separate pass (k_prep16: fp32 -> prologue -> bf16, HBM-bound) and
// the GEMM streams both operands with LDS-DMA:
//   * tile BM x BN x 64, waves as WM x WN, each wave (32 TM) x (32 TN) outputs in v_mfma_f32_32x32x16_bf16 accumulators;
//   * both operand tiles are [rows][64] bf16 = 128-byte rows, filled by global_load_lds_dwordx4 (8 rows per instruction),
//     two LDS stages: the DMA of chunk k+1 runs under the MFMAs of chunk k, one barrier per chunk;
//   * the 16-byte slots of a row are XOR-swizzled with (row >> 1) & 7 THROUGH THE SOURCE ADDRESS (the LDS image of a DMA is
//     lane-linear), which makes the ds_read_b128 fragment reads of 16 consecutive rows hit 16 different bank groups;
//   * 256 x 256 tiles when they fill the chip (128 flop per staged byte), else 256 x 128 / 128 x 128;
//   * epilogue straight from the accumulators: for a fixed register the 32 lanes of a half-wave hold 32 consecutive
//     columns of one row (128-byte segments).
#include <cstdio>
#include <cstdlib>

#include "mdt_kernels.h"

namespace mdt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};   // source of padded rows

__device__ __forceinline__ float gelu16(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(64 * WM * WN) void k_gemm_b16(Gemm16Args g) {
  constexpr int NW = WM * WN, BM = 32 * TM * WM, BN = 32 * TN * WN, BK = 64;
  constexpr int NA = BM / 8 / NW, NB = BN / 8 / NW;          // DMA instructions per wave and tile (8 rows each)
  constexpr int ABYTES = BM * 128, STAGE = (BM + BN) * 128;
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split over the waves");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt = (g.N + BN - 1) / BN;
  int id;
  {   // workgroups that share an XCD (bid % 8) take consecutive tiles: the A row block stays in that XCD's L2
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int m0 = (id / nt) * BM, n0 = (id % nt) * BN;
  const int K = g.taps * g.cin;
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(g_zero16);

  // ---- per-lane DMA sources (lane l of an instruction: LDS row R0 + l / 8, physical slot l % 8) ----
  // (fixed bounds: hipcc's host pass silently drops the kernel stub when a lambda captures an array of dependent size)
  static_assert(NA <= 4 && NB <= 4, "at most 4 DMA instructions per wave and operand");
  const unsigned char* abase[4];
  int arow[4];                   // row inside the sample, or a large negative number for rows past M
  const unsigned char* wsrc[4];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int R = (j * NW + wave) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((R >> 1) & 7);
    const int m = m0 + R;
    const bool ok = m < g.M;
    const int b = ok ? m / g.rows : 0;
    const int rr = ok ? m - b * g.rows : 0;
    arow[j] = ok ? rr : -(1 << 28);
    abase[j] = reinterpret_cast<const unsigned char*>(g.A) + (((int64_t)b * g.rows + rr) * g.lda + g.a_col + 8 * c) * 2;
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int R = (j * NW + wave) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((R >> 1) & 7);
    const int n = min(n0 + R, g.N - 1);
    wsrc[j] = reinterpret_cast<const unsigned char*>(g.W) + ((int64_t)n * K + 8 * c) * 2;
  }

  auto issue = [&](int kc, int stage) {
    const int k0 = kc * BK;
    const int tap = k0 / g.cin;
    const int ci = k0 - tap * g.cin;
    const int delta = tap * g.t_dj + g.t_off;
    const int64_t aoff = ((int64_t)delta * g.lda + ci) * 2;
    unsigned char* sa = smem + stage * STAGE + wave * 1024;
    unsigned char* sb = sa + ABYTES;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const bool ok = (unsigned)(arow[j] + delta) < (unsigned)g.rows;
      const unsigned char* src = ok ? abase[j] + aoff : zero;
      __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(sa + j * NW * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j)
      __builtin_amdgcn_global_load_lds(wsrc[j] + (int64_t)k0 * 2, (__attribute__((address_space(3))) void*)(sb + j * NW * 1024), 16, 0, 0);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  const int li = lane & 31, lh = lane >> 5;
  const int wr = wave / WN, wc = wave % WN;
  const int swz = (li >> 1) & 7;                  // (row >> 1) & 7: tile rows start at multiples of 32
  const int aoffs = (wr * 32 * TM + li) * 128, boffs = ABYTES + (wc * 32 * TN + li) * 128;

  auto compute = [&](int stage) {
    const unsigned char* cur = smem + stage * STAGE;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int slot = ((2 * ks + lh) ^ swz) << 4;
      bf16x8 af[TM], bf[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a) af[a] = *reinterpret_cast<const bf16x8*>(cur + aoffs + a * 4096 + slot);
#pragma unroll
      for (int b = 0; b < TN; ++b) bf[b] = *reinterpret_cast<const bf16x8*>(cur + boffs + b * 4096 + slot);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
  };

  const int nk = K / BK;
  issue(0, 0);
  for (int kc = 0; kc < nk; ++kc) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();          // chunk kc has landed for every wave; every wave is done reading the other stage
    if (kc + 1 < nk) issue(kc + 1, (kc + 1) & 1);
    compute(kc & 1);
  }

  // ---- epilogue.  32x32 C/D layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) ----
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int col = n0 + wc * 32 * TN + b * 32 + li;
    const bool cok = col < g.N;
    const float bias = (g.bias && cok) ? g.bias[col] : 0.f;
    // Register pairs (r, r + 1) are two consecutive rows of one column.  fp32 output: plain stores.  bf16 output (out16: the
    // only reader is another bf16 x bf16 GEMM; ldc in bf16 elements) or bf16 COPY of the fp32 output (copy16: the residual
    // stream as the next GEMM's A operand): neighbouring lanes swap one value of each pair, so that a lane stores two
    // consecutive columns of ONE row as 4 bytes.
    unsigned short* o16 = g.out16 ? reinterpret_cast<unsigned short*>(g.out) : g.copy16;
    const int ld16 = g.out16 ? g.ldc : g.N, oc16 = g.out16 ? g.o_col : 0;
    const bool odd = li & 1;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const int mb = m0 + wr * 32 * TM + a * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
        const bool ok0 = cok && mb < g.M, ok1 = cok && mb + 1 < g.M;
        float v0 = acc[a][b][r] + bias, v1 = acc[a][b][r + 1] + bias;
        if (g.act == 1) { v0 = gelu16(v0); v1 = gelu16(v1); }
        if (g.res) {
          if (ok0) v0 += g.res[(int64_t)mb * g.ldr + col];
          if (ok1) v1 += g.res[(int64_t)(mb + 1) * g.ldr + col];
        }
        if (!g.out16) {
          if (ok0) g.out[(int64_t)mb * g.ldc + g.o_col + col] = v0;
          if (ok1) g.out[(int64_t)(mb + 1) * g.ldc + g.o_col + col] = v1;
        }
        if (o16) {
          const float recv = __shfl_xor(odd ? v0 : v1, 1, 64);
          const float lo = odd ? recv : v0, hi = odd ? v1 : recv;
          const int m = mb + (odd ? 1 : 0);
          if (cok && m < g.M) {
            const unsigned pk = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)lo) |
                                ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)hi) << 16);
            *reinterpret_cast<unsigned*>(o16 + (int64_t)m * ld16 + oc16 + (col & ~1)) = pk;
          }
        }
      }
  }
}

template <int WM, int WN, int TM, int TN>
static hipError_t launch16(const Gemm16Args& g, hipStream_t s) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  const int mt = (g.M + BM - 1) / BM, nt = (g.N + BN - 1) / BN;
  const size_t smem = 2 * (size_t)(BM + BN) * 128;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_b16<WM, WN, TM, TN>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  hipLaunchKernelGGL((k_gemm_b16<WM, WN, TM, TN>), dim3((unsigned)(mt * nt)), dim3(64 * WM * WN), smem, s, g);
  return hipGetLastError();
}

bool gemm_b16_supported(int cin, int taps, int lda, int a_col) {
  return cin > 0 && cin % 64 == 0 && taps >= 1 && lda % 8 == 0 && a_col % 8 == 0;
}

// MDT_TILE16 = 0 (256 x 256), 1 (256 x 128), 2 (128 x 128) forces a tile (tuning aid, read once); mdt_set_tuning("tile16", v)
// does the same at run time (tests walk the configurations in one process; -1 = automatic)
static int g_force_tile16 = -1;
void set_tile16(int v) { g_force_tile16 = v; }
hipError_t launch_gemm_b16(const Gemm16Args& g, hipStream_t s) {
  if (g.M <= 0) return hipSuccess;
  if (!gemm_b16_supported(g.cin, g.taps, g.lda, g.a_col)) return hipErrorInvalidValue;
  static int force0 = -2;
  if (force0 == -2) {
    force0 = -1;
    if (const char* e = getenv("MDT_TILE16")) force0 = atoi(e);
  }
  const int force = g_force_tile16 >= 0 ? g_force_tile16 : force0;
  auto tiles = [&](int bm, int bn) { return (int64_t)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn); };
  int cfg = tiles(256, 256) >= 256 ? 0 : (tiles(256, 128) >= 256 ? 1 : 2);     // the largest tile that still fills 256 CUs
  if (force >= 0) cfg = force;
  switch (cfg) {
    case 0: return launch16<2, 4, 4, 2>(g, s);
    case 1: return launch16<4, 2, 2, 2>(g, s);
    default: return launch16<2, 2, 2, 2>(g, s);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// k_prep16: the A operand of k_gemm_b16.  out16[row][c] = bf16(prologue(a[row][a_col + c])), c < cin, with the prologues of
// the GEMM operator (mdt_hip.h: LayerNorm over the row / GroupNorm from precomputed statistics + FiLM [+ SiLU] / SiLU / none).
// 16 lanes per row, 8 channels per lane and step: 512 contiguous bytes in, 256 out per row pass.
// ------------------------------------------------------------------------------------------------------------------
template <int PRO>
__global__ __launch_bounds__(256) void k_prep16(Prep16Args g) {
  const int row = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
  if (row >= g.total_rows) return;
  const int b = row / g.rows;
  const float* src = g.a + (int64_t)row * g.lda + g.a_col;
  unsigned short* dst = g.out + (int64_t)row * g.cin;
  float mean = 0.f, rstd = 1.f;
  if constexpr (PRO == 1) {
    float s = 0.f;
    for (int e = sub; e < g.cin / 4; e += 16) {
      const float4 v = reinterpret_cast<const float4*>(src)[e];
      s += (v.x + v.y) + (v.z + v.w);
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) s += __shfl_xor(s, off, 16);
    mean = s / (float)g.cin;
    float ss = 0.f;
    for (int e = sub; e < g.cin / 4; e += 16) {
      const float4 v = reinterpret_cast<const float4*>(src)[e];
      const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
      ss += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 16);
    rstd = 1.0f / sqrtf(ss / (float)g.cin + g.eps);
  }
  for (int e = sub; e < g.cin / 8; e += 16) {
    const int c0 = 8 * e;
    const float4 v0 = *reinterpret_cast<const float4*>(src + c0), v1 = *reinterpret_cast<const float4*>(src + c0 + 4);
    float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    if constexpr (PRO == 1 || PRO == 2) {
      const float4 g0 = *reinterpret_cast<const float4*>(g.p0 + c0), g1 = *reinterpret_cast<const float4*>(g.p0 + c0 + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(g.p1 + c0), b1 = *reinterpret_cast<const float4*>(g.p1 + c0 + 4);
      const float ga[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const float be[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      if constexpr (PRO == 1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = (x[k] - mean) * rstd * ga[k] + be[k];
      } else {
        const float4 s0 = *reinterpret_cast<const float4*>(g.p3 + c0), s1 = *reinterpret_cast<const float4*>(g.p3 + c0 + 4);
        const float4 h0 = *reinterpret_cast<const float4*>(g.p3 + g.cin + c0), h1 = *reinterpret_cast<const float4*>(g.p3 + g.cin + c0 + 4);
        const float fa[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        const float fs[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
        const float* st = g.p2 + (int64_t)b * g.groups * 2;
        float mu[8], rs[8];
        if (g.gsize % 8 == 0) {           // the 8 channels of a lane share a group (all layers of the network)
          const int grp = min(c0 / g.gsize, g.groups - 1);
          const float2 mr = *reinterpret_cast<const float2*>(st + grp * 2);
#pragma unroll
          for (int k = 0; k < 8; ++k) { mu[k] = mr.x; rs[k] = mr.y; }
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int grp = min((c0 + k) / g.gsize, g.groups - 1);
            mu[k] = st[grp * 2]; rs[k] = st[grp * 2 + 1];
          }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float sc = rs[k] * ga[k];
          x[k] = x[k] * sc + (be[k] - sc * mu[k]);
          x[k] = x[k] * (fa[k] + 1.0f) + fs[k];
        }
        if (g.pro_silu) {
#pragma unroll
          for (int k = 0; k < 8; ++k) x[k] = x[k] / (1.0f + expf(-x[k]));
        }
      }
    } else if constexpr (PRO == 3) {
#pragma unroll
      for (int k = 0; k < 8; ++k) x[k] = x[k] / (1.0f + expf(-x[k]));
    }
    uint4 o;
    unsigned short h[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) h[k] = __builtin_bit_cast(unsigned short, (__bf16)x[k]);
    o.x = h[0] | ((unsigned)h[1] << 16); o.y = h[2] | ((unsigned)h[3] << 16);
    o.z = h[4] | ((unsigned)h[5] << 16); o.w = h[6] | ((unsigned)h[7] << 16);
    *reinterpret_cast<uint4*>(dst + c0) = o;
  }
}

hipError_t launch_prep16(const Prep16Args& g, hipStream_t s) {
  if (g.total_rows <= 0) return hipSuccess;
  if (g.cin % 8 || g.lda % 4 || g.a_col % 4) return hipErrorInvalidValue;
  const dim3 grid((unsigned)((g.total_rows + 15) / 16)), block(256);
  switch (g.pro) {
    case 0: hipLaunchKernelGGL(k_prep16<0>, grid, block, 0, s, g); break;
    case 1: hipLaunchKernelGGL(k_prep16<1>, grid, block, 0, s, g); break;
    case 2: hipLaunchKernelGGL(k_prep16<2>, grid, block, 0, s, g); break;
    case 3: hipLaunchKernelGGL(k_prep16<3>, grid, block, 0, s, g); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace mdt
