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
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};   // source of padded rows

// exact-erf GELU with the branch-free erf of the ring kernels (Abramowitz-Stegun 7.1.26, |error| < 1.5e-7: far inside this mode's
// bf16 budget): the erff of round 3 made the feed-forward up-projection's epilogue VALU-bound
__device__ __forceinline__ float gelu16(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erfa = 1.0f - poly * __expf(-z * z);
  return 0.5f * x * (1.0f + copysignf(erfa, x));
}

// MDT_UB (tools/ubench/gemm16_phases.hip ONLY; the library never defines it): phase ablations of the main loop for timing -- bit 0: only
// chunk 0 is requested (no DMA stream), bit 1: no MFMAs / fragment reads, bit 2: no epilogue; results are WRONG with any bit set.
#ifndef MDT_UB
#define MDT_UB 0
#endif
#if MDT_UB != 0 || defined(MDT_UB_CLOCK)
__device__ unsigned long long g_ub_clk[4];          // shader clock / 100 MHz wall clock at the start and the end of workgroup 0
#endif
#ifndef MDT_B16_EP8_PAD
#define MDT_B16_EP8_PAD 4         // W16 epilogue: parked-row pitch 32 TN + 4 floats (two rows' 8-column pieces interleave over the banks)
#endif
#ifndef MDT_B16_EPI_GROUPS
#define MDT_B16_EPI_GROUPS 8      // (4: the residual of half a 32-row block in flight per wave -- the epilogue was bound by bytes in flight)
#endif

// W16 (round 6): the all-bf16 epilogue -- bf16 output, bf16 residual or none, no second copy.  A lane owns 8 columns of a row
// (16-byte accesses) and the residual of the tile is REQUESTED UNDER THE MAIN LOOP: the generic epilogue asks for a block's
// residual, waits a round trip, stores, and does that TM times per wave -- with 8 bytes per lane in flight it was bound by
// latency, not bytes (K = 512, M = 65,536: 82-86 us with a residual, bf16 or fp32, against 48-50 us without).
template <int WM, int WN, int TM, int TN, bool W16>
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
  // Round 6: a chunk's sources are a wave-UNIFORM base (tap, channel offset) plus per-lane 32-bit byte offsets that never
  // change, so a DMA instruction takes the scalar-base form (global_load_lds_dwordx4 v_off, s[base]) and a chunk costs 16 DMA
  // instructions + scalar arithmetic.  Before, every instruction had its own 64-bit VALU add and a zero-source select, and the
  // chunk's tap came from an integer division: ~100 instructions between the barrier and the chunk's first MFMA, on both waves
  // of a SIMD at the same time (parked 42 % + issue-stalled 37 % of the wave cycles, profiles/r6_cfg4_pmc_sq.csv).
  // (fixed bounds: hipcc's host pass silently drops the kernel stub when a lambda captures an array of dependent size)
  static_assert(NA <= 8 && NB <= 8, "at most 8 DMA instructions per wave and operand");
  unsigned aoff[8], woff[8];     // byte offsets from g.A / g.W (launch_gemm_b16 checks that they fit 32 bits)
  int arow[8];                   // row inside the sample, or a large negative number for rows past M
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int R = (j * NW + wave) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((R >> 1) & 7);
    const int m = m0 + R;
    const bool ok = m < g.M;
    const int b = ok ? m / g.rows : 0;
    const int rr = ok ? m - b * g.rows : 0;
    arow[j] = ok ? rr : -(1 << 28);
    aoff[j] = (unsigned)((((int64_t)b * g.rows + rr) * g.lda + g.a_col + 8 * c) * 2);
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int R = (j * NW + wave) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((R >> 1) & 7);
    const int n = min(n0 + R, g.N - 1);
    woff[j] = (unsigned)(((int64_t)n * K + 8 * c) * 2);
  }
  // no padded source anywhere: one tap on its own row and whole row tiles
  const bool plain = g.taps == 1 && g.t_off == 0 && g.M % BM == 0;
  int nx_tap = 0, nx_ci = 0, nx_k0 = 0;          // the next chunk to request (chunks are requested in order)

  auto issue = [&](int stage) {
    const int delta = nx_tap * g.t_dj + g.t_off;
    const unsigned char* ab = reinterpret_cast<const unsigned char*>(g.A) + ((int64_t)delta * g.lda + nx_ci) * 2;     // wave-uniform
    const unsigned char* wb = reinterpret_cast<const unsigned char*>(g.W) + (int64_t)nx_k0 * 2;
    unsigned char* sa = smem + stage * STAGE + wave * 1024;
    unsigned char* sb = sa + ABYTES;
    // (the offsets pass through an empty asm so that their zero-extension stays in this basic block: k_res256.hip, issue_w)
    unsigned ao[8], wo[8];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      ao[j] = aoff[j];
      asm volatile("" : "+v"(ao[j]));
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      wo[j] = woff[j];
      asm volatile("" : "+v"(wo[j]));
    }
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const unsigned char* src = ab + ao[j];
      // a padded row (outside its sample for this tap, or past M) reads the zero line; ONE request per piece either way (the
      // hand-counted vmcnt of the W16 epilogue counts them)
      if (!plain && !((unsigned)(arow[j] + delta) < (unsigned)g.rows)) src = zero;
      __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(sa + j * NW * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j)
      __builtin_amdgcn_global_load_lds(wb + wo[j], (__attribute__((address_space(3))) void*)(sb + j * NW * 1024), 16, 0, 0);
    nx_k0 += BK;
    nx_ci += BK;
    if (nx_ci == g.cin) {
      nx_ci = 0;
      ++nx_tap;
    }
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

  // (Round 6, measured before leaving this loop alone -- tools/ubench/gemm16_phases.hip, profiles/r6_ubench_gemm16_phases.txt: the
  //  MFMAs alone take 2,400-2,500 cycles per 256 x 256 x 64 chunk at the 1.75-1.95 GHz the chip holds under them, the DMA stream
  //  alone 2,350 (1,500 with two chunks in flight), together 3,400-3,500 -- and that figure does not move with hand
  //  double-buffered fragments, a third fewer fragment reads, two chunks in flight, or the requests spread between the MFMAs.)
  const bool fold = W16 && g.csum != nullptr;         // LayerNorm of the A rows folded into the GEMM (below, "W16")
  float st_s[4] = {0.f, 0.f, 0.f, 0.f}, st_q[4] = {0.f, 0.f, 0.f, 0.f};      // this lane's row of each row block: sum, sum of squares
  static_assert(TM <= 4, "statistics registers");
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
      if (fold && (ks % WN) == wc) {                   // the WN waves of a row group share the k-steps of the row statistics
        const bf16x2 one2 = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bf16x2 pr = {af[a][2 * e], af[a][2 * e + 1]};
            st_s[a] = __builtin_amdgcn_fdot2_f32_bf16(pr, one2, st_s[a], false);
            st_q[a] = __builtin_amdgcn_fdot2_f32_bf16(pr, pr, st_q[a], false);
          }
      }
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
  };

  // ---- W16: the wave's residual, 8 columns (16 bytes) per lane, LPR8 lanes per row, NP8 requests per 32-row block.  Blocks
  // 0 and 1 are requested two chunks before the end of the main loop (32 registers), blocks 2 and 3 when the loop is over (the
  // fragment and DMA-pointer registers are dead by then).
  // The requests are unconditional on clamped addresses: their COUNT is what the hand-written vmcnt below relies on.
  constexpr int LPR8 = 4 * TN, RP8 = 64 / LPR8, NP8 = 32 / RP8;
  static_assert(!W16 || (TM <= 4 && NP8 <= 4), "residual registers");
  const int er8 = lane / LPR8, ec8 = (lane % LPR8) * 8;
  const bool res8 = W16 && g.res != nullptr;
  uint4 rq[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int p = 0; p < 4; ++p) rq[a][p] = make_uint4(0u, 0u, 0u, 0u);
  auto prefetch = [&](int a) {
    const int n = min(n0 + wc * 32 * TN + ec8, g.N - 8);
#pragma unroll
    for (int p = 0; p < (NP8 < 4 ? NP8 : 4); ++p) {
      const int m = min(m0 + wr * 32 * TM + a * 32 + p * RP8 + er8, g.M - 1);
      rq[a][p] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(g.res) + (int64_t)m * g.ldr + n);
    }
  };
  constexpr int NPRE = (TM < 2 ? TM : 2) * NP8;     // requests in flight behind the last chunk's DMA
  float bia8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (W16 && g.bias) {                              // requested before the first chunk: no round trip in the epilogue
    const int n = min(n0 + wc * 32 * TN + ec8, g.N - 8);
    const float4 b0 = *reinterpret_cast<const float4*>(g.bias + n), b1 = *reinterpret_cast<const float4*>(g.bias + n + 4);
    bia8[0] = b0.x; bia8[1] = b0.y; bia8[2] = b0.z; bia8[3] = b0.w; bia8[4] = b1.x; bia8[5] = b1.y; bia8[6] = b1.z; bia8[7] = b1.w;
  }
  // LayerNorm folded into the GEMM (W16 only).  The A operand is the RAW bf16 residual stream and W (g xn + b) = rstd ((W g) x
  // - mean rowsum(W g)) + W b: the epilogue scales and shifts the accumulators per row.  The rows' (sum, sum of squares) come from
  // the A fragments the waves read for the MFMAs anyway -- wave column wc takes the k-steps ks = wc mod WN, two v_dot2c_f32_bf16
  // per pair of values, in the MFMAs' shadow -- and meet in LDS behind the main loop.  What it replaces: an MDT_OP_PREP16 launch
  // that read the stream and wrote a normalised copy for this GEMM to read again (134 MB of 201 at M = 65,536, C = 512).
  float cs8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float2* srow = reinterpret_cast<float2*>(smem + 2 * STAGE);          // (mean, rstd) of the tile's BM rows, behind the two stages
  float2* sred = srow + BM;                                            // [WN][BM] partial (sum, sum of squares)
  if (fold) {
    const int n = min(n0 + wc * 32 * TN + ec8, g.N - 8);
    const float4 c0 = *reinterpret_cast<const float4*>(g.csum + n), c1 = *reinterpret_cast<const float4*>(g.csum + n + 4);
    cs8[0] = c0.x; cs8[1] = c0.y; cs8[2] = c0.z; cs8[3] = c0.w; cs8[4] = c1.x; cs8[5] = c1.y; cs8[6] = c1.z; cs8[7] = c1.w;
  }

  const int nk = K / BK;
#if MDT_UB != 0 || defined(MDT_UB_CLOCK)
  if (blockIdx.x == 0 && tid == 0) {
    unsigned long long c0, r0;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
    g_ub_clk[0] = c0;
    g_ub_clk[2] = r0;
  }
#endif
  issue(0);
  if (res8 && nk == 1) {
    prefetch(0);
    if (TM > 1) prefetch(1);
  }
  for (int kc = 0; kc < nk; ++kc) {
    // (a bare s_barrier behind hand-written counters: __syncthreads() makes the compiler wait for vmcnt(0), residual included)
    if (res8 && kc == nk - 1) {
      // loads retire in order: the last chunk's DMA was issued before the NPRE residual requests
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NPRE) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();          // chunk kc has landed for every wave; every wave is done reading the other stage
    asm volatile("" ::: "memory");
    if (kc + 1 < nk && !(MDT_UB & 1)) issue((kc + 1) & 1);
    if (res8 && kc == nk - 2) {
      prefetch(0);
      if (TM > 1) prefetch(1);
    }
    if (!(MDT_UB & 2)) compute(kc & 1);
  }
#if MDT_UB != 0 || defined(MDT_UB_CLOCK)
  if (blockIdx.x == 0 && tid == 0) {
    unsigned long long c1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
    g_ub_clk[1] = c1;
    g_ub_clk[3] = r1;
  }
  if ((MDT_UB & 4) && g.act != 77) return;           // (the accumulators stay live: act is never 77)
#endif

  // ---- epilogue.  32x32 C/D layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5): for a fixed register a
  // half-wave holds 32 consecutive columns of ONE row.  Written straight from the accumulators that is 128-byte segments, one
  // row per instruction, and -- what hurt -- the residual comes back the same way: 4-byte loads, one round trip per register pair
  // (round 3: the GEMMs with a residual ran at 160-220 TFLOP/s against 480-600 for the same shapes without, 14 % of a configs[4]
  // evaluation).  Now every wave parks one 32 x (32 TN) block of its accumulators in its own corner of the (idle) staging LDS as
  // fp32 rows of pitch 72 floats (both the accumulator-order writes and the row-order reads are conflict-free) and walks it row
  // by row: 16 lanes per row, float4 each -- bias, GELU, residual (float4 loads, all requested before the first store of the
  // pass) and the fp32 store and / or the bf16 store as 16- / 8-byte accesses of 256- / 128-byte row segments.
  if (fold) {                                       // the rows' statistics meet: halves of k (lh), then the WN waves of the row group
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const float s_ = st_s[a] + __shfl_xor(st_s[a], 32), q_ = st_q[a] + __shfl_xor(st_q[a], 32);
      if (lh == 0) sred[wc * BM + wr * 32 * TM + a * 32 + li] = make_float2(s_, q_);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                     // every wave is done reading the last stage (no DMA is in flight any more)
  asm volatile("" ::: "memory");
  if (fold) {
    for (int r = tid; r < BM; r += 64 * NW) {
      float s_ = 0.f, q_ = 0.f;
#pragma unroll
      for (int w = 0; w < WN; ++w) {
        const float2 v = sred[w * BM + r];
        s_ += v.x;
        q_ += v.y;
      }
      const float mean = s_ / (float)g.cin;
      const float var = fmaxf(q_ / (float)g.cin - mean * mean, 0.f);
      srow[r] = make_float2(mean, 1.0f / sqrtf(var + g.eps));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  constexpr int EP = 32 * TN + (W16 ? MDT_B16_EP8_PAD : 8);   // floats per parked row (72 / 136: pitch = 8 mod 32 banks)
  float* ws = reinterpret_cast<float*>(smem) + wave * (32 * EP);
  static_assert(NW * 32 * EP * 4 <= 2 * STAGE, "parking area inside the staging buffers");
  if constexpr (W16) {
    unsigned short* o8 = reinterpret_cast<unsigned short*>(g.out);
    const int ncol = n0 + wc * 32 * TN + ec8;
    const bool cok = ncol < g.N;                    // (N % 8 == 0, gemm_b16_w16_ok)
    {
      // The compiler guards the first LDS read after an LDS-DMA stream with vmcnt(0) (the DMA's LDS writes may alias it).  Take
      // that wait HERE, where only the requests of blocks 0 / 1 are in flight (two chunks old), and request the other blocks
      // behind it: they land under the first blocks' passes, and every later wait is counted exactly.
      const float first = *reinterpret_cast<volatile float*>(ws);
      asm volatile("" ::"v"(first) : "memory");
      if (res8) {
        if (TM > 2) prefetch(2);
        if (TM > 3) prefetch(3);
      }
    }
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) ws[((r & 3) + 8 * (r >> 2) + 4 * lh) * EP + b * 32 + li] = acc[a][b][r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int mrow0 = m0 + wr * 32 * TM + a * 32;
#pragma unroll
      for (int p = 0; p < NP8; ++p) {
        const int row = p * RP8 + er8, m = mrow0 + row;
        const float4 v0 = *reinterpret_cast<const float4*>(ws + row * EP + ec8);
        const float4 v1 = *reinterpret_cast<const float4*>(ws + row * EP + ec8 + 4);
        float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        if (fold) {
          const float2 mr = srow[wr * 32 * TM + a * 32 + row];
#pragma unroll
          for (int k = 0; k < 8; ++k) x[k] = mr.y * (x[k] - mr.x * cs8[k]);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] += bia8[k];
        if (g.act == 1) {
#pragma unroll
          for (int k = 0; k < 8; ++k) x[k] = gelu16(x[k]);
        }
        const unsigned w[4] = {rq[a][p].x, rq[a][p].y, rq[a][p].z, rq[a][p].w};       // zeros without a residual
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          x[2 * q] += __uint_as_float(w[q] << 16);
          x[2 * q + 1] += __uint_as_float(w[q] & 0xffff0000u);
        }
        unsigned short h[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) h[k] = __builtin_bit_cast(unsigned short, (__bf16)x[k]);
        const bool okrow = cok && m < g.M;
        if (okrow) {
          *reinterpret_cast<uint4*>(o8 + (int64_t)m * g.ldc + g.o_col + ncol) =
              make_uint4(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16), h[4] | ((unsigned)h[5] << 16), h[6] | ((unsigned)h[7] << 16));
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the block has been read before the next one is parked
    }
    return;
  }
  unsigned short* o16 = g.out16 ? reinterpret_cast<unsigned short*>(g.out) : g.copy16;
  const int ld16 = g.out16 ? g.ldc : g.N, oc16 = g.out16 ? g.o_col : 0;
  constexpr int LPR = 8 * TN;                       // lanes per parked row (float4 each)
  const int er = lane / LPR, ec = (lane % LPR) * 4; // this lane's row inside a group of 64 / LPR rows, its first column
#pragma unroll
  for (int a = 0; a < TM; ++a) {
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) ws[((r & 3) + 8 * (r >> 2) + 4 * lh) * EP + b * 32 + li] = acc[a][b][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // same wave writes and reads: no barrier needed
    constexpr int CW = 32 * TN;                                   // columns of the block (64)
    constexpr int RPP = 64 / (CW / 4);                            // rows per pass of the wave (4)
    const int mrow0 = m0 + wr * 32 * TM + a * 32, ncol = n0 + wc * CW + ec;
    const bool cok = ec < CW && ncol < g.N;                       // (N % 4 == 0, gemm_b16_epilogue_ok: a float4 never straddles it)
    float4 bia = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.bias && cok) bia = *reinterpret_cast<const float4*>(g.bias + ncol);
    constexpr int NB = MDT_B16_EPI_GROUPS;                          // row groups per batch: their residual loads fly together
#pragma unroll
    for (int p0 = 0; p0 < 32; p0 += NB * RPP) {
      float4 v[NB], rs[NB];
      bool ok[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int row = p0 + u * RPP + er, m = mrow0 + row;
        ok[u] = cok && m < g.M;
        v[u] = *reinterpret_cast<const float4*>(ws + row * EP + ec);
        rs[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.res && ok[u]) {
          if (g.res16) {                                          // bf16 residual stream: 8 bytes per lane, widened exactly
            const uint2 rb = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(g.res) + (int64_t)m * g.ldr + ncol);
            rs[u] = make_float4(__uint_as_float(rb.x << 16), __uint_as_float(rb.x & 0xffff0000u),
                                __uint_as_float(rb.y << 16), __uint_as_float(rb.y & 0xffff0000u));
          } else {
            rs[u] = *reinterpret_cast<const float4*>(g.res + (int64_t)m * g.ldr + ncol);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int m = mrow0 + p0 + u * RPP + er;
        float x[4] = {v[u].x + bia.x, v[u].y + bia.y, v[u].z + bia.z, v[u].w + bia.w};
        if (g.act == 1) {
#pragma unroll
          for (int k = 0; k < 4; ++k) x[k] = gelu16(x[k]);
        }
        x[0] += rs[u].x; x[1] += rs[u].y; x[2] += rs[u].z; x[3] += rs[u].w;
        if (ok[u]) {
          if (!g.out16) *reinterpret_cast<float4*>(g.out + (int64_t)m * g.ldc + g.o_col + ncol) = make_float4(x[0], x[1], x[2], x[3]);
          if (o16) {
            unsigned short h[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) h[k] = __builtin_bit_cast(unsigned short, (__bf16)x[k]);
            *reinterpret_cast<uint2*>(o16 + (int64_t)m * ld16 + oc16 + ncol) =
                make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the block has been read before the next one is parked
  }
}

template <int WM, int WN, int TM, int TN, bool W16>
static hipError_t launch16w(const Gemm16Args& g, hipStream_t s) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  const int mt = (g.M + BM - 1) / BM, nt = (g.N + BN - 1) / BN;
  const size_t smem = 2 * (size_t)(BM + BN) * 128 + (W16 ? (1 + WN) * BM * sizeof(float2) : 0);   // + a folded LayerNorm's row statistics
  static DevOnce attr_once;                          // per device (mdt_kernels.h)
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_b16<WM, WN, TM, TN, W16>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  hipLaunchKernelGGL((k_gemm_b16<WM, WN, TM, TN, W16>), dim3((unsigned)(mt * nt)), dim3(64 * WM * WN), smem, s, g);
  return hipGetLastError();
}

// the all-bf16 epilogue (8 columns per lane): bf16 output, no second copy, a bf16 residual or none, every pitch / offset a
// multiple of 8 elements and the tensors 16-byte aligned; anything else takes the generic epilogue.  MDT_W16=0 (tuning aid) /
// mdt_set_tuning("w16", 0) turn it off.
static int g_w16 = -1;
void set_w16(int v) { g_w16 = v; }
static bool gemm_b16_w16_ok(const Gemm16Args& g) {
  auto al16 = [](const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; };
  if (g_w16 < 0) {
    const char* e = mdt_tuning_env("MDT_W16");
    g_w16 = e ? atoi(e) : 1;
  }
  if ((!g_w16 && !g.csum) || !g.out16 || g.copy16 || (g.res && !g.res16)) return false;   // (the folded LayerNorm lives in this epilogue only)
  if (g.N % 8 || g.ldc % 8 || g.o_col % 8 || !al16(g.out)) return false;
  if (g.res && (g.ldr % 8 || !al16(g.res))) return false;
  return true;
}

template <int WM, int WN, int TM, int TN>
static hipError_t launch16(const Gemm16Args& g, hipStream_t s) {
  if (gemm_b16_w16_ok(g)) return launch16w<WM, WN, TM, TN, true>(g, s);
  if (g.csum) return hipErrorInvalidValue;          // a folded LayerNorm must not be dropped silently
  return launch16w<WM, WN, TM, TN, false>(g, s);
}

bool gemm_b16_supported(int cin, int taps, int lda, int a_col) {
  return cin > 0 && cin % 64 == 0 && taps >= 1 && lda % 8 == 0 && a_col % 8 == 0;
}

// The epilogue walks the parked accumulators 4 columns per lane: float4 bias / residual loads, float4 (fp32) or 8-byte (bf16)
// stores at column n0 + 4 k.  So N, the output pitch and column offset and the residual pitch must be multiples of 4, the
// tensors 16-byte aligned (8 for a bf16 output / copy) -- otherwise the accesses are misaligned and a float4 reaches past column N.
bool gemm_b16_epilogue_ok(const Gemm16Args& g) {
  auto al = [](const void* p, size_t a) { return (reinterpret_cast<size_t>(p) & (a - 1)) == 0; };
  if (g.N <= 0 || g.N % 4 || g.ldc % 4 || g.o_col % 4) return false;
  if (g.res && (g.ldr % 4 || !al(g.res, g.res16 ? 8 : 16))) return false;
  if (g.bias && !al(g.bias, 16)) return false;
  if (!al(g.out, g.out16 ? 8 : 16) || (g.copy16 && !al(g.copy16, 8))) return false;
  return true;
}

// MDT_TILE16 = 0 (256 x 256), 1 (256 x 128), 2 (128 x 128) forces a tile (tuning aid, read once); mdt_set_tuning("tile16", v)
// does the same at run time (tests walk the configurations in one process; -1 = automatic)
static int g_force_tile16 = -1;
void set_tile16(int v) { g_force_tile16 = v; }
hipError_t launch_gemm_b16(const Gemm16Args& g, hipStream_t s) {
  if (g.M <= 0) return hipSuccess;
  if (!gemm_b16_supported(g.cin, g.taps, g.lda, g.a_col) || !gemm_b16_epilogue_ok(g)) return hipErrorInvalidValue;
  // folded LayerNorm: one tap over the whole normalised row
  if (g.csum && (g.taps != 1 || g.t_off || (reinterpret_cast<size_t>(g.csum) & 15))) return hipErrorInvalidValue;
  // the kernel addresses both operands with 32-bit byte offsets from their bases (4 GB of bf16 rows: no layer comes near)
  if (((int64_t)g.M * g.lda + g.a_col) * 2 >= (1ll << 32) || (int64_t)g.N * g.taps * g.cin * 2 >= (1ll << 32)) return hipErrorInvalidValue;
  static int force0 = -2;
  if (force0 == -2) {
    force0 = -1;
    if (const char* e = mdt_tuning_env("MDT_TILE16")) force0 = atoi(e);
  }
  const int force = g_force_tile16 >= 0 ? g_force_tile16 : force0;
  auto tiles = [&](int bm, int bn) { return (int64_t)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn); };
  int cfg = tiles(256, 256) >= 256 ? 0 : (tiles(256, 128) >= 256 ? 1 : 2);     // the largest tile that still fills 256 CUs
  if (force >= 0) cfg = force;
  switch (cfg) {
    // (256 x 256 on FOUR waves of 128 x 128 -- launch16<2, 2, 4, 4>, 16 MFMAs per 8 KB of fragment reads, 256 accumulator
    //  registers -- measured 5-30 % SLOWER than the eight-wave form on every configs[4] shape, round 5: one wave per SIMD
    //  leaves nothing to run under its fragment reads)
    case 0: return launch16<2, 4, 4, 2>(g, s);
    case 1: return launch16<4, 2, 2, 2>(g, s);
#ifdef MDT_UB_TILE3
    case 3: return launch16w<2, 2, 4, 4, false>(g, s);       // (ubench only: 256 x 256 on four waves, generic epilogue)
#endif
    default: return launch16<2, 2, 2, 2>(g, s);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// k_prep16: the A operand of k_gemm_b16.  out16[row][c] = bf16(prologue(a[row][a_col + c])), c < cin, with the prologues of
// the GEMM operator (mdt_hip.h: LayerNorm over the row / GroupNorm from precomputed statistics + FiLM [+ SiLU] / SiLU / none).
// 16 lanes per row, 8 channels per lane and step: 512 contiguous bytes in, 256 out per row pass.
// ------------------------------------------------------------------------------------------------------------------
// bf16 input (round 6: the bf16 residual stream of the plain-bf16 mode): LayerNorm of a bf16 row of up to 1024 channels, statistics and
// normalisation in fp32 on the exactly widened values; 16 lanes per row, 8 channels (16 bytes) per lane and step
__global__ __launch_bounds__(256) void k_prep16_ln_in16(Prep16Args g) {
  const int row = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
  if (row >= g.total_rows) return;
  const unsigned short* src = reinterpret_cast<const unsigned short*>(g.a) + (int64_t)row * g.lda + g.a_col;
  unsigned short* dst = g.out + (int64_t)row * g.cin;
  const int ng = g.cin / 8;
  float x[8][8];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int e = sub + 16 * k;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (e < ng) v = *reinterpret_cast<const uint4*>(src + 8 * e);
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      x[k][2 * q] = __uint_as_float(w[q] << 16);
      x[k][2 * q + 1] = __uint_as_float(w[q] & 0xffff0000u);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) s += x[k][q];
  }
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) s += __shfl_xor(s, off, 16);
  const float mean = s / (float)g.cin;
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (sub + 16 * k < ng) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float d = x[k][q] - mean;
        ss += d * d;
      }
    }
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 16);
  const float rstd = 1.0f / sqrtf(ss / (float)g.cin + g.eps);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int e = sub + 16 * k;
    if (e < ng) {
      const int c0 = 8 * e;
      const float4 g0 = *reinterpret_cast<const float4*>(g.p0 + c0), g1 = *reinterpret_cast<const float4*>(g.p0 + c0 + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(g.p1 + c0), b1 = *reinterpret_cast<const float4*>(g.p1 + c0 + 4);
      const float ga[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const float be[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      unsigned short h[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) h[q] = __builtin_bit_cast(unsigned short, (__bf16)((x[k][q] - mean) * rstd * ga[q] + be[q]));
      uint4 o;
      o.x = h[0] | ((unsigned)h[1] << 16); o.y = h[2] | ((unsigned)h[3] << 16);
      o.z = h[4] | ((unsigned)h[5] << 16); o.w = h[6] | ((unsigned)h[7] << 16);
      *reinterpret_cast<uint4*>(dst + c0) = o;
    }
  }
}

template <int PRO>
__global__ __launch_bounds__(256) void k_prep16(Prep16Args g) {
  const int row = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
  if (row >= g.total_rows) return;
  const int b = row / g.rows;
  const float* src = g.a + (int64_t)row * g.lda + g.a_col;
  unsigned short* dst = g.out + (int64_t)row * g.cin;
  float mean = 0.f, rstd = 1.f;
  if constexpr (PRO == 1) {
    if (g.cin <= 1024) {
      // LayerNorm of a row of up to 1024 channels with the row held in registers: ONE pass over memory instead of three
      // (round 3: 2.9-3.1 TB/s for this kernel; the statistics re-read the row from the L2 twice).  Lane `sub` of the row's 16
      // lanes owns the 8-channel groups sub, sub + 16, ...: the same 32-byte pieces it converts and writes below.
      float4 xa[8], xb[8];
      const int ng = g.cin / 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int e = sub + 16 * k;
        const bool in = e < ng;
        xa[k] = in ? *reinterpret_cast<const float4*>(src + 8 * e) : make_float4(0.f, 0.f, 0.f, 0.f);
        xb[k] = in ? *reinterpret_cast<const float4*>(src + 8 * e + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) s += ((xa[k].x + xa[k].y) + (xa[k].z + xa[k].w)) + ((xb[k].x + xb[k].y) + (xb[k].z + xb[k].w));
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) s += __shfl_xor(s, off, 16);
      mean = s / (float)g.cin;
      float ss = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (sub + 16 * k < ng) {
          const float d[8] = {xa[k].x - mean, xa[k].y - mean, xa[k].z - mean, xa[k].w - mean,
                              xb[k].x - mean, xb[k].y - mean, xb[k].z - mean, xb[k].w - mean};
          ss += ((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) + ((d[4] * d[4] + d[5] * d[5]) + (d[6] * d[6] + d[7] * d[7]));
        }
      }
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 16);
      rstd = 1.0f / sqrtf(ss / (float)g.cin + g.eps);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int e = sub + 16 * k;
        if (e < ng) {
          const int c0 = 8 * e;
          const float4 g0 = *reinterpret_cast<const float4*>(g.p0 + c0), g1 = *reinterpret_cast<const float4*>(g.p0 + c0 + 4);
          const float4 b0 = *reinterpret_cast<const float4*>(g.p1 + c0), b1 = *reinterpret_cast<const float4*>(g.p1 + c0 + 4);
          const float x[8] = {(xa[k].x - mean) * rstd * g0.x + b0.x, (xa[k].y - mean) * rstd * g0.y + b0.y,
                              (xa[k].z - mean) * rstd * g0.z + b0.z, (xa[k].w - mean) * rstd * g0.w + b0.w,
                              (xb[k].x - mean) * rstd * g1.x + b1.x, (xb[k].y - mean) * rstd * g1.y + b1.y,
                              (xb[k].z - mean) * rstd * g1.z + b1.z, (xb[k].w - mean) * rstd * g1.w + b1.w};
          unsigned short h[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) h[q] = __builtin_bit_cast(unsigned short, (__bf16)x[q]);
          uint4 o;
          o.x = h[0] | ((unsigned)h[1] << 16); o.y = h[2] | ((unsigned)h[3] << 16);
          o.z = h[4] | ((unsigned)h[5] << 16); o.w = h[6] | ((unsigned)h[7] << 16);
          *reinterpret_cast<uint4*>(dst + c0) = o;
        }
      }
      return;
    }
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
  if (g.in16) {                                        // bf16 input: LayerNorm only (a plain conversion would be a copy), rows <= 1024 channels
    if (g.pro != 1 || g.cin > 1024 || g.lda % 8 || g.a_col % 8 || !g.p0 || !g.p1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_prep16_ln_in16, grid, block, 0, s, g);
    return hipGetLastError();
  }
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
