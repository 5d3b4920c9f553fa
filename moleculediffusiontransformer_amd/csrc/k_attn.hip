// Attention core of AttentionBase.forward (modules.py:350-363): one wave per (sample, head),
//   out[i, :] = softmax_j( (q_i . k_j) * scale ) @ V,          head dim 64, n, m <= 64.
//
// Both contractions run on the matrix cores in exact fp32 (v_mfma_f32_16x16x4_f32), entirely out of
// registers -- no LDS, no cross-lane transposes:
//   * S^T = K Q^T  (A = K rows, B = Q rows).  Lane (j = l&15, kq = l>>4) feeds K[j][16kq + s] at k-step s,
//     i.e. 16 CONTIGUOUS floats of its key row (and likewise for Q); any bijection of the 64 features onto
//     (step, lane-quarter) is valid as long as A and B use the same one.
//   * The 16x16 accumulator holds S^T[j = 4g + r][i = l&15] (g = l>>4, r = register).  Softmax over j for a
//     fixed query i is therefore per-lane over r and over g = lanes l, l^16, l^32, l^48: two shuffles.
//   * O^T = V^T P^T  (A = V^T, B = P^T).  With the k-mapping j = 4*kq + s the B operand of step s for lane
//     quarter kq is exactly that lane's own accumulator register r = s: the probabilities never move.
//   * O^T lands as [d = 16dt + 4g + r][i]: each lane stores 4 consecutive features (16 B) of its query row.
#include <algorithm>

#include "mdt_kernels.h"

namespace mdt {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ---- operand loads.  H = the tensor is bf16 (plain-bf16 mode: q | k | v come out of a bf16 x bf16 GEMM as bf16, MDT_A_IN16;
// the values are widened to fp32 in registers -- exact -- and both contractions stay fp32 MFMAs): half the bytes of the launch,
// which is HBM-bound (3.3 TB/s with fp32 rows at B = 2048, profiles/r5_cfg4_op_profile_b2048.txt).
template <bool H>
__device__ __forceinline__ void ld_row16(const unsigned char* base, int64_t elem, float (&d)[16]) {   // 16 consecutive features
  if constexpr (H) {
    const uint4* p = reinterpret_cast<const uint4*>(base + elem * 2);
    const uint4 u0 = p[0], u1 = p[1];
    const unsigned w[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      d[2 * c] = __builtin_bit_cast(float, w[c] << 16);
      d[2 * c + 1] = __builtin_bit_cast(float, w[c] & 0xffff0000u);
    }
  } else {
    const float4* p = reinterpret_cast<const float4*>(base + elem * 4);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4 t = p[c];
      d[4 * c] = t.x; d[4 * c + 1] = t.y; d[4 * c + 2] = t.z; d[4 * c + 3] = t.w;
    }
  }
}
template <bool H>
__device__ __forceinline__ void ld_row4(const unsigned char* base, int64_t elem, float* d, bool live) {   // 4 consecutive features
  if (!live) { d[0] = d[1] = d[2] = d[3] = 0.f; return; }
  if constexpr (H) {
    const uint2 u = *reinterpret_cast<const uint2*>(base + elem * 2);
    d[0] = __builtin_bit_cast(float, u.x << 16); d[1] = __builtin_bit_cast(float, u.x & 0xffff0000u);
    d[2] = __builtin_bit_cast(float, u.y << 16); d[3] = __builtin_bit_cast(float, u.y & 0xffff0000u);
  } else {
    const float4 t = *reinterpret_cast<const float4*>(base + elem * 4);
    d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
  }
}
// O^T tile dt, row 4 g + r of lane (query lo, quarter g) is feature 16 g + 4 r + dt (the PV contraction leaves the output-row
// order free: with "row i of tile dt = feature 4 i + dt" the A operand of the four tiles at one k-step is ONE 16-byte piece of the
// value row -- round 5; it used to be four 4-byte loads 64 bytes apart): 16 consecutive features per lane.
__device__ __forceinline__ void st_out16(const AttnArgs& a, int b, int h, int i, int g, const f32x4 (&acc)[4], float inv) {
  constexpr int D = 64;
  if (a.out16) {
    unsigned short* o16 = reinterpret_cast<unsigned short*>(a.out) + ((int64_t)b * a.T + i) * a.ldo + h * D + 16 * g;
#pragma unroll
    for (int rp = 0; rp < 2; ++rp) {
      unsigned w[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int r = 2 * rp + (c >> 1), dt = 2 * (c & 1);
        const unsigned short h0 = __builtin_bit_cast(unsigned short, (__bf16)(acc[dt][r] * inv));
        const unsigned short h1 = __builtin_bit_cast(unsigned short, (__bf16)(acc[dt + 1][r] * inv));
        w[c] = h0 | ((unsigned)h1 << 16);
      }
      *reinterpret_cast<uint4*>(o16 + 8 * rp) = make_uint4(w[0], w[1], w[2], w[3]);
    }
  } else {
    float* o = a.out + ((int64_t)b * a.T + i) * a.ldo + h * D + 16 * g;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *reinterpret_cast<float4*>(o + 4 * r) = make_float4(acc[0][r] * inv, acc[1][r] * inv, acc[2][r] * inv, acc[3][r] * inv);
  }
}

template <int KTM, int IN16>   // key tiles held in registers: 1 (Tk <= 16), 2 (Tk <= 32) or 4 (Tk <= 64); IN16: bit 0 = q is bf16, bit 1 = k | v are
__global__ __launch_bounds__(256) void k_attn(AttnArgs a) {
  constexpr int D = 64;
  constexpr bool HQ = IN16 & 1, HK = (IN16 & 2) != 0;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= a.batch * a.heads) return;
  const int b = wid / a.heads, h = wid % a.heads;
  const int lane = threadIdx.x & 63;
  const int lo = lane & 15, g = lane >> 4;
  const unsigned char* qb = reinterpret_cast<const unsigned char*>(a.q);
  const unsigned char* kb = reinterpret_cast<const unsigned char*>(a.k);
  const int64_t q0 = (int64_t)b * a.T * a.ldq + h * D, k0e = (int64_t)b * a.kv_bstride * a.ldkv + h * D, v0e = k0e + a.heads * D;
  const int KT = (a.Tk + 15) >> 4, QT = (a.T + 15) >> 4;

  // K rows (A operand of S^T) and V pieces (A operand of O^T) of every key tile: fetched ONCE per (sample, head) -- they do not
  // depend on the query tile -- and up-front, so that all global loads of the head are in flight together.
  float kr[KTM][16], vr[KTM][16];
#pragma unroll
  for (int kt = 0; kt < KTM; ++kt) {
    if (kt < KT) {
      const int j = kt * 16 + lo;
      ld_row16<HK>(kb, k0e + (int64_t)(j < a.Tk ? j : 0) * a.ldkv + 16 * g, kr[kt]);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int jj = kt * 16 + 4 * g + s;           // key row this lane quarter feeds at PV step s
        ld_row4<HK>(kb, v0e + (int64_t)(jj < a.Tk ? jj : 0) * a.ldkv + 4 * lo, &vr[kt][4 * s], jj < a.Tk);
      }
    }
  }
  for (int qt = 0; qt < QT; ++qt) {
    // B operand of S^T: this lane's 16 features of query row i
    const int i = qt * 16 + lo;
    float qr[16];
    ld_row16<HQ>(qb, q0 + (int64_t)(i < a.T ? i : 0) * a.ldq + 16 * g, qr);
    f32x4 st[KTM];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < KTM; ++kt) {
      st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kt < KT) {
        f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
        for (int s = 0; s < 16; s += 2) {
          s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[kt][s], qr[s], s0, 0, 0, 0);
          s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[kt][s + 1], qr[s + 1], s1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int jj = kt * 16 + 4 * g + r;          // key index of accumulator register r
          const float sv = jj < a.Tk ? (s0[r] + s1[r]) * a.scale : -INFINITY;
          st[kt][r] = sv;
          mx = fmaxf(mx, sv);
        }
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < KTM; ++kt) {
      if (kt < KT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = expf(st[kt][r] - mx);           // exp(-inf) = 0 for masked keys
          st[kt][r] = e;
          sum += e;
        }
      }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    f32x4 acc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < KTM; ++kt) {
      if (kt < KT) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float pr = st[kt][s] / sum;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
            acc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[kt][4 * s + dt], pr, acc[dt], 0, 0, 0);
        }
      }
    }
    if (i < a.T) st_out16(a, b, h, i, g, acc, 1.0f);
  }
}

// The same operator for LONG sequences (more than 64 queries or keys per sample: the reference's default max_length = 1024 puts
// 256 tokens on the first attention level, generative.py:720-776): one wave per (sample, head, 16-query tile), the keys in
// chunks of 64 with a running maximum / sum (online softmax) -- the probabilities of a chunk are formed against the running
// maximum, the accumulated O^T and sum are rescaled when it moves; the result is divided by the sum once at the end.  Layouts,
// MFMA operand maps and exactness (fp32 MFMA, expf) as k_attn above; mathematically the same softmax, rounding differs from
// the two-pass form by a few ulp.
template <int IN16>
__global__ __launch_bounds__(256) void k_attn_long(AttnArgs a) {
  constexpr int D = 64, KTM = 4;
  constexpr bool HQ = IN16 & 1, HK = (IN16 & 2) != 0;
  const int QT = (a.T + 15) >> 4;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= (int64_t)a.batch * a.heads * QT) return;
  const int qt = (int)(wid % QT);
  const int bh = (int)(wid / QT);
  const int b = bh / a.heads, h = bh % a.heads;
  const int lane = threadIdx.x & 63;
  const int lo = lane & 15, g = lane >> 4;
  const unsigned char* qb = reinterpret_cast<const unsigned char*>(a.q);
  const unsigned char* kb = reinterpret_cast<const unsigned char*>(a.k);
  const int64_t q0 = (int64_t)b * a.T * a.ldq + h * D, k0e = (int64_t)b * a.kv_bstride * a.ldkv + h * D, v0e = k0e + a.heads * D;
  const int i = qt * 16 + lo;
  float qr[16];
  ld_row16<HQ>(qb, q0 + (int64_t)(i < a.T ? i : 0) * a.ldq + 16 * g, qr);
  f32x4 acc[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  for (int k0 = 0; k0 < a.Tk; k0 += 16 * KTM) {
    float kr[KTM][16], vr[KTM][16];
#pragma unroll
    for (int kt = 0; kt < KTM; ++kt) {
      const int j = k0 + kt * 16 + lo;
      ld_row16<HK>(kb, k0e + (int64_t)(j < a.Tk ? j : 0) * a.ldkv + 16 * g, kr[kt]);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int jj = k0 + kt * 16 + 4 * g + s;
        ld_row4<HK>(kb, v0e + (int64_t)(jj < a.Tk ? jj : 0) * a.ldkv + 4 * lo, &vr[kt][4 * s], jj < a.Tk);
      }
    }
    f32x4 st[KTM];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < KTM; ++kt) {
      f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
      for (int s = 0; s < 16; s += 2) {
        s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[kt][s], qr[s], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[kt][s + 1], qr[s + 1], s1, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int jj = k0 + kt * 16 + 4 * g + r;
        const float sv = jj < a.Tk ? (s0[r] + s1[r]) * a.scale : -INFINITY;
        st[kt][r] = sv;
        mx = fmaxf(mx, sv);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);                 // finite: every chunk holds at least one real key
    const float alpha = expf(m_run - m_new);              // exp(-inf) = 0 on the first chunk
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < KTM; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = expf(st[kt][r] - m_new);
        st[kt][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    l_run = l_run * alpha + sum;
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) acc[dt] *= alpha;
#pragma unroll
    for (int kt = 0; kt < KTM; ++kt)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          acc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[kt][4 * s + dt], st[kt][s], acc[dt], 0, 0, 0);
  }
  if (i >= a.T) return;
  st_out16(a, b, h, i, g, acc, 1.0f / l_run);
}

template <int IN16>
static hipError_t launch_attn_in(const AttnArgs& a, hipStream_t s) {
  if (a.T > 64 || a.Tk > 64) {
    const int64_t w = (int64_t)a.batch * a.heads * ((a.T + 15) / 16);
    if ((w + 3) / 4 > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_attn_long<IN16>, dim3((unsigned)((w + 3) / 4)), dim3(256), 0, s, a);
    return hipGetLastError();
  }
  const int waves = a.batch * a.heads;
  if (a.Tk <= 16)
    hipLaunchKernelGGL((k_attn<1, IN16>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, a);
  else if (a.Tk <= 32)   // (round 6: two key tiles in registers instead of four -- 64 registers fewer, more waves per SIMD under the loads)
    hipLaunchKernelGGL((k_attn<2, IN16>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((k_attn<4, IN16>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_attn(const AttnArgs& a, hipStream_t s) {
  if (a.batch <= 0) return hipSuccess;
  if (a.T <= 0 || a.Tk <= 0 || a.T > 8192 || a.Tk > 8192 || a.ldo % 4) return hipErrorInvalidValue;
  if (a.in16 < 0 || a.in16 > 3) return hipErrorInvalidValue;
  if (a.ldq % ((a.in16 & 1) ? 8 : 4) || a.ldkv % ((a.in16 & 2) ? 8 : 4)) return hipErrorInvalidValue;   // 16-byte row pieces
  if (a.out16 && a.ldo % 8) return hipErrorInvalidValue;
  switch (a.in16) {
    case 0: return launch_attn_in<0>(a, s);
    case 1: return launch_attn_in<1>(a, s);
    case 2: return launch_attn_in<2>(a, s);
    default: return launch_attn_in<3>(a, s);
  }
}


// ------------------------------------------------------------------------------------------------------------------
// Cross-attention against the NORMALISED CONTEXT ITSELF (MDT_OP_ATTN_CTX): the per-layer key / value projections are
// folded into the query and output projections on the host (compiler.py::attention_layer_folded),
//   S_h = (LN(x) Mq_h^T) c^T,   out = sum_h (softmax(S_h) c) N_h^T,     c = (ctx - mean) / std  (no affine),
// so every head of every layer attends to the same Tk x 128 matrix c instead of its own hoisted K / V rows
// (QMDiffusionForward: 64 keys x 1024 floats per sample and layer = 1 GB per layer at B = 4096, the launch was HBM-bound).
// K = V = c is shared by the heads, hence the (token, head) pairs of a sample are simply ROWS of one problem:
//   Q' [R = T * heads rows][128]  x  c [Tk <= 64][128]   ->   out [R][128]
// one wave per (sample, 16-row tile of R); both contractions in exact fp32 MFMA out of registers.  The MFMA contraction
// index is free to permute, so both loads are laid out for coalescing instead of for the formula:
//   S^T = c Q'^T : step s = 4 cc + e of lane quarter g contracts feature 16 cc + 4 g + e -> one dwordx4 per cc, the four
//                  quarters of a row read 64 contiguous bytes;
//   O^T = c^T P^T: output row i of tile dt is feature 4 i + dt -> the A operand of the four tiles is ONE dwordx4 of key
//                  row 16 kt + 4 g + s at floats [64 half + 4 lo, +4), a full 256 B row segment per lane quarter, and
//                  lane (query, g) ends with out[query][64 half + 16 g + 4 r .. +4) per r = one dwordx4 store.
// ------------------------------------------------------------------------------------------------------------------
// lane-group exchanges over +-16 / +-32 lanes with the gfx950 permlane swaps (k_rconv.hip: VALU, no LDS round trip)
#define MDT_XG(NAME, INSN, COMBINE)                                                      \
  __device__ __forceinline__ float NAME(float v) {                                       \
    float a = v, b = v;                                                                  \
    asm("s_nop 1\n\t" INSN " %0, %1" : "+v"(a), "+v"(b));                                \
    return COMBINE;                                                                      \
  }
MDT_XG(xg16_add, "v_permlane16_swap_b32", a + b)
MDT_XG(xg32_add, "v_permlane32_swap_b32", a + b)
MDT_XG(xg16_max, "v_permlane16_swap_b32", fmaxf(a, b))
MDT_XG(xg32_max, "v_permlane32_swap_b32", fmaxf(a, b))
#undef MDT_XG

__device__ __forceinline__ void lds_read_f4(f32x4& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const unsigned char* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)p;
}

// Round 4: the context streams through LDS.  The first form loaded its operands straight from global memory into registers
// (one dependent round trip per 16-key tile, then a second pass over the same rows for the value side): 3.1-3.8 TB/s with the
// waves stalled on issue.  Now every wave owns a two-slot ring of 8 KB in LDS and walks a flat stream of
// (work unit, 16-key chunk) pairs: the chunk two ahead is in flight by LDS-DMA (1 KB per instruction, fully coalesced) while
// the current one is used for BOTH contractions out of LDS -- an online softmax (running maximum and sum per query row,
// accumulators rescaled per chunk) needs each context row once.  The ring is private to the wave: no barrier anywhere, a
// counted s_waitcnt vmcnt is the only synchronisation (loads return in order; the output stores of the previous unit are issued
// a whole chunk before the next wait so that they are never what it waits for).
//   LDS image of a chunk: 16 rows x 512 B, the 16-byte slots of a row XOR-swizzled THROUGH THE SOURCE ADDRESS of the DMA (its LDS
//   side is lane-linear) with ctx_swz(row) = row ^ ((row & 4) << 1).  Round 5: the swizzle used to be the row number itself,
//   which is conflict-free for CONTIGUOUS 16-lane groups -- but a ds_read_b128 is banked over the four groups {0-3, 12-15, 20-27},
//   {4-11, 16-19, 28-31}, {32-35, 44-47, 52-59}, {36-43, 48-51, 60-63} (MI355X_MICROARCH.md, LDS): a group mixes two quarters g
//   of the wave, i.e. on the O side two ROWS 4 apart, whose slots lo ^ row collided pairwise (2-way: SQ_LDS_BANK_CONFLICT 29-31 %
//   of the LDS cycles in profiles/r4_cfg3_pmc_sq.csv).  With bit 3 of the swizzle flipped for rows with bit 2 set, the two rows
//   of a group take complementary slot sets on the O side (lanes {0-3, 12-15} ^ x and {4-11} ^ x' are disjoint when bits 3:2 of
//   x ^ x' are 11) and the S side (16 rows, slot 4 cc + g) keeps its two quarters in opposite halves of the bank row.
#if defined(MDT_TUNING) && defined(MDT_ABL_CTX_CLOCK)   // probe only: shader-clock and 100 MHz stamps of workgroup 0, wave 0
__device__ unsigned long long g_ctx_stamp[4 + 2 * 1024];   // [0..1]: workgroup 0's ticks; then (start, end) of every workgroup on the 100 MHz clock
#endif

// SPLIT: the scores S = q' c^T as split-bf16 products (hi hi + hi lo + lo hi on v_mfma_f32_16x16x32_bf16: 12 MFMAs of 16 cycles per
// row tile and chunk instead of 32 of 32 cycles; the default mode's arithmetic, like every other GEMM of that mode); the output
// side P c stays exact fp32 (its contraction runs over keys: the bf16 form would need the context tile transposed).  The operand
// registers ARE the fp32 fragments: k-slot e of step s is feature 16 (2 s + (e >> 2)) + 4 g + (e & 3) for q' and c alike.
__device__ __forceinline__ int ctx_swz(int row) { return row ^ ((row & 4) << 1); }

template <int RT, bool SPLIT>
__global__ __launch_bounds__(256, 2) void k_attn_ctx(AttnArgs a) {
  constexpr int D = 128, CH = 8192;
#if defined(MDT_TUNING) && defined(MDT_ABL_CTX_CLOCK)
  unsigned long long t0c, t0r;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0c), "=s"(t0r)::"memory");
#endif
  __shared__ __attribute__((aligned(1024))) unsigned char lds_all[4 * 2 * CH];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lo = lane & 15, g = lane >> 4;
  unsigned char* ring = lds_all + wv * (2 * CH);
  const int R = a.T * a.heads;                         // rows per sample
  const int QG = (R + 16 * RT - 1) / (16 * RT);        // work units per sample
  const int NCH = (a.Tk + 15) >> 4;                    // 16-key chunks per unit
  const int W = gridDim.x * 4, total = a.batch * QG;
  int u = blockIdx.x * 4 + wv;                         // this wave's units: u, u + W, ...
  if (u >= total) return;
  const int u0 = u;

  // ---- the DMA side.  Stream elements of a unit: its RT query tiles (16 rows of q' each), then its NCH context chunks; every
  // element is 16 rows x 512 B = one slot = eight 1 KB instructions, element sj goes to slot sj & 1.  The query rows travel the
  // same way (and are copied LDS -> registers when their element is consumed) so that the launch has NO compiler-visible vector
  // load: with one pending, hipcc guards its use -- and every LDS read behind an LDS-DMA -- with s_waitcnt vmcnt(0), which is
  // the whole look-ahead.  Past the wave's last element the same eight instructions re-read an element of its first unit into
  // the (free) slot: the counted wait below is vmcnt(8) on every path.
  int su = u, se = 0, sj = 0;                          // next stream element to issue: unit, element of the unit, stream index
  const int l5 = lane >> 5, p15 = lane & 15, p16 = lane & 16;
  auto issue_next = [&]() {
#if defined(MDT_TUNING) && defined(MDT_ABL_CTX_NODMA)   // timing only: the ring is filled twice and never again
    if (sj >= 2) { ++sj; if (++se == RT + NCH) { se = 0; su += W; } return; }
#endif
    const int uu = su < total ? su : u0;
    const int b = uu / QG, qg = uu - b * QG;
    const bool isq = se < RT;
    const unsigned char* base = isq ? reinterpret_cast<const unsigned char*>(a.q + (int64_t)b * R * D)
                                    : reinterpret_cast<const unsigned char*>(a.k + (int64_t)b * a.kv_bstride * a.ldkv);
    const int row0 = isq ? (qg * RT + se) * 16 : (se - RT) * 16, nrows = isq ? R : a.Tk, ld4 = (isq ? D : a.ldkv) * 4;
    unsigned char* dst = ring + (sj & 1) * CH;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = 2 * i + l5;                        // row of the element this lane's 16 bytes belong to
      int row = row0 + r;
      row = row < nrows ? row : 0;                     // rows past the end: any valid row (masked scores / unstored outputs)
      const int logical = p16 | (p15 ^ ctx_swz(r));
      __builtin_amdgcn_global_load_lds(base + (int64_t)row * ld4 + logical * 16, (lds_ptr)(dst + i * 1024), 16, 0, 0);
    }
    ++sj;
    if (++se == RT + NCH) { se = 0; su += W; }
  };

  // S side: lane (key lo, quarter g) reads slot 4 cc + g of row lo (cc & 4 is the upper half of the row: +256 bytes);
  // O side: lane (feature group lo, quarter g) reads slot 16 half + lo of row 4 g + s
  unsigned aS[4], aV[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    aS[c] = lo * 512 + (((4 * c + g) ^ ctx_swz(lo)) << 4);
    const int jr = 4 * g + c;
    aV[c] = jr * 512 + ((lo ^ ctx_swz(jr)) << 4);
  }
  f32x4 qr[SPLIT ? 1 : RT][8];                         // qr[t][cc][e] = q'[row lo of tile t][16 cc + 4 g + e]   (exact form)
  bf16x8 qh[SPLIT ? RT : 1][4], ql[SPLIT ? RT : 1][4]; // the same values as bf16 hi / lo planes, one register pair per 32-wide step
  int qi[RT];
  auto split8 = [](const f32x4& u, const f32x4& v, bf16x8& hi, bf16x8& lo) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = e < 4 ? u[e & 3] : v[e & 3];
      const __bf16 h = (__bf16)x;
      hi[e] = h;
      lo[e] = (__bf16)(x - (float)h);
    }
  };

  f32x4 acc[RT][2][4];
  float mx[RT], ls[RT];
  int prev_unit = -1, prev_qi[RT];
  auto store_prev = [&]() {                            // normalise and write the finished unit out of the accumulators
    const int b = prev_unit / QG;
    float* o = a.out + (int64_t)b * R * D;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      const float inv = 1.0f / xg32_add(xg16_add(ls[t]));
      if (prev_qi[t] < R) {
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            *reinterpret_cast<float4*>(o + (int64_t)prev_qi[t] * D + 64 * half + 16 * g + 4 * r) =
                make_float4(acc[t][half][0][r] * inv, acc[t][half][1][r] * inv, acc[t][half][2][r] * inv, acc[t][half][3][r] * inv);
      }
    }
  };

  issue_next();
  issue_next();
  int j = 0;                                           // stream index of the element being consumed
  for (; u < total; u += W) {
    {
      const int b = u / QG, qg = u - b * QG;
#pragma unroll
      for (int t = 0; t < RT; ++t, ++j) {              // the unit's query tiles: LDS -> registers (the S-side read pattern)
        qi[t] = (qg * RT + t) * 16 + lo;
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        const unsigned cur = lds_addr(ring + (j & 1) * CH);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) lds_read_f4(qr[SPLIT ? 0 : t][cc], cur + aS[cc & 3] + (cc & 4) * 64);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (SPLIT) {
#pragma unroll
          for (int st = 0; st < 4; ++st) split8(qr[0][2 * st], qr[0][2 * st + 1], qh[t][st], ql[t][st]);
        }
        issue_next();
      }
    }
    for (int k = 0; k < NCH; ++k, ++j) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // element j has landed (eight younger loads may be in flight)
      // fragment reads as inline asm with hand-counted lgkmcnt waits (the ring kernels' way, tools/isa_lint.py checks the
      // landing registers): a compiler-visible LDS read behind an LDS-DMA costs an s_waitcnt vmcnt(0) -- the whole look-ahead
      const unsigned cur = lds_addr(ring + (j & 1) * CH);
      f32x4 st[RT];
      f32x4 kr4[8], vr[2][4];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) lds_read_f4(kr4[cc], cur + aS[cc & 3] + (cc & 4) * 64);
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) lds_read_f4(vr[0][s_], cur + aV[s_]);
      asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 kh[SPLIT ? 4 : 1], kl[SPLIT ? 4 : 1];
      if constexpr (SPLIT) {
#pragma unroll
        for (int st = 0; st < 4; ++st) split8(kr4[2 * st], kr4[2 * st + 1], kh[st], kl[st]);
      }
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0;
        if constexpr (SPLIT) {
#pragma unroll
          for (int st = 0; st < 4; ++st) {
            s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl[st], qh[t][st], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh[st], ql[t][st], s1, 0, 0, 0);
            s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh[st], qh[t][st], s0, 0, 0, 0);
          }
        } else
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
#if defined(MDT_TUNING) && defined(MDT_ABL_CTX_NOS)      // timing only: one of the 32 score MFMAs
          if (cc) { s0[0] += kr4[cc][0] * qr[t][cc][1]; continue; }
#endif
          s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kr4[cc][0], qr[t][cc][0], s0, 0, 0, 0);
          s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kr4[cc][1], qr[t][cc][1], s1, 0, 0, 0);
          s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kr4[cc][2], qr[t][cc][2], s0, 0, 0, 0);
          s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kr4[cc][3], qr[t][cc][3], s1, 0, 0, 0);
        }
        st[t] = s0 + s1;
      }
      if (k == 0) {
        if (prev_unit >= 0) store_prev();              // a whole chunk of MFMAs before the next counted wait
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          mx[t] = -INFINITY; ls[t] = 0.f; prev_qi[t] = qi[t];
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) acc[t][h][c4] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        prev_unit = u;
      }
#pragma unroll
      for (int t = 0; t < RT; ++t) {                   // online softmax: lane (query lo, quarter g) holds keys 4 g + r
        float cm = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float sv = (k * 16 + 4 * g + r) < a.Tk ? st[t][r] * a.scale : -INFINITY;
          st[t][r] = sv;
          cm = fmaxf(cm, sv);
        }
#if defined(MDT_TUNING) && defined(MDT_ABL_CTX_NOSM)     // timing only: no running maximum, no exponentials, no rescale
        ls[t] += st[t][0]; continue;
#endif
        cm = xg32_max(xg16_max(cm));
        const float mn = fmaxf(mx[t], cm);             // finite: every chunk holds at least one key
        const float sc = __expf(mx[t] - mn);
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __expf(st[t][r] - mn);
          st[t][r] = e;
          sum += e;
        }
        ls[t] = ls[t] * sc + sum;
        mx[t] = mn;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int c4 = 0; c4 < 4; ++c4) acc[t][h][c4] *= sc;
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // the value rows of half 0 (issued with the key rows)
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) lds_read_f4(vr[1][s_], cur + aV[s_] + 256);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (half == 1) {
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
          for (int s_ = 0; s_ < 4; ++s_) {
            const float pr = st[t][s_];
#if defined(MDT_TUNING) && defined(MDT_ABL_CTX_NOPV)     // timing only: a quarter of the output MFMAs
            if (s_) { acc[t][half][0][0] += vr[half][s_][0] * pr; continue; }
#endif
            acc[t][half][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[half][s_][0], pr, acc[t][half][0], 0, 0, 0);
            acc[t][half][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[half][s_][1], pr, acc[t][half][1], 0, 0, 0);
            acc[t][half][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[half][s_][2], pr, acc[t][half][2], 0, 0, 0);
            acc[t][half][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[half][s_][3], pr, acc[t][half][3], 0, 0, 0);
          }
      }
      __builtin_amdgcn_sched_barrier(0);
      // (every read of this slot has returned: the last wait above) -- refill it
      issue_next();
    }
  }
  store_prev();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the look-ahead past the last element has landed before the ring is given back
#if defined(MDT_TUNING) && defined(MDT_ABL_CTX_CLOCK)
  unsigned long long t1c, t1r;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1c), "=s"(t1r)::"memory");
  if (blockIdx.x == 0 && threadIdx.x == 0) { g_ctx_stamp[0] = t1c - t0c; g_ctx_stamp[1] = t1r - t0r; }
  if (threadIdx.x == 0 && blockIdx.x < 1024) { g_ctx_stamp[4 + 2 * blockIdx.x] = t0r; g_ctx_stamp[5 + 2 * blockIdx.x] = t1r; }
#endif
}

hipError_t launch_attn_ctx(const AttnArgs& a, hipStream_t s) {
  if (a.batch <= 0) return hipSuccess;
  if (a.Tk <= 0 || a.Tk > 64 || a.ldkv % 4 || a.T <= 0 || a.heads <= 0) return hipErrorInvalidValue;
  static int wgs_of[64] = {0};                         // two workgroups (8 waves, 128 KB of rings) per CU, per DEVICE
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (!wgs_of[dev]) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    wgs_of[dev] = 2 * cus;
#if defined(MDT_TUNING) && defined(MDT_ABL_CTX_OCC1)
    wgs_of[dev] = cus;
#endif
  }
  const int wgs = wgs_of[dev];
  const int R = a.T * a.heads;
  if (R > 16) {                                        // two row tiles of a sample share the context stream
    const dim3 grid((unsigned)std::min(wgs, (a.batch * ((R + 31) / 32) + 3) / 4));
    if (a.split_scores) hipLaunchKernelGGL((k_attn_ctx<2, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_attn_ctx<2, false>), grid, dim3(256), 0, s, a);
  } else {
    const dim3 grid((unsigned)std::min(wgs, (a.batch + 3) / 4));
    if (a.split_scores) hipLaunchKernelGGL((k_attn_ctx<1, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_attn_ctx<1, false>), grid, dim3(256), 0, s, a);
  }
  return hipGetLastError();
}

}  // namespace mdt
