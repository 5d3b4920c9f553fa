// EXPERIMENT (round 2, not part of the build): MDT_OP_TF256 on 16-row workgroups.  Correct (every test_fused_transformer case
// at C = 256 passes against the interpreter and the module arithmetic, bitwise reproducible), but it has no winning regime:
// configs[1] evaluation at B = 1024: 2.250 ms against 2.149 ms for the head-split launches (k_tblock32.hip) and 2.396 ms for
// the 32-row k_tf256.hip; at B = 1536 / 2048: 3.88 / 4.07 ms against 3.12 / 3.33 ms for k_tf256.hip.  With a quarter of the
// MFMAs per wave the streamed part does run at the DMA rate, but a head is then only ~3400 cycles of streaming between ~5000
// cycles of per-head serial work and synchronisation (ten workgroup barriers, two LDS exchanges, softmax) -- DESIGN.md 3.5.
// To build it: add the file to build.py's SOURCES, declare launch_tf256q in mdt_kernels.h and dispatch to it in mdt_api.cpp.
//
// MDT_OP_TF256 with 16-ROW workgroups: the small-batch form of k_tf256.hip (same op, same tile / descriptor / vector streams).
//
// Why: at B = 1024 the 256-channel level has 4096 rows = 128 workgroups of 32 rows; k_tf256.hip leaves half the CUs idle
// there and the head-split launches of k_tblock32.hip pay a prologue, an epilogue and a launch gap per SUB-BLOCK (44 launches,
// ~8 of their ~24 us each).  The ring kernels are issue-bound, not stream-bound: the time of a sub-tile grows
// with the MFMA units a wave has to issue for it -- tools/ubench/proj_phase.hip: 1597 / 968 / 620 cycles per 32 KB for 8 / 4 / 2
// units per wave, the last one AT the pure LDS-DMA rate.  So: 16 rows per workgroup (256 workgroups at B = 1024), every tile's
// OUTPUT features split over the four compute waves (2 units per wave and sub-tile), a whole transformer per launch.
//
//   * compute wave fq (0..3) produces a quarter of every tile's outputs from COMPLETE inputs: 16 of a projection chunk's 64
//     features (K = the row's 256 channels, two sub-tiles), 32 of an output sub-tile's 128 channels (K = the chunk's 64
//     features).  No partial sums over K anywhere except S^T = K Q^T (contraction over the head's features: four partial
//     16 x 16 tiles meet in LDS, summed in a fixed order);
//   * what a wave needs from the others it takes from LDS: the attention output / hidden chunk (16 features per wave -> the
//     64-feature operand of the output projection: 4 KB per head), and at every sub-block boundary the residual row (each
//     wave owns 64 channels, writes them into the SCRATCH TILE the stream carries behind every sub-block, and reads the whole
//     row back: LayerNorm statistics and the operands of the next projections are computed by every wave for itself);
//   * everything else -- ring protocol, sub-tile formats, accumulator -> operand map, vectors, K / V tiles -- is k_tf256.hip's.
//
// A whole Transformer1d (modules.py:469-524) of a C = 256 level in ONE launch (MDT_OP_TF256), 16-row workgroups:
//
//   x = Conv1d_1x1(GroupNorm32(x))                       to_in            (:485-490, :520)
//   per TransformerBlock (:456-461):  x += Attention(x);  [x += Attention(x, context);]  x += FeedForward(x)
//   x = Conv1d_1x1(x)                                    to_out           (:512-516, :524), folded into the last FF
//
// Why: the 256-channel level has 4 tokens per sample, 4096 rows at B = 1024.  As one launch per sub-block (k_tblock32.hip)
// every launch pays ~5 us of prologue / epilogue plus the launch gap for 13..23 us of work, the heads are split over two
// workgroups to fill the chip and the partial sums travel through HBM between the launches.  Here a workgroup keeps its 32
// rows for the whole transformer: no split, no partial-sum tensors, one prologue and one epilogue per transformer, and the
// loader waves stream the weights of ALL sub-blocks through the LDS ring.  (128 workgroups at B = 1024: the launch is bound
// by the per-CU L2 -> LDS stream either way, which does not depend on how many rows share a workgroup.)
//
//   * compute wave w = (row tile rt = w >> 1, feature half fh = w & 1) as in k_tblock32.hip: 16 rows x 32 of each chunk's
//     64 features; partial S^T = K Q^T is exchanged through LDS; the output projection accumulates PARTIAL sums over the
//     wave's 32-feature k-slice for all 256 output channels;
//   * the residual stream lives in those accumulators (lane (i, g): x[row i][16 ct + 4 g + r], as in k_tf128.hip): wave
//     fh = 0 starts a sub-block from x + bias, wave fh = 1 from 0; at the end of the sub-block the two partial sums are
//     exchanged through two SCRATCH TILES of the ring (descriptor kind 4 / 5: the loaders issue no weight DMA for them, the
//     slots carry the accumulators instead) and added in a fixed order, so both waves hold the identical new row;
//   * the next projection's operands come from the accumulators without lane movement (K columns of the consuming tiles
//     permuted on the host, k_tf128.hip); each sub-block's vectors (biases) arrive in a double-buffered 3 KB LDS area by
//     LDS-DMA with the scratch tile in front of the sub-block.
//
// Ring protocol, sub-tile formats and the attention core are those of k_tblock32.hip / k_tblock_lw.hip.
#include <cstdlib>
#include <type_traits>

#include "mdt_kernels.h"

namespace mdt {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(4))) const unsigned* cu32p;   // constant address space: scalar loads

__device__ __forceinline__ void store_nt(float* p, float4 v) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(f4{v.x, v.y, v.z, v.w}, reinterpret_cast<f4*>(p));
}

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

enum { K_T = 0, K_N = 1, K_O = 2 };   // transposed projection, un-transposed projection, output projection
enum { D_P = 0, D_O = 1, D_K = 2, D_V = 3, D_SCRATCH = 4, D_SCRATCH_VEC = 5 };   // tile descriptor kinds (3 bits)

#define MDT_MFMA_BF16 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define MDT_MFMA_F32 __builtin_amdgcn_mfma_f32_16x16x4f32

__device__ __forceinline__ float gelu_tf(float x) {   // exact-erf GELU, branch-free erf (A&S 7.1.26), see k_tblock.hip
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erfa = 1.0f - poly * __expf(-z * z);
  return 0.5f * x * (1.0f + copysignf(erfa, x));
}

__device__ __forceinline__ void split8_tf(const float v[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 h = (__bf16)v[e];
    hi[e] = h;
    lo[e] = (__bf16)(v[e] - (float)h);
  }
}

template <int OFF>
__device__ __forceinline__ void lds_read16_off(bf16x8& dst, unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read_b128 offset field");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}

template <int OFF>
__device__ __forceinline__ void lds_read_f4_off(f32x4& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}

__device__ __forceinline__ unsigned lds_addr(const unsigned char* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)p;
}

template <int N>
__device__ __forceinline__ void lgkm_wait() {   // at most N LDS/scalar operations still in flight
  if constexpr (N >= 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
  else if constexpr (N >= 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

constexpr int C = 256;          // channels
constexpr int CS = 128;         // sub-tile width (k_tblock32.hip)
constexpr int SLOT = 256 * CS;  // bytes per sub-tile (bf16 hi plane + lo plane)
constexpr int NS = 4;           // ring slots
constexpr int IPT = CS / 16;    // DMA pieces per sub-tile per loader wave
constexpr int NST = C / 32;     // k-steps of a full projection
constexpr int NCT = C / 16;     // 16-row tiles of the output projection
constexpr int NU = 2;           // units (4 fragment reads + 6 MFMAs) per sub-tile per wave
constexpr int KTM = 3;          // key tiles per wave (cross): at most 48 context rows per 16 token rows
constexpr int RED_BYTES = KTM * 4 * 64 * 16;   // partial S^T exchange [key tile][4 waves][64 lanes] f32x4
constexpr int OG_BYTES = 4 * 64 * 16;          // attention output / hidden chunk gather [4 waves][64 lanes] f32x4
constexpr int VEC_FLOATS = 768;                // vectors of one sub-block: [bq 512 | bo 256], [b1 512 | b2 256], [b_in 256]
constexpr int VEC_BYTES = VEC_FLOATS * 4;

}  // namespace

// NPW: LDS-DMA pieces per loader wave per K / V tile = ceil(context rows of the workgroup / 16); 0 = no cross-attention
template <int NPW>
__global__ __launch_bounds__(512) void k_tf256q(TFArgs a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* red_b = smem + NS * SLOT;
  unsigned char* vec_b = red_b + RED_BYTES;          // two parities of VEC_BYTES
  unsigned char* og_b = vec_b + 2 * VEC_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NT = a.NT;
  const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.w);

  if (wave >= 4) {
    // ================= loader waves (k_tblock32.hip, descriptor-driven as k_tf128.hip) =================
    const int iw = wave - 4;
    __builtin_amdgcn_s_setprio(MDT_LOADER_PRIO);
    const cu32p tiles = (cu32p)a.tiles;              // kind (3 bits) | aux << 3
    const int lpP = lane >> 5;
    const int xP = (lane & 15) ^ lpP;
    const int baseP = ((lane >> 4) & 1) * (128 * CS) + lpP * (2 * CS);
    const int xO = (lane & 7) ^ (lane >> 4);
    const int baseO = (lane >> 3) * 128;
    unsigned voffP[IPT], voffO[IPT];
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
      const int inst = iw + 4 * q;
      const int U = 2 * inst;
      voffP[q] = (unsigned)(U * (2 * CS) + ((xP ^ (U & 15)) << 4) + baseP);
      voffO[q] = (unsigned)(((inst * 8) / CS) * (128 * CS) + ((inst * 8) % CS) * 128 + ((xO ^ (4 * (inst & 1))) << 4) + baseO);
    }
    const int sample0 = blockIdx.x * (16 / a.T);
    const bool second = a.kv2 && sample0 >= a.nsamples / 2;      // dual batch: shared K / V rows for the second half
    unsigned voffKV[4];
    if constexpr (NPW > 0) {
      const int kv_rows = (16 / a.T) * a.Tk;
      const int bstr = second ? 0 : a.kv_bstride;
#pragma unroll
      for (int q = 0; q < NPW; ++q) {
        const int R = 4 * (iw + 4 * q) + (lane >> 4);
        const int Rc = min(R, kv_rows - 1);
        const int sm = min(Rc / a.Tk, a.nsamples - 1 - sample0), key = Rc % a.Tk;
        voffKV[q] = (unsigned)(((sm * bstr + key) * a.ldkv + 4 * ((lane & 15) ^ (R & 15))) * 4);
      }
    }
    // vector-memory operations THIS wave issues for a tile (the counted waits below are per wave)
    auto pieces_of = [&](unsigned d) -> int {
      const unsigned kind = d & 7u;
      if (kind == D_SCRATCH) return 0;
      if (kind == D_SCRATCH_VEC) return iw < VEC_BYTES / 1024 ? 1 : 0;
      if (kind >= D_K) return NPW;
      return IPT;
    };
    auto issue_vec = [&](unsigned aux) {             // aux = (float offset / 256) << 1 | parity
      if (iw < VEC_BYTES / 1024)
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const unsigned char*>(a.vec) + (aux >> 1) * 1024 + iw * 1024 + lane * 16,
                                         (__attribute__((address_space(3))) void*)(vec_b + (aux & 1u) * VEC_BYTES + iw * 1024), 16, 0, 0);
    };
    auto issue_tile = [&](int tau, unsigned d) {
      unsigned char* slot = smem + (tau % NS) * SLOT + iw * 1024;
      const unsigned kind = d & 7u, aux = d >> 3;
      if (kind == D_SCRATCH) return;
      if (kind == D_SCRATCH_VEC) { issue_vec(aux); return; }
      if (kind >= D_K) {
        if constexpr (NPW > 0) {
          const int layer = (int)(aux >> 4), head = (int)(aux & 15u);
          const float* lb = second ? a.kv2 + (int64_t)layer * a.kv2_lstride
                                   : a.kv + (int64_t)layer * a.kv_lstride + (int64_t)sample0 * a.kv_bstride * a.ldkv;
          const unsigned char* base = reinterpret_cast<const unsigned char*>(lb + 64 * head + (kind == D_V ? 64 * a.nheads : 0));
#pragma unroll
          for (int q = 0; q < NPW; ++q)
            __builtin_amdgcn_global_load_lds(base + voffKV[q], (__attribute__((address_space(3))) void*)(slot + q * 4096), 16, 0, 0);
        }
        return;
      }
      const unsigned char* tile = wsrc + (int64_t)aux * SLOT;   // wave-uniform
      const bool ptile = kind == D_P;
#pragma unroll
      for (int q = 0; q < IPT; ++q) {
        const unsigned off = ptile ? voffP[q] : voffO[q];
        __builtin_amdgcn_global_load_lds(tile + off, (__attribute__((address_space(3))) void*)(slot + q * 4096), 16, 0, 0);
      }
    };
    auto wait_vm = [&](int allow) {
      switch (allow) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      }
    };
    const unsigned d0 = tiles[0], d1 = NT > 1 ? tiles[1] : 0u;
    __builtin_amdgcn_s_barrier();   // P: the compute waves' row loads are queued ahead of the stream
    issue_vec(0u);                  // the first sub-block's vectors (parity 0), ahead of tile 0: covered by the first wait
    issue_tile(0, d0);
    if (NT > 1) issue_tile(1, d1);
    unsigned dn = d1;                                                    // descriptor of tile k + 1
    for (int k = 0; k < NT; ++k) {
      const unsigned d2 = k + 2 < NT ? tiles[k + 2] : 0u;
      wait_vm(k + 1 < NT ? pieces_of(dn) : 0);                           // tile k landed; tile k+1 may be in flight
      __builtin_amdgcn_s_barrier();                                      // B(k)
      if (k + 2 < NT) issue_tile(k + 2, d2);
      dn = d2;
    }
    prefetch_next_weights(a.pf_ptr, a.pf_lines, iw * 64 + lane);
    return;
  }


  // ================= compute waves =================
  const int i = lane & 15, g = lane >> 4;
  const int fq = wave;                               // feature quarter
  const int row0 = blockIdx.x * 16;
  const int m = row0 + i;
  const bool mvalid = m < a.M;
  const int mc = mvalid ? m : a.M - 1;
  f32x4* red = reinterpret_cast<f32x4*>(red_b);
  f32x4* og = reinterpret_cast<f32x4*>(og_b);

  // the row, complete in every wave at the start of a sub-block: xf[ct][r] = x[row i][16 ct + 4 g + r]
  f32x4 xf[NCT];
  {
    const float* xp = a.x + (int64_t)mc * C + 4 * g;
    float4 xr[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) xr[ct] = *reinterpret_cast<const float4*>(xp + 16 * ct);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                    // P
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) xf[ct] = f32x4{xr[ct].x, xr[ct].y, xr[ct].z, xr[ct].w};
  }
  // this wave's channels of the residual stream / of every output: tiles own_ct(k), k = 0..3 = 8 (k >> 1) + 2 fq + (k & 1)
  // (output sub-tile s = k >> 1 holds channels 128 s .. 128 s + 127; the wave takes its rows 32 fq .. 32 fq + 31).  Between
  // sub-blocks out[] holds the wave's channels of the row (picked by ADDRESS from memory / the LDS copy: indexing xf with the
  // wave number would put that array in scratch)
  f32x4 out[4];
  {
    const float* xp = a.x + (int64_t)mc * C + 4 * g;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(xp + 16 * (8 * (k >> 1) + 2 * fq + (k & 1)));
      out[k] = f32x4{v.x, v.y, v.z, v.w};
    }
  }

  // fragment addressing inside a sub-tile (k_tblock32.hip): projection sub-tile [64 features][128 k], this wave's 16 features;
  // output sub-tile [128 channels][64 k], this wave's 32 channels
  int aP[4], aO[2];
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    const int lc = 4 * st + g;
    aP[st] = fq * (16 * 4 * CS) + i * (4 * CS) + ((lc & ~15) | ((lc & 15) ^ i)) * 16;
  }
#pragma unroll
  for (int sp = 0; sp < 2; ++sp) aO[sp] = (2 * fq) * (16 * 128) + i * 128 + ((4 * sp + g) ^ ((i >> 1) & 7)) * 16;

  bf16x8 frh[3][2], frl[3][2];
  // read j of unit u: projection tile: k-step 2 u + (j >> 1), plane j & 1; output tile: k-step u, channel tile j >> 1, plane j & 1
  auto frag_read = [&](auto kind, unsigned base, int set, auto jc) __attribute__((always_inline)) {
    constexpr int KIND = decltype(kind)::value, j = decltype(jc)::value;
    constexpr int q = j >> 1, lo = j & 1;
    constexpr int off = (KIND == K_O) ? (q * 16 * 128 + lo * (CS * 128)) : (lo * (2 * CS));
    lds_read16_off<off>(lo ? frl[set][q] : frh[set][q], base);
  };
  using J0 = std::integral_constant<int, 0>;
  using J1 = std::integral_constant<int, 1>;
  using J2 = std::integral_constant<int, 2>;
  using J3 = std::integral_constant<int, 3>;
  // base address of read j of unit u inside the slot at LDS address l
  auto rbase = [&](auto kind, unsigned l, int u, int j) __attribute__((always_inline)) -> unsigned {
    constexpr int KIND = decltype(kind)::value;
    return l + (KIND == K_O ? aO[u] : aP[2 * u + (j >> 1)]);
  };
  auto prefetch2 = [&](auto kind, const unsigned char* slot, int off) __attribute__((always_inline)) {
    const unsigned l = lds_addr(slot);
    frag_read(kind, rbase(kind, l, 0, 0), off % 3, J0{}); frag_read(kind, rbase(kind, l, 0, 1), off % 3, J1{});
    frag_read(kind, rbase(kind, l, 0, 2), off % 3, J2{}); frag_read(kind, rbase(kind, l, 0, 3), off % 3, J3{});
    frag_read(kind, rbase(kind, l, 1, 0), (off + 1) % 3, J0{}); frag_read(kind, rbase(kind, l, 1, 1), (off + 1) % 3, J1{});
    frag_read(kind, rbase(kind, l, 1, 2), (off + 1) % 3, J2{}); frag_read(kind, rbase(kind, l, 1, 3), (off + 1) % 3, J3{});
  };

  int tau = 0;
  auto slot_of = [&](int t) -> unsigned char* { return smem + (t % NS) * SLOT; };

  // One MFMA phase over a sub-tile: 2 units of 4 fragment reads + 6 MFMAs.  Projection tile: unit u = k-steps 2 u, 2 u + 1 of the
  // wave's 16 features into the accumulator pair acc[0] / acc[1] (two dependency chains; the caller adds them); output tile:
  // unit u = k-step u of the wave's two channel tiles acc[0], acc[1].  bar: execute B(tau + 1) (at the first unit: every read
  // of this phase goes to the NEXT tile); pre: prefetch the next tile (of kind nkind).  Both compile-time (see below).
  auto phase = [&](auto kind, auto offc, auto nkind, auto barc, auto prec, f32x4* acc, const bf16x8* bh,
                   const bf16x8* bl) __attribute__((always_inline)) {
    constexpr int KIND = decltype(kind)::value, OFF = decltype(offc)::value;
    constexpr bool bar = decltype(barc)::value, pre = decltype(prec)::value;
    const unsigned ln = lds_addr(slot_of(tau + 1));
    auto unit = [&](auto uc) __attribute__((always_inline)) {
      constexpr int u = decltype(uc)::value;
      if constexpr (u == 0 && bar) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                // B(tau + 1)
        __builtin_amdgcn_sched_barrier(0);
      }
      constexpr int s0 = (OFF + u) % 3, s2 = (OFF + u + 2) % 3;
      constexpr bool later = (u + 1 < NU) || pre;
      if constexpr (later) lgkm_wait<4>(); else lgkm_wait<0>();
      auto rd = [&](auto jc) __attribute__((always_inline)) {
        if constexpr (pre) {
          __builtin_amdgcn_sched_barrier(0);
          frag_read(nkind, rbase(nkind, ln, u, decltype(jc)::value), s2, jc);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      auto mm = [&](const bf16x8& w, const bf16x8& x, int q) __attribute__((always_inline)) {
        if constexpr (KIND == K_N) acc[q] = MDT_MFMA_BF16(x, w, acc[q], 0, 0, 0);
        else acc[q] = MDT_MFMA_BF16(w, x, acc[q], 0, 0, 0);
      };
      // projection: the two fragment pairs are the unit's two k-steps (operand k-steps 2 u, 2 u + 1); output: its two channel
      // tiles (operand k-step u)
      constexpr int b0 = (KIND == K_O) ? u : 2 * u, b1 = (KIND == K_O) ? u : 2 * u + 1;
      mm(frl[s0][0], bh[b0], 0); rd(J0{});
      mm(frl[s0][1], bh[b1], 1); rd(J1{});
      mm(frh[s0][0], bl[b0], 0); rd(J2{});
      mm(frh[s0][1], bl[b1], 1); rd(J3{});
      mm(frh[s0][0], bh[b0], 0);
      mm(frh[s0][1], bh[b1], 1);
      __builtin_amdgcn_sched_barrier(0);
    };
    unit(std::integral_constant<int, 0>{}); unit(std::integral_constant<int, 1>{});
    ++tau;
  };
  // Every in-flight fragment read (inline asm: the compiler does not know its result lands later) is issued AND consumed
  // inside one straight-line region: a read that is live across a loop back-edge or a run-time branch may be followed by a
  // register copy (phi elimination) that moves the not-yet-landed register -- seen here as run-to-run differences of ~1e-3 in
  // the first projection after the back-edge of the feed-forward chunk loop.  Hence every head / chunk iteration starts its own
  // prefetch chain (begin_tile) and ends it (last phase without prefetch); bar / pre are compile-time.
  using Yes = std::true_type;
  using No = std::false_type;
  auto begin_tile = [&](auto kind) __attribute__((always_inline)) {
    __builtin_amdgcn_s_barrier();                    // B(tau)
    prefetch2(kind, slot_of(tau), 0);
  };
  using IC0 = std::integral_constant<int, 0>;
  using IC1 = std::integral_constant<int, 1>;
  using IC2 = std::integral_constant<int, 2>;
  const IC0 kT{};
  const IC1 kN{};
  const IC2 kO{};

  const int samp_q = i / a.T;
  float kmask[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) kmask[r] = ((4 * g + r) / a.T == samp_q) ? 0.f : -INFINITY;
  const float scale2 = a.scale * 1.44269504088896340736f;
  int nkeys = 0;
  unsigned okbits = 0;
  if constexpr (NPW > 0) {
    nkeys = (16 / a.T) * a.Tk;
#pragma unroll
    for (int kt = 0; kt < KTM; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int jj = 16 * kt + 4 * g + r;
        if (jj < nkeys && (jj / a.Tk) == samp_q) okbits |= 1u << (4 * kt + r);
      }
  }
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  const float t1 = a.T > 1 ? 1.f : 0.f, t2 = a.T > 2 ? 1.f : 0.f, t4 = a.T > 4 ? 1.f : 0.f, t8 = a.T > 8 ? 1.f : 0.f;
  auto dpp_fma = [](float v, float f, auto ctrl) {
    const int mm_ = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, true);
    return __builtin_fmaf(__builtin_bit_cast(float, mm_), f, v);
  };
  auto token_sum = [&](float (&s)[NCT]) __attribute__((always_inline)) {   // sums over the sample's token lanes (k_rconv.hip)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) s[ct] = dpp_fma(s[ct], t1, std::integral_constant<int, 0xB1>{});
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) s[ct] = dpp_fma(s[ct], t2, std::integral_constant<int, 0x4E>{});
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) s[ct] = dpp_fma(s[ct], t4, std::integral_constant<int, 0x141>{});
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) s[ct] = dpp_fma(s[ct], t8, std::integral_constant<int, 0x140>{});
  };

  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned vec_l0 = lds_addr(vec_b);
  int vpar = 0;                                      // parity of the current sub-block's vector area

  bf16x8 xh[NST], xl[NST];
  auto make_operands = [&](bool layernorm) __attribute__((always_inline)) {
    float mean = 0.f, rstd = 1.f;
    if (layernorm) {
      float s = 0.f;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) s += (xf[ct][0] + xf[ct][1]) + (xf[ct][2] + xf[ct][3]);
      s = xg16_add(s);
      s = xg32_add(s);
      mean = s / (float)C;
      float ss = 0.f;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = xf[ct][r] - mean;
          ss += d * d;
        }
      ss = xg16_add(ss);
      ss = xg32_add(ss);
      rstd = 1.0f / sqrtf(ss / (float)C + a.eps_ln);
    }
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = mvalid ? (xf[2 * st + (e >> 2)][e & 3] - mean) * rstd : 0.f;
      split8_tf(v, xh[st], xl[st]);
    }
  };
  // LDS addresses (floats) of the wave's own tiles k = 0..3 inside a [ct][lane] f32x4 image, and of its bias floats
  auto own_ct = [&](int k) __attribute__((always_inline)) -> int { return 8 * (k >> 1) + 2 * fq + (k & 1); };
  // start of a sub-block: out = [own channels of the row] + output bias (or the bias alone)
  auto start_acc = [&](int off, bool keep_residual) __attribute__((always_inline)) {
    const float* p = reinterpret_cast<const float*>(vec_b + vpar * VEC_BYTES) + off + 4 * g;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 b = *reinterpret_cast<const float4*>(p + 16 * own_ct(k));
      const f32x4 bb = f32x4{b.x, b.y, b.z, b.w};
      out[k] = keep_residual ? out[k] + bb : bb;
    }
  };
  // end of a sub-block: every wave writes its tiles (positions `pos(k)`) into the first scratch tile that follows the sub-block
  // in the stream and reads the whole row back; the second scratch tile only carries the next sub-block's vectors
  auto exchange = [&](auto pos) __attribute__((always_inline)) {
    f32x4* ex = reinterpret_cast<f32x4*>(slot_of(tau));
#pragma unroll
    for (int k = 0; k < 4; ++k) ex[pos(k) * 64 + lane] = out[k];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                    // B(scratch tile 1): every wave's channels are in LDS
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) xf[ct] = ex[ct * 64 + lane];
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = ex[own_ct(k) * 64 + lane];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    ++tau;
    __builtin_amdgcn_s_barrier();                    // B(scratch tile 2): the next sub-block's vectors have landed
    ++tau;
    vpar ^= 1;
  };
  // the four waves' 16-feature pieces (attention output / hidden chunk, f32x4 per lane) -> the 64-feature operand of the
  // output projection; the CALLER has written og[fq] and passed a barrier
  bf16x8 oh[2], ol[2];
  auto gather_o = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
      const f32x4 t0 = og[(2 * sp) * 64 + lane], t1_ = og[(2 * sp + 1) * 64 + lane];
      const float v[8] = {t0[0], t0[1], t0[2], t0[3], t1_[0], t1_[1], t1_[2], t1_[3]};
      split8_tf(v, oh[sp], ol[sp]);
    }
  };

  // ---- Transformer1d.to_in: GroupNorm(32 groups of 8 channels, over the sample's tokens) + Conv1d(k = 1) ----
  if (a.has_in) {
    float gm[NCT], gv[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) gm[ct] = xg16_add((xf[ct][0] + xf[ct][1]) + (xf[ct][2] + xf[ct][3]));   // lanes g, g ^ 1
    token_sum(gm);
    const float inv_n = 1.0f / (float)(8 * a.T);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      gm[ct] *= inv_n;
      float ss = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = xf[ct][r] - gm[ct];
        ss += d * d;
      }
      gv[ct] = xg16_add(ss);
    }
    token_sum(gv);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const float rs = __builtin_amdgcn_rsqf(gv[ct] * inv_n + a.eps_gn);
#pragma unroll
      for (int r = 0; r < 4; ++r) xf[ct][r] = (xf[ct][r] - gm[ct]) * rs;
    }
    make_operands(false);
    // projection chunk c (64 output channels, two K-half sub-tiles): this wave's 16 channels = tile 4 c + fq
    const float* bp = reinterpret_cast<const float*>(vec_b + vpar * VEC_BYTES) + 4 * g;
    auto in_chunk = [&](auto cc, auto o0, auto o1, auto more) __attribute__((always_inline)) {
      constexpr int c = decltype(cc)::value;
      f32x4 t[2] = {zero4, zero4};
      phase(kT, o0, kT, Yes{}, Yes{}, t, xh, xl);            // K half 0
      phase(kT, o1, kT, more, more, t, xh + 4, xl + 4);      // K half 1
      const float4 b = *reinterpret_cast<const float4*>(bp + 16 * (4 * c + fq));
      out[c] = (t[0] + t[1]) + f32x4{b.x, b.y, b.z, b.w};
    };
    begin_tile(kT);
    in_chunk(IC0{}, IC0{}, IC2{}, Yes{});
    in_chunk(IC1{}, IC1{}, IC0{}, Yes{});
    in_chunk(IC2{}, IC2{}, IC1{}, Yes{});
    in_chunk(std::integral_constant<int, 3>{}, IC0{}, IC2{}, No{});
    exchange([&](int k) { return 4 * k + fq; });
  }

  const int nheads = a.nheads, nff = a.nff;
  for (int blk = 0; blk < a.nblocks; ++blk) {
    const bool last_blk = blk + 1 == a.nblocks;
    const unsigned bias_l = 64u * (unsigned)fq + 16u * (unsigned)g;      // + parity base + 256 h: this wave's 16 features
    // ================= x += Attention(x) =================
    {
      make_operands(true);
      start_acc(64 * nheads, true);
      const unsigned bl = vec_l0 + vpar * VEC_BYTES + bias_l;
      for (int h = 0; h < nheads; ++h) {
        f32x4 qa[2] = {zero4, zero4}, ka[2] = {zero4, zero4}, va[2] = {zero4, zero4};
        begin_tile(kT);
        phase(kT, IC0{}, kT, Yes{}, Yes{}, qa, xh, xl);
        phase(kT, IC2{}, kT, Yes{}, Yes{}, qa, xh + 4, xl + 4);
        phase(kT, IC1{}, kT, Yes{}, Yes{}, ka, xh, xl);
        phase(kT, IC0{}, kN, Yes{}, Yes{}, ka, xh + 4, xl + 4);
        phase(kN, IC2{}, kN, Yes{}, Yes{}, va, xh, xl);
        phase(kN, IC1{}, kN, No{}, No{}, va, xh + 4, xl + 4);
        f32x4 bq;
        lds_read_f4_off<0>(bq, bl + 256 * h);
        lgkm_wait<0>();
        const f32x4 qT = (qa[0] + qa[1]) + bq, kTt = ka[0] + ka[1], vT = va[0] + va[1];
        f32x4 sp = zero4;
#pragma unroll
        for (int r = 0; r < 4; ++r) sp = MDT_MFMA_F32(kTt[r], qT[r], sp, 0, 0, 0);
        red[wave * 64 + lane] = sp;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                         // B(first output sub-tile) + partial exchange
        prefetch2(kO, slot_of(tau), 2);
        const f32x4 s01 = (red[lane] + red[64 + lane]) + (red[128 + lane] + red[192 + lane]);
        f32x4 st;
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float sv = s01[r] * scale2 + kmask[r];
          st[r] = sv;
          mx = fmaxf(mx, sv);
        }
        mx = xg16_max(mx);
        mx = xg32_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(st[r] - mx);
          st[r] = e;
          sum += e;
        }
        sum = xg16_add(sum);
        sum = xg32_add(sum);
        const float inv = __builtin_amdgcn_rcpf(sum);
        f32x4 oT = zero4;
#pragma unroll
        for (int r = 0; r < 4; ++r) oT = MDT_MFMA_F32(vT[r], st[r] * inv, oT, 0, 0, 0);
        og[wave * 64 + lane] = oT;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // (also: the partials above are read)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                         // B(second output sub-tile) + the four output pieces
        __builtin_amdgcn_sched_barrier(0);
        gather_o();
        phase(kO, IC2{}, kO, No{}, Yes{}, out, oh, ol);       // output channels 0..127 (this wave: 32 of them)
        phase(kO, IC1{}, kT, No{}, No{}, out + 2, oh, ol);    // output channels 128..255
      }
      exchange(own_ct);
    }
    // ================= x += Attention(x, context) =================
    if constexpr (NPW > 0) {
      make_operands(true);
      start_acc(64 * nheads, true);
      const unsigned bl = vec_l0 + vpar * VEC_BYTES + bias_l;
      for (int h = 0; h < nheads; ++h) {
        f32x4 qa[2] = {zero4, zero4};
        begin_tile(kT);
        phase(kT, IC0{}, kT, Yes{}, Yes{}, qa, xh, xl);
        phase(kT, IC2{}, kT, No{}, No{}, qa, xh + 4, xl + 4);
        __builtin_amdgcn_s_barrier();                         // B(K tile)
        const unsigned char* sk = slot_of(tau);
        f32x4 bq;
        lds_read_f4_off<0>(bq, bl + 256 * h);
        lgkm_wait<0>();
        const f32x4 qT = (qa[0] + qa[1]) + bq;
        f32x4 st[KTM];
        float4 kq[KTM];
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt) {
          const int R = min(16 * kt + i, nkeys - 1);
          kq[kt] = *reinterpret_cast<const float4*>(sk + R * 256 + (((4 * fq + g) ^ (R & 15)) << 4));
        }
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt) st[kt] = zero4;
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt) st[kt] = MDT_MFMA_F32(kq[kt].x, qT[0], st[kt], 0, 0, 0);
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt) st[kt] = MDT_MFMA_F32(kq[kt].y, qT[1], st[kt], 0, 0, 0);
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt) st[kt] = MDT_MFMA_F32(kq[kt].z, qT[2], st[kt], 0, 0, 0);
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt) st[kt] = MDT_MFMA_F32(kq[kt].w, qT[3], st[kt], 0, 0, 0);
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt) red[(kt * 4 + wave) * 64 + lane] = st[kt];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ++tau;
        __builtin_amdgcn_s_barrier();                         // B(V tile) + partial exchange
        const unsigned char* sv = slot_of(tau);
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt) {
          const f32x4* rp = red + kt * 4 * 64 + lane;
          const f32x4 s01 = (rp[0] + rp[64]) + (rp[128] + rp[192]);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float sv2 = ((okbits >> (4 * kt + r)) & 1u) ? s01[r] * scale2 : -INFINITY;
            st[kt][r] = sv2;
            mx = fmaxf(mx, sv2);
          }
        }
        mx = xg16_max(mx);
        mx = xg32_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(st[kt][r] - mx);
            st[kt][r] = e;
            sum += e;
          }
        sum = xg16_add(sum);
        sum = xg32_add(sum);
        const float inv = __builtin_amdgcn_rcpf(sum);
        f32x4 vq[KTM];
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int R = min(16 * kt + 4 * g + r, nkeys - 1);
            vq[kt][r] = *reinterpret_cast<const float*>(sv + R * 256 + (i & 3) * 4 + (((4 * fq + (i >> 2)) ^ (R & 15)) << 4));
          }
        f32x4 oT = zero4;
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) oT = MDT_MFMA_F32(vq[kt][r], st[kt][r] * inv, oT, 0, 0, 0);
        og[wave * 64 + lane] = oT;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // V reads complete before the slot can be refilled
        ++tau;
        __builtin_amdgcn_s_barrier();                         // B(first output sub-tile) + the four output pieces
        prefetch2(kO, slot_of(tau), 2);
        gather_o();
        phase(kO, IC2{}, kO, Yes{}, Yes{}, out, oh, ol);
        phase(kO, IC1{}, kT, No{}, No{}, out + 2, oh, ol);
      }
      exchange(own_ct);
    }
    // ================= x += FeedForward(x)  (last block: the closing convolution folded in) =================
    {
      const int npost = last_blk ? a.npost : 0;
      make_operands(false);
      start_acc(64 * nff, npost == 0);               // folded closing convolution: no residual (Wout x rides as tiles)
      const unsigned bl = vec_l0 + vpar * VEC_BYTES + bias_l;
      for (int h = 0; h < nff; ++h) {
        f32x4 ha[2] = {zero4, zero4};
        begin_tile(kT);
        phase(kT, IC0{}, kT, Yes{}, Yes{}, ha, xh, xl);       // K half 0
        phase(kT, IC2{}, kT, No{}, No{}, ha, xh + 4, xl + 4); // K half 1
        f32x4 b1;
        lds_read_f4_off<0>(b1, bl + 256 * h);
        lgkm_wait<0>();
        f32x4 hT = (ha[0] + ha[1]) + b1;
#pragma unroll
        for (int r = 0; r < 4; ++r) hT[r] = gelu_tf(hT[r]);
        og[wave * 64 + lane] = hT;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                         // B(first W2 sub-tile) + the four hidden pieces
        prefetch2(kO, slot_of(tau), 2);
        gather_o();
        phase(kO, IC2{}, kO, Yes{}, Yes{}, out, oh, ol);
        phase(kO, IC1{}, kO, No{}, No{}, out + 2, oh, ol);
      }
      if (npost > 0) {
        // + Wout x: per 64-channel k chunk kc of the raw-x operands (k-steps 2 kc, 2 kc + 1) two output sub-tiles
        begin_tile(kO);
        phase(kO, IC0{}, kO, Yes{}, Yes{}, out, xh, xl);          phase(kO, IC2{}, kO, Yes{}, Yes{}, out + 2, xh, xl);
        phase(kO, IC1{}, kO, Yes{}, Yes{}, out, xh + 2, xl + 2);  phase(kO, IC0{}, kO, Yes{}, Yes{}, out + 2, xh + 2, xl + 2);
        phase(kO, IC2{}, kO, Yes{}, Yes{}, out, xh + 4, xl + 4);  phase(kO, IC1{}, kO, Yes{}, Yes{}, out + 2, xh + 4, xl + 4);
        phase(kO, IC0{}, kO, Yes{}, Yes{}, out, xh + 6, xl + 6);  phase(kO, IC2{}, kT, No{}, No{}, out + 2, xh + 6, xl + 6);
      }
      exchange(own_ct);
    }
  }

  // ---- the residual stream leaves the kernel once: every wave stores its own 64 channels of its rows ----
  if (mvalid) {
    float* xo = a.out + (int64_t)m * C + 4 * g;
#pragma unroll
    for (int k = 0; k < 4; ++k) store_nt(xo + 16 * own_ct(k), make_float4(out[k][0], out[k][1], out[k][2], out[k][3]));
  }
}

template <int NPW>
static hipError_t launch_tf2q(const TFArgs& a, hipStream_t s) {
  const size_t smem = (size_t)NS * SLOT + RED_BYTES + 2 * VEC_BYTES + OG_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tf256q<NPW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((k_tf256q<NPW>), dim3((unsigned)((a.M + 15) / 16)), dim3(512), smem, s, a);
  return hipGetLastError();
}

hipError_t launch_tf256q(const TFArgs& a, hipStream_t s) {
  if (a.M <= 0) return hipSuccess;
  const bool cross = a.kv != nullptr;
  if (!tf256_supported(a.T, a.Tk, a.nheads, a.nff, cross) || a.nblocks <= 0 || a.NT <= 0) return hipErrorInvalidValue;
  if (a.npost != 0 && a.npost != 8) return hipErrorInvalidValue;
  if (!cross) return launch_tf2q<0>(a, s);
  switch (((16 / a.T) * a.Tk + 15) / 16) {
    case 1: return launch_tf2q<1>(a, s);
    case 2: return launch_tf2q<2>(a, s);
    case 3: return launch_tf2q<3>(a, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace mdt
