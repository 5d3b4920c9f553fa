// A whole Transformer1d (modules.py:469-524) of a C = 256 level in ONE launch (MDT_OP_TF256), 32-row workgroups:
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
//
// NSPLIT = 2 (the form for batches that do not fill the chip with 32-row workgroups, e.g. 128 row blocks at B = 1024): the
// heads / hidden chunks / to_in output chunks of a row block are split over a PAIR of workgroups, each with its own tile
// descriptor table (tiles + hh * NT) over the shared weight stream.  Both keep the whole residual row; at the end of every
// sub-block, behind the wave pairs' exchange, the two workgroups HAND their 32 x 256 partial sums TO EACH OTHER inside the
// launch and add them in a fixed order (half 0 + half 1), so both hold bitwise identical rows -- no partial-sum tensors between
// launches, one launch per Transformer1d instead of one per sub-block.  The hand-off is the form MI355X_MICROARCH.md lists as
// measured-valid for any placement of the two workgroups ("Hand-offs measured with sc1 loads", first row): 16-byte sc1 stores of
// whole 128-byte lines, every storing wave s_waitcnt vmcnt(0), workgroup barrier, ONE lane's sc1 flag store; ONE wave polls the
// partner's flag with sc1 loads, workgroup barrier, 16-byte sc1 loads.  tools/ubench/pair_handoff.hip: 2.2 us per 32 KB round
// for partners on one XCD (workgroup ids 8 apart), 3.8 us on different XCDs, 0 stale pieces in 2 x 10^8 checks either way under
// uneven load; correctness never depends on the placement, PAIR_STRIDE only picks the fast one.  The buffers are double-
// buffered by the hand-off count of the launch, the flags count hand-offs monotonically across launches (both partners always
// make the same number), a poll gives up after ~0.3 s and raises xflags[0] instead of hanging the GPU.  The stores / loads go
// through __builtin_amdgcn_raw_buffer_* (aux = sc1): hipcc tracks their waits and the store-data hazard (an inline-asm
// global_store_dwordx4 followed within two wait states by a VALU write to its data registers stores garbage: first version of
// the probe).
#include <cstdlib>
#include <type_traits>

#include "mdt_kernels.h"

// Compiled twice: as is (split-bf16 products, launch_tf256) and through k_tf256_f32.hip with MDT_TF_F32 = 1 (exact fp32 MFMA
// products, launch_tf256_f32) -- two translation units that build in parallel.
#ifndef MDT_TF_F32
#define MDT_TF_F32 0
#endif

// cache policy of the K / V row DMA (read once per evaluation, 1 GB in all): 2 = nt (streaming)
#ifndef MDT_KV_CPOL
#define MDT_KV_CPOL 2
#endif

// ring slot of tile t (run-time t): a mask, not the signed modulo (7 scalar instructions per use)
#ifdef MDT_SLOT_MOD
#define MDT_SLOT_IDX(t) ((t) % NS)
#else
#define MDT_SLOT_IDX(t) ((t) & (NS - 1))
#endif

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

__device__ __forceinline__ float gelu_tf(float x) {   // exact-erf GELU, branch-free erf (A&S 7.1.26, |error| < 1.5e-7)
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erfa = 1.0f - poly * __expf(-z * z);
  return 0.5f * x * (1.0f + copysignf(erfa, x));
}

// 8 values of one k-step -> its two 128-bit operand registers: bf16 hi / lo planes, or (F32) the values themselves, slots
// 0..3 in `hi`, 4..7 in `lo` (k_tf128.hip)
template <bool F32>
__device__ __forceinline__ void split8_tf(const float v[8], bf16x8& hi, bf16x8& lo) {
  if constexpr (F32) {
    hi = __builtin_bit_cast(bf16x8, f32x4{v[0], v[1], v[2], v[3]});
    lo = __builtin_bit_cast(bf16x8, f32x4{v[4], v[5], v[6], v[7]});
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const __bf16 h = (__bf16)v[e];
      hi[e] = h;
      lo[e] = (__bf16)(v[e] - (float)h);
    }
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
#ifdef MDT_ABL_LGKM   // ablation (WRONG results, timing only): no wait for fragment reads -- what the waits cost
  __builtin_amdgcn_sched_barrier(0);
  return;
#endif
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
constexpr int NU = 4;           // units (4 fragment reads + 6 MFMAs) per sub-tile per wave
constexpr int KTM = 3;          // key tiles per wave (cross): at most 48 context rows per 16 token rows
constexpr int RED_BYTES = KTM * 4 * 64 * 16;   // partial S^T exchange [key tile][4 waves][64 lanes] f32x4
constexpr int VEC_FLOATS = 768;                // vectors of one sub-block: [bq 512 | bo 256], [b1 512 | b2 256], [b_in 256]
constexpr int VEC_BYTES = VEC_FLOATS * 4;
// raw_buffer_* cache policies of the hand-off.  MDT_XH_MODE (tuning builds): 0 = sc1 stores, sc1 loads (the guide's measured row);
// 1 = sc0 sc1 (system scope) both sides; 2 = sc1 stores, sc0 sc1 loads (what the volatile bit gives)
#ifndef MDT_XH_MODE
#define MDT_XH_MODE 0
#endif
#ifdef MDT_ABL_BAR            // ablation (WRONG results, timing only): no workgroup barrier anywhere -- what the per-tile synchronisation costs
#define MDT_BARRIER() __builtin_amdgcn_sched_barrier(0)
#else
#define MDT_BARRIER() __builtin_amdgcn_s_barrier()
#endif
#ifndef MDT_STREAM_PF
#define MDT_STREAM_PF 8      // tiles ahead of the one being consumed at which ONE workgroup per (XCD, half) touches the weight stream's
#endif                       // lines (0: off).  An evaluation moves ~4 GB through the 256 MB Infinity Cache between two uses of a layer, so
                             // every launch streams its weights from HBM: 240 against 208 us per 4-block launch (tools/pair_launch_outliers.py)
#ifndef MDT_RING_AHEAD
#define MDT_RING_AHEAD 2     // tiles the loader waves run ahead of the one being consumed (2 or 3; NS = 4 slots)
#endif
#ifndef MDT_XH_ADDR
#define MDT_XH_ADDR 1        // 1: whole address in the VECTOR offset (scalar offset 0), as tools/ubench/pair_handoff.hip does
#endif
constexpr int AUX_ST = MDT_XH_MODE == 1 ? 17 : (MDT_XH_MODE == 3 ? 18 : (MDT_XH_MODE == 4 ? 19 : 16));   // 18 = sc1 nt, 19 = sc0 sc1 nt
constexpr int AUX_LD = MDT_XH_MODE == 0 ? 16 : (MDT_XH_MODE == 3 ? 18 : (MDT_XH_MODE == 4 ? 19 : 17));
constexpr unsigned XBLOCK = 32 * C * 4;            // bytes one workgroup hands over per round
constexpr unsigned long long POLL_TIMEOUT = 30000000ull;   // s_memrealtime ticks (100 MHz): 0.3 s

}  // namespace

// NPW: LDS-DMA pieces per loader wave per K / V tile = ceil(context rows of the workgroup / 16); 0 = no cross-attention
// NSPLIT: 1 = one workgroup per 32-row block; 2 = a pair of workgroups per block (see the head of the file)
// F32: fp32 fragment sub-tiles and exact fp32 MFMA products (v_mfma_f32_16x16x4_f32), as k_tf128.hip: a [64][128] projection
//      sub-tile is fragments (feature tile ft, k-step st, half lo) at ft * 8192 + st * 2048 + lo * 1024, a [128][64] output
//      sub-tile (row tile ct, k-step sp, half lo) at ct * 4096 + sp * 2048 + lo * 1024; lane (i, g) float r of a fragment =
//      W[16 rt + i][k-slot 32 st + 8 g + 4 lo + r].  The loader waves copy such a sub-tile linearly.
template <int NPW, int NSPLIT, bool F32>
__global__ __launch_bounds__(512) void k_tf256(TFArgs a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* red_b = smem + NS * SLOT;
  unsigned char* vec_b = red_b + RED_BYTES;          // two parities of VEC_BYTES

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NT = a.NT;
  const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.w);
  // row block and half of this workgroup: linear id = (group, half, member) with PAIR_STRIDE members per group, so partners are
  // PAIR_STRIDE ids apart (8: same XCD under the observed round-robin placement; 1: neighbours on different XCDs)
  int rb = blockIdx.x, hh = 0;
  const int nrb = (a.M + 31) / 32;
  if constexpr (NSPLIT == 2) {
    const int S = a.pair_stride, id = blockIdx.x;
    const int grp = id / (2 * S), w = id - grp * 2 * S;
    hh = w / S;
    rb = a.rb_base + grp * S + (w - hh * S);        // (rb_base: first row block of this launch, see launch_tf2)
    if (rb >= nrb) return;                          // padding of the last group: both partners leave
  }

  if (wave >= 4) {
    // ================= loader waves (k_tblock32.hip, descriptor-driven as k_tf128.hip) =================
    const int iw = wave - 4;
    __builtin_amdgcn_s_setprio(MDT_LOADER_PRIO);
    const cu32p tiles = (cu32p)(a.tiles + (NSPLIT == 2 ? hh * NT : 0));   // kind (3 bits) | aux << 3
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
      voffP[q] = F32 ? (unsigned)(inst * 1024 + lane * 16) : (unsigned)(U * (2 * CS) + ((xP ^ (U & 15)) << 4) + baseP);
      voffO[q] = F32 ? (unsigned)(inst * 1024 + lane * 16)
                     : (unsigned)(((inst * 8) / CS) * (128 * CS) + ((inst * 8) % CS) * 128 + ((xO ^ (4 * (inst & 1))) << 4) + baseO);
    }
    const int sample0 = rb * (32 / a.T);
    const bool second = a.kv2 && sample0 >= a.nsamples / 2;      // dual batch: shared K / V rows for the second half
    unsigned voffKV[8];
    if constexpr (NPW > 0) {
      const int kv_rows = (32 / a.T) * a.Tk;
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
      unsigned char* slot = smem + MDT_SLOT_IDX(tau) * SLOT + iw * 1024;
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
            __builtin_amdgcn_global_load_lds(base + voffKV[q], (__attribute__((address_space(3))) void*)(slot + q * 4096), 16, 0, MDT_KV_CPOL);
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
#ifdef MDT_ABL_VMWAIT   // ablation (WRONG results, timing only): a tile is published whether or not its DMA has landed
      return;
#endif
      switch (allow) {
#define MDT_VMW(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
        MDT_VMW(0) MDT_VMW(1) MDT_VMW(2) MDT_VMW(3) MDT_VMW(4) MDT_VMW(5) MDT_VMW(6) MDT_VMW(7) MDT_VMW(8) MDT_VMW(9) MDT_VMW(10)
        MDT_VMW(11) MDT_VMW(12) MDT_VMW(13) MDT_VMW(14) MDT_VMW(15)
#undef MDT_VMW
        default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
      }
    };
    // Tile k + AHEAD is issued behind B(k): the compute waves are past tile k - 1 there, and with AHEAD = 3 its slot is tile
    // k - 1's (NS = 4).  The only accesses to that slot that may still be in flight are fragment reads issued BEFORE the barrier
    // (phase(): units 2, 3 of a tile are read during units 0, 1), which the compute waves drain before they arrive (MDT_RING_AHEAD
    // == 3 adds that wait to phase()).
    constexpr int AHEAD = MDT_RING_AHEAD;
    unsigned dq[AHEAD];                                                  // descriptors of tiles k + 1 .. k + AHEAD - 1 (+ one scratch)
    MDT_BARRIER();   // P: the compute waves' row loads are queued ahead of the stream
    issue_vec(0u);                  // the first sub-block's vectors (parity 0), ahead of tile 0: covered by the first wait
#pragma unroll
    for (int j = 0; j < AHEAD; ++j) {
      const unsigned d = j < NT ? tiles[j] : 0u;
      if (j < NT) issue_tile(j, d);
      if (j > 0) dq[j - 1] = d;
    }
    // L2 prefetch of the stream (MDT_STREAM_PF): workgroup ids 8 (16 with the pair split) apart share an XCD and a half under the
    // observed round-robin placement (speed only); behind tile k + AHEAD's DMA each loader wave of the workgroup on duty
    // touches 64 lines of weight tile k + MDT_STREAM_PF with one dword load into a register nothing else uses.  That load counts in
    // vmcnt like the DMA pieces and returns in order, so the counted waits below allow for the ones issued in the last two turns.
    constexpr bool PF_ON = MDT_STREAM_PF > 0 && AHEAD == 2;
    // the duty rotates: of the workgroups of an (XCD, half) -- ids 8 NSPLIT apart -- number g takes the tiles with index % count == g,
    // so that no workgroup's own stream pays for more than its share of the touches
    const int pf_groups = PF_ON ? max(1, (int)gridDim.x / (8 * NSPLIT)) : 1;
    const int pf_mine = (int)blockIdx.x / (8 * NSPLIT);
    int pf_turn = MDT_STREAM_PF % pf_groups;                             // (k + MDT_STREAM_PF) % pf_groups, kept without a division per tile
    unsigned char* pf_sink_b = vec_b + 2 * VEC_BYTES;                     // 1 KB behind the vector areas (launch_tf2 sizes it)
    int pf1 = 0, pf2 = 0;                                                // prefetch loads issued one / two turns ago
    for (int k = 0; k < NT; ++k) {
      const unsigned dnew = k + AHEAD < NT ? tiles[k + AHEAD] : 0u;
      int allow = 0;
#pragma unroll
      for (int j = 0; j < AHEAD - 1; ++j) allow += k + 1 + j < NT ? pieces_of(dq[j]) : 0;
      if (PF_ON && k + 1 < NT) allow += pf1 + pf2;
      // tile k landed; tiles k + 1 .. may be in flight.  Round 5: the common cases -- a weight sub-tile or a K / V tile next, no
      // touch of this workgroup in flight -- are tested first: as compiled, the 17-way switch of wait_vm is a cascade of ~27
      // scalar branches, ~300 cycles of a loader turn (stamps of the same loader in k_res256.hip, where the loader IS what a
      // sub-tile waits for: 1100 -> 890 cycles per sub-tile).  Here the compute waves arrive at the barrier last: same-box A/B of
      // this change on the headline 4506 / 4487 -> 4519 / 4526 molecules/s, evaluation 1.766 ms either way (profiles/r5_res256_ab.txt)
      if (allow == IPT) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (NPW > 0 && allow == NPW) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW > 0 ? NPW : 1) : "memory");      // a K / V tile next
      else wait_vm(allow);
      MDT_BARRIER();                                      // B(k)
      if (k + AHEAD < NT) issue_tile(k + AHEAD, dnew);
      if constexpr (PF_ON) {
        pf2 = pf1;
        pf1 = 0;
        const bool mine = pf_turn == pf_mine;
        if (++pf_turn == pf_groups) pf_turn = 0;
        if (mine && k + MDT_STREAM_PF < NT) {
          const unsigned dp = tiles[k + MDT_STREAM_PF];
          const unsigned kp = dp & 7u;
          if (kp != D_SCRATCH && kp != D_SCRATCH_VEC && kp < D_K) {      // a weight sub-tile
            const unsigned char* t = wsrc + (int64_t)(dp >> 3) * SLOT + (iw * 64 + lane) * 128;
            // one dword per lane by LDS-DMA into a sink nothing reads: no landing REGISTER at all (an inline-asm global_load with
            // a live "+v" register is the hazard class of mdt_kernels.h: prefetch_next_weights, and a load whose value the
            // compiler can see is waited for with vmcnt(0) on the spot -- HBM latency in the loader's per-tile turn).  Counts in
            // vmcnt like the DMA pieces and completes in order with them.
            __builtin_amdgcn_global_load_lds(t, (__attribute__((address_space(3))) void*)(pf_sink_b + iw * 256), 4, 0, 0);
            pf1 = 1;
          }
        }
      }
#pragma unroll
      for (int j = 0; j + 1 < AHEAD - 1; ++j) dq[j] = dq[j + 1];
      dq[AHEAD - 2] = dnew;
    }
    prefetch_next_weights(a.pf_ptr, a.pf_lines, iw * 64 + lane);
    return;
  }

  // ================= compute waves =================
  const int i = lane & 15, g = lane >> 4;
  const int rt = wave >> 1, fh = wave & 1;
  const int row0 = rb * 32 + rt * 16;
  const int m = row0 + i;
  const bool mvalid = m < a.M;
  const int mc = mvalid ? m : a.M - 1;
  f32x4* red = reinterpret_cast<f32x4*>(red_b);

  // the residual stream: accT[ct][r] = x[row i][16 ct + 4 g + r]; both waves of a row tile hold the whole row
  f32x4 accT[NCT];
#ifdef MDT_STAMPS   // tuning build: wave 0 of the two workgroups of row block 0 record (source line << 48 | shader clock) at phase
                    // boundaries behind the flag lines: xflags[64 + 64 nrb + 512 hh ...] (tools/tf256_bench.py allocates the room)
  unsigned long long* stamps = reinterpret_cast<unsigned long long*>(a.xflags + 64 + 64 * nrb + 512 * hh);
  int nstamp = 0;
#define MDT_STAMP()                                                                                           \
  do {                                                                                                        \
    if (NSPLIT == 2 && rb == 0 && wave == 0 && nstamp < 250) {                                                \
      unsigned long long t_;                                                                                  \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                              \
      if (lane == 0) stamps[nstamp] = (t_ & 0xffffffffffffull) | ((unsigned long long)__LINE__ << 48);        \
      ++nstamp;                                                                                               \
    }                                                                                                         \
  } while (0)
#else
#define MDT_STAMP() do {} while (0)
#endif
  // pair hand-off state (NSPLIT == 2): flag words xflags[64 + 32 * (2 rb + hh)] (one 128-byte line each; xflags[0..63] are
  // diagnostics), hand-off blocks xbuf[parity][rb][hh][row tile][accumulator tile][lane] f32x4
#ifdef MDT_XH_LOG
  unsigned xlog_first = 0, xlog_last = 0, xlog_got = 0, xlog_epoch = 0;
#endif
  unsigned xround = 0;                               // my flag's value = hand-offs I have completed, ever
  int xn = 0;                                        // hand-offs of this launch (buffer parity)
  __amdgpu_buffer_rsrc_t xres, fres;
  const unsigned fown = (64u + 32u * (unsigned)(2 * rb + hh)) * 4u;
  if constexpr (NSPLIT == 2) {
    xres = __builtin_amdgcn_make_buffer_rsrc(a.xbuf, 0, 0x7fffffff, 0x00020000);
    fres = __builtin_amdgcn_make_buffer_rsrc(a.xflags, 0, 0x7fffffff, 0x00020000);
    xround = (unsigned)__builtin_amdgcn_readfirstlane(__builtin_amdgcn_raw_buffer_load_b32(fres, fown, 0, AUX_LD));   // written by an earlier LAUNCH
  }
#ifdef MDT_XH_LOG
  xlog_epoch = xround;
#endif
  {
    const float* xp = a.x + (int64_t)mc * C + 4 * g;
    float4 xr[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) xr[ct] = *reinterpret_cast<const float4*>(xp + 16 * ct);
    __builtin_amdgcn_sched_barrier(0);
    MDT_BARRIER();                    // P
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) accT[ct] = f32x4{xr[ct].x, xr[ct].y, xr[ct].z, xr[ct].w};
  }

  // fragment addressing inside a sub-tile (k_tblock32.hip), this wave's feature half folded in
  // aP(st) = fh * 16384 + i * 512 + ((4 st + g) ^ i) * 16 = aP0 ^ (64 st): ONE register instead of four (4 st only touches bits 6-7 of
  // the address, which nothing else carries into)
  // F32: fragment tiles -- the lane's 16 bytes, this wave's feature tiles (projection: ft = 2 fh + q) or k-step (output: sp = fh)
  // and the k-step of a projection unit in the base, the rest in the immediate
  const int aP0 = F32 ? lane * 16 + fh * 16384 : fh * (2 * 16 * 4 * CS) + i * (4 * CS) + ((g ^ i) & 15) * 16;
  auto aP = [&](int st) -> int { return F32 ? aP0 : aP0 ^ (64 * st); };          // (F32: the k-step is in the immediate too)
  const int aO = F32 ? lane * 16 + fh * 2048 : i * 128 + ((4 * fh + g) ^ ((i >> 1) & 7)) * 16;

  bf16x8 frh[3][2], frl[3][2];
  auto frag_read = [&](auto kind, unsigned base, auto uc, int set, auto jc) {
    constexpr int KIND = decltype(kind)::value, u = decltype(uc)::value, j = decltype(jc)::value;
    constexpr int q = j >> 1, lo = j & 1;
    constexpr int off = F32 ? ((KIND == K_O) ? ((2 * u + q) * 4096 + lo * 1024) : (q * 8192 + u * 2048 + lo * 1024))
                            : ((KIND == K_O) ? ((2 * u + q) * 16 * 128 + lo * (CS * 128)) : (q * 16 * 4 * CS + lo * (2 * CS)));
#ifdef MDT_ABL_LDSBC   // ablation (WRONG results, timing only): every lane reads the same 16 bytes -- what the fragment reads cost the LDS
    lds_read16_off<off>(lo ? frl[set][q] : frh[set][q], base & 0x18000u);
#else
    if constexpr (F32) lds_read16_off<off>(frh[set][q], base);            // (exact fp32: one pair per set, see phase())
    else lds_read16_off<off>(lo ? frl[set][q] : frh[set][q], base);
#endif
  };
  using J0 = std::integral_constant<int, 0>;
  using J1 = std::integral_constant<int, 1>;
  using J2 = std::integral_constant<int, 2>;
  using J3 = std::integral_constant<int, 3>;
  // LDS addresses of the fragment reads of units 2 / 3 of the NEXT phase(): computed in the shadow of the running phase's last MFMAs
  // (or by prefetch2 where the pipeline restarts) -- address arithmetic between two phases is exposed issue time, ~5 cycles an instruction
  unsigned pb2 = 0, pb3 = 0;
  auto prefetch2 = [&](auto kind, const unsigned char* slot, int off) {
    constexpr int KIND = decltype(kind)::value;
    const unsigned l = lds_addr(slot);
    if constexpr (F32) {                             // two sets of ONE fragment pair, half a unit ahead (phase()): half-unit 0 -> set 0
      const unsigned b = l + (KIND == K_O ? aO : aP0);
      frag_read(kind, b, J0{}, 0, J0{}); frag_read(kind, b, J0{}, 0, J2{});
      return;
    }
    const unsigned b0 = l + (KIND == K_O ? aO : aP(0)), b1 = l + (KIND == K_O ? aO : aP(1));
    pb2 = l + (KIND == K_O ? aO : aP(2));
    pb3 = l + (KIND == K_O ? aO : aP(3));
    frag_read(kind, b0, J0{}, off % 3, J0{}); frag_read(kind, b0, J0{}, off % 3, J1{});
    frag_read(kind, b0, J0{}, off % 3, J2{}); frag_read(kind, b0, J0{}, off % 3, J3{});
    frag_read(kind, b1, J1{}, (off + 1) % 3, J0{}); frag_read(kind, b1, J1{}, (off + 1) % 3, J1{});
    frag_read(kind, b1, J1{}, (off + 1) % 3, J2{}); frag_read(kind, b1, J1{}, (off + 1) % 3, J3{});
  };

  int tau = 0;
  auto slot_of = [&](int t) -> unsigned char* { return smem + MDT_SLOT_IDX(t) * SLOT; };
  // An opaque copy of the lane id.  Addresses that are needed once per sub-block or head (vector rows, the exchange areas, the
  // hand-off blocks, the final store) are formed from it WHERE THEY ARE USED: computed once at the top of the kernel from the
  // real lane id they are live for the whole launch and -- this kernel sits at the 256-register limit -- are what hipcc spills
  // after to_in and reloads in front of every use (up to 120 bytes of scratch per lane).
  // (The exact-fp32 instantiations have 16 registers to spare -- two fragment sets -- and compile to zero scratch with these
  //  values hoisted; made opaque there, the same code spills 76-136 bytes around the S^T MFMAs.  So: opaque for split-bf16 only.)
  auto lane_now = [&]() -> int {
    if constexpr (F32) return lane;
    // read from the hardware (v_mbcnt: the number of lanes below this one), not from `lane`: threadIdx.x then has no late use
    // either, and it was the last register spilled across the block loop
    int l = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(l));
    return l;
  };

  // One MFMA phase over a sub-tile (k_tblock32.hip): 4 units of 4 fragment reads + 6 MFMAs
  auto phase = [&](auto kind, auto offc, auto nkind, bool has_next, f32x4* acc, const bf16x8* bh, const bf16x8* bl) {
    constexpr int KIND = decltype(kind)::value, OFF = decltype(offc)::value, NK = decltype(nkind)::value;
    if constexpr (F32) {
      // Exact fp32 products: a fragment pair (the two feature tiles q of one operand half) feeds 8 MFMAs of 8 passes = 256
      // MFMA-pipe cycles, so HALF a unit of read-ahead covers the LDS latency: the pipeline keeps two sets of ONE pair in
      // flight (frh[set][q]: 16 registers against the 48 of the split form's three full sets -- this kernel sits at the
      // 256-register limit, and with the slack every fp32 instantiation compiles to zero scratch).  Half-unit v = 2 u + half
      // reads pair v + 1 into set (v + 1) & 1 between its MFMAs; the barrier that publishes the next sub-tile sits in front of
      // the last half-unit.  2 NU is even: every phase starts on set 0 and OFF is not used.  All fragment addresses are slot
      // base + one lane constant + immediate.
      const unsigned lc = lds_addr(slot_of(tau)) + (KIND == K_O ? aO : aP0);
      const unsigned lnx = lds_addr(slot_of(tau + 1)) + (NK == K_O ? aO : aP0);
      auto half32 = [&](auto vc) {
        constexpr int v = decltype(vc)::value, u = v >> 1, hl = v & 1;
        if (v == 2 * NU - 1 && has_next) {
          __builtin_amdgcn_sched_barrier(0);
          MDT_BARRIER();                // B(tau + 1)
          __builtin_amdgcn_sched_barrier(0);
        }
        lgkm_wait<0>();                 // pair v (read during the previous half-unit) has landed
        constexpr int s0 = v & 1, s1 = (v + 1) & 1;
        constexpr bool in_phase = v + 1 < 2 * NU;
        const bool pre = in_phase || has_next;
        constexpr int ia = (KIND == K_O) ? 2 * u : 0, ib = (KIND == K_O) ? 0 : u;
        auto rd = [&](auto qc) {        // fragment (q, half) of half-unit v + 1: read j = 2 q + half
          if (!pre) return;
          constexpr int q = decltype(qc)::value;
          __builtin_amdgcn_sched_barrier(0);
          constexpr int un = (v + 1) / 2, jn = 2 * q + ((v + 1) & 1);
          if constexpr (in_phase) frag_read(kind, lc, std::integral_constant<int, un>{}, s1, std::integral_constant<int, jn>{});
          else frag_read(nkind, lnx, std::integral_constant<int, 0>{}, s1, std::integral_constant<int, 2 * q>{});
          __builtin_amdgcn_sched_barrier(0);
        };
        // pair x operand half: four 16x16x4 MFMAs per feature tile (r = contraction sub-step), the two accumulators alternating
        const f32x4 a0 = __builtin_bit_cast(f32x4, frh[s0][0]), a1 = __builtin_bit_cast(f32x4, frh[s0][1]);
        const f32x4 xb = __builtin_bit_cast(f32x4, hl ? bl[ib] : bh[ib]);
        auto mm2 = [&](auto r0c) {
          constexpr int r0 = decltype(r0c)::value;
#pragma unroll
          for (int r = r0; r < r0 + 2; ++r) {
            if constexpr (KIND == K_N) {
              acc[ia] = MDT_MFMA_F32(xb[r], a0[r], acc[ia], 0, 0, 0);
              acc[ia + 1] = MDT_MFMA_F32(xb[r], a1[r], acc[ia + 1], 0, 0, 0);
            } else {
              acc[ia] = MDT_MFMA_F32(a0[r], xb[r], acc[ia], 0, 0, 0);
              acc[ia + 1] = MDT_MFMA_F32(a1[r], xb[r], acc[ia + 1], 0, 0, 0);
            }
          }
        };
        mm2(J0{}); rd(J0{});
        mm2(J2{}); rd(J1{});
        __builtin_amdgcn_sched_barrier(0);
      };
      half32(std::integral_constant<int, 0>{}); half32(std::integral_constant<int, 1>{});
      half32(std::integral_constant<int, 2>{}); half32(std::integral_constant<int, 3>{});
      half32(std::integral_constant<int, 4>{}); half32(std::integral_constant<int, 5>{});
      half32(std::integral_constant<int, 6>{}); half32(std::integral_constant<int, 7>{});
      ++tau;
      return;
    }
    unsigned ln = 0, bn[2] = {0u, 0u};
    auto unit = [&](auto uc) {
      constexpr int u = decltype(uc)::value;
      if (u == NU - 2 && has_next) {
        __builtin_amdgcn_sched_barrier(0);
#ifndef MDT_RING_NODRAIN   // (timing experiment: without the drain the refill races this tile's last fragment reads in principle)
        if constexpr (MDT_RING_AHEAD >= 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this tile's slot is refilled behind B
#endif
        MDT_BARRIER();                // B(tau + 1)
#ifdef MDT_STAGGER   // experiment: the feature-half-1 waves fall MDT_STAGGER x 16 cycles behind after every tile barrier
        if (fh) { for (int d_ = 0; d_ < MDT_STAGGER; ++d_) asm volatile("s_nop 15" ::: "memory"); }
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
      constexpr int s0 = (OFF + u) % 3, s2 = (OFF + u + 2) % 3;
      constexpr bool in_phase = u + 2 < NU;
      const bool pre = in_phase || has_next;
      const bool later = (u + 1 < NU) || has_next;
      if (later) lgkm_wait<4>(); else lgkm_wait<0>();
      constexpr int ia = (KIND == K_O) ? 2 * u : 0, ib = (KIND == K_O) ? 0 : u;
      auto rd = [&](auto jc) {
        if (!pre) return;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (in_phase) frag_read(kind, u == 0 ? pb2 : pb3, std::integral_constant<int, u + 2>{}, s2, jc);
        else frag_read(nkind, bn[u + 2 - NU], std::integral_constant<int, u + 2 - NU>{}, s2, jc);
        __builtin_amdgcn_sched_barrier(0);
      };
      auto mm = [&](const bf16x8& w, const bf16x8& x, int q) {
        if constexpr (KIND == K_N) acc[ia + q] = MDT_MFMA_BF16(x, w, acc[ia + q], 0, 0, 0);
        else acc[ia + q] = MDT_MFMA_BF16(w, x, acc[ia + q], 0, 0, 0);
      };
      mm(frl[s0][0], bh[ib], 0); rd(J0{});
      mm(frl[s0][1], bh[ib], 1); rd(J1{});
      mm(frh[s0][0], bl[ib], 0); rd(J2{});
      mm(frh[s0][1], bl[ib], 1); rd(J3{});
      mm(frh[s0][0], bh[ib], 0);
      if constexpr (u == 1) {                        // the next tile's slot: needed from unit 2 on
        __builtin_amdgcn_sched_barrier(0);
        ln = lds_addr(slot_of(tau + 1));
        bn[0] = ln + (NK == K_O ? aO : aP(0));
        bn[1] = ln + (NK == K_O ? aO : aP(1));
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (u == NU - 1) {                   // this phase's own reads are all issued: pb2 / pb3 move on to the next tile
        __builtin_amdgcn_sched_barrier(0);
        pb2 = ln + (NK == K_O ? aO : aP(2));
        pb3 = ln + (NK == K_O ? aO : aP(3));
        __builtin_amdgcn_sched_barrier(0);
      }
      mm(frh[s0][1], bh[ib], 1);
      __builtin_amdgcn_sched_barrier(0);
    };
    unit(std::integral_constant<int, 0>{}); unit(std::integral_constant<int, 1>{});
    unit(std::integral_constant<int, 2>{}); unit(std::integral_constant<int, 3>{});
    ++tau;
  };
  using IC0 = std::integral_constant<int, 0>;
  using IC1 = std::integral_constant<int, 1>;
  using IC2 = std::integral_constant<int, 2>;
  const IC0 kT{};
  const IC1 kN{};
  const IC2 kO{};

  // (the self-attention mask -- bit r: key 4 g + r belongs to this lane's sample -- is formed per head, see lane_now())
  const float scale2 = a.scale * 1.44269504088896340736f;
  // K / V rows of this wave inside a K / V tile: row R = Rw + 16 kt + (i | 4 g + r), 256 bytes, 16-byte chunks XOR-swizzled with
  // R & 15 (independent of kt): ONE base per operand, the key tile kt is an immediate offset of 4096 bytes and the second
  // 16-feature half is the base ^ 64.  Rows past the wave's keys (up to row 95 of the 128-row slot) hold other samples' rows or
  // older tiles -- finite fp32 bit patterns in either case (every slot is filled by weight sub-tiles before the first K / V
  // tile: bf16 pairs whose upper half is a finite bf16) -- and meet a probability of exactly 0.
  // These six lane constants (kb0, vb0[4], okbits) are RECOMPUTED at every cross-attention head from an opaque copy of the lane
  // id (cross_consts below, ~30 VALU instructions): kept live across the whole launch they were the registers hipcc spilled
  // after to_in and reloaded at every head (44 of the 120 bytes of scratch of the pair-split instantiations).
  const int nkeys = NPW > 0 ? (16 / a.T) * a.Tk : 0;
  const int tsh = __builtin_ctz((unsigned)a.T);      // T is a power of two dividing 16 (launch_tf256 checks)
  // bit r: key 4 g + r belongs to this lane's sample (self-attention mask)
  auto self_mask = [&](int l) -> unsigned {
    const int sq = (l & 15) >> tsh, g4 = 4 * (l >> 4);
    unsigned kb = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) kb |= (((g4 + r) >> tsh) == sq) ? (1u << r) : 0u;
    return kb;
  };
  auto cross_consts = [&](int l, unsigned& kb0, unsigned (&vb0)[4], unsigned& okbits) {
    const int i_ = l & 15, g_ = l >> 4;
    const int Rw = rt * nkeys;
    kb0 = (unsigned)((Rw + i_) * 256 + (((8 * fh + g_) ^ ((Rw + i_) & 15)) << 4));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int R = Rw + 4 * g_ + r;
      vb0[r] = (unsigned)(R * 256 + (i_ & 3) * 4 + (((8 * fh + (i_ >> 2)) ^ (R & 15)) << 4));
    }
    const int k_lo = (i_ >> tsh) * a.Tk, k_hi = k_lo + a.Tk;       // this lane's sample owns keys [k_lo, k_hi) of the wave's
    okbits = 0;
#pragma unroll
    for (int kt = 0; kt < KTM; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int jj = 16 * kt + 4 * g_ + r;
        if (jj >= k_lo && jj < k_hi) okbits |= 1u << (4 * kt + r);
      }
  };
  // The variables themselves live here; the split-bf16 instantiations ASSIGN them at every head (from the opaque lane id, so
  // that they are dead in between), the exact-fp32 ones once, now.  (No copies into per-head arrays: a register array that is
  // copied element-wise through a reference ends up in scratch memory -- 148 bytes per lane when this was tried.)
  unsigned kbits = 0, kb0 = 0, vb0[4] = {0, 0, 0, 0}, okbits = 0;
  if constexpr (F32) {
    kbits = self_mask(lane);
    if constexpr (NPW > 0) cross_consts(lane, kb0, vb0, okbits);
  }
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  const float t1 = a.T > 1 ? 1.f : 0.f, t2 = a.T > 2 ? 1.f : 0.f, t4 = a.T > 4 ? 1.f : 0.f, t8 = a.T > 8 ? 1.f : 0.f;
  auto dpp_fma = [](float v, float f, auto ctrl) {
    const int mm_ = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, true);
    return __builtin_fmaf(__builtin_bit_cast(float, mm_), f, v);
  };
  auto token_sum = [&](float (&s)[NCT]) {            // sums over the sample's token lanes (k_rconv.hip)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) s[ct] = dpp_fma(s[ct], t1, std::integral_constant<int, 0xB1>{});
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) s[ct] = dpp_fma(s[ct], t2, std::integral_constant<int, 0x4E>{});
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) s[ct] = dpp_fma(s[ct], t4, std::integral_constant<int, 0x141>{});
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) s[ct] = dpp_fma(s[ct], t8, std::integral_constant<int, 0x140>{});
  };

  MDT_STAMP();                                       // entry -> row loads issued, lane constants
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  MDT_STAMP();                                       // rows arrived
  MDT_BARRIER();                      // B(0)
  prefetch2(kT, slot_of(0), 0);
  const unsigned vec_l0 = lds_addr(vec_b);
  int vpar = 0;                                      // parity of the current sub-block's vector area

  bf16x8 xh[NST], xl[NST];
  auto make_operands = [&](bool layernorm) {
    float mean = 0.f, rstd = 1.f;
    if (layernorm) {
      float s = 0.f;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) s += (accT[ct][0] + accT[ct][1]) + (accT[ct][2] + accT[ct][3]);
      s = xg16_add(s);
      s = xg32_add(s);
      mean = s / (float)C;
      float ss = 0.f;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = accT[ct][r] - mean;
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
      for (int e = 0; e < 8; ++e) v[e] = mvalid ? (accT[2 * st + (e >> 2)][e & 3] - mean) * rstd : 0.f;
      split8_tf<F32>(v, xh[st], xl[st]);
    }
  };
  // start of a sub-block: wave fh = 0 carries residual + output bias (or the bias alone), wave fh = 1 starts from zero
  auto start_acc = [&](int off, bool keep_residual) {
    const float* p = reinterpret_cast<const float*>(vec_b + vpar * VEC_BYTES) + off + 4 * (lane_now() >> 4);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const float4 b = *reinterpret_cast<const float4*>(p + 16 * ct);
      const f32x4 bb = f32x4{b.x, b.y, b.z, b.w};
      const f32x4 mine = keep_residual ? accT[ct] + bb : bb;
      accT[ct] = (fh | hh) ? zero4 : mine;
    }
  };
  // end of a sub-block: the two feature-half waves of a row tile add their partial accumulators through the two scratch
  // tiles that follow the sub-block in the stream (8 accumulator tiles = 8 KB per wave and round); fixed order
  // (half 0 + half 1), so both waves end up with bitwise identical rows.  With `last` the barriers of the scratch tiles are
  // still executed (the loaders count them) but nothing follows.
  auto exchange_half = [&](auto rdc) {               // accumulator tiles 8 rd .. 8 rd + 7 (compile-time register indices)
    constexpr int rd_ = decltype(rdc)::value;
    f32x4* ex = reinterpret_cast<f32x4*>(slot_of(tau)) + lane_now();
#pragma unroll
    for (int c = 0; c < 8; ++c) ex[(wave * 8 + c) * 64] = accT[8 * rd_ + c];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    MDT_BARRIER();                    // B(scratch tile): the partner's partials (and the next vectors) are in LDS
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const f32x4 other = ex[((wave ^ 1) * 8 + c) * 64];
      accT[8 * rd_ + c] = fh ? (other + accT[8 * rd_ + c]) : (accT[8 * rd_ + c] + other);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // read before the slot can be refilled (two barriers later)
    ++tau;
  };
  // NSPLIT == 2: the two workgroups of the pair hand each other their 32 x 256 partial sums (head of the file); two
  // barrier-only tiles of the stream (descriptor kind 4) keep the loader waves in step.  Every wave holds its row tile's 16
  // accumulator tiles after the wave pairs' exchange; wave (rt, fh) publishes tiles 8 fh .. 8 fh + 7 and reads all 16 of the
  // partner's.
  auto pair_handoff = [&]() {
    if constexpr (NSPLIT == 2) {
      // wave-uniform part of the addresses in the scalar offset (an SGPR), the lane's 16 bytes in the vector offset
      const unsigned sbase = __builtin_amdgcn_readfirstlane(
          (((unsigned)(xn & 1) * (unsigned)nrb + (unsigned)rb) * 2u + (unsigned)hh) * XBLOCK + (unsigned)rt * (16u * 1024u));
      const unsigned vlane = (unsigned)lane_now() * 16u;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const f32x4 lo_ = accT[c], hi_ = accT[8 + c];
        const f32x4 v = f32x4{fh ? hi_[0] : lo_[0], fh ? hi_[1] : lo_[1], fh ? hi_[2] : lo_[2], fh ? hi_[3] : lo_[3]};
#if MDT_XH_ADDR
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), xres, vlane + sbase + (unsigned)(8 * fh + c) * 1024u, 0, AUX_ST);
#else   // the form that FAILS at model level (kept for the record, -DMDT_XH_ADDR=0): wave-uniform part in the scalar offset
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), xres, vlane, sbase + (unsigned)(8 * fh + c) * 1024u, AUX_ST);
#endif
      }
      MDT_STAMP();                                       // hand-off: stores issued
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave, in front of the barrier the flag store follows
      MDT_STAMP();                                       // ... drained
      MDT_BARRIER();                      // B(first hand-off tile)
      MDT_STAMP();                                       // ... every wave of the workgroup drained
      ++xround;
      if (wave == 0) {
        __builtin_amdgcn_raw_buffer_store_b32((int)xround, fres, fown, 0, AUX_ST);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
#ifdef MDT_XH_LOG
        int xtries = 0;
#endif
        for (;;) {
#ifdef MDT_XH_LOG
          ++xtries;
#endif
          asm volatile("" ::: "memory");           // a fresh load every turn (the builtin is not volatile: that bit would make it sc0 sc1)
          const unsigned got = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(fres, fown ^ 128u, 0, AUX_LD);
          if ((int)(got - xround) >= 0) {
#ifdef MDT_XH_LOG
            if (xtries == 1) ++xlog_first;
            xlog_got = got;
#endif
            break;
          }
          __builtin_amdgcn_s_sleep(2);
          if (__builtin_amdgcn_s_memrealtime() - t0 > POLL_TIMEOUT) {   // never hang the GPU: flag the launch and go on
            if (lane_now() == 0) atomicOr(a.xflags, 1u);
            break;
          }
        }
      }
      MDT_STAMP();                                       // ... partner's flag seen
#ifdef MDT_XH_D2       // stress build: 135 us between the poll and the loads (made EVERY call fail with the scalar-offset form)
      if (wave == 0) for (int d = 0; d < 40; ++d) __builtin_amdgcn_s_sleep(127);
#endif
      MDT_BARRIER();                      // B(second hand-off tile): the partner's block is complete
      asm volatile("" ::: "memory");
      const unsigned sother = sbase ^ XBLOCK;            // the same block of half hh ^ 1
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        f32x4 o[8];
#pragma unroll
        for (int c = 0; c < 8; ++c)
#if MDT_XH_ADDR
          o[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, vlane + sother + (unsigned)(8 * half + c) * 1024u, 0,
                                                                                 AUX_LD));
#else
          o[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, vlane, sother + (unsigned)(8 * half + c) * 1024u,
                                                                                 AUX_LD));
#endif
#pragma unroll
        for (int c = 0; c < 8; ++c) accT[8 * half + c] = hh ? (o[c] + accT[8 * half + c]) : (accT[8 * half + c] + o[c]);
      }
      ++xn;
      tau += 2;
#ifdef MDT_XH_LOG
      xlog_last = xround;
#endif
      MDT_STAMP();                                       // ... partner's block added
    }
  };
  auto exchange = [&]() {
    MDT_STAMP();                                         // sub-block streamed
    exchange_half(std::integral_constant<int, 0>{});
    exchange_half(std::integral_constant<int, 1>{});
    vpar ^= 1;
    MDT_STAMP();                                         // wave pairs exchanged
    pair_handoff();
  };
  auto next_subblock = [&]() {                       // first tile of the next sub-block: always a projection sub-tile
    MDT_BARRIER();                    // B(tau)
    prefetch2(kT, slot_of(tau), 0);
    MDT_STAMP();                                     // next sub-block's first tile there
  };

  // ---- Transformer1d.to_in: GroupNorm(32 groups of 8 channels, over the sample's tokens) + Conv1d(k = 1) ----
  if (a.has_in) {
    float gm[NCT], gv[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) gm[ct] = xg16_add((accT[ct][0] + accT[ct][1]) + (accT[ct][2] + accT[ct][3]));   // lanes g, g ^ 1
    token_sum(gm);
    const float inv_n = 1.0f / (float)(8 * a.T);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      gm[ct] *= inv_n;
      float ss = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = accT[ct][r] - gm[ct];
        ss += d * d;
      }
      gv[ct] = xg16_add(ss);
    }
    token_sum(gv);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const float rs = __builtin_amdgcn_rsqf(gv[ct] * inv_n + a.eps_gn);
#pragma unroll
      for (int r = 0; r < 4; ++r) accT[ct][r] = (accT[ct][r] - gm[ct]) * rs;
    }
    make_operands(false);
    // this wave produces output channels 64 c + 32 fh + 16 q + (4 g + r), i.e. accumulator tiles 4 c + 2 fh + q, complete
    // sums (both K halves); the other tiles stay zero and arrive through the exchanges; bias on wave 0 (of half 0) only
    const float* bp = reinterpret_cast<const float*>(vec_b + vpar * VEC_BYTES) + 4 * (lane_now() >> 4);
    if constexpr (NSPLIT == 1) {
      auto in_chunk = [&](auto cc, auto o0, auto o1, bool more) {
        constexpr int c = decltype(cc)::value;
        f32x4 t[2] = {zero4, zero4};
        phase(kT, o0, kT, true, t, xh, xl);            // K half 0
        phase(kT, o1, kT, more, t, xh + 4, xl + 4);    // K half 1
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 b = *reinterpret_cast<const float4*>(bp + 16 * (4 * c + q));
          const f32x4 bb = fh ? zero4 : f32x4{b.x, b.y, b.z, b.w};
          const f32x4 mine = (q >> 1) == 0 ? (fh ? zero4 : t[q & 1]) : (fh ? t[q & 1] : zero4);
          accT[4 * c + q] = mine + bb;
        }
      };
      in_chunk(IC0{}, IC0{}, IC1{}, true);
      in_chunk(IC1{}, IC2{}, IC0{}, true);
      in_chunk(IC2{}, IC1{}, IC2{}, true);
      in_chunk(std::integral_constant<int, 3>{}, IC0{}, IC1{}, false);
    } else {
      // half hh owns output chunks 2 hh and 2 hh + 1 (its table holds only their sub-tiles): tiles 8 hh + 4 j + 2 fh + q
      f32x4 t0[2] = {zero4, zero4}, t1[2] = {zero4, zero4};
      phase(kT, IC0{}, kT, true, t0, xh, xl);
      phase(kT, IC1{}, kT, true, t0, xh + 4, xl + 4);
      phase(kT, IC2{}, kT, true, t1, xh, xl);
      phase(kT, IC0{}, kT, false, t1, xh + 4, xl + 4);
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        const int hv = ct >> 3, j = (ct >> 2) & 1, fv = (ct >> 1) & 1, q = ct & 1;     // compile-time after unrolling
        const float4 b = *reinterpret_cast<const float4*>(bp + 16 * ct);
        const bool own = (hh == hv) && (fh == fv);
        const f32x4 tv = j ? t1[q] : t0[q];
        const bool first = (hh | fh) == 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float bias_r = r == 0 ? b.x : (r == 1 ? b.y : (r == 2 ? b.z : b.w));
          accT[ct][r] = (own ? tv[r] : 0.f) + (first ? bias_r : 0.f);
        }
      }
    }
    exchange();
    next_subblock();
  }

  // this workgroup's heads / hidden chunks: h0 .. h0 + nheads - 1 of the module's a.nheads (biases are indexed globally)
  const int nheads = a.nheads / NSPLIT, nff = a.nff / NSPLIT;
  const int h0 = hh * nheads, f0 = hh * nff;
  for (int blk = 0; blk < a.nblocks; ++blk) {
    const bool last_blk = blk + 1 == a.nblocks;
    const unsigned bias_l = 128u * (unsigned)fh + 16u * (unsigned)(lane_now() >> 4);     // + parity base + 256 (global head / chunk)
    // ================= x += Attention(x) =================
    {
      make_operands(true);
      start_acc(64 * a.nheads, true);
      MDT_STAMP();                                     // LayerNorm + operand split done
      const unsigned bl = vec_l0 + vpar * VEC_BYTES + bias_l + 256u * (unsigned)h0;
      for (int h = 0; h < nheads; ++h) {
        const bool more = h + 1 < nheads;
        f32x4 oT[2];
        f32x4 qT[2], kTt[2], vT[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) { kTt[q] = zero4; vT[q] = zero4; qT[q] = zero4; }
#ifdef MDT_STAMPS_HEAD
#define MDT_HSTAMP() do { if (blk == 0) MDT_STAMP(); } while (0)
#else
#define MDT_HSTAMP() do {} while (0)
#endif
        phase(kT, IC0{}, kT, true, qT, xh, xl);
        MDT_HSTAMP();
        phase(kT, IC1{}, kT, true, qT, xh + 4, xl + 4);
        MDT_HSTAMP();
        phase(kT, IC2{}, kT, true, kTt, xh, xl);
        MDT_HSTAMP();
        phase(kT, IC0{}, kN, true, kTt, xh + 4, xl + 4);
        MDT_HSTAMP();
        phase(kN, IC1{}, kN, true, vT, xh, xl);
        MDT_HSTAMP();
        phase(kN, IC2{}, kN, false, vT, xh + 4, xl + 4);
        MDT_HSTAMP();
        {
          f32x4 bq[2];
          lds_read_f4_off<0>(bq[0], bl + 256 * h); lds_read_f4_off<64>(bq[1], bl + 256 * h);
          lgkm_wait<0>();
          qT[0] += bq[0];
          qT[1] += bq[1];
        }
        f32x4 sp0 = zero4, sp1 = zero4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sp0 = MDT_MFMA_F32(kTt[0][r], qT[0][r], sp0, 0, 0, 0);
          sp1 = MDT_MFMA_F32(kTt[1][r], qT[1][r], sp1, 0, 0, 0);
        }
        const f32x4 mine = sp0 + sp1;
        f32x4* redl = red + lane_now();
        redl[wave * 64] = mine;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        MDT_BARRIER();                         // B(first output sub-tile) + partial exchange
        const f32x4 other = redl[(wave ^ 1) * 64];
        prefetch2(kO, slot_of(tau), 1);
        const f32x4 s01 = fh ? (other + mine) : (mine + other);
        f32x4 st;
        float mx = -INFINITY;
        if constexpr (!F32) kbits = self_mask(lane_now());
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float sv = ((kbits >> r) & 1u) ? s01[r] * scale2 : -INFINITY;
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
        oT[0] = zero4; oT[1] = zero4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = st[r] * inv;
          oT[0] = MDT_MFMA_F32(vT[0][r], p, oT[0], 0, 0, 0);
          oT[1] = MDT_MFMA_F32(vT[1][r], p, oT[1], 0, 0, 0);
        }
        bf16x8 oh[1], ol[1];
        {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = oT[e >> 2][e & 3];
          split8_tf<F32>(v, oh[0], ol[0]);
        }
        MDT_HSTAMP();                                           // attention core done
        phase(kO, IC1{}, kO, true, accT, oh, ol);               // output rows 0..127
        MDT_HSTAMP();
        phase(kO, IC2{}, kT, more, accT + 8, oh, ol);           // output rows 128..255
        MDT_HSTAMP();
      }
      exchange();
      next_subblock();
    }
    // ================= x += Attention(x, context) =================
    if constexpr (NPW > 0) {
      make_operands(true);
      start_acc(64 * a.nheads, true);
      MDT_STAMP();                                     // LayerNorm + operand split done
      const unsigned bl = vec_l0 + vpar * VEC_BYTES + bias_l + 256u * (unsigned)h0;
      for (int h = 0; h < nheads; ++h) {
        const bool more = h + 1 < nheads;
        f32x4 oT[2];
        f32x4 qT[2];
        qT[0] = zero4; qT[1] = zero4;
        phase(kT, IC0{}, kT, true, qT, xh, xl);
        phase(kT, IC1{}, kT, false, qT, xh + 4, xl + 4);
        MDT_BARRIER();                         // B(K tile)
        const unsigned char* sk = slot_of(tau);
        if constexpr (!F32) cross_consts(lane_now(), kb0, vb0, okbits);      // per head, from the opaque lane id (see lane_now())
        {
          f32x4 bq[2];
          lds_read_f4_off<0>(bq[0], bl + 256 * h); lds_read_f4_off<64>(bq[1], bl + 256 * h);
          lgkm_wait<0>();
          qT[0] += bq[0];
          qT[1] += bq[1];
        }
        f32x4 st[KTM];
        float4 k0[KTM], k1[KTM];
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt) {
          k0[kt] = *reinterpret_cast<const float4*>(sk + kb0 + 4096 * kt);
          k1[kt] = *reinterpret_cast<const float4*>(sk + (kb0 ^ 64u) + 4096 * kt);
        }
        f32x4 sp0[KTM], sp1[KTM];
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt) { sp0[kt] = zero4; sp1[kt] = zero4; }
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt) { sp0[kt] = MDT_MFMA_F32(k0[kt].x, qT[0][0], sp0[kt], 0, 0, 0); sp1[kt] = MDT_MFMA_F32(k1[kt].x, qT[1][0], sp1[kt], 0, 0, 0); }
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt) { sp0[kt] = MDT_MFMA_F32(k0[kt].y, qT[0][1], sp0[kt], 0, 0, 0); sp1[kt] = MDT_MFMA_F32(k1[kt].y, qT[1][1], sp1[kt], 0, 0, 0); }
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt) { sp0[kt] = MDT_MFMA_F32(k0[kt].z, qT[0][2], sp0[kt], 0, 0, 0); sp1[kt] = MDT_MFMA_F32(k1[kt].z, qT[1][2], sp1[kt], 0, 0, 0); }
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt) { sp0[kt] = MDT_MFMA_F32(k0[kt].w, qT[0][3], sp0[kt], 0, 0, 0); sp1[kt] = MDT_MFMA_F32(k1[kt].w, qT[1][3], sp1[kt], 0, 0, 0); }
        f32x4* redl = red + lane_now();
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt) {
          st[kt] = sp0[kt] + sp1[kt];
          redl[(kt * 4 + wave) * 64] = st[kt];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ++tau;
        MDT_BARRIER();                         // B(V tile) + partial exchange
        const unsigned char* sv = slot_of(tau);
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt) {
          const f32x4 other = redl[(kt * 4 + (wave ^ 1)) * 64];
          const f32x4 s01 = fh ? (other + st[kt]) : (st[kt] + other);
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
        oT[0] = zero4; oT[1] = zero4;
        f32x4 v0[KTM], v1[KTM];
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v0[kt][r] = *reinterpret_cast<const float*>(sv + vb0[r] + 4096 * kt);
            v1[kt][r] = *reinterpret_cast<const float*>(sv + (vb0[r] ^ 64u) + 4096 * kt);
          }
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = st[kt][r] * inv;
            oT[0] = MDT_MFMA_F32(v0[kt][r], p, oT[0], 0, 0, 0);
            oT[1] = MDT_MFMA_F32(v1[kt][r], p, oT[1], 0, 0, 0);
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // V reads complete before the slot can be refilled
        ++tau;
        MDT_BARRIER();                         // B(first output sub-tile)
        prefetch2(kO, slot_of(tau), 1);
        bf16x8 oh[1], ol[1];
        {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = oT[e >> 2][e & 3];
          split8_tf<F32>(v, oh[0], ol[0]);
        }
        phase(kO, IC1{}, kO, true, accT, oh, ol);
        phase(kO, IC2{}, kT, more, accT + 8, oh, ol);
      }
      exchange();
      next_subblock();
    }
    // ================= x += FeedForward(x)  (last block: the closing convolution folded in) =================
    {
      const int npost = last_blk ? a.npost : 0;
      make_operands(false);
      start_acc(64 * a.nff, npost == 0);             // folded closing convolution: no residual (Wout x rides as tiles)
      MDT_STAMP();                                     // LayerNorm + operand split done
      const unsigned bl = vec_l0 + vpar * VEC_BYTES + bias_l + 256u * (unsigned)f0;
      for (int h = 0; h < nff; ++h) {
        const bool more = h + 1 < nff;
        f32x4 oT[2];
        oT[0] = zero4; oT[1] = zero4;
        phase(kT, IC0{}, kT, true, oT, xh, xl);               // K half 0
        phase(kT, IC1{}, kT, false, oT, xh + 4, xl + 4);      // K half 1
        MDT_BARRIER();                         // B(first W2 sub-tile)
        prefetch2(kO, slot_of(tau), 1);
        {
          f32x4 b1[2];
          lds_read_f4_off<0>(b1[0], bl + 256 * h); lds_read_f4_off<64>(b1[1], bl + 256 * h);
          lgkm_wait<0>();
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) oT[q][r] = gelu_tf(oT[q][r] + b1[q][r]);
        }
        bf16x8 oh[1], ol[1];
        {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = oT[e >> 2][e & 3];
          split8_tf<F32>(v, oh[0], ol[0]);
        }
        phase(kO, IC1{}, kO, true, accT, oh, ol);
        if (npost > 0 && !more) phase(kO, IC2{}, kO, true, accT + 8, oh, ol);   // the folded convolution's sub-tiles follow
        else phase(kO, IC2{}, kT, more, accT + 8, oh, ol);
      }
      if (npost > 0) {
        // + Wout x: per 64-channel k chunk of the raw-x operands one output tile = two row-half sub-tiles; this wave's
        // k-step is the chunk's half fh (a register select: fh is not a compile-time index)
        bf16x8 oxh[1], oxl[1];
        auto pick = [&](int kc) {
          const i32x4 h0v = __builtin_bit_cast(i32x4, xh[2 * kc]), h1v = __builtin_bit_cast(i32x4, xh[2 * kc + 1]);
          const i32x4 l0v = __builtin_bit_cast(i32x4, xl[2 * kc]), l1v = __builtin_bit_cast(i32x4, xl[2 * kc + 1]);
          i32x4 hv, lv;
#pragma unroll
          for (int k = 0; k < 4; ++k) { hv[k] = fh ? h1v[k] : h0v[k]; lv[k] = fh ? l1v[k] : l0v[k]; }
          oxh[0] = __builtin_bit_cast(bf16x8, hv);
          oxl[0] = __builtin_bit_cast(bf16x8, lv);
        };
        if constexpr (NSPLIT == 1) {
          pick(0); phase(kO, IC0{}, kO, true, accT, oxh, oxl); phase(kO, IC1{}, kO, true, accT + 8, oxh, oxl);
          pick(1); phase(kO, IC2{}, kO, true, accT, oxh, oxl); phase(kO, IC0{}, kO, true, accT + 8, oxh, oxl);
          pick(2); phase(kO, IC1{}, kO, true, accT, oxh, oxl); phase(kO, IC2{}, kO, true, accT + 8, oxh, oxl);
          pick(3); phase(kO, IC0{}, kO, true, accT, oxh, oxl); phase(kO, IC1{}, kT, false, accT + 8, oxh, oxl);
        } else {
          // half hh takes the k chunks 2 hh and 2 hh + 1 of Wout x (register selects: hh and fh are not compile-time)
          auto pick2 = [&](int j) {
            const i32x4 a0 = __builtin_bit_cast(i32x4, xh[2 * j]), a1 = __builtin_bit_cast(i32x4, xh[2 * j + 1]);
            const i32x4 a2 = __builtin_bit_cast(i32x4, xh[4 + 2 * j]), a3 = __builtin_bit_cast(i32x4, xh[5 + 2 * j]);
            const i32x4 b0 = __builtin_bit_cast(i32x4, xl[2 * j]), b1 = __builtin_bit_cast(i32x4, xl[2 * j + 1]);
            const i32x4 b2 = __builtin_bit_cast(i32x4, xl[4 + 2 * j]), b3 = __builtin_bit_cast(i32x4, xl[5 + 2 * j]);
            i32x4 hv, lv;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              hv[k] = hh ? (fh ? a3[k] : a2[k]) : (fh ? a1[k] : a0[k]);
              lv[k] = hh ? (fh ? b3[k] : b2[k]) : (fh ? b1[k] : b0[k]);
            }
            oxh[0] = __builtin_bit_cast(bf16x8, hv);
            oxl[0] = __builtin_bit_cast(bf16x8, lv);
          };
          pick2(0); phase(kO, IC0{}, kO, true, accT, oxh, oxl); phase(kO, IC1{}, kO, true, accT + 8, oxh, oxl);
          pick2(1); phase(kO, IC2{}, kO, true, accT, oxh, oxl); phase(kO, IC0{}, kT, false, accT + 8, oxh, oxl);
        }
      }
      exchange();
      if (!last_blk) next_subblock();
    }
  }

  // ---- the residual stream leaves the kernel once: wave (rt, fh) stores channels 128 fh .. 128 fh + 127 of its rows (NSPLIT
  // == 2: both workgroups hold the same rows; half hh stores channels 128 fh + 64 hh .. + 63) ----
  MDT_STAMP();
#ifdef MDT_XH_LOG   // one record per launch and role (row blocks 0..3): [epoch read at entry, XCC id, polls that passed at the first
                    // try, last flag value seen, my last flag value] behind the flag lines (tools/_dbg5.py allocates the room)
  if (NSPLIT == 2 && wave == 0 && lane == 0 && rb < 4) {
    unsigned* lg = a.xflags + 64 + 64 * nrb + 4096 * (2 * rb + hh);
    const unsigned slot = atomicAdd(lg, 1u);
    if (slot < 500) {
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      unsigned* e = lg + 8 + 8 * slot;
      e[0] = xlog_epoch; e[1] = xcc & 15u; e[2] = xlog_first; e[3] = xlog_got; e[4] = xlog_last; e[5] = blockIdx.x;
    }
  }
#endif
  const int ln_end = lane_now();                     // (row index and lane group re-derived here: see lane_now())
  const int m_end = rb * 32 + rt * 16 + (ln_end & 15);
  if (m_end < a.M) {
    if constexpr (NSPLIT == 1) {
      float* xo = a.out + (int64_t)m_end * C + 4 * (ln_end >> 4) + 128 * fh;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        // (component-wise value selects: a select between the two array ELEMENTS becomes a select of their addresses and
        //  keeps the whole accumulator array in scratch memory)
        const f32x4 lo_ = accT[c], hi_ = accT[8 + c];
        store_nt(xo + 16 * c, make_float4(fh ? hi_[0] : lo_[0], fh ? hi_[1] : lo_[1], fh ? hi_[2] : lo_[2], fh ? hi_[3] : lo_[3]));
      }
    } else {
      float* xo = a.out + (int64_t)m_end * C + 4 * (ln_end >> 4) + 128 * fh + 64 * hh;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 q0 = accT[c], q1 = accT[4 + c], q2 = accT[8 + c], q3 = accT[12 + c];
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fh ? (hh ? q3[r] : q2[r]) : (hh ? q1[r] : q0[r]);
        store_nt(xo + 16 * c, make_float4(v[0], v[1], v[2], v[3]));
      }
    }
  }
}

#if MDT_TF_F32
extern int g_pair_capacity_override;
#else
int g_pair_capacity_override = 0;     // tests (mdt_set_tuning("pair_capacity", v)): pretend the device runs only v workgroups at once
#endif

template <int NPW, int NSPLIT, bool F32>
static hipError_t launch_tf2(const TFArgs& a, hipStream_t s) {
  const size_t smem = (size_t)NS * SLOT + RED_BYTES + 2 * VEC_BYTES + 1024;   // ring, S^T exchange, vectors, prefetch sink
  static DevOnce attr_once;                          // per device (mdt_kernels.h)
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tf256<NPW, NSPLIT, F32>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(160 * 1024));
  }
  const int nrb = (a.M + 31) / 32;
  if constexpr (NSPLIT == 1) {
    hipLaunchKernelGGL((k_tf256<NPW, NSPLIT, F32>), dim3((unsigned)nrb), dim3(512), smem, s, a);
    return hipGetLastError();
  } else {
    // The two workgroups of a pair spin on each other's flag, so BOTH must be resident at the same time: a launch may not hold
    // more workgroups than the device runs at once.  capacity = compute units x workgroups of THIS kernel per compute unit
    // (hipOccupancyMaxActiveBlocksPerMultiprocessor: 1, the LDS ring fills the CU), asked once per device.  A batch with more
    // row blocks runs as several launches over consecutive row-block ranges (stream order = one after the other; every pair
    // lives inside one launch, flags and hand-off blocks are indexed by the global row block) -- the results are those of a
    // single launch, so a pinned 'narrow' kernel choice stays valid, and bitwise shard-invariant, at any batch size.
    // (What this cannot see -- compute units held by another stream or process, CU masks -- ends in the polls' time-out and the
    //  status word that sample() checks before it returns, engine.py.)
    static int cap_of[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (cap_of[dev] == 0) {
      int cus = 0, per_cu = 0;
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return hipErrorInvalidDevice;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&k_tf256<NPW, NSPLIT, F32>), 512, smem) != hipSuccess)
        return hipErrorInvalidValue;
      cap_of[dev] = cus * per_cu > 0 ? cus * per_cu : -1;
    }
    const int S = a.pair_stride;
    const int groups_fit = g_pair_capacity_override > 0 ? g_pair_capacity_override / (2 * S) : cap_of[dev] / (2 * S);
    if (groups_fit < 1) return hipErrorLaunchOutOfResources;          // not even one group of 2 S workgroups is co-resident
    const int ngroups = (nrb + S - 1) / S;
    for (int g0 = 0; g0 < ngroups; g0 += groups_fit) {
      TFArgs b = a;
      b.rb_base = g0 * S;
      const int ng = ngroups - g0 < groups_fit ? ngroups - g0 : groups_fit;
      hipLaunchKernelGGL((k_tf256<NPW, NSPLIT, F32>), dim3(2u * (unsigned)S * (unsigned)ng), dim3(512), smem, s, b);
      const hipError_t e = hipGetLastError();
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  }
}

template <int NSPLIT>
static hipError_t launch_tf256_n(const TFArgs& a, hipStream_t s) {
  constexpr bool kF32 = MDT_TF_F32 != 0;
  const bool cross = a.kv != nullptr;
  if (!cross) return launch_tf2<0, NSPLIT, kF32>(a, s);
  switch (((32 / a.T) * a.Tk + 15) / 16) {
    case 1: return launch_tf2<1, NSPLIT, kF32>(a, s);
    case 2: return launch_tf2<2, NSPLIT, kF32>(a, s);
    case 3: return launch_tf2<3, NSPLIT, kF32>(a, s);
    case 4: return launch_tf2<4, NSPLIT, kF32>(a, s);
    case 5: return launch_tf2<5, NSPLIT, kF32>(a, s);
    case 6: return launch_tf2<6, NSPLIT, kF32>(a, s);
    default: return hipErrorInvalidValue;
  }
}

#if MDT_TF_F32
hipError_t launch_tf256_f32(const TFArgs& a, hipStream_t s) {
#else
// workgroups of a pair-split launch that are resident at the same time on the current device (0 if it cannot be determined)
int tf256_pair_capacity() {
  if (g_pair_capacity_override > 0) return g_pair_capacity_override;
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  const size_t smem = (size_t)NS * SLOT + RED_BYTES + 2 * VEC_BYTES + 1024;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tf256<6, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&k_tf256<6, 2, false>), 512, smem) != hipSuccess) return 0;
  return cus * per_cu;
}

bool tf256_supported(int T, int Tk, int nheads, int nff, bool cross) {
  if (T <= 0 || 16 % T || nheads != 8 || nff != 8) return false;        // vectors: [bq 512 | bo 256] / [b1 512 | b2 256]
  if (cross && (Tk <= 0 || (16 / T) * Tk > 48)) return false;           // three key tiles per wave (k_tblock32.hip)
  return true;
}

hipError_t launch_tf256(const TFArgs& a, hipStream_t s) {
  if (a.wf32) return launch_tf256_f32(a, s);           // exact-fp32 products: the instantiations of k_tf256_f32.hip
#endif
  if (a.M <= 0) return hipSuccess;
  const bool cross = a.kv != nullptr;
  if (!tf256_supported(a.T, a.Tk, a.nheads, a.nff, cross) || a.nblocks <= 0 || a.NT <= 0) return hipErrorInvalidValue;
  if (a.npost != 0 && a.npost != 8) return hipErrorInvalidValue;
  if (a.nsplit == 2) {
    if (!a.xbuf || !a.xflags || a.pair_stride <= 0 || a.pair_stride > 64) return hipErrorInvalidValue;
    return launch_tf256_n<2>(a, s);
  }
  return launch_tf256_n<1>(a, s);
}

}  // namespace mdt
