// A whole Transformer1d (modules.py:469-524) of a C = 128 level in ONE launch (MDT_OP_TF128):
//
//   x = Conv1d_1x1(GroupNorm32(x))                       to_in            (:485-490, :520)
//   per TransformerBlock (:456-461):  x += Attention(x);  [x += Attention(x, context);]  x += FeedForward(x)
//   x = Conv1d_1x1(x)                                    to_out           (:512-516, :524), folded into the last FF
//
// Why one launch: as separate launches (k_tblock_lw.hip, k_rconv.hip) every sub-block pays ~5 us of prologue / epilogue
// latency (rows in, LayerNorm, ring fill ... bias + residual round trip, stores) plus a 1.65 us dependent-launch gap for
// 8..22 us of streamed MFMA work, 13 times per transformer.  A wave of k_tblock_lw owns 16 token rows = whole samples and
// all 128 channels, so NOTHING in a transformer crosses waves except the weight stream: here the residual stream stays in
// the output accumulators (fp32, transposed: lane (i, g) holds x[row i][16 ct + 4 g + r]) from the first sub-block to the
// last, and the four loader waves stream the weights of ALL sub-blocks through the LDS ring without ever draining it.
//
//   * accumulator -> operand without lane movement: the next projection's k-slot (st, g, e) is mapped to feature
//     16 (2 st + (e >> 2)) + 4 g + (e & 3), i.e. to registers accT[2 st][0..3], accT[2 st + 1][0..3] of the same lane; the
//     host permutes the K columns of every projection tile that consumes the residual stream accordingly (compiler.py);
//   * LayerNorm / GroupNorm statistics on that layout: 32 values per lane, the 4 lane groups g by v_permlane swaps, the
//     token lanes of a sample by DPP (GroupNorm: 4 channels x T tokens per group = one float4 per lane and token);
//   * accumulators start from x + bias, so a sub-block has no epilogue at all;
//   * the sub-block sequence is fixed by the module (to_in, then per block self-attention, [cross-attention], feed-forward);
//     the loader waves follow a table of tile descriptors (weight tile index | K / V rows of (layer, head)) with scalar
//     loads; every sub-block's vectors (biases) are staged into LDS behind the ring once.
//
// Ring protocol, fragment layouts, attention core and GELU are those of k_tblock_lw.hip (read that file first).
//
// RES > 0: ResnetBlock1d blocks (modules.py:145-205) of the same level run IN FRONT of the transformer in the same launch --
// a wave owns whole samples, so their GroupNorm statistics, FiLM, SiLU and the +-1 taps of the k = 3 convolutions (the
// operand registers shifted by one lane inside the 16-lane row, k_rconv.hip) are wave-local as well:
//   RES = 1 (down path):  x = Block(x), every block's output also stored as a skip tensor          (6 tiles / convolution pair 12)
//   RES = 2 (up path):    x = Block(cat([x, s * skip]))  -- the 2C-channel input is never built: both sources are normalised
//                         and convolved separately into the same accumulators, the 1x1 residual convolution to_out likewise
//                         (24 tiles per block: 2 to_out on x, the skip rows themselves, 2 to_out on the skip, 6 block1 on x, the
//                         skip rows again, 6 block1 on the skip, 6 block2)
// The first convolution's output lives in a second accumulator set hT; the second convolution accumulates into the residual
// stream.  Vectors per block: RES 1 [g1 | b1 | bias1 | g2 | b2 | bias2] (6 C), RES 2 [g1 (2C) | b1 (2C) | bias1 | bias_r | g2 | b2
// | bias2] (9 C); the FiLM (scale | shift) rows of the blocks come from the shared time-mapping row (a.film, 2 C per block).
#include <cstdlib>
#include <type_traits>

#include "mdt_kernels.h"

// This file is compiled twice: as is (split-bf16 products, launch_tf128) and through k_tf128_f32.hip with MDT_TF_F32 = 1 (the
// exact-fp32 instantiations, launch_tf128_f32) -- two translation units that build in parallel.
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

#define MDT_MFMA_BF16 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define MDT_MFMA_F32 __builtin_amdgcn_mfma_f32_16x16x4f32

__device__ __forceinline__ float gelu_tf(float x) {   // exact-erf GELU, branch-free erf (A&S 7.1.26, |error| < 1.5e-7)
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erfa = 1.0f - poly * __expf(-z * z);
  return 0.5f * x * (1.0f + copysignf(erfa, x));
}

// 8 values of one k-step -> the two 128-bit operand registers of the step.  Split-bf16 products (F32 = false): bf16 hi plane /
// lo plane (v = hi + lo to 2^-17).  Exact fp32 products (F32 = true): the values themselves, slots e = 0..3 in `hi`, 4..7 in
// `lo` (bit casts: the operand arrays keep one type for both instantiations; an fp32 k-step is eight 16x16x4 MFMAs, slot
// (g, e = 4 lo + r) of the bf16 step being contraction index g of MFMA (lo, r))
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
  if constexpr (N >= 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
  else if constexpr (N >= 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

constexpr int C = 128;
constexpr int SLOT = 256 * C;   // bytes per weight tile (bf16 hi plane + lo plane)
constexpr int NS = 4;           // ring slots
constexpr int IPT = C / 16;     // DMA pieces per tile per loader wave
constexpr int NST = C / 32;     // k-steps of a projection
constexpr int NCT = C / 16;     // 16-row tiles of the output projection
constexpr int NU = 8;           // units (4 fragment reads + 6 MFMAs) per tile, all three kinds

}  // namespace

template <bool SHR>      // operand of the neighbouring token row (k_rconv.hip): lane i takes lane i - 1 (SHR) / i + 1, 0 at the ends
__device__ __forceinline__ bf16x8 row_shift_tf(const bf16x8& v, bool keep) {
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const i32x4 s = __builtin_bit_cast(i32x4, v);
  i32x4 r;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int t = __builtin_amdgcn_update_dpp(0, s[k], SHR ? 0x111 : 0x101, 0xf, 0xf, true);
    r[k] = keep ? t : 0;
  }
  return __builtin_bit_cast(bf16x8, r);
}

// NPW: LDS-DMA pieces per loader wave per K / V tile = ceil(context rows of the workgroup / 16); 0 = no cross segment
// RES: ResNet blocks in front of the transformer (0 none, 1 single source + skip stores, 2 two sources)
// F32: weight tiles are fp32 FRAGMENT tiles and every projection / convolution product is an exact fp32 MFMA
//      (v_mfma_f32_16x16x4_f32) -- the reference's arithmetic (modules.py:314-320, :350-364, :386-391, :105-112).  A tile holds the
//      same 64 x 128 (or 128 x 64) weights in the same 32 KB: fragment (row tile rt, k-step st, half lo) = 1 KB at
//      ((rt * steps + st) * 2 + lo) * 1024, lane (i, g) float r = W[16 rt + i][k-slot 32 st + 8 g + 4 lo + r] (packed by
//      compiler.py::_tile_f32), so a conflict-free ds_read_b128 at lane * 16 feeds four MFMAs and the loader waves copy the
//      tile linearly.  Same ring, same barriers, same operand registers; 16 / 3 x the MFMA cycles per tile (MFMA-bound).
template <int NPW, int RES, bool F32>
__global__ __launch_bounds__(512) void k_tf128(TFArgs a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NT = a.NT;
  const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.w);

  if (wave >= 4) {
    // ================= loader waves: the weight / K / V stream of every segment (k_tblock_lw.hip) =================
    const int iw = wave - 4;
    __builtin_amdgcn_s_setprio(MDT_LOADER_PRIO);
    const cu32p tiles = (cu32p)a.tiles;              // descriptors: kind (0 P, 1 O, 2 K, 3 V) | aux << 2, scalar loads
    const int lpP = lane >> 5;
    const int xP = (lane & 15) ^ lpP;
    const int baseP = ((lane >> 4) & 1) * (128 * C) + lpP * (2 * C);
    const int xO = (lane & 7) ^ (lane >> 4);
    const int baseO = (lane >> 3) * 128;
    unsigned voffP[IPT], voffO[IPT];
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
      const int inst = iw + 4 * q;
      const int U = 2 * inst;
      voffP[q] = F32 ? (unsigned)(inst * 1024 + lane * 16) : (unsigned)(U * (2 * C) + ((xP ^ (U & 15)) << 4) + baseP);
      voffO[q] = F32 ? (unsigned)(inst * 1024 + lane * 16)
                     : (unsigned)(((inst * 8) / C) * (128 * C) + ((inst * 8) % C) * 128 + ((xO ^ (4 * (inst & 1))) << 4) + baseO);
    }
    // K / V tiles: row R = (sample, key) of the workgroup's samples, 256 B per row and head, chunks swizzled with R & 15
    const int sample0 = blockIdx.x * (64 / a.T);
    const bool second = a.kv2 && sample0 >= a.nsamples / 2;      // dual batch: shared K / V rows for the second half
    unsigned voffKV[4];
    if constexpr (NPW > 0) {
      const int kv_rows = (64 / a.T) * a.Tk;
      const int bstr = second ? 0 : a.kv_bstride;
#pragma unroll
      for (int q = 0; q < NPW; ++q) {
        const int R = 4 * (iw + 4 * q) + (lane >> 4);
        const int Rc = min(R, kv_rows - 1);
        const int sm = min(Rc / a.Tk, a.nsamples - 1 - sample0), key = Rc % a.Tk;
        voffKV[q] = (unsigned)(((sm * bstr + key) * a.ldkv + 4 * ((lane & 15) ^ (R & 15))) * 4);
      }
    }
    // RES = 2: the skip rows of a block travel through the ring like a weight tile (kind 0, bit 20 of aux set, low bits = block):
    // the workgroup's 64 rows x 512 B = one slot; 16-byte chunks XOR-swizzled with row & 15 inside each half row (conflict-free
    // float4 reads of 16 rows).  A plain global load of these rows by the compute waves in the middle of the stream made hipcc
    // spill ~390 registers per lane around it (incl. in-flight fragment registers).
    const int rowb = blockIdx.x * 64;       // (offsets computed per piece: a third per-lane offset array next to voffP / voffO
                                            // made hipcc index a merged array dynamically = scratch + vmcnt(0) per piece)
    auto pieces_of = [&](unsigned d) -> int { return (NPW > 0 && (d & 2u)) ? NPW : IPT; };
    auto issue_tile = [&](int tau, unsigned d) {
      unsigned char* slot = smem + MDT_SLOT_IDX(tau) * SLOT + iw * 1024;
      const unsigned kind = d & 3u, aux = d >> 2;
      if (NPW > 0 && kind >= 2u) {
        if constexpr (NPW > 0) {
          const int layer = (int)(aux >> 4), head = (int)(aux & 15u);
          const float* lb = second ? a.kv2 + (int64_t)layer * a.kv2_lstride
                                   : a.kv + (int64_t)layer * a.kv_lstride + (int64_t)sample0 * a.kv_bstride * a.ldkv;
          const unsigned char* base = reinterpret_cast<const unsigned char*>(lb + 64 * head + (kind == 3u ? 64 * a.nheads : 0));
#pragma unroll
          for (int q = 0; q < NPW; ++q)
            __builtin_amdgcn_global_load_lds(base + voffKV[q], (__attribute__((address_space(3))) void*)(slot + q * 4096), 16, 0, MDT_KV_CPOL);
        }
      } else if (RES == 2 && kind == 0u && (aux >> 20)) {
        if constexpr (RES == 2) {
          const unsigned char* base = reinterpret_cast<const unsigned char*>(
              a.skip + (int64_t)(aux & 0xffu) * a.skip_stride + (int64_t)blockIdx.x * 64 * C);
#pragma unroll
          for (int q = 0; q < IPT; ++q) {
            const int R = 2 * (iw + 4 * q) + (lane >> 5), pch = lane & 31;
            const int c = (pch & 16) | ((pch & 15) ^ (R & 15));
            const unsigned off = (unsigned)((min(rowb + R, a.M - 1) - rowb) * (C * 4) + c * 16);
            __builtin_amdgcn_global_load_lds(base + off, (__attribute__((address_space(3))) void*)(slot + q * 4096), 16, 0, 0);
          }
        }
      } else {
        const unsigned char* tile = wsrc + (int64_t)aux * SLOT;   // wave-uniform
        const bool ptile = kind == 0u;
#pragma unroll
        for (int q = 0; q < IPT; ++q) {
          const unsigned off = ptile ? voffP[q] : voffO[q];
          __builtin_amdgcn_global_load_lds(tile + off, (__attribute__((address_space(3))) void*)(slot + q * 4096), 16, 0, 0);
        }
      }
    };
    auto wait_vm = [&](int allow) {                  // at most `allow` of this wave's vector-memory operations in flight
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
    // the sub-blocks' vectors (nvec floats, a multiple of 256) -> LDS behind the ring, 1 KB pieces, ahead of tile 0: the
    // first counted wait below (all but tile 1's pieces landed) covers them, B(0) publishes them with tile 0
    for (int q = iw; q * 256 < a.nvec; q += 4)
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const unsigned char*>(a.vec) + q * 1024 + lane * 16,
                                       (__attribute__((address_space(3))) void*)(smem + NS * SLOT + q * 1024), 16, 0, 0);
    if constexpr (RES > 0) {                     // the ResNet blocks' FiLM rows (shared arena) behind the vectors
      for (int q = iw; q * 256 < a.nfilm; q += 4)
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const unsigned char*>(a.film) + q * 1024 + lane * 16,
                                         (__attribute__((address_space(3))) void*)(smem + NS * SLOT + a.nvec * 4 + q * 1024), 16, 0, 0);
    }
    issue_tile(0, d0);
    if (NT > 1) issue_tile(1, d1);
    unsigned dn = d1;                                                    // descriptor of tile k + 1
    for (int k = 0; k < NT; ++k) {
      const unsigned d2 = k + 2 < NT ? tiles[k + 2] : 0u;
      // tile k landed; tile k + 1 may be in flight (the common case -- a weight tile next: 8 pieces -- first: as compiled the
      // switch of wait_vm is a cascade of scalar branches, k_tf256.hip)
      if (k + 1 < NT && pieces_of(dn) == IPT) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (NPW > 0 && k + 1 < NT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW > 0 ? NPW : 1) : "memory");          // a K / V tile next
      else wait_vm(k + 1 < NT ? pieces_of(dn) : 0);
      __builtin_amdgcn_s_barrier();                                      // B(k)
      if (k + 2 < NT) issue_tile(k + 2, d2);
      dn = d2;
    }
    prefetch_next_weights(a.pf_ptr, a.pf_lines, iw * 64 + lane);
    return;
  }

  // ================= compute waves =================
  const int i = lane & 15, g = lane >> 4;
  const int row0 = blockIdx.x * 64 + wave * 16;
  const int m = row0 + i;
  const bool mvalid = m < a.M;
  const int mc = mvalid ? m : a.M - 1;
  float* vec_s = reinterpret_cast<float*>(smem + NS * SLOT);             // per-segment vectors behind the ring

  // the residual stream: accT[ct][r] = x[row i][16 ct + 4 g + r], for the whole launch
  f32x4 accT[NCT];
  {
    const float* xp = a.x + (int64_t)mc * C + 4 * g;
    float4 xr[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) xr[ct] = *reinterpret_cast<const float4*>(xp + 16 * ct);
    // (every sub-block's vectors reach LDS by the loader waves' DMA ahead of tile 0: no register, no issue slot here)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                    // P
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) accT[ct] = f32x4{xr[ct].x, xr[ct].y, xr[ct].z, xr[ct].w};
  }

  // Fragment addressing (k_tblock_lw.hip): lane-dependent swizzled part per k-step (projection tiles) / per k-half
  // (output tiles); the 16-row tile and the hi/lo plane are compile-time immediates of the ds_read_b128.
  int aP[NST], aO[2];
#pragma unroll
  for (int st = 0; st < NST; ++st) {
    const int lc = 4 * st + g;
    aP[st] = F32 ? lane * 16 : i * (4 * C) + ((lc & ~15) | ((lc & 15) ^ i)) * 16;      // F32: the k-step is in the immediate
  }
#pragma unroll
  for (int sp = 0; sp < 2; ++sp) aO[sp] = F32 ? lane * 16 : i * 128 + ((4 * sp + g) ^ ((i >> 1) & 7)) * 16;

  bf16x8 fh[3][2], fl[3][2];
  auto frag_read = [&](auto kind, unsigned base, auto uc, int set, auto jc) __attribute__((always_inline)) {
    constexpr int KIND = decltype(kind)::value, u = decltype(uc)::value, j = decltype(jc)::value;
    constexpr int q = j >> 1, lo = j & 1;
    constexpr int off = F32 ? ((KIND == K_O) ? ((2 * (u % (NCT / 2)) + q) * 4096 + (u / (NCT / 2)) * 2048 + lo * 1024)
                                             : ((2 * (u & 1) + q) * 8192 + (u >> 1) * 2048 + lo * 1024))
                            : ((KIND == K_O) ? ((2 * (u % (NCT / 2)) + q) * 16 * 128 + lo * (C * 128))
                                             : ((2 * (u & 1) + q) * 16 * 4 * C + lo * (2 * C)));
#ifdef MDT_ABL_LDSBC   // ablation (WRONG results, timing only): every lane reads the same 16 bytes -- what the fragment reads cost the LDS
    lds_read16_off<off>(lo ? fl[set][q] : fh[set][q], base & 0x18000u);
#else
    lds_read16_off<off>(lo ? fl[set][q] : fh[set][q], base);
#endif
  };
  using J0 = std::integral_constant<int, 0>;
  using J1 = std::integral_constant<int, 1>;
  using J2 = std::integral_constant<int, 2>;
  using J3 = std::integral_constant<int, 3>;
  auto prefetch2 = [&](auto kind, const unsigned char* slot, int off) __attribute__((always_inline)) {   // units 0 and 1 of a phase, as a burst
    constexpr int KIND = decltype(kind)::value;
    const unsigned base = lds_addr(slot) + (KIND == K_O ? aO[0] : aP[0]);
    frag_read(kind, base, J0{}, off % 3, J0{}); frag_read(kind, base, J0{}, off % 3, J1{});
    frag_read(kind, base, J0{}, off % 3, J2{}); frag_read(kind, base, J0{}, off % 3, J3{});
    frag_read(kind, base, J1{}, (off + 1) % 3, J0{}); frag_read(kind, base, J1{}, (off + 1) % 3, J1{});
    frag_read(kind, base, J1{}, (off + 1) % 3, J2{}); frag_read(kind, base, J1{}, (off + 1) % 3, J3{});
  };

  int tau = 0;                                       // tile being consumed
  auto slot_of = [&](int t) -> const unsigned char* { return smem + MDT_SLOT_IDX(t) * SLOT; };

  // One MFMA phase over the tile `tau` (k_tblock_lw.hip): 8 units; the reads of unit u+2 ride between the MFMAs of unit
  // u; for u+2 >= NU they belong to units 0/1 of the NEXT tile (kind NK), published by the barrier before unit NU-2.
  auto phase = [&](auto kind, auto offc, auto nkind, bool has_next, f32x4* acc, const bf16x8* bh, const bf16x8* bl) __attribute__((always_inline)) {
    constexpr int KIND = decltype(kind)::value, OFF = decltype(offc)::value, NK = decltype(nkind)::value;
    const unsigned lc = lds_addr(slot_of(tau)), ln = lds_addr(slot_of(tau + 1));
    unsigned bc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) bc[k] = lc + (KIND == K_O ? aO[k & 1] : aP[k]);
    const unsigned bn = ln + (NK == K_O ? aO[0] : aP[0]);
    auto unit = [&](auto uc) __attribute__((always_inline)) {
      constexpr int u = decltype(uc)::value;
      if (u == NU - 2 && has_next) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                // B(tau + 1)
        __builtin_amdgcn_sched_barrier(0);
      }
      constexpr int s0 = (OFF + u) % 3, s2 = (OFF + u + 2) % 3;
      constexpr bool in_phase = u + 2 < NU;
      const bool pre = in_phase || has_next;
      const bool later = (u + 1 < NU) || has_next;
      if (later) lgkm_wait<4>(); else lgkm_wait<0>();
      constexpr int ia = (KIND == K_O) ? 2 * (u % (NCT / 2)) : 2 * (u & 1);
      constexpr int ib = (KIND == K_O) ? u / (NCT / 2) : (u >> 1);
      auto rd = [&](auto jc) __attribute__((always_inline)) {
        if (!pre) return;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (in_phase) {
          constexpr int u2 = u + 2;
          frag_read(kind, bc[KIND == K_O ? u2 / (NCT / 2) : (u2 >> 1)], std::integral_constant<int, u2>{}, s2, jc);
        } else {
          frag_read(nkind, bn, std::integral_constant<int, u + 2 - NU>{}, s2, jc);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      auto mm = [&](const bf16x8& w, const bf16x8& x, int q) __attribute__((always_inline)) {
        if constexpr (KIND == K_N) acc[ia + q] = MDT_MFMA_BF16(x, w, acc[ia + q], 0, 0, 0);
        else acc[ia + q] = MDT_MFMA_BF16(w, x, acc[ia + q], 0, 0, 0);
      };
      if constexpr (F32) {
        // exact fp32: fragment (q, half) x operand half, four 16x16x4 MFMAs each (r = contraction sub-step); the two
        // accumulators alternate so that no MFMA waits for the one issued just before it
        auto mm4 = [&](const bf16x8& w0, const bf16x8& w1, const bf16x8& x, auto r0c) __attribute__((always_inline)) {
          constexpr int r0 = decltype(r0c)::value;
          const f32x4 a0 = __builtin_bit_cast(f32x4, w0), a1 = __builtin_bit_cast(f32x4, w1), xb = __builtin_bit_cast(f32x4, x);
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
        mm4(fh[s0][0], fh[s0][1], bh[ib], J0{}); rd(J0{});
        mm4(fh[s0][0], fh[s0][1], bh[ib], J2{}); rd(J1{});
        mm4(fl[s0][0], fl[s0][1], bl[ib], J0{}); rd(J2{});
        mm4(fl[s0][0], fl[s0][1], bl[ib], J2{}); rd(J3{});
      } else {
        mm(fl[s0][0], bh[ib], 0); rd(J0{});
        mm(fl[s0][1], bh[ib], 1); rd(J1{});
        mm(fh[s0][0], bl[ib], 0); rd(J2{});
        mm(fh[s0][1], bl[ib], 1); rd(J3{});
        mm(fh[s0][0], bh[ib], 0);
        mm(fh[s0][1], bh[ib], 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    unit(std::integral_constant<int, 0>{}); unit(std::integral_constant<int, 1>{});
    unit(std::integral_constant<int, 2>{}); unit(std::integral_constant<int, 3>{});
    unit(std::integral_constant<int, 4>{}); unit(std::integral_constant<int, 5>{});
    unit(std::integral_constant<int, 6>{}); unit(std::integral_constant<int, 7>{});
    ++tau;
  };
  using IC0 = std::integral_constant<int, 0>;
  using IC1 = std::integral_constant<int, 1>;
  using IC2 = std::integral_constant<int, 2>;
  const IC0 kT{};   // K_T
  const IC1 kN{};   // K_N
  const IC2 kO{};   // K_O

  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  // GroupNorm of the to_in segment: token lanes of a sample by DPP inside the 16-lane row (k_rconv.hip)
  const float t1 = a.T > 1 ? 1.f : 0.f, t2 = a.T > 2 ? 1.f : 0.f, t4 = a.T > 4 ? 1.f : 0.f, t8 = a.T > 8 ? 1.f : 0.f;
  auto dpp_fma = [](float v, float f, auto ctrl) {
    const int mm_ = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, true);
    return __builtin_fmaf(__builtin_bit_cast(float, mm_), f, v);
  };
  auto token_sum = [&](float (&s)[NCT]) __attribute__((always_inline)) {            // stage-major: every stage is one batch of independent exchanges
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) s[ct] = dpp_fma(s[ct], t1, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) s[ct] = dpp_fma(s[ct], t2, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) s[ct] = dpp_fma(s[ct], t4, std::integral_constant<int, 0x141>{});   // row_half_mirror
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) s[ct] = dpp_fma(s[ct], t8, std::integral_constant<int, 0x140>{});   // row_mirror
  };

  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                      // B(0) (also: every wave's vectors are in LDS)
  prefetch2(kT, slot_of(0), 0);
  const unsigned vec_l = lds_addr(reinterpret_cast<const unsigned char*>(vec_s));

  bf16x8 xh[NST], xl[NST];
  // operands of the next projection from the residual stream: k-slot (st, g, e) <-> accT[2 st + (e >> 2)][e & 3]
  auto make_operands = [&](bool layernorm) __attribute__((always_inline)) {
    float mean = 0.f, rstd = 1.f;
    if (layernorm) {                                 // nn.LayerNorm statistics, two-pass; gain / bias folded into the weights
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
  // accT += vec[off + 16 ct + 4 g + r] (the sub-block's output bias: accumulators start from residual + bias)
  auto add_vec = [&](int off, bool replace) __attribute__((always_inline)) {
    const float* p = vec_s + off + 4 * g;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const float4 b = *reinterpret_cast<const float4*>(p + 16 * ct);
      const f32x4 bb = f32x4{b.x, b.y, b.z, b.w};
      accT[ct] = replace ? bb : accT[ct] + bb;
    }
  };


  int voff = 0;                                      // running offset into the vectors: [ResNet blocks] [to_in bias] then per
                                                     // block [bq | bo] (self), [bq | bo] (cross), [b1 | b2] (feed-forward)
  // ================= ResNet blocks in front of the transformer =================
  if constexpr (RES > 0) {
    const float* film_s = vec_s + a.nvec;
    const bool keep_l = (i & (a.T - 1)) != 0, keep_r = (i & (a.T - 1)) != a.T - 1;   // i % T, T a power of two (16 % T == 0)
    // GroupNorm statistics of groups of 16 (pair = false) or 32 (pair = true) channels over the sample's tokens, on the
    // accumulator layout: a group is one / two 16-channel tiles ct -- 4 registers, the 4 lane groups, the sample's token lanes
    auto gn_stats = [&](const f32x4* src, bool pair, float (&mu)[NCT], float (&rs)[NCT]) __attribute__((always_inline)) {
      const float inv_n = 1.0f / (float)(a.T * (pair ? 32 : 16));
      auto reduce = [&](float (&v)[NCT]) __attribute__((always_inline)) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) v[ct] = xg16_add(v[ct]);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) v[ct] = xg32_add(v[ct]);
        token_sum(v);
        if (pair) {
#pragma unroll
          for (int k = 0; k < NCT; k += 2) {
            const float t = v[k] + v[k + 1];
            v[k] = t;
            v[k + 1] = t;
          }
        }
      };
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) mu[ct] = (src[ct][0] + src[ct][1]) + (src[ct][2] + src[ct][3]);
      reduce(mu);
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        mu[ct] *= inv_n;
        float ss = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = src[ct][r] - mu[ct];
          ss += d * d;
        }
        rs[ct] = ss;
      }
      reduce(rs);
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) rs[ct] = __builtin_amdgcn_rsqf(rs[ct] * inv_n + a.eps_res);
    };
    // operands of a convolution: silu(GroupNorm(src) [* (scale + 1) + shift]) in k-slot order (make_operands' mapping)
    auto gn_operands = [&](const f32x4* src, const float (&mu)[NCT], const float (&rs)[NCT], const float* gam, const float* bet,
                           const float* film) __attribute__((always_inline)) {
      // one k-step (two 16-channel tiles) at a time: a whole normalised copy of src next to src, two operand sets, the
      // fragment sets and two more accumulator sets does not fit the register file
#pragma unroll
      for (int st = 0; st < NST; ++st) {
        float v[8];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int ct = 2 * st + hf;
          const float4 gv = *reinterpret_cast<const float4*>(gam + 16 * ct + 4 * g);
          const float4 bv = *reinterpret_cast<const float4*>(bet + 16 * ct + 4 * g);
          const float g4[4] = {gv.x, gv.y, gv.z, gv.w}, b4[4] = {bv.x, bv.y, bv.z, bv.w};
          float u[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float sc = rs[ct] * g4[r];
            u[r] = src[ct][r] * sc + (b4[r] - sc * mu[ct]);
          }
          if (film) {
            const float4 fv = *reinterpret_cast<const float4*>(film + 16 * ct + 4 * g);
            const float4 hv = *reinterpret_cast<const float4*>(film + C + 16 * ct + 4 * g);
            const float f4[4] = {fv.x, fv.y, fv.z, fv.w}, h4[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) u[r] = u[r] * f4[r] + (u[r] + h4[r]);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) v[4 * hf + r] = mvalid ? u[r] * __builtin_amdgcn_rcpf(1.0f + __expf(-u[r])) : 0.f;
        }
        split8_tf<F32>(v, xh[st], xl[st]);
      }
    };
    auto raw_operands = [&](const f32x4* src) __attribute__((always_inline)) {
#pragma unroll
      for (int st = 0; st < NST; ++st) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = mvalid ? src[2 * st + (e >> 2)][e & 3] : 0.f;
        split8_tf<F32>(v, xh[st], xl[st]);
      }
    };
    auto set_vec = [&](f32x4* acc, const float* p, bool add) __attribute__((always_inline)) {
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        const float4 b = *reinterpret_cast<const float4*>(p + 16 * ct + 4 * g);
        const f32x4 bb = f32x4{b.x, b.y, b.z, b.w};
        acc[ct] = add ? acc[ct] + bb : bb;
      }
    };
    // Conv1d(k = 3) on the operands xh / xl into acc: tiles in order (tap, output half); the taps +-1 are the operands
    // shifted by one lane.  O0 = fragment-set rotation at entry; six phases leave it unchanged.
    auto conv3 = [&](auto o0, f32x4* acc, bool has_next) __attribute__((always_inline)) {
      constexpr int O0 = decltype(o0)::value;
      using A0 = std::integral_constant<int, O0 % 3>;
      using A1 = std::integral_constant<int, (O0 + 2) % 3>;
      using A2 = std::integral_constant<int, (O0 + 1) % 3>;
      bf16x8 sh[NST], sl[NST];
#pragma unroll
      for (int st = 0; st < NST; ++st) { sh[st] = row_shift_tf<true>(xh[st], keep_l); sl[st] = row_shift_tf<true>(xl[st], keep_l); }
      phase(kT, A0{}, kT, true, acc, sh, sl);
      phase(kT, A1{}, kT, true, acc + 4, sh, sl);
      phase(kT, A2{}, kT, true, acc, xh, xl);
      phase(kT, A0{}, kT, true, acc + 4, xh, xl);
#pragma unroll
      for (int st = 0; st < NST; ++st) { sh[st] = row_shift_tf<false>(xh[st], keep_r); sl[st] = row_shift_tf<false>(xl[st], keep_r); }
      phase(kT, A1{}, kT, true, acc, sh, sl);
      phase(kT, A2{}, kT, has_next, acc + 4, sh, sl);
    };
    const bool tf_follows = a.has_in || a.nblocks > 0;
    float mu[NCT], rs[NCT];
    for (int rb = 0; rb < a.n_res; ++rb) {
      const bool more = rb + 1 < a.n_res || tf_follows;
      const float* pv = vec_s + voff;
      f32x4 hT[NCT];
      if constexpr (RES == 1) {
        gn_stats(accT, a.res_pair1 != 0, mu, rs);
        gn_operands(accT, mu, rs, pv, pv + C, nullptr);
        set_vec(hT, pv + 2 * C, false);
        conv3(IC0{}, hT, true);
        gn_stats(hT, a.res_pair2 != 0, mu, rs);
        gn_operands(hT, mu, rs, pv + 3 * C, pv + 4 * C, film_s + rb * 2 * C);
        set_vec(accT, pv + 5 * C, true);                 // the stream is the block's residual
        conv3(IC0{}, accT, more);
        if (mvalid) {                                    // every block's output is a skip of the up path
          float* so = a.skip + (int64_t)rb * a.skip_stride + (int64_t)m * C + 4 * g;
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct)
            store_nt(so + 16 * ct, make_float4(accT[ct][0], accT[ct][1], accT[ct][2], accT[ct][3]));
        }
        voff += 6 * C;
      } else {
        // Order chosen for the register budget (never more than three accumulator-sized sets next to two operand sets and the
        // fragment sets): the skip rows arrive as a ring tile TWICE, once for the residual convolution, once for block1.
        // Tiles: to_out_a (2), skip rows, to_out_b (2), conv1_a (6), skip rows, conv1_b (6), conv2 (6).
        auto skip_rows = [&](f32x4* xbT) __attribute__((always_inline)) {
          __builtin_amdgcn_s_barrier();                  // B(skip rows)
          const unsigned char* ss_ = slot_of(tau) + (wave * 16 + i) * (C * 4);
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct) {
            const int ch = 4 * ct + g;
            const float4 xr = *reinterpret_cast<const float4*>(ss_ + (((ch & 16) | ((ch & 15) ^ i)) << 4));
            xbT[ct] = f32x4{xr.x, xr.y, xr.z, xr.w} * a.skip_scale;
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads are complete before the slot can be refilled
          ++tau;
          __builtin_amdgcn_s_barrier();                  // B(next weight tile)
        };
        f32x4 rT[NCT];
        raw_operands(accT);                              // residual = to_out(cat) (k = 1)
        set_vec(rT, pv + 5 * C, false);
        phase(kT, IC0{}, kT, true, rT, xh, xl);
        phase(kT, IC2{}, kT, false, rT + 4, xh, xl);
        {
          f32x4 xbT[NCT];
          skip_rows(xbT);
          prefetch2(kT, slot_of(tau), 0);
          raw_operands(xbT);
        }
        phase(kT, IC0{}, kT, true, rT, xh, xl);
        phase(kT, IC2{}, kT, true, rT + 4, xh, xl);      // rotation 1
        // block1 over both sources (GroupNorm groups of the 2C-channel input never straddle the halves)
        gn_stats(accT, a.res_pair1 != 0, mu, rs);
        gn_operands(accT, mu, rs, pv, pv + 2 * C, nullptr);
        set_vec(hT, pv + 4 * C, false);
        conv3(IC1{}, hT, false);
        {
          f32x4 xbT[NCT];
          skip_rows(xbT);
          gn_stats(xbT, a.res_pair1 != 0, mu, rs);
          gn_operands(xbT, mu, rs, pv + C, pv + 3 * C, nullptr);
          prefetch2(kT, slot_of(tau), 0);
        }
        conv3(IC0{}, hT, true);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) accT[ct] = rT[ct];
        gn_stats(hT, a.res_pair2 != 0, mu, rs);
        gn_operands(hT, mu, rs, pv + 6 * C, pv + 7 * C, film_s + rb * 2 * C);
        set_vec(accT, pv + 8 * C, true);
        conv3(IC0{}, accT, more);
        voff += 9 * C;
      }
    }
  }
  // loop-invariant softmax pieces (k_tblock_lw.hip); computed behind the ResNet blocks, whose register demand would
  // otherwise push them to scratch and back
  const int samp_q = i / a.T;
  float kmask[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) kmask[r] = ((4 * g + r) / a.T == samp_q) ? 0.f : -INFINITY;
  const float scale2 = a.scale * 1.44269504088896340736f;
  int aK = 0, xK = 0, aV[4] = {0, 0, 0, 0}, xV[4] = {0, 0, 0, 0};
  bool kok[4] = {false, false, false, false};
  if constexpr (NPW > 0) {
    const int nkeys = (16 / a.T) * a.Tk;             // this wave's context rows
    const int Rw = wave * nkeys;
    const int Rk = Rw + min(i, nkeys - 1);
    aK = Rk * 256;
    xK = Rk & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int jj = 4 * g + r;
      const int Rv = Rw + min(jj, nkeys - 1);
      aV[r] = Rv * 256 + (i & 3) * 4;
      xV[r] = Rv & 15;
      kok[r] = jj < nkeys && (jj / a.Tk) == samp_q;
    }
  }
  // ---- Transformer1d.to_in: GroupNorm(32 groups of 4 channels, over the sample's tokens) + Conv1d(k = 1) ----
  if (a.has_in) {
    // the lane's float4 accT[ct] is exactly one group at one token; gain / bias are folded into the weights
    float gm[NCT], gv[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) gm[ct] = (accT[ct][0] + accT[ct][1]) + (accT[ct][2] + accT[ct][3]);
    token_sum(gm);
    const float inv_n = 1.0f / (float)(4 * a.T);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      gm[ct] *= inv_n;
      float ss = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = accT[ct][r] - gm[ct];
        ss += d * d;
      }
      gv[ct] = ss;
    }
    token_sum(gv);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const float rs = __builtin_amdgcn_rsqf(gv[ct] * inv_n + a.eps_gn);
#pragma unroll
      for (int r = 0; r < 4; ++r) accT[ct][r] = (accT[ct][r] - gm[ct]) * rs;
    }
    make_operands(false);
    add_vec(voff, true);                             // accT = bias (the convolution REPLACES the stream)
    phase(kT, IC0{}, kT, true, accT, xh, xl);        // output channels 0..63
    phase(kT, IC2{}, kT, false, accT + 4, xh, xl);   // output channels 64..127
    __builtin_amdgcn_s_barrier();                    // B(next tile); realigns the fragment-set rotation for the blocks
    prefetch2(kT, slot_of(tau), 0);
    voff += C;
  }

  const int nheads = a.nheads, nff = a.nff;
  for (int blk = 0; blk < a.nblocks; ++blk) {
    const bool last_blk = blk + 1 == a.nblocks;
    // ================= x += Attention(x) =================
    {
      make_operands(true);
      add_vec(voff + 64 * nheads, false);            // accumulators start from residual + output bias
      const unsigned bias_l = vec_l + 4u * (unsigned)voff + 16u * (unsigned)g;   // bq: + 256 h
      for (int h = 0; h < nheads; ++h) {
        f32x4 oT[4];
        f32x4 qT[4], kTt[4], vT[4];
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) { qT[ft] = zero4; kTt[ft] = zero4; vT[ft] = zero4; }
        phase(kT, IC0{}, kT, true, qT, xh, xl);        // q^T
        phase(kT, IC2{}, kN, true, kTt, xh, xl);       // k^T
        phase(kN, IC1{}, kN, false, vT, xh, xl);       // v (un-transposed)
        __builtin_amdgcn_s_barrier();                  // B(output tile)
        prefetch2(kO, slot_of(tau), 1);
        {
          f32x4 bq[4];     // the host folds the k bias away (softmax-invariant) and the v bias into the output bias
          lds_read_f4_off<0>(bq[0], bias_l + 256 * h); lds_read_f4_off<64>(bq[1], bias_l + 256 * h);
          lds_read_f4_off<128>(bq[2], bias_l + 256 * h); lds_read_f4_off<192>(bq[3], bias_l + 256 * h);
          lgkm_wait<0>();
#pragma unroll
          for (int ft = 0; ft < 4; ++ft) qT[ft] += bq[ft];
        }
#ifdef MDT_ABL_ATTN   // timing experiment only (wrong results): the attention core of self-attention (S, softmax, P V) skipped
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) oT[dt] = qT[dt] + kTt[dt] + vT[dt];
#else
        f32x4 s0 = zero4, s1 = zero4;
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
          s0 = MDT_MFMA_F32(kTt[ft][0], qT[ft][0], s0, 0, 0, 0);
          s1 = MDT_MFMA_F32(kTt[ft][1], qT[ft][1], s1, 0, 0, 0);
          s0 = MDT_MFMA_F32(kTt[ft][2], qT[ft][2], s0, 0, 0, 0);
          s1 = MDT_MFMA_F32(kTt[ft][3], qT[ft][3], s1, 0, 0, 0);
        }
        f32x4 st;
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {                  // key token 4 g + r (within the wave's 16 rows)
          const float sv = (s0[r] + s1[r]) * scale2 + kmask[r];
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
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) oT[dt] = zero4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = st[r] * inv;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) oT[dt] = MDT_MFMA_F32(vT[dt][r], p, oT[dt], 0, 0, 0);
        }
#endif
        bf16x8 oh[2], ol[2];
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = oT[2 * sp + (e >> 2)][e & 3];
          split8_tf<F32>(v, oh[sp], ol[sp]);
        }
        phase(kO, IC1{}, kT, true, accT, oh, ol);      // a tile always follows (cross / feed-forward of this block)
      }
      voff += 64 * nheads + C;
    }
    // ================= x += Attention(x, context): K / V rows hoisted out of the sampling loop =================
    if constexpr (NPW > 0) {
      make_operands(true);
      add_vec(voff + 64 * nheads, false);
      const unsigned bias_l = vec_l + 4u * (unsigned)voff + 16u * (unsigned)g;
      for (int h = 0; h < nheads; ++h) {
        f32x4 oT[4];
        f32x4 qT[4];
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) qT[ft] = zero4;
        phase(kT, IC0{}, kT, false, qT, xh, xl);       // q^T
        __builtin_amdgcn_s_barrier();                  // B(K tile)
        const unsigned char* sk = slot_of(tau);
        float4 kk[4];                                  // A operand of S^T: K[key i][64 h + 16 ft + 4 g + s]
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
          kk[ft] = *reinterpret_cast<const float4*>(sk + aK + (((4 * ft + g) ^ xK) << 4));
        {
          f32x4 bq[4];
          lds_read_f4_off<0>(bq[0], bias_l + 256 * h); lds_read_f4_off<64>(bq[1], bias_l + 256 * h);
          lds_read_f4_off<128>(bq[2], bias_l + 256 * h); lds_read_f4_off<192>(bq[3], bias_l + 256 * h);
          lgkm_wait<0>();
#pragma unroll
          for (int ft = 0; ft < 4; ++ft) qT[ft] += bq[ft];
        }
        f32x4 s0 = zero4, s1 = zero4;
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
          s0 = MDT_MFMA_F32(kk[ft].x, qT[ft][0], s0, 0, 0, 0);
          s1 = MDT_MFMA_F32(kk[ft].y, qT[ft][1], s1, 0, 0, 0);
          s0 = MDT_MFMA_F32(kk[ft].z, qT[ft][2], s0, 0, 0, 0);
          s1 = MDT_MFMA_F32(kk[ft].w, qT[ft][3], s1, 0, 0, 0);
        }
        ++tau;
        __builtin_amdgcn_s_barrier();                  // B(V tile)
        const unsigned char* sv = slot_of(tau);
        f32x4 vT[4];                                   // A operand of O^T: V[key 4 g + r][64 h + 16 dt + i]
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            vT[dt][r] = *reinterpret_cast<const float*>(sv + aV[r] + (((4 * dt + (i >> 2)) ^ xV[r]) << 4));
        f32x4 st;
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float sv2 = kok[r] ? (s0[r] + s1[r]) * scale2 : -INFINITY;
          st[r] = sv2;
          mx = fmaxf(mx, sv2);
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
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) oT[dt] = zero4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = st[r] * inv;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) oT[dt] = MDT_MFMA_F32(vT[dt][r], p, oT[dt], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the V reads are complete before the slot can be refilled
        ++tau;
        __builtin_amdgcn_s_barrier();                  // B(output tile)
        prefetch2(kO, slot_of(tau), 1);
        bf16x8 oh[2], ol[2];
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = oT[2 * sp + (e >> 2)][e & 3];
          split8_tf<F32>(v, oh[sp], ol[sp]);
        }
        phase(kO, IC1{}, kT, true, accT, oh, ol);      // the feed-forward block's tiles follow
      }
      voff += 64 * nheads + C;
    }
    // ================= x += FeedForward(x)  (last block: the closing convolution folded in) =================
    {
      const int npost = last_blk ? a.npost : 0;
      make_operands(false);
      add_vec(voff + 64 * nff, npost > 0);             // folded closing convolution: no residual (Wout x rides as tiles)
      const unsigned bias_l = vec_l + 4u * (unsigned)voff + 16u * (unsigned)g;   // b1: + 256 h
      for (int h = 0; h < nff; ++h) {
        const bool more = h + 1 < nff;
        f32x4 oT[4];
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) oT[ft] = zero4;
        phase(kT, IC0{}, kT, false, oT, xh, xl);       // hidden chunk^T = W1 x^T
        __builtin_amdgcn_s_barrier();                  // B(w2 tile)
        prefetch2(kO, slot_of(tau), 1);
        {
          f32x4 b1[4];
          lds_read_f4_off<0>(b1[0], bias_l + 256 * h); lds_read_f4_off<64>(b1[1], bias_l + 256 * h);
          lds_read_f4_off<128>(b1[2], bias_l + 256 * h); lds_read_f4_off<192>(b1[3], bias_l + 256 * h);
          lgkm_wait<0>();
#pragma unroll
          for (int ft = 0; ft < 4; ++ft)
#pragma unroll
#ifdef MDT_ABL_GELU   // timing experiment only (wrong results): what the serial GELU costs
            for (int r = 0; r < 4; ++r) oT[ft][r] = oT[ft][r] + b1[ft][r];
#else
            for (int r = 0; r < 4; ++r) oT[ft][r] = gelu_tf(oT[ft][r] + b1[ft][r]);
#endif
        }
        bf16x8 oh[2], ol[2];
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = oT[2 * sp + (e >> 2)][e & 3];
          split8_tf<F32>(v, oh[sp], ol[sp]);
        }
        if (npost > 0 && !more) phase(kO, IC1{}, kO, true, accT, oh, ol);   // the folded convolution's tiles follow
        else phase(kO, IC1{}, kT, more || !last_blk, accT, oh, ol);
      }
      if (npost > 0) {                                 // + Wout x: two more output tiles on the raw-x operands
        phase(kO, IC0{}, kO, true, accT, xh, xl);
        phase(kO, IC2{}, kT, false, accT, xh + 2, xl + 2);
      }
      voff += 64 * nff + C;
    }
  }

  // ---- the residual stream leaves the kernel once ----
  if (mvalid) {
    float* xo = a.out + (int64_t)m * C + 4 * g;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
      store_nt(xo + 16 * ct, make_float4(accT[ct][0], accT[ct][1], accT[ct][2], accT[ct][3]));
  }
}

template <int NPW, int RES, bool F32>
static hipError_t launch_tf(const TFArgs& a, hipStream_t s) {
  const size_t smem = (size_t)NS * SLOT + (size_t)(a.nvec + (RES > 0 ? a.nfilm : 0)) * sizeof(float);   // ring + vectors [+ FiLM rows]
  static DevOnce attr_once;                          // per device (mdt_kernels.h)
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tf128<NPW, RES, F32>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(160 * 1024));
  }
  hipLaunchKernelGGL((k_tf128<NPW, RES, F32>), dim3((unsigned)((a.M + 63) / 64)), dim3(512), smem, s, a);
  return hipGetLastError();
}

template <int RES, bool F32>
static hipError_t launch_tf_res(const TFArgs& a, hipStream_t s, bool cross) {
  if (!cross) return launch_tf<0, RES, F32>(a, s);
  switch (((64 / a.T) * a.Tk + 15) / 16) {
    case 1: return launch_tf<1, RES, F32>(a, s);
    case 2: return launch_tf<2, RES, F32>(a, s);
    case 3: return launch_tf<3, RES, F32>(a, s);
    case 4: return launch_tf<4, RES, F32>(a, s);
    default: return hipErrorInvalidValue;
  }
}

#if MDT_TF_F32
hipError_t launch_tf128_f32(const TFArgs& a, hipStream_t s) {
#else
bool tf128_supported(int T, int Tk, int nvec, bool cross) {
  if (T <= 0 || 16 % T || nvec <= 0 || nvec % 256 || nvec > 8192) return false;     // 32 KB of vectors behind the 128 KB ring
  if (cross && (Tk <= 0 || (16 / T) * Tk > 16)) return false;                         // one key tile per wave (k_tblock_lw.hip)
  return true;
}

hipError_t launch_tf128(const TFArgs& a, hipStream_t s) {
  if (a.wf32) return launch_tf128_f32(a, s);           // exact-fp32 products: the instantiations of k_tf128_f32.hip
#endif
  if (a.M <= 0) return hipSuccess;
  const bool cross = a.kv != nullptr;
  if (!tf128_supported(a.T, a.Tk, a.nvec, cross) || a.nblocks < 0 || a.NT <= 0 || a.nheads <= 0 || a.nff <= 0)
    return hipErrorInvalidValue;
  if (a.npost != 0 && a.npost != 2) return hipErrorInvalidValue;
  if (a.n_res < 0 || a.res_kind < 0 || a.res_kind > 2 || (a.res_kind == 0) != (a.n_res == 0)) return hipErrorInvalidValue;
  if (a.n_res == 0 && a.nblocks == 0) return hipErrorInvalidValue;
  if (a.n_res > 0 && (!a.skip || !a.film || a.nfilm % 256 || a.nfilm < 2 * C * a.n_res || a.nvec + a.nfilm > 8192))
    return hipErrorInvalidValue;
  if (a.nblocks == 0 && a.has_in) return hipErrorInvalidValue;
  constexpr bool kF32 = MDT_TF_F32 != 0;
  switch (a.res_kind) {
    case 0: return launch_tf_res<0, kF32>(a, s, cross);
    case 1: return launch_tf_res<1, kF32>(a, s, cross);
    default: return launch_tf_res<2, kF32>(a, s, cross);
  }
}

}  // namespace mdt
