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
#include <cstdlib>
#include <type_traits>

#include "mdt_kernels.h"

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

constexpr int C = 128;
constexpr int SLOT = 256 * C;   // bytes per weight tile (bf16 hi plane + lo plane)
constexpr int NS = 4;           // ring slots
constexpr int IPT = C / 16;     // DMA pieces per tile per loader wave
constexpr int NST = C / 32;     // k-steps of a projection
constexpr int NCT = C / 16;     // 16-row tiles of the output projection
constexpr int NU = 8;           // units (4 fragment reads + 6 MFMAs) per tile, all three kinds

}  // namespace

// NPW: LDS-DMA pieces per loader wave per K / V tile = ceil(context rows of the workgroup / 16); 0 = no cross segment
template <int NPW>
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
      voffP[q] = (unsigned)(U * (2 * C) + ((xP ^ (U & 15)) << 4) + baseP);
      voffO[q] = (unsigned)(((inst * 8) / C) * (128 * C) + ((inst * 8) % C) * 128 + ((xO ^ (4 * (inst & 1))) << 4) + baseO);
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
    auto pieces_of = [&](unsigned d) -> int { return (NPW > 0 && (d & 2u)) ? NPW : IPT; };
    auto issue_tile = [&](int tau, unsigned d) {
      unsigned char* slot = smem + (tau % NS) * SLOT + iw * 1024;
      const unsigned kind = d & 3u, aux = d >> 2;
      if (NPW > 0 && kind >= 2u) {
        if constexpr (NPW > 0) {
          const int layer = (int)(aux >> 4), head = (int)(aux & 15u);
          const float* lb = second ? a.kv2 + (int64_t)layer * a.kv2_lstride
                                   : a.kv + (int64_t)layer * a.kv_lstride + (int64_t)sample0 * a.kv_bstride * a.ldkv;
          const unsigned char* base = reinterpret_cast<const unsigned char*>(lb + 64 * head + (kind == 3u ? 64 * a.nheads : 0));
#pragma unroll
          for (int q = 0; q < NPW; ++q)
            __builtin_amdgcn_global_load_lds(base + voffKV[q], (__attribute__((address_space(3))) void*)(slot + q * 4096), 16, 0, 0);
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
    aP[st] = i * (4 * C) + ((lc & ~15) | ((lc & 15) ^ i)) * 16;
  }
#pragma unroll
  for (int sp = 0; sp < 2; ++sp) aO[sp] = i * 128 + ((4 * sp + g) ^ ((i >> 1) & 7)) * 16;

  bf16x8 fh[3][2], fl[3][2];
  auto frag_read = [&](auto kind, unsigned base, auto uc, int set, auto jc) {
    constexpr int KIND = decltype(kind)::value, u = decltype(uc)::value, j = decltype(jc)::value;
    constexpr int q = j >> 1, lo = j & 1;
    constexpr int off = (KIND == K_O) ? ((2 * (u % (NCT / 2)) + q) * 16 * 128 + lo * (C * 128))
                                      : ((2 * (u & 1) + q) * 16 * 4 * C + lo * (2 * C));
    lds_read16_off<off>(lo ? fl[set][q] : fh[set][q], base);
  };
  using J0 = std::integral_constant<int, 0>;
  using J1 = std::integral_constant<int, 1>;
  using J2 = std::integral_constant<int, 2>;
  using J3 = std::integral_constant<int, 3>;
  auto prefetch2 = [&](auto kind, const unsigned char* slot, int off) {   // units 0 and 1 of a phase, as a burst
    constexpr int KIND = decltype(kind)::value;
    const unsigned base = lds_addr(slot) + (KIND == K_O ? aO[0] : aP[0]);
    frag_read(kind, base, J0{}, off % 3, J0{}); frag_read(kind, base, J0{}, off % 3, J1{});
    frag_read(kind, base, J0{}, off % 3, J2{}); frag_read(kind, base, J0{}, off % 3, J3{});
    frag_read(kind, base, J1{}, (off + 1) % 3, J0{}); frag_read(kind, base, J1{}, (off + 1) % 3, J1{});
    frag_read(kind, base, J1{}, (off + 1) % 3, J2{}); frag_read(kind, base, J1{}, (off + 1) % 3, J3{});
  };

  int tau = 0;                                       // tile being consumed
  auto slot_of = [&](int t) -> const unsigned char* { return smem + (t % NS) * SLOT; };

  // One MFMA phase over the tile `tau` (k_tblock_lw.hip): 8 units; the reads of unit u+2 ride between the MFMAs of unit
  // u; for u+2 >= NU they belong to units 0/1 of the NEXT tile (kind NK), published by the barrier before unit NU-2.
  auto phase = [&](auto kind, auto offc, auto nkind, bool has_next, f32x4* acc, const bf16x8* bh, const bf16x8* bl) {
    constexpr int KIND = decltype(kind)::value, OFF = decltype(offc)::value, NK = decltype(nkind)::value;
    const unsigned lc = lds_addr(slot_of(tau)), ln = lds_addr(slot_of(tau + 1));
    unsigned bc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) bc[k] = lc + (KIND == K_O ? aO[k & 1] : aP[k]);
    const unsigned bn = ln + (NK == K_O ? aO[0] : aP[0]);
    auto unit = [&](auto uc) {
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
      auto rd = [&](auto jc) {
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
      auto mm = [&](const bf16x8& w, const bf16x8& x, int q) {
        if constexpr (KIND == K_N) acc[ia + q] = MDT_MFMA_BF16(x, w, acc[ia + q], 0, 0, 0);
        else acc[ia + q] = MDT_MFMA_BF16(w, x, acc[ia + q], 0, 0, 0);
      };
      mm(fl[s0][0], bh[ib], 0); rd(J0{});
      mm(fl[s0][1], bh[ib], 1); rd(J1{});
      mm(fh[s0][0], bl[ib], 0); rd(J2{});
      mm(fh[s0][1], bl[ib], 1); rd(J3{});
      mm(fh[s0][0], bh[ib], 0);
      mm(fh[s0][1], bh[ib], 1);
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

  // loop-invariant softmax pieces (k_tblock_lw.hip)
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
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  // GroupNorm of the to_in segment: token lanes of a sample by DPP inside the 16-lane row (k_rconv.hip)
  const float t1 = a.T > 1 ? 1.f : 0.f, t2 = a.T > 2 ? 1.f : 0.f, t4 = a.T > 4 ? 1.f : 0.f, t8 = a.T > 8 ? 1.f : 0.f;
  auto dpp_fma = [](float v, float f, auto ctrl) {
    const int mm_ = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, true);
    return __builtin_fmaf(__builtin_bit_cast(float, mm_), f, v);
  };
  auto token_sum = [&](float (&s)[NCT]) {            // stage-major: every stage is one batch of independent exchanges
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
  auto make_operands = [&](bool layernorm) {
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
      split8_tf(v, xh[st], xl[st]);
    }
  };
  // accT += vec[off + 16 ct + 4 g + r] (the sub-block's output bias: accumulators start from residual + bias)
  auto add_vec = [&](int off, bool replace) {
    const float* p = vec_s + off + 4 * g;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const float4 b = *reinterpret_cast<const float4*>(p + 16 * ct);
      const f32x4 bb = f32x4{b.x, b.y, b.z, b.w};
      accT[ct] = replace ? bb : accT[ct] + bb;
    }
  };

  int voff = 0;                                      // running offset into the vectors: [to_in bias] then per block
                                                     // [bq | bo] (self), [bq | bo] (cross), [b1 | b2] (feed-forward)
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
    add_vec(0, true);                                // accT = bias (the convolution REPLACES the stream)
    phase(kT, IC0{}, kT, true, accT, xh, xl);        // output channels 0..63
    phase(kT, IC2{}, kT, false, accT + 4, xh, xl);   // output channels 64..127
    __builtin_amdgcn_s_barrier();                    // B(next tile); realigns the fragment-set rotation for the blocks
    prefetch2(kT, slot_of(tau), 0);
    voff = C;
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
        bf16x8 oh[2], ol[2];
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = oT[2 * sp + (e >> 2)][e & 3];
          split8_tf(v, oh[sp], ol[sp]);
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
          split8_tf(v, oh[sp], ol[sp]);
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
            for (int r = 0; r < 4; ++r) oT[ft][r] = gelu_tf(oT[ft][r] + b1[ft][r]);
        }
        bf16x8 oh[2], ol[2];
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = oT[2 * sp + (e >> 2)][e & 3];
          split8_tf(v, oh[sp], ol[sp]);
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

template <int NPW>
static hipError_t launch_tf(const TFArgs& a, hipStream_t s) {
  const size_t smem = (size_t)NS * SLOT + (size_t)a.nvec * sizeof(float);   // ring + every segment's vectors
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tf128<NPW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((k_tf128<NPW>), dim3((unsigned)((a.M + 63) / 64)), dim3(512), smem, s, a);
  return hipGetLastError();
}

bool tf128_supported(int T, int Tk, int nvec, bool cross) {
  if (T <= 0 || 16 % T || nvec <= 0 || nvec % 256 || nvec > 7168) return false;     // 28 KB of vectors behind the 128 KB ring
  if (cross && (Tk <= 0 || (16 / T) * Tk > 16)) return false;                         // one key tile per wave (k_tblock_lw.hip)
  return true;
}

hipError_t launch_tf128(const TFArgs& a, hipStream_t s) {
  if (a.M <= 0) return hipSuccess;
  const bool cross = a.kv != nullptr;
  if (!tf128_supported(a.T, a.Tk, a.nvec, cross) || a.nblocks <= 0 || a.NT <= 0 || a.nheads <= 0 || a.nff <= 0)
    return hipErrorInvalidValue;
  if (a.npost != 0 && a.npost != 2) return hipErrorInvalidValue;
  if (!cross) return launch_tf<0>(a, s);
  switch (((64 / a.T) * a.Tk + 15) / 16) {
    case 1: return launch_tf<1>(a, s);
    case 2: return launch_tf<2>(a, s);
    case 3: return launch_tf<3>(a, s);
    case 4: return launch_tf<4>(a, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace mdt
