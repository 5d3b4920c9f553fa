// Row-stationary projection (MDT_OP_GEMM with MDT_G_WFMT = 16): out[M][N] = (LayerNorm(x) | x) W^T + bias (+ res),
// K = 128 | 256 input channels, N a multiple of 64 -- the q' / folded self-attention / to_q / to_kv layers that sit BETWEEN the fused
// kernels of configs[2] (QMDiffusionForward: its 64-key cross-attention runs layer by layer).
//
// Why not k_gemm_as (the A-stationary tiled form these layers used through round 4): M is 4096 rows on the one-token level and
// 16384 on the four-token level at B = 4096, i.e. 16 - 64 rows per CU; the launch is a serial latency chain (rows -> LayerNorm ->
// LDS -> first weight chunk, fetched through registers, one chunk in flight, a __syncthreads per chunk) and ran at 37 - 97 TFLOP/s:
// 26 us for 6.4 GFLOP of split-bf16 products (M = 4096, N = 1024, K = 256), 45 us for M = 16384, N = 1024, K = 128.
// Here the structure of k_rconv.hip is reused with a RUN-TIME loop over 64-feature output chunks:
//   * a compute wave keeps its 16 rows, normalised, as bf16 hi / lo MFMA operands in registers for the whole launch (lane (i, g)
//     holds x[i][32 st + 8 g + e]; LayerNorm statistics: in-lane sums + two permlane swaps, two-pass variance);
//   * the weights stream as 32 KB tiles [64 features][128 k] (hi plane | lo plane, order chunk / K half) through the 4-slot LDS
//     ring filled by four loader waves with LDS-DMA, the stream starts while the rows are still being fetched;
//   * per chunk the accumulators are written straight out (lane (i, g) holds 4 consecutive features of row i: one 16-byte store),
//     bias and residual requested one chunk ahead;
//   * RTW = 4 (K = 128): 64-row workgroups, wave = row tile; RTW = 2 (K = 256): 32-row workgroups, wave = (row tile, feature half),
//     both waves of a row tile normalise the full row themselves (no exchange); gridDim.y workgroups share a row block, each
//     taking N / 64 / gridDim.y chunks, while the row blocks alone do not fill the chip.
// Fragment sets rotate mod 4 (units per tile are 4 or 8), so the loop body is the same code for every chunk.
#include <cstdlib>
#include <type_traits>

#include "mdt_kernels.h"

namespace mdt {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define MDT_XGP(NAME, INSN)                                                              \
  __device__ __forceinline__ float NAME(float v) {                                       \
    float a = v, b = v;                                                                  \
    asm("s_nop 1\n\t" INSN " %0, %1" : "+v"(a), "+v"(b));                                \
    return a + b;                                                                        \
  }
MDT_XGP(pj_xg16_add, "v_permlane16_swap_b32")
MDT_XGP(pj_xg32_add, "v_permlane32_swap_b32")
#undef MDT_XGP

constexpr int CS = 128;         // k-width of a weight tile
constexpr int SLOT = 256 * CS;  // bytes per tile (bf16 hi plane + lo plane)
constexpr int NS = 4;           // ring slots
constexpr int IPT = CS / 16;    // DMA pieces per tile per loader wave
constexpr int PJ_BIAS_PASSES = 8;   // bias slice of a workgroup in LDS: up to 8 x 256 floats = 32 chunks

// 8 values of one k-step -> its two 128-bit operand registers: bf16 hi / lo planes, or (F32) the values themselves, slots 0..3 in
// `hi`, 4..7 in `lo` (k_rconv.hip / k_tf128.hip)
template <bool F32>
__device__ __forceinline__ void pj_split8(const float v[8], bf16x8& hi, bf16x8& lo) {
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

template <int OFF>      // fragment read with the (feature tile, plane) part of the address as immediate offset
__device__ __forceinline__ void pj_lds_read16(bf16x8& dst, unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read_b128 offset field");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}

__device__ __forceinline__ unsigned pj_lds_addr(const unsigned char* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)p;
}

template <int N>
__device__ __forceinline__ void pj_lgkm_wait() {
  if constexpr (N >= 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

}  // namespace

template <int RTW, int KH, int LN, bool HASR, bool F32>
__global__ __launch_bounds__(512) void k_proj(ProjArgs a) {
  constexpr int NST = 4 * KH;               // k-steps of the input channels
  constexpr int NFT = (RTW == 4) ? 4 : 2;   // feature tiles per chunk per wave
  constexpr int NU = 2 * NFT;               // units (4 fragment reads + 6 MFMAs) per tile per wave
  constexpr int K = 128 * KH;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nch = a.nch;                              // 64-feature chunks of this workgroup
  const int c0 = (int)blockIdx.y * nch;
  const int TT = nch * KH;                            // tiles this workgroup consumes

  if (wave >= 4) {
    // ================= loader waves: the weight stream (k_rconv.hip) =================
    const int iw = wave - 4;
    __builtin_amdgcn_s_setprio(MDT_LOADER_PRIO);
    const int lpP = lane >> 5;
    const int xP = (lane & 15) ^ lpP;
    const int baseP = ((lane >> 4) & 1) * (128 * CS) + lpP * (2 * CS);
    unsigned voffP[IPT];
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
      const int U = 2 * (iw + 4 * q);
      // (F32: fp32 fragment tiles are stored in LDS order, a linear copy)
      voffP[q] = F32 ? (unsigned)((iw + 4 * q) * 1024 + lane * 16) : (unsigned)(U * (2 * CS) + ((xP ^ (U & 15)) << 4) + baseP);
    }
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.w) + (int64_t)c0 * KH * SLOT;
    auto issue_tile = [&](int tau) {
      const unsigned char* tile = wsrc + (int64_t)tau * SLOT;
      unsigned char* slot = smem + (tau & (NS - 1)) * SLOT + iw * 1024;
#pragma unroll
      for (int q = 0; q < IPT; ++q) {
        unsigned off = voffP[q];
        asm volatile("" : "+v"(off));                // keep the 32-bit offset form (k_res256.hip)
        __builtin_amdgcn_global_load_lds(tile + off, (__attribute__((address_space(3))) void*)(slot + q * 4096), 16, 0, 0);
      }
    };
    __builtin_amdgcn_s_barrier();                                        // P: the compute waves' row loads are queued first
    issue_tile(0);
    if (TT > 1) issue_tile(1);
    // the workgroup's bias slice goes to LDS behind the ring: inside the chunk loop it is a counted ds_read like the fragments, the
    // compute waves' loop has no vector-memory wait for it (the wait hipcc places in front of these ds_writes also covers tiles 0 / 1,
    // which B(0) needs anyway)
    {
      float* bias_s = reinterpret_cast<float*>(smem + NS * SLOT);
      for (int e = iw * 64 + lane; e < 64 * nch; e += 256) bias_s[e] = a.bias ? a.bias[64 * c0 + e] : 0.f;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    for (int k = 0; k < TT; ++k) {
      if (k + 1 < TT) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // tile k landed; tile k + 1 may be in flight
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                                      // B(k)
      if (k + 2 < TT) issue_tile(k + 2);
    }
    return;
  }

  // ================= compute waves =================
  const int i = lane & 15, g = lane >> 4;
  const int rt = (RTW == 4) ? wave : (wave >> 1), fh = (RTW == 4) ? 0 : (wave & 1);
  const int row0 = blockIdx.x * (16 * RTW) + rt * 16;
  const int m = row0 + i;
  const bool mvalid = m < a.M;
  const int mc = mvalid ? m : a.M - 1;

  // fragment addressing inside a tile: row = 16 ft + i, 16-byte chunk = 4 st + g, XOR-swizzled with i (k_rconv.hip)
  int aP[4];
#pragma unroll
  for (int st = 0; st < 4; ++st)      // F32: fragment (feature tile ft, k-step st, half lo) at ft * 8192 + st * 2048 + lo * 1024, the lane's 16 bytes inside
    aP[st] = F32 ? lane * 16 + fh * 16384 + st * 2048 : fh * (2 * 16 * 4 * CS) + i * (4 * CS) + (((4 * st + g) ^ i) << 4);
  bf16x8 frh[4][2], frl[4][2];
  auto frag_read = [&](unsigned base, auto uc, int set, auto jc) {
    constexpr int u = decltype(uc)::value, j = decltype(jc)::value;
    constexpr int q = j >> 1, lo = j & 1;
    constexpr int off = F32 ? ((RTW == 4) ? ((2 * (u & 1) + q) * 8192 + lo * 1024) : (q * 8192 + lo * 1024))
                            : ((RTW == 4) ? ((2 * (u & 1) + q) * 16 * 4 * CS + lo * (2 * CS)) : (q * 16 * 4 * CS + lo * (2 * CS)));
    pj_lds_read16<off>(lo ? frl[set][q] : frh[set][q], base);
  };
  using J0 = std::integral_constant<int, 0>;
  using J1 = std::integral_constant<int, 1>;
  using J2 = std::integral_constant<int, 2>;
  using J3 = std::integral_constant<int, 3>;
  auto slot_of = [&](int t) -> const unsigned char* { return smem + (t & (NS - 1)) * SLOT; };

  // ---- the wave's 16 rows: load, LayerNorm, split into bf16 hi / lo MFMA operands ----
  bf16x8 xh[NST], xl[NST];
  {
    const float* xp = a.x + (int64_t)mc * a.lda + 8 * g;
    float4 xu[NST], xw[NST];
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      xu[st] = *reinterpret_cast<const float4*>(xp + 32 * st);
      xw[st] = *reinterpret_cast<const float4*>(xp + 32 * st + 4);
    }
    // LN = 1: LayerNorm WITHOUT affine (the compiler folds gain into the weights and bias into the bias: no per-channel vectors here);
    // LN = 2: with gain / bias vectors -- K = 256 then holds 48 float4 per lane ahead of barrier P, more than the registers take: hipcc
    // splits the requests around a wait and the weight stream starts a memory latency late (P at 7.4 k cycles instead of ~1.5 k)
    float4 ga[LN == 2 ? NST : 1][2], be[LN == 2 ? NST : 1][2];
    if constexpr (LN == 2) {
#pragma unroll
      for (int st = 0; st < NST; ++st)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int c = 32 * st + 8 * g + 4 * hf;
          ga[st][hf] = *reinterpret_cast<const float4*>(a.gamma + c);
          be[st][hf] = *reinterpret_cast<const float4*>(a.beta + c);
        }
    }
    __builtin_amdgcn_sched_barrier(0);               // the loads' first uses (and their waits) stay behind the barrier
    __builtin_amdgcn_s_barrier();                    // P: the weight stream starts behind this wave's requests
    __builtin_amdgcn_sched_barrier(0);
    float xr[NST][8];
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      const float4 u = xu[st], w = xw[st];
      const float sc = mvalid ? 1.0f : 0.f;
      xr[st][0] = u.x * sc; xr[st][1] = u.y * sc; xr[st][2] = u.z * sc; xr[st][3] = u.w * sc;
      xr[st][4] = w.x * sc; xr[st][5] = w.y * sc; xr[st][6] = w.z * sc; xr[st][7] = w.w * sc;
    }
    if constexpr (LN != 0) {
      float s = 0.f;
#pragma unroll
      for (int st = 0; st < NST; ++st)
        s += ((xr[st][0] + xr[st][1]) + (xr[st][2] + xr[st][3])) + ((xr[st][4] + xr[st][5]) + (xr[st][6] + xr[st][7]));
      s = pj_xg32_add(pj_xg16_add(s));
      const float mean = s * (1.0f / (float)K);
      float ss = 0.f;
#pragma unroll
      for (int st = 0; st < NST; ++st)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = xr[st][e] - mean;
          ss += d * d;
        }
      ss = pj_xg32_add(pj_xg16_add(ss));
      const float rstd = __builtin_amdgcn_rsqf(ss * (1.0f / (float)K) + a.eps);     // v_rsq_f32: 1 ulp (as k_rconv.hip)
      if constexpr (LN == 2) {
#pragma unroll
        for (int st = 0; st < NST; ++st)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const float4 gv = ga[st][hf], bv = be[st][hf];
            const float g4[4] = {gv.x, gv.y, gv.z, gv.w}, b4[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) xr[st][4 * hf + e] = (xr[st][4 * hf + e] - mean) * rstd * g4[e] + b4[e];
          }
      } else {
#pragma unroll
        for (int st = 0; st < NST; ++st)
#pragma unroll
          for (int e = 0; e < 8; ++e) xr[st][e] = (xr[st][e] - mean) * rstd;
      }
    }
#pragma unroll
    for (int st = 0; st < NST; ++st) pj_split8<F32>(xr[st], xh[st], xl[st]);
  }

  __builtin_amdgcn_s_barrier();                      // B(0): tile 0 has landed, the bias slice is in LDS (loader waves)
  {
    const unsigned l0 = pj_lds_addr(slot_of(0));
    const unsigned p0 = l0 + aP[0], p1 = l0 + aP[(RTW == 4) ? 0 : 1];
    frag_read(p0, J0{}, 0, J0{}); frag_read(p0, J0{}, 0, J1{}); frag_read(p0, J0{}, 0, J2{}); frag_read(p0, J0{}, 0, J3{});
    frag_read(p1, J1{}, 1, J0{}); frag_read(p1, J1{}, 1, J1{}); frag_read(p1, J1{}, 1, J2{}); frag_read(p1, J1{}, 1, J3{});
  }

  // The residual of a chunk (HASR) is requested ONE CHUNK AHEAD into the other of two register sets, and the stores are ordinary ones:
  // gfx950 counts loads and stores in the same vmcnt, in order -- requested at the start of their own chunk the wait in front of the
  // epilogue also covered the previous chunk's (then non-temporal) stores: 5.9 k cycles per chunk in the first version.  Without a
  // residual the loop has no vector-memory load at all.
  f32x4 acc[NFT];
  float4 irA[HASR ? NFT : 1], irB[HASR ? NFT : 1];
  const float* rsrc = HASR ? a.res + (int64_t)mc * a.ldr + 64 * c0 + 16 * (NFT * fh) + 4 * g : nullptr;
  auto request = [&](int cc, float4 (&ir)[HASR ? NFT : 1]) {
    if constexpr (HASR) {
#pragma unroll
      for (int q = 0; q < NFT; ++q) ir[q] = *reinterpret_cast<const float4*>(rsrc + 64 * cc + 16 * q);
    }
  };
  const unsigned bias_l = pj_lds_addr(smem + NS * SLOT) + (16 * (NFT * fh) + 4 * g) * 4;
  auto chunk = [&](auto lastc, int cc, float4 (&ir)[HASR ? NFT : 1], float4 (&irn)[HASR ? NFT : 1]) {
    constexpr bool LAST = decltype(lastc)::value;               // the workgroup's last chunk: nothing behind its last tile
    const int f0 = 64 * (c0 + cc) + 16 * (NFT * fh) + 4 * g;   // this lane's first feature of feature tile 0
    if constexpr (!LAST) request(cc + 1, irn);
    f32x4 ib[NFT];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NFT; ++q)                               // behind the prefetched fragments of units 0 / 1: landed by unit 1's wait
      asm volatile("ds_read_b128 %0, %1" : "=v"(ib[q]) : "v"(bias_l + (64 * cc + 16 * q) * 4) : "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NFT; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < KH; ++kh) {
      const int tau = cc * KH + kh;
      constexpr bool more_ct = !LAST;                           // (kh < KH - 1 is always followed by a tile)
      const bool more = (kh + 1 < KH) || more_ct;
      const unsigned lc = pj_lds_addr(slot_of(tau)), ln = pj_lds_addr(slot_of(tau + 1));
      unsigned bc[4], bn[2];
#pragma unroll
      for (int k = 0; k < 4; ++k) bc[k] = lc + aP[k];
      bn[0] = ln + aP[0];
      bn[1] = ln + aP[(RTW == 4) ? 0 : 1];
      auto unit = [&](auto uc) {
        constexpr int u = decltype(uc)::value;
        if (u == NU - 2 && more) {
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_barrier();              // B(tau + 1)
          __builtin_amdgcn_sched_barrier(0);
        }
        constexpr int s0 = u % 4, s2 = (u + 2) % 4;  // NU is 4 or 8: every tile starts on set 0
        constexpr bool in_tile = u + 2 < NU;
        const bool pre = in_tile || more;
        const bool later = (u + 1 < NU) || more;
        if (later) pj_lgkm_wait<4>(); else pj_lgkm_wait<0>();
        constexpr int ia = (RTW == 4) ? 2 * (u & 1) : 0, ib_ = (RTW == 4) ? (u >> 1) : u;
        const bf16x8 oph = xh[4 * kh + ib_], opl = xl[4 * kh + ib_];
        auto rd = [&](auto jc) {
          if (!pre) return;
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (in_tile) {
            constexpr int u2 = u + 2;
            frag_read(bc[(RTW == 4) ? (u2 >> 1) : u2], std::integral_constant<int, u2>{}, s2, jc);
          } else {
            frag_read(bn[u + 2 - NU], std::integral_constant<int, u + 2 - NU>{}, s2, jc);
          }
          __builtin_amdgcn_sched_barrier(0);
        };
        auto mm = [&](const bf16x8& w, const bf16x8& x, int q) {
          acc[ia + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, acc[ia + q], 0, 0, 0);
        };
        if constexpr (F32) {
          // exact fp32: fragment (q, half) x operand half, four 16x16x4 MFMAs each, the two accumulators alternating (k_rconv.hip)
          auto mm4 = [&](const bf16x8& w0, const bf16x8& w1, const bf16x8& x, auto r0c) {
            constexpr int r0 = decltype(r0c)::value;
            const f32x4 a0 = __builtin_bit_cast(f32x4, w0), a1 = __builtin_bit_cast(f32x4, w1), xb = __builtin_bit_cast(f32x4, x);
#pragma unroll
            for (int r = r0; r < r0 + 2; ++r) {
              acc[ia] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[r], xb[r], acc[ia], 0, 0, 0);
              acc[ia + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[r], xb[r], acc[ia + 1], 0, 0, 0);
            }
          };
          mm4(frh[s0][0], frh[s0][1], oph, J0{}); rd(J0{});
          mm4(frh[s0][0], frh[s0][1], oph, J2{}); rd(J1{});
          mm4(frl[s0][0], frl[s0][1], opl, J0{}); rd(J2{});
          mm4(frl[s0][0], frl[s0][1], opl, J2{}); rd(J3{});
        } else {
          mm(frl[s0][0], oph, 0); rd(J0{});
          mm(frl[s0][1], oph, 1); rd(J1{});
          mm(frh[s0][0], opl, 0); rd(J2{});
          mm(frh[s0][1], opl, 1); rd(J3{});
          mm(frh[s0][0], oph, 0);
          mm(frh[s0][1], oph, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      unit(std::integral_constant<int, 0>{}); unit(std::integral_constant<int, 1>{});
      unit(std::integral_constant<int, 2>{}); unit(std::integral_constant<int, 3>{});
      if constexpr (NU == 8) {
        unit(std::integral_constant<int, 4>{}); unit(std::integral_constant<int, 5>{});
        unit(std::integral_constant<int, 6>{}); unit(std::integral_constant<int, 7>{});
      }
    }
    // ---- out[m][64 chunk + 16 ft + 4 g + r] = acc + bias (+ res) ----
    if (mvalid) {
#pragma unroll
      for (int q = 0; q < NFT; ++q) {
        f32x4 v = acc[q] + ib[q];
        if constexpr (HASR) v += f32x4{ir[q].x, ir[q].y, ir[q].z, ir[q].w};
        *reinterpret_cast<float4*>(a.out + (int64_t)m * a.ldc + f0 + 16 * q) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  using NotLast = std::false_type;
  using Last = std::true_type;
  // (every copy of the chunk body is ~2.5 KB of code fetched cold once per launch: without a residual there are two, not five)
  if constexpr (!HASR) {
    int cc = 0;
#pragma unroll 1
    for (; cc + 1 < nch; ++cc) chunk(NotLast{}, cc, irA, irB);
    chunk(Last{}, cc, irA, irB);
  } else {
    request(0, irA);
    int cc = 0;
#pragma unroll 1
    for (; cc + 2 < nch; cc += 2) {
      chunk(NotLast{}, cc, irA, irB);
      chunk(NotLast{}, cc + 1, irB, irA);
    }
    if (cc + 1 < nch) {
      chunk(NotLast{}, cc, irA, irB);
      chunk(Last{}, cc + 1, irB, irA);
    } else {
      chunk(Last{}, cc, irA, irB);
    }
  }
}

template <int RTW, int KH, int LN, bool HASR, bool F32>
static hipError_t launch_pj(const ProjArgs& a0, hipStream_t s) {
  ProjArgs a = a0;
  static DevOnce attr_once;                          // per device (mdt_kernels.h)
  constexpr int SMEM = NS * SLOT + PJ_BIAS_PASSES * 256 * 4;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_proj<RTW, KH, LN, HASR, F32>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
  }
  const int rows = 16 * RTW, rb = (a.M + rows - 1) / rows, nchunks = a.N / 64;
  int nsplit = 1;                                    // workgroups per row block while the row blocks alone leave CUs idle
  while ((rb * nsplit < 256 && nchunks % (2 * nsplit) == 0) || nchunks / nsplit > PJ_BIAS_PASSES * 4) {
    if (nchunks % (2 * nsplit)) return hipErrorInvalidValue;       // (N / 64 > 32 chunks per workgroup and no even split)
    nsplit *= 2;
  }
  a.nch = nchunks / nsplit;
  hipLaunchKernelGGL((k_proj<RTW, KH, LN, HASR, F32>), dim3((unsigned)rb, (unsigned)nsplit), dim3(512), (size_t)SMEM, s, a);
  return hipGetLastError();
}

template <int RTW, int KH, bool F32>
static hipError_t launch_pj2(const ProjArgs& a, hipStream_t s) {
  if (a.ln && a.gamma) return a.res ? launch_pj<RTW, KH, 2, true, F32>(a, s) : launch_pj<RTW, KH, 2, false, F32>(a, s);
  if (a.ln) return a.res ? launch_pj<RTW, KH, 1, true, F32>(a, s) : launch_pj<RTW, KH, 1, false, F32>(a, s);
  return a.res ? launch_pj<RTW, KH, 0, true, F32>(a, s) : launch_pj<RTW, KH, 0, false, F32>(a, s);
}

bool proj_supported(int K, int N, int lda, int ldc, int ldr) {
  // (N <= 2048: a workgroup's bias slice in LDS holds 32 chunks, and a workgroup takes all of a row block's chunks when the row blocks
  //  fill the chip)
  return (K == 128 || K == 256) && N > 0 && N <= 64 * 4 * PJ_BIAS_PASSES && N % 64 == 0 && lda % 4 == 0 && ldc % 4 == 0 && ldr % 4 == 0;
}

hipError_t launch_proj(const ProjArgs& a, hipStream_t s) {
  if (a.M <= 0) return hipSuccess;
  if (!proj_supported(a.K, a.N, a.lda, a.ldc, a.ldr) || (a.ln && (!a.gamma != !a.beta))) return hipErrorInvalidValue;
  auto al16 = [](const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; };
  if (!al16(a.x) || !al16(a.out) || !al16(a.w) || (a.bias && !al16(a.bias)) || (a.res && !al16(a.res))) return hipErrorInvalidValue;
  if (a.wf32) return a.K == 128 ? launch_pj2<4, 1, true>(a, s) : launch_pj2<2, 2, true>(a, s);   // fp32 fragment tiles, exact fp32 products
  return a.K == 128 ? launch_pj2<4, 1, false>(a, s) : launch_pj2<2, 2, false>(a, s);
}

}  // namespace mdt
