// Row-stationary convolution for the C = 128 / C = 256 levels of the U-Net (MDT_OP_RCONV):
//
//   xn  = in_scale * x                                              (skip scaling of UpsampleBlock1d.add_skip)
//   xn  = silu( GroupNorm(xn) * (scale + 1) + shift )               (ConvBlock1d.forward, modules.py:117-121; optional)
//   out = bias + sum_tap W[:, tap, :] xn[t + tap - taps/2]  (+ res)  (Conv1d k = 1 | 3, zero padding inside the sample)
//
// with C output channels and one or two C-channel input sources per launch: the 2C-channel input cat([x, s * skip])
// of the up path is never materialised, the launch runs prologue + taps for source a, then for source b, into the
// same accumulators (GroupNorm groups never straddle the two halves; b's gain/bias/weight tiles follow a's).
//
// Why not the tiled GEMM (k_gemm_bf16x3.hip): at B = 1024 these layers are 4096..16384 rows x 128..256 columns,
// one 64x64 tile per CU, and the tiled kernel spends ~1800 cycles per 32-deep k-step on load -> LDS -> MFMA
// latency (17.6 us for K = 768, 9x its MFMA time), after a separate GroupNorm launch.  Here the structure of the
// fused transformer blocks (k_tblock_lw.hip / k_tblock32.hip) is reused:
//   * a wave keeps its 16 token rows, normalised, as bf16 hi/lo MFMA operands in registers for the whole launch;
//     the GroupNorm statistics are wave-local (a wave owns whole samples and all C channels);
//   * the +-1 taps are the same registers shifted by one lane inside the 16-lane row (DPP row_shr / row_shl, zero
//     fill at the row ends, one select at sample boundaries inside the row);
//   * weights arrive as 32 KB tiles [64 features][128 k] (C = 128 tile format) through the 4-slot LDS ring filled
//     by four loader waves; fragment reads are interleaved with the MFMAs and pipelined across tiles.
// Tile order: tap, K half, 64-feature chunk (the shifted operands of a (tap, K half) serve all chunks).
//   RTW = 4: 64-row workgroups, wave = row tile, 4 feature tiles per chunk per wave   (16384-row level)
//   RTW = 2: 32-row workgroups, wave = (row tile, feature half)                       ( 4096-row level)
#include <cstdlib>
#include <type_traits>

#include "mdt_kernels.h"

// Compiled twice: as is (split-bf16 products, launch_rconv) and through k_rconv_f32.hip with MDT_TF_F32 = 1 (exact fp32 MFMA
// products on fp32 fragment tiles, launch_rconv_f32; tile format and MFMA order as k_tf128.hip).
#ifndef MDT_TF_F32
#define MDT_TF_F32 0
#endif

namespace mdt {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// streaming store: the output is consumed by the next launch (through the memory side: the per-XCD L2s are written
// back / invalidated at every kernel boundary anyway), so it need not stay dirty in this XCD's L2 until kernel end
__device__ __forceinline__ void store_nt(float* p, float4 v) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(f4{v.x, v.y, v.z, v.w}, reinterpret_cast<f4*>(p));
}

// lane-group exchanges over +-16 / +-32 lanes with the gfx950 permlane swaps (VALU, no LDS round trip). The swap is in
// place on two registers: fed the same value twice, v_permlane16_swap leaves (rows 0,0,2,2) and (rows 1,1,3,3),
// v_permlane32_swap (halves lo,lo) and (hi,hi); combining the two gives every lane the pair it would get from xor 16 /
// xor 32. Written as asm: through __builtin_amdgcn_permlane*_swap hipcc 7.2 folds the two results into one register.
// The s_nop covers the VALU-write -> permlane-swap-read hazard for the copies the compiler places just before.
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
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define MDT_MFMA_BF16 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define MDT_MFMA_F32 __builtin_amdgcn_mfma_f32_16x16x4f32
constexpr bool F32 = MDT_TF_F32 != 0;   // product type of this translation unit

constexpr int CS = 128;         // k-width of a weight tile
constexpr int SLOT = 256 * CS;  // bytes per tile (bf16 hi plane + lo plane)
constexpr int NS = 4;           // ring slots
constexpr int IPT = CS / 16;    // DMA pieces per tile per loader wave

// 8 values of one k-step -> its two 128-bit operand registers: bf16 hi / lo planes, or (F32) the values themselves, slots
// 0..3 in `hi`, 4..7 in `lo` (k_tf128.hip)
__device__ __forceinline__ void split8_rc(const float v[8], bf16x8& hi, bf16x8& lo) {
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

__device__ __forceinline__ void lds_read16_rc(bf16x8& dst, const unsigned char* p) {
  const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)p;
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory");
}

template <int OFF>      // fragment read with the (tile, plane) part of the address as immediate offset (k_tblock_lw.hip)
__device__ __forceinline__ void lds_read16_off_rc(bf16x8& dst, unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read_b128 offset field");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}

__device__ __forceinline__ unsigned lds_addr_rc(const unsigned char* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)p;
}

template <int N>
__device__ __forceinline__ void lgkm_wait_rc() {
  if constexpr (N >= 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// operand of the neighbouring token row: lane i takes lane i - 1 (SHR) or i + 1 inside its 16-lane row, 0 at the ends
template <bool SHR>
__device__ __forceinline__ bf16x8 row_shift(const bf16x8& v, bool keep) {
  const i32x4 s = __builtin_bit_cast(i32x4, v);
  i32x4 r;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int t = __builtin_amdgcn_update_dpp(0, s[k], SHR ? 0x111 : 0x101, 0xf, 0xf, true);
    r[k] = keep ? t : 0;
  }
  return __builtin_bit_cast(bf16x8, r);
}

}  // namespace

// NSPLIT workgroups (blockIdx.y) share a row block, each producing C / NSPLIT of the output channels: no reduction,
// half the weight stream per workgroup, twice the workgroups (the 4096-row level has only 128 row blocks).
// PRO: 2 = GroupNorm + FiLM + SiLU prologue (all run-time optional), 1 = without FiLM, 0 = no prologue at all -- the
// leaner instantiations exist for the two-source form, which the parameter registers push over the register budget
template <int RTW, int C, int TAPS, int NSPLIT, int NSRC, int PRO, bool F32_>
__global__ __launch_bounds__(512) void k_rconv(RConvArgs a) {
  static_assert(F32_ == F32, "one product type per translation unit");
  constexpr int NST = C / 32;               // k-steps of the input channels
  constexpr int NKH = C / CS;               // K halves (tiles per tap per chunk)
  constexpr int NCHT = C / 64;              // 64-feature output chunks in total
  constexpr int NCH = NCHT / NSPLIT;        // ... of this workgroup
  constexpr int NFT = (RTW == 4) ? 4 : 2;   // feature tiles per chunk per wave
  constexpr int NU = 2 * NFT;               // units (4 fragment reads + 6 MFMAs) per tile per wave
  constexpr int NT = TAPS * NKH * NCH;      // tiles of this workgroup
  constexpr int NSTW = (RTW == 2) ? NST / 2 : NST;   // k-steps normalised by this wave (RTW = 2: the wave pair splits them)
  constexpr int NTF = TAPS * NKH * NCHT;    // tiles per source in the stream (all workgroups)
  constexpr int nsrc = NSRC;                // input sources (2: a concatenated input, never materialised)
  constexpr int TT = nsrc * NT;             // tiles this workgroup consumes
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.w);

  if (wave >= 4) {
    // ================= loader waves: the weight stream (k_tblock_lw.hip) =================
    const int iw = wave - 4;
    __builtin_amdgcn_s_setprio(MDT_LOADER_PRIO);
    const int lpP = lane >> 5;
    const int xP = (lane & 15) ^ lpP;
    const int baseP = ((lane >> 4) & 1) * (128 * CS) + lpP * (2 * CS);
    unsigned voffP[IPT];
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
      const int U = 2 * (iw + 4 * q);
      voffP[q] = F32 ? (unsigned)((iw + 4 * q) * 1024 + lane * 16) : (unsigned)(U * (2 * CS) + ((xP ^ (U & 15)) << 4) + baseP);
    }
    auto issue_tile = [&](int tau) {
      // stream order (tap, K half, chunk): this workgroup's chunks are NCH consecutive ones of every (tap, K half)
      const int tl = tau % NT;
      // (a.nb > 1: blockIdx.y = output block x NSPLIT + split; the blocks' tile streams follow each other)
      const int ts = ((int)blockIdx.y / NSPLIT) * (nsrc * NTF) + (tau / NT) * NTF + (tl / NCH) * NCHT + ((int)blockIdx.y % NSPLIT) * NCH + tl % NCH;
      const unsigned char* tile = wsrc + (int64_t)ts * SLOT;
      unsigned char* slot = smem + (tau % NS) * SLOT + iw * 1024;
#pragma unroll
      for (int q = 0; q < IPT; ++q)
        __builtin_amdgcn_global_load_lds(tile + voffP[q], (__attribute__((address_space(3))) void*)(slot + q * 4096), 16, 0, 0);
    };
    __builtin_amdgcn_s_barrier();                                        // P: the compute waves' row loads are queued first
    issue_tile(0);
    if (TT > 1) issue_tile(1);
    if (RTW == 2) __builtin_amdgcn_s_barrier();                          // X: operand exchange of the compute waves
    for (int k = 0; k < TT; ++k) {
      if (k + 1 < TT) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // tile k landed; tile k+1 may be in flight
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                                      // B(k)
      if (k + 2 < TT) issue_tile(k + 2);
      if (RTW == 2 && nsrc >= 2 && k > 0 && k % NT == 0) __builtin_amdgcn_s_barrier();   // X of the next source (follows its first B)
    }
    prefetch_next_weights(a.pf_ptr, a.pf_lines, iw * 64 + lane);
    return;
  }

  // ================= compute waves =================
  // first output channel of this workgroup: output block (MDT_R_NB: blockIdx.y / NSPLIT, C channels each), then its split's chunks
  const int ocol0 = ((int)blockIdx.y / NSPLIT) * C + ((int)blockIdx.y % NSPLIT) * (64 * NCH);
#ifdef MDT_STAMPS   // tuning build: wave 0 of workgroup 0 records the shader clock into the film buffer's tail (never in a real run)
  unsigned long long* stamps = reinterpret_cast<unsigned long long*>(const_cast<float*>(a.dbgbuf));
  int nstamp = 0;
#define MDT_STAMP()                                                                   \
  do {                                                                                \
    if (stamps && blockIdx.x == 0 && wave == 0 && nstamp < 60) {                      \
      unsigned long long t_;                                                          \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");      \
      if (lane == 0) stamps[nstamp] = t_;                                             \
      ++nstamp;                                                                       \
    }                                                                                 \
  } while (0)
#else
#define MDT_STAMP() do {} while (0)
#endif
  MDT_STAMP();
  const int i = lane & 15, g = lane >> 4;
  const int rt = (RTW == 4) ? wave : (wave >> 1), fh = (RTW == 4) ? 0 : (wave & 1);
  const int row0 = blockIdx.x * (16 * RTW) + rt * 16;
  const int m = row0 + i;
  const bool mvalid = m < a.M;
  const int mc = mvalid ? m : a.M - 1;

  // neighbours inside the sample: row i - 1 exists unless i starts a sample, row i + 1 unless i ends one
  const int it = i & (a.T - 1);                      // i % T: 16 % T == 0 (host check), T is a power of two
  const bool has_prev = it != 0, has_next_row = it != a.T - 1;

  // fragment addressing inside a tile: row = 16 ft + i, chunk = 4 st + g (C = 128 tile format, k_tblock_lw.hip)
  int aP[4];
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    const int lc = 4 * st + g;
    // F32: fragment (feature tile ft, k-step st, half lo) at ft * 8192 + st * 2048 + lo * 1024, the lane's 16 bytes inside
    aP[st] = F32 ? lane * 16 + fh * 16384 + st * 2048 : fh * (2 * 16 * 4 * CS) + i * (4 * CS) + ((lc & ~15) | ((lc & 15) ^ i)) * 16;
  }
  bf16x8 frh[3][2], frl[3][2];
  // read j (= 2 q + plane) of unit u; `base` = LDS address of the slot + the lane's swizzled part for the unit's
  // k-step (RTW = 4: aP[u >> 1], RTW = 2: aP[u]); the (feature tile, plane) part is the instruction's immediate offset
  auto frag_read = [&](unsigned base, auto uc, int set, auto jc) {
    constexpr int u = decltype(uc)::value, j = decltype(jc)::value;
    constexpr int q = j >> 1, lo = j & 1;
    constexpr int off = F32 ? ((RTW == 4) ? ((2 * (u & 1) + q) * 8192 + lo * 1024) : (q * 8192 + lo * 1024))
                            : ((RTW == 4) ? ((2 * (u & 1) + q) * 16 * 4 * CS + lo * (2 * CS)) : (q * 16 * 4 * CS + lo * (2 * CS)));
    lds_read16_off_rc<off>(lo ? frl[set][q] : frh[set][q], base);
  };
  using J0 = std::integral_constant<int, 0>;
  using J1 = std::integral_constant<int, 1>;
  using J2 = std::integral_constant<int, 2>;
  using J3 = std::integral_constant<int, 3>;
  auto slot_of = [&](int t) -> const unsigned char* { return smem + (t % NS) * SLOT; };

  f32x4 acc[NCH][NFT];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int q = 0; q < NFT; ++q) acc[c][q] = f32x4{0.f, 0.f, 0.f, 0.f};

  // two sources without a GroupNorm prologue: the second source's rows are requested together with the first's (in
  // front of barrier P) and wait in registers while the first source's tiles run
  constexpr bool GN = PRO > 0;
  // NSRC = 4 | 8 (round 5, MDT_R_KSRC): the sources are consecutive C-channel BLOCKS of one tensor [M][NSRC C] -- a K = NSRC C projection
  // onto C outputs (configs[2]'s output projections behind the 8 x 128 attention rows), no prologue; NSRC = 2 stays the concatenation
  constexpr bool KSRC = NSRC > 2;
  static_assert(!KSRC || (PRO == 0 && TAPS == 1), "K blocks: a plain projection");
  // NSRC = 2 serves both: x | x2 (the concatenations of the up path), or -- a.ksrc == 2 -- the two C-channel blocks of one tensor
  // (a strided convolution in patch form: 2 C = factor x channels, see compiler.py::down_patch)
  const bool ks = KSRC || (NSRC == 2 && a.ksrc == 2);
  constexpr bool EARLY2 = NSRC >= 2 && !GN;          // the NEXT source's rows are requested while this one's tiles run
  float4 xu2[EARLY2 ? NSTW : 1], xw2[EARLY2 ? NSTW : 1];
#pragma unroll 1
  for (int src = 0; src < nsrc; ++src) {
  const float* xsrc = ks ? a.x + src * C : (src ? a.x2 : a.x);
  const int lda = (src && !ks) ? a.lda2 : a.lda;
  const float in_scale = (src && !ks) ? a.in_scale2 : a.in_scale;
  const float* gamma = a.gamma + src * C;            // source b's gain / bias follow source a's
  const float* beta = a.beta + src * C;
  // ---- the wave's 16 rows: load, (GroupNorm + FiLM + SiLU), split into bf16 hi/lo MFMA operands ----
  // lane (i, g) holds x[i][32 st + 8 g + e].  RTW = 2: the two waves of a row tile normalise half of the k-steps each
  // (GroupNorm groups are at most 64 channels wide, a half is 128) and exchange the operands through 32 KB of LDS
  // behind the ring.
  bf16x8 xh[NST], xl[NST];
  {
    const int st0 = (RTW == 2) ? fh * NSTW : 0;
    float xr[NSTW][8];
    const float* xp = xsrc + (int64_t)mc * lda + 32 * st0 + 8 * g;
    float4 xu[NSTW], xw[NSTW];
    if (EARLY2 && src >= 1) {
#pragma unroll
      for (int st = 0; st < NSTW; ++st) { xu[st] = xu2[EARLY2 ? st : 0]; xw[st] = xw2[EARLY2 ? st : 0]; }
    } else {
#pragma unroll
      for (int st = 0; st < NSTW; ++st) {
        xu[st] = *reinterpret_cast<const float4*>(xp + 32 * st);
        xw[st] = *reinterpret_cast<const float4*>(xp + 32 * st + 4);
      }
    }
    if constexpr (EARLY2) {
      if (src + 1 < nsrc) {
        const float* xq = (ks ? a.x + (int64_t)mc * a.lda + (src + 1) * C : a.x2 + (int64_t)mc * a.lda2) + 32 * st0 + 8 * g;
#pragma unroll
        for (int st = 0; st < NSTW; ++st) {
          xu2[st] = *reinterpret_cast<const float4*>(xq + 32 * st);
          xw2[st] = *reinterpret_cast<const float4*>(xq + 32 * st + 4);
        }
      }
    }
    // per-channel parameters of the lane's channels, requested together with the rows: behind the statistics (where
    // they are used) their 32 loads sat in blocks of their own, each a round trip queued behind the loaders' stream
    float4 ga[NSTW][2], be[NSTW][2], fs[NSTW][2], fsh[NSTW][2];
    if (GN && a.gsize > 0) {
#pragma unroll
      for (int st = 0; st < NSTW; ++st)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int c = 32 * (st0 + st) + 8 * g + 4 * hf;
          ga[st][hf] = *reinterpret_cast<const float4*>(gamma + c);
          be[st][hf] = *reinterpret_cast<const float4*>(beta + c);
        }
      if (PRO == 2 && a.film) {
#pragma unroll
        for (int st = 0; st < NSTW; ++st)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const int c = 32 * (st0 + st) + 8 * g + 4 * hf;
            fs[st][hf] = *reinterpret_cast<const float4*>(a.film + c);
            fsh[st][hf] = *reinterpret_cast<const float4*>(a.film + a.film_ld + c);
          }
      }
    }
    // the accumulators start from bias (+ residual), requested here with everything else: loaded in the epilogue they
    // were one more exposed round trip per launch
    float4 ib[NCH][NFT], ir[NCH][NFT];
    if (src == 0) {
      if (a.bias) {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
          for (int q = 0; q < NFT; ++q)
            ib[c][q] = *reinterpret_cast<const float4*>(a.bias + ocol0 + 64 * c + 16 * (NFT * fh + q) + 4 * g);
      }
      if (a.res) {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
          for (int q = 0; q < NFT; ++q)
            ir[c][q] = *reinterpret_cast<const float4*>(a.res + (int64_t)mc * a.ldr + ocol0 + 64 * c + 16 * (NFT * fh + q) + 4 * g);
      }
    }
    // P: the loader waves start the weight stream only now, behind this wave's requests (a row load queued behind the
    // stream's first tiles came back ~2000 cycles later)
    __builtin_amdgcn_sched_barrier(0);               // the loads' first uses (and their waits) stay behind the barrier
    if (src == 0) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#ifdef MDT_STAMPS
    MDT_STAMP();                                     // loads issued, past barrier P
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MDT_STAMP();                                     // rows and per-channel parameters arrived
#endif
    if (src == 0) {
      if (a.bias) {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
          for (int q = 0; q < NFT; ++q) acc[c][q] = f32x4{ib[c][q].x, ib[c][q].y, ib[c][q].z, ib[c][q].w};
      }
      if (a.res) {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
          for (int q = 0; q < NFT; ++q) acc[c][q] += f32x4{ir[c][q].x, ir[c][q].y, ir[c][q].z, ir[c][q].w};
      }
    }
#pragma unroll
    for (int st = 0; st < NSTW; ++st) {
      const float4 u = xu[st], w = xw[st];
      const float sc = mvalid ? in_scale : 0.f;
      xr[st][0] = u.x * sc; xr[st][1] = u.y * sc; xr[st][2] = u.z * sc; xr[st][3] = u.w * sc;
      xr[st][4] = w.x * sc; xr[st][5] = w.y * sc; xr[st][6] = w.z * sc; xr[st][7] = w.w * sc;
    }
    if (GN && a.gsize > 0) {
      // group statistics over (tokens of the sample) x (gsize channels): the lane's 4-value halves, the lane groups g,
      // the paired k-steps, then the sample's token lanes.  Stage-major: every stage is one batch of independent
      // exchanges with a compile-time distance (a dependent chain per value, value after value, cost 30k cycles).
      // Which stages apply depends on run-time gsize / T: every stage is computed and then selected with a uniform
      // bit mask (or, for the DPP stages, added through an fma with a 0 / 1 factor) -- as `if (gs >= 16)` blocks the
      // ~20 tiny basic blocks cost more in branches, phi copies and hazard nops (700 instructions) than the sums.
      const int gs = a.gsize;
      const unsigned m16 = gs >= 16 ? ~0u : 0u, m32 = gs >= 32 ? ~0u : 0u, m64 = gs >= 64 ? ~0u : 0u;
      const float t1 = a.T > 1 ? 1.f : 0.f, t2 = a.T > 2 ? 1.f : 0.f, t4 = a.T > 4 ? 1.f : 0.f, t8 = a.T > 8 ? 1.f : 0.f;
      auto sel = [](unsigned m, float x, float y) {
        return __builtin_bit_cast(float, (m & __builtin_bit_cast(unsigned, x)) | (~m & __builtin_bit_cast(unsigned, y)));
      };
      // token lanes: DPP moves inside the 16-lane row (no LDS round trip).  Once the quads are uniform, mirroring the
      // half row / the row adds exactly the lanes that xor 4 / xor 8 would.
      auto dpp_fma = [](float v, float f, auto ctrl) {
        const int m = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, true);
        return __builtin_fmaf(__builtin_bit_cast(float, m), f, v);
      };
      auto reduce = [&](auto nhc, float (&s)[NSTW][2]) {
        constexpr int NH = decltype(nhc)::value;           // 1: the lane's two halves belong to one group (gsize >= 8)
        auto all = [&](auto fn) {
#pragma unroll
          for (int st = 0; st < NSTW; ++st)
#pragma unroll
            for (int hf = 0; hf < NH; ++hf) s[st][hf] = fn(s[st][hf]);
        };
        if (NH == 1) {
#pragma unroll
          for (int st = 0; st < NSTW; ++st) s[st][0] += s[st][1];
        }
        all([&](float v) { return sel(m16, xg16_add(v), v); });
        all([&](float v) { return sel(m32, xg32_add(v), v); });
#pragma unroll
        for (int st = 0; st + 1 < NSTW; st += 2)
#pragma unroll
          for (int hf = 0; hf < NH; ++hf) {
            const float t = s[st][hf] + s[st + 1][hf];
            s[st][hf] = sel(m64, t, s[st][hf]);
            s[st + 1][hf] = sel(m64, t, s[st + 1][hf]);
          }
        all([&](float v) { return dpp_fma(v, t1, std::integral_constant<int, 0xB1>{}); });    // quad_perm [1,0,3,2]
        all([&](float v) { return dpp_fma(v, t2, std::integral_constant<int, 0x4E>{}); });    // quad_perm [2,3,0,1]
        all([&](float v) { return dpp_fma(v, t4, std::integral_constant<int, 0x141>{}); });   // row_half_mirror
        all([&](float v) { return dpp_fma(v, t8, std::integral_constant<int, 0x140>{}); });   // row_mirror
        if (NH == 1) {
#pragma unroll
          for (int st = 0; st < NSTW; ++st) s[st][1] = s[st][0];
        }
      };
      auto group_sum = [&](float (&s)[NSTW][2]) {
        if (gs >= 8) reduce(std::integral_constant<int, 1>{}, s);
        else reduce(std::integral_constant<int, 2>{}, s);
      };
      const float inv_n = 1.0f / (float)(a.T * gs);
      float mean[NSTW][2], rstd[NSTW][2];
#pragma unroll
      for (int st = 0; st < NSTW; ++st) {
        mean[st][0] = (xr[st][0] + xr[st][1]) + (xr[st][2] + xr[st][3]);
        mean[st][1] = (xr[st][4] + xr[st][5]) + (xr[st][6] + xr[st][7]);
      }
      group_sum(mean);
#pragma unroll
      for (int st = 0; st < NSTW; ++st)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          mean[st][hf] *= inv_n;
          float ss = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float d = xr[st][4 * hf + e] - mean[st][hf];
            ss += d * d;
          }
          rstd[st][hf] = ss;
        }
      group_sum(rstd);
      MDT_STAMP();                                   // group statistics
#pragma unroll
      for (int st = 0; st < NSTW; ++st)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) rstd[st][hf] = __builtin_amdgcn_rsqf(rstd[st][hf] * inv_n + a.eps);   // v_rsq_f32: 1 ulp
#pragma unroll
      for (int st = 0; st < NSTW; ++st)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const float4 gv = ga[st][hf], bv = be[st][hf];
          const float g4[4] = {gv.x, gv.y, gv.z, gv.w}, b4[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float sc = rstd[st][hf] * g4[e];
            xr[st][4 * hf + e] = xr[st][4 * hf + e] * sc + (b4[e] - sc * mean[st][hf]);
          }
        }
      // FiLM and SiLU as passes of their own under one uniform branch each (a per-element select otherwise).  Rows past
      // M (clamped duplicates of the last row) are carried along: their columns are never stored.
      if (PRO == 2 && a.film) {
#pragma unroll
        for (int st = 0; st < NSTW; ++st)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const float4 fv = fs[st][hf], hv = fsh[st][hf];
            const float f4[4] = {fv.x, fv.y, fv.z, fv.w}, h4[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {              // t (scale + 1) + shift, with nothing to precompute on the loaded
              const float t = xr[st][4 * hf + e];      // values alone (hipcc hoists `scale + 1` up to the loads and
              xr[st][4 * hf + e] = t * f4[e] + (t + h4[e]);   // waits for them there, in front of barrier P)
            }
          }
      }
      if (a.silu) {
#pragma unroll
        for (int st = 0; st < NSTW; ++st)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float t = xr[st][e];
            xr[st][e] = t * __builtin_amdgcn_rcpf(1.0f + __expf(-t));
          }
      }
    }
    MDT_STAMP();                                     // normalised, FiLM, SiLU
    if constexpr (RTW == 2) {
      // exchange: [wave][k-step][hi | lo][lane] 16-byte entries; every wave reads back all NST k-steps of its row tile
      unsigned char* ex = smem + NS * SLOT;            // 32 KB behind the ring
#pragma unroll
      for (int st = 0; st < NSTW; ++st) {
        bf16x8 h, l;
        split8_rc(xr[st], h, l);
        *reinterpret_cast<bf16x8*>(ex + (((wave * NSTW + st) * 2 + 0) * 64 + lane) * 16) = h;
        *reinterpret_cast<bf16x8*>(ex + (((wave * NSTW + st) * 2 + 1) * 64 + lane) * 16) = l;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                    // X
#pragma unroll
      for (int st = 0; st < NST; ++st) {
        const int src = (2 * rt + st / NSTW) * NSTW + st % NSTW;
        xh[st] = *reinterpret_cast<const bf16x8*>(ex + ((src * 2 + 0) * 64 + lane) * 16);
        xl[st] = *reinterpret_cast<const bf16x8*>(ex + ((src * 2 + 1) * 64 + lane) * 16);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
#pragma unroll
      for (int st = 0; st < NST; ++st) split8_rc(xr[st], xh[st], xl[st]);
    }
  }
  MDT_STAMP();
  if (src == 0) {
    __builtin_amdgcn_s_barrier();                    // B(0)
    const unsigned l0 = lds_addr_rc(slot_of(0));
    const unsigned p0 = l0 + aP[0], p1 = l0 + aP[(RTW == 4) ? 0 : 1];
    frag_read(p0, J0{}, 0, J0{}); frag_read(p0, J0{}, 0, J1{}); frag_read(p0, J0{}, 0, J2{}); frag_read(p0, J0{}, 0, J3{});
    frag_read(p1, J1{}, 1, J0{}); frag_read(p1, J1{}, 1, J1{}); frag_read(p1, J1{}, 1, J2{}); frag_read(p1, J1{}, 1, J3{});
  }
  MDT_STAMP();

  // tiles in stream order; everything below is fully unrolled, so tile index, fragment-set rotation (tau NU mod 3)
  // and accumulator indices are compile-time
#pragma unroll
  for (int tap = 0; tap < TAPS; ++tap) {
#pragma unroll
    for (int kh = 0; kh < NKH; ++kh) {
      MDT_STAMP();
      bf16x8 oph[4], opl[4];                         // the 4 k-steps of this K half as seen by this tap
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int st = 4 * kh + k;
        if (TAPS == 1 || tap == 1) { oph[k] = xh[st]; opl[k] = xl[st]; }
        else if (tap == 0) { oph[k] = row_shift<true>(xh[st], has_prev); opl[k] = row_shift<true>(xl[st], has_prev); }
        else { oph[k] = row_shift<false>(xh[st], has_next_row); opl[k] = row_shift<false>(xl[st], has_next_row); }
      }
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int tau = (tap * NKH + kh) * NCH + c;       // tile within this source (compile-time)
        const int off = (tau * NU) % 3;
        const bool more = (tau + 1 < NT) || (src + 1 < nsrc);
        const unsigned lc = lds_addr_rc(slot_of(src * NT + tau)), ln = lds_addr_rc(slot_of(src * NT + tau + 1));
        unsigned bc[4], bn[2];
#pragma unroll
        for (int k = 0; k < 4; ++k) bc[k] = lc + aP[k];
        bn[0] = ln + aP[0];
        bn[1] = ln + aP[(RTW == 4) ? 0 : 1];
        auto unit = [&](auto uc) {
          constexpr int u = decltype(uc)::value;
          if (u == NU - 2 && more) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();            // B(src NT + tau + 1)
            __builtin_amdgcn_sched_barrier(0);
          }
          const int s0 = (off + u) % 3, s2 = (off + u + 2) % 3;
          constexpr bool in_tile = u + 2 < NU;
          const bool pre = in_tile || more;
          const bool later = (u + 1 < NU) || more;
          if (later) lgkm_wait_rc<4>(); else lgkm_wait_rc<0>();
          constexpr int ia = (RTW == 4) ? 2 * (u & 1) : 0, ib = (RTW == 4) ? (u >> 1) : u;
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
            acc[c][ia + q] = MDT_MFMA_BF16(w, x, acc[c][ia + q], 0, 0, 0);
          };
          if constexpr (F32) {
            // exact fp32: fragment (q, half) x operand half, four 16x16x4 MFMAs each, the two accumulators alternating
            auto mm4 = [&](const bf16x8& w0, const bf16x8& w1, const bf16x8& x, auto r0c) {
              constexpr int r0 = decltype(r0c)::value;
              const f32x4 a0 = __builtin_bit_cast(f32x4, w0), a1 = __builtin_bit_cast(f32x4, w1), xb = __builtin_bit_cast(f32x4, x);
#pragma unroll
              for (int r = r0; r < r0 + 2; ++r) {
                acc[c][ia] = MDT_MFMA_F32(a0[r], xb[r], acc[c][ia], 0, 0, 0);
                acc[c][ia + 1] = MDT_MFMA_F32(a1[r], xb[r], acc[c][ia + 1], 0, 0, 0);
              }
            };
            mm4(frh[s0][0], frh[s0][1], oph[ib], J0{}); rd(J0{});
            mm4(frh[s0][0], frh[s0][1], oph[ib], J2{}); rd(J1{});
            mm4(frl[s0][0], frl[s0][1], opl[ib], J0{}); rd(J2{});
            mm4(frl[s0][0], frl[s0][1], opl[ib], J2{}); rd(J3{});
          } else {
            mm(frl[s0][0], oph[ib], 0); rd(J0{});
            mm(frl[s0][1], oph[ib], 1); rd(J1{});
            mm(frh[s0][0], opl[ib], 0); rd(J2{});
            mm(frh[s0][1], opl[ib], 1); rd(J3{});
            mm(frh[s0][0], oph[ib], 0);
            mm(frh[s0][1], oph[ib], 1);
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
    }
  }

  if (src + 1 < nsrc && (NT * NU) % 3 != 0) {
    // the next source's code expects units 0 / 1 of its first tile in fragment sets 0 / 1
    constexpr int R = (NT * NU) % 3;
    bf16x8 th[3][2], tl[3][2];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int q = 0; q < 2; ++q) { th[j][q] = frh[(j + R) % 3][q]; tl[j][q] = frl[(j + R) % 3][q]; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the in-flight prefetch writes the old sets
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int q = 0; q < 2; ++q) { frh[j][q] = th[j][q]; frl[j][q] = tl[j][q]; }
  }
  }   // sources

  MDT_STAMP();
  // ---- out[m][64 c + 16 ft + 4 g + r] = acc (which started from bias + res) ----
  if (mvalid) {
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int q = 0; q < NFT; ++q) {
        const int f = ocol0 + 64 * c + 16 * (NFT * fh + q) + 4 * g;
        store_nt(a.out + (int64_t)m * a.ldc + f, make_float4(acc[c][q][0], acc[c][q][1], acc[c][q][2], acc[c][q][3]));
      }
  }
}

template <int RTW, int C, int TAPS, int NSPLIT, int NSRC, int PRO>
static hipError_t launch_rc2(const RConvArgs& a, hipStream_t s) {
  const size_t smem = (size_t)NS * SLOT + (RTW == 2 ? SLOT : 0);      // ring (+ operand exchange area)
  static DevOnce attr_once;                          // per device (mdt_kernels.h)
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rconv<RTW, C, TAPS, NSPLIT, NSRC, PRO, F32>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
  }
  const int rows = 16 * RTW;
  // half_out (NSPLIT = 2 forms only): ONE workgroup per row block, the one that produces output channels 0 .. C / 2 - 1
  hipLaunchKernelGGL((k_rconv<RTW, C, TAPS, NSPLIT, NSRC, PRO, F32>), dim3((unsigned)((a.M + rows - 1) / rows), a.half_out ? 1 : NSPLIT * (a.nb > 1 ? a.nb : 1)),
                     dim3(512), smem, s, a);
  return hipGetLastError();
}

template <int RTW, int C, int TAPS, int NSPLIT>
static hipError_t launch_rc(const RConvArgs& a, hipStream_t s) {
  if (!a.x2) return launch_rc2<RTW, C, TAPS, NSPLIT, 1, 2>(a, s);
  if (a.film) return hipErrorInvalidValue;            // a concatenated input never carries FiLM (block1 of a ResNet)
  return a.gsize > 0 ? launch_rc2<RTW, C, TAPS, NSPLIT, 2, 1>(a, s) : launch_rc2<RTW, C, TAPS, NSPLIT, 2, 0>(a, s);
}

#if MDT_TF_F32
hipError_t launch_rconv_f32(const RConvArgs& a, hipStream_t s) {
#else
bool rconv_supported(int C, int T, int taps, int gsize) {
  if (C != 128 && C != 256) return false;
  if (T <= 0 || 16 % T || (taps != 1 && taps != 3)) return false;
  return gsize == 0 || gsize == 4 || gsize == 8 || gsize == 16 || gsize == 32 || gsize == 64;
}

hipError_t launch_rconv(const RConvArgs& a, hipStream_t s) {
  if (a.wf32) return launch_rconv_f32(a, s);           // exact-fp32 products: the instantiations of k_rconv_f32.hip
#endif
  if (a.M <= 0) return hipSuccess;
  if (!rconv_supported(a.C, a.T, a.taps, a.gsize) || (a.gsize > 0 && (!a.gamma || !a.beta))) return hipErrorInvalidValue;
  if (a.x2 && a.film) return hipErrorInvalidValue;   // FiLM only ever precedes a single-source convolution
  if (a.nb > 1 && (a.nb > 8 || a.half_out || a.ksrc > 1 || a.x2 || a.film || a.gsize > 0)) return hipErrorInvalidValue;
  if (a.nb > 1) {                                    // NB x C output channels: NB convolutions of the same rows, no prologue
    if (a.C == 128) return a.taps == 3 ? launch_rc2<4, 128, 3, 1, 1, 0>(a, s) : launch_rc2<4, 128, 1, 1, 1, 0>(a, s);
    return a.taps == 3 ? launch_rc2<2, 256, 3, 2, 1, 0>(a, s) : launch_rc2<2, 256, 1, 2, 1, 0>(a, s);
  }
  if (a.half_out && (a.C != 256 || a.x2 || a.film || a.gsize > 0 || a.ksrc > 1)) return hipErrorInvalidValue;
  if (a.half_out) return a.taps == 3 ? launch_rc2<2, 256, 3, 2, 1, 0>(a, s) : launch_rc2<2, 256, 1, 2, 1, 0>(a, s);
  if (a.ksrc == 2) {                                 // two C-channel blocks of one tensor, taps as usual, no prologue
    if (a.x2 || a.film || a.gsize > 0 || a.C != 256) return hipErrorInvalidValue;
    return a.taps == 3 ? launch_rc2<2, 256, 3, 2, 2, 0>(a, s) : launch_rc2<2, 256, 1, 2, 2, 0>(a, s);   // (the unsplit forms spill)
  }
  if (a.ksrc > 1) {                                  // K = ksrc C input channels in consecutive C-channel blocks (MDT_R_KSRC)
    if (a.x2 || a.film || a.gsize > 0 || a.taps != 1 || a.ksrc * a.C != 1024) return hipErrorInvalidValue;
    if (a.C == 128) return launch_rc2<4, 128, 1, 1, 8, 0>(a, s);
    return launch_rc2<2, 256, 1, 2, 4, 0>(a, s);     // (always two workgroups per row block: the unsplit form spills 68 bytes)
  }
  // C = 128: 64-row workgroups (wave = row tile); C = 256: 32-row workgroups, features split over wave pairs (the
  // 64-row form needs 64 operand + 64 accumulator + 32 shifted-operand registers per lane and spills)
  if (a.C == 128) return a.taps == 3 ? launch_rc<4, 128, 3, 1>(a, s) : launch_rc<4, 128, 1, 1>(a, s);
  // two workgroups per row block (each half of the output channels) while that still fits one wave of workgroups
  if ((a.M + 31) / 32 <= 128) return a.taps == 3 ? launch_rc<2, 256, 3, 2>(a, s) : launch_rc<2, 256, 1, 2>(a, s);
  return a.taps == 3 ? launch_rc<2, 256, 3, 1>(a, s) : launch_rc<2, 256, 1, 1>(a, s);
}

}  // namespace mdt
