// Fused transformer sub-blocks (MDT_OP_TBLOCK variant 0; the maths and the register-chained layout: DESIGN.md 3.1), C = 128,
// 64-row workgroups, restructured around two measurements on gfx950 (tools/ubench/proj_phase.hip):
//
//   * with one wave per SIMD, an LDS-DMA instruction costs its issuing wave ~60 cycles, so the 8 pieces per tile
//     per wave that the compute waves used to issue cost ~500 cycles per tile, none of it overlapped.  Here FOUR
//     LOADER WAVES (waves 4..7, one per SIMD, next to the compute waves) own the whole weight stream; the compute
//     waves issue no vector-memory instruction between their prologue and epilogue except bias loads.
//   * fragment reads issued as a burst of 4 ds_read_b128 in front of 6 MFMAs stall the in-order issue while the LDS
//     pipe (shared by 4 waves) takes them: 27 cycles per MFMA.  Issued one at a time BETWEEN the MFMAs they hide in
//     the MFMA shadow, and when the read pipeline also runs across tile boundaries (the next tile is published two
//     units before the current one ends) the phase runs at ~22-24 cycles per MFMA (17.5 = bare issue rate).
//
// Ring protocol (4 slots of 32 KB, issue distance 2 tiles).  Barrier B(k) publishes tile k:
//   loader : wait until its own pieces of tile k have landed (vmcnt), s_barrier, issue tile k+2 into the slot of
//            tile k-2 (every compute wave is past unit 5 of tile k-1 at least, hence done with tile k-2);
//   compute: s_barrier at the point where it first wants to prefetch fragments of tile k.
// Every barrier is executed by all 8 waves, NT + 0 times in total.
#include <cstdlib>
#include <type_traits>

#include "mdt_kernels.h"

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

enum { TB_SELF = 0, TB_CROSS = 1, TB_FF = 2 };
enum { K_T = 0, K_N = 1, K_O = 2 };   // transposed projection, un-transposed projection, output projection

#define MDT_MFMA_BF16 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define MDT_MFMA_F32 __builtin_amdgcn_mfma_f32_16x16x4f32

__device__ __forceinline__ float gelu_lw(float x) {   // exact-erf GELU, branch-free erf (A&S 7.1.26, |error| < 1.5e-7)
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erfa = 1.0f - poly * __expf(-z * z);
  return 0.5f * x * (1.0f + copysignf(erfa, x));
}

// F32 (round 5, MDT_B_WF32): the values themselves, slots 0..3 in `hi`, 4..7 in `lo` -- operands of exact fp32 MFMAs on fp32 fragment
// tiles, as in k_tf128.hip (which grew out of this kernel and carries the same branches)
template <bool F32>
__device__ __forceinline__ void split8_lw(const float v[8], bf16x8& hi, bf16x8& lo) {
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

__device__ __forceinline__ void lds_read16(bf16x8& dst, const unsigned char* p) {
  const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)p;
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory");
}

// fragment read with the (tile, plane) part of the address in the instruction's immediate offset: per read there is no
// address arithmetic left (one v_add_u32 per ds_read_b128 was 127 of the 679 instructions of a self-attention head,
// in a kernel whose compute waves are issue-bound)
template <int OFF>
__device__ __forceinline__ void lds_read16_off(bf16x8& dst, unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read_b128 offset field");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}

// per-head bias vectors live in LDS behind the ring and are read like fragments (asm, counted in the lgkmcnt waits): a
// global load issued by a compute wave queues behind the loader waves' DMA traffic and stalls its issue ~60 cycles
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

// NPW (MODE_CROSS only): LDS-DMA pieces per loader wave per K / V tile = ceil(context rows of the workgroup / 16)
template <int MODE, int NPW, bool F32>
__global__ __launch_bounds__(512) void k_tblock_lw(TBlockArgs a) {
  constexpr int TPC = (MODE == TB_FF) ? 2 : 4;     // tiles per chunk: q k v o | q K V o (K, V = hoisted context rows) | w1 w2
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NX = (MODE == TB_FF) ? a.post : 0;     // extra output tiles of a folded 1x1 convolution (mdt_kernels.h)
  const int NT = a.nchunk * TPC + NX;
  const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.w);

  if (wave >= 4) {
    // ================= loader waves: the weight stream =================
    // Piece `inst` (= iw + 4 q) fills slot bytes [inst*1024, +1024), lane l supplies bytes inst*1024 + 16 l.  The
    // XOR swizzle that makes the fragment reads conflict-free is applied through the SOURCE address
    // (the address factors into a lane-only and a wave-uniform part).
    const int iw = wave - 4;
    __builtin_amdgcn_s_setprio(3);   // few instructions, all on the critical path of the stream: issue ahead of the MFMA waves
    const int lpP = lane >> 5;
    const int xP = (lane & 15) ^ lpP;
    const int baseP = ((lane >> 4) & 1) * (128 * C) + lpP * (2 * C);
    const int xO = (lane & 7) ^ (lane >> 4);
    const int baseO = (lane >> 3) * 128;
    // Per-lane source offsets of this wave's pieces inside a tile, computed ONCE: under MFMA load every VALU
    // instruction of a loader wave waits for an issue slot, and address arithmetic per piece is what made a piece
    // cost ~140 cycles instead of ~60.  Per piece there is now one scalar base + one 32-bit VGPR offset.
    unsigned voffP[IPT], voffO[IPT];
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
      const int inst = iw + 4 * q;
      const int U = 2 * inst;
      // (F32: fp32 fragment tiles are stored in LDS order, a linear copy)
      voffP[q] = F32 ? (unsigned)(inst * 1024 + lane * 16) : (unsigned)(U * (2 * C) + ((xP ^ (U & 15)) << 4) + baseP);
      voffO[q] = F32 ? (unsigned)(inst * 1024 + lane * 16)
                     : (unsigned)(((inst * 8) / C) * (128 * C) + ((inst * 8) % C) * 128 + ((xO ^ (4 * (inst & 1))) << 4) + baseO);
    }
    // MODE_CROSS: tiles 1 and 2 of a head are the K and V rows of the workgroup's samples, head h: row R = (sample,
    // key) concatenated, 256 B (64 fp32) per row, 16-byte chunks XOR-swizzled with R & 15 (conflict-free
    // ds_read_b128 over 16 keys, ds_read_b32 over 16 features x 4 keys).  A piece = 4 rows.
    const int kv_rows = (MODE == TB_CROSS) ? (64 / a.T) * a.Tk : 0;
    constexpr int npw = NPW;                         // rows are padded to 16 NPW with clamped (valid, unused) rows
    unsigned voffKV[4];                              // NPW <= 4 used (a fixed bound: hipcc's host pass silently drops the
                                                     // kernel stub when a lambda captures an array of dependent size)
    if constexpr (MODE == TB_CROSS) {
      const int sample0 = blockIdx.x * (64 / a.T);
      // dual batch (classifier-free guidance, both passes in one launch): the samples of the second half read the
      // batch-invariant K / V rows a.kv2 (a workgroup never straddles the halves: the host checks B % 16 == 0)
      const int bstr = (a.kv2 && sample0 >= a.nsamples / 2) ? 0 : a.kv_bstride;
#pragma unroll
      for (int q = 0; q < NPW; ++q) {
        const int R = 4 * (iw + 4 * q) + (lane >> 4);
        const int Rc = min(R, kv_rows - 1);
        const int sm = min(Rc / a.Tk, a.nsamples - 1 - sample0), key = Rc % a.Tk;
        voffKV[q] = (unsigned)(((sm * bstr + key) * a.ldkv + 4 * ((lane & 15) ^ (R & 15))) * 4);
      }
    }
    auto pieces_of = [&](int tau) -> int {           // vector-memory operations this wave issues for tile tau
      if constexpr (MODE == TB_CROSS) return ((tau & 3) == 1 || (tau & 3) == 2) ? npw : IPT;
      return IPT;
    };
    auto issue_kv = [&](int tau) {
      if constexpr (MODE == TB_CROSS) {
        unsigned char* slot = smem + (tau % NS) * SLOT + iw * 1024;
        const int sample0 = blockIdx.x * (64 / a.T);
        const bool second = a.kv2 && sample0 >= a.nsamples / 2;
        const unsigned char* base = reinterpret_cast<const unsigned char*>(
            (second ? a.kv2 : a.kv + (int64_t)sample0 * a.kv_bstride * a.ldkv) + 64 * (tau >> 2) + ((tau & 3) == 2 ? 64 * a.nheads : 0));
#pragma unroll
        for (int q = 0; q < NPW; ++q)
          __builtin_amdgcn_global_load_lds(base + voffKV[q], (__attribute__((address_space(3))) void*)(slot + q * 4096), 16, 0, 0);
      }
    };
    auto issue_w = [&](int tau) {
      unsigned char* slot = smem + (tau % NS) * SLOT + iw * 1024;
      const int j = tau % TPC;
      const int wt = (MODE == TB_CROSS) ? 2 * (tau >> 2) + (j == 3) : tau;      // index into the weight-tile stream
      const unsigned char* tile = wsrc + (int64_t)wt * SLOT;       // wave-uniform
      // per element select (not two loops over two arrays: hipcc then indexes a merged array dynamically, puts it in
      // scratch and waits for every scratch load with vmcnt(0), which serialises the whole DMA stream)
      const bool ptile = (j != TPC - 1 && tau < a.nchunk * TPC);
#pragma unroll
      for (int q = 0; q < IPT; ++q) {
        const unsigned off = ptile ? voffP[q] : voffO[q];
        __builtin_amdgcn_global_load_lds(tile + off, (__attribute__((address_space(3))) void*)(slot + q * 4096), 16, 0, 0);
      }
    };
    auto issue_tile = [&](int tau) {
      const bool kv = (MODE == TB_CROSS) && ((tau & 3) == 1 || (tau & 3) == 2);
      if (kv) issue_kv(tau);
      else issue_w(tau);
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
    __builtin_amdgcn_s_barrier();   // P: the compute waves' row and bias loads are queued ahead of the stream
    issue_tile(0);
    if (NT > 1) issue_tile(1);
#ifdef MDT_STAMPS
    unsigned long long* lst = reinterpret_cast<unsigned long long*>(const_cast<float*>(a.dbgbuf)) + 128;
    int nl = 0;
#define MDT_LSTAMP()                                                                  \
  do {                                                                                \
    if (a.dbgbuf && blockIdx.x == 0 && wave == 4 && nl < 120) {                       \
      unsigned long long t_;                                                          \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");      \
      if (lane == 0) lst[nl] = t_;                                                    \
      ++nl;                                                                           \
    }                                                                                 \
  } while (0)
#else
#define MDT_LSTAMP() do {} while (0)
#endif
    for (int k = 0; k < NT; ++k) {
      MDT_LSTAMP();
      // tile k landed; tile k + 1 may be in flight (the weight-tile case first: the switch is a cascade of scalar branches, k_tf256.hip)
      if (k + 1 < NT && pieces_of(k + 1) == IPT) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else wait_vm(k + 1 < NT ? pieces_of(k + 1) : 0);
      MDT_LSTAMP();
      __builtin_amdgcn_s_barrier();                                      // B(k)
      MDT_LSTAMP();
      if (k + 2 < NT) issue_tile(k + 2);
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

  // this wave's 16 rows in MFMA operand layout: lane (i, g) holds x[i][32 st + 8 g + e]
  bf16x8 xh[NST], xl[NST];
  constexpr int NBV = 4;                 // per-chunk bias vectors: 64 nchunk floats over 256 lanes, nchunk <= 16
  float bv[NBV];
  {
    float xr[NST][8];
    const float* xp = a.x + (int64_t)mc * a.ldx + 8 * g;
    float s = 0.f;
    float4 xu[NST], xw[NST];
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      xu[st] = *reinterpret_cast<const float4*>(xp + 32 * st);
      xw[st] = *reinterpret_cast<const float4*>(xp + 32 * st + 4);
    }
#pragma unroll
    for (int k = 0; k < NBV; ++k) bv[k] = (tid + 256 * k < 64 * a.nchunk) ? a.bias[tid + 256 * k] : 0.f;
    // P: the loader waves start the weight stream only now, behind this wave's requests (queued behind the stream's
    // first two tiles the rows came back ~1500 cycles later).  sched_barrier: without it hipcc moves the rows' first
    // uses (and with them the wait for the loads) in front of the barrier, i.e. the stream starts a round trip late
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      const float4 u = xu[st], w = xw[st];
      xr[st][0] = u.x; xr[st][1] = u.y; xr[st][2] = u.z; xr[st][3] = u.w;
      xr[st][4] = w.x; xr[st][5] = w.y; xr[st][6] = w.z; xr[st][7] = w.w;
#pragma unroll
      for (int e = 0; e < 8; ++e) s += xr[st][e];
    }
    float mean = 0.f, rstd = 1.f;
    if constexpr (MODE != TB_FF) {       // nn.LayerNorm statistics (two-pass; gain/bias folded into the weights)
      s = xg16_add(s);
      s = xg32_add(s);
      mean = s / (float)C;
      float ss = 0.f;
#pragma unroll
      for (int st = 0; st < NST; ++st)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = xr[st][e] - mean;
          ss += d * d;
        }
      ss = xg16_add(ss);
      ss = xg32_add(ss);
      rstd = 1.0f / sqrtf(ss / (float)C + a.eps);
    }
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = mvalid ? (xr[st][e] - mean) * rstd : 0.f;
      split8_lw<F32>(v, xh[st], xl[st]);
    }
  }

  // Fragment addressing: lane-dependent swizzled part per k-step (projection tiles) / per k-half (output tiles);
  // the 16-row tile and the hi/lo plane are compile-time immediates of the ds_read_b128.
  //   projection tile: row = 16 t + i, chunk = 4 st + g :  row*4C + plane*2C + ((chunk&~15) | ((chunk&15)^(row&15)))*16
  //   output tile:     row = 16 t + i, chunk = 4 sp + g :  plane*C*128 + row*128 + (chunk ^ ((row>>1)&7))*16
  int aP[NST], aO[2];
#pragma unroll
  for (int st = 0; st < NST; ++st) {
    const int lc = 4 * st + g;
    aP[st] = F32 ? lane * 16 : i * (4 * C) + ((lc & ~15) | ((lc & 15) ^ i)) * 16;      // F32: the k-step is in the immediate
  }
#pragma unroll
  for (int sp = 0; sp < 2; ++sp) aO[sp] = F32 ? lane * 16 : i * 128 + ((4 * sp + g) ^ ((i >> 1) & 7)) * 16;

  bf16x8 fh[3][2], fl[3][2];   // three fragment sets: unit u of a phase with set offset OFF lives in set (OFF + u) % 3
  // Read j (= 2 q + plane) of unit u of a tile of kind KIND; `base` = LDS address of the slot + the lane's swizzled part
  // for the unit's k-step (projection tiles: aP[u >> 1]) or k-half (output tiles: aO[u / 4]); the rest is immediate.
  auto frag_read = [&](auto kind, unsigned base, auto uc, int set, auto jc) {
    constexpr int KIND = decltype(kind)::value, u = decltype(uc)::value, j = decltype(jc)::value;
    constexpr int q = j >> 1, lo = j & 1;
    constexpr int off = F32 ? ((KIND == K_O) ? ((2 * (u % (NCT / 2)) + q) * 4096 + (u / (NCT / 2)) * 2048 + lo * 1024)
                                             : ((2 * (u & 1) + q) * 8192 + (u >> 1) * 2048 + lo * 1024))
                            : ((KIND == K_O) ? ((2 * (u % (NCT / 2)) + q) * 16 * 128 + lo * (C * 128))
                                             : ((2 * (u & 1) + q) * 16 * 4 * C + lo * (2 * C)));
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

#ifdef MDT_STAMPS   // tuning build: wave 0 of workgroup 0 records the shader clock at phase boundaries into dbgbuf
  unsigned long long* stamps = reinterpret_cast<unsigned long long*>(const_cast<float*>(a.dbgbuf));
  int nstamp = 0;
#define MDT_STAMP()                                                                   \
  do {                                                                                \
    if (stamps && blockIdx.x == 0 && wave == 0 && nstamp < 120) {                     \
      unsigned long long t_;                                                          \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");      \
      if (lane == 0) stamps[nstamp] = (t_ & 0xffffffffffffull) | ((unsigned long long)__LINE__ << 48);  \
      ++nstamp;                                                                       \
    }                                                                                 \
  } while (0)
#else
#define MDT_STAMP() do {} while (0)
#endif
  int tau = 0;                                       // tile being consumed
  auto slot_of = [&](int t) -> const unsigned char* { return smem + (t % NS) * SLOT; };

  // One MFMA phase over the tile in `cur`: 8 units.  acc[] are the phase's accumulators, bh/bl the register-resident
  // operand (activations) per k-step.  The reads of unit u+2 ride between the MFMAs of unit u; for u+2 >= NU they
  // belong to units 0/1 of the NEXT phase (kind NK, in tile tau+1), published by the barrier before unit NU-2.
  auto phase = [&](auto kind, auto offc, auto nkind, bool has_next, f32x4* acc, const bf16x8* bh, const bf16x8* bl) {
    constexpr int KIND = decltype(kind)::value, OFF = decltype(offc)::value, NK = decltype(nkind)::value;
    const unsigned lc = lds_addr(slot_of(tau)), ln = lds_addr(slot_of(tau + 1));
    unsigned bc[4];                                  // per k-step (projection) / k-half (output: two used)
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
      const bool later = (u + 1 < NU) || has_next;   // unit u+1's reads are in flight behind unit u's
      if (later) lgkm_wait<4>(); else lgkm_wait<0>();
      constexpr int ia = (KIND == K_O) ? 2 * (u % (NCT / 2)) : 2 * (u & 1);   // accumulator index
      constexpr int ib = (KIND == K_O) ? u / (NCT / 2) : (u >> 1);             // operand index
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
      if constexpr (F32) {
        // exact fp32: fragment (q, half) x operand half, four 16x16x4 MFMAs each (r = contraction sub-step); the two accumulators
        // alternate so that no MFMA waits for the one issued just before it (k_tf128.hip)
        auto mm4 = [&](const bf16x8& w0, const bf16x8& w1, const bf16x8& x, auto r0c) {
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
#ifdef MDT_STAMPS_UNITS
      MDT_STAMP();
#endif
    };
    unit(std::integral_constant<int, 0>{}); unit(std::integral_constant<int, 1>{});
    unit(std::integral_constant<int, 2>{}); unit(std::integral_constant<int, 3>{});
    unit(std::integral_constant<int, 4>{}); unit(std::integral_constant<int, 5>{});
    unit(std::integral_constant<int, 6>{}); unit(std::integral_constant<int, 7>{});
    ++tau;
    MDT_STAMP();
  };
  using IC0 = std::integral_constant<int, 0>;
  using IC1 = std::integral_constant<int, 1>;
  using IC2 = std::integral_constant<int, 2>;
  const IC0 kT{};   // K_T
  const IC1 kN{};   // K_N
  const IC2 kO{};   // K_O

  f32x4 accT[NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) accT[ct] = f32x4{0.f, 0.f, 0.f, 0.f};

  const float* bias = a.bias;                        // [bq | bk | bv | bo] / [b1 | b2], global (L2-resident)
  const int bo_off = (MODE == TB_SELF) ? 3 * 64 * a.nchunk : 64 * a.nchunk;   // [bq | bk | bv | bo] / [bq | bo] / [b1 | b2]
  // Loop-invariant softmax pieces: additive mask of key j = 4 g + r against query column i (other samples of the
  // wave's 16 rows -> -inf), and the logit scale folded with log2(e) so that the exponential is one v_exp_f32.
  const int samp_q = i / a.T;                        // sample (within the wave's 16 rows) of query column i
  float kmask[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) kmask[r] = ((4 * g + r) / a.T == samp_q) ? 0.f : -INFINITY;
  const float scale2 = a.scale * 1.44269504088896340736f;
  // MODE_CROSS (one key tile: at most 16 context rows per wave): LDS addresses of this wave's K / V rows inside a
  // K or V tile and the validity of key column 4 g + r for query column i.  Rows past the wave's keys are clamped to
  // a real row (finite data): their scores are masked by select, their probabilities are exactly 0.
  int aK = 0, xK = 0, aV[4] = {0, 0, 0, 0}, xV[4] = {0, 0, 0, 0};
  bool kok[4] = {false, false, false, false};
  if constexpr (MODE == TB_CROSS) {
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

  // the first 64 * nchunk bias entries (bq | b1: the per-chunk vectors) -> LDS behind the ring
  float* bias_s = reinterpret_cast<float*>(smem + NS * SLOT);
#pragma unroll
  for (int k = 0; k < NBV; ++k)
    if (tid + 256 * k < 64 * a.nchunk) bias_s[tid + 256 * k] = bv[k];
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                      // B(0)
  prefetch2(kT, slot_of(0), 0);
  const unsigned bias_l = lds_addr(reinterpret_cast<const unsigned char*>(bias_s)) + 16 * g;   // + 256 h per chunk

  MDT_STAMP();
  for (int h = 0; h < a.nchunk; ++h) {
    const bool more = h + 1 < a.nchunk;
    f32x4 oT[4];
    if constexpr (MODE == TB_FF) {
      f32x4 b1[4];
      lds_read_f4_off<0>(b1[0], bias_l + 256 * h); lds_read_f4_off<64>(b1[1], bias_l + 256 * h);
      lds_read_f4_off<128>(b1[2], bias_l + 256 * h); lds_read_f4_off<192>(b1[3], bias_l + 256 * h);
#pragma unroll
      for (int ft = 0; ft < 4; ++ft) oT[ft] = zero4;
      phase(kT, IC0{}, kT, false, oT, xh, xl);       // hidden chunk^T = W1 x^T
      __builtin_amdgcn_s_barrier();                  // B(w2 tile)
      prefetch2(kO, slot_of(tau), 1);
#pragma unroll
      for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int r = 0; r < 4; ++r) oT[ft][r] = gelu_lw(oT[ft][r] + b1[ft][r]);
    } else if constexpr (MODE == TB_CROSS) {
      f32x4 qT[4], bq[4];
      lds_read_f4_off<0>(bq[0], bias_l + 256 * h); lds_read_f4_off<64>(bq[1], bias_l + 256 * h);
      lds_read_f4_off<128>(bq[2], bias_l + 256 * h); lds_read_f4_off<192>(bq[3], bias_l + 256 * h);
#pragma unroll
      for (int ft = 0; ft < 4; ++ft) qT[ft] = zero4;
      phase(kT, IC0{}, kT, false, qT, xh, xl);       // q^T
      __builtin_amdgcn_s_barrier();                  // B(K tile)
      const unsigned char* sk = slot_of(tau);
      float4 kk[4];                                  // A operand of S^T: K[key i][64 h + 16 ft + 4 g + s]
#pragma unroll
      for (int ft = 0; ft < 4; ++ft)
        kk[ft] = *reinterpret_cast<const float4*>(sk + aK + (((4 * ft + g) ^ xK) << 4));
#pragma unroll
      for (int ft = 0; ft < 4; ++ft) qT[ft] += bq[ft];
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
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the V reads above are complete before the slot can be refilled
      ++tau;
      __builtin_amdgcn_s_barrier();                  // B(output tile)
      prefetch2(kO, slot_of(tau), 1);
    } else {
      f32x4 qT[4], kTt[4], vT[4];
      f32x4 bq[4];       // the host folds the k bias away (softmax-invariant) and the v bias into the output bias
      lds_read_f4_off<0>(bq[0], bias_l + 256 * h); lds_read_f4_off<64>(bq[1], bias_l + 256 * h);
      lds_read_f4_off<128>(bq[2], bias_l + 256 * h); lds_read_f4_off<192>(bq[3], bias_l + 256 * h);
#pragma unroll
      for (int ft = 0; ft < 4; ++ft) { qT[ft] = zero4; kTt[ft] = zero4; vT[ft] = zero4; }
      phase(kT, IC0{}, kT, true, qT, xh, xl);        // q^T
      phase(kT, IC2{}, kN, true, kTt, xh, xl);       // k^T
      phase(kN, IC1{}, kN, false, vT, xh, xl);       // v (un-transposed)
      __builtin_amdgcn_s_barrier();                  // B(output tile)
      prefetch2(kO, slot_of(tau), 1);
#pragma unroll
      for (int ft = 0; ft < 4; ++ft) qT[ft] += bq[ft];
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
    }
    MDT_STAMP();
    // ---- output projection of this chunk: accT[c][i] += sum_d Wo[c][64 h + d] * o[d][i] ----
    bf16x8 oh[2], ol[2];
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = oT[2 * sp + (e >> 2)][e & 3];
      split8_lw<F32>(v, oh[sp], ol[sp]);
    }
    if (NX > 0 && !more) phase(kO, IC1{}, kO, true, accT, oh, ol);   // the folded convolution's tiles follow
    else phase(kO, IC1{}, kT, more, accT, oh, ol);
  }
  if constexpr (MODE == TB_FF) {
    if (NX > 0) {                                    // + Wout x: two more output tiles on the raw x operands
      phase(kO, IC0{}, kO, true, accT, xh, xl);
      phase(kO, IC2{}, kT, false, accT, xh + 2, xl + 2);
    }
  }

  // ---- residual + output bias: x[m][16 ct + 4 g + r] += accT[ct][r] + bo[..] ----
  // Every load is requested before the first store: the output aliases the residual rows (and, for all hipcc knows,
  // the bias), and with loads and stores alternating it kept them in order -- eight exposed round trips.
  if (mvalid) {
    const float* xi = a.x + (int64_t)m * a.ldx + 4 * g;
    float* xo = (NX > 0 ? a.xout : a.x) + (int64_t)m * a.ldx + 4 * g;   // folded convolution: separate output tensor
    float4 bo[NCT], xr[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) bo[ct] = *reinterpret_cast<const float4*>(bias + bo_off + 16 * ct + 4 * g);
    if (NX == 0) {
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) xr[ct] = *reinterpret_cast<const float4*>(xi + 16 * ct);
    } else {                                         // folded convolution: no residual
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) xr[ct] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
      store_nt(xo + 16 * ct, make_float4(accT[ct][0] + bo[ct].x + xr[ct].x, accT[ct][1] + bo[ct].y + xr[ct].y,
                                         accT[ct][2] + bo[ct].z + xr[ct].z, accT[ct][3] + bo[ct].w + xr[ct].w));
  }
}

template <int MODE, int NPW, bool F32>
static hipError_t launch_lw2(const TBlockArgs& a, hipStream_t s) {
  const size_t smem = (size_t)NS * SLOT + (size_t)64 * a.nchunk * sizeof(float);   // ring + per-chunk bias vectors
  static DevOnce attr_once;                          // per device (mdt_kernels.h)
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tblock_lw<MODE, NPW, F32>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
  }
  hipLaunchKernelGGL((k_tblock_lw<MODE, NPW, F32>), dim3((unsigned)((a.M + 63) / 64)), dim3(512), smem, s, a);
  return hipGetLastError();
}

template <int MODE, int NPW = 0>
static hipError_t launch_lw(const TBlockArgs& a, hipStream_t s) {
  return a.wf32 ? launch_lw2<MODE, NPW, true>(a, s) : launch_lw2<MODE, NPW, false>(a, s);   // fp32 fragment tiles: exact fp32 products
}

bool tblock_lw_supported(const TBlockArgs& a) {
  if (a.C != 128 || a.T <= 0 || 16 % a.T || a.nchunk <= 0 || a.nchunk > 16) return false;
  if (a.post && (a.mode != TB_FF || a.post != 2 || !a.xout)) return false;
  if (a.mode == TB_CROSS) return a.Tk > 0 && (16 / a.T) * a.Tk <= 16;   // one key tile per wave, K / V tile <= 64 rows
  return a.mode == TB_SELF || a.mode == TB_FF;
}

hipError_t launch_tblock_lw(const TBlockArgs& a, hipStream_t s) {
  if (a.M <= 0) return hipSuccess;
  if (!tblock_lw_supported(a)) return hipErrorInvalidValue;
  if (a.mode == TB_CROSS) {
    switch (((64 / a.T) * a.Tk + 15) / 16) {
      case 1: return launch_lw<TB_CROSS, 1>(a, s);
      case 2: return launch_lw<TB_CROSS, 2>(a, s);
      case 3: return launch_lw<TB_CROSS, 3>(a, s);
      case 4: return launch_lw<TB_CROSS, 4>(a, s);
      default: return hipErrorInvalidValue;          // unreachable: at most 16 context rows per wave = 64 per workgroup
    }
  }
  return a.mode == TB_SELF ? launch_lw<TB_SELF>(a, s) : launch_lw<TB_FF>(a, s);
}

}  // namespace mdt
