// MDT_OP_RES256: a CHAIN of ResnetBlock1d blocks (modules.py:145-205) of a 256-channel level in ONE launch.
//
//   kind 1 (down path / bottleneck):  x = Block(x), N_RES times; every block's output is also stored as a skip tensor
//   kind 2 (up path):                 x = Block(cat([x, s * skip[rb]]))   (UpsampleBlock1d.add_skip, modules.py:828-829)
//   Block(u) = conv2(silu(GroupNorm(h) * (scale + 1) + shift)) + bias2 + to_out(u),  h = conv1(silu(GroupNorm(u))) + bias1
//
// Why: as one k_rconv launch per convolution the 256-channel level's ResNet path is 26 launches per evaluation at BASELINE
// configs[1] (28 at configs[2]'s one-token level), each ~9-11 us of kernel time for 1-3 us of streamed MFMA work -- kernel
// arguments, the rows' round trip, GroupNorm, the ring fill and the drain are paid per launch (profiles/r4_rconv_stamps.txt: 9.7 k
// of 17 k cycles pass before the first MFMA; a one-workgroup launch takes as long as a full one) -- plus a 1.65 us launch gap.
// Here a 32-row workgroup keeps its rows for the whole chain; per convolution only the weight stream and the GroupNorm /
// operand exchange between the two waves of a row tile remain.  Every workgroup computes ALL 256 output channels of every
// convolution (the per-launch form splits them over two workgroups: a chain cannot, the next convolution needs all channels of
// these rows): twice the stream per workgroup and half the workgroups at B = 1024 -- still the better trade, the launches were
// latency-, not throughput-bound.
//
// Structure = k_rconv.hip (RTW = 2: wave = (row tile rt, feature half fh), operands exchanged between the two waves of a row tile
// through LDS) + the descriptor-driven loader of k_tf128.hip / k_tf256.hip:
//   * the residual stream, the first convolution's output hT and the residual convolution's output rT live in ACCUMULATOR layout:
//     wave (rt, fh), acc[c][q][r] of lane (i, g) = value[row 16 rt + i][channel 128 fh + 32 c + 16 q + 4 g + r].  The host packs
//     the rows of every [64 features][128 k] weight sub-tile accordingly (rows 0..31 = channels 32 c .., rows 32..63 = channels
//     128 + 32 c ..), so that a wave owns a CONTIGUOUS half of the channels: GroupNorm groups of 32 channels are one chunk c, groups
//     of 64 (the 2C-channel input of kind 2: 512 / 8) two chunks, both wave-local (4 lane groups by v_permlane swaps, the sample's
//     token lanes by DPP);
//   * accumulator -> operand without lane movement (k_tf128.hip): k-slot (st, g, e) of the next convolution = channel
//     16 (2 st + (e >> 2)) + 4 g + (e & 3), i.e. k-step st = 4 fh + c is built from acc[c][0..1] of the same lane; the host permutes
//     the K columns of every sub-tile (compiler.py::_ACC_PERM, 256 entries).  A wave builds its four k-steps, the pair exchanges
//     them through a SCRATCH tile of the ring (descriptor kind 2 / 3: no weight DMA, one barrier);
//   * the +-1 taps are the operand registers shifted by one lane inside the 16-lane row (k_rconv.hip);
//   * kind 2: the skip rows of a block (32 rows x 1 KB = one ring slot) travel through the ring as a tile of their own (kind 1),
//     twice per block (residual convolution, block1), so that they are never live across a k = 3 convolution;
//   * per-block vectors ([g1 | b1 | bias1 | g2 | b2 | bias2], kind 2: [g1 (2C) | b1 (2C) | bias1 | bias_r | g2 | b2 | bias2]) and the
//     block's FiLM row arrive by LDS-DMA in a double-buffered area behind the ring, one block ahead (descriptor kind 3).
// Tile stream per block, TAPS = 1 | 3 (1: one token per sample, only the centre tap sees data), nt = 8 TAPS sub-tiles per
// convolution in (tap, K half, chunk) order:
//   kind 1:  X  nt (block1)  X  nt (block2)
//   kind 2:  X  nt (block1 on x)  X  8 (to_out on x)  S  X  8 (to_out on skip)  S  X  nt (block1 on skip)  X  nt (block2)
// preceded by ONE vector tile (kind 3, block 0).  X = scratch tile (kind 2, or 3 = + the NEXT block's vectors), S = skip rows.
//
// NSPLIT = 2 (round 6; the form of batches whose 32-row blocks do not fill the chip: the narrow program, B <= 1024 at 4 tokens per
// sample): a PAIR of workgroups per row block, as k_tf256.hip's pair split.  Half hh streams only the sub-tiles of output chunks
// 2 hh, 2 hh + 1 of every convolution (half the weight stream per compute unit -- what the per-convolution launches of k_rconv.hip
// bought by splitting the output channels over two workgroups -- without their 26 launches); behind every COMPLETED accumulator
// set (the first convolution's output; the block's output sum) the two workgroups hand each other their chunks inside the launch
// (16 KB each way, wave by wave: 16-byte sc1 stores, s_waitcnt vmcnt(0), the wave's flag store; poll of the partner wave's flag,
// sc1 loads -- the forms, the flag lines and the hand-off blocks of k_tf256.hip, which see for the safety argument; see handoff()),
// so both hold identical full rows for GroupNorm and the next convolution's operands.  The tile stream is the unsplit one with half
// the weight runs; the chain's last output leaves the kernel by halves, without a hand-off.
// One descriptor table serves both halves (equal run lengths); half hh's sub-tiles start at w + hh * (sub-tiles per half).
#include <cstdlib>
#include <type_traits>

#include "mdt_kernels.h"

#define MDT_SLOT_IDX(t) ((t) & (NS - 1))

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
#undef MDT_XG

enum { D_W = 0, D_SKIP = 1, D_X = 2, D_XV = 3 };     // tile descriptor kinds (2 bits), aux = descriptor >> 2

#define MDT_MFMA_BF16 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define MDT_MFMA_F32 __builtin_amdgcn_mfma_f32_16x16x4f32

template <bool F32>
__device__ __forceinline__ void split8_rs(const float v[8], bf16x8& hi, bf16x8& lo) {
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

__device__ __forceinline__ unsigned lds_addr(const unsigned char* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)p;
}

template <int N>
__device__ __forceinline__ void lgkm_wait() {   // at most N LDS operations still in flight
  if constexpr (N >= 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

template <bool SHR>      // operand of the neighbouring token row (k_rconv.hip): lane i takes lane i - 1 (SHR) / i + 1, 0 at the ends
__device__ __forceinline__ bf16x8 row_shift_rs(const bf16x8& v, bool keep) {
  const i32x4 s = __builtin_bit_cast(i32x4, v);
  i32x4 r;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int t = __builtin_amdgcn_update_dpp(0, s[k], SHR ? 0x111 : 0x101, 0xf, 0xf, true);
    r[k] = keep ? t : 0;
  }
  return __builtin_bit_cast(bf16x8, r);
}

constexpr int C = 256;          // channels
constexpr int CS = 128;         // sub-tile k-width
constexpr int SLOT = 256 * CS;  // bytes per sub-tile (bf16 hi plane + lo plane, or fp32 fragments)
constexpr int NS = 4;           // ring slots
constexpr int IPT = CS / 16;    // DMA pieces per sub-tile per loader wave
constexpr int NST = C / 32;     // k-steps of the input channels
constexpr int NU = 4;           // units (4 fragment reads + 6 MFMAs) per sub-tile per wave
constexpr int NCH = C / 64;     // 64-feature chunks of an output
constexpr int VPAR = 12 * 1024; // bytes of one parity of the vector area (kind 2: 9 C + 2 C floats = 11 KB)
// pair hand-off (NSPLIT = 2): k_tf256.hip's blocks and flag lines (one 32 KB block per (parity, row block, half); this kernel fills
// the first 16 KB), its cache policies (sc1 stores / sc1 loads: the measured-valid row of MI355X_MICROARCH.md) and its time-out
constexpr unsigned XBLOCK = 32 * C * 4;
constexpr int AUX_ST = 16, AUX_LD = 16;
constexpr unsigned long long POLL_TIMEOUT = 30000000ull;   // s_memrealtime ticks (100 MHz): 0.3 s

}  // namespace

// RES: 1 single source + skip stores, 2 two sources (see the head of the file).  TAPS: 3, or 1 for one token per sample (the
// block convolutions' centre tap only; the residual convolution of kind 2 is always k = 1).  F32: fp32 fragment sub-tiles and
// exact fp32 MFMA products, as k_tf256.hip.
// NSPLIT: 1 = one workgroup per 32-row block; 2 = a pair of workgroups per block (head of the file).
template <int RES, int TAPS, bool F32, int NSPLIT>
__global__ __launch_bounds__(512) void k_res256(TFArgs a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* vec_b = smem + NS * SLOT;             // two parities of VPAR bytes: [block vectors | FiLM row (2 C)]
  constexpr int VB = (RES == 1 ? 6 : 9) * C;           // floats of a block's vectors
  constexpr int NVP = VB / 256 + 2;                    // 1 KB pieces of a block's vectors + its FiLM row

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NT = a.NT;
  const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.w);
  // row block and half of this workgroup (k_tf256.hip): linear id = (group, half, member), partners PAIR_STRIDE ids apart
  int rb_ = blockIdx.x, hh = 0;
  const int nrb = (a.M + 31) / 32;
  if constexpr (NSPLIT == 2) {
    const int S = a.pair_stride, id = blockIdx.x;
    const int grp = id / (2 * S), w = id - grp * 2 * S;
    hh = w / S;
    rb_ = a.rb_base + grp * S + (w - hh * S);
    if (rb_ >= nrb) return;                            // padding of the last group: both partners leave
  }
  const int rowb = rb_ * 32;

  if (wave >= 4) {
    // ================= loader waves: descriptor-driven stream (k_tf128.hip / k_tf256.hip) =================
    const int iw = wave - 4;
    __builtin_amdgcn_s_setprio(MDT_LOADER_PRIO);
    const cu32p tiles = (cu32p)a.tiles;
    const int lpP = lane >> 5;
    const int xP = (lane & 15) ^ lpP;
    const int baseP = ((lane >> 4) & 1) * (128 * CS) + lpP * (2 * CS);
    unsigned voffP[IPT];
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
      const int inst = iw + 4 * q;
      const int U = 2 * inst;
      voffP[q] = F32 ? (unsigned)(inst * 1024 + lane * 16) : (unsigned)(U * (2 * CS) + ((xP ^ (U & 15)) << 4) + baseP);
    }
    auto pieces_of = [&](unsigned d) -> int {
      const unsigned kind = d & 3u;
      if (kind == D_X) return 0;
      if (kind == D_XV) return (NVP - iw + 3) / 4;
      return IPT;
    };
    auto issue_w = [&](unsigned char* slot, const unsigned char* tile) {       // tile: wave-uniform
      // The per-lane offsets pass through an empty asm so that their zero-extension stays in THIS basic block: hoisted out of the
      // tile loop as 64-bit register pairs, the address became a 64-bit VALU add per piece and the instruction took the
      // vector-address form with ONE register pair rewritten between the pieces.  With a 32-bit offset next to a scalar base hipcc
      // selects global_load_lds_dwordx4 v, s[base] (k_rconv.hip's form).
      unsigned off[IPT];                                 // (all eight first: one register each, no piece waits for the previous
#pragma unroll                                           //  one to release its address register)
      for (int q = 0; q < IPT; ++q) {
        off[q] = voffP[q];
        asm volatile("" : "+v"(off[q]));
      }
#pragma unroll
      for (int q = 0; q < IPT; ++q)
        __builtin_amdgcn_global_load_lds(tile + off[q], (__attribute__((address_space(3))) void*)(slot + q * 4096), 16, 0, 0);
    };
    auto issue_tile = [&](int tau, unsigned d) {
      unsigned char* slot = smem + MDT_SLOT_IDX(tau) * SLOT + iw * 1024;
      const unsigned kind = d & 3u, aux = d >> 2;
      if (kind == D_W) {                               // (the common case first: every test is a scalar branch in the turn)
        issue_w(slot, wsrc + (int64_t)aux * SLOT);
        return;
      }
      if (kind == D_X) return;
      if (kind == D_XV) {                              // block aux: vectors -> parity aux & 1, FiLM row behind them
        unsigned char* dst = vec_b + (aux & 1u) * VPAR;
        const unsigned char* vsrc = reinterpret_cast<const unsigned char*>(a.vec + (int64_t)aux * VB);
        const unsigned char* fsrc = reinterpret_cast<const unsigned char*>(a.film + (int64_t)aux * 2 * C);
        for (int q = iw; q < NVP; q += 4) {
          const unsigned char* s = q < VB / 256 ? vsrc + q * 1024 : fsrc + (q - VB / 256) * 1024;
          __builtin_amdgcn_global_load_lds(s + lane * 16, (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, 0, 0);
        }
        return;
      }
      if (kind == D_SKIP) {
        if constexpr (RES == 2) {
          // skip rows of block aux: row R of the workgroup = piece R (1 KB); 16-byte chunks XOR-swizzled with R & 15 inside each 256
          // bytes through the SOURCE address (conflict-free float4 reads of 16 rows x one chunk, any of the real b128 lane groups)
          const unsigned char* base = reinterpret_cast<const unsigned char*>(
              a.skip + (int64_t)aux * a.skip_stride + (int64_t)rowb * C);
#pragma unroll
          for (int q = 0; q < IPT; ++q) {
            const int R = iw + 4 * q;
            const int ch = (lane & 48) | ((lane & 15) ^ (R & 15));
            const unsigned off = (unsigned)((min(rowb + R, a.M - 1) - rowb) * (C * 4) + ch * 16);
            __builtin_amdgcn_global_load_lds(base + off, (__attribute__((address_space(3))) void*)(slot + q * 4096), 16, 0, 0);
          }
        }
        return;
      }
    };
    auto wait_vm = [&](int allow) {
      switch (allow) {
#define MDT_VMW(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
        MDT_VMW(0) MDT_VMW(1) MDT_VMW(2) MDT_VMW(3) MDT_VMW(4) MDT_VMW(5) MDT_VMW(6) MDT_VMW(7) MDT_VMW(8) MDT_VMW(9)
#undef MDT_VMW
        default: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
      }
    };
    // The stream is described by SEGMENTS (a.nheads of them): a weight segment stands for `aux` consecutive sub-tiles of the weight
    // stream (which is stored in consumption order), every other descriptor for one tile.  Two cursors walk the list: tile k + 1
    // (what may stay in flight at the wait of turn k) and tile k + 2 (what is issued behind B(k)).  While both are inside weight
    // segments a turn is `s_waitcnt vmcnt(8); s_barrier; eight pieces` and some scalar arithmetic -- no descriptor load, no decode:
    // k_rconv.hip's turn.  (First version: one descriptor per tile, loaded and decoded every turn, the wait dispatched by a switch on
    // the piece count: 1100 cycles per sub-tile in the in-kernel stamps of tools/res256_bench.py, of which 548 are the LDS-DMA issue
    // itself -- the CU's 64 B/clk vector-memory path; 890 with the switch gone.)
    const int NSEG = a.nheads;
    struct Cur { int seg, rem; unsigned kind, aux; };
    auto load_seg = [&](Cur& c) {
      if (c.seg < NSEG) {
        const unsigned d = tiles[c.seg];
        c.kind = d & 3u;
        c.aux = d >> 2;
        c.rem = c.kind == D_W ? max((int)c.aux, 1) : 1;   // a weight segment has aux >= 1 (mdt_hip.h); a 0 would never count down
      } else {
        c.kind = D_X;                                  // past the end: nothing to issue, nothing in flight
        c.aux = 0;
        c.rem = 1 << 30;
      }
    };
    auto advance = [&](Cur& c) {
      if (--c.rem == 0) {
        ++c.seg;
        load_seg(c);
      }
    };
    unsigned wnext = NSPLIT == 2 ? (unsigned)(hh * a.nff) : 0u;   // next sub-tile of the weight stream to issue (a.nff = sub-tiles of one half)
    auto issue_cur = [&](int tau, const Cur& c) {      // the tile under the issue cursor -> slot of tile tau
      if (c.kind == D_W) {
        issue_w(smem + MDT_SLOT_IDX(tau) * SLOT + iw * 1024, wsrc + (int64_t)wnext * SLOT);
        ++wnext;
      } else {
        issue_tile(tau, c.kind | (c.aux << 2));
      }
    };
    Cur ci = {0, 0, 0u, 0u}, cw;
    load_seg(ci);
    __builtin_amdgcn_s_barrier();   // P: the compute waves' row loads are queued ahead of the stream
    issue_cur(0, ci);
    advance(ci);
    cw = ci;                                           // tile 1
    if (NT > 1) {
      issue_cur(1, ci);
      advance(ci);                                     // tile 2
    }
    for (int k = 0; k < NT; ++k) {
#ifdef MDT_STAMPS   // loader wave 4 of workgroup 0: (turn start, tile landed, barrier passed, pieces issued) for tiles 24..47
      unsigned long long ls0 = 0, ls1 = 0, ls2 = 0, ls3 = 0;
      const bool lst = a.dbgbuf && blockIdx.x == 0 && iw == 0 && k >= 24 && k < 48;
      if (lst) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ls0)::"memory");
#endif
      if (cw.kind == D_W && ci.kind == D_W && cw.rem > 1 && ci.rem > 1 && k + 2 < NT) {
        // ---- the common turn: weight sub-tiles on both cursors ----
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                 // tile k landed; tile k + 1 (8 pieces) may be in flight
#ifdef MDT_STAMPS
        if (lst) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ls1)::"memory");
#endif
        __builtin_amdgcn_s_barrier();                                    // B(k)
#ifdef MDT_STAMPS
        if (lst) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ls2)::"memory");
#endif
        issue_w(smem + MDT_SLOT_IDX(k + 2) * SLOT + iw * 1024, wsrc + (int64_t)wnext * SLOT);
        ++wnext;
        --cw.rem;
        --ci.rem;
      } else {
        // ---- a segment boundary, a scratch / vector / skip tile, or the end of the stream ----
        if (k + 1 < NT && (cw.kind == D_W || cw.kind == D_SKIP)) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (k + 1 < NT && cw.kind == D_XV) wait_vm(pieces_of(D_XV));
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef MDT_STAMPS
        if (lst) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ls1)::"memory");
#endif
        __builtin_amdgcn_s_barrier();                                    // B(k)
#ifdef MDT_STAMPS
        if (lst) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ls2)::"memory");
#endif
        if (k + 2 < NT) issue_cur(k + 2, ci);
        advance(cw);
        advance(ci);
      }
#ifdef MDT_STAMPS
      if (lst) {
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ls3)::"memory");
        unsigned long long* lo_ = reinterpret_cast<unsigned long long*>(const_cast<float*>(a.dbgbuf)) + 128 + 4 * (k - 24);
        if (lane == 0) { lo_[0] = ls0; lo_[1] = ls1; lo_[2] = ls2; lo_[3] = ls3; }
      }
#endif
    }
    prefetch_next_weights(a.pf_ptr, a.pf_lines, iw * 64 + lane);
    return;
  }

  // ================= compute waves =================
#ifdef MDT_STAMPS   // tuning build: wave 0 of workgroup 0 records the shader clock at phase boundaries of block 0 (tools/res256_bench.py)
  unsigned long long* stamps = reinterpret_cast<unsigned long long*>(const_cast<float*>(a.dbgbuf));
  int nstamp = 0;
  bool stamp_on = true;
#define MDT_STAMP()                                                                   \
  do {                                                                                \
    if (stamps && stamp_on && blockIdx.x == 0 && wave == 0 && nstamp < 120) {         \
      unsigned long long t_;                                                          \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");      \
      if (lane == 0) stamps[nstamp] = (t_ & 0xffffffffffffull) | ((unsigned long long)__LINE__ << 48); \
      ++nstamp;                                                                       \
    }                                                                                 \
  } while (0)
#else
#define MDT_STAMP() do {} while (0)
#endif
  MDT_STAMP();
  const int i = lane & 15, g = lane >> 4;
  const int rt = wave >> 1, fh = wave & 1;
  const int m = rowb + rt * 16 + i;
  const bool mvalid = m < a.M;
  const int mc = mvalid ? m : a.M - 1;
  const int ch0 = 128 * fh + 4 * g;                   // channel of acc[c][q][r] = ch0 + 32 c + 16 q + r

  // the residual stream, in accumulator layout
  f32x4 acc[NCH][2];
  {
    const float* xp = a.x + (int64_t)mc * C + ch0;
    float4 xr[NCH][2];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int q = 0; q < 2; ++q) xr[c][q] = *reinterpret_cast<const float4*>(xp + 32 * c + 16 * q);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                    // P
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int q = 0; q < 2; ++q) acc[c][q] = f32x4{xr[c][q].x, xr[c][q].y, xr[c][q].z, xr[c][q].w};
  }

  // fragment addressing inside a sub-tile (k_tf256.hip), this wave's feature half folded in
  const int aP0 = F32 ? lane * 16 + fh * 16384 : fh * (2 * 16 * 4 * CS) + i * (4 * CS) + ((g ^ i) & 15) * 16;
  auto aP = [&](int st) -> int { return F32 ? aP0 : aP0 ^ (64 * st); };

  bf16x8 frh[3][2], frl[3][2];
  auto frag_read = [&](unsigned base, auto uc, int set, auto jc) __attribute__((always_inline)) {
    constexpr int u = decltype(uc)::value, j = decltype(jc)::value;
    constexpr int q = j >> 1, lo = j & 1;
    constexpr int off = F32 ? (q * 8192 + u * 2048 + lo * 1024) : (q * 16 * 4 * CS + lo * (2 * CS));
    if constexpr (F32) lds_read16_off<off>(frh[set][q], base);            // (exact fp32: one pair per set, see phase())
    else lds_read16_off<off>(lo ? frl[set][q] : frh[set][q], base);
  };
  using J0 = std::integral_constant<int, 0>;
  using J1 = std::integral_constant<int, 1>;
  using J2 = std::integral_constant<int, 2>;
  using J3 = std::integral_constant<int, 3>;
  unsigned pb2 = 0, pb3 = 0;
  int tau = 0;
  auto slot_of = [&](int t) -> unsigned char* { return smem + MDT_SLOT_IDX(t) * SLOT; };
  auto prefetch2 = [&](const unsigned char* slot) __attribute__((always_inline)) {     // units 0 and 1 of a convolution's first sub-tile
    const unsigned l = lds_addr(slot);
    if constexpr (F32) {
      const unsigned b = l + aP0;
      frag_read(b, J0{}, 0, J0{}); frag_read(b, J0{}, 0, J2{});
      return;
    }
    const unsigned b0 = l + aP(0), b1 = l + aP(1);
    pb2 = l + aP(2);
    pb3 = l + aP(3);
    frag_read(b0, J0{}, 0, J0{}); frag_read(b0, J0{}, 0, J1{}); frag_read(b0, J0{}, 0, J2{}); frag_read(b0, J0{}, 0, J3{});
    frag_read(b1, J1{}, 1, J0{}); frag_read(b1, J1{}, 1, J1{}); frag_read(b1, J1{}, 1, J2{}); frag_read(b1, J1{}, 1, J3{});
  };

  // One MFMA phase over a sub-tile [64 features][128 k] (k_tf256.hip, transposed projection): 4 units of 4 fragment reads + 6 MFMAs;
  // acc2 = this wave's two feature tiles of the chunk, bh / bl = the 4 k-steps of the K half
  auto phase = [&](auto offc, bool has_next, f32x4* acc2, const bf16x8* bh, const bf16x8* bl) __attribute__((always_inline)) {
    constexpr int OFF = decltype(offc)::value;
    if constexpr (F32) {
      const unsigned lc = lds_addr(slot_of(tau)) + aP0;
      const unsigned lnx = lds_addr(slot_of(tau + 1)) + aP0;
      auto half32 = [&](auto vc) __attribute__((always_inline)) {
        constexpr int v = decltype(vc)::value, u = v >> 1, hl = v & 1;
        if (v == 2 * NU - 1 && has_next) {
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_barrier();                // B(tau + 1)
          __builtin_amdgcn_sched_barrier(0);
        }
        lgkm_wait<0>();
        constexpr int s0 = v & 1, s1 = (v + 1) & 1;
        constexpr bool in_phase = v + 1 < 2 * NU;
        const bool pre = in_phase || has_next;
        auto rd = [&](auto qc) __attribute__((always_inline)) {
          if (!pre) return;
          constexpr int q = decltype(qc)::value;
          __builtin_amdgcn_sched_barrier(0);
          constexpr int un = (v + 1) / 2, jn = 2 * q + ((v + 1) & 1);
          if constexpr (in_phase) frag_read(lc, std::integral_constant<int, un>{}, s1, std::integral_constant<int, jn>{});
          else frag_read(lnx, std::integral_constant<int, 0>{}, s1, std::integral_constant<int, 2 * q>{});
          __builtin_amdgcn_sched_barrier(0);
        };
        const f32x4 a0 = __builtin_bit_cast(f32x4, frh[s0][0]), a1 = __builtin_bit_cast(f32x4, frh[s0][1]);
        const f32x4 xb = __builtin_bit_cast(f32x4, hl ? bl[u] : bh[u]);
        auto mm2 = [&](auto r0c) __attribute__((always_inline)) {
          constexpr int r0 = decltype(r0c)::value;
#pragma unroll
          for (int r = r0; r < r0 + 2; ++r) {
            acc2[0] = MDT_MFMA_F32(a0[r], xb[r], acc2[0], 0, 0, 0);
            acc2[1] = MDT_MFMA_F32(a1[r], xb[r], acc2[1], 0, 0, 0);
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
    auto unit = [&](auto uc) __attribute__((always_inline)) {
      constexpr int u = decltype(uc)::value;
      if (u == NU - 2 && has_next) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                  // B(tau + 1)
        __builtin_amdgcn_sched_barrier(0);
      }
      constexpr int s0 = (OFF + u) % 3, s2 = (OFF + u + 2) % 3;
      constexpr bool in_phase = u + 2 < NU;
      const bool pre = in_phase || has_next;
      const bool later = (u + 1 < NU) || has_next;
      if (later) lgkm_wait<4>(); else lgkm_wait<0>();
      auto rd = [&](auto jc) __attribute__((always_inline)) {
        if (!pre) return;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (in_phase) frag_read(u == 0 ? pb2 : pb3, std::integral_constant<int, u + 2>{}, s2, jc);
        else frag_read(bn[u + 2 - NU], std::integral_constant<int, u + 2 - NU>{}, s2, jc);
        __builtin_amdgcn_sched_barrier(0);
      };
      auto mm = [&](const bf16x8& w, const bf16x8& x, int q) __attribute__((always_inline)) {
        acc2[q] = MDT_MFMA_BF16(w, x, acc2[q], 0, 0, 0);
      };
      mm(frl[s0][0], bh[u], 0); rd(J0{});
      mm(frl[s0][1], bh[u], 1); rd(J1{});
      mm(frh[s0][0], bl[u], 0); rd(J2{});
      mm(frh[s0][1], bl[u], 1); rd(J3{});
      mm(frh[s0][0], bh[u], 0);
      if constexpr (u == 1) {                          // the next sub-tile's slot: needed from unit 2 on
        __builtin_amdgcn_sched_barrier(0);
        ln = lds_addr(slot_of(tau + 1));
        bn[0] = ln + aP(0);
        bn[1] = ln + aP(1);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (u == NU - 1) {
        __builtin_amdgcn_sched_barrier(0);
        pb2 = ln + aP(2);
        pb3 = ln + aP(3);
        __builtin_amdgcn_sched_barrier(0);
      }
      mm(frh[s0][1], bh[u], 1);
      __builtin_amdgcn_sched_barrier(0);
    };
    unit(std::integral_constant<int, 0>{}); unit(std::integral_constant<int, 1>{});
    unit(std::integral_constant<int, 2>{}); unit(std::integral_constant<int, 3>{});
    ++tau;
  };

  // neighbours inside the sample: row i - 1 exists unless i starts a sample, row i + 1 unless i ends one (T a power of two)
  const int it = i & (a.T - 1);
  const bool has_prev = it != 0, has_next_row = it != a.T - 1;
  const float t1 = a.T > 1 ? 1.f : 0.f, t2 = a.T > 2 ? 1.f : 0.f, t4 = a.T > 4 ? 1.f : 0.f, t8 = a.T > 8 ? 1.f : 0.f;
  auto dpp_fma = [](float v, float f, auto ctrl) {
    const int mm_ = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, true);
    return __builtin_fmaf(__builtin_bit_cast(float, mm_), f, v);
  };
  // sums over (the 4 lane groups g) x (the sample's token lanes) of one value per chunk; pair64: groups of 64 channels = two chunks
  auto group_reduce = [&](float (&s)[NCH], bool pair64) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) s[c] = xg16_add(s[c]);
#pragma unroll
    for (int c = 0; c < NCH; ++c) s[c] = xg32_add(s[c]);
#pragma unroll
    for (int c = 0; c < NCH; ++c) s[c] = dpp_fma(s[c], t1, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
#pragma unroll
    for (int c = 0; c < NCH; ++c) s[c] = dpp_fma(s[c], t2, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
#pragma unroll
    for (int c = 0; c < NCH; ++c) s[c] = dpp_fma(s[c], t4, std::integral_constant<int, 0x141>{});   // row_half_mirror
#pragma unroll
    for (int c = 0; c < NCH; ++c) s[c] = dpp_fma(s[c], t8, std::integral_constant<int, 0x140>{});   // row_mirror
    if (pair64) {
#pragma unroll
      for (int c = 0; c < NCH; c += 2) {
        const float t = s[c] + s[c + 1];
        s[c] = t;
        s[c + 1] = t;
      }
    }
  };
  // GroupNorm statistics (two-pass) of src in accumulator layout: groups of 32 (one chunk) or 64 (two chunks) channels x T tokens
  auto gn_stats = [&](const f32x4 (&src)[NCH][2], bool pair64, float (&mu)[NCH], float (&rs)[NCH], float in_scale = 1.0f) __attribute__((always_inline)) {
    const float inv_n = 1.0f / (float)(a.T * (pair64 ? 64 : 32));
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      mu[c] = (((src[c][0][0] + src[c][0][1]) + (src[c][0][2] + src[c][0][3])) + ((src[c][1][0] + src[c][1][1]) + (src[c][1][2] + src[c][1][3]))) * in_scale;
    group_reduce(mu, pair64);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      mu[c] *= inv_n;
      float ss = 0.f;
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = src[c][q][r] * in_scale - mu[c];
          ss += d * d;
        }
      rs[c] = ss;
    }
    group_reduce(rs, pair64);
#pragma unroll
    for (int c = 0; c < NCH; ++c) rs[c] = __builtin_amdgcn_rsqf(rs[c] * inv_n + a.eps_res);
    MDT_STAMP();                                     // group statistics
  };

  bf16x8 xh[NST], xl[NST];
  // Operands of the next convolution from `src` (accumulator layout), through the scratch tile `tau` of the ring: this wave
  // builds its k-steps 4 fh + c -- raw (gam == nullptr), or silu(GroupNorm(src) [* (scale + 1) + shift]) -- the pair exchanges them.
  auto make_operands = [&](const f32x4 (&src)[NCH][2], const float (&mu)[NCH], const float (&rs)[NCH], const float* gam, const float* bet,
                           const float* film, float in_scale) __attribute__((always_inline)) {
    unsigned char* ex = slot_of(tau);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float u[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) u[r] = src[c][q][r] * in_scale;
        if (gam) {
          const float4 gv = *reinterpret_cast<const float4*>(gam + ch0 + 32 * c + 16 * q);
          const float4 bv = *reinterpret_cast<const float4*>(bet + ch0 + 32 * c + 16 * q);
          const float g4[4] = {gv.x, gv.y, gv.z, gv.w}, b4[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float sc = rs[c] * g4[r];
            u[r] = u[r] * sc + (b4[r] - sc * mu[c]);
          }
          if (film) {
            const float4 fv = *reinterpret_cast<const float4*>(film + ch0 + 32 * c + 16 * q);
            const float4 hv = *reinterpret_cast<const float4*>(film + C + ch0 + 32 * c + 16 * q);
            const float f4[4] = {fv.x, fv.y, fv.z, fv.w}, h4[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) u[r] = u[r] * f4[r] + (u[r] + h4[r]);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) u[r] = u[r] * __builtin_amdgcn_rcpf(1.0f + __expf(-u[r]));
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[4 * q + r] = mvalid ? u[r] : 0.f;
      }
      bf16x8 h, l;
      split8_rs<F32>(v, h, l);
      *reinterpret_cast<bf16x8*>(ex + (((wave * 4 + c) * 2 + 0) * 64 + lane) * 16) = h;
      *reinterpret_cast<bf16x8*>(ex + (((wave * 4 + c) * 2 + 1) * 64 + lane) * 16) = l;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                    // B(scratch tile): the partner's k-steps are in LDS
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      const int sw = (2 * rt + st / 4) * 4 + st % 4;
      xh[st] = *reinterpret_cast<const bf16x8*>(ex + ((sw * 2 + 0) * 64 + lane) * 16);
      xl[st] = *reinterpret_cast<const bf16x8*>(ex + ((sw * 2 + 1) * 64 + lane) * 16);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // read before the slot can be refilled (two barriers later)
    ++tau;
    MDT_STAMP();                                     // operands exchanged
  };

  // A convolution of NTAPS taps on the operands xh / xl into dst (tile order: tap, K half, chunk).  Starts the fragment pipeline
  // afresh (the operands were not there while the previous convolution's last sub-tiles streamed) and drains it at the end.
  // NSPLIT = 2: half hh streams chunks 2 hh, 2 hh + 1 only; they are computed in a local pair of chunk accumulators (register
  // arrays cannot be indexed by hh) that starts from and returns to dst by selects on the wave-uniform hh.
  auto sel4 = [](bool c_, const f32x4& x, const f32x4& y) __attribute__((always_inline)) {
    return f32x4{c_ ? x[0] : y[0], c_ ? x[1] : y[1], c_ ? x[2] : y[2], c_ ? x[3] : y[3]};
  };
  auto conv = [&](auto ntaps, f32x4 (&dst)[NCH][2]) __attribute__((always_inline)) {
    constexpr int NTAPS = decltype(ntaps)::value;
    constexpr int NCL = NCH / NSPLIT;                  // chunks this workgroup streams per (tap, K half)
    f32x4 dl[NCL][2];
#pragma unroll
    for (int k = 0; k < NCL; ++k)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if constexpr (NSPLIT == 2) dl[k][q] = sel4(hh != 0, dst[NCL + k][q], dst[k][q]);
        else dl[k][q] = dst[k][q];
      }
    MDT_STAMP();                                     // operands ready
    __builtin_amdgcn_s_barrier();                    // B(first sub-tile)
    prefetch2(slot_of(tau));
    MDT_STAMP();                                     // first sub-tile there
    auto tap_kh = [&](auto pc) __attribute__((always_inline)) {
      constexpr int P0 = decltype(pc)::value;          // first phase of this (tap, K half): NCL phases follow
      constexpr int tap = P0 / (2 * NCL), kh = (P0 / NCL) & 1;
      bf16x8 oph[4], opl[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int st = 4 * kh + k;
        if (NTAPS == 1 || tap == 1) { oph[k] = xh[st]; opl[k] = xl[st]; }
        else if (tap == 0) { oph[k] = row_shift_rs<true>(xh[st], has_prev); opl[k] = row_shift_rs<true>(xl[st], has_prev); }
        else { oph[k] = row_shift_rs<false>(xh[st], has_next_row); opl[k] = row_shift_rs<false>(xl[st], has_next_row); }
      }
      constexpr int LAST = NTAPS * 2 * NCL - 1;
      phase(std::integral_constant<int, (P0 + 0) % 3>{}, P0 + 0 < LAST, dl[0], oph, opl);
      phase(std::integral_constant<int, (P0 + 1) % 3>{}, P0 + 1 < LAST, dl[1], oph, opl);
      if constexpr (NCL == 4) {
        phase(std::integral_constant<int, (P0 + 2) % 3>{}, P0 + 2 < LAST, dl[2], oph, opl);
        phase(std::integral_constant<int, (P0 + 3) % 3>{}, P0 + 3 < LAST, dl[3], oph, opl);
      }
      MDT_STAMP();                                     // the sub-tiles of one (tap, K half)
    };
    tap_kh(std::integral_constant<int, 0>{});
    tap_kh(std::integral_constant<int, NCL>{});
    if constexpr (NTAPS == 3) {
      tap_kh(std::integral_constant<int, 2 * NCL>{});
      tap_kh(std::integral_constant<int, 3 * NCL>{});
      tap_kh(std::integral_constant<int, 4 * NCL>{});
      tap_kh(std::integral_constant<int, 5 * NCL>{});
    }
#pragma unroll
    for (int k = 0; k < NCL; ++k)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if constexpr (NSPLIT == 2) {
          dst[k][q] = sel4(hh != 0, dst[k][q], dl[k][q]);
          dst[NCL + k][q] = sel4(hh != 0, dl[k][q], dst[NCL + k][q]);
        } else {
          dst[k][q] = dl[k][q];
        }
      }
  };

  // ---- NSPLIT = 2: the pair hands each other the chunks it computed of a COMPLETED accumulator set ----
  // k_tf256.hip's protocol PER WAVE: wave w of half hh needs exactly what wave w of half hh ^ 1 computed (same rows, same feature
  // half, the other two chunks), so every wave owns a 4 KB piece of the pair's hand-off blocks and a flag word of its own (words
  // 4 + w of the (row block, half) flag line whose word 0 is MDT_OP_TF256's): 16-byte sc1 stores, s_waitcnt vmcnt(0), the wave's flag
  // store; poll of the partner wave's flag with sc1 loads; 16-byte sc1 loads.  No workgroup barrier, no wave waits for another wave
  // of its own workgroup, no tile of the stream is spent on it (first version: the two barriers + one polling wave of k_tf256 --
  // ~6 us per hand-off with the waves' skew added up at the barriers; profiles/r6_res256_split_ab.txt).
  unsigned xround = 0;                               // this wave's flag value = hand-offs it has completed, ever
  int xn = 0;                                        // hand-offs of this launch (buffer parity)
  __amdgpu_buffer_rsrc_t xres, fres;
  const unsigned fown = (64u + 32u * (unsigned)(2 * rb_ + hh) + 4u + (unsigned)wave) * 4u;
  if constexpr (NSPLIT == 2) {
    xres = __builtin_amdgcn_make_buffer_rsrc(a.xbuf, 0, 0x7fffffff, 0x00020000);
    fres = __builtin_amdgcn_make_buffer_rsrc(a.xflags, 0, 0x7fffffff, 0x00020000);
    xround = (unsigned)__builtin_amdgcn_readfirstlane(__builtin_amdgcn_raw_buffer_load_b32(fres, fown, 0, AUX_LD));   // written by an earlier LAUNCH
  }
  auto handoff = [&](f32x4 (&dst)[NCH][2]) __attribute__((always_inline)) {
    if constexpr (NSPLIT == 2) {
      // block of (parity, row block, half): [wave][chunk k of the half][q][lane] f32x4 = 16 KB; the whole byte offset goes into
      // the instructions' VECTOR offset (a rewritten SGPR as scalar offset lost pieces: k_tf256.hip, round 3)
      const unsigned sbase = __builtin_amdgcn_readfirstlane(
          (((unsigned)(xn & 1) * (unsigned)nrb + (unsigned)rb_) * 2u + (unsigned)hh) * XBLOCK + (unsigned)wave * 4096u);
      const unsigned vlane = (unsigned)lane * 16u;
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const f32x4 v = sel4(hh != 0, dst[2 + k][q], dst[k][q]);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), xres, vlane + sbase + (unsigned)(2 * k + q) * 1024u, 0, AUX_ST);
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores are drained in front of ITS flag store
      ++xround;
      __builtin_amdgcn_raw_buffer_store_b32((int)xround, fres, fown, 0, AUX_ST);
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      for (;;) {
        asm volatile("" ::: "memory");                 // a fresh load every turn
        const unsigned got = (unsigned)__builtin_amdgcn_readfirstlane(__builtin_amdgcn_raw_buffer_load_b32(fres, fown ^ 128u, 0, AUX_LD));
        if ((int)(got - xround) >= 0) break;
        __builtin_amdgcn_s_sleep(1);
        if (__builtin_amdgcn_s_memrealtime() - t0 > POLL_TIMEOUT) {   // never hang the GPU: flag the launch and go on
          if (lane == 0) atomicOr(a.xflags, 1u);
          break;
        }
      }
      asm volatile("" ::: "memory");
      const unsigned sother = sbase ^ XBLOCK;            // the same piece of half hh ^ 1
      f32x4 o[2][2];
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          o[k][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, vlane + sother + (unsigned)(2 * k + q) * 1024u, 0, AUX_LD));
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          dst[k][q] = sel4(hh != 0, o[k][q], dst[k][q]);              // half 1 receives chunks 0, 1
          dst[2 + k][q] = sel4(hh != 0, dst[2 + k][q], o[k][q]);      // half 0 receives chunks 2, 3
        }
      ++xn;
      MDT_STAMP();                                     // partner's chunks received
    }
  };
  using N1 = std::integral_constant<int, 1>;
  using NB = std::integral_constant<int, TAPS>;

  auto set_vec = [&](f32x4 (&dst)[NCH][2], const float* p, bool add) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float4 b = *reinterpret_cast<const float4*>(p + ch0 + 32 * c + 16 * q);
        const f32x4 bb = f32x4{b.x, b.y, b.z, b.w};
        dst[c][q] = add ? dst[c][q] + bb : bb;
      }
  };
  // the skip rows of this block: tile `tau` of the ring -> accumulator layout (unscaled: the scale rides on make_operands)
  auto skip_rows = [&](f32x4 (&xb)[NCH][2]) __attribute__((always_inline)) {
    __builtin_amdgcn_s_barrier();                    // B(skip rows)
    const unsigned char* ss_ = slot_of(tau) + (rt * 16 + i) * (C * 4);
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int ch = 32 * fh + 8 * c + 4 * q + g;    // 16-byte chunk of the row
        const float4 xr = *reinterpret_cast<const float4*>(ss_ + (((ch & 48) | ((ch & 15) ^ i)) << 4));
        xb[c][q] = f32x4{xr.x, xr.y, xr.z, xr.w};
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads are complete before the slot can be refilled
    ++tau;
  };

  MDT_STAMP();                                       // row loads issued, past barrier P, lane constants
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  MDT_STAMP();                                       // rows arrived
  __builtin_amdgcn_s_barrier();                      // B(0): the vector tile of block 0
  ++tau;
  MDT_STAMP();

  float mu[NCH], rs[NCH];
  const float one[NCH] = {1.f, 1.f, 1.f, 1.f}, zero[NCH] = {0.f, 0.f, 0.f, 0.f};
  for (int rb = 0; rb < a.n_res; ++rb) {
#ifdef MDT_STAMPS
    stamp_on = rb == 0;
#endif
    const float* pv = reinterpret_cast<const float*>(vec_b + (rb & 1) * VPAR);
    const float* film_s = pv + VB;
    f32x4 hT[NCH][2];
    if constexpr (RES == 1) {
      // vectors: [g1 | b1 | bias1 | g2 | b2 | bias2]
      gn_stats(acc, false, mu, rs);
      make_operands(acc, mu, rs, pv, pv + C, nullptr, 1.0f);
      set_vec(hT, pv + 2 * C, false);
      conv(NB{}, hT);
      handoff(hT);
      gn_stats(hT, false, mu, rs);
      make_operands(hT, mu, rs, pv + 3 * C, pv + 4 * C, film_s, 1.0f);
      set_vec(acc, pv + 5 * C, true);                  // the stream is the block's residual
      conv(NB{}, acc);
      if (rb + 1 < a.n_res) handoff(acc);              // (the last block's output leaves by halves, below)
      if (mvalid) {                                    // every block's output is a skip of the up path (NSPLIT = 2: the chunks this half
        float* so = a.skip + (int64_t)rb * a.skip_stride + (int64_t)m * C + ch0;      // computed; BEHIND the hand-off: its drain does not wait for them)
#pragma unroll
        for (int c = 0; c < NCH; ++c)
          if (NSPLIT == 1 || (c >> 1) == hh) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
              store_nt(so + 32 * c + 16 * q, make_float4(acc[c][q][0], acc[c][q][1], acc[c][q][2], acc[c][q][3]));
          }
      }
    } else {
      // vectors: [g1 (2C) | b1 (2C) | bias1 | bias_r | g2 | b2 | bias2]; groups of block1's 2C-channel input are 64 channels wide.
      // Order chosen for the register budget (never more than TWO accumulator-sized sets next to the operands, the shifted
      // operands of a K half and the fragment sets): block1 on x first; x is dead once its raw operands are built, and the
      // residual convolution, its bias and block2 all accumulate into the SAME set.  The skip rows arrive twice.
      gn_stats(acc, true, mu, rs);
      make_operands(acc, mu, rs, pv, pv + 2 * C, nullptr, 1.0f);
      set_vec(hT, pv + 4 * C, false);
      conv(NB{}, hT);                                                          // block1 on x
      make_operands(acc, one, zero, nullptr, nullptr, nullptr, 1.0f);          // raw x: the residual convolution to_out(cat) (k = 1)
      set_vec(acc, pv + 5 * C, false);                                         // ... from here on acc is the block's OUTPUT sum
      set_vec(acc, pv + 8 * C, true);
      conv(N1{}, acc);
      {
        f32x4 xb[NCH][2];
        skip_rows(xb);
        make_operands(xb, one, zero, nullptr, nullptr, nullptr, a.skip_scale);
      }
      conv(N1{}, acc);
      {
        f32x4 xb[NCH][2];
        skip_rows(xb);
        gn_stats(xb, true, mu, rs, a.skip_scale);
        make_operands(xb, mu, rs, pv + C, pv + 3 * C, nullptr, a.skip_scale);
      }
      conv(NB{}, hT);                                                          // block1 on the skip
      handoff(hT);
      gn_stats(hT, false, mu, rs);
      make_operands(hT, mu, rs, pv + 6 * C, pv + 7 * C, film_s, 1.0f);
      conv(NB{}, acc);                                                         // block2
      if (rb + 1 < a.n_res) handoff(acc);                                      // (the last block's output leaves by halves, below)
    }
  }

  // ---- the residual stream leaves the kernel (NSPLIT = 2: every half stores the chunks it owns; kind 1 has exchanged them, both hold
  // the same values) ----
  if (mvalid) {
    float* xo = a.out + (int64_t)m * C + ch0;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      if (NSPLIT == 1 || (c >> 1) == hh) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
          store_nt(xo + 32 * c + 16 * q, make_float4(acc[c][q][0], acc[c][q][1], acc[c][q][2], acc[c][q][3]));
      }
  }
}

template <int RES, int TAPS, bool F32, int NSPLIT>
static hipError_t launch_rs2(const TFArgs& a, hipStream_t s) {
  const size_t smem = (size_t)NS * SLOT + 2 * VPAR;              // ring, vector areas
  static DevOnce attr_once;                          // per device (mdt_kernels.h)
  if (attr_once.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_res256<RES, TAPS, F32, NSPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(160 * 1024));
  const int nrb = (a.M + 31) / 32;
  if constexpr (NSPLIT == 1) {
    hipLaunchKernelGGL((k_res256<RES, TAPS, F32, NSPLIT>), dim3((unsigned)nrb), dim3(512), smem, s, a);
    return hipGetLastError();
  } else {
    // Both workgroups of a pair must be resident at the same time: never more workgroups in a launch than the device runs at once;
    // a larger batch runs as several launches over consecutive row-block ranges (k_tf256.hip::launch_tf2, same reasoning)
    // (one 512-thread workgroup per compute unit for either kernel -- the LDS ring --: k_tf256's capacity, asked once per device;
    //  the test override, mdt_set_tuning("pair_capacity"), is looked at on every launch)
    static int cap_of[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    int cap = g_pair_capacity_override;
    if (cap <= 0) {
      if (cap_of[dev] == 0) {
        const int c_ = tf256_pair_capacity();
        cap_of[dev] = c_ > 0 ? c_ : -1;
      }
      cap = cap_of[dev];
    }
    if (cap <= 0) return hipErrorLaunchOutOfResources;
    const int S = a.pair_stride;
    const int groups_fit = cap / (2 * S);
    if (groups_fit < 1) return hipErrorLaunchOutOfResources;
    const int ngroups = (nrb + S - 1) / S;
    for (int g0 = 0; g0 < ngroups; g0 += groups_fit) {
      TFArgs b = a;
      b.rb_base = g0 * S;
      const int ng = ngroups - g0 < groups_fit ? ngroups - g0 : groups_fit;
      hipLaunchKernelGGL((k_res256<RES, TAPS, F32, NSPLIT>), dim3(2u * (unsigned)S * (unsigned)ng), dim3(512), smem, s, b);
      const hipError_t e = hipGetLastError();
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  }
}

template <int RES, int TAPS, bool F32>
static hipError_t launch_rs(const TFArgs& a, hipStream_t s) {
  return a.nsplit == 2 ? launch_rs2<RES, TAPS, F32, 2>(a, s) : launch_rs2<RES, TAPS, F32, 1>(a, s);
}

bool res256_supported(int T, int kind, int n_res, int taps) {
  if (T <= 0 || 16 % T || (kind != 1 && kind != 2) || n_res <= 0 || n_res > 255) return false;
  return taps == 3 || (taps == 1 && T == 1);
}

// a.res_kind / n_res / T as MDT_OP_TF128's; a.npost carries the taps of the block convolutions (1 | 3); a.nvec = floats of ALL
// blocks' vectors (n_res x 6 C | 9 C), a.nfilm >= 2 C n_res; a.nsplit = 2: the pair-split form (a.nff = weight sub-tiles of ONE half,
// a.xflags / a.xbuf / a.pair_stride as MDT_OP_TF256's)
hipError_t launch_res256(const TFArgs& a, hipStream_t s) {
  if (a.M <= 0) return hipSuccess;
  const int taps = a.npost;
  if (!res256_supported(a.T, a.res_kind, a.n_res, taps) || a.NT <= 0 || a.nheads <= 0 || !a.skip || !a.film || !a.vec || !a.tiles) return hipErrorInvalidValue;
  if (a.nsplit == 2 && (!a.xbuf || !a.xflags || a.pair_stride <= 0 || a.pair_stride > 64 || a.nff <= 0)) return hipErrorInvalidValue;
  if (a.wf32) {                                       // exact fp32 products on fp32 fragment sub-tiles (MDT_F_WF32)
    if (a.res_kind == 1) return taps == 3 ? launch_rs<1, 3, true>(a, s) : launch_rs<1, 1, true>(a, s);
    return taps == 3 ? launch_rs<2, 3, true>(a, s) : launch_rs<2, 1, true>(a, s);
  }
  if (a.res_kind == 1) return taps == 3 ? launch_rs<1, 3, false>(a, s) : launch_rs<1, 1, false>(a, s);
  return taps == 3 ? launch_rs<2, 3, false>(a, s) : launch_rs<2, 1, false>(a, s);
}

}  // namespace mdt
