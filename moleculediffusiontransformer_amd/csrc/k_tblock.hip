// Fused transformer sub-blocks of TransformerBlock.forward (modules.py:456-461) on gfx950:
//
//   MODE_SELF   x += to_out( softmax(q k^T / 8) v ),  q = to_q(LN(x)),  k,v = to_kv(LN_ctx(x))    (:401-410, :350-364)
//   MODE_CROSS  x += to_out( softmax(q K^T / 8) V ),  q = to_q(LN(x)),  K,V = hoisted context projections
//   MODE_FF     x += W2 gelu(W1 x + b1) + b2                                                     (:314-320)
//
// Why: in the layer-by-layer program the q / kv / attention-output / FF-hidden tensors (up to 67 MB each at
// B = 1024) are written to and re-read from HBM between ~10 launches per block, and that traffic -- not
// the matrix cores -- bounds the step.  Here they never leave the register file:
//
//   * a workgroup owns 64 consecutive token rows, each of its 4 waves 16 of them (whole samples, T | 16);
//   * projections are computed TRANSPOSED (A = weight rows, B = the wave's 16 normalised rows held in
//     registers as bf16 hi/lo), so the 16x16 accumulator holds [feature 16t+4g+r][token l&15]:
//       - q^T and k^T accumulators are, register for register, the B and A operands of the fp32
//         v_mfma_f32_16x16x4 that forms S^T = K Q^T (k-slot <-> feature 16t + 4(l>>4) + r);
//       - S^T[j = 4g+r][i] puts a query's scores in one lane column: softmax = 4 registers + 2 shuffles;
//       - v is computed un-transposed, which makes its accumulator the A operand of O^T = V^T P^T with the
//         probabilities taken from the lane's own registers;
//       - O^T (or the GELU'd FF hidden chunk) re-packed to bf16 hi/lo in registers is the B operand of the
//         output projection, whose columns the host pre-permutes to the accumulator's feature order.
//   * LayerNorm gains/biases are folded into the projection weights/biases on the host; the kernel only
//     normalises.  Products use the split-bf16 scheme of k_gemm_bf16x3.hip (hi*hi + hi*lo + lo*hi, fp32
//     accumulation); the attention core is exact fp32.
//   * the only shared resource is the weight stream: 256*C-byte tiles ([64][C] projection tiles, [C][64]
//     output tiles, bf16 hi plane + lo plane) are pulled by LDS-DMA (global_load_lds_dwordx4, no VGPRs) into
//     a ring of LDS slots, up to three tiles (96 KB) in flight per CU, XOR-swizzled through the per-lane
//     source address so that the 16-row MFMA fragment reads (ds_read_b128) are bank-conflict free.
//     Protocol per tile: counted s_waitcnt vmcnt (own DMAs) -> s_barrier (everyone's) -> refill the slot
//     released by the previous tile -> consume.
#include <cstdlib>

#include "mdt_kernels.h"

namespace mdt {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { TB_SELF = 0, TB_CROSS = 1, TB_FF = 2 };

// Exact-erf GELU with a branch-free erf (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7, i.e. fp32 rounding
// level): the library erff costs ~200 cycles per value here (divergent polynomial branches) and dominated
// the feed-forward chunk (3270 of 7700 cycles, measured with s_memtime stamps).
__device__ __forceinline__ float gelu_tb(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);   // 1 ulp; the A&S fit itself is 1.5e-7
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erfa = 1.0f - poly * __expf(-z * z);         // erf(|x|/sqrt2)
  return 0.5f * x * (1.0f + copysignf(erfa, x));
}

__device__ __forceinline__ void split8(const float v[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 h = (__bf16)v[e];
    hi[e] = h;
    lo[e] = (__bf16)(v[e] - (float)h);
  }
}

#define MDT_MFMA_BF16 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define MDT_MFMA_F32 __builtin_amdgcn_mfma_f32_16x16x4f32

template <int MODE, int C>
__global__ __launch_bounds__(256) void k_tblock(TBlockArgs a, int dbg) {
  constexpr int SLOT = 256 * C;                    // bytes per weight tile (hi plane + lo plane)
  constexpr int NS = (C == 128) ? 4 : 2;           // ring slots
  constexpr int IPT = C / 16;                      // DMA instructions per tile per wave
  constexpr int TPC = (MODE == TB_SELF) ? 4 : 2;   // tiles per chunk (head / hidden chunk)
  constexpr int NST = C / 32;                      // k-steps of a projection
  constexpr int NCT = C / 16;                      // 16-row tiles of the output projection
  constexpr int KTM = (MODE == TB_CROSS) ? 4 : 1;  // key tiles held in registers

  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  float* bias_s = reinterpret_cast<float*>(smem + NS * SLOT);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int row0 = blockIdx.x * 64 + wave * 16;
  const int m = row0 + i;
  const bool mvalid = m < a.M;
  const int mc = mvalid ? m : a.M - 1;
  const int NT = a.nchunk * TPC;
  const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.w);

  // ---- biases -> LDS (ordinary loads, before any DMA is in flight) ----
  for (int t = tid; t < a.nbias; t += 256) bias_s[t] = a.bias[t];

  // ---- this wave's 16 rows in MFMA operand layout: lane (i, g) holds x[i][32 st + 8 g + e] ----
  bf16x8 xh[NST], xl[NST];
  {
    float xr[NST][8];
    const float* xp = a.x + (int64_t)mc * a.ldx + 8 * g;
    float s = 0.f;
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      const float4 u = *reinterpret_cast<const float4*>(xp + 32 * st);
      const float4 w = *reinterpret_cast<const float4*>(xp + 32 * st + 4);
      xr[st][0] = u.x; xr[st][1] = u.y; xr[st][2] = u.z; xr[st][3] = u.w;
      xr[st][4] = w.x; xr[st][5] = w.y; xr[st][6] = w.z; xr[st][7] = w.w;
#pragma unroll
      for (int e = 0; e < 8; ++e) s += xr[st][e];
    }
    float mean = 0.f, rstd = 1.f;
    if constexpr (MODE != TB_FF) {       // nn.LayerNorm statistics (two-pass; gain/bias folded into the weights)
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      mean = s / (float)C;
      float ss = 0.f;
#pragma unroll
      for (int st = 0; st < NST; ++st)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = xr[st][e] - mean;
          ss += d * d;
        }
      ss += __shfl_xor(ss, 16, 64);
      ss += __shfl_xor(ss, 32, 64);
      rstd = 1.0f / sqrtf(ss / (float)C + a.eps);
    }
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = mvalid ? (xr[st][e] - mean) * rstd : 0.f;
      split8(v, xh[st], xl[st]);
    }
  }
  __syncthreads();   // bias_s visible; all ordinary loads retired before the DMA pipeline starts

  // ---- weight ring ----
  // DMA source addressing.  Instruction `inst` (= wave + 4 q) fills slot bytes [inst*1024, +1024); lane l
  // supplies bytes inst*1024 + 16 l.  Un-swizzling that position to a source address factors into a
  // lane-only part (computed once) and a wave-uniform part (scalar ALU), coupled by ONE xor:
  //   projection tile, row pitch 4C: row = U + lp (U = 2 inst | inst, lp = lane>>5 | 0),
  //       chunk = (pc & ~15) | ((pc & 15) ^ (row & 15)),  and (row & 15) = (U & 15) | lp  since U is even
  //       whenever lp can be 1  ->  chunk_lo = ((pc & 15) ^ lp) ^ (U & 15)
  //   output tile, row pitch 128: row = 8 inst % C + (l >> 3), chunk = (l & 7) ^ ((row >> 1) & 7)
  //       = ((l & 7) ^ (l >> 4)) ^ 4 (inst & 1)
  const int lpP = (C == 128) ? (lane >> 5) : 0;
  const int xP = (lane & 15) ^ lpP;
  const int baseP = ((C == 128) ? ((lane >> 4) & 1) : (lane >> 5)) * (128 * C) + lpP * (2 * C) +
                    ((C == 128) ? 0 : (lane & 16) * 16);
  const int xO = (lane & 7) ^ (lane >> 4);
  const int baseO = (lane >> 3) * 128;
  auto issue_tile = [&](int tau) {
    if (dbg & 1) return;
    const unsigned char* tile = wsrc + (int64_t)tau * SLOT;
    unsigned char* slot = smem + (tau % NS) * SLOT;
    const bool otile = (tau % TPC) == TPC - 1;
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
      const int inst = wave + 4 * q;                  // wave-uniform
      if (!otile) {
        const int U = (C == 128) ? 2 * inst : inst;
        const int v = ((xP ^ (U & 15)) << 4) + baseP;
        __builtin_amdgcn_global_load_lds(tile + (int64_t)U * (2 * C) + v,
                                         (__attribute__((address_space(3))) void*)(slot + inst * 1024), 16, 0, 0);
      } else {
        const int u = ((inst * 8) / C) * (128 * C) + ((inst * 8) % C) * 128;
        const int v = ((xO ^ (4 * (inst & 1))) << 4) + baseO;
        __builtin_amdgcn_global_load_lds(tile + u + v, (__attribute__((address_space(3))) void*)(slot + inst * 1024),
                                         16, 0, 0);
      }
    }
  };
  // Own DMAs of the tile about to be consumed have landed once at most `allow` newer vector-memory operations
  // of this wave are still outstanding (vmcnt retires in issue order): the DMAs of the tiles issued after it
  // plus, in CROSS mode, the K/V loads requested just before.
  auto wait_vm = [&](int allow) {
    if (allow >= 56) asm volatile("s_waitcnt vmcnt(56)" ::: "memory");
    else if (allow >= 36) asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
    else if (allow >= 28) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
    else if (allow >= 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if (allow >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (allow >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  int tau = 0;
  auto acquire = [&](int extra_vm = 0) -> const unsigned char* {   // slot holding tile `tau`; then advances
    const int after = min(NS - 2, NT - 1 - tau);
    if (!(dbg & 1)) wait_vm(after * IPT + extra_vm);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // every wave's DMAs landed; slot of tile tau-1 is free
    if (tau + NS - 1 < NT) issue_tile(tau + NS - 1);
    const unsigned char* slot = smem + (tau % NS) * SLOT;
    ++tau;
    return slot;
  };
  // Fragment addressing.  The swizzled LDS address of a fragment splits into a lane-dependent part that only
  // depends on the k-step (precomputed once, NST + 2 registers) and a compile-time part (16-row tile, plane)
  // that lands in the ds_read_b128 immediate; per fragment there is then no address arithmetic left, which
  // matters because this kernel runs one wave per SIMD and is VALU-issue bound otherwise.
  //   projection tile: row = 16 t + i, chunk = 4 st + g :  row*4C + plane*2C + ((chunk&~15) | ((chunk&15)^(row&15)))*16
  //   output tile:     row = 16 t + i, chunk = 4 sp + g :  plane*C*128 + row*128 + (chunk ^ ((row>>1)&7))*16
  int aP[NST], aO[2];
#pragma unroll
  for (int st = 0; st < NST; ++st) {
    const int lc = 4 * st + g;
    aP[st] = i * (4 * C) + ((lc & ~15) | ((lc & 15) ^ i)) * 16;
  }
#pragma unroll
  for (int sp = 0; sp < 2; ++sp) aO[sp] = i * 128 + ((4 * sp + g) ^ ((i >> 1) & 7)) * 16;
  // Fragment reads are issued as inline asm: hipcc's waitcnt pass only ever emits lgkmcnt(0) in this kernel
  // (it then waits for the prefetch just issued and exposes one LDS latency per 12 MFMAs: 27 instead of 16.7
  // cycles per MFMA).  Opaque asm reads + hand-counted s_waitcnt lgkmcnt(N) keep two units in flight.
  auto lds_read = [&](bf16x8& dst, const unsigned char* p) {
    // generic -> LDS address-space cast yields the 32-bit LDS byte address ds_read expects
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)p;
    asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory");
  };
  auto lgkm_wait = [&](int pending) {                  // reads still allowed in flight
    if (pending >= 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    else if (pending >= 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto fragP = [&](const unsigned char* slot, int tile16, int st, int plane) -> bf16x8 {
    return *reinterpret_cast<const bf16x8*>(slot + aP[st] + (tile16 * 16 * 4 * C + plane * 2 * C));
  };
  auto fragO = [&](const unsigned char* slot, int tile16, int sp, int plane) -> bf16x8 {
    return *reinterpret_cast<const bf16x8*>(slot + aO[sp] + (tile16 * 16 * 128 + plane * C * 128));
  };
  // The three MFMA loops below are software-pipelined by hand: with one wave per SIMD nothing else hides the
  // LDS latency, and left alone hipcc emits ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma per fragment.  Fragments
  // are fetched in batches of 4 k-steps (8 x ds_read_b128) into one of two register sets; the batch for unit
  // u+1 is issued before the 12 MFMAs of unit u, and sched_barrier keeps the two groups apart.
  // Unit = one k-step of TWO 16-feature tiles: 4 fragment reads + 6 MFMAs on two independent accumulators.
  // Fragments are prefetched two units ahead into three register sets, so at most 12 ds_reads are outstanding:
  // lgkmcnt is a 4-bit counter, and with 16 outstanding (8 + 8) hipcc can only emit lgkmcnt(0), which waits
  // for the prefetch it has just issued (measured: 29 instead of 16.7 cycles per MFMA).
  // transposed projection: out[ft][r] = (W x^T)[feature 16 ft + 4 g + r][token i]
  auto proj_T = [&](const unsigned char* slot, f32x4 out[4], const float* bias) {
    constexpr int NU = 2 * NST;                        // units: u = 2 st + half
    bf16x8 fh[3][2], fl[3][2];
    auto load = [&](int u, int set) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        lds_read(fh[set][q], slot + aP[u >> 1] + ((2 * (u & 1) + q) * 16 * 4 * C));
        lds_read(fl[set][q], slot + aP[u >> 1] + ((2 * (u & 1) + q) * 16 * 4 * C + 2 * C));
      }
    };
    load(0, 0);
    load(1, 1);
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) out[ft] = *reinterpret_cast<const f32x4*>(bias + 16 * ft + 4 * g);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      if (u + 2 < NU) load(u + 2, (u + 2) % 3);
      lgkm_wait(4 * min(2, NU - 1 - u));
      const int st = u >> 1, f0 = 2 * (u & 1);
#pragma unroll
      for (int q = 0; q < 2; ++q) out[f0 + q] = MDT_MFMA_BF16(fl[u % 3][q], xh[st], out[f0 + q], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 2; ++q) out[f0 + q] = MDT_MFMA_BF16(fh[u % 3][q], xl[st], out[f0 + q], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 2; ++q) out[f0 + q] = MDT_MFMA_BF16(fh[u % 3][q], xh[st], out[f0 + q], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // un-transposed projection: out[dt][r] = (x W^T)[token 4 g + r][feature 16 dt + i]
  auto proj_N = [&](const unsigned char* slot, f32x4 out[4], const float* bias) {
    constexpr int NU = 2 * NST;
    bf16x8 fh[3][2], fl[3][2];
    auto load = [&](int u, int set) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        lds_read(fh[set][q], slot + aP[u >> 1] + ((2 * (u & 1) + q) * 16 * 4 * C));
        lds_read(fl[set][q], slot + aP[u >> 1] + ((2 * (u & 1) + q) * 16 * 4 * C + 2 * C));
      }
    };
    load(0, 0);
    load(1, 1);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const float b = bias[16 * dt + i];
      out[dt] = f32x4{b, b, b, b};
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      if (u + 2 < NU) load(u + 2, (u + 2) % 3);
      lgkm_wait(4 * min(2, NU - 1 - u));
      const int st = u >> 1, d0 = 2 * (u & 1);
#pragma unroll
      for (int q = 0; q < 2; ++q) out[d0 + q] = MDT_MFMA_BF16(xl[st], fh[u % 3][q], out[d0 + q], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 2; ++q) out[d0 + q] = MDT_MFMA_BF16(xh[st], fl[u % 3][q], out[d0 + q], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 2; ++q) out[d0 + q] = MDT_MFMA_BF16(xh[st], fh[u % 3][q], out[d0 + q], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  f32x4 accT[NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) accT[ct] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int t = 0; t < NS - 1; ++t)
    if (t < NT) issue_tile(t);

  const int bo_off = (MODE == TB_SELF) ? 3 * 64 * a.nchunk : 64 * a.nchunk;   // [bq | bk | bv | bo] / [b1 | b2]
  const int samp_q = i / a.T;                        // sample (within the wave's 16 rows) of query column i
  // diagnostic build only (MDT_DBG & 8): wave 0 of workgroup 0 stamps the shader clock at phase boundaries into
  // the kv pointer's buffer reinterpreted as a scratch area -- never enabled in a timed or parity run
  unsigned long long* stamps = reinterpret_cast<unsigned long long*>(const_cast<float*>(a.dbgbuf));
  int nstamp = 0;
  auto stamp = [&]() {
#ifdef MDT_STAMPS
    if ((dbg & 8) && stamps && blockIdx.x == 0 && wave == 0 && nstamp < 64) {
      unsigned long long t;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      if (lane == 0) stamps[nstamp] = t;
      ++nstamp;
    }
#endif
  };

  for (int h = 0; h < a.nchunk; ++h) {
    f32x4 oT[4];
    if (dbg & 2) {   // stream-only ablation: consume the tiles without computing
      for (int t = 0; t < TPC; ++t) (void)acquire();
      continue;
    }
    stamp();
    if constexpr (MODE == TB_FF) {
      const unsigned char* s1 = acquire();
      stamp();
      proj_T(s1, oT, bias_s + 64 * h);
      stamp();
#pragma unroll
      for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int r = 0; r < 4; ++r) oT[ft][r] = gelu_tb(oT[ft][r]);
      stamp();
    } else {
      f32x4 qT[4];
      f32x4 st[KTM];
      f32x4 vT[KTM][4];
      float mx = -INFINITY;
      if constexpr (MODE == TB_SELF) {
        f32x4 kT[4];
        const unsigned char* sq = acquire();
        stamp();
        proj_T(sq, qT, bias_s + 64 * h);
        stamp();
        const unsigned char* sk = acquire();
        stamp();
        proj_T(sk, kT, bias_s + 64 * (a.nchunk + h));
        stamp();
        const unsigned char* sv = acquire();
        stamp();
        proj_N(sv, vT[0], bias_s + 64 * (2 * a.nchunk + h));
        stamp();
        f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
          s0 = MDT_MFMA_F32(kT[ft][0], qT[ft][0], s0, 0, 0, 0);
          s1 = MDT_MFMA_F32(kT[ft][1], qT[ft][1], s1, 0, 0, 0);
          s0 = MDT_MFMA_F32(kT[ft][2], qT[ft][2], s0, 0, 0, 0);
          s1 = MDT_MFMA_F32(kT[ft][3], qT[ft][3], s1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 4 * g + r;                    // key token (within the wave's 16 rows)
          const float sv2 = (j / a.T == samp_q) ? (s0[r] + s1[r]) * a.scale : -INFINITY;
          st[0][r] = sv2;
          mx = fmaxf(mx, sv2);
        }
      } else {   // TB_CROSS: keys/values are the hoisted context projections of the wave's samples
        const int nsamp = 16 / a.T;                   // samples per wave
        const int nkeys = nsamp * a.Tk;
        const int sample0 = row0 / a.T;
        // K and V fragments of this head are requested FIRST, so that their latency is covered by the q
        // projection (left after it, they cost ~5800 exposed cycles per head).
        float4 kk[KTM][4];
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt) {
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) vT[kt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};   // p = 0 there, but 0 * garbage may be NaN
#pragma unroll
          for (int ft = 0; ft < 4; ++ft) kk[kt][ft] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (kt * 16 < nkeys) {
            // A operand of S^T: lane (key jl = 16 kt + i, quarter g) holds K[key][64 h + 16 ft + 4 g + s]
            const int jl = kt * 16 + i;
            const int js = min(jl / a.Tk, nsamp - 1), jk = jl % a.Tk;
            const float* kp = a.kv + ((int64_t)min(sample0 + js, a.nsamples - 1) * a.kv_bstride + jk) * a.ldkv +
                              64 * h + 4 * g;
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) kk[kt][ft] = *reinterpret_cast<const float4*>(kp + 16 * ft);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              // A operand of O^T: lane (feature 16 dt + i, quarter g) holds V[key 4 g + r][64 h + 16 dt + i]
              const int jj = kt * 16 + 4 * g + r;
              const int vs = min(jj / a.Tk, nsamp - 1), vk = jj % a.Tk;
              const float* vp = a.kv + ((int64_t)min(sample0 + vs, a.nsamples - 1) * a.kv_bstride + vk) * a.ldkv +
                                64 * a.nheads + 64 * h + i;
#pragma unroll
              for (int dt = 0; dt < 4; ++dt) {
                const float vv = vp[16 * dt];
                vT[kt][dt][r] = jj < nkeys ? vv : 0.f;
              }
            }
          }
        }
        const int nkv = 20 * ((nkeys + 15) >> 4);      // 4 K + 16 V loads per active key tile, issued above
        const unsigned char* sq = acquire(nkv);
        proj_T(sq, qT, bias_s + 64 * h);
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt) {
          st[kt] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
          if (kt * 16 < nkeys) {
            f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) {
              s0 = MDT_MFMA_F32(kk[kt][ft].x, qT[ft][0], s0, 0, 0, 0);
              s1 = MDT_MFMA_F32(kk[kt][ft].y, qT[ft][1], s1, 0, 0, 0);
              s0 = MDT_MFMA_F32(kk[kt][ft].z, qT[ft][2], s0, 0, 0, 0);
              s1 = MDT_MFMA_F32(kk[kt][ft].w, qT[ft][3], s1, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int jj = kt * 16 + 4 * g + r;     // concatenated key index of accumulator register r
              const bool ok = jj < nkeys && (jj / a.Tk) == samp_q;
              const float sv2 = ok ? (s0[r] + s1[r]) * a.scale : -INFINITY;
              st[kt][r] = sv2;
              mx = fmaxf(mx, sv2);
            }
          }
        }
      }
      // softmax over the keys of query column i: registers r (and key tiles), then lane quarters g
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < KTM; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = expf(st[kt][r] - mx);
          st[kt][r] = e;
          sum += e;
        }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) oT[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < KTM; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = st[kt][r] / sum;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) oT[dt] = MDT_MFMA_F32(vT[kt][dt][r], p, oT[dt], 0, 0, 0);
        }
    }
    stamp();
    // ---- output projection of this chunk: accT[c][i] += sum_d Wo[c][64 h + d] * o[d][i] ----
    const unsigned char* so = acquire();
    stamp();
    bf16x8 oh[2], ol[2];
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = oT[2 * sp + (e >> 2)][e & 3];
      split8(v, oh[sp], ol[sp]);
    }
    {
      constexpr int CH = NCT / 2;                      // units of 2 row tiles per k-step
      constexpr int NU = 2 * CH;
      bf16x8 fh[3][2], fl[3][2];
      auto load = [&](int u, int set) {
        const int sp = u / CH, ct0 = 2 * (u % CH);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          lds_read(fh[set][q], so + aO[sp] + ((ct0 + q) * 16 * 128));
          lds_read(fl[set][q], so + aO[sp] + ((ct0 + q) * 16 * 128 + C * 128));
        }
      };
      load(0, 0);
      load(1, 1);
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int sp = u / CH, ct0 = 2 * (u % CH);
        if (u + 2 < NU) load(u + 2, (u + 2) % 3);
        lgkm_wait(4 * min(2, NU - 1 - u));
#pragma unroll
        for (int q = 0; q < 2; ++q) accT[ct0 + q] = MDT_MFMA_BF16(fl[u % 3][q], oh[sp], accT[ct0 + q], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 2; ++q) accT[ct0 + q] = MDT_MFMA_BF16(fh[u % 3][q], ol[sp], accT[ct0 + q], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 2; ++q) accT[ct0 + q] = MDT_MFMA_BF16(fh[u % 3][q], oh[sp], accT[ct0 + q], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    stamp();
  }

  // ---- residual + output bias: x[m][16 ct + 4 g + r] += accT[ct][r] + bo[..] ----
  if (mvalid) {
    float* xo = a.x + (int64_t)m * a.ldx + 4 * g;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const float4 xr = *reinterpret_cast<const float4*>(xo + 16 * ct);
      const float4 bo = *reinterpret_cast<const float4*>(bias_s + bo_off + 16 * ct + 4 * g);
      *reinterpret_cast<float4*>(xo + 16 * ct) =
          make_float4(accT[ct][0] + bo.x + xr.x, accT[ct][1] + bo.y + xr.y, accT[ct][2] + bo.z + xr.z,
                      accT[ct][3] + bo.w + xr.w);
    }
  }
}

template <int MODE, int C>
static hipError_t launch_tb(const TBlockArgs& a, hipStream_t s) {
  constexpr int NS = (C == 128) ? 4 : 2;
  const size_t smem = (size_t)NS * 256 * C + (size_t)((a.nbias + 3) / 4 * 4) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tblock<MODE, C>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    attr_set = true;
  }
  if (smem > 160 * 1024) return hipErrorInvalidValue;
  static const int dbg = getenv("MDT_DBG") ? atoi(getenv("MDT_DBG")) : 0;   // ablation switches (tuning aid)
  hipLaunchKernelGGL((k_tblock<MODE, C>), dim3((unsigned)((a.M + 63) / 64)), dim3(256), smem, s, a, dbg);
  return hipGetLastError();
}

hipError_t launch_tblock(const TBlockArgs& a, hipStream_t s) {
  if (a.M <= 0) return hipSuccess;
  static const bool use_lw = !(getenv("MDT_TB_LW") && atoi(getenv("MDT_TB_LW")) == 0);   // loader-wave kernels (default)
  if (use_lw && tblock_lw_supported(a)) return launch_tblock_lw(a, s);
  if (a.post || a.kv2) return hipErrorInvalidValue; // folded closing convolution / dual batch: ring kernels only
  if ((a.C != 128 && a.C != 256) || a.T <= 0 || 16 % a.T || a.nchunk <= 0) return hipErrorInvalidValue;
  if (a.mode == TB_CROSS && (a.Tk <= 0 || (16 / a.T) * a.Tk > 64)) return hipErrorInvalidValue;
#define MDT_TB_CASE(MD)                                                         \
  case MD:                                                                      \
    return a.C == 128 ? launch_tb<MD, 128>(a, s) : launch_tb<MD, 256>(a, s);
  switch (a.mode) {
    MDT_TB_CASE(TB_SELF)
    MDT_TB_CASE(TB_CROSS)
    MDT_TB_CASE(TB_FF)
    default: return hipErrorInvalidValue;
  }
#undef MDT_TB_CASE
}

}  // namespace mdt
