// Fused transformer sub-blocks for the C = 256 level (4 tokens per sample: 4096 rows at B = 1024), 32-row
// workgroups.  Why 32: at 16 rows per workgroup (k_tblock16.hip) the weight stream is re-read by 256 workgroups
// and the launch sits on the L2 -> LDS bandwidth roof (feed-forward: 512 MB per launch = 17.6 TB/s, measured);
// 64 rows leave 192 CUs idle.  32 rows halve the stream and keep 128 CUs busy with loader waves beside them.
//
//   * compute wave w = (row tile rt = w >> 1, feature half fh = w & 1): 16 rows x 32 of each chunk's 64 features;
//     the two waves of a row tile exchange the partial S^T = K Q^T (sum over features) through 4 KB of LDS and
//     sum their output-projection accumulators once, at the end;
//   * a wave's 32 features are exactly one k-step of the output projection (no zero padding as in k_tblock16);
//   * weights arrive as 32 KB sub-tiles in the C = 128 tile format (host: compiler.py, variant 2), so the ring,
//     the swizzle and the fragment addressing are those of k_tblock_lw.hip: 4 loader waves, 4 slots, issue
//     distance 2, fragment reads interleaved with the MFMAs and pipelined across sub-tile boundaries.
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
typedef int i32x4 __attribute__((ext_vector_type(4)));

enum { TB_SELF = 0, TB_CROSS = 1, TB_FF = 2 };
enum { K_T = 0, K_N = 1, K_O = 2 };   // transposed projection, un-transposed projection, output projection

#define MDT_MFMA_BF16 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define MDT_MFMA_F32 __builtin_amdgcn_mfma_f32_16x16x4f32

__device__ __forceinline__ float gelu_32(float x) {   // exact-erf GELU, branch-free erf (A&S 7.1.26, |error| < 1.5e-7)
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erfa = 1.0f - poly * __expf(-z * z);
  return 0.5f * x * (1.0f + copysignf(erfa, x));
}

// F32 (round 6, MDT_B_WF32 for variants 2..4): the values themselves, slots 0..3 in `hi`, 4..7 in `lo` -- operands of exact fp32
// MFMAs on fp32 fragment sub-tiles, as in k_tblock_lw.hip / k_tf256.hip
template <bool F32>
__device__ __forceinline__ void split8_32(const float v[8], bf16x8& hi, bf16x8& lo) {
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

__device__ __forceinline__ void lds_read16_32(bf16x8& dst, const unsigned char* p) {
  const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)p;
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory");
}

template <int OFF>      // fragment read with the (tile, plane) part of the address as immediate offset (k_tblock_lw.hip)
__device__ __forceinline__ void lds_read16_off(bf16x8& dst, unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read_b128 offset field");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}

template <int OFF>      // per-chunk bias vector from LDS, read like a fragment (k_tblock_lw.hip)
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
constexpr int CS = 128;         // sub-tile width: a [64][256] projection tile is streamed as two K halves, a
                                // [256][64] output tile as two row halves, both in the C = 128 tile format
constexpr int SLOT = 256 * CS;  // bytes per sub-tile (bf16 hi plane + lo plane)
constexpr int NS = 4;           // ring slots
constexpr int IPT = CS / 16;    // DMA pieces per sub-tile per loader wave
constexpr int NST = C / 32;     // k-steps of a full projection
constexpr int NCT = C / 16;     // 16-row tiles of the output projection
constexpr int NU = 4;           // units (4 fragment reads + 6 MFMAs) per sub-tile per wave
#ifdef MDT_TB32_PLAIN_PROLOGUE   // tuning build: every wave loads / normalises / splits whole rows (A/B of the shared prologue)
constexpr bool SHARED_PROLOGUE = false;
#else
constexpr bool SHARED_PROLOGUE = true;
#endif

}  // namespace


// x[m][c] += bias[c] + sum_k part[k][m][c]  (fixed order: the result does not depend on which workgroup finished first)
__global__ __launch_bounds__(256) void k_tb_reduce(float* x, const float* part, const float* bo, int M, int nsplit) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (int64_t)M * (C / 4)) return;
  const int c4 = (int)(t % (C / 4));
  float4 v = reinterpret_cast<const float4*>(part)[t];
  for (int k = 1; k < nsplit; ++k) {
    const float4 p = reinterpret_cast<const float4*>(part)[(int64_t)k * M * (C / 4) + t];
    v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
  }
  const float4 b = reinterpret_cast<const float4*>(bo)[c4];
  float4 xr = reinterpret_cast<float4*>(x)[t];
  xr.x = v.x + b.x + xr.x; xr.y = v.y + b.y + xr.y; xr.z = v.z + b.z + xr.z; xr.w = v.w + b.w + xr.w;
  store_nt(x + 4 * t, xr);
}

// NPW (MODE_CROSS only): LDS-DMA pieces per loader wave per K / V tile = ceil(context rows of the workgroup / 16)
// CHAIN: variant 4 (input = x + pin, outputs xout / pout); a separate instantiation so that the in-place kernels
// keep their exact code (the chained form costs them 1.8 % when folded in as run-time branches)
// F32 (round 6): fp32 FRAGMENT sub-tiles and exact fp32 MFMA products (v_mfma_f32_16x16x4_f32), the format of k_tf256.hip: a [64][128]
// projection sub-tile is fragments (feature tile ft, k-step st, half lo) at ft * 8192 + st * 2048 + lo * 1024, a [128][64] output
// sub-tile (row tile ct, k-step sp, half lo) at ct * 4096 + sp * 2048 + lo * 1024; lane (i, g) float r of a fragment =
// W[16 rt + i][k-slot 32 st + 8 g + 4 lo + r].  The loader waves copy such a sub-tile linearly; the attention core is fp32 either way.
template <int MODE, int NPW, bool CHAIN, bool F32>
__global__ __launch_bounds__(512) void k_tblock32(TBlockArgs a) {
  // sub-tiles per chunk: q0 q1 k0 k1 v0 v1 o0 o1 | q0 q1 K V o0 o1 (K, V = hoisted context rows) | w1a w1b w2a w2b
  constexpr int SPC = (MODE == TB_SELF) ? 8 : (MODE == TB_CROSS) ? 6 : 4;
  constexpr int KTM = 3;                           // key tiles per wave (MODE_CROSS): at most 48 context rows per 16 token rows
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  f32x4* red = reinterpret_cast<f32x4*>(smem + NS * SLOT);   // [key tile][4 waves][64 lanes] partial S^T

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // head / hidden-chunk range of this workgroup: blockIdx.y of gridDim.y workgroups share the row block (nsplit)
  const int h0 = (int)blockIdx.y * (a.nchunk / (int)gridDim.y), h1 = h0 + a.nchunk / (int)gridDim.y;
  const int NX = (MODE == TB_FF) ? a.post : 0;     // extra output sub-tiles of a folded closing convolution (mdt_kernels.h)
  const int NT = (h1 - h0) * SPC + NX;             // tau counts this workgroup's sub-tiles; the stream index is tau + h0 SPC
  const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.w);

  if (wave >= 4) {
    // ================= loader waves (see k_tblock_lw.hip) =================
    const int iw = wave - 4;
    __builtin_amdgcn_s_setprio(MDT_LOADER_PRIO);   // few instructions, all on the critical path of the stream: issue ahead of the MFMA waves
    const int lpP = lane >> 5;
    const int xP = (lane & 15) ^ lpP;
    const int baseP = ((lane >> 4) & 1) * (128 * CS) + lpP * (2 * CS);
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
      voffP[q] = F32 ? (unsigned)(inst * 1024 + lane * 16) : (unsigned)(U * (2 * CS) + ((xP ^ (U & 15)) << 4) + baseP);
      voffO[q] = F32 ? (unsigned)(inst * 1024 + lane * 16)
                     : (unsigned)(((inst * 8) / CS) * (128 * CS) + ((inst * 8) % CS) * 128 + ((xO ^ (4 * (inst & 1))) << 4) + baseO);
    }
    // MODE_CROSS: sub-tiles 2 and 3 of a head are the K and V rows (row = (sample, key), 256 B, chunks swizzled with
    // the row index) of the workgroup's samples -- layout and padding as in k_tblock_lw.hip
    const int ltl = 31 - __builtin_clz((unsigned)a.T);   // T is a power of two (host check)
    const int kv_rows = (MODE == TB_CROSS) ? (32 >> ltl) * a.Tk : 0;
    unsigned voffKV[8];                              // NPW <= 8 used (fixed bound, see k_tblock_lw.hip)
    if constexpr (MODE == TB_CROSS) {
      const int sample0 = blockIdx.x * (32 >> ltl);
      const float inv_tk = 1.0f / (float)a.Tk;
      // dual batch (classifier-free guidance, both passes in one launch): the samples of the second half read the
      // batch-invariant K / V rows a.kv2 (a workgroup never straddles the halves: the host checks B % 16 == 0)
      const int bstr = (a.kv2 && sample0 >= a.nsamples / 2) ? 0 : a.kv_bstride;
#pragma unroll
      for (int q = 0; q < NPW; ++q) {
        const int R = 4 * (iw + 4 * q) + (lane >> 4);
        const int Rc = min(R, kv_rows - 1);
        const int smq = (int)(((float)Rc + 0.5f) * inv_tk);       // Rc / Tk for Rc < 96: (Rc + 0.5) / Tk is >= 0.5 / Tk from an integer
        const int sm = min(smq, a.nsamples - 1 - sample0), key = Rc - smq * a.Tk;
        voffKV[q] = (unsigned)(((sm * bstr + key) * a.ldkv + 4 * ((lane & 15) ^ (R & 15))) * 4);
      }
    }
    auto is_kv = [&](int tau) -> bool { return (MODE == TB_CROSS) && ((tau % SPC) == 2 || (tau % SPC) == 3); };
    auto issue_kv = [&](int tau) {
      if constexpr (MODE == TB_CROSS) {
        unsigned char* slot = smem + (tau % NS) * SLOT + iw * 1024;
        const int sample0 = blockIdx.x * (32 >> ltl);
        const bool second = a.kv2 && sample0 >= a.nsamples / 2;
        const unsigned char* base = reinterpret_cast<const unsigned char*>(
            (second ? a.kv2 : a.kv + (int64_t)sample0 * a.kv_bstride * a.ldkv) + 64 * (h0 + tau / SPC) + ((tau % SPC) == 3 ? 64 * a.nheads : 0));
#pragma unroll
        for (int q = 0; q < NPW; ++q)
          __builtin_amdgcn_global_load_lds(base + voffKV[q], (__attribute__((address_space(3))) void*)(slot + q * 4096), 16, 0, 0);
      }
    };
    auto issue_w = [&](int tau) {
      const int j = tau % SPC;
      const int wt = (MODE == TB_CROSS) ? 4 * (h0 + tau / SPC) + (j < 2 ? j : j - 2) : tau + h0 * SPC;   // index into the weight stream
      const unsigned char* tile = wsrc + (int64_t)wt * SLOT;       // wave-uniform
      unsigned char* slot = smem + (tau % NS) * SLOT + iw * 1024;
      // per element select (not two loops over two arrays: hipcc then indexes a merged array dynamically, puts it in
      // scratch and waits for every scratch load with vmcnt(0), which serialises the whole DMA stream)
      const bool ptile = (j < SPC - 2 && tau < (h1 - h0) * SPC);
#pragma unroll
      for (int q = 0; q < IPT; ++q) {
        const unsigned off = ptile ? voffP[q] : voffO[q];
        __builtin_amdgcn_global_load_lds(tile + off, (__attribute__((address_space(3))) void*)(slot + q * 4096), 16, 0, 0);
      }
    };
    auto issue_tile = [&](int tau) {
      if (is_kv(tau)) issue_kv(tau);
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
    if constexpr (MODE != TB_FF && SHARED_PROLOGUE) {
      __builtin_amdgcn_s_barrier();   // S1: LayerNorm statistics pooled between the two waves of a row tile
      __builtin_amdgcn_s_barrier();   // S2: operand planes exchanged (compute waves' prologue; the feed-forward kernel keeps the plain one)
    }
    for (int k = 0; k < NT; ++k) {
      // tile k landed; tile k + 1 may be in flight (the weight-tile case first: the switch is a cascade of scalar branches, k_tf256.hip)
      if (k + 1 < NT && !is_kv(k + 1)) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else wait_vm(k + 1 < NT ? NPW : 0);
      __builtin_amdgcn_s_barrier();                                      // B(k)
      if (k + 2 < NT) issue_tile(k + 2);
    }
    prefetch_next_weights(a.pf_ptr, a.pf_lines, iw * 64 + lane);
    return;
  }

  // ================= compute waves =================
  const int i = lane & 15, g = lane >> 4;
  const int rt = wave >> 1, fh = wave & 1;
  const int row0 = blockIdx.x * 32 + rt * 16;
  const int m = row0 + i;
  const bool mvalid = m < a.M;
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
  MDT_STAMP();                                     // kernel entry
#ifdef MDT_STAMPS   // every workgroup: 100 MHz real-time clock at 8 points (dispatch spread, where the slow workgroups lose
                    // their time, tail) -> dbgbuf[256 + 8 id + k]
#define MDT_RSTAMP(K)                                                                                   \
  do {                                                                                                  \
    if (stamps && wave == 0) {                                                                          \
      unsigned long long t_, c_;                                                                        \
      asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(c_)::"memory"); \
      if (lane == 0) {                                                                                  \
        stamps[256 + 8 * (blockIdx.x + gridDim.x * blockIdx.y) + (K)] = t_;                             \
        stamps[256 + 8 * 1024 + 8 * (blockIdx.x + gridDim.x * blockIdx.y) + (K)] = c_;                  \
      }                                                                                                 \
    }                                                                                                   \
  } while (0)
#else
#define MDT_RSTAMP(K) do {} while (0)
#endif
  MDT_RSTAMP(0);
  const int mc = mvalid ? m : a.M - 1;

  bf16x8 xh[NST], xl[NST];
  constexpr int NBV = 4;                 // per-chunk bias vectors: 64 (h1 - h0) floats over 256 lanes, nchunk <= 16
  float bv[NBV];
  if constexpr (MODE == TB_FF || !SHARED_PROLOGUE)
  {
    float xr[NST][8];
    const float* xp = a.x + (int64_t)mc * a.ldx + 8 * g;
    float s = 0.f;
    float4 xu[NST], xw[NST], pu[NST], pw[NST];
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      xu[st] = *reinterpret_cast<const float4*>(xp + 32 * st);
      xw[st] = *reinterpret_cast<const float4*>(xp + 32 * st + 4);
    }
    if (CHAIN && a.pin) {                            // chained form (variant 4): block input = x + pin
      const float* pp = a.pin + (int64_t)mc * C + 8 * g;
#pragma unroll
      for (int st = 0; st < NST; ++st) {
        pu[st] = *reinterpret_cast<const float4*>(pp + 32 * st);
        pw[st] = *reinterpret_cast<const float4*>(pp + 32 * st + 4);
      }
    }
#pragma unroll
    for (int k = 0; k < NBV; ++k) bv[k] = (tid + 256 * k < 64 * (h1 - h0)) ? a.bias[64 * h0 + tid + 256 * k] : 0.f;
    // P: the loader waves start the weight stream only now, behind this wave's requests (queued behind the stream's
    // first two tiles the rows came back ~1500 cycles later).  sched_barrier: without it hipcc moves the rows' first
    // uses (and with them the wait for the loads) in front of the barrier, i.e. the stream starts a round trip late
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#ifdef MDT_STAMPS
    MDT_STAMP();                                     // past barrier P: row / bias loads issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MDT_STAMP();                                     // rows and biases arrived
    MDT_RSTAMP(1);
#endif
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      const float4 u = xu[st], w = xw[st];
      xr[st][0] = u.x; xr[st][1] = u.y; xr[st][2] = u.z; xr[st][3] = u.w;
      xr[st][4] = w.x; xr[st][5] = w.y; xr[st][6] = w.z; xr[st][7] = w.w;
    }
    if (CHAIN && a.pin) {
#pragma unroll
      for (int st = 0; st < NST; ++st) {
        xr[st][0] += pu[st].x; xr[st][1] += pu[st].y; xr[st][2] += pu[st].z; xr[st][3] += pu[st].w;
        xr[st][4] += pw[st].x; xr[st][5] += pw[st].y; xr[st][6] += pw[st].z; xr[st][7] += pw[st].w;
      }
    }
#pragma unroll
    for (int st = 0; st < NST; ++st)
#pragma unroll
      for (int e = 0; e < 8; ++e) s += xr[st][e];
    float mean = 0.f, rstd = 1.f;
    if constexpr (MODE != TB_FF) {
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
      split8_32<F32>(v, xh[st], xl[st]);
    }
  }
  else
  {
    // The two waves of a row tile need the same 16 normalised rows as MFMA operands.  Each of them loads, normalises and
    // splits only HALF of every row (channels 128 fh .. 128 fh + 127 = k-steps 4 fh .. 4 fh + 3) and hands the bf16 planes to
    // the other through ring slot 2 (free until B(0)); the LayerNorm statistics of the two halves are pooled exactly
    // (mean and sum of squared deviations per half, Chan et al.) through the head of slot 3.  Before, both waves did
    // all of it: 128 KB of row requests per workgroup through a 64 B/clk path and ~500 VALU instructions per lane ahead
    // of the first MFMA (launch timeline).
    constexpr int NH = NST / 2;
    float xr[NH][8];
    const float* xp = a.x + (int64_t)mc * a.ldx + 8 * g + 128 * fh;
    float4 xu[NH], xw[NH], pu[NH], pw[NH];
#pragma unroll
    for (int st = 0; st < NH; ++st) {
      xu[st] = *reinterpret_cast<const float4*>(xp + 32 * st);
      xw[st] = *reinterpret_cast<const float4*>(xp + 32 * st + 4);
    }
    if (CHAIN && a.pin) {                            // chained form (variant 4): block input = x + pin
      const float* pp = a.pin + (int64_t)mc * C + 8 * g + 128 * fh;
#pragma unroll
      for (int st = 0; st < NH; ++st) {
        pu[st] = *reinterpret_cast<const float4*>(pp + 32 * st);
        pw[st] = *reinterpret_cast<const float4*>(pp + 32 * st + 4);
      }
    }
#pragma unroll
    for (int k = 0; k < NBV; ++k) bv[k] = (tid + 256 * k < 64 * (h1 - h0)) ? a.bias[64 * h0 + tid + 256 * k] : 0.f;
    // P: the loader waves start the weight stream only now, behind this wave's requests (queued behind the stream's
    // first two tiles the rows came back ~1500 cycles later).  sched_barrier: without it hipcc moves the rows' first
    // uses (and with them the wait for the loads) in front of the barrier, i.e. the stream starts a round trip late
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#ifdef MDT_STAMPS
    MDT_STAMP();                                     // past barrier P: row / bias loads issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MDT_STAMP();                                     // rows and biases arrived
    MDT_RSTAMP(1);
#endif
#pragma unroll
    for (int st = 0; st < NH; ++st) {
      const float4 u = xu[st], w = xw[st];
      xr[st][0] = u.x; xr[st][1] = u.y; xr[st][2] = u.z; xr[st][3] = u.w;
      xr[st][4] = w.x; xr[st][5] = w.y; xr[st][6] = w.z; xr[st][7] = w.w;
    }
    if (CHAIN && a.pin) {
#pragma unroll
      for (int st = 0; st < NH; ++st) {
        xr[st][0] += pu[st].x; xr[st][1] += pu[st].y; xr[st][2] += pu[st].z; xr[st][3] += pu[st].w;
        xr[st][4] += pw[st].x; xr[st][5] += pw[st].y; xr[st][6] += pw[st].z; xr[st][7] += pw[st].w;
      }
    }
    float mean = 0.f, rstd = 1.f;
    {
      float s = 0.f;
#pragma unroll
      for (int st = 0; st < NH; ++st)
#pragma unroll
        for (int e = 0; e < 8; ++e) s += xr[st][e];
      s = xg16_add(s);
      s = xg32_add(s);
      const float mean_h = s / (float)(C / 2);
      float m2 = 0.f;
#pragma unroll
      for (int st = 0; st < NH; ++st)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = xr[st][e] - mean_h;
          m2 += d * d;
        }
      m2 = xg16_add(m2);
      m2 = xg32_add(m2);
      float2* stat = reinterpret_cast<float2*>(smem + 3 * SLOT);      // [wave][row]
      if (g == 0) stat[wave * 16 + i] = make_float2(mean_h, m2);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                  // S1 (the loader waves take part)
      const float2 other = stat[(wave ^ 1) * 16 + i];
      const float dm = mean_h - other.x;
      mean = 0.5f * (mean_h + other.x);
      const float ss = m2 + other.y + (float)(C / 4) * dm * dm;
      rstd = 1.0f / sqrtf(ss / (float)C + a.eps);
    }
    unsigned char* xch = smem + 2 * SLOT + rt * (NST * 2048) + lane * 16;     // [row tile][k-step][plane][lane] 16-byte operands
    const float rs = mvalid ? rstd : 0.f;            // rows past M: zero operands (their inputs are a real row's, finite)
#pragma unroll
    for (int st = 0; st < NH; ++st) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (xr[st][e] - mean) * rs;
      bf16x8 hh, ll;
      split8_32<F32>(v, hh, ll);
      *reinterpret_cast<bf16x8*>(xch + (NH * fh + st) * 2048) = hh;
      *reinterpret_cast<bf16x8*>(xch + (NH * fh + st) * 2048 + 1024) = ll;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                    // S2
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      xh[st] = *reinterpret_cast<const bf16x8*>(xch + st * 2048);
      xl[st] = *reinterpret_cast<const bf16x8*>(xch + st * 2048 + 1024);
    }
  }
  // fragment addressing inside a sub-tile (C = 128 tile format), this wave's feature half folded in:
  //   projection sub-tile: row = 32 fh + 16 q + i, chunk = 4 st + g ; output sub-tile: row = 16 ct + i, chunk = 4 fh + g
  int aP[4];
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    const int lc = 4 * st + g;
    // F32: the lane's 16 bytes of a fragment, this wave's feature tiles ft = 2 fh + q (q in the immediate) and the k-step
    aP[st] = F32 ? lane * 16 + fh * 16384 + st * 2048 : fh * (2 * 16 * 4 * CS) + i * (4 * CS) + ((lc & ~15) | ((lc & 15) ^ i)) * 16;
  }
  const int aO = F32 ? lane * 16 + fh * 2048 : i * 128 + ((4 * fh + g) ^ ((i >> 1) & 7)) * 16;   // F32: this wave's k-step sp = fh

  bf16x8 frh[3][2], frl[3][2];
  // read j (= 2 q + plane) of unit u; `base` = LDS address of the slot + the lane's swizzled part (projection
  // sub-tiles: aP[u], output sub-tiles: aO), the rest is the instruction's immediate offset
  auto frag_read = [&](auto kind, unsigned base, auto uc, int set, auto jc) {
    constexpr int KIND = decltype(kind)::value, u = decltype(uc)::value, j = decltype(jc)::value;
    constexpr int q = j >> 1, lo = j & 1;
    constexpr int off = F32 ? ((KIND == K_O) ? ((2 * u + q) * 4096 + lo * 1024) : (q * 8192 + lo * 1024))
                            : ((KIND == K_O) ? ((2 * u + q) * 16 * 128 + lo * (CS * 128)) : (q * 16 * 4 * CS + lo * (2 * CS)));
    lds_read16_off<off>(lo ? frl[set][q] : frh[set][q], base);
  };
  using J0 = std::integral_constant<int, 0>;
  using J1 = std::integral_constant<int, 1>;
  using J2 = std::integral_constant<int, 2>;
  using J3 = std::integral_constant<int, 3>;
  auto prefetch2 = [&](auto kind, const unsigned char* slot, int off) {
    constexpr int KIND = decltype(kind)::value;
    const unsigned l = lds_addr(slot);
    const unsigned b0 = l + (KIND == K_O ? aO : aP[0]), b1 = l + (KIND == K_O ? aO : aP[1]);
    frag_read(kind, b0, J0{}, off % 3, J0{}); frag_read(kind, b0, J0{}, off % 3, J1{});
    frag_read(kind, b0, J0{}, off % 3, J2{}); frag_read(kind, b0, J0{}, off % 3, J3{});
    frag_read(kind, b1, J1{}, (off + 1) % 3, J0{}); frag_read(kind, b1, J1{}, (off + 1) % 3, J1{});
    frag_read(kind, b1, J1{}, (off + 1) % 3, J2{}); frag_read(kind, b1, J1{}, (off + 1) % 3, J3{});
  };

  int tau = 0;
  auto slot_of = [&](int t) -> const unsigned char* { return smem + (t % NS) * SLOT; };

  // One MFMA phase over a sub-tile: 4 units of 4 fragment reads + 6 MFMAs.  Projection kinds: unit = k-step u of the
  // K half, accumulators acc[0..1] = the wave's two feature tiles, operands bh[u]/bl[u].  Output kind: unit = row
  // tiles 2u, 2u+1 of the row half, accumulators acc[2u..2u+1], the single operand bh[0]/bl[0].
  auto phase = [&](auto kind, auto offc, auto nkind, bool has_next, f32x4* acc, const bf16x8* bh, const bf16x8* bl) {
    constexpr int KIND = decltype(kind)::value, OFF = decltype(offc)::value, NK = decltype(nkind)::value;
    const unsigned lc = lds_addr(slot_of(tau)), ln = lds_addr(slot_of(tau + 1));
    unsigned bc[4], bn[2];                           // per unit of this sub-tile / of the first two units of the next
#pragma unroll
    for (int k = 0; k < 4; ++k) bc[k] = lc + (KIND == K_O ? aO : aP[k]);
#pragma unroll
    for (int k = 0; k < 2; ++k) bn[k] = ln + (NK == K_O ? aO : aP[k]);
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
      constexpr int ia = (KIND == K_O) ? 2 * u : 0, ib = (KIND == K_O) ? 0 : u;
      auto rd = [&](auto jc) {
        if (!pre) return;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (in_phase) frag_read(kind, bc[u + 2], std::integral_constant<int, u + 2>{}, s2, jc);
        else frag_read(nkind, bn[u + 2 - NU], std::integral_constant<int, u + 2 - NU>{}, s2, jc);
        __builtin_amdgcn_sched_barrier(0);
      };
      auto mm = [&](const bf16x8& w, const bf16x8& x, int q) {
        if constexpr (KIND == K_N) acc[ia + q] = MDT_MFMA_BF16(x, w, acc[ia + q], 0, 0, 0);
        else acc[ia + q] = MDT_MFMA_BF16(w, x, acc[ia + q], 0, 0, 0);
      };
      if constexpr (F32) {
        // exact fp32: fragment (q, half) x operand half, four 16x16x4 MFMAs each (r = contraction sub-step); the two accumulators
        // alternate so that no MFMA waits for the one issued just before it (k_tblock_lw.hip)
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
        mm4(frh[s0][0], frh[s0][1], bh[ib], J0{}); rd(J0{});
        mm4(frh[s0][0], frh[s0][1], bh[ib], J2{}); rd(J1{});
        mm4(frl[s0][0], frl[s0][1], bl[ib], J0{}); rd(J2{});
        mm4(frl[s0][0], frl[s0][1], bl[ib], J2{}); rd(J3{});
      } else {
        mm(frl[s0][0], bh[ib], 0); rd(J0{});
        mm(frl[s0][1], bh[ib], 1); rd(J1{});
        mm(frh[s0][0], bl[ib], 0); rd(J2{});
        mm(frh[s0][1], bl[ib], 1); rd(J3{});
        mm(frh[s0][0], bh[ib], 0);
        mm(frh[s0][1], bh[ib], 1);
      }
      __builtin_amdgcn_sched_barrier(0);
#ifdef MDT_STAMPS_UNITS
      MDT_STAMP();
#elif defined(MDT_STAMPS)
      if (tau == 0) MDT_STAMP();                     // the first sub-tile unit by unit: where the stream's start-up wait sits
#endif
    };
    unit(std::integral_constant<int, 0>{}); unit(std::integral_constant<int, 1>{});
    unit(std::integral_constant<int, 2>{}); unit(std::integral_constant<int, 3>{});
    ++tau;
    MDT_STAMP();
  };
  using IC0 = std::integral_constant<int, 0>;
  using IC1 = std::integral_constant<int, 1>;
  using IC2 = std::integral_constant<int, 2>;
  const IC0 kT{};
  const IC1 kN{};
  const IC2 kO{};

  f32x4 accT[NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) accT[ct] = f32x4{0.f, 0.f, 0.f, 0.f};

  const float* bias = a.bias;
  const int bo_off = (MODE == TB_SELF) ? 3 * 64 * a.nchunk : 64 * a.nchunk;   // [bq | bk | bv | bo] / [bq | bo] / [b1 | b2]
  const int lt = 31 - __builtin_clz((unsigned)a.T);  // 16 % T == 0 (host check): T is a power of two, x / T = x >> lt.  A run-time
                                                     // integer division is ~35 VALU instructions; the 17 of this section cost ~2500 cycles
  const int samp_q = i >> lt;                        // loop-invariant softmax pieces, see k_tblock_lw.hip
  float kmask[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) kmask[r] = (((4 * g + r) >> lt) == samp_q) ? 0.f : -INFINITY;
  const float scale2 = a.scale * 1.44269504088896340736f;
  // MODE_CROSS: this row tile's context rows inside a K / V tile start at row rt * nkeys; lane (i, g) reads K row
  // 16 kt + i and V rows 16 kt + 4 g + r (clamped to a real row past the end: masked scores, zero probabilities).
  // Validity of key column 16 kt + 4 g + r for query column i is bit 4 kt + r of okbits.
  int nkeys = 0, Rw = 0;
  unsigned okbits = 0;
  if constexpr (MODE == TB_CROSS) {
    nkeys = (16 >> lt) * a.Tk;
    Rw = rt * nkeys;
#pragma unroll
    for (int kt = 0; kt < KTM; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int jj = 16 * kt + 4 * g + r;
        if (jj < nkeys && jj >= samp_q * a.Tk && jj < (samp_q + 1) * a.Tk) okbits |= 1u << (4 * kt + r);   // jj / Tk == samp_q
      }
  }
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};

  // this workgroup's per-chunk bias vectors (bq | b1 of chunks h0 .. h1-1) -> LDS behind the ring and the exchange area
  float* bias_s = reinterpret_cast<float*>(smem + NS * SLOT + (MODE == TB_CROSS ? 3 : 1) * 4 * 64 * 16);
#pragma unroll
  for (int k = 0; k < NBV; ++k)
    if (tid + 256 * k < 64 * (h1 - h0)) bias_s[tid + 256 * k] = bv[k];
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  MDT_STAMP();                                       // rows loaded, normalised and split
  MDT_RSTAMP(2);
  __builtin_amdgcn_s_barrier();                      // B(0)
  prefetch2(kT, slot_of(0), 0);
  const unsigned bias_l = lds_addr(reinterpret_cast<const unsigned char*>(bias_s)) + 128 * fh + 16 * g;   // + 256 (h - h0)

  MDT_STAMP();
  for (int h = h0; h < h1; ++h) {
    const bool more = h + 1 < h1;
    f32x4 oT[2];      // this wave's 32 features of the chunk: [feature 32 fh + 16 q + 4 g + r][token i]
    if constexpr (MODE == TB_FF) {
      f32x4 b1[2];
      lds_read_f4_off<0>(b1[0], bias_l + 256 * (h - h0)); lds_read_f4_off<64>(b1[1], bias_l + 256 * (h - h0));
      oT[0] = zero4; oT[1] = zero4;
      phase(kT, IC0{}, kT, true, oT, xh, xl);               // K half 0
      phase(kT, IC1{}, kT, false, oT, xh + 4, xl + 4);      // K half 1
      __builtin_amdgcn_s_barrier();                         // B(first W2 sub-tile)
      prefetch2(kO, slot_of(tau), 1);
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) oT[q][r] = gelu_32(oT[q][r] + b1[q][r]);
    } else if constexpr (MODE == TB_CROSS) {
      f32x4 qT[2], bq[2];
      lds_read_f4_off<0>(bq[0], bias_l + 256 * (h - h0)); lds_read_f4_off<64>(bq[1], bias_l + 256 * (h - h0));
      qT[0] = zero4; qT[1] = zero4;
      phase(kT, IC0{}, kT, true, qT, xh, xl);
      phase(kT, IC1{}, kT, false, qT, xh + 4, xl + 4);
      __builtin_amdgcn_s_barrier();                         // B(K tile)
      MDT_STAMP();
      const unsigned char* sk = slot_of(tau);
      qT[0] += bq[0];
      qT[1] += bq[1];
      // Partial S^T over this wave's 32 features for all key tiles: branch-free (rows past the wave's keys are clamped
      // to a real row, their scores are masked below), so that the six K reads go out together and the six
      // independent MFMA chains interleave instead of read -> wait -> 8 dependent MFMAs -> write per key tile.
      f32x4 st[KTM];
      float4 k0[KTM], k1[KTM];
#pragma unroll
      for (int kt = 0; kt < KTM; ++kt) {
        const int R = Rw + min(16 * kt + i, nkeys - 1);
        const unsigned char* kp = sk + R * 256;
        k0[kt] = *reinterpret_cast<const float4*>(kp + (((8 * fh + g) ^ (R & 15)) << 4));
        k1[kt] = *reinterpret_cast<const float4*>(kp + (((8 * fh + 4 + g) ^ (R & 15)) << 4));
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
#pragma unroll
      for (int kt = 0; kt < KTM; ++kt) {
        st[kt] = sp0[kt] + sp1[kt];
        red[(kt * 4 + wave) * 64 + lane] = st[kt];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      ++tau;
      MDT_STAMP();
      __builtin_amdgcn_s_barrier();                         // B(V tile) + partial exchange
      MDT_STAMP();
      const unsigned char* sv = slot_of(tau);
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < KTM; ++kt) {
        const f32x4 other = red[(kt * 4 + (wave ^ 1)) * 64 + lane];
        const f32x4 s01 = fh ? (other + st[kt]) : (st[kt] + other);   // same sum in both waves of the row tile
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float sv2 = ((okbits >> (4 * kt + r)) & 1u) ? s01[r] * scale2 : -INFINITY;   // okbits: key exists, same sample
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
      MDT_STAMP();
      oT[0] = zero4; oT[1] = zero4;
      f32x4 v0[KTM], v1[KTM];                               // V[key 16 kt + 4 g + r][32 fh + 16 dt + i], dt = 0, 1
#pragma unroll
      for (int kt = 0; kt < KTM; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int R = Rw + min(16 * kt + 4 * g + r, nkeys - 1);
          const unsigned char* vp = sv + R * 256 + (i & 3) * 4;
          v0[kt][r] = *reinterpret_cast<const float*>(vp + (((8 * fh + (i >> 2)) ^ (R & 15)) << 4));
          v1[kt][r] = *reinterpret_cast<const float*>(vp + (((8 * fh + 4 + (i >> 2)) ^ (R & 15)) << 4));
        }
#pragma unroll
      for (int kt = 0; kt < KTM; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = st[kt][r] * inv;                  // exactly 0 for masked keys
          oT[0] = MDT_MFMA_F32(v0[kt][r], p, oT[0], 0, 0, 0);
          oT[1] = MDT_MFMA_F32(v1[kt][r], p, oT[1], 0, 0, 0);
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // V reads complete before the slot can be refilled
      ++tau;
      MDT_STAMP();
      __builtin_amdgcn_s_barrier();                         // B(first output sub-tile)
      MDT_STAMP();
      prefetch2(kO, slot_of(tau), 1);
    } else {
      f32x4 qT[2], kTt[2], vT[2];
      f32x4 bq[2];
      // (the host folds the k bias away -- softmax-invariant -- and the v bias into the output bias)
      lds_read_f4_off<0>(bq[0], bias_l + 256 * (h - h0)); lds_read_f4_off<64>(bq[1], bias_l + 256 * (h - h0));
#pragma unroll
      for (int q = 0; q < 2; ++q) { kTt[q] = zero4; vT[q] = zero4; qT[q] = zero4; }
      phase(kT, IC0{}, kT, true, qT, xh, xl);
      phase(kT, IC1{}, kT, true, qT, xh + 4, xl + 4);
      phase(kT, IC2{}, kT, true, kTt, xh, xl);
      phase(kT, IC0{}, kN, true, kTt, xh + 4, xl + 4);
      phase(kN, IC1{}, kN, true, vT, xh, xl);
      phase(kN, IC2{}, kN, false, vT, xh + 4, xl + 4);
      qT[0] += bq[0];
      qT[1] += bq[1];
      // partial S^T over this wave's 32 features; the partner wave (other feature half) holds the rest
      f32x4 sp0 = zero4, sp1 = zero4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sp0 = MDT_MFMA_F32(kTt[0][r], qT[0][r], sp0, 0, 0, 0);
        sp1 = MDT_MFMA_F32(kTt[1][r], qT[1][r], sp1, 0, 0, 0);
      }
      const f32x4 mine = sp0 + sp1;
      red[wave * 64 + lane] = mine;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                         // B(first output sub-tile) + partial exchange
      const f32x4 other = red[(wave ^ 1) * 64 + lane];
      prefetch2(kO, slot_of(tau), 1);
      // both waves of a row tile must form the SAME sum: add in feature-half order
      const f32x4 s01 = fh ? (other + mine) : (mine + other);
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
      oT[0] = zero4; oT[1] = zero4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = st[r] * inv;
        oT[0] = MDT_MFMA_F32(vT[0][r], p, oT[0], 0, 0, 0);
        oT[1] = MDT_MFMA_F32(vT[1][r], p, oT[1], 0, 0, 0);
      }
    }
    MDT_STAMP();
    // ---- output projection: k-step = this wave's 32 features of the chunk ----
    bf16x8 oh[1], ol[1];
    {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = oT[e >> 2][e & 3];
      split8_32<F32>(v, oh[0], ol[0]);
    }
    phase(kO, IC1{}, kO, true, accT, oh, ol);               // output rows 0..127
    if (NX > 0 && !more) phase(kO, IC2{}, kO, true, accT + 8, oh, ol);   // the folded convolution's sub-tiles follow
    else phase(kO, IC2{}, kT, more, accT + 8, oh, ol);      // output rows 128..255
    if (h - h0 < 4) MDT_RSTAMP(3 + (h - h0));
  }
  if constexpr (MODE == TB_FF) {
    if (NX > 0) {
      // + Wout x (folded closing convolution): per 64-channel k chunk of the raw x operands one output tile = two row-half
      // sub-tiles; this wave's k-step is the chunk's half fh (a register select: fh is not a compile-time index)
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
      pick(0); phase(kO, IC0{}, kO, true, accT, oxh, oxl); phase(kO, IC1{}, kO, true, accT + 8, oxh, oxl);
      pick(1); phase(kO, IC2{}, kO, true, accT, oxh, oxl); phase(kO, IC0{}, kO, true, accT + 8, oxh, oxl);
      pick(2); phase(kO, IC1{}, kO, true, accT, oxh, oxl); phase(kO, IC2{}, kO, true, accT + 8, oxh, oxl);
      pick(3); phase(kO, IC0{}, kO, true, accT, oxh, oxl); phase(kO, IC1{}, kT, false, accT + 8, oxh, oxl);
    }
  }

  // ---- sum the two feature halves' accumulators through LDS (the ring is idle now), add bias + residual ----
  // wave (rt, fh) finalises row tiles 8 fh .. 8 fh + 7 (read back from LDS: a register array cannot be indexed by fh)
  __builtin_amdgcn_s_barrier();                       // every wave is done with the last weight slot
  // who writes what: in-place / chained head group 0 -> block output (partial + bias + input); every other head
  // group -> its bare partial (variant 3: slot blockIdx.y of `part` for k_tb_reduce; chained: pout)
  const bool bare = gridDim.y > 1 && !(CHAIN && blockIdx.y == 0);
  const bool full = mvalid && !bare;
  // The residual rows and the output bias are requested HERE, between the two halves of the accumulator exchange through LDS:
  // their ~1 us round trip (in-kernel stamps: 3700 cycles from this barrier to the last store, ~2500 of them waiting
  // for these loads behind the second barrier) overlaps the exchange.  Every load of the epilogue is requested before
  // the first store: the output aliases the residual rows (in place), and with loads and stores alternating hipcc kept
  // them in order -- eight exposed round trips.
  f32x4* part = reinterpret_cast<f32x4*>(smem);       // [4 waves][16][64 lanes] = 64 KB
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) part[(wave * NCT + ct) * 64 + lane] = accT[ct];
  __builtin_amdgcn_sched_barrier(0);
  float4 bo[8], xr[8], pr[8];
  {
    if (full) {
      const float* xi = a.x + (int64_t)m * a.ldx + 4 * g + 128 * fh;
      const float* bi = bias + bo_off + 4 * g + 128 * fh;
#pragma unroll
      for (int c = 0; c < 8; ++c) bo[c] = *reinterpret_cast<const float4*>(bi + 16 * c);
      if (NX == 0) {                                         // folded closing convolution: Wout x is already in the sum
#pragma unroll
        for (int c = 0; c < 8; ++c) xr[c] = *reinterpret_cast<const float4*>(xi + 16 * c);
      }
      if (CHAIN && a.pin && NX == 0) {                       // chained form: the block input was x + pin
        const float* pi = a.pin + (int64_t)m * C + 4 * g + 128 * fh;
#pragma unroll
        for (int c = 0; c < 8; ++c) pr[c] = *reinterpret_cast<const float4*>(pi + 16 * c);
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (mvalid) {
    auto halves = [&](int c) -> f32x4 {                       // feature half 0 first
      const int ct = 8 * fh + c;
      return part[((2 * rt) * NCT + ct) * 64 + lane] + part[((2 * rt + 1) * NCT + ct) * 64 + lane];
    };
    if (bare) {
      float* po = CHAIN ? a.pout + (int64_t)m * C + 4 * g : a.part + ((int64_t)blockIdx.y * a.M + m) * C + 4 * g;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const f32x4 sum = halves(c);
        store_nt(po + 16 * (8 * fh + c), make_float4(sum[0], sum[1], sum[2], sum[3]));
      }
    } else {
      float* xo = ((CHAIN || NX > 0) ? a.xout : a.x) + (int64_t)m * a.ldx + 4 * g + 128 * fh;
      if (NX != 0) {
#pragma unroll
        for (int c = 0; c < 8; ++c) xr[c] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (CHAIN && a.pin && NX == 0) {
#pragma unroll
        for (int c = 0; c < 8; ++c) { xr[c].x += pr[c].x; xr[c].y += pr[c].y; xr[c].z += pr[c].z; xr[c].w += pr[c].w; }
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const f32x4 sum = halves(c);
        store_nt(xo + 16 * c, make_float4(sum[0] + bo[c].x + xr[c].x, sum[1] + bo[c].y + xr[c].y,
                                          sum[2] + bo[c].z + xr[c].z, sum[3] + bo[c].w + xr[c].w));
      }
    }
  }
#ifdef MDT_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  MDT_STAMP();                                       // outputs stored
  MDT_RSTAMP(7);
#endif
}

template <int MODE, int NPW, bool CHAIN, bool F32>
static hipError_t launch_32f(const TBlockArgs& a, hipStream_t s) {
  const size_t smem = (size_t)NS * SLOT + (MODE == TB_CROSS ? 3 : 1) * 4 * 64 * 16 + (size_t)64 * a.nchunk * sizeof(float);
  static DevOnce attr_once;                          // per device (mdt_kernels.h)
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tblock32<MODE, NPW, CHAIN, F32>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
  }
  const int nsplit = a.nsplit > 1 ? a.nsplit : 1;
  hipLaunchKernelGGL((k_tblock32<MODE, NPW, CHAIN, F32>), dim3((unsigned)((a.M + 31) / 32), (unsigned)nsplit), dim3(512), smem, s, a);
  if (nsplit > 1 && !a.xout) {                        // variant 3: separate fixed-order reduce (the chained form needs none)
    const int bo_off = (MODE == TB_SELF) ? 3 * 64 * a.nchunk : 64 * a.nchunk;
    const int64_t n4 = (int64_t)a.M * (C / 4);
    hipLaunchKernelGGL(k_tb_reduce, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, a.x, a.part, a.bias + bo_off, a.M, nsplit);
  }
  return hipGetLastError();
}

template <int MODE, int NPW, bool CHAIN>
static hipError_t launch_32c(const TBlockArgs& a, hipStream_t s) {
  return a.wf32 ? launch_32f<MODE, NPW, CHAIN, true>(a, s) : launch_32f<MODE, NPW, CHAIN, false>(a, s);   // fp32 fragment sub-tiles: exact products
}

template <int MODE, int NPW = 0>
static hipError_t launch_32(const TBlockArgs& a, hipStream_t s) {
  // the chained instantiation also serves a folded closing convolution whose input is x + pin
  return (a.xout && (!a.post || a.pin)) ? launch_32c<MODE, NPW, true>(a, s) : launch_32c<MODE, NPW, false>(a, s);
}

hipError_t launch_tblock32(const TBlockArgs& a, hipStream_t s) {
  if (a.M <= 0) return hipSuccess;
  if (a.C != 256 || a.T <= 0 || 16 % a.T || a.nchunk <= 0 || a.nchunk > 16) return hipErrorInvalidValue;
  if (a.post && (a.mode != TB_FF || a.post != 8 || a.nsplit > 1 || !a.xout)) return hipErrorInvalidValue;
  if (a.nsplit > 1 && (a.nchunk % a.nsplit || !(a.xout ? (void*)a.pout : (void*)a.part))) return hipErrorInvalidValue;
  if (a.xout && (a.nsplit > 2 || a.xout == a.x)) return hipErrorInvalidValue;
  if (a.mode == TB_CROSS) {
    if (a.Tk <= 0 || (16 / a.T) * a.Tk > 48) return hipErrorInvalidValue;   // three key tiles per wave at most
    switch (((32 / a.T) * a.Tk + 15) / 16) {
      case 1: return launch_32<TB_CROSS, 1>(a, s);
      case 2: return launch_32<TB_CROSS, 2>(a, s);
      case 3: return launch_32<TB_CROSS, 3>(a, s);
      case 4: return launch_32<TB_CROSS, 4>(a, s);
      case 5: return launch_32<TB_CROSS, 5>(a, s);
      case 6: return launch_32<TB_CROSS, 6>(a, s);
      default: return hipErrorInvalidValue;
    }
  }
  if (a.mode != TB_SELF && a.mode != TB_FF) return hipErrorInvalidValue;
  return a.mode == TB_SELF ? launch_32<TB_SELF>(a, s) : launch_32<TB_FF>(a, s);
}

}  // namespace mdt
