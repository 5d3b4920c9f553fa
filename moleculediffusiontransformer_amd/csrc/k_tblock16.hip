// Fused transformer sub-blocks, feature-split variant for levels with FEW rows (e.g. C = 256, 4 tokens per
// sample: 4096 rows at B = 1024).  Same operator, weight-tile stream and LDS-DMA ring as k_tblock.hip, but the
// 64-row workgroup of that kernel would give only 64 workgroups and a ~70 us serial chain per wave.  Here a
// workgroup owns 16 rows (256 workgroups at B = 1024) and its four waves split every 64-feature chunk:
//
//   * every wave holds the same 16 normalised rows (bf16 hi/lo fragments in registers);
//   * wave w computes feature tile w (16 features) of q^T, k^T, v (or of the FF hidden chunk), i.e. one quarter
//     of each projection tile, so the per-wave MFMA chain per head shrinks from 416 to 128 instructions;
//   * S^T = K Q^T is a sum over features: each wave forms the partial product over its 16 features
//     (4 fp32 MFMAs), the four 16x16 partials are summed through 4 KB of LDS, and every wave runs the
//     (cheap) softmax on the full tile;
//   * O^T: wave w produces output features 16w..16w+15 from its own v columns;
//   * output projection: wave w contributes the K-slice of its 16 features (the other half of the 32-deep
//     bf16 MFMA k-step is fed zeros) into a private [C][16] accumulator; the four accumulators are summed
//     once, after the last chunk, through LDS, together with bias and residual.
#include <cstdlib>

#include "mdt_kernels.h"

namespace mdt {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { TB16_SELF = 0, TB16_CROSS = 1, TB16_FF = 2 };

__device__ __forceinline__ float gelu_tb16(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erfa = 1.0f - poly * __expf(-z * z);
  return 0.5f * x * (1.0f + copysignf(erfa, x));
}

__device__ __forceinline__ void split8_16(const float v[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 h = (__bf16)v[e];
    hi[e] = h;
    lo[e] = (__bf16)(v[e] - (float)h);
  }
}

#define MDT16_MFMA_BF16 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define MDT16_MFMA_F32 __builtin_amdgcn_mfma_f32_16x16x4f32

template <int MODE, int C>
__global__ __launch_bounds__(256) void k_tblock16(TBlockArgs a) {
  constexpr int SLOT = 256 * C;
  constexpr int NS = (C == 128) ? 4 : 2;
  constexpr int IPT = C / 16;
  constexpr int TPC = (MODE == TB16_SELF) ? 4 : 2;
  constexpr int NST = C / 32;
  constexpr int NCT = C / 16;
  constexpr int KTM = (MODE == TB16_CROSS) ? 4 : 1;

  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  float* bias_s = reinterpret_cast<float*>(smem + NS * SLOT);
  f32x4* red = reinterpret_cast<f32x4*>(smem + NS * SLOT + ((a.nbias + 3) / 4) * 16);   // [KTM][4 waves][64 lanes]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int row0 = blockIdx.x * 16;
  const int m = row0 + i;
  const bool mvalid = m < a.M;
  const int mc = mvalid ? m : a.M - 1;
  const int NT = a.nchunk * TPC;
  const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.w);

  for (int t = tid; t < a.nbias; t += 256) bias_s[t] = a.bias[t];

  // ---- the workgroup's 16 rows, replicated in every wave, in MFMA operand layout ----
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
    if constexpr (MODE != TB16_FF) {
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
      split8_16(v, xh[st], xl[st]);
    }
  }
  __syncthreads();

  // ---- weight ring (identical to k_tblock.hip: see there for the swizzle algebra) ----
  const int lpP = (C == 128) ? (lane >> 5) : 0;
  const int xP = (lane & 15) ^ lpP;
  const int baseP = ((C == 128) ? ((lane >> 4) & 1) : (lane >> 5)) * (128 * C) + lpP * (2 * C) +
                    ((C == 128) ? 0 : (lane & 16) * 16);
  const int xO = (lane & 7) ^ (lane >> 4);
  const int baseO = (lane >> 3) * 128;
  auto issue_tile = [&](int tau) {
    const unsigned char* tile = wsrc + (int64_t)tau * SLOT;
    unsigned char* slot = smem + (tau % NS) * SLOT;
    const bool otile = (tau % TPC) == TPC - 1;
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
      const int inst = wave + 4 * q;
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
  auto wait_vm = [&](int allow) {
    if (allow >= 56) asm volatile("s_waitcnt vmcnt(56)" ::: "memory");
    else if (allow >= 36) asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
    else if (allow >= 28) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
    else if (allow >= 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if (allow >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (allow >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (allow >= 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  int tau = 0;
  auto acquire = [&](int extra_vm = 0) -> const unsigned char* {
    const int after = min(NS - 2, NT - 1 - tau);
    wait_vm(after * IPT + extra_vm);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (tau + NS - 1 < NT) issue_tile(tau + NS - 1);
    const unsigned char* slot = smem + (tau % NS) * SLOT;
    ++tau;
    return slot;
  };
  int aP[NST], aO[2];
#pragma unroll
  for (int st = 0; st < NST; ++st) {
    const int lc = 4 * st + g;
    aP[st] = i * (4 * C) + ((lc & ~15) | ((lc & 15) ^ i)) * 16;
  }
#pragma unroll
  for (int sp = 0; sp < 2; ++sp) aO[sp] = i * 128 + ((4 * sp + g) ^ ((i >> 1) & 7)) * 16;
  // fragments of feature tile `wave` of a projection tile / of row tile ct of an output tile
  auto fragP = [&](const unsigned char* slot, int st, int plane) -> bf16x8 {
    return *reinterpret_cast<const bf16x8*>(slot + aP[st] + wave * (16 * 4 * C) + plane * (2 * C));
  };
  auto fragO = [&](const unsigned char* slot, int ct, int sp, int plane) -> bf16x8 {
    return *reinterpret_cast<const bf16x8*>(slot + aO[sp] + (ct * 16 * 128 + plane * C * 128));
  };
  // this wave's 16 features of a transposed projection: out[r] = (W x^T)[feature 16 wave + 4 g + r][token i]
  auto proj_T = [&](const unsigned char* slot, const float* bias16) -> f32x4 {
    bf16x8 fh[3][2], fl[3][2];
    auto load = [&](int u, int set) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        fh[set][q] = fragP(slot, 2 * u + q, 0);
        fl[set][q] = fragP(slot, 2 * u + q, 1);
      }
    };
    constexpr int NU = NST / 2;
    load(0, 0);
    if (NU > 1) load(1, 1);
    f32x4 acc = *reinterpret_cast<const f32x4*>(bias16 + 4 * g);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      if (u + 2 < NU) load(u + 2, (u + 2) % 3);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int st = 2 * u + q;
        acc = MDT16_MFMA_BF16(fl[u % 3][q], xh[st], acc, 0, 0, 0);
        acc = MDT16_MFMA_BF16(fh[u % 3][q], xl[st], acc, 0, 0, 0);
        acc = MDT16_MFMA_BF16(fh[u % 3][q], xh[st], acc, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    return acc;
  };
  // un-transposed: out[r] = (x W^T)[token 4 g + r][feature 16 wave + i]
  auto proj_N = [&](const unsigned char* slot, const float* bias16) -> f32x4 {
    bf16x8 fh[3][2], fl[3][2];
    auto load = [&](int u, int set) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        fh[set][q] = fragP(slot, 2 * u + q, 0);
        fl[set][q] = fragP(slot, 2 * u + q, 1);
      }
    };
    constexpr int NU = NST / 2;
    load(0, 0);
    if (NU > 1) load(1, 1);
    const float b = bias16[i];
    f32x4 acc = f32x4{b, b, b, b};
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      if (u + 2 < NU) load(u + 2, (u + 2) % 3);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int st = 2 * u + q;
        acc = MDT16_MFMA_BF16(xl[st], fh[u % 3][q], acc, 0, 0, 0);
        acc = MDT16_MFMA_BF16(xh[st], fl[u % 3][q], acc, 0, 0, 0);
        acc = MDT16_MFMA_BF16(xh[st], fh[u % 3][q], acc, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    return acc;
  };

  f32x4 accT[NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) accT[ct] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int t = 0; t < NS - 1; ++t)
    if (t < NT) issue_tile(t);

  const int bo_off = (MODE == TB16_SELF) ? 3 * 64 * a.nchunk : 64 * a.nchunk;
  const int samp_q = i / a.T;

  for (int h = 0; h < a.nchunk; ++h) {
    f32x4 oT;      // this wave's 16 features of the chunk: [feature 16 wave + 4 g + r][token i]
    if constexpr (MODE == TB16_FF) {
      const unsigned char* s1 = acquire();
      oT = proj_T(s1, bias_s + 64 * h + 16 * wave);
#pragma unroll
      for (int r = 0; r < 4; ++r) oT[r] = gelu_tb16(oT[r]);
    } else {
      f32x4 qT, st[KTM], vT[KTM];
      int nkt = 1;
      if constexpr (MODE == TB16_SELF) {
        const unsigned char* sq = acquire();
        qT = proj_T(sq, bias_s + 64 * h + 16 * wave);
        const unsigned char* sk = acquire();
        const f32x4 kT = proj_T(sk, bias_s + 64 * (a.nchunk + h) + 16 * wave);
        const unsigned char* sv = acquire();
        vT[0] = proj_N(sv, bias_s + 64 * (2 * a.nchunk + h) + 16 * wave);
        f32x4 sp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) sp = MDT16_MFMA_F32(kT[s], qT[s], sp, 0, 0, 0);
        red[wave * 64 + lane] = sp;
      } else {
        const int nsamp = 16 / a.T, nkeys = nsamp * a.Tk;
        const int sample0 = row0 / a.T;
        nkt = (nkeys + 15) >> 4;
        float4 kk[KTM];
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt) {
          vT[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
          kk[kt] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (kt < nkt) {
            const int jl = kt * 16 + i;
            const int js = min(jl / a.Tk, nsamp - 1), jk = jl % a.Tk;
            kk[kt] = *reinterpret_cast<const float4*>(
                a.kv + ((int64_t)min(sample0 + js, a.nsamples - 1) * a.kv_bstride + jk) * a.ldkv + 64 * h + 16 * wave + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int jj = kt * 16 + 4 * g + r;
              const int vs = min(jj / a.Tk, nsamp - 1), vk = jj % a.Tk;
              const float vv = a.kv[((int64_t)min(sample0 + vs, a.nsamples - 1) * a.kv_bstride + vk) * a.ldkv +
                                    64 * a.nheads + 64 * h + 16 * wave + i];
              vT[kt][r] = jj < nkeys ? vv : 0.f;
            }
          }
        }
        const unsigned char* sq = acquire(5 * nkt);
        qT = proj_T(sq, bias_s + 64 * h + 16 * wave);
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt) {
          f32x4 sp = f32x4{0.f, 0.f, 0.f, 0.f};
          if (kt < nkt) {
            sp = MDT16_MFMA_F32(kk[kt].x, qT[0], sp, 0, 0, 0);
            sp = MDT16_MFMA_F32(kk[kt].y, qT[1], sp, 0, 0, 0);
            sp = MDT16_MFMA_F32(kk[kt].z, qT[2], sp, 0, 0, 0);
            sp = MDT16_MFMA_F32(kk[kt].w, qT[3], sp, 0, 0, 0);
            red[(kt * 4 + wave) * 64 + lane] = sp;
          }
        }
      }
      // ---- sum the four feature-partials of S^T (every wave ends up with the full tile) ----
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      float mx = -INFINITY;
      const int nkeys_c = (MODE == TB16_SELF) ? 16 : (16 / a.T) * a.Tk;
#pragma unroll
      for (int kt = 0; kt < KTM; ++kt) {
        st[kt] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        if (kt < nkt) {
          const f32x4 p0 = red[(kt * 4 + 0) * 64 + lane], p1 = red[(kt * 4 + 1) * 64 + lane];
          const f32x4 p2 = red[(kt * 4 + 2) * 64 + lane], p3 = red[(kt * 4 + 3) * 64 + lane];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int jj = kt * 16 + 4 * g + r;
            const int jsamp = (MODE == TB16_SELF) ? jj / a.T : jj / a.Tk;
            const bool ok = jj < nkeys_c && jsamp == samp_q;
            const float sv2 = ok ? ((p0[r] + p1[r]) + (p2[r] + p3[r])) * a.scale : -INFINITY;
            st[kt][r] = sv2;
            mx = fmaxf(mx, sv2);
          }
        }
      }
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
      oT = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < KTM; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) oT = MDT16_MFMA_F32(vT[kt][r], st[kt][r] / sum, oT, 0, 0, 0);
    }
    // ---- output projection: this wave's K-slice (its 16 features) of the chunk ----
    const unsigned char* so = acquire();
    float v8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v8[e] = ((e >> 2) == (wave & 1)) ? oT[e & 3] : 0.f;
    bf16x8 oh, ol;
    split8_16(v8, oh, ol);
    {
      const int sp = wave >> 1;
      constexpr int NU = NCT / 2;
      bf16x8 fh[3][2], fl[3][2];
      auto load = [&](int u, int set) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          fh[set][q] = fragO(so, 2 * u + q, sp, 0);
          fl[set][q] = fragO(so, 2 * u + q, sp, 1);
        }
      };
      load(0, 0);
      load(1, 1);
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        if (u + 2 < NU) load(u + 2, (u + 2) % 3);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 2; ++q) accT[2 * u + q] = MDT16_MFMA_BF16(fl[u % 3][q], oh, accT[2 * u + q], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 2; ++q) accT[2 * u + q] = MDT16_MFMA_BF16(fh[u % 3][q], ol, accT[2 * u + q], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 2; ++q) accT[2 * u + q] = MDT16_MFMA_BF16(fh[u % 3][q], oh, accT[2 * u + q], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // ---- sum the four waves' [C][16] accumulators through LDS (the ring is idle now), add bias + residual ----
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                       // every wave is done with the last weight slot
  f32x4* part = reinterpret_cast<f32x4*>(smem);       // [4 waves][NCT][64 lanes]
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) part[(wave * NCT + ct) * 64 + lane] = accT[ct];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (mvalid) {
    float* xo = a.x + (int64_t)m * a.ldx + 4 * g;
#pragma unroll
    for (int q = 0; q < NCT / 4; ++q) {
      const int ct = wave * (NCT / 4) + q;
      const f32x4 p0 = part[(0 * NCT + ct) * 64 + lane], p1 = part[(1 * NCT + ct) * 64 + lane];
      const f32x4 p2 = part[(2 * NCT + ct) * 64 + lane], p3 = part[(3 * NCT + ct) * 64 + lane];
      const float4 xr = *reinterpret_cast<const float4*>(xo + 16 * ct);
      const float4 bo = *reinterpret_cast<const float4*>(bias_s + bo_off + 16 * ct + 4 * g);
      *reinterpret_cast<float4*>(xo + 16 * ct) =
          make_float4(((p0[0] + p1[0]) + (p2[0] + p3[0])) + bo.x + xr.x, ((p0[1] + p1[1]) + (p2[1] + p3[1])) + bo.y + xr.y,
                      ((p0[2] + p1[2]) + (p2[2] + p3[2])) + bo.z + xr.z, ((p0[3] + p1[3]) + (p2[3] + p3[3])) + bo.w + xr.w);
    }
  }
}

template <int MODE, int C>
static hipError_t launch_tb16(const TBlockArgs& a, hipStream_t s) {
  constexpr int NS = (C == 128) ? 4 : 2;
  const size_t smem = (size_t)NS * 256 * C + (size_t)((a.nbias + 3) / 4) * 16 + 4 * 4 * 64 * 16;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tblock16<MODE, C>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    attr_set = true;
  }
  if (smem > 160 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL((k_tblock16<MODE, C>), dim3((unsigned)((a.M + 15) / 16)), dim3(256), smem, s, a);
  return hipGetLastError();
}

hipError_t launch_tblock16(const TBlockArgs& a, hipStream_t s) {
  if (a.M <= 0) return hipSuccess;
  if ((a.C != 128 && a.C != 256) || a.T <= 0 || 16 % a.T || a.nchunk <= 0 || a.post || a.kv2) return hipErrorInvalidValue;
  if (a.mode == TB16_CROSS && (a.Tk <= 0 || (16 / a.T) * a.Tk > 64)) return hipErrorInvalidValue;
#define MDT_TB16_CASE(MD)                                                         \
  case MD:                                                                        \
    return a.C == 128 ? launch_tb16<MD, 128>(a, s) : launch_tb16<MD, 256>(a, s);
  switch (a.mode) {
    MDT_TB16_CASE(TB16_SELF)
    MDT_TB16_CASE(TB16_CROSS)
    MDT_TB16_CASE(TB16_FF)
    default: return hipErrorInvalidValue;
  }
#undef MDT_TB16_CASE
}

}  // namespace mdt
