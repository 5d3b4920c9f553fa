// MDT_OP_RESBLOCK: a whole ResnetBlock1d of the 64-token level (Patcher / Unpatcher of the U-Net, modules.py:145-205,
// 208-257) in one launch:
//     a1 = silu(GroupNorm_1group(x))                       h = conv_k3(a1) + b1
//     a2 = silu(GroupNorm_1group(h) (scale + 1) + shift)   y = conv_k3(a2) + b2 + to_out_1x1(x) + b_out
// for (CIN, COUT) = (16, 64) | (64, 16) | (16, 16); round 4: the channel counts are the PADDED ones (multiples of 16), the
// GroupNorm statistics run over the first cin_real / cout_real channels only (QMDiffusionForward's Patcher takes 2 real
// channels in 16, its Unpatcher leaves 1 in 16: padded gains, biases and weights are zero, so are the padded outputs).
// As separate launches (two GroupNorm-apply passes, three GEMMs) the five ops
// move 16 MB tensors through HBM between launches of 10-20 us each: 76 + 72 us per U-Net evaluation at batch 1024.
//
// One sample (64 tokens) per group of 4 waves, two samples in flight per workgroup (8 waves = 2 per SIMD: while one
// sample's waves sit in a reduction or a barrier the other's issue), persistent over the batch.  All weights (at most
// 72 KB as bf16 hi/lo fragments) are staged into LDS once per workgroup.  Activations go through LDS as bf16 hi/lo
// planes [token + 1][channel] with zero rows on both ends (the convolution's padding), row pitch 2 C + 16 bytes
// (conflict-free ds_read_b128 of 16 consecutive rows).  Convolutions run transposed, out^T[n][t] = W[n][k] act^T[k][t]:
// wave w owns tokens 16 w .. 16 w + 15 as MFMA columns, the accumulator of lane (i, g) holds 4 consecutive channels of
// token 16 w + i -- the layout GroupNorm + FiLM + SiLU + the 8-byte LDS write of the next operand want, and the float4
// the final store wants.  The k index of a 32-wide MFMA step enumerates (tap, channel) pairs; the host packs the weight
// fragments in exactly that order (compiler.py: UNetCompiler.resblock), zero where a step is padded.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "mdt_kernels.h"

namespace mdt {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define MDT_MFMA_BF16 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define MDT_MFMA_F32 __builtin_amdgcn_mfma_f32_16x16x4f32

constexpr int T = 64;                    // tokens per sample
constexpr int ROWS = T + 2;              // + one zero row on either end

__host__ __device__ constexpr int pitch_of(int c) { return 2 * c + 16; }      // bf16 planes
__host__ __device__ constexpr int pitch32_of(int c) { return 4 * c + 16; }    // fp32 planes (F32): conflict-free float4 reads of 16 rows
__host__ __device__ constexpr int ksteps_of(int c) { return c == 64 ? 6 : 2; }   // k = 3 convolution over c channels
__host__ __device__ constexpr int rsteps_of(int c) { return c == 64 ? 2 : 1; }   // 1 x 1 convolution over c channels

// +-16 / +-32 lane exchanges with the gfx950 permlane swaps (k_tblock_lw.hip), the rest of a wave sum with DPP moves
// inside the 16-lane row: no LDS round trip per step
#define MDT_XG(NAME, INSN)                                                               \
  __device__ __forceinline__ float NAME(float v) {                                       \
    float a = v, b = v;                                                                  \
    asm("s_nop 1\n\t" INSN " %0, %1" : "+v"(a), "+v"(b));                                \
    return a + b;                                                                        \
  }
MDT_XG(xg16_add, "v_permlane16_swap_b32")
MDT_XG(xg32_add, "v_permlane32_swap_b32")
#undef MDT_XG

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  const int m = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true);
  return v + __builtin_bit_cast(float, m);
}

__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add<0xB1>(v);      // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);      // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);     // row_half_mirror (quads are uniform: adds the lanes xor 4 would)
  v = dpp_add<0x140>(v);     // row_mirror
  v = xg16_add(v);
  return xg32_add(v);
}

__device__ __forceinline__ void split4(const float v[4], bf16x4& hi, bf16x4& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const __bf16 h = (__bf16)v[e];
    hi[e] = h;
    lo[e] = (__bf16)(v[e] - (float)h);
  }
}

__device__ __forceinline__ float silu(float t) { return t * __builtin_amdgcn_rcpf(1.0f + __expf(-t)); }

// byte offset (inside a plane) of the 16-byte B fragment of lane (i, g) for k-step s of a k = 3 convolution over C
// channels per tap; tokens 16 w + i.  Row r of a plane is token r - 1; row 0 is zero.
// F32: one fp32 plane instead of the hi / lo pair; the lane's 8 k-slots are 32 bytes = two float4 (slots 0..3 | 4..7)
template <int C, bool F32>
__device__ __forceinline__ int conv_frag_off(int s, int w, int i, int g) {
  constexpr int P = F32 ? pitch32_of(C) : pitch_of(C), E = F32 ? 4 : 2;
  if constexpr (C == 64) {
    const int tap = s >> 1;
    return (16 * w + i + tap) * P + (32 * (s & 1) + 8 * g) * E;
  } else {
    const int tap = 2 * s + (g >> 1);                // step 0: taps 0 | 1, step 1: tap 2 | nothing
    return (tap < 3 ? (16 * w + i + tap) * P : 0) + 8 * (g & 1) * E;
  }
}
// ... of the 1 x 1 convolution (centre tap) over C channels
template <int C, bool F32>
__device__ __forceinline__ int res_frag_off(int s, int w, int i, int g) {
  constexpr int P = F32 ? pitch32_of(C) : pitch_of(C), E = F32 ? 4 : 2;
  if constexpr (C == 64) return (16 * w + i + 1) * P + (32 * s + 8 * g) * E;
  else return (g < 2 ? (16 * w + i + 1) * P : 0) + 8 * (g & 1) * E;
}

}  // namespace

// F32: the weights are fp32 MFMA fragments [step][row tile][half][64 lanes][4] (lane (i, g) float r of half lo = W[16 rt + i][the
// step's pair 8 g + 4 lo + r]), the activation planes fp32, every product an exact fp32 MFMA (v_mfma_f32_16x16x4_f32: slot
// (g, 4 lo + r) of a 32-wide bf16 step is contraction index g of MFMA (lo, r)) -- the reference's arithmetic (modules.py:105-112)
template <int CIN, int COUT, bool F32>
__global__ __launch_bounds__(512) void k_resblock(ResBlockArgs a) {
  constexpr int RT = COUT / 16;                          // 16-channel row tiles of both convolutions' outputs
  constexpr int K1 = ksteps_of(CIN);                     // conv1 k-steps
  constexpr int K2 = ksteps_of(COUT);                    // conv2 k-steps ...
  constexpr int KR = rsteps_of(CIN);                     // ... followed by the 1 x 1 to_out steps on the raw input
  constexpr int WBYTES = (K1 + K2 + KR) * RT * 2 * 1024; // fragment tiles: [step][row tile][hi | lo][64 lanes][16 B]
  constexpr int PIN = F32 ? pitch32_of(CIN) : pitch_of(CIN), POUT = F32 ? pitch32_of(COUT) : pitch_of(COUT);
  constexpr int PL_IN = ROWS * PIN, PL_OUT = ROWS * POUT;   // bytes per plane
  constexpr int NPL = F32 ? 1 : 2;                          // planes per tensor
  constexpr int SAMPLE = 2 * NPL * PL_IN + NPL * PL_OUT;    // x hi | x lo | a1 hi | a1 lo | a2 hi | a2 lo   (F32: x | a1 | a2)
  constexpr int NV = T * CIN / 4 / 256;                     // float4 per thread of one sample's input
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* wl = smem;
  float* red = reinterpret_cast<float*>(smem + WBYTES + 2 * SAMPLE);   // [half][GroupNorm 0 | 1][wave 0..3] (mean, M2)

  const int tid = threadIdx.x, lane = tid & 63, t = tid & 255;
  const int half = tid >> 8, w = (tid >> 6) & 3;
  const int i = lane & 15, g = lane >> 4;
  unsigned char* sm = smem + WBYTES + half * SAMPLE;
  unsigned char* xh = sm, *xl = sm + (NPL - 1) * PL_IN, *a1h = sm + NPL * PL_IN, *a1l = sm + (2 * NPL - 1) * PL_IN;
  unsigned char* a2h = sm + 2 * NPL * PL_IN, *a2l = sm + 2 * NPL * PL_IN + (NPL - 1) * PL_OUT;     // (F32: the "l" pointers are unused)

  // ---- once per workgroup: weights -> LDS, zero rows, per-channel vectors -> registers ----
  {
    // all requests first: as a rolled load -> wait -> store loop the staging was 9 round trips in a row
    const float4* src = reinterpret_cast<const float4*>(a.w);
    float4* dst = reinterpret_cast<float4*>(wl);
    constexpr int NW = (WBYTES / 16 + 511) / 512;
    float4 wv[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) wv[k] = src[(tid + 512 * k < WBYTES / 16) ? tid + 512 * k : 0];
#pragma unroll
    for (int k = 0; k < NW; ++k)
      if (tid + 512 * k < WBYTES / 16) dst[tid + 512 * k] = wv[k];
    // rows 0 and ROWS - 1 of the six planes of this half
    for (int k = t; k < 2 * NPL * (PIN / 4) ; k += 256) {
      const int pl = k / (PIN / 4), o = k % (PIN / 4);
      reinterpret_cast<float*>(sm + pl * PL_IN)[o] = 0.f;
      reinterpret_cast<float*>(sm + pl * PL_IN + (ROWS - 1) * PIN)[o] = 0.f;
    }
    for (int k = t; k < NPL * (POUT / 4); k += 256) {
      const int pl = k / (POUT / 4), o = k % (POUT / 4);
      reinterpret_cast<float*>(sm + 2 * NPL * PL_IN + pl * PL_OUT)[o] = 0.f;
      reinterpret_cast<float*>(sm + 2 * NPL * PL_IN + pl * PL_OUT + (ROWS - 1) * POUT)[o] = 0.f;
    }
  }
  // vec = gamma1[CIN] | beta1[CIN] | b1[COUT] | gamma2[COUT] | beta2[COUT] | bout[COUT]  (bout = b2 + to_out bias)
  const int c4 = t % (CIN / 4);                           // the thread's input float4 column (the same for all its rows)
  const float4 g1 = *reinterpret_cast<const float4*>(a.vec + 4 * c4);
  const float4 be1 = *reinterpret_cast<const float4*>(a.vec + CIN + 4 * c4);
  float4 b1[RT], fa[RT], fb[RT], bo[RT];
  {
    float4 g2[RT], be2[RT], fs[RT], fh[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int c = 16 * rt + 4 * g;
      b1[rt] = *reinterpret_cast<const float4*>(a.vec + 2 * CIN + c);
      g2[rt] = *reinterpret_cast<const float4*>(a.vec + 2 * CIN + COUT + c);
      be2[rt] = *reinterpret_cast<const float4*>(a.vec + 2 * CIN + 2 * COUT + c);
      bo[rt] = *reinterpret_cast<const float4*>(a.vec + 2 * CIN + 3 * COUT + c);
      fs[rt] = make_float4(0.f, 0.f, 0.f, 0.f);
      fh[rt] = fs[rt];
    }
    if (a.film) {                                          // one block for all row tiles: one round trip
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        fs[rt] = *reinterpret_cast<const float4*>(a.film + 16 * rt + 4 * g);
        fh[rt] = *reinterpret_cast<const float4*>(a.film + a.film_ld + 16 * rt + 4 * g);
      }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      // (n gamma + beta) (scale + 1) + shift = n [gamma (scale + 1)] + [beta (scale + 1) + shift]
      fa[rt] = make_float4(g2[rt].x * (fs[rt].x + 1.0f), g2[rt].y * (fs[rt].y + 1.0f), g2[rt].z * (fs[rt].z + 1.0f),
                           g2[rt].w * (fs[rt].w + 1.0f));
      fb[rt] = make_float4(be2[rt].x * (fs[rt].x + 1.0f) + fh[rt].x, be2[rt].y * (fs[rt].y + 1.0f) + fh[rt].y,
                           be2[rt].z * (fs[rt].z + 1.0f) + fh[rt].z, be2[rt].w * (fs[rt].w + 1.0f) + fh[rt].w);
    }
  }
  // fragment addresses
  int o1[K1], o2[K2], orr[KR];
#pragma unroll
  for (int s = 0; s < K1; ++s) o1[s] = conv_frag_off<CIN, F32>(s, w, i, g);
#pragma unroll
  for (int s = 0; s < K2; ++s) o2[s] = conv_frag_off<COUT, F32>(s, w, i, g);
#pragma unroll
  for (int s = 0; s < KR; ++s) orr[s] = res_frag_off<CIN, F32>(s, w, i, g);
  const unsigned char* wlane = wl + lane * 16;

  // mean and 1 / sqrt(var + eps) over the sample (4 waves x 64 lanes x N values): two-pass inside the wave, then the
  // waves' (mean, M2) pairs merged through LDS with ONE barrier (Chan et al.: M2 = sum M2_k + n sum (mean_k - mean)^2)
  // (`real(k)`: the value belongs to a real channel; `cnt`: real values per wave -- the same for the four waves of a sample)
  auto sample_stats = [&](auto&& value, auto&& real, auto nc, float cnt, int slot, float& mean, float& rstd) {
    constexpr int N = decltype(nc)::value;
    const float icnt = 1.0f / cnt;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < N; ++k) s += real(k) ? value(k) : 0.f;
    const float mw = wave_sum(s) * icnt;
    float m2 = 0.f;
#pragma unroll
    for (int k = 0; k < N; ++k) { const float d = real(k) ? value(k) - mw : 0.f; m2 += d * d; }
    m2 = wave_sum(m2);
    if (lane == 0) reinterpret_cast<float2*>(red)[(half * 2 + slot) * 4 + w] = make_float2(mw, m2);
    __syncthreads();
    const float4 p01 = reinterpret_cast<const float4*>(red)[(half * 2 + slot) * 2];
    const float4 p23 = reinterpret_cast<const float4*>(red)[(half * 2 + slot) * 2 + 1];
    mean = 0.25f * ((p01.x + p01.z) + (p23.x + p23.z));
    const float d0 = p01.x - mean, d1 = p01.z - mean, d2 = p23.x - mean, d3 = p23.z - mean;
    const float M2 = ((p01.y + p01.w) + (p23.y + p23.w)) + cnt * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
    rstd = __builtin_amdgcn_rsqf(M2 * (0.25f * icnt) + a.eps);
  };
  const float cnt1 = 16.0f * (float)a.cin_real, cnt2 = 16.0f * (float)a.cout_real;   // a wave holds 16 tokens of its sample

  __syncthreads();
  const int step = 2 * gridDim.x;
  const int niter = (a.B - 2 * (int)blockIdx.x + step - 1) / step;      // the same for both halves (barriers)
  float4 xn[NV];                                     // the next pass's input rows
  auto request_rows = [&](int it) {
    const int b = 2 * blockIdx.x + it * step + half;
    const float* xb = a.x + (int64_t)(b < a.B ? b : a.B - 1) * T * CIN;
    if (a.patch_in > 1) {
      // the input is still PATCHED (MDT_K_PATCH_IN = p: [T / p][CIN p], y[l][c p + q] = x[l p + q][c]; the Unpatcher's rearrange,
      // modules.py / a_unet Patcher): the four channels of a float4 are p floats apart -- four 4-byte loads instead of a launch
      const int p_ = a.patch_in;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int e = t + 256 * j, tok = e / (CIN / 4), cc = 4 * (e % (CIN / 4));
        const float* s_ = xb + (tok / p_) * (p_ * CIN) + cc * p_ + tok % p_;
        xn[j] = make_float4(s_[0], s_[p_], s_[2 * p_], s_[3 * p_]);
      }
    } else {
      const float4* xp = reinterpret_cast<const float4*>(xb);
#pragma unroll
      for (int j = 0; j < NV; ++j) xn[j] = xp[t + 256 * j];
    }
  };
  request_rows(0);
  for (int it = 0; it < niter; ++it) {
    const int b = 2 * blockIdx.x + it * step + half;
    const bool live = b < a.B;
    // ---- input rows (requested during the previous pass), GroupNorm 1 (one group: the whole sample), SiLU, split ----
    float xv[NV][4];
#pragma unroll
    for (int j = 0; j < NV; ++j) { xv[j][0] = xn[j].x; xv[j][1] = xn[j].y; xv[j][2] = xn[j].z; xv[j][3] = xn[j].w; }
    float mean1, rstd1;
    sample_stats([&](int k) { return xv[k >> 2][k & 3]; }, [&](int k) { return 4 * c4 + (k & 3) < a.cin_real; },
                 std::integral_constant<int, 4 * NV>{}, cnt1, 0, mean1, rstd1);
    {
      const float ga[4] = {g1.x, g1.y, g1.z, g1.w}, be[4] = {be1.x, be1.y, be1.z, be1.w};
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int tok = (t + 256 * j) / (CIN / 4);
        float av[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) av[e] = silu((xv[j][e] - mean1) * rstd1 * ga[e] + be[e]);
        if constexpr (F32) {
          *reinterpret_cast<f32x4*>(xh + (tok + 1) * PIN + 16 * c4) = f32x4{xv[j][0], xv[j][1], xv[j][2], xv[j][3]};
          *reinterpret_cast<f32x4*>(a1h + (tok + 1) * PIN + 16 * c4) = f32x4{av[0], av[1], av[2], av[3]};
        } else {
          bf16x4 h, l;
          split4(xv[j], h, l);
          *reinterpret_cast<bf16x4*>(xh + (tok + 1) * PIN + 8 * c4) = h;
          *reinterpret_cast<bf16x4*>(xl + (tok + 1) * PIN + 8 * c4) = l;
          split4(av, h, l);
          *reinterpret_cast<bf16x4*>(a1h + (tok + 1) * PIN + 8 * c4) = h;
          *reinterpret_cast<bf16x4*>(a1l + (tok + 1) * PIN + 8 * c4) = l;
        }
      }
    }
    if (it + 1 < niter) request_rows(it + 1);        // lands under the two convolutions
    __syncthreads();
    // ---- conv1 (transposed): h^T[16 rt + 4 g + r][16 w + i] ----
    f32x4 acc[RT], acc2[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) { acc[rt] = f32x4{b1[rt].x, b1[rt].y, b1[rt].z, b1[rt].w}; acc2[rt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    // one 32-wide k-step: B operand at byte offset `off` of the plane pair (ph, pl), weight fragments of step `ws`
    auto kstep = [&](const unsigned char* ph, const unsigned char* pl, int off, int ws) __attribute__((always_inline)) {
      if constexpr (F32) {
        // exact fp32: the lane's 8 k-slots are two float4 of ONE plane; (half, r) = one 16x16x4 MFMA per row tile.  Two partial
        // accumulators per row tile (even / odd r) so that no MFMA waits for the one issued just before it (RT = 1: the whole
        // convolution would otherwise be one dependent chain)
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(ph + off), b1 = *reinterpret_cast<const f32x4*>(ph + off + 16);
        f32x4 w0[RT], w1[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          w0[rt] = *reinterpret_cast<const f32x4*>(wlane + ((ws * RT + rt) * 2 + 0) * 1024);
          w1[rt] = *reinterpret_cast<const f32x4*>(wlane + ((ws * RT + rt) * 2 + 1) * 1024);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            if (r & 1) acc2[rt] = MDT_MFMA_F32(w0[rt][r], b0[r], acc2[rt], 0, 0, 0);
            else acc[rt] = MDT_MFMA_F32(w0[rt][r], b0[r], acc[rt], 0, 0, 0);
          }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            if (r & 1) acc2[rt] = MDT_MFMA_F32(w1[rt][r], b1[r], acc2[rt], 0, 0, 0);
            else acc[rt] = MDT_MFMA_F32(w1[rt][r], b1[r], acc[rt], 0, 0, 0);
          }
      } else {
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(ph + off);
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(pl + off);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const bf16x8 wh = *reinterpret_cast<const bf16x8*>(wlane + ((ws * RT + rt) * 2 + 0) * 1024);
          const bf16x8 wlo = *reinterpret_cast<const bf16x8*>(wlane + ((ws * RT + rt) * 2 + 1) * 1024);
          acc[rt] = MDT_MFMA_BF16(wlo, bh, acc[rt], 0, 0, 0);
          acc[rt] = MDT_MFMA_BF16(wh, bl, acc[rt], 0, 0, 0);
          acc[rt] = MDT_MFMA_BF16(wh, bh, acc[rt], 0, 0, 0);
        }
      }
    };
    auto fold_acc2 = [&]() __attribute__((always_inline)) {
      if constexpr (F32) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) { acc[rt] += acc2[rt]; acc2[rt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      }
    };
#pragma unroll
    for (int s1 = 0; s1 < K1; ++s1) kstep(a1h, a1l, o1[s1], s1);
    fold_acc2();
    // ---- GroupNorm 2 + FiLM + SiLU on the accumulators, split -> a2 planes ----
    float mean2, rstd2;
    sample_stats([&](int k) { return acc[k >> 2][k & 3]; }, [&](int k) { return 16 * (k >> 2) + 4 * g + (k & 3) < a.cout_real; },
                 std::integral_constant<int, 4 * RT>{}, cnt2, 1, mean2, rstd2);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const float fav[4] = {fa[rt].x, fa[rt].y, fa[rt].z, fa[rt].w}, fbv[4] = {fb[rt].x, fb[rt].y, fb[rt].z, fb[rt].w};
      float av[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) av[r] = silu((acc[rt][r] - mean2) * rstd2 * fav[r] + fbv[r]);
      if constexpr (F32) {
        *reinterpret_cast<f32x4*>(a2h + (16 * w + i + 1) * POUT + (16 * rt + 4 * g) * 4) = f32x4{av[0], av[1], av[2], av[3]};
      } else {
        bf16x4 h, l;
        split4(av, h, l);
        *reinterpret_cast<bf16x4*>(a2h + (16 * w + i + 1) * POUT + (16 * rt + 4 * g) * 2) = h;
        *reinterpret_cast<bf16x4*>(a2l + (16 * w + i + 1) * POUT + (16 * rt + 4 * g) * 2) = l;
      }
    }
    __syncthreads();
    // ---- conv2 + to_out (1 x 1 on the raw input) ----
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{bo[rt].x, bo[rt].y, bo[rt].z, bo[rt].w};
#pragma unroll
    for (int s2 = 0; s2 < K2 + KR; ++s2) {
      if (s2 < K2) kstep(a2h, a2l, o2[s2 < K2 ? s2 : 0], K1 + s2);
      else kstep(xh, xl, orr[s2 < K2 ? 0 : s2 - K2], K1 + s2);
    }
    fold_acc2();
    if (live && a.patch_out > 1) {
      // the output goes out PATCHED (MDT_K_PATCH_OUT = p: [T / p][COUT p], the Patcher's rearrange): token 16 w + i, channel c at
      // [(token / p)][c p + token % p]
      const int p_ = a.patch_out, tok = 16 * w + i;
      float* yo = a.out + (int64_t)b * T * COUT + (tok / p_) * (p_ * COUT) + (4 * g) * p_ + tok % p_;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) yo[(16 * rt + r) * p_] = acc[rt][r];
    } else if (live) {
      float* yo = a.out + ((int64_t)b * T + 16 * w + i) * COUT + 4 * g;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        __builtin_nontemporal_store(acc[rt], reinterpret_cast<f32x4*>(yo + 16 * rt));   // streaming: consumed by the next launch
    }
    // the next iteration's plane writes come after its two reduction barriers: every wave is past its reads by then
  }
}

bool resblock_supported(int T_, int cin, int cout) {
  return T_ == T && ((cin == 16 && cout == 64) || (cin == 64 && cout == 16) || (cin == 16 && cout == 16));
}

template <int CIN, int COUT, bool F32>
static hipError_t launch_rb(const ResBlockArgs& a, hipStream_t s) {
  constexpr int RT = COUT / 16;
  constexpr int WBYTES = (ksteps_of(CIN) + ksteps_of(COUT) + rsteps_of(CIN)) * RT * 2 * 1024;
  constexpr int SAMPLE = F32 ? 2 * ROWS * pitch32_of(CIN) + ROWS * pitch32_of(COUT) : 4 * ROWS * pitch_of(CIN) + 2 * ROWS * pitch_of(COUT);
  const size_t smem = (size_t)WBYTES + 2 * SAMPLE + 2 * 4 * 4 * sizeof(float);
  static DevOnce attr_once;                          // per device (mdt_kernels.h)
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_resblock<CIN, COUT, F32>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(160 * 1024));
  }
  const int pairs = (a.B + 1) / 2;
  hipLaunchKernelGGL((k_resblock<CIN, COUT, F32>), dim3((unsigned)(pairs < 256 ? pairs : 256)), dim3(512), smem, s, a);
  return hipGetLastError();
}

hipError_t launch_resblock(const ResBlockArgs& a, hipStream_t s) {
  if (a.B <= 0) return hipSuccess;
  if (!resblock_supported(a.T, a.cin, a.cout)) return hipErrorInvalidValue;
  if (a.cin_real <= 0 || a.cin_real > a.cin || a.cout_real <= 0 || a.cout_real > a.cout) return hipErrorInvalidValue;
  if (a.cin == 16 && a.cout == 16) return a.wf32 ? launch_rb<16, 16, true>(a, s) : launch_rb<16, 16, false>(a, s);
  if (a.wf32) return a.cin == 16 ? launch_rb<16, 64, true>(a, s) : launch_rb<64, 16, true>(a, s);
  return a.cin == 16 ? launch_rb<16, 64, false>(a, s) : launch_rb<64, 16, false>(a, s);
}

}  // namespace mdt
