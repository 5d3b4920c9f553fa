// Implicit-GEMM kernel for every dense contraction of the 1-D U-Net on gfx950 (MI355X).
//
//   out[m, n] = epilogue( sum_{tap, ci} prologue(A)[row(m, tap), ci] * W[n, tap*cin + ci] + bias[n] )
//
// A is a token-major activation (rows = (sample, position), channels contiguous), W is [N][K] with
// K = taps*cin contiguous (nn.Linear's native layout; convolutions are re-packed to [Cout][tap][Cin]).
// One instantiation serves nn.Linear, Conv1d k=1/k=3/k=9-stride-4 and the four output phases of
// ConvTranspose1d k=8 s=4 (reference: modules.py:40-81, :105-112, :135, :188, :317-319, :386-391, :486-516).
//
// Arithmetic is exact fp32 on the matrix cores: v_mfma_f32_32x32x2_f32 (64 FLOP/clk/SIMD, the fp32 peak
// of CDNA4; there is no TF32 on gfx950).  Workgroup = 4 waves, tile 128(M) x 64(N); wave w owns rows
// [32w, 32w+32) x 64 columns = two 32x32 accumulators.  A and W tiles are staged through registers into
// LDS (rows padded by 4 floats: conflict-free ds_read_b128 for the MFMA operands); the next K-chunk's
// global loads are issued before the current chunk's MFMAs.  The prologue (LayerNorm / GroupNorm-apply +
// FiLM + SiLU) runs on the A tile while it is staged, the epilogue (bias, exact GELU, residual) on the
// accumulators, so normalisation/activation tensors never round-trip through HBM.
#include <type_traits>

#include "mdt_kernels.h"

namespace mdt {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BN = 64;
constexpr int NTHREADS = 256;

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ float group16_sum(float v) {     // over the 16 lanes of a lane's group
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// BM = 128: wave w owns rows [32 w, 32 w + 32) x 64 columns (two accumulators).  BM = 64: waves as 2 x 2, rows 32 (w & 1) .., columns
// 32 (w >> 1) .. (one accumulator) -- twice the workgroups for the layers whose 128-row tiles leave half the CUs idle (M = 4096).
template <int PRO, int BK, int BM>
__global__ __launch_bounds__(NTHREADS) void k_gemm(GemmArgs g) {
  constexpr int LDT = BK + 4;           // padded LDS row (floats)
  constexpr int C4 = BK / 4;            // float4 per tile row
  constexpr int ROWSTEP = NTHREADS / C4;
  constexpr int RPT = BM / ROWSTEP;     // A rows staged per thread
  constexpr int WPT = BN / ROWSTEP;     // W rows staged per thread

  __shared__ __attribute__((aligned(16))) float As[BM * LDT];
  __shared__ __attribute__((aligned(16))) float Bs[BN * LDT];
  __shared__ float rstat[PRO == 1 ? BM * 2 : 2];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  gemm_select_phase(g);

  // XCD-aware tile order: blocks are dealt round-robin over the 8 XCDs, so give each XCD a contiguous
  // run of logical tiles (neighbouring N-tiles of one M-tile then share that XCD's L2 copy of A).
  const int nt = (g.N + BN - 1) / BN;
  int id;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int m0 = (id / nt) * BM;
  const int n0 = (id % nt) * BN;
  const int K = g.taps * g.cin;

  // ---- per-thread staging coordinates ----
  const int c4 = tid % C4;
  const int r0 = tid / C4;
  int rb[RPT], rs[RPT];
  bool rv[RPT];
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int m = m0 + r0 + i * ROWSTEP;
    rv[i] = m < g.M;
    const int b = rv[i] ? m / g.r_out : 0;
    rb[i] = b;
    rs[i] = m - b * g.r_out;
  }

  // ---- LayerNorm row statistics (two-pass, a 16-lane group per row, values held in registers) ----
  // NQ = 64-channel pieces of a row actually present: the U-Net's LayerNorm rows are 128 / 256 channels wide, and walking all
  // 32 pieces of the 2048-channel envelope for each of a wave's 32 rows cost more than the tile's MFMAs (pro = 1 GEMMs ran at a
  // third of the rate of the same shapes without a prologue)
  if constexpr (PRO == 1) {
    auto row_stats = [&](auto nqc) {
      constexpr int NQ = decltype(nqc)::value;
      // four rows per wave and turn: a 16-lane group per row, lane l of the group holds channels 64 q + 4 l .. + 3
      const int l16 = lane & 15;
      for (int it = 0; it < BM / 16; ++it) {
        const int row = wave * (BM / 4) + it * 4 + (lane >> 4);
        const int m = m0 + row;
        const bool valid = m < g.M;
        const int b = valid ? m / g.r_out : 0;
        const int src = valid ? (m - b * g.r_out) * g.t_stride + g.t_off : 0;
        const float* p = g.A + ((int64_t)b * g.r_in + src) * g.lda + g.a_col;
        float4 v[NQ];
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int e = q * 64 + l16 * 4;
          v[q] = (valid && e < g.cin) ? *reinterpret_cast<const float4*>(p + e) : make_float4(0.f, 0.f, 0.f, 0.f);
          s += (v[q].x + v[q].y) + (v[q].z + v[q].w);
        }
        const float mean = group16_sum(s) / (float)g.cin;
        float ss = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          if (q * 64 + l16 * 4 < g.cin) {
            const float dx = v[q].x - mean, dy = v[q].y - mean, dz = v[q].z - mean, dw = v[q].w - mean;
            ss += (dx * dx + dy * dy) + (dz * dz + dw * dw);
          }
        }
        const float var = group16_sum(ss) / (float)g.cin;
        if (l16 == 0) {
          rstat[row * 2] = valid ? mean : 0.f;
          rstat[row * 2 + 1] = valid ? 1.0f / sqrtf(var + g.eps) : 0.f;
        }
      }
    };
    if (g.cin <= 128) row_stats(std::integral_constant<int, 2>{});
    else if (g.cin <= 256) row_stats(std::integral_constant<int, 4>{});
    else if (g.cin <= 512) row_stats(std::integral_constant<int, 8>{});
    else row_stats(std::integral_constant<int, 32>{});
    __syncthreads();
  }

  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    acc0[i] = 0.f;
    acc1[i] = 0.f;
  }

  float4 ra[RPT], rw[WPT];
  bool va[RPT];

  auto load_chunk = [&](int kc) {
    const int k0 = kc * BK;
    const int tap = k0 / g.cin;
    const int ci = k0 - tap * g.cin + c4 * 4;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int src = rs[i] * g.t_stride + tap * g.t_dj + g.t_off;
      va[i] = rv[i] && src >= 0 && src < g.r_in;
      ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (va[i])
        ra[i] = *reinterpret_cast<const float4*>(g.A + ((int64_t)rb[i] * g.r_in + src) * g.lda + g.a_col + ci);
    }
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int n = n0 + r0 + i * ROWSTEP;
      rw[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < g.N) rw[i] = *reinterpret_cast<const float4*>(g.W + (int64_t)n * K + k0 + c4 * 4);
    }
  };

  auto store_chunk = [&](int kc) {
    const int k0 = kc * BK;
    const int tap = k0 / g.cin;
    const int ci = k0 - tap * g.cin + c4 * 4;  // first channel (within the normalised tensor) of this float4
    float4 gam, bet, fsc, fsh;
    if constexpr (PRO == 1 || PRO == 2) {
      gam = *reinterpret_cast<const float4*>(g.p0 + ci);
      bet = *reinterpret_cast<const float4*>(g.p1 + ci);
    }
    if constexpr (PRO == 2) {
      if (g.p3) {
        fsc = *reinterpret_cast<const float4*>(g.p3 + ci);
        fsh = *reinterpret_cast<const float4*>(g.p3 + g.cin + ci);
      }
    }
    int grp[4] = {0, 0, 0, 0};
    if constexpr (PRO == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) grp[e] = min((ci + e) / g.gsize, g.groups - 1);  // padded channels: gain 0
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      float x[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
      if (va[i]) {
        if constexpr (PRO == 1) {
          const int row = r0 + i * ROWSTEP;
          const float mean = rstat[row * 2], rstd = rstat[row * 2 + 1];
          const float ga[4] = {gam.x, gam.y, gam.z, gam.w}, be[4] = {bet.x, bet.y, bet.z, bet.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = (x[e] - mean) * rstd * ga[e] + be[e];
        } else if constexpr (PRO == 2) {
          const float ga[4] = {gam.x, gam.y, gam.z, gam.w}, be[4] = {bet.x, bet.y, bet.z, bet.w};
          const float* st = g.p2 + (int64_t)rb[i] * g.groups * 2;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float mean = st[grp[e] * 2], rstd = st[grp[e] * 2 + 1];
            const float sc = rstd * ga[e];
            x[e] = x[e] * sc + (be[e] - sc * mean);
          }
          if (g.p3) {
            const float a[4] = {fsc.x, fsc.y, fsc.z, fsc.w}, s[4] = {fsh.x, fsh.y, fsh.z, fsh.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = x[e] * (a[e] + 1.0f) + s[e];
          }
          if (g.pro_silu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = silu_f(x[e]);
          }
        } else if constexpr (PRO == 3) {
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = silu_f(x[e]);
        }
      }
      *reinterpret_cast<float4*>(&As[(r0 + i * ROWSTEP) * LDT + c4 * 4]) = make_float4(x[0], x[1], x[2], x[3]);
    }
#pragma unroll
    for (int i = 0; i < WPT; ++i)
      *reinterpret_cast<float4*>(&Bs[(r0 + i * ROWSTEP) * LDT + c4 * 4]) = rw[i];
  };

  const int nk = K / BK;
  const int li = lane & 31, lh = lane >> 5;
  load_chunk(0);
  for (int kc = 0; kc < nk; ++kc) {
    store_chunk(kc);
    __syncthreads();
    if (kc + 1 < nk) load_chunk(kc + 1);
    // MFMA k-slot mapping: the two lane halves take k = kb*8 + 4*lh + s for step s; A and W use the same
    // permutation of k, so the sum over the chunk is complete whatever the order.
#pragma unroll
    for (int kb = 0; kb < BK / 8; ++kb) {
      if constexpr (BM == 128) {
        const float4 a = *reinterpret_cast<const float4*>(&As[(wave * 32 + li) * LDT + kb * 8 + 4 * lh]);
        const float4 b0 = *reinterpret_cast<const float4*>(&Bs[li * LDT + kb * 8 + 4 * lh]);
        const float4 b1 = *reinterpret_cast<const float4*>(&Bs[(32 + li) * LDT + kb * 8 + 4 * lh]);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
      } else {
        const float4 a = *reinterpret_cast<const float4*>(&As[((wave & 1) * 32 + li) * LDT + kb * 8 + 4 * lh]);
        const float4 b0 = *reinterpret_cast<const float4*>(&Bs[((wave >> 1) * 32 + li) * LDT + kb * 8 + 4 * lh]);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) ----
  const int nA = n0 + (BM == 128 ? 0 : (wave >> 1) * 32) + li, nB = n0 + 32 + li;
  const float biasA = (g.bias && nA < g.N) ? g.bias[nA] : 0.f;
  const float biasB = (BM == 128 && g.bias && nB < g.N) ? g.bias[nB] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (BM == 128 ? wave : (wave & 1)) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    const int m = m0 + row;
    if (m >= g.M) continue;
    const int b = m / g.r_out;
    const int64_t orow = (int64_t)b * g.o_rows + (int64_t)(m - b * g.r_out) * g.o_stride + g.o_off;
    float vA = acc0[r] + biasA, vB = acc1[r] + biasB;
    if (g.act == 1) {
      vA = gelu_f(vA);
      if constexpr (BM == 128) vB = gelu_f(vB);
    }
    if (g.res) {
      if (nA < g.N) vA += g.res[orow * g.ldr + nA];
      if (BM == 128 && nB < g.N) vB += g.res[orow * g.ldr + nB];
    }
    if (nA < g.N) g.out[orow * g.ldc + g.o_col + nA] = vA;
    if (BM == 128 && nB < g.N) g.out[orow * g.ldc + g.o_col + nB] = vB;
  }
}

template <int PRO, int BM>
static hipError_t launch_pro_bm(const GemmArgs& g, hipStream_t s) {
  const int mt = (g.M + BM - 1) / BM, nt = (g.N + BN - 1) / BN;
  dim3 grid((unsigned)(mt * nt), 1, (unsigned)(g.phases > 1 ? g.phases : 1)), block(NTHREADS);
  // 64-deep chunks where the channels allow it: one chunk is then 4096 cycles of MFMA work per wave against a fixed ~4000 cycles
  // of staging (wait for the next chunk's loads, LDS writes, two barriers) with one workgroup per CU
  if (g.cin % 64 == 0 && g.taps * g.cin >= 128)
    hipLaunchKernelGGL((k_gemm<PRO, 64, BM>), grid, block, 0, s, g);
  else if (g.cin % 32 == 0)
    hipLaunchKernelGGL((k_gemm<PRO, 32, BM>), grid, block, 0, s, g);
  else
    hipLaunchKernelGGL((k_gemm<PRO, 16, BM>), grid, block, 0, s, g);
  return hipGetLastError();
}

template <int PRO>
static hipError_t launch_pro(const GemmArgs& g, hipStream_t s) {
  // 64-row tiles while 128-row tiles would give fewer than two workgroups per CU
  const int64_t wg128 = (int64_t)((g.M + 127) / 128) * ((g.N + BN - 1) / BN) * (g.phases > 1 ? g.phases : 1);
  return wg128 < 512 ? launch_pro_bm<PRO, 64>(g, s) : launch_pro_bm<PRO, 128>(g, s);
}

hipError_t launch_gemm(const GemmArgs& g, hipStream_t s) {
  if (g.M <= 0) return hipSuccess;
  switch (g.pro) {
    case 0: return launch_pro<0>(g, s);
    case 1: return launch_pro<1>(g, s);
    case 2: return launch_pro<2>(g, s);
    case 3: return launch_pro<3>(g, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace mdt
