// Attention core of AttentionBase.forward (modules.py:350-363): one wave per (sample, head),
//   out[i, :] = softmax_j( (q_i . k_j) * scale ) @ V,          head dim 64, n, m <= 64.
//
// Both contractions run on the matrix cores in exact fp32 (v_mfma_f32_16x16x4_f32), entirely out of
// registers -- no LDS, no cross-lane transposes:
//   * S^T = K Q^T  (A = K rows, B = Q rows).  Lane (j = l&15, kq = l>>4) feeds K[j][16kq + s] at k-step s,
//     i.e. 16 CONTIGUOUS floats of its key row (and likewise for Q); any bijection of the 64 features onto
//     (step, lane-quarter) is valid as long as A and B use the same one.
//   * The 16x16 accumulator holds S^T[j = 4g + r][i = l&15] (g = l>>4, r = register).  Softmax over j for a
//     fixed query i is therefore per-lane over r and over g = lanes l, l^16, l^32, l^48: two shuffles.
//   * O^T = V^T P^T  (A = V^T, B = P^T).  With the k-mapping j = 4*kq + s the B operand of step s for lane
//     quarter kq is exactly that lane's own accumulator register r = s: the probabilities never move.
//   * O^T lands as [d = 16dt + 4g + r][i]: each lane stores 4 consecutive features (16 B) of its query row.
#include "mdt_kernels.h"

namespace mdt {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KTM>   // key tiles held in registers: 1 (Tk <= 16) or 4 (Tk <= 64)
__global__ __launch_bounds__(256) void k_attn(AttnArgs a) {
  constexpr int D = 64;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= a.batch * a.heads) return;
  const int b = wid / a.heads, h = wid % a.heads;
  const int lane = threadIdx.x & 63;
  const int lo = lane & 15, g = lane >> 4;
  const float* q = a.q + (int64_t)b * a.T * a.ldq + h * D;
  const float* k = a.k + (int64_t)b * a.kv_bstride * a.ldkv + h * D;
  const float* v = k + a.heads * D;
  float* o = a.out + (int64_t)b * a.T * a.ldo + h * D;
  const int KT = (a.Tk + 15) >> 4, QT = (a.T + 15) >> 4;

  for (int qt = 0; qt < QT; ++qt) {
    // B operand of S^T: this lane's 16 features of query row i
    const int i = qt * 16 + lo;
    float qr[16];
    {
      const float4* p = reinterpret_cast<const float4*>(q + (int64_t)(i < a.T ? i : 0) * a.ldq + 16 * g);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 t = p[c];
        qr[4 * c] = t.x; qr[4 * c + 1] = t.y; qr[4 * c + 2] = t.z; qr[4 * c + 3] = t.w;
      }
    }
    // K rows (A operand of S^T) and V columns (A operand of O^T) of every key tile are fetched up-front so
    // that all global loads of the head are in flight together.
    float kr[KTM][16], vr[KTM][16];
#pragma unroll
    for (int kt = 0; kt < KTM; ++kt) {
      if (kt < KT) {
        const int j = kt * 16 + lo;
        const float4* p = reinterpret_cast<const float4*>(k + (int64_t)(j < a.Tk ? j : 0) * a.ldkv + 16 * g);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float4 t = p[c];
          kr[kt][4 * c] = t.x; kr[kt][4 * c + 1] = t.y; kr[kt][4 * c + 2] = t.z; kr[kt][4 * c + 3] = t.w;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int jj = kt * 16 + 4 * g + s;           // key row this lane quarter feeds at PV step s
          const float* vrow = v + (int64_t)(jj < a.Tk ? jj : 0) * a.ldkv + lo;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) vr[kt][4 * s + dt] = jj < a.Tk ? vrow[16 * dt] : 0.f;
        }
      }
    }
    f32x4 st[KTM];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < KTM; ++kt) {
      st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kt < KT) {
        f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
        for (int s = 0; s < 16; s += 2) {
          s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[kt][s], qr[s], s0, 0, 0, 0);
          s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[kt][s + 1], qr[s + 1], s1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int jj = kt * 16 + 4 * g + r;          // key index of accumulator register r
          const float sv = jj < a.Tk ? (s0[r] + s1[r]) * a.scale : -INFINITY;
          st[kt][r] = sv;
          mx = fmaxf(mx, sv);
        }
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < KTM; ++kt) {
      if (kt < KT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = expf(st[kt][r] - mx);           // exp(-inf) = 0 for masked keys
          st[kt][r] = e;
          sum += e;
        }
      }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    f32x4 acc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < KTM; ++kt) {
      if (kt < KT) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float pr = st[kt][s] / sum;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
            acc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[kt][4 * s + dt], pr, acc[dt], 0, 0, 0);
        }
      }
    }
    if (i < a.T && a.out16) {
      unsigned short* o16 = reinterpret_cast<unsigned short*>(a.out) + (int64_t)b * a.T * a.ldo + h * D;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        unsigned short hh[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) hh[r] = __builtin_bit_cast(unsigned short, (__bf16)acc[dt][r]);
        *reinterpret_cast<uint2*>(o16 + (int64_t)i * a.ldo + 16 * dt + 4 * g) =
            make_uint2(hh[0] | ((unsigned)hh[1] << 16), hh[2] | ((unsigned)hh[3] << 16));
      }
    } else if (i < a.T) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        *reinterpret_cast<float4*>(o + (int64_t)i * a.ldo + 16 * dt + 4 * g) =
            make_float4(acc[dt][0], acc[dt][1], acc[dt][2], acc[dt][3]);
    }
  }
}

// The same operator for LONG sequences (more than 64 queries or keys per sample: the reference's default max_length = 1024 puts
// 256 tokens on the first attention level, generative.py:720-776): one wave per (sample, head, 16-query tile), the keys in
// chunks of 64 with a running maximum / sum (online softmax) -- the probabilities of a chunk are formed against the running
// maximum, the accumulated O^T and sum are rescaled when it moves; the result is divided by the sum once at the end.  Layouts,
// MFMA operand maps and exactness (fp32 MFMA, expf) as k_attn above; mathematically the same softmax, rounding differs from
// the two-pass form by a few ulp.
__global__ __launch_bounds__(256) void k_attn_long(AttnArgs a) {
  constexpr int D = 64, KTM = 4;
  const int QT = (a.T + 15) >> 4;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= (int64_t)a.batch * a.heads * QT) return;
  const int qt = (int)(wid % QT);
  const int bh = (int)(wid / QT);
  const int b = bh / a.heads, h = bh % a.heads;
  const int lane = threadIdx.x & 63;
  const int lo = lane & 15, g = lane >> 4;
  const float* q = a.q + (int64_t)b * a.T * a.ldq + h * D;
  const float* k = a.k + (int64_t)b * a.kv_bstride * a.ldkv + h * D;
  const float* v = k + a.heads * D;
  const int i = qt * 16 + lo;
  float qr[16];
  {
    const float4* p = reinterpret_cast<const float4*>(q + (int64_t)(i < a.T ? i : 0) * a.ldq + 16 * g);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4 t = p[c];
      qr[4 * c] = t.x; qr[4 * c + 1] = t.y; qr[4 * c + 2] = t.z; qr[4 * c + 3] = t.w;
    }
  }
  f32x4 acc[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  for (int k0 = 0; k0 < a.Tk; k0 += 16 * KTM) {
    float kr[KTM][16], vr[KTM][16];
#pragma unroll
    for (int kt = 0; kt < KTM; ++kt) {
      const int j = k0 + kt * 16 + lo;
      const float4* p = reinterpret_cast<const float4*>(k + (int64_t)(j < a.Tk ? j : 0) * a.ldkv + 16 * g);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 t = p[c];
        kr[kt][4 * c] = t.x; kr[kt][4 * c + 1] = t.y; kr[kt][4 * c + 2] = t.z; kr[kt][4 * c + 3] = t.w;
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int jj = k0 + kt * 16 + 4 * g + s;
        const float* vrow = v + (int64_t)(jj < a.Tk ? jj : 0) * a.ldkv + lo;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) vr[kt][4 * s + dt] = jj < a.Tk ? vrow[16 * dt] : 0.f;
      }
    }
    f32x4 st[KTM];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < KTM; ++kt) {
      f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
      for (int s = 0; s < 16; s += 2) {
        s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[kt][s], qr[s], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[kt][s + 1], qr[s + 1], s1, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int jj = k0 + kt * 16 + 4 * g + r;
        const float sv = jj < a.Tk ? (s0[r] + s1[r]) * a.scale : -INFINITY;
        st[kt][r] = sv;
        mx = fmaxf(mx, sv);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);                 // finite: every chunk holds at least one real key
    const float alpha = expf(m_run - m_new);              // exp(-inf) = 0 on the first chunk
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < KTM; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = expf(st[kt][r] - m_new);
        st[kt][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    l_run = l_run * alpha + sum;
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) acc[dt] *= alpha;
#pragma unroll
    for (int kt = 0; kt < KTM; ++kt)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          acc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[kt][4 * s + dt], st[kt][s], acc[dt], 0, 0, 0);
  }
  if (i >= a.T) return;
  const float inv = 1.0f / l_run;
  if (a.out16) {
    unsigned short* o16 = reinterpret_cast<unsigned short*>(a.out) + (int64_t)b * a.T * a.ldo + h * D;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      unsigned short hh[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) hh[r] = __builtin_bit_cast(unsigned short, (__bf16)(acc[dt][r] * inv));
      *reinterpret_cast<uint2*>(o16 + (int64_t)i * a.ldo + 16 * dt + 4 * g) =
          make_uint2(hh[0] | ((unsigned)hh[1] << 16), hh[2] | ((unsigned)hh[3] << 16));
    }
  } else {
    float* o = a.out + (int64_t)b * a.T * a.ldo + h * D;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      *reinterpret_cast<float4*>(o + (int64_t)i * a.ldo + 16 * dt + 4 * g) =
          make_float4(acc[dt][0] * inv, acc[dt][1] * inv, acc[dt][2] * inv, acc[dt][3] * inv);
  }
}

hipError_t launch_attn(const AttnArgs& a, hipStream_t s) {
  if (a.batch <= 0) return hipSuccess;
  if (a.T <= 0 || a.Tk <= 0 || a.T > 8192 || a.Tk > 8192 || a.ldq % 4 || a.ldkv % 4 || a.ldo % 4) return hipErrorInvalidValue;
  if (a.T > 64 || a.Tk > 64) {
    const int64_t w = (int64_t)a.batch * a.heads * ((a.T + 15) / 16);
    if ((w + 3) / 4 > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_attn_long, dim3((unsigned)((w + 3) / 4)), dim3(256), 0, s, a);
    return hipGetLastError();
  }
  const int waves = a.batch * a.heads;
  if (a.Tk <= 16)
    hipLaunchKernelGGL(k_attn<1>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(k_attn<4>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, a);
  return hipGetLastError();
}


// ------------------------------------------------------------------------------------------------------------------
// Cross-attention against the NORMALISED CONTEXT ITSELF (MDT_OP_ATTN_CTX): the per-layer key / value projections are
// folded into the query and output projections on the host (compiler.py::attention_layer_folded),
//   S_h = (LN(x) Mq_h^T) c^T,   out = sum_h (softmax(S_h) c) N_h^T,     c = (ctx - mean) / std  (no affine),
// so every head of every layer attends to the same Tk x 128 matrix c instead of its own hoisted K / V rows
// (QMDiffusionForward: 64 keys x 1024 floats per sample and layer = 1 GB per layer at B = 4096, the launch was HBM-bound).
// K = V = c is shared by the heads, hence the (token, head) pairs of a sample are simply ROWS of one problem:
//   Q' [R = T * heads rows][128]  x  c [Tk <= 64][128]   ->   out [R][128]
// one wave per (sample, 16-row tile of R); both contractions in exact fp32 MFMA out of registers.  The MFMA contraction
// index is free to permute, so both loads are laid out for coalescing instead of for the formula:
//   S^T = c Q'^T : step s = 4 cc + e of lane quarter g contracts feature 16 cc + 4 g + e -> one dwordx4 per cc, the four
//                  quarters of a row read 64 contiguous bytes;
//   O^T = c^T P^T: output row i of tile dt is feature 4 i + dt -> the A operand of the four tiles is ONE dwordx4 of key
//                  row 16 kt + 4 g + s at floats [64 half + 4 lo, +4), a full 256 B row segment per lane quarter, and
//                  lane (query, g) ends with out[query][64 half + 16 g + 4 r .. +4) per r = one dwordx4 store.
// ------------------------------------------------------------------------------------------------------------------
// RT row tiles of ONE sample per wave: the context rows (both operand layouts) are loaded once per wave, so with RT = 2 a
// sample with 17..32 (token, head) rows reads its 2 x 32 KB of context once instead of twice -- the launch is bound by that
// L2 -> CU traffic (4096 samples x 2 tiles x 72 KB = 590 MB per launch), not by its MFMAs.
template <int KT, int RT>
__global__ __launch_bounds__(256) void k_attn_ctx(AttnArgs a) {
  constexpr int D = 128;
  const int R = a.T * a.heads;                         // rows per sample
  const int QG = ((R + 15) / 16 + RT - 1) / RT;        // waves per sample
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= a.batch * QG) return;
  const int b = wid / QG, qg = wid % QG;
  const int lane = threadIdx.x & 63;
  const int lo = lane & 15, g = lane >> 4;
  const float* q = a.q + (int64_t)b * R * D;           // rows (token, head) are contiguous 128-float vectors
  const float* c = a.k + (int64_t)b * a.kv_bstride * a.ldkv;
  float* o = a.out + (int64_t)b * R * D;

  int i[RT];
  float qr[RT][32];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    i[t] = (qg * RT + t) * 16 + lo;
    const float4* p = reinterpret_cast<const float4*>(q + (int64_t)(i[t] < R ? i[t] : 0) * D + 4 * g);
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
      const float4 v = p[4 * cc];
      qr[t][4 * cc] = v.x; qr[t][4 * cc + 1] = v.y; qr[t][4 * cc + 2] = v.z; qr[t][4 * cc + 3] = v.w;
    }
  }
  f32x4 st[RT][KT];
  float mx[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) mx[t] = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int j = kt * 16 + lo;                        // A operand of S^T: key row j
    const float4* p = reinterpret_cast<const float4*>(c + (int64_t)(j < a.Tk ? j : 0) * a.ldkv + 4 * g);
    float kr[32];
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
      const float4 v = p[4 * cc];
      kr[4 * cc] = v.x; kr[4 * cc + 1] = v.y; kr[4 * cc + 2] = v.z; kr[4 * cc + 3] = v.w;
    }
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
      for (int s_ = 0; s_ < 32; s_ += 2) {
        s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[s_], qr[t][s_], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[s_ + 1], qr[t][s_ + 1], s1, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int jj = kt * 16 + 4 * g + r;
        const float sv = jj < a.Tk ? (s0[r] + s1[r]) * a.scale : -INFINITY;
        st[t][kt][r] = sv;
        mx[t] = fmaxf(mx[t], sv);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    mx[t] = fmaxf(mx[t], __shfl_xor(mx[t], 16, 64));
    mx[t] = fmaxf(mx[t], __shfl_xor(mx[t], 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = expf(st[t][kt][r] - mx[t]);
        st[t][kt][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) st[t][kt] *= inv;  // masked keys: exactly 0, times a finite (clamped) row below
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    f32x4 acc[RT][4];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) acc[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      float4 vr[4];
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) {
        const int jj = kt * 16 + 4 * g + s_;           // key row this lane quarter feeds at step s
        vr[s_] = *reinterpret_cast<const float4*>(c + (int64_t)(jj < a.Tk ? jj : 0) * a.ldkv + 64 * half + 4 * lo);
      }
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
          const float pr = st[t][kt][s_];
          acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[s_].x, pr, acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[s_].y, pr, acc[t][1], 0, 0, 0);
          acc[t][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[s_].z, pr, acc[t][2], 0, 0, 0);
          acc[t][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[s_].w, pr, acc[t][3], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < RT; ++t)
      if (i[t] < R) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          *reinterpret_cast<float4*>(o + (int64_t)i[t] * D + 64 * half + 16 * g + 4 * r) =
              make_float4(acc[t][0][r], acc[t][1][r], acc[t][2][r], acc[t][3][r]);
      }
  }
}

template <int RT>
static void launch_attn_ctx_rt(const AttnArgs& a, hipStream_t s) {
  const int qg = ((a.T * a.heads + 15) / 16 + RT - 1) / RT;
  const dim3 grid((unsigned)((a.batch * qg + 3) / 4)), block(256);
  switch ((a.Tk + 15) / 16) {
    case 1: hipLaunchKernelGGL((k_attn_ctx<1, RT>), grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL((k_attn_ctx<2, RT>), grid, block, 0, s, a); break;
    case 3: hipLaunchKernelGGL((k_attn_ctx<3, RT>), grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL((k_attn_ctx<4, RT>), grid, block, 0, s, a); break;
  }
}

hipError_t launch_attn_ctx(const AttnArgs& a, hipStream_t s) {
  if (a.batch <= 0) return hipSuccess;
  if (a.Tk <= 0 || a.Tk > 64 || a.ldkv % 4 || a.T <= 0 || a.heads <= 0) return hipErrorInvalidValue;
  if (a.T * a.heads > 16) launch_attn_ctx_rt<2>(a, s);     // two row tiles of a sample share the context loads
  else launch_attn_ctx_rt<1>(a, s);
  return hipGetLastError();
}

}  // namespace mdt
