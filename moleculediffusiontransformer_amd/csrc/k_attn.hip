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
    if (i < a.T) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        *reinterpret_cast<float4*>(o + (int64_t)i * a.ldo + 16 * dt + 4 * g) =
            make_float4(acc[dt][0], acc[dt][1], acc[dt][2], acc[dt][3]);
    }
  }
}

hipError_t launch_attn(const AttnArgs& a, hipStream_t s) {
  if (a.batch <= 0) return hipSuccess;
  if (a.T > 64 || a.Tk > 64 || a.ldq % 4 || a.ldkv % 4 || a.ldo % 4) return hipErrorInvalidValue;
  const int waves = a.batch * a.heads;
  if (a.Tk <= 16)
    hipLaunchKernelGGL(k_attn<1>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(k_attn<4>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace mdt
