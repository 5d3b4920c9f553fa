// Attention core of AttentionBase.forward (modules.py:350-363) for one (sample, head) per wave:
//   out[i, :] = softmax_j( (q_i . k_j) * scale ) @ V          head dim fixed at 64 = one lane per feature.
// The problem per head is tiny (n <= 64 queries, m <= 64 keys), so everything lives in registers/LDS:
// lane j keeps key row j in 64 VGPRs (scores are one dot product per lane), the softmax max/sum are
// wave-shuffle reductions across the 64 lanes, and for P@V lane d keeps column d of V in registers while
// the probabilities are broadcast with v_readlane.  K is staged through LDS with a 65-float row pitch so
// the row-per-lane read is bank-conflict free.
#include "mdt_kernels.h"

namespace mdt {

template <int TKM>
__global__ __launch_bounds__(64) void k_attn(AttnArgs a) {
  constexpr int D = 64;
  __shared__ float ks[TKM * (D + 1)];
  __shared__ __attribute__((aligned(16))) float qs[64 * D];
  const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
  const int lane = threadIdx.x;
  const float* q = a.q + (int64_t)b * a.T * a.ldq + h * D;
  const float* k = a.k + (int64_t)b * a.kv_bstride * a.ldkv + h * D;
  const float* v = k + a.heads * D;
  float* o = a.out + (int64_t)b * a.T * a.ldo + h * D;

  // V column `lane` in registers; K and Q through LDS (coalesced 256-B row reads).
  float vr[TKM];
#pragma unroll
  for (int j = 0; j < TKM; ++j) {
    vr[j] = 0.f;
    if (j < a.Tk) {
      vr[j] = v[(int64_t)j * a.ldkv + lane];
      ks[j * (D + 1) + lane] = k[(int64_t)j * a.ldkv + lane];
    }
  }
  for (int i = 0; i < a.T; ++i) qs[i * D + lane] = q[(int64_t)i * a.ldq + lane];
  __syncthreads();
  float kr[D];
  const int jrow = lane < a.Tk ? lane : 0;
#pragma unroll
  for (int d = 0; d < D; ++d) kr[d] = ks[jrow * (D + 1) + d];

  for (int i = 0; i < a.T; ++i) {
    float s = 0.f;
#pragma unroll
    for (int d4 = 0; d4 < D / 4; ++d4) {
      const float4 qv = *reinterpret_cast<const float4*>(&qs[i * D + d4 * 4]);
      s += qv.x * kr[d4 * 4] + qv.y * kr[d4 * 4 + 1] + qv.z * kr[d4 * 4 + 2] + qv.w * kr[d4 * 4 + 3];
    }
    s = lane < a.Tk ? s * a.scale : -INFINITY;
    float mx = s;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    const float e = lane < a.Tk ? expf(s - mx) : 0.f;
    float sum = e;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
    const float p = e / sum;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < TKM; ++j) {
      const float pj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p), j));
      acc += pj * vr[j];
    }
    o[(int64_t)i * a.ldo + lane] = acc;
  }
}

hipError_t launch_attn(const AttnArgs& a, hipStream_t s) {
  if (a.batch <= 0) return hipSuccess;
  if (a.T > 64 || a.Tk > 64) return hipErrorInvalidValue;
  dim3 grid((unsigned)(a.batch * a.heads)), block(64);
  if (a.Tk <= 16)
    hipLaunchKernelGGL((k_attn<16>), grid, block, 0, s, a);
  else
    hipLaunchKernelGGL((k_attn<64>), grid, block, 0, s, a);
  return hipGetLastError();
}

}  // namespace mdt
