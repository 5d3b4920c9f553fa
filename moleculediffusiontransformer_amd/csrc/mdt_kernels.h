// Internal launch interface between the C ABI / program executor (mdt_api.cpp)
// and the gfx950 kernels (*.hip).  Not part of the public ABI (include/mdt_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mdt {

struct GemmArgs {
  const float* A;
  const float* W;     // fp32 [N][K], or the bf16 hi plane when W_lo != nullptr
  const float* W_lo;  // bf16 lo plane [N][K] (split-bf16 GEMM) or nullptr (exact fp32 MFMA)
  const float* bias;
  float* out;
  const float* res;
  const float* p0;  // gain
  const float* p1;  // bias of the norm
  const float* p2;  // GroupNorm stats [batch][G][2] = (mean, rstd)
  const float* p3;  // FiLM [scale(cin) | shift(cin)]
  int M, r_out, r_in, lda, cin, taps, t_stride, t_dj, t_off;
  int N, ldc, o_rows, o_stride, o_off, ldr;
  int pro, groups, gsize, pro_silu, act, a_col, o_col;
  float eps;
};
hipError_t launch_gemm(const GemmArgs& g, hipStream_t s);          // exact fp32 MFMA (k_gemm.hip)
hipError_t launch_gemm_bf16x3(const GemmArgs& g, hipStream_t s);   // split-bf16 MFMA (k_gemm_bf16x3.hip)

struct GnStatsArgs {
  const float* x;
  float* stats;  // [batch][G][2]
  int batch, rows, ld, groups, gsize;
  float eps;
};
hipError_t launch_gn_stats(const GnStatsArgs& g, hipStream_t s);

struct AttnArgs {
  const float* q;
  const float* k;  // v = k + heads*64
  float* out;
  int batch, T, Tk, heads, ldq, ldkv, ldo, kv_bstride;
  float scale;
};
hipError_t launch_attn(const AttnArgs& a, hipStream_t s);

hipError_t launch_concat(const float* a, const float* b, float* out, int64_t rows, int ca, int cb, float scale_b,
                         hipStream_t s);
hipError_t launch_patch(const float* in, float* out, int batch, int rows_in, int c_in, int ld_in, int ld_out,
                        int patch, int inverse, hipStream_t s);
hipError_t launch_time_embed(const float* cn, const float* w, float* out, int rows, int half, int ld,
                             hipStream_t s);

}  // namespace mdt
