// HBM-bound kernels of the sampling path on gfx950: k-diffusion preconditioning, the ADPM2 sampler update,
// noise generation, conditioning/time embeddings and the layout shuffles around the U-Net.
//
// Built with -ffp-contract=off: the sampler arithmetic keeps the reference's separate fp32 multiplies and
// adds (diffusion.py:502-515, :810-814) instead of fused multiply-adds.
//
// Sampler state, noise and results are (B, C, L) channel-major as in the reference; the U-Net consumes and
// produces token-major (B, L, Cp) tiles.  One workgroup handles one sample: the (L x Cp) tile is transposed
// through LDS so that both the channel-major and the token-major side are read and written as 16-byte
// coalesced accesses.
#include "mdt_kernels.h"
#include "../../include/mdt_hip.h"

namespace mdt {

// ------------------------------------------------------------------------------------------------
// Counter-based normal generator: Philox4x32-10 (Salmon et al., SC'11) + Box-Muller.
// One counter per 4 consecutive elements of the flattened GLOBAL (sample, channel, position) index, so
// the stream a sample sees does not depend on how the batch is sharded over GPUs.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float4 normal4(uint64_t seed, uint32_t step, uint64_t quad_index) {
  uint32_t r[4];
  philox4x32_10((uint32_t)quad_index, (uint32_t)(quad_index >> 32), step, 0u, (uint32_t)seed,
                (uint32_t)(seed >> 32), r);
  const float k = 2.3283064365386963e-10f;  // 2^-32
  const float u0 = ((float)r[0] + 1.0f) * k, u1 = (float)r[1] * k;   // u0 in (0, 1]
  const float u2 = ((float)r[2] + 1.0f) * k, u3 = (float)r[3] * k;
  const float ra = sqrtf(-2.0f * logf(u0)), rb = sqrtf(-2.0f * logf(u2));
  float sa, ca, sb, cb;
  sincosf(6.283185307179586f * u1, &sa, &ca);
  sincosf(6.283185307179586f * u3, &sb, &cb);
  return make_float4(ra * ca, ra * sa, rb * cb, rb * sb);
}

__device__ __forceinline__ float clamp1(float v) { return fminf(fmaxf(v, -1.0f), 1.0f); }
// clip() of the denoised value (diffusion.py:75-88): static clamp to [-1, 1], or -- dynamic thresholding, s = the sample's
// max(quantile(|x_denoised|, q), 1) from k_dyn_scale -- clamp to [-s, s] and divide by s
__device__ __forceinline__ float clip_dyn(float v, float s) { return s > 0.f ? fminf(fmaxf(v, -s), s) / s : clamp1(v); }

// Token-major tile <-> LDS helpers (tile pitch Cp + 1 floats).
__device__ __forceinline__ void tile_load(float* tile, const float* src, int L, int Cp) {
  const int n4 = L * Cp / 4, c4n = Cp / 4;
  for (int i = threadIdx.x; i < n4; i += blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(src)[i];
    const int l = i / c4n, c = (i - l * c4n) * 4;
    float* t = tile + l * (Cp + 1) + c;
    t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
  }
}
__device__ __forceinline__ void tile_store(const float* tile, float* dst, int L, int Cp) {
  const int n4 = L * Cp / 4, c4n = Cp / 4;
  for (int i = threadIdx.x; i < n4; i += blockDim.x) {
    const int l = i / c4n, c = (i - l * c4n) * 4;
    const float* t = tile + l * (Cp + 1) + c;
    reinterpret_cast<float4*>(dst)[i] = make_float4(t[0], t[1], t[2], t[3]);
  }
}
__device__ __forceinline__ void tile_zero_pad(float* tile, int C, int L, int Cp) {
  const int np = Cp - C;
  for (int i = threadIdx.x; i < L * np; i += blockDim.x) {
    const int l = i / np, c = C + (i - l * np);
    tile[l * (Cp + 1) + c] = 0.f;
  }
}

// xin[b,l,c] = c_in * x[b,c,l]                                             (diffusion.py:810)
__global__ __launch_bounds__(256) void k_precond_in(const float* x, float* xin, float c_in, int C, int L, int Cp) {
  extern __shared__ float tile[];
  const int b = blockIdx.x;
  const float* xb = x + (int64_t)b * C * L;
  const int l4n = L / 4;
  for (int e = threadIdx.x; e < C * l4n; e += blockDim.x) {
    const int c = e / l4n, l = (e - c * l4n) * 4;
    const float4 v = *reinterpret_cast<const float4*>(xb + c * L + l);
    tile[(l + 0) * (Cp + 1) + c] = c_in * v.x;
    tile[(l + 1) * (Cp + 1) + c] = c_in * v.y;
    tile[(l + 2) * (Cp + 1) + c] = c_in * v.z;
    tile[(l + 3) * (Cp + 1) + c] = c_in * v.w;
  }
  tile_zero_pad(tile, C, L, Cp);
  __syncthreads();
  tile_store(tile, xin + (int64_t)b * L * Cp, L, Cp);
}

// D = clamp(c_skip*x + c_out*pred, -1, 1)                                  (diffusion.py:811-814)
__global__ __launch_bounds__(256) void k_precond_out(const float* x, const float* pred, float* D, float c_skip,
                                                      float c_out, int C, int L, int Cp, const float* dscale) {
  extern __shared__ float tile[];
  const int b = blockIdx.x;
  const float ds = dscale ? dscale[b] : 0.f;
  tile_load(tile, pred + (int64_t)b * L * Cp, L, Cp);
  __syncthreads();
  const int l4n = L / 4;
  for (int e = threadIdx.x; e < C * l4n; e += blockDim.x) {
    const int c = e / l4n, l = (e - c * l4n) * 4;
    const int64_t o = (int64_t)b * C * L + c * L + l;
    const float4 v = *reinterpret_cast<const float4*>(x + o);
    float4 d;
    d.x = clip_dyn(c_skip * v.x + c_out * tile[(l + 0) * (Cp + 1) + c], ds);
    d.y = clip_dyn(c_skip * v.y + c_out * tile[(l + 1) * (Cp + 1) + c], ds);
    d.z = clip_dyn(c_skip * v.z + c_out * tile[(l + 2) * (Cp + 1) + c], ds);
    d.w = clip_dyn(c_skip * v.w + c_out * tile[(l + 3) * (Cp + 1) + c], ds);
    *reinterpret_cast<float4*>(D + o) = d;
  }
}

// Dynamic thresholding (clip() with dynamic_threshold = q > 0, diffusion.py:78-88): per sample
//   scale[b] = max(torch.quantile(|c_skip x + c_out pred|.flatten(), q), 1)
// One workgroup per sample: the N = C L magnitudes are sorted in LDS (bitonic, padded with +inf to a power of two) and the
// quantile is torch's linear interpolation between the two neighbouring order statistics: rank = q (N - 1) in fp32,
// lerp(v[floor], v[ceil], rank - floor) with ATen's two-sided formula.  The consumers (k_precond_out / k_adpm2_mid / k_adpm2_next)
// then clamp to [-scale, scale] and divide.  Rarely used (every class of the reference passes 0.0): simple, not tuned.
__global__ __launch_bounds__(256) void k_dyn_scale(const float* x, const float* pred, float* scale, float c_skip, float c_out,
                                                    float q, int C, int L, int Cp, int npad) {
  extern __shared__ float tile[];
  const int b = blockIdx.x, N = C * L;
  const float* xb = x + (int64_t)b * N;
  const float* pb = pred + (int64_t)b * L * Cp;
  for (int e = threadIdx.x; e < npad; e += blockDim.x) {
    float v = INFINITY;
    if (e < N) {
      const int c = e / L, l = e - c * L;
      v = fabsf(c_skip * xb[e] + c_out * pb[l * Cp + c]);
    }
    tile[e] = v;
  }
  __syncthreads();
  for (int k = 2; k <= npad; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int e = threadIdx.x; e < npad; e += blockDim.x) {
        const int p = e ^ j;
        if (p > e) {
          const float a0 = tile[e], a1 = tile[p];
          const bool up = (e & k) == 0;
          if ((a0 > a1) == up) { tile[e] = a1; tile[p] = a0; }
        }
      }
      __syncthreads();
    }
  if (threadIdx.x == 0) {
    const float rank = q * (float)(N - 1);
    const float below = floorf(rank), above = ceilf(rank);
    const float w = rank - below;
    const float v0 = tile[(int)below], v1 = tile[(int)above];
    const float diff = v1 - v0;
    const float qv = fabsf(w) < 0.5f ? v0 + w * diff : v1 - diff * (1.0f - w);
    scale[b] = fmaxf(qv, 1.0f);
  }
}

// First half of ADPM2Sampler.step fused with denoise_fn's output stage   (diffusion.py:506-508, :811-814)
__global__ __launch_bounds__(256) void k_adpm2_mid(const float* x, const float* pred, float* x_mid, float* xin_mid,
                                                    float c_skip, float c_out, float sigma, float dt_mid,
                                                    float c_in_mid, int C, int L, int Cp, const float* dscale) {
  extern __shared__ float tile[];
  const int b = blockIdx.x;
  const float ds = dscale ? dscale[b] : 0.f;
  tile_load(tile, pred + (int64_t)b * L * Cp, L, Cp);
  __syncthreads();
  const int l4n = L / 4;
  for (int e = threadIdx.x; e < C * l4n; e += blockDim.x) {
    const int c = e / l4n, l = (e - c * l4n) * 4;
    const int64_t o = (int64_t)b * C * L + c * L + l;
    const float4 v = *reinterpret_cast<const float4*>(x + o);
    const float xv[4] = {v.x, v.y, v.z, v.w};
    float xm[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float* t = tile + (l + q) * (Cp + 1) + c;
      const float den = clip_dyn(c_skip * xv[q] + c_out * (*t), ds);
      const float d = (xv[q] - den) / sigma;
      xm[q] = xv[q] + d * dt_mid;
      *t = c_in_mid * xm[q];   // same thread owns (l, c): in-place reuse of the tile for xin_mid
    }
    *reinterpret_cast<float4*>(x_mid + o) = make_float4(xm[0], xm[1], xm[2], xm[3]);
  }
  tile_zero_pad(tile, C, L, Cp);
  __syncthreads();
  tile_store(tile, xin_mid + (int64_t)b * L * Cp, L, Cp);
}

// Second half of ADPM2Sampler.step                                        (diffusion.py:510-515)
// tokens != nullptr (last step of a call): also the decode step after the path, tokens[b,l] = argmax_c x[b,c,l] of the
// final x (generative.py:1212-1213, :1690-1691: permute(0,2,1) -> argmax(dim=2); first maximum, NaN wins, as torch.argmax),
// taken from the values this kernel has in LDS anyway instead of a second pass over the (B, C, L) result.
__global__ __launch_bounds__(256) void k_adpm2_next(float* x, const float* x_mid, const float* pred,
                                                     const float* noise, float* xin_next, float c_skip, float c_out,
                                                     float sigma_mid, float dt_down, float sigma_up, float c_in_next,
                                                     uint64_t seed, uint32_t step, int64_t sample0, int C, int L,
                                                     int Cp, int32_t* tokens, const float* dscale) {
  extern __shared__ float tile[];
  const int b = blockIdx.x;
  const float ds = dscale ? dscale[b] : 0.f;
  tile_load(tile, pred + (int64_t)b * L * Cp, L, Cp);
  __syncthreads();
  const int l4n = L / 4;
  const bool keep_x = tokens && !xin_next;       // the tile then carries x itself (not c_in_next * x) for the argmax
  for (int e = threadIdx.x; e < C * l4n; e += blockDim.x) {
    const int c = e / l4n, l = (e - c * l4n) * 4;
    const int64_t o = (int64_t)b * C * L + c * L + l;
    const float4 v = *reinterpret_cast<const float4*>(x + o);
    const float4 m = *reinterpret_cast<const float4*>(x_mid + o);
    float4 nz;
    if (noise) nz = *reinterpret_cast<const float4*>(noise + o);
    else nz = normal4(seed, step, (uint64_t)(((sample0 + b) * C + c) * (int64_t)L + l) >> 2);
    const float xv[4] = {v.x, v.y, v.z, v.w}, mv[4] = {m.x, m.y, m.z, m.w}, nv[4] = {nz.x, nz.y, nz.z, nz.w};
    float xn[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float* t = tile + (l + q) * (Cp + 1) + c;
      const float den = clip_dyn(c_skip * mv[q] + c_out * (*t), ds);
      const float d = (mv[q] - den) / sigma_mid;
      float xx = xv[q] + d * dt_down;
      xx = xx + nv[q] * sigma_up;
      xn[q] = xx;
      *t = keep_x ? xx : c_in_next * xx;
    }
    *reinterpret_cast<float4*>(x + o) = make_float4(xn[0], xn[1], xn[2], xn[3]);
  }
  if (xin_next) {
    tile_zero_pad(tile, C, L, Cp);
    __syncthreads();
    tile_store(tile, xin_next + (int64_t)b * L * Cp, L, Cp);
  } else if (tokens) {
    __syncthreads();
    for (int l = threadIdx.x; l < L; l += blockDim.x) {
      const float* t = tile + l * (Cp + 1);
      float best = t[0];
      int arg = 0;
      for (int c = 1; c < C; ++c) {
        const float v = t[c];
        if (v > best || (v != v && best == best)) { best = v; arg = c; }
      }
      tokens[(int64_t)b * L + l] = arg;
    }
  }
}

// Flat (B*C*L) kernels: 4 elements per thread.
__global__ __launch_bounds__(256) void k_init_noise(float* x, const float* noise, float sigma0, uint64_t seed,
                                                     uint32_t step, int64_t elem0, int64_t n4) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 nz = noise ? reinterpret_cast<const float4*>(noise)[i] : normal4(seed, step, (uint64_t)((elem0 >> 2) + i));
    reinterpret_cast<float4*>(x)[i] = make_float4(sigma0 * nz.x, sigma0 * nz.y, sigma0 * nz.z, sigma0 * nz.w);
  }
}

__global__ __launch_bounds__(256) void k_add_noise(float* x, const float* noise, float s, uint64_t seed, uint32_t step,
                                                    int64_t elem0, int64_t n4) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 nz = noise ? reinterpret_cast<const float4*>(noise)[i] : normal4(seed, step, (uint64_t)((elem0 >> 2) + i));
    float4 v = reinterpret_cast<float4*>(x)[i];
    v.x = v.x + s * nz.x; v.y = v.y + s * nz.y; v.z = v.z + s * nz.z; v.w = v.w + s * nz.w;
    reinterpret_cast<float4*>(x)[i] = v;
  }
}

// One Euler move of ADPM2Sampler.step with the denoised tensor supplied by the caller's fn (diffusion.py:506-515):
//   out = x_base + ((x_from - den) / sigma) * dt   [+ noise * sigma_up]
// first half: x_base = x_from = x, dt = sigma_mid - sigma; second half: x_base = x, x_from = x_mid, dt = sigma_down - sigma
__global__ __launch_bounds__(256) void k_adpm2_euler(const float* xb, const float* xf, const float* den, const float* noise,
                                                      float* out, float sigma, float dt, float sigma_up, int noise_mode,
                                                      uint64_t seed, uint32_t step, int64_t elem0, int64_t n4) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 b = reinterpret_cast<const float4*>(xb)[i], f = reinterpret_cast<const float4*>(xf)[i];
    const float4 d = reinterpret_cast<const float4*>(den)[i];
    float4 o;
    o.x = b.x + ((f.x - d.x) / sigma) * dt; o.y = b.y + ((f.y - d.y) / sigma) * dt;
    o.z = b.z + ((f.z - d.z) / sigma) * dt; o.w = b.w + ((f.w - d.w) / sigma) * dt;
    if (noise_mode) {
      const float4 nz = noise_mode == 1 ? reinterpret_cast<const float4*>(noise)[i]
                                        : normal4(seed, step, (uint64_t)((elem0 >> 2) + i));
      o.x = o.x + nz.x * sigma_up; o.y = o.y + nz.y * sigma_up; o.z = o.z + nz.z * sigma_up; o.w = o.w + nz.w * sigma_up;
    }
    reinterpret_cast<float4*>(out)[i] = o;
  }
}

// x = where(mask, src + sigma*noise, x)                                    (diffusion.py:539-542, :549)
__global__ __launch_bounds__(256) void k_inpaint_merge(float* x, const float* src, const uint8_t* mask,
                                                        const float* noise, float sigma, uint64_t seed, uint32_t step,
                                                        int64_t elem0, int64_t n4) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 nz = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sigma != 0.0f)
      nz = noise ? reinterpret_cast<const float4*>(noise)[i] : normal4(seed, step, (uint64_t)((elem0 >> 2) + i));
    const float4 sv = reinterpret_cast<const float4*>(src)[i];
    const uchar4 mk = reinterpret_cast<const uchar4*>(mask)[i];
    float4 v = reinterpret_cast<float4*>(x)[i];
    if (sigma != 0.0f) {
      if (mk.x) v.x = sv.x + sigma * nz.x;
      if (mk.y) v.y = sv.y + sigma * nz.y;
      if (mk.z) v.z = sv.z + sigma * nz.z;
      if (mk.w) v.w = sv.w + sigma * nz.w;
    } else {
      if (mk.x) v.x = sv.x;
      if (mk.y) v.y = sv.y;
      if (mk.z) v.z = sv.z;
      if (mk.w) v.w = sv.w;
    }
    reinterpret_cast<float4*>(x)[i] = v;
  }
}

// dst[0..n) = src[0..n): the FiLM rows of ONE evaluation out of the table the time program fills once per call.  One small launch on
// the caller's stream (hipMemcpyAsync of the same bytes ran as up to three runtime copy kernels of ~4 us each in front of every
// evaluation graph: profiles/r6_kernel_stats.csv, __amd_rocclr_copyBuffer).
__global__ __launch_bounds__(256) void k_copy_f32(float* __restrict__ dst, const float* __restrict__ src, int64_t n, int vec) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (vec) {
    if (t < n / 4) reinterpret_cast<float4*>(dst)[t] = reinterpret_cast<const float4*>(src)[t];
  } else {
    for (int64_t k = t; k < n; k += (int64_t)gridDim.x * 256) dst[k] = src[k];
  }
}

__global__ __launch_bounds__(256) void k_clamp(float* x, float lo, float hi, int64_t n4) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 v = reinterpret_cast<float4*>(x)[i];
    v.x = fminf(fmaxf(v.x, lo), hi); v.y = fminf(fmaxf(v.y, lo), hi);
    v.z = fminf(fmaxf(v.z, lo), hi); v.w = fminf(fmaxf(v.w, lo), hi);
    reinterpret_cast<float4*>(x)[i] = v;
  }
}

// out = um + (cond - um) * scale                                           (modules.py:1253)
__global__ __launch_bounds__(256) void k_cfg_mix(const float* cond, const float* um, float* out, float scale,
                                                  int64_t n4) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 c = reinterpret_cast<const float4*>(cond)[i], u = reinterpret_cast<const float4*>(um)[i];
    reinterpret_cast<float4*>(out)[i] = make_float4(u.x + (c.x - u.x) * scale, u.y + (c.y - u.y) * scale,
                                                    u.z + (c.z - u.z) * scale, u.w + (c.w - u.w) * scale);
  }
}

// tokens[b,l] = argmax_c x[b,c,l] (first maximum, as torch.argmax)          (generative.py:1212-1213)
__global__ __launch_bounds__(256) void k_argmax(const float* x, int32_t* tok, int B, int C, int L) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * L) return;
  const int b = (int)(i / L), l = (int)(i - (int64_t)b * L);
  const float* p = x + (int64_t)b * C * L + l;
  float best = p[0];
  int arg = 0;
  for (int c = 1; c < C; ++c) {
    const float v = p[(int64_t)c * L];
    if (v > best || (v != v && best == best)) { best = v; arg = c; }
  }
  tok[i] = arg;
}

// e[b,i,:D1] = gelu(w*s+b), e[b,i,D1:] = [sin(i f) | cos(i f)]             (generative.py:838-850, transformer.py:3456-3470)
// add != 0 (pos_emb_fourier_add, generative.py:844-846): e[b,i,f] = gelu(w*s+b)[f] + PositionalEncoding1D(D2)[i,f], f < D1 <= D2
// (the reference's encoding returns emb[:, :, :orig_ch], the FIRST D1 columns of [sin (D2/2) | cos (D2/2)], transformer.py:3470)
__global__ __launch_bounds__(256) void k_cond_embed(const float* seq, const float* w, const float* bias,
                                                     const float* inv_freq, float* out, int B, int n, int D1, int D2, int add) {
  const int F = add ? D1 : D1 + D2;
  const int64_t total = (int64_t)B * n * F;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(i % F);
    const int64_t row = i / F;
    const int pos = (int)(row % n);
    float v = 0.f;
    if (f < D1) {
      const float h = seq[row] * w[f] + bias[f];
      v = 0.5f * h * (1.0f + erff(h * 0.70710678118654752440f));
    }
    if (add || f >= D1) {
      const int j = add ? f : f - D1, half = D2 / 2;
      const float arg = (float)pos * inv_freq[j < half ? j : j - half];
      const float pe = j < half ? sinf(arg) : cosf(arg);
      v = add ? v + pe : pe;
    }
    out[i] = v;
  }
}

// LearnedPositionalEmbedding: [t, sin(t w 2 pi), cos(t w 2 pi)] padded to ld  (modules.py:554-559)
__global__ __launch_bounds__(256) void k_time_embed(const float* cn, const float* w, float* out, int rows, int half,
                                                     int ld) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * ld) return;
  const int r = i / ld, c = i - r * ld;
  const float t = cn[r];
  float v = 0.f;
  if (c == 0) v = t;
  else if (c <= 2 * half) {
    const int j = c - 1;
    const float fr = t * w[j < half ? j : j - half] * 2.0f * 3.14159265358979323846f;
    v = j < half ? sinf(fr) : cosf(fr);
  }
  out[i] = v;
}

// out[row, :ca] = a[row, :], out[row, ca:ca+cb] = b[row, :] * scale_b       (modules.py:828-829)
__global__ __launch_bounds__(256) void k_concat(const float* a, const float* b, float* out, int64_t rows, int ca,
                                                 int cb, float scale_b) {
  const int w4 = (ca + cb) / 4, ca4 = ca / 4;
  const int64_t total = rows * w4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / w4;
    const int c = (int)(i - row * w4);
    float4 v;
    if (c < ca4) v = reinterpret_cast<const float4*>(a)[row * ca4 + c];
    else {
      v = reinterpret_cast<const float4*>(b)[row * (cb / 4) + (c - ca4)];
      v.x *= scale_b; v.y *= scale_b; v.z *= scale_b; v.w *= scale_b;
    }
    reinterpret_cast<float4*>(out)[i] = v;
  }
}

// Patcher 'b c (l p) -> b (c p) l' / Unpatcher ' b (c p) l -> b c (l p) ' on token-major tensors
// (modules.py:230, :255): fwd  y[b, l, c*p + q] = x[b, l*p + q, c];  inverse the other way round.
__global__ __launch_bounds__(256) void k_patch(const float* in, float* out, int batch, int rows_in, int c_in, int ld_in,
                                                int ld_out, int patch, int inverse) {
  // rows_in / c_in always describe the UNPATCHED side (long sequence, few channels).
  const int64_t total = (int64_t)batch * rows_in * c_in;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c_in);
    const int64_t t = i / c_in;
    const int r = (int)(t % rows_in);
    const int64_t b = t / rows_in;
    const int l = r / patch, q = r - l * patch;
    const int64_t long_idx = (b * rows_in + r) * (int64_t)(inverse ? ld_out : ld_in) + c;
    const int64_t short_idx = (b * (rows_in / patch) + l) * (int64_t)(inverse ? ld_in : ld_out) + c * patch + q;
    if (inverse) out[long_idx] = in[short_idx];
    else out[short_idx] = in[long_idx];
  }
}

static inline unsigned grid_for(int64_t n, int block = 256, int cap = 256 * 8) {
  int64_t g = (n + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (unsigned)g;
}

hipError_t launch_concat(const float* a, const float* b, float* out, int64_t rows, int ca, int cb, float scale_b,
                         hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_concat, dim3(grid_for(rows * (ca + cb) / 4)), dim3(256), 0, s, a, b, out, rows, ca, cb, scale_b);
  return hipGetLastError();
}
hipError_t launch_patch(const float* in, float* out, int batch, int rows_in, int c_in, int ld_in, int ld_out, int patch,
                        int inverse, hipStream_t s) {
  if (batch <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_patch, dim3(grid_for((int64_t)batch * rows_in * c_in)), dim3(256), 0, s, in, out, batch,
                     rows_in, c_in, ld_in, ld_out, patch, inverse);
  return hipGetLastError();
}
hipError_t launch_time_embed(const float* cn, const float* w, float* out, int rows, int half, int ld, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_time_embed, dim3((rows * ld + 255) / 256), dim3(256), 0, s, cn, w, out, rows, half, ld);
  return hipGetLastError();
}

}  // namespace mdt

// ------------------------------------------------------------------------------------------------
// C ABI entry points of this translation unit (declared in include/mdt_hip.h)
// ------------------------------------------------------------------------------------------------
extern "C" __attribute__((visibility("hidden"))) void mdt_set_error(const char* msg);  // mdt_api.cpp (not exported)

namespace {
inline int finish(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    mdt_set_error(buf);
    return 1;
  }
  return 0;
}
inline int bad(const char* msg) {
  mdt_set_error(msg);
  return 2;
}
inline size_t tile_bytes(int L, int Cp) { return (size_t)L * (Cp + 1) * sizeof(float); }
}  // namespace

extern "C" {

int mdt_cond_embed(const float* seq, const float* fc1_w, const float* fc1_b, const float* inv_freq, float* out,
                   int32_t B, int32_t n, int32_t D1, int32_t D2, void* stream) {
  if (B <= 0) return 0;
  if (D2 % 2) return bad("mdt_cond_embed: D2 must be even");
  hipLaunchKernelGGL(mdt::k_cond_embed, dim3(mdt::grid_for((int64_t)B * n * (D1 + D2))), dim3(256), 0,
                     (hipStream_t)stream, seq, fc1_w, fc1_b, inv_freq, out, B, n, D1, D2, 0);
  return finish("mdt_cond_embed");
}

int mdt_cond_embed_add(const float* seq, const float* fc1_w, const float* fc1_b, const float* inv_freq, float* out,
                       int32_t B, int32_t n, int32_t D1, int32_t D2, void* stream) {
  if (!seq || !fc1_w || !fc1_b || !inv_freq || !out) return bad("mdt_cond_embed_add: null pointer");
  if (B <= 0) return 0;
  if (D1 <= 0 || D2 <= 0 || D2 % 2 || D1 > D2) return bad("mdt_cond_embed_add: need 0 < D1 <= D2, D2 even");
  hipLaunchKernelGGL(mdt::k_cond_embed, dim3(mdt::grid_for((int64_t)B * n * D1)), dim3(256), 0, (hipStream_t)stream, seq,
                     fc1_w, fc1_b, inv_freq, out, B, n, D1, D2, 1);
  return finish("mdt_cond_embed_add");
}

// the sampler kernels stage one sample's (L x Cp) tile in LDS: up to the CU's 160 KiB (the default limit of a launch is 64 KiB:
// raised once per kernel and device).  max_length = 1024 (the reference constructors' default) at 16 padded channels is 68 KiB.
#define MDT_CHECK_TILE(name)                                                                   \
  if (B <= 0) return 0;                                                                        \
  if (L % 4 || Cp % 16 || Cp < C) return bad(name ": need L % 4 == 0, Cp % 16 == 0, Cp >= C"); \
  if (tile_bytes(L, Cp) > 160 * 1024) return bad(name ": (L, Cp) tile exceeds the 160 KiB of LDS of a compute unit");
#define MDT_BIG_LDS(kernel)                                                                                          \
  do {                                                                                                                \
    static mdt::DevOnce once_;                                    /* per device, not per process (mdt_kernels.h) */     \
    if (once_.first())                                                                                                \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
  } while (0)

int mdt_precond_in(const float* x, float* xin, float c_in, int32_t B, int32_t C, int32_t L, int32_t Cp,
                   void* stream) {
  MDT_CHECK_TILE("mdt_precond_in")
  MDT_BIG_LDS(mdt::k_precond_in);
  hipLaunchKernelGGL(mdt::k_precond_in, dim3(B), dim3(256), tile_bytes(L, Cp), (hipStream_t)stream, x, xin, c_in, C, L,
                     Cp);
  return finish("mdt_precond_in");
}

int mdt_precond_out(const float* x, const float* pred, float* D, float c_skip, float c_out, int32_t B, int32_t C,
                    int32_t L, int32_t Cp, const float* dyn_scale, void* stream) {
  MDT_CHECK_TILE("mdt_precond_out")
  MDT_BIG_LDS(mdt::k_precond_out);
  hipLaunchKernelGGL(mdt::k_precond_out, dim3(B), dim3(256), tile_bytes(L, Cp), (hipStream_t)stream, x, pred, D,
                     c_skip, c_out, C, L, Cp, dyn_scale);
  return finish("mdt_precond_out");
}

int mdt_dyn_scale(const float* x, const float* pred, float* scale, float c_skip, float c_out, float q, int32_t B, int32_t C,
                  int32_t L, int32_t Cp, void* stream) {
  if (B <= 0) return 0;
  if (!x || !pred || !scale) return bad("mdt_dyn_scale: null pointer");
  if (!(q > 0.0f && q <= 1.0f)) return bad("mdt_dyn_scale: the quantile must lie in (0, 1]");
  if (C <= 0 || L <= 0 || Cp < C) return bad("mdt_dyn_scale: bad dims");
  int npad = 1;
  while (npad < C * L) npad <<= 1;
  if ((size_t)npad * sizeof(float) > 160 * 1024) return bad("mdt_dyn_scale: C * L exceeds 32768 values (the sort runs in one compute unit's LDS)");
  MDT_BIG_LDS(mdt::k_dyn_scale);
  hipLaunchKernelGGL(mdt::k_dyn_scale, dim3(B), dim3(256), (size_t)npad * sizeof(float), (hipStream_t)stream, x, pred, scale,
                     c_skip, c_out, q, C, L, Cp, npad);
  return finish("mdt_dyn_scale");
}

int mdt_adpm2_mid(const float* x, const float* pred, float* x_mid, float* xin_mid, float c_skip, float c_out,
                  float sigma, float dt_mid, float c_in_mid, int32_t B, int32_t C, int32_t L, int32_t Cp,
                  const float* dyn_scale, void* stream) {
  MDT_CHECK_TILE("mdt_adpm2_mid")
  MDT_BIG_LDS(mdt::k_adpm2_mid);
  hipLaunchKernelGGL(mdt::k_adpm2_mid, dim3(B), dim3(256), tile_bytes(L, Cp), (hipStream_t)stream, x, pred, x_mid,
                     xin_mid, c_skip, c_out, sigma, dt_mid, c_in_mid, C, L, Cp, dyn_scale);
  return finish("mdt_adpm2_mid");
}

int mdt_adpm2_next(float* x, const float* x_mid, const float* pred, const float* noise, float* xin_next, float c_skip,
                   float c_out, float sigma_mid, float dt_down, float sigma_up, float c_in_next, uint64_t seed,
                   uint32_t step, int64_t sample0, int32_t B, int32_t C, int32_t L, int32_t Cp, int32_t* tokens,
                   const float* dyn_scale, void* stream) {
  MDT_CHECK_TILE("mdt_adpm2_next")
  if (tokens && xin_next) return bad("mdt_adpm2_next: tokens are decoded on the LAST update of a call (xin_next == NULL)");
  MDT_BIG_LDS(mdt::k_adpm2_next);
  hipLaunchKernelGGL(mdt::k_adpm2_next, dim3(B), dim3(256), tile_bytes(L, Cp), (hipStream_t)stream, x, x_mid, pred,
                     noise, xin_next, c_skip, c_out, sigma_mid, dt_down, sigma_up, c_in_next, seed, step, sample0, C, L,
                     Cp, tokens, dyn_scale);
  return finish("mdt_adpm2_next");
}

int mdt_init_noise(float* x, const float* noise, float sigma0, uint64_t seed, uint32_t step, int64_t sample0, int32_t B,
                   int32_t C, int32_t L, void* stream) {
  if (B <= 0) return 0;
  if (L % 4) return bad("mdt_init_noise: L % 4 != 0");
  const int64_t n4 = (int64_t)B * C * L / 4;
  hipLaunchKernelGGL(mdt::k_init_noise, dim3(mdt::grid_for(n4)), dim3(256), 0, (hipStream_t)stream, x, noise, sigma0,
                     seed, step, sample0 * C * L, n4);
  return finish("mdt_init_noise");
}

int mdt_add_noise(float* x, const float* noise, float s, uint64_t seed, uint32_t step, int64_t sample0, int32_t B,
                  int32_t C, int32_t L, void* stream) {
  if (B <= 0) return 0;
  if (L % 4) return bad("mdt_add_noise: L % 4 != 0");
  const int64_t n4 = (int64_t)B * C * L / 4;
  hipLaunchKernelGGL(mdt::k_add_noise, dim3(mdt::grid_for(n4)), dim3(256), 0, (hipStream_t)stream, x, noise, s, seed,
                     step, sample0 * C * L, n4);
  return finish("mdt_add_noise");
}

int mdt_inpaint_merge(float* x, const float* src, const uint8_t* mask, const float* noise, float sigma, uint64_t seed,
                      uint32_t step, int64_t sample0, int32_t B, int32_t C, int32_t L, void* stream) {
  if (B <= 0) return 0;
  if (L % 4) return bad("mdt_inpaint_merge: L % 4 != 0");
  const int64_t n4 = (int64_t)B * C * L / 4;
  hipLaunchKernelGGL(mdt::k_inpaint_merge, dim3(mdt::grid_for(n4)), dim3(256), 0, (hipStream_t)stream, x, src, mask,
                     noise, sigma, seed, step, sample0 * C * L, n4);
  return finish("mdt_inpaint_merge");
}

int mdt_adpm2_euler(const float* x_base, const float* x_from, const float* denoised, const float* noise, float* out,
                    float sigma, float dt, float sigma_up, int32_t noise_mode, uint64_t seed, uint32_t step,
                    int64_t sample0, int32_t B, int32_t C, int32_t L, void* stream) {
  if (B <= 0) return 0;
  if (L % 4) return bad("mdt_adpm2_euler: L % 4 != 0");
  if (noise_mode < 0 || noise_mode > 2 || (noise_mode == 1 && !noise)) return bad("mdt_adpm2_euler: bad noise mode");
  const int64_t n4 = (int64_t)B * C * L / 4;
  hipLaunchKernelGGL(mdt::k_adpm2_euler, dim3(mdt::grid_for(n4)), dim3(256), 0, (hipStream_t)stream, x_base, x_from,
                     denoised, noise, out, sigma, dt, sigma_up, noise_mode, seed, step, sample0 * C * L, n4);
  return finish("mdt_adpm2_euler");
}

int mdt_copy_f32(float* dst, const float* src, int64_t n, void* stream) {
  if (n <= 0) return 0;
  if (!dst || !src) return bad("mdt_copy_f32: null pointer");
  const bool vec = n % 4 == 0 && ((reinterpret_cast<size_t>(dst) | reinterpret_cast<size_t>(src)) & 15) == 0;
  const int64_t items = vec ? n / 4 : (n < 65536 ? n : 65536);
  hipLaunchKernelGGL(mdt::k_copy_f32, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dst, src, n, vec ? 1 : 0);
  return finish("mdt_copy_f32");
}

int mdt_clamp(float* x, float lo, float hi, int64_t n, void* stream) {
  if (n <= 0) return 0;
  if (n % 4) return bad("mdt_clamp: n % 4 != 0");
  hipLaunchKernelGGL(mdt::k_clamp, dim3(mdt::grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, x, lo, hi, n / 4);
  return finish("mdt_clamp");
}

int mdt_cfg_mix(const float* cond, const float* uncond, float* out, float scale, int64_t n, void* stream) {
  if (n <= 0) return 0;
  if (n % 4) return bad("mdt_cfg_mix: n % 4 != 0");
  hipLaunchKernelGGL(mdt::k_cfg_mix, dim3(mdt::grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, cond, uncond, out,
                     scale, n / 4);
  return finish("mdt_cfg_mix");
}

int mdt_argmax_tokens(const float* x, int32_t* tokens, int32_t B, int32_t C, int32_t L, void* stream) {
  if (B <= 0) return 0;
  const int64_t n = (int64_t)B * L;
  hipLaunchKernelGGL(mdt::k_argmax, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, tokens, B,
                     C, L);
  return finish("mdt_argmax_tokens");
}

}  // extern "C"
