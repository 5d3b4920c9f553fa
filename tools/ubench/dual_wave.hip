// Premise of the large-batch form of the ring kernels (round 4): ALL eight waves of a workgroup compute -- two per SIMD, each
// with its own 16 rows and the whole 32 KB tile (48 split-bf16 MFMAs per wave per tile, twice the rows per weight byte) -- and
// issue the LDS-DMA stream themselves (4 pieces per wave per tile) instead of leaving it to four dedicated loader waves.
// Compared with the shipped structure (kpipe<1> of proj_phase.hip: 4 compute + 4 loader waves, 1147 cycles per tile for 64
// rows) at equal work per ROW: the dual form processes 128 rows per tile.
//
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 tools/ubench/dual_wave.hip -o /tmp/dual_wave && /tmp/dual_wave
//
// Variants: VALU = n GELU-like values per lane between tiles (the serial sections of the real kernels: softmax, GELU, LayerNorm),
// FRAG = 1: fragment-ordered tiles (linear DMA, ds_read_b128 at lane * 16 + immediate).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA __builtin_amdgcn_mfma_f32_16x16x32_bf16

constexpr int C = 128, SLOT = 256 * C, NS = 4, NU = 8;

template <int OFF>
__device__ __forceinline__ void lds_read16_off(bf16x8& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}

// NCW compute waves (4: + 4 loader waves, the shipped structure; 8: every wave computes and loads)
// SKEW = 1 (NCW = 8): waves 4-7 run their serial VALU section BEFORE the tile's MFMA units, waves 0-3 after them: inside every barrier
// interval one wave of a SIMD issues MFMAs while the other is in its VALU section (the lock-step of the per-tile barrier is kept, the
// phases inside the interval are swapped)
template <int NCW, int VALU, int SKEW = 0>
__global__ __launch_bounds__(512) void kdual(const unsigned char* w, float* out, unsigned long long* cyc, int ntiles, int wtiles) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NLW = NCW == 8 ? 8 : 4;             // waves that issue DMA
  constexpr int PIECES = 32 / NLW;
  const bool loader_only = NCW == 4 && wave >= 4;
  const int iw = NCW == 8 ? wave : wave - 4;
  auto issue_tile = [&](int tau) {
    const unsigned char* tile = w + (int64_t)(tau % wtiles) * SLOT;
    unsigned char* slot = smem + (tau & (NS - 1)) * SLOT;
#pragma unroll
    for (int q = 0; q < PIECES; ++q) {
      const int inst = iw + NLW * q;
      __builtin_amdgcn_global_load_lds(tile + inst * 1024 + lane * 16, (__attribute__((address_space(3))) void*)(slot + inst * 1024), 16, 0, 0);
    }
  };
  auto wait_mine = [&](bool more) {                 // my pieces of the tile about to be published have landed
    if (more) {
      if constexpr (PIECES == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };
  unsigned long long t0 = 0, t1 = 0;
  if (loader_only) {
    __builtin_amdgcn_s_setprio(3);
    issue_tile(0);
    issue_tile(1);
    float lv[VALU > 0 ? VALU : 1];
    for (int e = 0; e < VALU; ++e) lv[e] = 0.001f * (lane + e);
    for (int k = 0; k < ntiles; ++k) {
      wait_mine(k + 1 < ntiles);
      __builtin_amdgcn_s_barrier();
      if (k + 2 < ntiles) issue_tile(k + 2);
      if constexpr (SKEW == 2 && VALU > 0) {       // the serial section handed to the (otherwise parked) loader wave of the SIMD
#pragma unroll
        for (int e = 0; e < VALU; ++e) {
          float x = lv[e];
          const float z = fabsf(x) * 0.70710678f;
          const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
          const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
          const float erfa = 1.0f - poly * __expf(-z * z);
          lv[e] = 0.5f * x * (1.0f + copysignf(erfa, x)) + 1e-3f;
        }
      }
    }
    if constexpr (SKEW == 2 && VALU > 0) {
      float r = 0.f;
      for (int e = 0; e < VALU; ++e) r += lv[e];
      out[blockIdx.x * 512 + tid] = r;
    }
    return;
  }
  bf16x8 xh[4], xl[4];
  for (int st = 0; st < 4; ++st)
    for (int e = 0; e < 8; ++e) { xh[st][e] = (__bf16)(float)(lane + e + st); xl[st][e] = (__bf16)(float)(lane - e); }
  f32x4 acc[4];
  for (int n = 0; n < 4; ++n) acc[n] = f32x4{0, 0, 0, 0};
  float vv[VALU > 0 ? VALU : 1];
  for (int e = 0; e < VALU; ++e) vv[e] = 0.001f * (lane + e);
  if (NCW == 8) { issue_tile(0); issue_tile(1); }
  bf16x8 fh[3][2], fl[3][2];
  // fragment-ordered tile: fragment (ft, st, plane) at ft * 8192 + st * 2048 + plane * 1024, the lane's 16 bytes inside
  auto frag = [&](unsigned base, auto uc, int set, auto jc) __attribute__((always_inline)) {
    constexpr int u = decltype(uc)::value, j = decltype(jc)::value;
    constexpr int q = j >> 1, lo = j & 1;
    constexpr int off = (2 * (u & 1) + q) * 8192 + (u >> 1) * 2048 + lo * 1024;
    lds_read16_off<off>(lo ? fl[set][q] : fh[set][q], base);
  };
  using J0 = std::integral_constant<int, 0>; using J1 = std::integral_constant<int, 1>;
  using J2 = std::integral_constant<int, 2>; using J3 = std::integral_constant<int, 3>;
  auto lds_of = [&](int t) -> unsigned { return (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(smem + (t & (NS - 1)) * SLOT) + lane * 16; };
  // B(0)
  if (NCW == 8) wait_mine(true);
  __builtin_amdgcn_s_barrier();
  if (NCW == 8) issue_tile(2);
  {
    const unsigned b = lds_of(0);
    frag(b, J0{}, 0, J0{}); frag(b, J0{}, 0, J1{}); frag(b, J0{}, 0, J2{}); frag(b, J0{}, 0, J3{});
    frag(b, J1{}, 1, J0{}); frag(b, J1{}, 1, J1{}); frag(b, J1{}, 1, J2{}); frag(b, J1{}, 1, J3{});
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(8)" : "=s"(t0)::"memory");
  int off = 0;
  for (int tau = 0; tau < ntiles; ++tau) {
    const bool has_next = tau + 1 < ntiles;
    const unsigned lc = lds_of(tau), ln = lds_of(tau + 1);
    auto unit = [&](auto uc, auto offc) __attribute__((always_inline)) {
      constexpr int u = decltype(uc)::value, OFF = decltype(offc)::value;
      if (u == NU - 2 && has_next) {
        __builtin_amdgcn_sched_barrier(0);
        if (NCW == 8) wait_mine(tau + 2 < ntiles);
        __builtin_amdgcn_s_barrier();                // B(tau + 1)
        if (NCW == 8 && tau + 3 < ntiles) issue_tile(tau + 3);
        __builtin_amdgcn_sched_barrier(0);
      }
      constexpr int s0 = (OFF + u) % 3, s2 = (OFF + u + 2) % 3;
      constexpr bool in_phase = u + 2 < NU;
      const bool pre = in_phase || has_next;
      const bool later = (u + 1 < NU) || has_next;
      if (later) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      constexpr int ia = 2 * (u & 1), ib = u >> 1;
      auto rd = [&](auto jc) __attribute__((always_inline)) {
        if (!pre) return;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (in_phase) frag(lc, std::integral_constant<int, u + 2>{}, s2, jc);
        else frag(ln, std::integral_constant<int, u + 2 - NU>{}, s2, jc);
        __builtin_amdgcn_sched_barrier(0);
      };
      acc[ia] = MFMA(fl[s0][0], xh[ib], acc[ia], 0, 0, 0); rd(J0{});
      acc[ia + 1] = MFMA(fl[s0][1], xh[ib], acc[ia + 1], 0, 0, 0); rd(J1{});
      acc[ia] = MFMA(fh[s0][0], xl[ib], acc[ia], 0, 0, 0); rd(J2{});
      acc[ia + 1] = MFMA(fh[s0][1], xl[ib], acc[ia + 1], 0, 0, 0); rd(J3{});
      acc[ia] = MFMA(fh[s0][0], xh[ib], acc[ia], 0, 0, 0);
      acc[ia + 1] = MFMA(fh[s0][1], xh[ib], acc[ia + 1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    };
    auto tile = [&](auto offc) __attribute__((always_inline)) {
      unit(std::integral_constant<int, 0>{}, offc); unit(std::integral_constant<int, 1>{}, offc);
      unit(std::integral_constant<int, 2>{}, offc); unit(std::integral_constant<int, 3>{}, offc);
      unit(std::integral_constant<int, 4>{}, offc); unit(std::integral_constant<int, 5>{}, offc);
      unit(std::integral_constant<int, 6>{}, offc); unit(std::integral_constant<int, 7>{}, offc);
    };
    auto valu = [&]() __attribute__((always_inline)) {
      if constexpr (VALU > 0) {
#pragma unroll
        for (int e = 0; e < VALU; ++e) {
          float x = vv[e] + acc[e & 3][e & 3] * 1e-30f;
          const float z = fabsf(x) * 0.70710678f;
          const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
          const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
          const float erfa = 1.0f - poly * __expf(-z * z);
          vv[e] = 0.5f * x * (1.0f + copysignf(erfa, x));
        }
      }
    };
    const bool early = SKEW == 1 && wave >= 4;
    if (early) valu();
    if (off == 0) tile(std::integral_constant<int, 0>{});
    else if (off == 1) tile(std::integral_constant<int, 1>{});
    else tile(std::integral_constant<int, 2>{});
    off = (off + NU) % 3;
    if (!early && SKEW != 2) valu();
    if constexpr (false) {       // the serial section of a head / hidden chunk: GELU-like work on VALU values per lane
#pragma unroll
      for (int e = 0; e < VALU; ++e) {
        float x = vv[e] + acc[e & 3][e & 3] * 1e-30f;
        const float z = fabsf(x) * 0.70710678f;
        const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
        const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
        const float erfa = 1.0f - poly * __expf(-z * z);
        vv[e] = 0.5f * x * (1.0f + copysignf(erfa, x));
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
  float r = s[0] + s[1] + s[2] + s[3];
  for (int e = 0; e < VALU; ++e) r += vv[e];
  out[blockIdx.x * 512 + tid] = r;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  unsigned char* w; float* out; unsigned long long* cyc;
  const int wtiles = 64, ntiles = 510;
  hipMalloc(&w, (size_t)(wtiles + 8) * SLOT); hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8 * 1024);
  hipMemset(w, 0, (size_t)(wtiles + 8) * SLOT);
  auto report = [&](const char* name, auto kern, int rows, int blocks) {
    const size_t smem = NS * SLOT;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), smem, 0, w, out, cyc, ntiles, wtiles);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), smem, 0, w, out, cyc, ntiles, wtiles);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[1024]; hipMemcpy(c, cyc, 8 * blocks, hipMemcpyDeviceToHost);
    double s = 0; for (int b = 0; b < blocks; ++b) s += c[b];
    const double per_tile = s / blocks / ntiles;
    // MFMA-pipe cycles per tile and SIMD: 48 MFMAs x 16 cycles per wave, NCW / 4 waves per SIMD
    const double mfma = 768.0 * rows / 64;
    printf("%-72s blocks %4d : %7.1f cyc per tile (%3d rows) = %6.1f cyc per 64 rows ; MFMA pipe %4.1f %% busy ; %.3f ms\n", name, blocks,
           per_tile, rows, per_tile * 64 / rows, 100.0 * mfma / per_tile, ms);
  };
  for (int blocks : {1, 256}) {
    report("4 compute + 4 loader waves (shipped structure), no VALU", kdual<4, 0>, 64, blocks);
    report("8 compute waves, DMA issued by them, no VALU", kdual<8, 0>, 128, blocks);
    report("4 compute + 4 loader waves, 16 GELU values per lane and tile", kdual<4, 16>, 64, blocks);
    report("8 compute waves, 16 GELU values per lane and tile", kdual<8, 16>, 128, blocks);
    report("8 compute waves, 16 GELU values, waves 4-7 SKEWED (VALU first)", kdual<8, 16, 1>, 128, blocks);
    report("4 compute + 4 loader waves, 16 GELU values per tile ON THE LOADER WAVES", kdual<4, 16, 2>, 64, blocks);
    report("4 compute + 4 loader waves, 32 GELU values per lane and tile", kdual<4, 32>, 64, blocks);
    report("4 compute + 4 loader waves, 32 GELU values per tile ON THE LOADER WAVES", kdual<4, 32, 2>, 64, blocks);
    report("8 compute waves, 32 GELU values per lane and tile", kdual<8, 32>, 128, blocks);
    report("8 compute waves, 32 GELU values, waves 4-7 SKEWED (VALU first)", kdual<8, 32, 1>, 128, blocks);
  }
  return 0;
}
