// Premise of a large-batch form of the ring kernels, measured before anything is built on it (round 4): TWO row tiles per compute
// wave -- every weight fragment read from LDS feeds the MFMAs of two 16-row tiles (4 reads : 12 MFMAs instead of 4 : 6), a 32 KB tile
// serves 128 rows of a workgroup instead of 64.  Same structure as kpipe<1> of proj_phase.hip (4 compute + 4 loader waves, 4-slot
// ring, one barrier per tile, reads two units ahead).
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 tools/ubench/two_row_tiles.hip -o /tmp/two_row_tiles && /tmp/two_row_tiles
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA __builtin_amdgcn_mfma_f32_16x16x32_bf16

constexpr int C = 128, SLOT = 256 * C, NS = 4, IPT = 8;   // 32 KB tiles, 8 DMA pieces per wave per tile (4 issuing waves)

template <int LOADERS, int RT>
__global__ __launch_bounds__(LOADERS ? 512 : 256) void kpipe_rt(const unsigned char* w, float* out, unsigned long long* cyc,
                                                             int ntiles, int wtiles, int unit_gap) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const bool loader = LOADERS && wave >= 4;
  const int iw = wave & 3;
  bf16x8 xh[RT][4], xl[RT][4];                       // RT row tiles of 16 rows per compute wave
  for (int t = 0; t < RT; ++t)
    for (int st = 0; st < 4; ++st)
      for (int e = 0; e < 8; ++e) { xh[t][st][e] = (__bf16)(float)(lane + e + st + t); xl[t][st][e] = (__bf16)(float)(lane - e - t); }
  for (int t = tid; t < NS * SLOT / 4; t += blockDim.x) ((float*)smem)[t] = 0.f;
  __syncthreads();
  unsigned long long t0 = 0, t1 = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  if (loader) {
    auto issue_tile = [&](int tau) {
      const unsigned char* tile = w + (int64_t)(tau % wtiles) * SLOT;
      unsigned char* slot = smem + (tau % NS) * SLOT;
#pragma unroll
      for (int q = 0; q < IPT; ++q) {
        const int inst = iw + 4 * q;
        __builtin_amdgcn_global_load_lds(tile + inst * 1024 + lane * 16,
                                         (__attribute__((address_space(3))) void*)(slot + inst * 1024), 16, 0, 0);
      }
    };
    issue_tile(0);
    issue_tile(1);
    for (int k = 0; k < ntiles; ++k) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // tile k landed (tile k+1 may be in flight)
      __builtin_amdgcn_s_barrier();
      issue_tile(k + 2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    int aP[4];
    for (int st = 0; st < 4; ++st) {
      const int lc = 4 * st + g;
      aP[st] = i * (4 * C) + ((lc & ~15) | ((lc & 15) ^ i)) * 16;
    }
    auto lds_read = [&](bf16x8& dst, const unsigned char* p) {
      const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)p;
      asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory");
    };
    f32x4 acc[RT][4];
    for (int t = 0; t < RT; ++t)
      for (int n = 0; n < 4; ++n) acc[t][n] = f32x4{0, 0, 0, 0};
    bf16x8 fh[3][2], fl[3][2];
    if (LOADERS) __builtin_amdgcn_s_barrier();             // B(0)
    {
      const unsigned char* slot = smem;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          lds_read(fh[u][q], slot + aP[u >> 1] + ((2 * (u & 1) + q) * 16 * 4 * C));
          lds_read(fl[u][q], slot + aP[u >> 1] + ((2 * (u & 1) + q) * 16 * 4 * C + 2 * C));
        }
    }
    for (int tau = 0; tau < ntiles; ++tau) {
      const unsigned char* cur = smem + (tau % NS) * SLOT;
      const unsigned char* nxt = smem + ((tau + 1) % NS) * SLOT;
      const bool last = tau + 1 == ntiles;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (u == 6 && LOADERS && !last) __builtin_amdgcn_s_barrier();   // B(tau+1)
        // sets rotate with the GLOBAL unit index; 8 % 3 = 2, so the set of unit u of tile tau is (2 tau + u) % 3
        const int un = (u + 2) & 7;
        const unsigned char* base = (u + 2 < 8 ? cur : nxt) + aP[un >> 1];
        const bool pre = (u + 2 < 8) || !last;
        // the register sets must be compile-time: unroll the three phases of tau % 3 via a switch-free trick
        // (rotate the arrays by value at the end of each tile instead)
        const int s0 = u % 3, s2 = (u + 2) % 3;
        if (pre || u < 7) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (last && u == 6) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        if (last && u == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const int st = u >> 1, f0 = 2 * (u & 1);
        for (int t = 0; t < RT; ++t) acc[t][f0] = MFMA(fl[s0][0], xh[t][st], acc[t][f0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (pre) lds_read(fh[s2][0], base + ((2 * (un & 1)) * 16 * 4 * C));
        __builtin_amdgcn_sched_barrier(0);
        for (int t = 0; t < RT; ++t) acc[t][f0 + 1] = MFMA(fl[s0][1], xh[t][st], acc[t][f0 + 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (pre) lds_read(fl[s2][0], base + ((2 * (un & 1)) * 16 * 4 * C + 2 * C));
        __builtin_amdgcn_sched_barrier(0);
        for (int t = 0; t < RT; ++t) acc[t][f0] = MFMA(fh[s0][0], xl[t][st], acc[t][f0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (pre) lds_read(fh[s2][1], base + ((2 * (un & 1) + 1) * 16 * 4 * C));
        __builtin_amdgcn_sched_barrier(0);
        for (int t = 0; t < RT; ++t) acc[t][f0 + 1] = MFMA(fh[s0][1], xl[t][st], acc[t][f0 + 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (pre) lds_read(fl[s2][1], base + ((2 * (un & 1) + 1) * 16 * 4 * C + 2 * C));
        __builtin_amdgcn_sched_barrier(0);
        for (int t = 0; t < RT; ++t) acc[t][f0] = MFMA(fh[s0][0], xh[t][st], acc[t][f0], 0, 0, 0);
        for (int t = 0; t < RT; ++t) acc[t][f0 + 1] = MFMA(fh[s0][1], xh[t][st], acc[t][f0 + 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // units 8, 9 of this tile are units 0, 1 of the next: they sit in sets 8%3 = 2 and 9%3 = 0 -> move to 0, 1
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        bf16x8 a = fh[2][q], b = fl[2][q];
        fh[1][q] = fh[0][q]; fl[1][q] = fl[0][q];
        fh[0][q] = a; fl[0][q] = b;
      }
    }
    f32x4 s = f32x4{0, 0, 0, 0};
    for (int t = 0; t < RT; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * 256 + (tid & 255)] = s[0] + s[1] + s[2] + s[3];
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}


int main() {
  unsigned char* w; float* out; unsigned long long* cyc;
  const int wtiles = 64;
  (void)hipMalloc(&w, (size_t)(wtiles + 8) * SLOT); (void)hipMalloc(&out, 1 << 22); (void)hipMalloc(&cyc, 8 * 1024);
  (void)hipMemset(w, 0, (size_t)(wtiles + 8) * SLOT);
  auto report = [&](const char* name, auto kern, int threads, int blocks, int ntiles, int rows) {
    const size_t smem = NS * SLOT;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), smem, 0, w, out, cyc, ntiles, wtiles, 0);
    (void)hipDeviceSynchronize();
    static unsigned long long c[1024]; (void)hipMemcpy(c, cyc, 8 * blocks, hipMemcpyDeviceToHost);
    double s = 0; for (int b = 0; b < blocks; ++b) s += c[b];
    const double per_tile = s / blocks / ntiles;
    printf("%-58s blocks %4d : %7.1f cycles per 32 KB tile for %3d rows = %6.2f cycles per tile and 16 rows\n", name, blocks, per_tile, rows,
           per_tile / (rows / 16));
  };
  for (int blocks : {1, 256}) {
    report("static LDS, 1 row tile / wave (P0)", kpipe_rt<0, 1>, 256, blocks, 510, 64);
    report("static LDS, 2 row tiles / wave", kpipe_rt<0, 2>, 256, blocks, 510, 128);
    report("ring + 4 loader waves, 1 row tile / wave (P1, shipped)", kpipe_rt<1, 1>, 512, blocks, 510, 64);
    report("ring + 4 loader waves, 2 row tiles / wave", kpipe_rt<1, 2>, 512, blocks, 510, 128);
    report("ring + 4 loader waves, 3 row tiles / wave", kpipe_rt<1, 3>, 512, blocks, 510, 192);
    report("ring + 4 loader waves, 4 row tiles / wave", kpipe_rt<1, 4>, 512, blocks, 510, 256);
  }
  return 0;
}
