// Round 5: the large-batch form of the ring kernels has NO loader waves (two row tiles per compute wave need the registers of the
// whole SIMD: 4 waves x up to 512).  What does it cost the compute waves to issue the LDS-DMA stream themselves?
//   DMA = 0: 4 compute + 4 loader waves (the shipped structure; two_row_tiles.hip)
//   DMA = 1: 4 compute waves, all 8 pieces of tile k + 2 issued in a burst behind barrier B(k)
//   DMA = 2: 4 compute waves, ONE piece per unit (8 units per tile), the first behind B(k)
//   DMA = 3: as 2, plus the tile's descriptor by a scalar load behind the barrier and s_waitcnt lgkmcnt(0) in front of it
//            (the compute waves' counted lgkmcnt waits cannot cover a scalar load that returns out of order)
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/two_row_own_dma.hip -o /tmp/two_row_own_dma && /tmp/two_row_own_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA __builtin_amdgcn_mfma_f32_16x16x32_bf16
// Round 6: the tables of round 5 were CYCLES on an all-zero weight stream and small-integer rows.  MI355X_MICROARCH.md (DVFS give-back,
// items 1 and 7): zero or trivial operands rank two MFMA shapes by cycles and miss the clock the chip holds under load -- on random
// data a 16x16x32 loop delivered ~1.15x the FLOP/s of the 32x32x16 loop at equal cycles per FLOP.  `RANDOM_DATA` (main: second pass)
// fills the weight stream with random finite bf16 pairs and the row operands with a per-lane pseudo-random sequence, and every
// variant is also timed in WALL time (hipEvents over >= 0.2 s of back-to-back launches, every CU busy): ns per 32 KB tile and 16 rows,
// and the clock the kernel ran at (in-kernel cycles / wall).
__device__ int g_random_rows = 0;
__device__ __forceinline__ float row_value(int a, int b) {
  if (!g_random_rows) return (float)(a + b);
  unsigned h = (unsigned)(a * 2654435761u) ^ (unsigned)(b * 40503u + 0x9e3779b9u);
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  return (float)(int)(h & 0xffffu) * (1.0f / 32768.0f) - 1.0f;      // [-1, 1)
}

constexpr int C = 128, SLOT = 256 * C, NS = 4, IPT = 8;   // 32 KB tiles, 8 DMA pieces per wave per tile (4 issuing waves)

// ACC4 (RT = 1 only): the six MFMAs of a unit go to FOUR accumulators instead of two (the lo x hi product of each feature tile to a
// second accumulator): is the unit's dependent accumulation (every other MFMA on the same registers) what the one-row-tile form waits for?
template <int DMA, int RT, int ACC4 = 0>
__global__ __launch_bounds__(DMA == 0 ? 512 : 256) void kpipe_rt(const unsigned char* w, float* out, unsigned long long* cyc,
                                                              int ntiles, int wtiles, const unsigned* desc) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  const bool loader = DMA == 0 && wave >= 4;
  const int iw = wave & 3;
  bf16x8 xh[RT][4], xl[RT][4];                       // RT row tiles of 16 rows per compute wave
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int e = 0; e < 8; ++e) { xh[t][st][e] = (__bf16)row_value(lane + e + st + t, 7 * t + e); xl[t][st][e] = (__bf16)row_value(lane - e - t, 13 * st + 1); }
  for (int t = tid; t < NS * SLOT / 4; t += blockDim.x) ((float*)smem)[t] = 0.f;
  __syncthreads();
  unsigned long long t0 = 0, t1 = 0;
  unsigned long long r0 = 0, r1 = 0;
  asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0), "=s"(t0)::"memory");
  const unsigned voff = (unsigned)(iw * 1024 + lane * 16);
  auto issue_piece = [&](int tau, unsigned d, int q) {
    const unsigned char* tile = w + (int64_t)d * SLOT;
    unsigned char* slot = smem + (tau & (NS - 1)) * SLOT + iw * 1024;
    __builtin_amdgcn_global_load_lds(tile + voff + q * 4096, (__attribute__((address_space(3))) void*)(slot + q * 4096), 16, 0, 0);
  };
  auto issue_tile = [&](int tau, unsigned d) {
#pragma unroll
    for (int q = 0; q < IPT; ++q) issue_piece(tau, d, q);
  };
  if (loader) {
    issue_tile(0, 0);
    issue_tile(1, 1);
    for (int k = 0; k < ntiles; ++k) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // tile k landed (tile k+1 may be in flight)
      __builtin_amdgcn_s_barrier();
      issue_tile(k + 2, (unsigned)((k + 2) % wtiles));
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
    f32x4 accb[4];
    for (int t = 0; t < RT; ++t)
      for (int n = 0; n < 4; ++n) { acc[t][n] = f32x4{0, 0, 0, 0}; accb[n] = f32x4{0, 0, 0, 0}; }
    bf16x8 fh[3][2], fl[3][2];
    unsigned dcur = 2u % (unsigned)wtiles;                 // descriptor of the tile being issued (k + 2)
    unsigned dnext = 3u % (unsigned)wtiles;
    if (DMA > 0) {
      issue_tile(0, 0);
      issue_tile(1, 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                          // B(0)
    if (DMA == 1) issue_tile(2, dcur);
    int pend_tau = 2, pend_q = DMA >= 2 ? 0 : IPT;         // DMA >= 2: pieces of tile pend_tau still to issue
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
        if (u == 6 && !last) {                             // B(tau+1), then the stream moves on to tile tau + 3
          if (DMA > 0) {
            if (DMA == 3) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
          }
          __builtin_amdgcn_s_barrier();
          if (DMA == 1) issue_tile(tau + 3, (unsigned)((tau + 3) % wtiles));
          if (DMA >= 2) { pend_tau = tau + 3; pend_q = 0; dcur = dnext; }
          if (DMA == 3) {
            asm volatile("s_load_dword %0, %1, 0x0" : "=s"(dnext) : "s"(desc + ((tau + 4) % wtiles)) : "memory");
          } else {
            dnext = (unsigned)((tau + 4) % wtiles);
          }
        }
        // sets rotate with the GLOBAL unit index; 8 % 3 = 2, so the set of unit u of tile tau is (2 tau + u) % 3
        const int un = (u + 2) & 7;
        const unsigned char* base = (u + 2 < 8 ? cur : nxt) + aP[un >> 1];
        const bool pre = (u + 2 < 8) || !last;
        const int s0 = u % 3, s2 = (u + 2) % 3;
        if (!(DMA == 3 && u == 6 && !last)) {
          if (pre || u < 7) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
          else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (last && u == 6) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        if (last && u == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const int st = u >> 1, f0 = 2 * (u & 1);
        if (ACC4) accb[f0] = MFMA(fl[s0][0], xh[0][st], accb[f0], 0, 0, 0);
        else for (int t = 0; t < RT; ++t) acc[t][f0] = MFMA(fl[s0][0], xh[t][st], acc[t][f0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (pre) lds_read(fh[s2][0], base + ((2 * (un & 1)) * 16 * 4 * C));
        __builtin_amdgcn_sched_barrier(0);
        if (ACC4) accb[f0 + 1] = MFMA(fl[s0][1], xh[0][st], accb[f0 + 1], 0, 0, 0);
        else for (int t = 0; t < RT; ++t) acc[t][f0 + 1] = MFMA(fl[s0][1], xh[t][st], acc[t][f0 + 1], 0, 0, 0);
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
        if (DMA >= 2 && pend_q < IPT) {                    // one piece of the stream per unit, in the shadow of the MFMAs
          __builtin_amdgcn_sched_barrier(0);
          issue_piece(pend_tau, dcur, pend_q);
          ++pend_q;
          __builtin_amdgcn_sched_barrier(0);
        }
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f32x4 s = f32x4{0, 0, 0, 0};
    for (int t = 0; t < RT; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    s += accb[0] + accb[1] + accb[2] + accb[3];
    out[blockIdx.x * 256 + (tid & 255)] = s[0] + s[1] + s[2] + s[3];
  }
  asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1), "=s"(t1)::"memory");
  if (tid == 0) { cyc[blockIdx.x] = t1 - t0; cyc[512 + blockIdx.x] = r1 - r0; }
}

// The same pipeline with v_mfma_f32_32x32x16_bf16 on ONE 32-row tile per wave (the rows of two 16-row tiles): per unit still 4
// fragment reads (32 features x 16 k each: k-half x hi / lo plane) but 6 MFMAs of 32 cycles instead of 12 of 16 -- half the MFMA
// issues per FLOP, and an MFMA holds the SIMD's vector issue for 8 of its 32 cycles instead of 8 of 16 (MI355X_MICROARCH.md).
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32 __builtin_amdgcn_mfma_f32_32x32x16_bf16
template <int DMA>
__global__ __launch_bounds__(DMA == 0 ? 512 : 256) void kpipe_m32(const unsigned char* w, float* out, unsigned long long* cyc,
                                                               int ntiles, int wtiles, const unsigned* desc) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = DMA == 0 && wave >= 4;
  const int iw = wave & 3;
  bf16x8 xh[4][2], xl[4][2];                         // k32-step, k16-half: the 32 rows' operand (B)
#pragma unroll
  for (int st = 0; st < 4; ++st)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int e = 0; e < 8; ++e) { xh[st][hf][e] = (__bf16)row_value(lane + e + st + hf, 7 * hf + e); xl[st][hf][e] = (__bf16)row_value(lane - e - hf, 13 * st + 1); }
  for (int t = tid; t < NS * SLOT / 4; t += blockDim.x) ((float*)smem)[t] = 0.f;
  __syncthreads();
  unsigned long long t0 = 0, t1 = 0;
  unsigned long long r0 = 0, r1 = 0;
  asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0), "=s"(t0)::"memory");
  const unsigned voff = (unsigned)(iw * 1024 + lane * 16);
  auto issue_piece = [&](int tau, unsigned d, int q) {
    const unsigned char* tile = w + (int64_t)d * SLOT;
    unsigned char* slot = smem + (tau & (NS - 1)) * SLOT + iw * 1024;
    __builtin_amdgcn_global_load_lds(tile + voff + q * 4096, (__attribute__((address_space(3))) void*)(slot + q * 4096), 16, 0, 0);
  };
  auto issue_tile = [&](int tau, unsigned d) {
#pragma unroll
    for (int q = 0; q < IPT; ++q) issue_piece(tau, d, q);
  };
  if (loader) {
    issue_tile(0, 0);
    issue_tile(1, 1);
    for (int k = 0; k < ntiles; ++k) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      issue_tile(k + 2, (unsigned)((k + 2) % wtiles));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    // fragment-ordered tile: fragment (unit u, j) is 1 KB at (4 u + j) * 1024, lane * 16 inside (conflict-free, linear DMA)
    auto lds_read = [&](bf16x8& dst, const unsigned char* p) {
      const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)p;
      asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory");
    };
    f32x16 acc[2];
    for (int n = 0; n < 2; ++n)
      for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    bf16x8 fr[3][4];                                 // three sets of a unit's 4 fragments: (k16 half 0 / 1) x (lo, hi)
    unsigned dcur = 2u % (unsigned)wtiles, dnext = 3u % (unsigned)wtiles;
    if (DMA > 0) {
      issue_tile(0, 0);
      issue_tile(1, 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                          // B(0)
    if (DMA == 1) issue_tile(2, dcur);
    int pend_tau = 2, pend_q = DMA >= 2 ? 0 : IPT;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) lds_read(fr[u][j], smem + (4 * u + j) * 1024 + lane * 16);
    for (int tau = 0; tau < ntiles; ++tau) {
      const unsigned char* cur = smem + (tau % NS) * SLOT;
      const unsigned char* nxt = smem + ((tau + 1) % NS) * SLOT;
      const bool last = tau + 1 == ntiles;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (u == 6 && !last) {
          if (DMA > 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          if (DMA == 1) issue_tile(tau + 3, (unsigned)((tau + 3) % wtiles));
          if (DMA >= 2) { pend_tau = tau + 3; pend_q = 0; dcur = dnext; }
          dnext = (unsigned)((tau + 4) % wtiles);
        }
        const int un = (u + 2) & 7;
        const unsigned char* base = (u + 2 < 8 ? cur : nxt) + un * 4096 + lane * 16;
        const bool pre = (u + 2 < 8) || !last;
        const int s0 = u % 3, s2 = (u + 2) % 3;
        if (pre || u < 7) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (last && u == 6) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        if (last && u == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const int st = u >> 1, f0 = u & 1;             // unit = (k32-step, feature half): both k16 halves of the step
        acc[f0] = MFMA32(fr[s0][0], xh[st][0], acc[f0], 0, 0, 0);      // lo x hi, k16 half 0
        __builtin_amdgcn_sched_barrier(0);
        if (pre) lds_read(fr[s2][0], base);
        __builtin_amdgcn_sched_barrier(0);
        acc[f0] = MFMA32(fr[s0][2], xh[st][1], acc[f0], 0, 0, 0);      // lo x hi, half 1
        __builtin_amdgcn_sched_barrier(0);
        if (pre) lds_read(fr[s2][1], base + 1024);
        __builtin_amdgcn_sched_barrier(0);
        acc[f0] = MFMA32(fr[s0][1], xl[st][0], acc[f0], 0, 0, 0);      // hi x lo
        __builtin_amdgcn_sched_barrier(0);
        if (pre) lds_read(fr[s2][2], base + 2048);
        __builtin_amdgcn_sched_barrier(0);
        acc[f0] = MFMA32(fr[s0][3], xl[st][1], acc[f0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (pre) lds_read(fr[s2][3], base + 3072);
        __builtin_amdgcn_sched_barrier(0);
        acc[f0] = MFMA32(fr[s0][1], xh[st][0], acc[f0], 0, 0, 0);      // hi x hi
        if (DMA >= 2 && pend_q < IPT) {
          __builtin_amdgcn_sched_barrier(0);
          issue_piece(pend_tau, dcur, pend_q);
          ++pend_q;
          __builtin_amdgcn_sched_barrier(0);
        }
        acc[f0] = MFMA32(fr[s0][3], xh[st][1], acc[f0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bf16x8 a = fr[2][j];
        fr[1][j] = fr[0][j];
        fr[0][j] = a;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int n = 0; n < 2; ++n)
      for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * 256 + (tid & 255)] = s;
  }
  asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1), "=s"(t1)::"memory");
  if (tid == 0) { cyc[blockIdx.x] = t1 - t0; cyc[512 + blockIdx.x] = r1 - r0; }
}

int main() {
  unsigned char* w; float* out; unsigned long long* cyc; unsigned* desc;
  const int wtiles = 64;
  (void)hipMalloc(&w, (size_t)(wtiles + 8) * SLOT); (void)hipMalloc(&out, 1 << 22); (void)hipMalloc(&cyc, 8 * 1024); (void)hipMemset(cyc, 0, 8 * 1024);
  (void)hipMalloc(&desc, 4 * wtiles);
  unsigned hd[64]; for (int k = 0; k < wtiles; ++k) hd[k] = (unsigned)k;
  (void)hipMemcpy(desc, hd, 4 * wtiles, hipMemcpyHostToDevice);
  (void)hipMemset(w, 0, (size_t)(wtiles + 8) * SLOT);
  auto report = [&](const char* name, auto kern, int threads, int blocks, int ntiles, int rows) {
    const size_t smem = NS * SLOT;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), smem, 0, w, out, cyc, ntiles, wtiles, desc);
    (void)hipDeviceSynchronize();
    static unsigned long long c[1024]; (void)hipMemcpy(c, cyc, 8 * blocks, hipMemcpyDeviceToHost);
    double s = 0; for (int b = 0; b < blocks; ++b) s += c[b];
    const double per_tile = s / blocks / ntiles;
    printf("%-64s blocks %4d : %7.1f cycles per 32 KB tile for %3d rows = %6.2f per 16 rows ; MFMA pipe %4.1f %% busy\n", name, blocks,
           per_tile, rows, per_tile / (rows / 16), 100.0 * (rows / 64) * 768.0 / per_tile);
  };
  auto wall = [&](const char* name, auto kern, int threads, int ntiles, int rows) {
    // every CU busy (256 workgroups), back-to-back launches for >= 0.2 s, hipEvent wall time; cycles from the kernel's own stamps
    const int blocks = 256;
    const size_t smem = NS * SLOT;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 50; ++rep) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), smem, 0, w, out, cyc, ntiles, wtiles, desc);
    (void)hipDeviceSynchronize();
    const int reps = 400;
    (void)hipEventRecord(e0, 0);
    for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), smem, 0, w, out, cyc, ntiles, wtiles, desc);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long c[1024]; (void)hipMemcpy(c, cyc, 8 * 1024, hipMemcpyDeviceToHost);
    double s = 0, sr = 0; for (int b = 0; b < blocks; ++b) { s += c[b]; sr += c[512 + b]; }
    // in-kernel: shader cycles (s_memtime) and 100 MHz ticks (s_memrealtime) of every workgroup's timed region; wall: hipEvents over
    // the back-to-back launches (includes each launch's LDS fill, drain and the gap to the next launch)
    const double cyc_launch = s / blocks, ns_kernel = 10.0 * sr / blocks, ns_launch = 1e6 * ms / reps;
    printf("%-58s : %7.1f cycles = %7.1f ns per tile in the kernel (clock %4.2f GHz) ; %6.2f ns per 16 rows ; %6.1f TFLOP/s in the kernel ; "
           "wall %7.1f ns per tile\n", name, cyc_launch / ntiles, ns_kernel / ntiles, cyc_launch / ns_kernel, ns_kernel / ntiles / (rows / 16),
           256.0 * rows * 8192.0 * 2 * 3 * ntiles / (ns_kernel * 1e-9) / 1e12, ns_launch / ntiles);
  };
  for (int random = 0; random < 2; ++random) {
    if (random) {
      // random finite bf16 pairs: exponent field 0x3c..0x3f, random sign and mantissa
      const size_t n16 = (size_t)(wtiles + 8) * SLOT / 2;
      unsigned short* hw = (unsigned short*)malloc(n16 * 2);
      unsigned st = 12345u;
      for (size_t k = 0; k < n16; ++k) { st = st * 1664525u + 1013904223u; hw[k] = (unsigned short)(((st >> 16) & 0x81ffu) | (0x3c00u + ((st >> 8) & 0x0300u))); }
      (void)hipMemcpy(w, hw, n16 * 2, hipMemcpyHostToDevice);
      free(hw);
      int one = 1; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_random_rows), &one, sizeof one);
    }
    printf("== WALL time, 256 workgroups, %s\n", random ? "RANDOM weight stream and rows" : "all-zero weight stream, small-integer rows (round 5's setting)");
    wall("4 compute + 4 loader waves, 1 row tile / wave (shipped)", kpipe_rt<0, 1>, 512, 510, 64);
    wall("4 compute + 4 loader waves, 2 row tiles / wave", kpipe_rt<0, 2>, 512, 510, 128);
    wall("4 compute waves, own DMA one piece per unit, 2 row tiles", kpipe_rt<2, 2>, 256, 510, 128);
    wall("32x32x16 MFMAs, one 32-row tile / wave, 4 loader waves", kpipe_m32<0>, 512, 510, 128);
    wall("32x32x16 MFMAs, one 32-row tile / wave, own DMA per unit", kpipe_m32<2>, 256, 510, 128);
    wall("4 compute + 4 loader waves, 1 row tile / wave (again)", kpipe_rt<0, 1>, 512, 510, 64);
  }
  if (getenv("WALL_ONLY")) return 0;
  (void)hipMemset(w, 0, (size_t)(wtiles + 8) * SLOT);
  { int zero = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_random_rows), &zero, sizeof zero); }
  for (int blocks : {1, 256}) {
    report("4 compute + 4 loader waves, 1 row tile / wave (shipped)", kpipe_rt<0, 1>, 512, blocks, 510, 64);
    report("  ... 1 row tile, four accumulators per unit instead of two", kpipe_rt<0, 1, 1>, 512, blocks, 510, 64);
    report("4 compute + 4 loader waves, 2 row tiles / wave", kpipe_rt<0, 2>, 512, blocks, 510, 128);
    report("4 compute waves, own DMA as a burst behind the barrier, 1 tile", kpipe_rt<1, 1>, 256, blocks, 510, 64);
    report("4 compute waves, own DMA as a burst behind the barrier, 2 tiles", kpipe_rt<1, 2>, 256, blocks, 510, 128);
    report("4 compute waves, own DMA one piece per unit, 1 row tile", kpipe_rt<2, 1>, 256, blocks, 510, 64);
    report("4 compute waves, own DMA one piece per unit, 2 row tiles", kpipe_rt<2, 2>, 256, blocks, 510, 128);
    report("  ... + descriptor by s_load, lgkmcnt(0) at the barrier, 2 tiles", kpipe_rt<3, 2>, 256, blocks, 510, 128);
    report("32x32x16 MFMAs, one 32-row tile / wave, 4 loader waves", kpipe_m32<0>, 512, blocks, 510, 128);
    report("32x32x16 MFMAs, one 32-row tile / wave, own DMA burst", kpipe_m32<1>, 256, blocks, 510, 128);
    report("32x32x16 MFMAs, one 32-row tile / wave, own DMA per unit", kpipe_m32<2>, 256, blocks, 510, 128);
    report("4 compute waves, own DMA one piece per unit, 3 row tiles", kpipe_rt<2, 3>, 256, blocks, 510, 192);
    report("4 compute waves, own DMA one piece per unit, 4 row tiles", kpipe_rt<2, 4>, 256, blocks, 510, 256);
  }
  return 0;
}
