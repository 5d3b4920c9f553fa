// Micro-benchmark of the fused-block inner structure on gfx950: a 4-slot LDS ring of 32 KB weight tiles filled by
// LDS-DMA from an L2-resident stream, consumed by 4 waves (one per SIMD) that each run 48 split-bf16 MFMAs
// (16x16x32) per tile with fragments read by ds_read_b128.  Variants move the DMA issue around:
//   0  burst issue by the consumers right after the tile barrier (what k_tblock does)
//   1  consumers issue one DMA piece after every 6-MFMA unit
//   2  four extra loader waves issue all DMAs (8 waves per workgroup)
//   3  no DMA at all (static LDS): the MFMA + ds_read phase alone
//   4  no DMA, no LDS reads (register operands): MFMA issue alone
// Build & run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/proj_phase.hip -o /tmp/proj_phase && /tmp/proj_phase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA __builtin_amdgcn_mfma_f32_16x16x32_bf16

constexpr int C = 128, SLOT = 256 * C, NS = 4, IPT = 8;   // 32 KB tiles, 8 DMA pieces per wave per tile (4 issuing waves)

template <int V, int IL = 0>
__global__ __launch_bounds__(V == 2 ? 512 : 256) void kproj(const unsigned char* w, float* out, unsigned long long* cyc,
                                                            int ntiles, int wtiles, int unit_gap) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const bool loader = (V == 2) && wave >= 4;
  const int iw = wave & 3;
  bf16x8 xh[4], xl[4];
  for (int st = 0; st < 4; ++st)
    for (int e = 0; e < 8; ++e) { xh[st][e] = (__bf16)(float)(lane + e + st); xl[st][e] = (__bf16)(float)(lane - e); }
  for (int t = tid; t < NS * SLOT / 4; t += blockDim.x) ((float*)smem)[t] = 0.f;
  __syncthreads();

  auto issue_piece = [&](int tau, int q) {
    const unsigned char* tile = w + (int64_t)(tau % wtiles) * SLOT;
    unsigned char* slot = smem + (tau % NS) * SLOT;
    const int inst = iw + 4 * q;
    __builtin_amdgcn_global_load_lds(tile + inst * 1024 + lane * 16,
                                     (__attribute__((address_space(3))) void*)(slot + inst * 1024), 16, 0, 0);
  };
  auto issue_tile = [&](int tau) {
#pragma unroll
    for (int q = 0; q < IPT; ++q) issue_piece(tau, q);
  };
  int aP[4];
  for (int st = 0; st < 4; ++st) {
    const int lc = 4 * st + g;
    aP[st] = i * (4 * C) + ((lc & ~15) | ((lc & 15) ^ i)) * 16;
  }
  auto lds_read = [&](bf16x8& dst, const unsigned char* p) {
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)p;
    asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory");
  };
  auto lgkm_wait = [&](int pending) {
    if (pending >= 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    else if (pending >= 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  f32x4 acc[4];
  for (int n = 0; n < 4; ++n) acc[n] = f32x4{0, 0, 0, 0};

  if (V <= 2 && (V != 2 || loader)) {
    for (int t = 0; t < NS - 1; ++t) issue_tile(t);
  }
  unsigned long long t0 = 0, t1 = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  if (loader) {
    for (int tau = 0; tau < ntiles; ++tau) {
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      issue_tile(tau + NS - 1);
    }
  } else {
    for (int tau = 0; tau < ntiles; ++tau) {
      if (V <= 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      if (V <= 2) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      if (V == 0) issue_tile(tau + NS - 1);
      const unsigned char* slot = smem + (tau % NS) * SLOT;
      if (V == 4) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int st = u >> 1, f0 = 2 * (u & 1);
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[f0 + q] = MFMA(xl[(st + r) & 3], xh[st], acc[f0 + q], 0, 0, 0);
        }
        continue;
      }
      bf16x8 fh[3][2], fl[3][2];
      auto load = [&](int u, int set) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          lds_read(fh[set][q], slot + aP[u >> 1] + ((2 * (u & 1) + q) * 16 * 4 * C));
          lds_read(fl[set][q], slot + aP[u >> 1] + ((2 * (u & 1) + q) * 16 * 4 * C + 2 * C));
        }
      };
      load(0, 0);
      load(1, 1);
      if (IL == 0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (u + 2 < 8) load(u + 2, (u + 2) % 3);
          lgkm_wait(4 * min(2, 7 - u));
          const int st = u >> 1, f0 = 2 * (u & 1);
#pragma unroll
          for (int q = 0; q < 2; ++q) acc[f0 + q] = MFMA(fl[u % 3][q], xh[st], acc[f0 + q], 0, 0, 0);
#pragma unroll
          for (int q = 0; q < 2; ++q) acc[f0 + q] = MFMA(fh[u % 3][q], xl[st], acc[f0 + q], 0, 0, 0);
          if (V == 1 && unit_gap == 0) issue_piece(tau + NS - 1, u);
#pragma unroll
          for (int q = 0; q < 2; ++q) acc[f0 + q] = MFMA(fh[u % 3][q], xh[st], acc[f0 + q], 0, 0, 0);
          if (V == 1 && unit_gap != 0) issue_piece(tau + NS - 1, u);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        // interleaved: the 4 fragment reads of unit u+2 are issued one by one behind the first MFMAs of unit u
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          lgkm_wait(u < 7 ? 4 : 0);   // unit u's fragments landed; unit u+1's (4 reads) may still be in flight
          const int st = u >> 1, f0 = 2 * (u & 1);
          const int un = u + 2, set = un % 3;
          const unsigned char* base = slot + aP[(un >> 1) & 3];
          acc[f0] = MFMA(fl[u % 3][0], xh[st], acc[f0], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (un < 8) lds_read(fh[set][0], base + ((2 * (un & 1)) * 16 * 4 * C));
          __builtin_amdgcn_sched_barrier(0);
          acc[f0 + 1] = MFMA(fl[u % 3][1], xh[st], acc[f0 + 1], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (un < 8) lds_read(fl[set][0], base + ((2 * (un & 1)) * 16 * 4 * C + 2 * C));
          __builtin_amdgcn_sched_barrier(0);
          acc[f0] = MFMA(fh[u % 3][0], xl[st], acc[f0], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (un < 8) lds_read(fh[set][1], base + ((2 * (un & 1) + 1) * 16 * 4 * C));
          __builtin_amdgcn_sched_barrier(0);
          acc[f0 + 1] = MFMA(fh[u % 3][1], xl[st], acc[f0 + 1], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (un < 8) lds_read(fl[set][1], base + ((2 * (un & 1) + 1) * 16 * 4 * C + 2 * C));
          __builtin_amdgcn_sched_barrier(0);
          acc[f0] = MFMA(fh[u % 3][0], xh[st], acc[f0], 0, 0, 0);
          acc[f0 + 1] = MFMA(fh[u % 3][1], xh[st], acc[f0 + 1], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (!loader) {
    f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
    out[blockIdx.x * 256 + (tid & 255)] = s[0] + s[1] + s[2] + s[3];
  }
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

// Continuous pipeline: fragment reads run two units ahead ACROSS tile boundaries.  The barrier that publishes tile
// k+1 sits before unit 6 of tile k (the first unit whose prefetch touches tile k+1); after it the loaders refill
// the slot of tile k-1 with tile k+3... (issue distance 2 tiles, 4 slots).  LOADERS = 0: static LDS, no DMA.
template <int LOADERS>
__global__ __launch_bounds__(LOADERS ? 512 : 256) void kpipe(const unsigned char* w, float* out, unsigned long long* cyc,
                                                             int ntiles, int wtiles, int unit_gap) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const bool loader = LOADERS && wave >= 4;
  const int iw = wave & 3;
  bf16x8 xh[4], xl[4];
  for (int st = 0; st < 4; ++st)
    for (int e = 0; e < 8; ++e) { xh[st][e] = (__bf16)(float)(lane + e + st); xl[st][e] = (__bf16)(float)(lane - e); }
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
    f32x4 acc[4];
    for (int n = 0; n < 4; ++n) acc[n] = f32x4{0, 0, 0, 0};
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
        acc[f0] = MFMA(fl[s0][0], xh[st], acc[f0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (pre) lds_read(fh[s2][0], base + ((2 * (un & 1)) * 16 * 4 * C));
        __builtin_amdgcn_sched_barrier(0);
        acc[f0 + 1] = MFMA(fl[s0][1], xh[st], acc[f0 + 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (pre) lds_read(fl[s2][0], base + ((2 * (un & 1)) * 16 * 4 * C + 2 * C));
        __builtin_amdgcn_sched_barrier(0);
        acc[f0] = MFMA(fh[s0][0], xl[st], acc[f0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (pre) lds_read(fh[s2][1], base + ((2 * (un & 1) + 1) * 16 * 4 * C));
        __builtin_amdgcn_sched_barrier(0);
        acc[f0 + 1] = MFMA(fh[s0][1], xl[st], acc[f0 + 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (pre) lds_read(fl[s2][1], base + ((2 * (un & 1) + 1) * 16 * 4 * C + 2 * C));
        __builtin_amdgcn_sched_barrier(0);
        acc[f0] = MFMA(fh[s0][0], xh[st], acc[f0], 0, 0, 0);
        acc[f0 + 1] = MFMA(fh[s0][1], xh[st], acc[f0 + 1], 0, 0, 0);
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
    f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
    out[blockIdx.x * 256 + (tid & 255)] = s[0] + s[1] + s[2] + s[3];
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

// Ring-geometry sweep for the loader-wave pipeline: TILE bytes per tile, NSLOT slots, UPT units (4 reads + 6 MFMAs)
// consumed per tile per wave, issue distance DIST tiles (tile k+DIST is issued after barrier B(k); needs
// NSLOT >= DIST + 2).  The tile loop is unrolled by 3 so that the fragment-set rotation is static.
// STAGE = 1: the loader waves stage through REGISTERS (global_load_dwordx4 -> ds_write_b128, two tiles in flight in two register
// sets) instead of LDS-DMA: tests whether the DMA's LDS write (vs a full-rate ds_write_b128) is what the LDS pipe pays for.
template <int TILE, int NSLOT, int UPT, int DIST, int NODMA = 0, int RDPAT = 0, int STAGE = 0>
__global__ __launch_bounds__(512) void kring(const unsigned char* w, float* out, unsigned long long* cyc,
                                             int ntiles, int wtiles, int unit_gap) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr int PIECES = TILE / 4096;                   // per loader wave
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const bool loader = wave >= 4;
  const int iw = wave & 3;
  for (int t = tid; t < NSLOT * TILE / 4; t += blockDim.x) ((float*)smem)[t] = 0.f;
  __syncthreads();
  unsigned long long t0 = 0, t1 = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  if (loader) {
    auto issue_tile = [&](int tau) {
      const unsigned char* tile = w + (int64_t)(tau % wtiles) * TILE;
      unsigned char* slot = smem + (tau % NSLOT) * TILE;
#pragma unroll
      for (int q = 0; q < PIECES; ++q) {
        const int inst = iw + 4 * q;
        __builtin_amdgcn_global_load_lds(tile + inst * 1024 + lane * 16,
                                         (__attribute__((address_space(3))) void*)(slot + inst * 1024), 16, 0, 0);
      }
    };
    if constexpr (STAGE == 1) {
      static_assert(DIST == 2 && PIECES == 8, "register staging: two tiles of 8 pieces per wave in flight");
      uint4 r0[PIECES], r1[PIECES];
      auto load_tile = [&](int tau, uint4 (&r)[PIECES]) {
        const unsigned char* tile = w + (int64_t)(tau % wtiles) * TILE;
#pragma unroll
        for (int q = 0; q < PIECES; ++q) r[q] = *reinterpret_cast<const uint4*>(tile + (iw + 4 * q) * 1024 + lane * 16);
      };
      auto write_tile = [&](int tau, const uint4 (&r)[PIECES]) {
        unsigned char* slot = smem + (tau % NSLOT) * TILE;
#pragma unroll
        for (int q = 0; q < PIECES; ++q) *reinterpret_cast<uint4*>(slot + (iw + 4 * q) * 1024 + lane * 16) = r[q];
      };
      load_tile(0, r0);
      load_tile(1, r1);
      for (int k = 0; k < ntiles; k += 2) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        write_tile(k, r0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        load_tile(k + 2, r0);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        write_tile(k + 1, r1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        load_tile(k + 3, r1);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 999) out[0] = (float)(r0[0].x + r1[0].x);
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
      if (tid == 0) cyc[blockIdx.x] = t1 - t0;
      return;
    }
    if (!NODMA) for (int t = 0; t < DIST; ++t) issue_tile(t);
    for (int k = 0; k < ntiles; ++k) {
      // tile k landed: at most (DIST - 1) newer tiles in flight
      constexpr int ALLOW = (DIST - 1) * PIECES;
      if constexpr (ALLOW >= 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
      else if constexpr (ALLOW >= 40) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
      else if constexpr (ALLOW >= 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      else if constexpr (ALLOW >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else if constexpr (ALLOW >= 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
      else if constexpr (ALLOW >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if constexpr (ALLOW >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if constexpr (ALLOW >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if constexpr (ALLOW >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (!NODMA) issue_tile(k + DIST);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    bf16x8 xh[4], xl[4];
    for (int st = 0; st < 4; ++st)
      for (int e = 0; e < 8; ++e) { xh[st][e] = (__bf16)(float)(lane + e + st); xl[st][e] = (__bf16)(float)(lane - e); }
    // conflict-free 16-B reads: 16 rows 256 B apart, chunk xor row
    int aP[4];
    for (int u = 0; u < 4; ++u) {
      if (RDPAT == 0) aP[u] = i * 256 + (((2 * u + (g & 1)) ^ i) & 15) * 16 + (g >> 1) * 4096 + (u >> 1) * 8192;
      else { const int lc = 4 * u + g; aP[u] = (i * 512 + ((lc & ~15) | ((lc & 15) ^ i)) * 16) % (TILE / 2); }   // k_tblock pattern
    }
    auto lds_read = [&](bf16x8& dst, const unsigned char* p) {
      const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)p;
      asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory");
    };
    f32x4 acc[4];
    for (int n = 0; n < 4; ++n) acc[n] = f32x4{0, 0, 0, 0};
    bf16x8 fh[3][2], fl[3][2];
    auto rd = [&](const unsigned char* slot, int u, int set, int j) {
      const unsigned char* p = slot + aP[u & 3] + (RDPAT == 0 ? (j & 1) * 128 + (j >> 1) * 16384 % TILE
                                                               : ((j & 1) * 256 + (j >> 1) * 8192) % (TILE / 2));
      if (j & 1) lds_read(fl[set][j >> 1], p); else lds_read(fh[set][j >> 1], p);
    };
    __builtin_amdgcn_s_barrier();                        // B(0)
    if (UPT == 1 && ntiles > 1) { __builtin_amdgcn_s_barrier(); }   // B(1) (a 1-unit tile prefetches two tiles ahead)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) rd(smem + ((u / UPT) % NSLOT) * TILE, u % UPT, u, j);
    for (int tau3 = 0; tau3 < ntiles; tau3 += 3) {
#pragma unroll
      for (int tt = 0; tt < 3; ++tt) {
        const int tau = tau3 + tt;
#pragma unroll
        for (int u = 0; u < UPT; ++u) {
          const int U = tt * UPT + u;                    // static unit index within the 3-tile group
          // unit u+2 lives in tile P = tau + (u+2)/UPT: publish it first if it is that tile's first unit
          const int ahead = (u + 2) / UPT;               // 0, 1 or 2 tiles ahead
          const int P = tau + ahead;
          const bool pre = P < ntiles;                   // (ntiles is a multiple of 3: no ragged group)
          if ((u + 2) % UPT == 0 && pre) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();                // B(P)
            __builtin_amdgcn_sched_barrier(0);
          }
          const int s0 = U % 3, s2 = (U + 2) % 3;
          // unit u+1's reads are behind unit u's iff unit u+1 exists
          const bool later = tau + (u + 1) / UPT < ntiles;
          if (later) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
          else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          const unsigned char* slot2 = smem + ((tau + ahead) % NSLOT) * TILE;
          const int u2 = (u + 2) % UPT;
          const int st = u & 3, f0 = 2 * (u & 1);
          acc[f0] = MFMA(fl[s0][0], xh[st], acc[f0], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (pre) rd(slot2, u2, s2, 0);
          __builtin_amdgcn_sched_barrier(0);
          acc[f0 + 1] = MFMA(fl[s0][1], xh[st], acc[f0 + 1], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (pre) rd(slot2, u2, s2, 1);
          __builtin_amdgcn_sched_barrier(0);
          acc[f0] = MFMA(fh[s0][0], xl[st], acc[f0], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (pre) rd(slot2, u2, s2, 2);
          __builtin_amdgcn_sched_barrier(0);
          acc[f0 + 1] = MFMA(fh[s0][1], xl[st], acc[f0 + 1], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (pre) rd(slot2, u2, s2, 3);
          __builtin_amdgcn_sched_barrier(0);
          acc[f0] = MFMA(fh[s0][0], xh[st], acc[f0], 0, 0, 0);
          acc[f0 + 1] = MFMA(fh[s0][1], xh[st], acc[f0 + 1], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
    out[blockIdx.x * 256 + (tid & 255)] = s[0] + s[1] + s[2] + s[3];
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

// Pure weight-stream rate: LW loader waves fill a ring of NSLOT tiles of TILE bytes as fast as slots free up; the
// 4 "compute" waves only take part in the per-tile barrier (the tile is released immediately).
template <int TILE, int NSLOT, int LW>
__global__ __launch_bounds__((4 + LW) * 64) void kdma(const unsigned char* w, float* out, unsigned long long* cyc,
                                                      int ntiles, int wtiles, int unit_gap) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr int PIECES = TILE / 1024 / LW;              // per loader wave
  constexpr int DIST = NSLOT - 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned long long t0 = 0, t1 = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  if (wave >= 4) {
    const int iw = wave - 4;
    auto issue_tile = [&](int tau) {
      const unsigned char* tile = w + (int64_t)(tau % wtiles) * TILE;
      unsigned char* slot = smem + (tau % NSLOT) * TILE;
#pragma unroll
      for (int q = 0; q < PIECES; ++q) {
        const int inst = iw + LW * q;
        __builtin_amdgcn_global_load_lds(tile + inst * 1024 + lane * 16,
                                         (__attribute__((address_space(3))) void*)(slot + inst * 1024), 16, 0, 0);
      }
    };
    for (int t = 0; t < DIST; ++t) issue_tile(t);
    for (int k = 0; k < ntiles; ++k) {
      constexpr int ALLOW = (DIST - 1) * PIECES;
      if constexpr (ALLOW >= 56) asm volatile("s_waitcnt vmcnt(56)" ::: "memory");
      else if constexpr (ALLOW >= 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
      else if constexpr (ALLOW >= 40) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
      else if constexpr (ALLOW >= 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      else if constexpr (ALLOW >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else if constexpr (ALLOW >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if constexpr (ALLOW >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if constexpr (ALLOW >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if constexpr (ALLOW >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if constexpr (ALLOW >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                       // tile k landed = tile k released (nobody reads it)
      issue_tile(k + DIST);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    for (int k = 0; k < ntiles; ++k) __builtin_amdgcn_s_barrier();
    if (tid < 256) out[blockIdx.x * 256 + tid] = (float)smem[tid * 16];
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

// Occupancy premise: the same per-tile work (48 MFMAs + a GELU-like VALU block + 8 DMA pieces per SIMD) done by ONE
// wave per SIMD (NW = 4) or split over TWO waves per SIMD (NW = 8, each half the MFMAs / VALU / pieces).  Ring of 4
// slots, burst DMA issue by every wave right after the tile barrier, plain fragment reads two units ahead.
template <int NW>
__global__ __launch_bounds__(NW * 64) void kpair(const unsigned char* w, float* out, unsigned long long* cyc,
                                                 int ntiles, int wtiles, int valu_iters) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr int PIECES = 32 / NW;                        // per wave per 32 KB tile
  constexpr int UNITS = 32 / NW;                         // 6-MFMA units per wave per tile (8 or 4)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  bf16x8 xh[4], xl[4];
  for (int st = 0; st < 4; ++st)
    for (int e = 0; e < 8; ++e) { xh[st][e] = (__bf16)(float)(lane + e + st); xl[st][e] = (__bf16)(float)(lane - e); }
  for (int t = tid; t < NS * SLOT / 4; t += blockDim.x) ((float*)smem)[t] = 0.f;
  __syncthreads();
  auto issue_tile = [&](int tau) {
    const unsigned char* tile = w + (int64_t)(tau % wtiles) * SLOT;
    unsigned char* slot = smem + (tau % NS) * SLOT;
#pragma unroll
    for (int q = 0; q < PIECES; ++q) {
      const int inst = wave + NW * q;
      __builtin_amdgcn_global_load_lds(tile + inst * 1024 + lane * 16,
                                       (__attribute__((address_space(3))) void*)(slot + inst * 1024), 16, 0, 0);
    }
  };
  int aP[4];
  for (int st = 0; st < 4; ++st) {
    const int lc = 4 * st + g;
    aP[st] = i * (4 * C) + ((lc & ~15) | ((lc & 15) ^ i)) * 16 + (NW == 8 ? (wave >> 2) * 2 * 16 * 4 * C : 0);
  }
  auto lds_read = [&](bf16x8& dst, const unsigned char* p) {
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)p;
    asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory");
  };
  f32x4 acc[4];
  for (int n = 0; n < 4; ++n) acc[n] = f32x4{0, 0, 0, 0};
  float vv[16];
  for (int e = 0; e < 16; ++e) vv[e] = 0.001f * (lane + e);
  for (int t = 0; t < NS - 1; ++t) issue_tile(t);
  unsigned long long t0 = 0, t1 = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int tau = 0; tau < ntiles; ++tau) {
    if constexpr (PIECES == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue_tile(tau + NS - 1);
    const unsigned char* slot = smem + (tau % NS) * SLOT;
    bf16x8 fh[3][2], fl[3][2];
    auto load = [&](int u, int set) {
      const int st = (NW == 8) ? u : (u >> 1), pr = (NW == 8) ? 0 : (u & 1);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        lds_read(fh[set][q], slot + aP[st] + ((2 * pr + q) * 16 * 4 * C));
        lds_read(fl[set][q], slot + aP[st] + ((2 * pr + q) * 16 * 4 * C + 2 * C));
      }
    };
    load(0, 0);
    load(1, 1);
#pragma unroll
    for (int u = 0; u < UNITS; ++u) {
      const int s0 = u % 3;
      if (u + 2 < UNITS) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
      else if (u + 1 < UNITS) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      const int st = (NW == 8) ? u : (u >> 1), f0 = (NW == 8) ? 0 : 2 * (u & 1);
      acc[f0] = MFMA(fl[s0][0], xh[st], acc[f0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (u + 2 < UNITS) { const int st2 = (NW == 8) ? u + 2 : ((u + 2) >> 1), pr2 = (NW == 8) ? 0 : ((u + 2) & 1);
        lds_read(fh[(u + 2) % 3][0], slot + aP[st2] + ((2 * pr2) * 16 * 4 * C)); }
      __builtin_amdgcn_sched_barrier(0);
      acc[f0 + 1] = MFMA(fl[s0][1], xh[st], acc[f0 + 1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (u + 2 < UNITS) { const int st2 = (NW == 8) ? u + 2 : ((u + 2) >> 1), pr2 = (NW == 8) ? 0 : ((u + 2) & 1);
        lds_read(fl[(u + 2) % 3][0], slot + aP[st2] + ((2 * pr2) * 16 * 4 * C + 2 * C)); }
      __builtin_amdgcn_sched_barrier(0);
      acc[f0] = MFMA(fh[s0][0], xl[st], acc[f0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (u + 2 < UNITS) { const int st2 = (NW == 8) ? u + 2 : ((u + 2) >> 1), pr2 = (NW == 8) ? 0 : ((u + 2) & 1);
        lds_read(fh[(u + 2) % 3][1], slot + aP[st2] + ((2 * pr2 + 1) * 16 * 4 * C)); }
      __builtin_amdgcn_sched_barrier(0);
      acc[f0 + 1] = MFMA(fh[s0][1], xl[st], acc[f0 + 1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (u + 2 < UNITS) { const int st2 = (NW == 8) ? u + 2 : ((u + 2) >> 1), pr2 = (NW == 8) ? 0 : ((u + 2) & 1);
        lds_read(fl[(u + 2) % 3][1], slot + aP[st2] + ((2 * pr2 + 1) * 16 * 4 * C + 2 * C)); }
      __builtin_amdgcn_sched_barrier(0);
      acc[f0] = MFMA(fh[s0][0], xh[st], acc[f0], 0, 0, 0);
      acc[f0 + 1] = MFMA(fh[s0][1], xh[st], acc[f0 + 1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // GELU-like VALU block on (64 / NW) values per lane
    if (valu_iters) {
#pragma unroll
      for (int e = 0; e < 64 / NW; ++e) {
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
  for (int e = 0; e < 16; ++e) r += vv[e];
  out[blockIdx.x * NW * 64 + tid] = r;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

static int g_arg = 0;
int main(int argc, char** argv) {
  unsigned char* w; float* out; unsigned long long* cyc;
  const int wtiles = 64;
  hipMalloc(&w, (size_t)(wtiles + 8) * SLOT); hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8 * 1024);
  hipMemset(w, 0, (size_t)(wtiles + 8) * SLOT);
  auto report = [&](const char* name, auto kern, int threads, int blocks, size_t smem, int ntiles, int tile_bytes, int mfma_per_tile) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), smem, 0, w, out, cyc, ntiles, wtiles, g_arg);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), smem, 0, w, out, cyc, ntiles, wtiles, g_arg);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[1024]; hipMemcpy(c, cyc, 8 * blocks, hipMemcpyDeviceToHost);
    double s = 0, mx = 0; for (int b = 0; b < blocks; ++b) { s += c[b]; if (c[b] > mx) mx = c[b]; }
    printf("%-60s blocks %4d : %7.1f cyc per 32 KB (max %7.1f) = %5.2f cyc/MFMA ; %6.2f TB/s L2->LDS\n", name, blocks,
           s / blocks / ntiles * (32768.0 / tile_bytes), mx / ntiles * (32768.0 / tile_bytes), s / blocks / ntiles / mfma_per_tile,
           (double)blocks * ntiles * tile_bytes / (ms * 1e-3) / 1e12);
  };
  const bool all = argc > 1;
  for (int blocks : {1, 256}) {
    if (all) {
      report("V4 MFMA only (register operands)", kproj<4>, 256, blocks, NS * SLOT, 510, SLOT, 48);
      report("V3 MFMA + ds_read_b128, static LDS", kproj<3>, 256, blocks, NS * SLOT, 510, SLOT, 48);
      report("V0 ring, burst DMA issue by consumers", kproj<0>, 256, blocks, NS * SLOT, 510, SLOT, 48);
      report("V2 ring, 4 loader waves", kproj<2>, 512, blocks, NS * SLOT, 510, SLOT, 48);
      report("V3i static LDS, reads interleaved with MFMAs", kproj<3, 1>, 256, blocks, NS * SLOT, 510, SLOT, 48);
      report("P0 static LDS, continuous cross-tile read pipeline", kpipe<0>, 256, blocks, NS * SLOT, 510, SLOT, 48);
      report("P1 ring + 4 loader waves, continuous pipeline", kpipe<1>, 512, blocks, NS * SLOT, 510, SLOT, 48);
    }
    g_arg = 1;
    report("O 4 waves: 48 MFMA + 16 GELU + 8 pieces / wave / tile", kpair<4>, 256, blocks, NS * SLOT, 512, 32768, 48);
    report("O 8 waves: 24 MFMA +  8 GELU + 4 pieces / wave / tile", kpair<8>, 512, blocks, NS * SLOT, 512, 32768, 48);
    g_arg = 0;
    report("O 4 waves: 48 MFMA + 8 pieces / wave / tile, no VALU", kpair<4>, 256, blocks, NS * SLOT, 512, 32768, 48);
    report("O 8 waves: 24 MFMA + 4 pieces / wave / tile, no VALU", kpair<8>, 512, blocks, NS * SLOT, 512, 32768, 48);
    report("D pure DMA 32KBx4, 4 loader waves", kdma<32768, 4, 4>, 512, blocks, 4 * 32768, 512, 32768, 48);
    report("D pure DMA 32KBx4, 8 loader waves", kdma<32768, 4, 8>, 768, blocks, 4 * 32768, 512, 32768, 48);
    report("D pure DMA 16KBx8, 4 loader waves", kdma<16384, 8, 4>, 512, blocks, 8 * 16384, 1024, 16384, 48);
    report("D pure DMA 16KBx8, 8 loader waves", kdma<16384, 8, 8>, 768, blocks, 8 * 16384, 1024, 16384, 48);
    report("D pure DMA  8KBx16, 4 loader waves", kdma<8192, 16, 4>, 512, blocks, 16 * 8192, 2048, 8192, 48);
    // 48 MFMAs per 32 KB per wave (k_tblock_lw: 64-row workgroups, every wave uses the whole tile)
    report("R 32KBx4 dist2, 8 units/tile   (k_tblock_lw now)", kring<32768, 4, 8, 2>, 512, blocks, 4 * 32768, 510, 32768, 48);
    report("R 32KBx4 dist2, 8 units/tile, REGISTER-staged by the loaders", kring<32768, 4, 8, 2, 0, 0, 1>, 512, blocks, 4 * 32768, 510, 32768, 48);
    report("R 32KBx4 dist2, 4 units/tile, REGISTER-staged by the loaders", kring<32768, 4, 4, 2, 0, 0, 1>, 512, blocks, 4 * 32768, 510, 32768, 24);
    report("R 32KBx4 dist2, 8 units/tile, NO DMA", kring<32768, 4, 8, 2, 1>, 512, blocks, 4 * 32768, 510, 32768, 48);
    report("R 32KBx4 dist2, 8 units/tile, NO DMA, k_tblock read pattern", kring<32768, 4, 8, 2, 1, 1>, 512, blocks, 4 * 32768, 510, 32768, 48);
    report("R 32KBx4 dist2, 8 units/tile, k_tblock read pattern", kring<32768, 4, 8, 2, 0, 1>, 512, blocks, 4 * 32768, 510, 32768, 48);
    report("P1 ring + 4 loader waves, continuous pipeline", kpipe<1>, 512, blocks, NS * SLOT, 510, SLOT, 48);
    report("R 32KBx4 dist2, 4 units/tile, NO DMA", kring<32768, 4, 4, 2, 1>, 512, blocks, 4 * 32768, 510, 32768, 24);
    report("R 32KBx4 dist2, 4 units/tile, k_tblock read pattern", kring<32768, 4, 4, 2, 0, 1>, 512, blocks, 4 * 32768, 510, 32768, 24);
    report("R 16KBx8 dist6, 4 units/tile", kring<16384, 8, 4, 6>, 512, blocks, 8 * 16384, 1020, 16384, 24);
    report("R 16KBx9 dist7, 4 units/tile", kring<16384, 9, 4, 7>, 512, blocks, 9 * 16384, 1020, 16384, 24);
    // 24 MFMAs per 32 KB per wave (k_tblock32: a wave uses half of each tile's features)
    report("R 32KBx4 dist2, 4 units/tile   (k_tblock32 now)", kring<32768, 4, 4, 2>, 512, blocks, 4 * 32768, 510, 32768, 24);
    // 12 MFMAs per 32 KB per wave: a 16-row workgroup whose four waves take a quarter of every tile's features each
    report("R 32KBx4 dist2, 2 units/tile   (16-row workgroups)", kring<32768, 4, 2, 2>, 512, blocks, 4 * 32768, 510, 32768, 12);
    report("R 32KBx4 dist2, 2 units/tile, NO DMA", kring<32768, 4, 2, 2, 1>, 512, blocks, 4 * 32768, 510, 32768, 12);
    report("R 16KBx8 dist6, 2 units/tile", kring<16384, 8, 2, 6>, 512, blocks, 8 * 16384, 1020, 16384, 12);
    report("R 16KBx9 dist7, 2 units/tile", kring<16384, 9, 2, 7>, 512, blocks, 9 * 16384, 1020, 16384, 12);
    report("R  8KBx18 dist15, 1 unit/tile", kring<8192, 18, 1, 15>, 512, blocks, 18 * 8192, 2040, 8192, 6);
  }
  return 0;
}
