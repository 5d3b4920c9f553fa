// Micro-benchmark: aggregate L2 -> LDS stream rate when MANY workgroups pull the SAME weight stream (what the fused
// transformer kernels do: every row block reads the same 1-2 MB of tiles in the same order at the same time).
//   mode 0  every workgroup streams tiles 0, 1, 2, ... of the same buffer (the kernels' pattern)
//   mode 1  workgroup b starts at tile (b * stride) % ntiles (same bytes in total, different tile per workgroup at any instant)
//   mode 2  every workgroup has its own private copy of the stream (no sharing at all; buffer = nwg x stream)
// 4 loader waves per workgroup (one per SIMD), 8 pieces of 1 KB per wave and 32 KB tile, 4-slot ring, at most 2 tiles in
// flight per wave (vmcnt), no consumers.  Prints bytes / clock / CU and the aggregate TB/s.
// Build & run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/ubench/stream_fanout.hip -o /tmp/sf && /tmp/sf
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

constexpr int SLOT = 32768, NS = 4;

__global__ __launch_bounds__(256) void kstream(const unsigned char* w, int ntiles_stream, int reps, int mode, int stride,
                                               unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int lane = threadIdx.x & 63, iw = threadIdx.x >> 6;
  const unsigned char* base = w + (mode == 2 ? (int64_t)blockIdx.x * ntiles_stream * SLOT : 0);
  const int start = mode == 1 ? (int)(((int64_t)blockIdx.x * stride) % ntiles_stream) : 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  const int total = ntiles_stream * reps;
  for (int tau = 0; tau < total; ++tau) {
    const unsigned char* tile = base + (int64_t)((tau + start) % ntiles_stream) * SLOT;
    unsigned char* slot = smem + (tau % NS) * SLOT;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int inst = iw + 4 * q;
      __builtin_amdgcn_global_load_lds(tile + inst * 1024 + lane * 16,
                                       (__attribute__((address_space(3))) void*)(slot + inst * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // one tile of this wave may stay in flight
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// Once-through (COLD L2) shared stream with an optional software prefetch: waves 4..7 stand in for the compute waves of
// the ring kernels and touch one dword per 128-byte line of tile tau + pf_dist (pf_who = 0: every workgroup, 1: one
// workgroup per XCD and tile, round robin), never waiting for the data; one workgroup barrier per tile as in the kernels.
__global__ __launch_bounds__(512) void kstream_pf(const unsigned char* w, int ntiles, int pf_dist, int pf_who,
                                                  unsigned long long* cyc, float* sink) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (wv < 4) {
    const int iw = wv;
    for (int tau = 0; tau < ntiles; ++tau) {
      const unsigned char* tile = w + (int64_t)tau * SLOT;
      unsigned char* slot = smem + (tau % NS) * SLOT;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int inst = iw + 4 * q;
        __builtin_amdgcn_global_load_lds(tile + inst * 1024 + lane * 16,
                                         (__attribute__((address_space(3))) void*)(slot + inst * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    const int p = wv - 4;
    const int xslot = (blockIdx.x >> 3) & 31;
    float d = 0.f;                       // one landing register for every prefetch, live until the end of the kernel
    for (int tau = 0; tau < ntiles; ++tau) {
      const int tp = tau + pf_dist;
      if (pf_dist > 0 && tp < ntiles && (pf_who == 0 || (tp & 31) == xslot)) {
        const unsigned char* line = w + (int64_t)tp * SLOT + (p * 64 + lane) * 128;
        asm volatile("global_load_dword %0, %1, off" : "+v"(d) : "v"(line) : "memory");   // never waited for inside the loop
      }
      __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(d));
    if (d == 123.f) sink[0] = d;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
  const int nwg = argc > 1 ? atoi(argv[1]) : 256;
  const int ntiles = 32;              // 1 MB stream (one head half of a C = 256 sub-block)
  const int reps = 16;
  unsigned char* w;
  unsigned long long* cyc;
  hipMalloc(&w, (size_t)nwg * ntiles * SLOT);
  hipMemset(w, 1, (size_t)nwg * ntiles * SLOT);
  hipMalloc(&cyc, nwg * sizeof(unsigned long long));
  hipFuncSetAttribute((const void*)kstream, hipFuncAttributeMaxDynamicSharedMemorySize, NS * SLOT);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int strides[] = {0, 1, 4, 8, 0};
  const int modes[] = {0, 1, 1, 1, 2};
  for (int v = 0; v < 5; ++v) {
    for (int it = 0; it < 3; ++it) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(kstream, dim3(nwg), dim3(256), NS * SLOT, 0, w, ntiles, reps, modes[v], strides[v], cyc);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[1024];
    hipMemcpy(h, cyc, nwg * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double avg = 0;
    for (int i = 0; i < nwg; ++i) avg += (double)h[i];
    avg /= nwg;
    const double bytes = (double)ntiles * reps * SLOT;
    printf("nwg %d mode %d stride %d: %.1f us, %.2f TB/s aggregate, %.1f B / counter tick / CU (s_memrealtime-free cycle counter)\n",
           nwg, modes[v], strides[v], ms * 1e3, bytes * nwg / (ms * 1e-3) / 1e12, bytes / avg);
  }
  // ---- cold streams: every workgroup reads the same stream ONCE, front to back ----
  hipFree(w);
  const size_t big = (size_t)768 << 20;
  hipMalloc(&w, big);
  hipMemset(w, 1, big);
  float* sink; hipMalloc(&sink, 64);
  hipFuncSetAttribute((const void*)kstream_pf, hipFuncAttributeMaxDynamicSharedMemorySize, NS * SLOT);
  auto run_pf = [&](const char* name, size_t off_mb, int nt, int dist, int who, int launches) {
    for (int it = 0; it < launches; ++it) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(kstream_pf, dim3(nwg), dim3(512), NS * SLOT, 0, w + (off_mb << 20), nt, dist, who, cyc, sink);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[1024];
      hipMemcpy(h, cyc, nwg * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      double avg = 0; for (int i = 0; i < nwg; ++i) avg += (double)h[i];
      avg /= nwg;
      printf("%-58s launch %d: %8.1f us, %6.2f TB/s, %6.1f ticks / tile\n", name, it, ms * 1e3,
             (double)nt * SLOT * nwg / (ms * 1e-3) / 1e12, avg / nt);
    }
  };
  // the cycle counter here is the 100 MHz one: 1 tick = 24 shader clocks at 2.4 GHz
  run_pf("1 MB stream, same buffer every launch (L2 kept across launches?)", 0, 32, 0, 0, 4);
  run_pf("2 MB stream, same buffer every launch", 0, 64, 0, 0, 3);
  run_pf("64 MB once through (cold L2, MALL/HBM), no prefetch", 64, 2048, 0, 0, 2);
  run_pf("64 MB once through, fresh region, no prefetch", 160, 2048, 0, 0, 1);
  for (int dist : {2, 4, 8, 16}) {
    char nm[96];
    snprintf(nm, sizeof nm, "64 MB once through, prefetch %d tiles ahead, every WG", dist);
    run_pf(nm, 256 + 0, 2048, dist, 0, 1);
    snprintf(nm, sizeof nm, "64 MB once through, prefetch %d tiles ahead, 1 WG / XCD / tile", dist);
    run_pf(nm, 352, 2048, dist, 1, 1);
  }
  run_pf("64 MB once through again (MALL-warm?), no prefetch", 448, 2048, 0, 0, 2);
  return 0;
}
