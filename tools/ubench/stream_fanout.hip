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
  return 0;
}
