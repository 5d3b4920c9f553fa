// What would a barrier across all resident workgroups cost inside a persistent launch, against the 1.65 us dependent-launch gap
// (+ dispatch, + kernel-argument fetch) of a HIP graph?  256 workgroups x 512 threads (one per CU, as the ring kernels), N rounds of
// { every workgroup: release its stores, one atomic add on a counter in device memory; thread 0 polls the generation word with
// s_sleep between polls; workgroup barrier }.  Reports the time per round; with `work` > 0 every workgroup first writes and then
// reads 32 KB of rows another workgroup of the same XCD wrote in the previous round (the producer -> consumer pattern of the
// sub-block chain), so the number includes the agent-scope release / acquire the data needs.
// Build & run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/ubench/grid_barrier.hip -o /tmp/gb && /tmp/gb
#include <hip/hip_runtime.h>
#include <cstdio>

template <int FENCE>
__global__ __launch_bounds__(512) void kbar(unsigned* ctr, float* buf, float* out, unsigned long long* res, int rounds, int work) {
  const unsigned nwg = gridDim.x;
  unsigned long long t0, t1;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r) {
    if (work) {
      // write my 32 KB, to be read by the workgroup 8 further (same XCD) in the next round
      float4* p = reinterpret_cast<float4*>(buf) + ((size_t)(r & 1) * nwg + blockIdx.x) * 2048 + threadIdx.x;
      for (int k = 0; k < 4; ++k) p[512 * k] = make_float4(r + acc, 1.f, 2.f, 3.f);
    }
    if (FENCE) __threadfence();                        // release (agent scope)
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned target = (unsigned)(r + 1) * nwg;
      __hip_atomic_fetch_add(ctr, 1u, FENCE ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(ctr, FENCE ? __ATOMIC_ACQUIRE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    if (FENCE) __threadfence();                        // acquire
    if (work) {
      const float4* q = reinterpret_cast<const float4*>(buf) + ((size_t)(r & 1) * nwg + (blockIdx.x + 8) % nwg) * 2048 + threadIdx.x;
      for (int k = 0; k < 4; ++k) { const float4 v = q[512 * k]; acc += v.x * 1e-9f; }
    }
  }
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  out[blockIdx.x * 512 + threadIdx.x] = acc;
  if (threadIdx.x == 0) res[blockIdx.x] = t1 - t0;
}

int main() {
  unsigned* ctr; float *buf, *out; unsigned long long* res;
  const int nwg = 256, rounds = 200;
  hipMalloc(&ctr, 64); hipMalloc(&buf, (size_t)2 * nwg * 32768); hipMalloc(&out, nwg * 512 * 4); hipMalloc(&res, nwg * 8);
  for (int mode = 0; mode < 3; ++mode) {
    const int work = mode == 2;
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(ctr, 0, 64);
      if (mode == 0) hipLaunchKernelGGL(kbar<0>, dim3(nwg), dim3(512), 0, 0, ctr, buf, out, res, rounds, 0);
      else hipLaunchKernelGGL(kbar<1>, dim3(nwg), dim3(512), 0, 0, ctr, buf, out, res, rounds, work);
      hipDeviceSynchronize();
      unsigned long long h[256]; hipMemcpy(h, res, nwg * 8, hipMemcpyDeviceToHost);
      double s = 0; for (int i = 0; i < nwg; ++i) s += h[i];
      printf("%s: %.2f us per round (256 workgroups, mean over workgroups, %d rounds)\n",
             mode == 0 ? "counter only (relaxed atomics, no fences: synchronises, publishes nothing)"
                       : work ? "release / acquire fences + 32 KB written / read across the barrier per workgroup" : "agent-scope release / acquire fences, no data",
             s / nwg / rounds / 100.0, rounds);
    }
  }
  return 0;
}
