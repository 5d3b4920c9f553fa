// Does a row written by one launch come out of the SAME XCD's L2 in the next launch?  (Same grid shape -> same workgroup ->
// XCD mapping, as between two fused sub-block launches.)  Kernel W writes 32 KB per workgroup (plain or non-temporal stores),
// kernel R (next launch, same grid) reads the same 32 KB with one dwordx4 per lane and reports the round trip in shader
// cycles; compared with re-reading a buffer that was only READ by the previous launch, and with a buffer nobody touched.
// Build & run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/ubench/producer_consumer_l2.hip -o /tmp/pc && /tmp/pc
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NT>
__global__ __launch_bounds__(256) void kwrite(float* buf, float v) {
  f4* p = reinterpret_cast<f4*>(buf) + (size_t)blockIdx.x * 2048 + threadIdx.x;
  for (int k = 0; k < 8; ++k) {
    const f4 val = f4{v, v + k, v, v};
    if (NT) __builtin_nontemporal_store(val, p + 256 * k); else p[256 * k] = val;
  }
}

__global__ __launch_bounds__(256) void kread(const float* buf, float* out, unsigned long long* res) {
  const f4* p = reinterpret_cast<const f4*>(buf) + (size_t)blockIdx.x * 2048 + threadIdx.x;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  f4 v[8];
  for (int k = 0; k < 8; ++k) v[k] = p[256 * k];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
  for (int k = 0; k < 8; ++k) s += v[k][0] + v[k][1] + v[k][2] + v[k][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) res[blockIdx.x] = t1 - t0;
}

int main() {
  const int nwg = 256;
  float *a, *b, *c, *out; unsigned long long* res;
  const size_t bytes = (size_t)nwg * 32768;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, 64 * bytes); hipMalloc(&out, nwg * 1024); hipMalloc(&res, nwg * 8);
  hipMemset(a, 0, bytes); hipMemset(b, 0, bytes); hipMemset(c, 0, 64 * bytes);
  auto report = [&](const char* name) {
    hipDeviceSynchronize();
    unsigned long long h[256]; hipMemcpy(h, res, nwg * 8, hipMemcpyDeviceToHost);
    double s = 0, mn = 1e18, mx = 0; for (int i = 0; i < nwg; ++i) { s += h[i]; if (h[i] < mn) mn = h[i]; if (h[i] > mx) mx = h[i]; }
    printf("%-72s round trip %6.0f / %6.0f / %6.0f cycles (min / mean / max over %d workgroups)\n", name, mn, s / nwg, mx, nwg);
  };
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(kwrite<0>, dim3(nwg), dim3(256), 0, 0, a, 1.0f);
    hipLaunchKernelGGL(kread, dim3(nwg), dim3(256), 0, 0, a, out, res);
    report("written by the previous launch, plain stores");
    hipLaunchKernelGGL(kwrite<1>, dim3(nwg), dim3(256), 0, 0, a, 2.0f);
    hipLaunchKernelGGL(kread, dim3(nwg), dim3(256), 0, 0, a, out, res);
    report("written by the previous launch, non-temporal stores");
    hipLaunchKernelGGL(kread, dim3(nwg), dim3(256), 0, 0, a, out, res);
    report("read (not written) by the previous launch");
    hipLaunchKernelGGL(kread, dim3(nwg), dim3(256), 0, 0, c + (size_t)(8 + 16 * rep) * nwg * 8192, out, res);
    report("untouched since its memset, far away in memory");
    // the same with 64 MB of other traffic between producer and consumer (as ~3 other launches' rows and weights)
    hipLaunchKernelGGL(kwrite<0>, dim3(nwg), dim3(256), 0, 0, a, 3.0f);
    hipLaunchKernelGGL(kread, dim3(nwg), dim3(256), 0, 0, b, out, res);
    hipLaunchKernelGGL(kread, dim3(nwg), dim3(256), 0, 0, a, out, res);
    report("written two launches ago, plain stores");
  }
  return 0;
}
