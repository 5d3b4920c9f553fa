// How long until a kernel's first global load can be ISSUED?  The address comes from the kernel arguments, which a wave
// normally fetches with s_load from the kernarg segment: a memory round trip before the first data request.  With
// -mllvm -amdgpu-kernarg-preload-count=N the command processor hands the first N argument dwords over in SGPRs.
// Each workgroup stamps the shader clock at entry, when the pointer argument is usable (after a dependent scalar op) and
// when the first loaded row is back.  Arguments: a big struct by value (as the fused kernels take) or leading scalars.
// Build & run on the GPU box (both variants):
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/kernarg_probe.hip -o /tmp/ka0 && /tmp/ka0
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=8 tools/ubench/kernarg_probe.hip -o /tmp/ka8 && /tmp/ka8
#include <hip/hip_runtime.h>
#include <cstdio>

struct Args { const float* x; float* out; unsigned long long* res; int M, ld; int pad[40]; };

__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

__global__ __launch_bounds__(256) void k_struct(Args a) {
  const unsigned long long t0 = now();
  const float* p = a.x + (size_t)(blockIdx.x * 256 + threadIdx.x) * 4;
  const unsigned long long t1 = now();          // (the s_load of a.x is waited for by now()'s lgkmcnt(0) only if issued before)
  const float4 v = *reinterpret_cast<const float4*>(p);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t2 = now();
  a.out[blockIdx.x * 256 + threadIdx.x] = v.x + v.y + v.z + v.w;
  if (threadIdx.x == 0) { a.res[3 * blockIdx.x] = t1 - t0; a.res[3 * blockIdx.x + 1] = t2 - t1; a.res[3 * blockIdx.x + 2] = t2 - t0; }
}

__global__ __launch_bounds__(256) void k_lead(const float* x, float* out, unsigned long long* res, Args a) {
  const unsigned long long t0 = now();
  const float* p = x + (size_t)(blockIdx.x * 256 + threadIdx.x) * 4;
  const unsigned long long t1 = now();
  const float4 v = *reinterpret_cast<const float4*>(p);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t2 = now();
  out[blockIdx.x * 256 + threadIdx.x] = v.x + v.y + v.z + v.w + (float)a.pad[3];
  if (threadIdx.x == 0) { res[3 * blockIdx.x] = t1 - t0; res[3 * blockIdx.x + 1] = t2 - t1; res[3 * blockIdx.x + 2] = t2 - t0; }
}

int main() {
  Args a{};
  float* x; hipMalloc(&x, 256 * 256 * 16); hipMemset(x, 0, 256 * 256 * 16);
  hipMalloc(&a.out, 256 * 256 * 4); hipMalloc(&a.res, 256 * 24);
  a.x = x; a.M = 1; a.ld = 1;
  for (int blocks : {1, 256}) {
    for (int which = 0; which < 2; ++which) {
      double s[3] = {0, 0, 0};
      const int reps = 20;
      for (int r = 0; r < reps; ++r) {
        hipMemsetAsync(x, 0, 256 * 256 * 16, 0);         // another kernel in between, as in the evaluation
        if (which == 0) hipLaunchKernelGGL(k_struct, dim3(blocks), dim3(256), 0, 0, a);
        else hipLaunchKernelGGL(k_lead, dim3(blocks), dim3(256), 0, 0, a.x, a.out, a.res, a);
        hipDeviceSynchronize();
        unsigned long long h[768]; hipMemcpy(h, a.res, blocks * 24, hipMemcpyDeviceToHost);
        for (int b = 0; b < blocks; ++b) for (int k = 0; k < 3; ++k) s[k] += (double)h[3 * b + k] / blocks / reps;
      }
      printf("%-34s %3d workgroups: entry -> address ready %6.0f cycles, -> row back %6.0f, total %6.0f\n",
             which ? "leading scalar arguments" : "struct by value", blocks, s[0], s[1], s[2]);
    }
  }
  return 0;
}
