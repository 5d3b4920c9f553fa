// Floor of a dependent kernel launch on gfx950: back-to-back launches of (nearly) empty kernels in one stream, as plain
// launches and as a captured graph, for the launch shapes of the fused kernels (512 threads, 128 KB LDS, 256 VGPRs).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k_small(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }

__global__ __launch_bounds__(512) void k_lds(float* p) {
  extern __shared__ float sm[];
  if (threadIdx.x == 0) sm[0] = 1.f;
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += sm[0];
}

__global__ __launch_bounds__(512) void k_regs(float* p) {   // forces a 256-VGPR allocation
  extern __shared__ float sm[];
  float v[200];
#pragma unroll
  for (int i = 0; i < 200; ++i) v[i] = p[(threadIdx.x + i) & 1023];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 200; ++i) s += v[i] * v[(i * 7) % 200];
  if (threadIdx.x == 0) sm[0] = s;
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += sm[0] * 1e-30f;
}

// one dependent global load -> store per thread (what every prologue / epilogue does at least once)
__global__ __launch_bounds__(512) void k_ldst(float* p) {
  extern __shared__ float sm[];
  const int t = blockIdx.x * 512 + threadIdx.x;
  const float4 v = reinterpret_cast<const float4*>(p)[t];
  reinterpret_cast<float4*>(p)[t + 262144] = make_float4(v.x + 1.f, v.y, v.z, v.w);
}

// the same plus a chain of 3 dependent loads (pointer-chase-like: x rows -> params -> weights)
__global__ __launch_bounds__(512) void k_chain(float* p) {
  extern __shared__ float sm[];
  const int t = blockIdx.x * 512 + threadIdx.x;
  float4 v = reinterpret_cast<const float4*>(p)[t];
  int j = ((int)v.x & 1023);
  v = reinterpret_cast<const float4*>(p)[t + 131072 + j];
  j = ((int)v.y & 1023);
  v = reinterpret_cast<const float4*>(p)[t + 65536 + j];
  reinterpret_cast<float4*>(p)[t + 262144] = v;
}

int main() {
  float* p; hipMalloc(&p, 1 << 24); hipMemset(p, 0, 1 << 24);
  hipStream_t st; hipStreamCreate(&st);
  const int N = 200;
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 10; ++i) launch();
    hipStreamSynchronize(st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st);
    for (int i = 0; i < N; ++i) launch();
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < N; ++i) launch();
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    hipGraphLaunch(ge, st);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    float msg; hipEventElapsedTime(&msg, e0, e1);
    printf("%-52s plain %6.2f us/launch   graph %6.2f us/launch\n", name, ms * 1e3 / N, msg * 1e3 / N);
  };
  hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)k_regs, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  run("256 thr x 1 WG, no LDS", [&] { hipLaunchKernelGGL(k_small, dim3(1), dim3(256), 0, st, p); });
  run("256 thr x 4096 WG, no LDS", [&] { hipLaunchKernelGGL(k_small, dim3(4096), dim3(256), 0, st, p); });
  run("512 thr x 256 WG, 4 KB LDS", [&] { hipLaunchKernelGGL(k_lds, dim3(256), dim3(512), 4096, st, p); });
  run("512 thr x 256 WG, 128 KB LDS", [&] { hipLaunchKernelGGL(k_lds, dim3(256), dim3(512), 131072, st, p); });
  run("512 thr x 128 WG, 128 KB LDS", [&] { hipLaunchKernelGGL(k_lds, dim3(128), dim3(512), 131072, st, p); });
  hipFuncSetAttribute((const void*)k_ldst, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)k_chain, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  run("512 thr x 256 WG, 128 KB LDS, load -> store", [&] { hipLaunchKernelGGL(k_ldst, dim3(256), dim3(512), 131072, st, p); });
  run("512 thr x 2 WG, 128 KB LDS, load -> store", [&] { hipLaunchKernelGGL(k_ldst, dim3(2), dim3(512), 131072, st, p); });
  run("512 thr x 256 WG, 128 KB LDS, 3 dependent loads -> store", [&] { hipLaunchKernelGGL(k_chain, dim3(256), dim3(512), 131072, st, p); });
  run("512 thr x 256 WG, 128 KB LDS, 256 VGPRs", [&] { hipLaunchKernelGGL(k_regs, dim3(256), dim3(512), 131072, st, p); });
  return 0;
}
