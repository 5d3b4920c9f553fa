// How fast does a wave run through code it executes for the FIRST time?  The fused kernels are 20-70 KB of mostly
// straight-line code, a large part of it executed once per launch (row prologue, epilogue) or once per head loop.
// A block of 4096 dependent 8-byte VALU instructions (32 KB, 4 cycles each when the instruction cache hits) is run
// three times inside one kernel, with the shader clock read around each pass: pass 0 is cold, passes 1 and 2 warm.
// Launched twice in a row (does the instruction cache survive a kernel boundary?), alternating with a second kernel of
// the same size (as consecutive launches of different fused kernels do), and with 1 / 8 waves per workgroup.
// Build & run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/ubench/icache_probe.hip -o /tmp/ic && /tmp/ic
#include <hip/hip_runtime.h>
#include <cstdio>

#define BODY(N) asm volatile(".rept " #N "\n\tv_fma_f32 %0, %0, %1, %2\n\t.endr" : "+v"(x) : "v"(a), "v"(b));

template <int ID>
__global__ __launch_bounds__(512) void kcode(float* out, unsigned long long* res, float a, float b) {
  float x = (float)threadIdx.x;
  unsigned long long t[4];
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t[0])::"memory");
  for (int pass = 0; pass < 3; ++pass) {
    BODY(4096)
    if (ID) x += 1.0f;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t[pass + 1])::"memory");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
  if (threadIdx.x == 0)
    for (int p = 0; p < 3; ++p) res[3 * blockIdx.x + p] = t[p + 1] - t[p];
}

int main() {
  float* out; unsigned long long* res;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&res, 256 * 3 * 8);
  auto show = [&](const char* name, int blocks) {
    hipDeviceSynchronize();
    unsigned long long h[768]; hipMemcpy(h, res, blocks * 24, hipMemcpyDeviceToHost);
    double p[3] = {0, 0, 0};
    for (int b = 0; b < blocks; ++b) for (int k = 0; k < 3; ++k) p[k] += (double)h[3 * b + k] / blocks;
    printf("%-64s cycles per pass of 4096 instructions (32 KB): %8.0f %8.0f %8.0f  -> cold pass costs %+.0f cycles\n", name, p[0], p[1], p[2], p[0] - p[2]);
  };
  for (int threads : {64, 512}) {
    for (int blocks : {1, 256}) {
      char nm[128];
      hipLaunchKernelGGL(kcode<0>, dim3(blocks), dim3(threads), 0, 0, out, res, 1.0f, 0.5f);
      snprintf(nm, sizeof nm, "%d wave(s) x %3d workgroups, first launch of kernel A", threads / 64, blocks); show(nm, blocks);
      hipLaunchKernelGGL(kcode<0>, dim3(blocks), dim3(threads), 0, 0, out, res, 1.0f, 0.5f);
      snprintf(nm, sizeof nm, "%d wave(s) x %3d workgroups, kernel A again", threads / 64, blocks); show(nm, blocks);
      hipLaunchKernelGGL(kcode<1>, dim3(blocks), dim3(threads), 0, 0, out, res, 1.0f, 0.5f);
      snprintf(nm, sizeof nm, "%d wave(s) x %3d workgroups, kernel B", threads / 64, blocks); show(nm, blocks);
      hipLaunchKernelGGL(kcode<0>, dim3(blocks), dim3(threads), 0, 0, out, res, 1.0f, 0.5f);
      snprintf(nm, sizeof nm, "%d wave(s) x %3d workgroups, kernel A after B", threads / 64, blocks); show(nm, blocks);
      // back to back without a host synchronisation in between (as inside a graph)
      hipLaunchKernelGGL(kcode<1>, dim3(blocks), dim3(threads), 0, 0, out, res, 1.0f, 0.5f);
      hipLaunchKernelGGL(kcode<0>, dim3(blocks), dim3(threads), 0, 0, out, res, 1.0f, 0.5f);
      snprintf(nm, sizeof nm, "%d wave(s) x %3d workgroups, A right behind B (no sync)", threads / 64, blocks); show(nm, blocks);
    }
  }
  return 0;
}
