// Round 6: where does k_gemm_b16's main loop spend its time?  The PRODUCT kernel (csrc/k_gemm_b16.hip, included as it is) on
// configs[4] layer shapes with random bf16 operands, every CU busy, timed in wall time over back-to-back launches, with phases
// switched off at compile time (MDT_UB: 1 = no DMA stream after chunk 0, 2 = no MFMAs / fragment reads, 4 = no epilogue) and the
// clock workgroup 0 ran at (s_memtime / s_memrealtime).  Results are wrong with any bit set: timing only.
//   for b in 0 1 2 4 5 6; do hipcc -O3 -std=c++17 --offload-arch=gfx950 -DMDT_UB=$b -DMDT_UB_CLOCK -DMDT_UB_TILE3 -I moleculediffusiontransformer_amd/csrc \
//       -I include tools/ubench/gemm16_phases.hip -o tools/ubench/bin/gemm16_phases_$b; done
//   tools/ubench/bin/gemm16_phases_<b> [tile: -1 auto, 0 256x256, 1 256x128, 2 128x128, 3 256x256 on four waves] [out16: 1 | 0] [launches per shape]
// The loop variants profiles/r6_ubench_gemm16_phases.txt also lists (fewer fragment reads, two chunks in flight, hand double-buffered
// fragments, no fragment reads, spread requests) were temporary edits of the kernel, measured and removed: DESIGN.md section 9.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
#include "k_gemm_b16.hip"

namespace mdt {
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

static unsigned short rnd_bf16(uint32_t& st, float scale) {
  st = st * 1664525u + 1013904223u;
  const float v = ((int)((st >> 8) & 0xffff) - 32768) * (scale / 32768.0f);
  uint32_t u;
  memcpy(&u, &v, 4);
  return (unsigned short)(u >> 16);
}

int main(int argc, char** argv) {
  struct Shape { int M, N, cin, taps, rows, res; };
  const Shape shapes[] = {{65536, 512, 512, 1, 32, 0},  {65536, 512, 512, 1, 32, 1},  {65536, 1536, 512, 1, 32, 0},
                          {16384, 1024, 2048, 1, 8, 1}, {16384, 1024, 1024, 3, 8, 0}, {16384, 1024, 2048, 3, 8, 0},
                          {65536, 512, 1024, 1, 32, 1}, {16384, 2048, 1024, 1, 8, 0}};
  const int tile = argc > 1 ? atoi(argv[1]) : -1, out16 = argc > 2 ? atoi(argv[2]) : 1;
  mdt::set_tile16(tile);
  printf("MDT_UB = %d  (1: no DMA stream, 2: no MFMA, 4: no epilogue)  tile %d  out16 %d\n", MDT_UB, tile, out16);
  for (const Shape& sh : shapes) {
    const int K = sh.cin * sh.taps;
    std::vector<unsigned short> hA((size_t)sh.M * sh.cin), hW((size_t)sh.N * K), hR((size_t)sh.M * sh.N);
    uint32_t st = 12345u;
    for (auto& v : hA) v = rnd_bf16(st, 1.0f);
    for (auto& v : hW) v = rnd_bf16(st, 0.05f);
    for (auto& v : hR) v = rnd_bf16(st, 1.0f);
    unsigned short *dA, *dW, *dR, *dO;
    float* dB;
    CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dW, hW.size() * 2)); CK(hipMalloc(&dR, hR.size() * 2));
    CK(hipMalloc(&dO, hR.size() * 4)); CK(hipMalloc(&dB, sh.N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dR, hR.data(), hR.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dB, 0, sh.N * 4));
    mdt::Gemm16Args g{};
    g.A = dA; g.W = dW; g.bias = dB; g.res = sh.res ? reinterpret_cast<const float*>(dR) : nullptr; g.out = reinterpret_cast<float*>(dO);
    g.M = sh.M; g.N = sh.N; g.cin = sh.cin; g.taps = sh.taps; g.rows = sh.rows; g.lda = sh.cin; g.a_col = 0;
    g.t_dj = sh.taps > 1 ? 1 : 0; g.t_off = -(sh.taps / 2); g.ldc = sh.N; g.ldr = sh.N; g.o_col = 0; g.act = 0;
    g.out16 = out16; g.res16 = sh.res; g.copy16 = nullptr;
    if (!out16 && sh.res) continue;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    for (int i = 0; i < 5; ++i) CK(mdt::launch_gemm_b16(g, s));
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = argc > 3 ? atoi(argv[3]) : 200;          // (a few thousand for tools/power_clock_sampler.sh: seconds per shape)
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) CK(mdt::launch_gemm_b16(g, s));
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long clk[4];
    CK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(mdt::g_ub_clk), sizeof(clk)));
    const double us = 1e3 * ms / reps, tf = 2.0 * sh.M * sh.N * K / (us * 1e-6) * 1e-12;
    const double wall_us = (clk[3] - clk[2]) * 0.01, ghz = (clk[1] - clk[0]) / (wall_us * 1e3);
    printf("M=%6d N=%5d K=%5d taps=%d res=%d: %8.1f us/launch %7.1f TFLOP/s | workgroup 0 main loop: %7.2f us, %5.0f cycles/chunk at %.2f GHz\n",
           sh.M, sh.N, K, sh.taps, sh.res, us, tf, wall_us, (double)(clk[1] - clk[0]) / (K / 64), ghz);
    hipFree(dA); hipFree(dW); hipFree(dR); hipFree(dO); hipFree(dB);
  }
  return 0;
}
