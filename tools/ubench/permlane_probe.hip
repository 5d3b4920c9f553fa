// What v_permlane16_swap / v_permlane32_swap return when both operands are the same register (gfx950), and the
// xor-16 / xor-32 reductions built from them.
#include <hip/hip_runtime.h>
#include <cstdio>
// lane-group exchanges over +-16 / +-32 lanes with the gfx950 permlane swaps (VALU, no LDS round trip). The swap is in
// place on two registers: fed the same value twice, v_permlane16_swap leaves (rows 0,0,2,2) and (rows 1,1,3,3),
// v_permlane32_swap (halves lo,lo) and (hi,hi); combining the two gives every lane the pair it would get from xor 16 /
// xor 32. Written as asm: through __builtin_amdgcn_permlane*_swap hipcc 7.2 folds the two results into one register.
// The s_nop covers the VALU-write -> permlane-swap-read hazard for the copies the compiler places just before.
#define MDT_XG(NAME, INSN, COMBINE)                                                      \
  __device__ __forceinline__ float NAME(float v) {                                       \
    float a = v, b = v;                                                                  \
    asm("s_nop 1\n\t" INSN " %0, %1" : "+v"(a), "+v"(b));                                \
    return COMBINE;                                                                      \
  }
MDT_XG(xg16_add, "v_permlane16_swap_b32", a + b)
MDT_XG(xg32_add, "v_permlane32_swap_b32", a + b)
MDT_XG(xg16_max, "v_permlane16_swap_b32", fmaxf(a, b))
MDT_XG(xg32_max, "v_permlane32_swap_b32", fmaxf(a, b))
#undef MDT_XG
__global__ void k(unsigned* out, float* f) {
  const unsigned v = threadIdx.x;
  auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  auto q = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1]; out[128 + threadIdx.x] = q[0]; out[192 + threadIdx.x] = q[1];
  float s = (float)threadIdx.x;
  const float a16 = xg16_add(s);
  const float a = xg32_add(a16);
  f[threadIdx.x] = a16; f[64 + threadIdx.x] = a;
  float t = (float)threadIdx.x, u = 100.f + threadIdx.x;        // two values reduced back to back (register pressure / reuse)
  t = xg16_add(t); u = xg16_add(u); t = xg32_add(t); u = xg32_add(u);
  f[128 + threadIdx.x] = t; f[192 + threadIdx.x] = u;
}
int main() {
  unsigned* d; float* f; hipMalloc(&d, 1024); hipMalloc(&f, 1024); k<<<1, 64>>>(d, f);
  unsigned h[256]; float g[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost); hipMemcpy(g, f, 1024, hipMemcpyDeviceToHost);
  const char* names[4] = {"p16 r0", "p16 r1", "p32 r0", "p32 r1"};
  for (int a = 0; a < 4; ++a) { printf("%s:", names[a]); for (int l = 0; l < 64; l += 4) printf(" %2u", h[64 * a + l]); printf("\n"); }
  const char* fn[4] = {"xor16 sum (want i + i^16)", "then xor32 (want 4i' + 96)", "t", "u (want t + 400)"};
  for (int a = 0; a < 4; ++a) { printf("%-28s:", fn[a]); for (int l = 0; l < 64; l += 4) printf(" %5.0f", g[64 * a + l]); printf("\n"); }
  return 0;
}
