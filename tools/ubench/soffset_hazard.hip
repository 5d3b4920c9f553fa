// Isolation of what broke the first form of k_tf256.hip's pair hand-off: buffer instructions whose SCALAR offset operand is an
// SGPR that a scalar add rewrites between consecutive instructions.  No second workgroup, no flags: every wave writes 8 x 1 KB
// "tiles" (tile c at byte offset base + 1024 c, lane * 16 inside) and 8 more of a second kind, drains, reads everything back with the
// whole offset in the VECTOR operand and checks (round, workgroup, wave, tile, lane) of every 16-byte piece; 256 workgroups of 8
// waves, many rounds, so that the vector-memory queues are under pressure.
//   FORM 0: __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, lane * 16, sbase + 1024 c, sc1)   -- what hipcc makes of the first form
//           (ISA: s_add_i32 s5, s4, imm ; buffer_store_dwordx4 ..., s5 offen sc1 ; s_add_i32 s5, s4, imm' ; buffer_store ... s5)
//   FORM 1: the same with the whole offset in the vector operand (scalar offset 0)             -- the shipped form
//   FORM 2: inline asm, ONE SGPR rewritten between the stores (store ; s_add ; s_nop 7 ; store ...): padding between the scalar add and
//           the store that READS its result, none between a store and the add that OVERWRITES its offset register
//   FORM 3: inline asm, ONE SGPR rewritten (store ; s_nop 7 ; s_add ; store ...): padding between a store and the add that overwrites its
//           offset register, none between the add and the store that reads its result
//   FORM 4: inline asm, EIGHT different SGPRs written up front, stores back to back
// Build & run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/ubench/soffset_hazard.hip -o /tmp/sh && /tmp/sh
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

template <int FORM>
__global__ __launch_bounds__(512) void kform(float* buf, unsigned* errs, float* dbg, int rounds) {
  const unsigned wg = blockIdx.x, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 0x7fffffff, 0x00020000);
  unsigned bad = 0;
  for (int r = 1; r <= rounds; ++r) {
    // 16 tiles per wave and round parity: [parity][wg][wave][16 tiles][1 KB]
    const unsigned sbase = __builtin_amdgcn_readfirstlane(((((unsigned)(r & 1) * gridDim.x + wg) * 8u + wave) * 16u) * 1024u);
    const unsigned vlane = lane * 16u;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      f4 v[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = f4{(float)r, (float)(wg * 8u + wave), (float)(8 * pass + c), (float)lane};
#pragma unroll
      for (int c = 0; c < 8; ++c) asm volatile("" : "+v"(v[c]));      // eight live register quads: no VALU write between the stores
      const unsigned pb = sbase + 8192u * pass;
      if (FORM == 0) {
#pragma unroll
        for (int c = 0; c < 8; ++c) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i4, v[c]), rs, vlane, pb + 1024u * c, 16);
      } else if (FORM == 1) {
#pragma unroll
        for (int c = 0; c < 8; ++c) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i4, v[c]), rs, vlane + pb + 1024u * c, 0, 16);
      } else if (FORM == 2 || FORM == 3) {
        unsigned so = pb;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          if (FORM == 2)
            asm volatile("s_nop 7\n\tbuffer_store_dwordx4 %1, %2, %3, %0 offen sc1\n\ts_add_u32 %0, %0, 0x400"
                         : "+s"(so) : "v"(v[c]), "v"(vlane), "s"(rs) : "memory", "scc");
          else
            asm volatile("buffer_store_dwordx4 %1, %2, %3, %0 offen sc1\n\ts_nop 7\n\ts_add_u32 %0, %0, 0x400"
                         : "+s"(so) : "v"(v[c]), "v"(vlane), "s"(rs) : "memory", "scc");
        }
      } else {
        unsigned so[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) { so[c] = pb + 1024u * c; asm volatile("" : "+s"(so[c])); }
        asm volatile("s_nop 7" ::: "memory");
#pragma unroll
        for (int c = 0; c < 8; ++c)
          asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen sc1\n\ts_nop 1" : : "v"(v[c]), "v"(vlane), "s"(rs), "s"(so[c]) : "memory");
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const f4 g = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, vlane + sbase + 1024u * c, 0, 16));
      if (g[0] != (float)r || g[1] != (float)(wg * 8u + wave) || g[2] != (float)c || g[3] != (float)lane) {
        if (!bad) {
          const unsigned slot = atomicAdd(errs + 1, 1u);
          if (slot < 4) { float* d = dbg + 8 * slot; d[0] = (float)r; d[1] = (float)(wg * 8u + wave); d[2] = (float)c; d[3] = (float)lane;
                          d[4] = g[0]; d[5] = g[1]; d[6] = g[2]; d[7] = g[3]; }
        }
        ++bad;
      }
    }
    __syncthreads();
  }
  if (bad) atomicAdd(errs, bad);
}

int main() {
  const int nwg = 256, rounds = 2000;
  float *buf, *dbg; unsigned* errs;
  hipMalloc(&buf, (size_t)2 * nwg * 8 * 16 * 1024); hipMalloc(&errs, 256); hipMalloc(&dbg, 256);
  const char* names[5] = {"builtin, wave-uniform part in the scalar offset (hipcc rewrites one SGPR between the stores)",
                          "builtin, whole offset in the vector operand", "asm, one SGPR rewritten, s_nop 7 between the add and the store reading it",
                          "asm, one SGPR rewritten, s_nop 7 between a store and the add overwriting its offset", "asm, eight SGPRs written up front"};
  for (int form = 0; form < 5; ++form) {
    hipMemset(errs, 0, 256); hipMemset(buf, 0, (size_t)2 * nwg * 8 * 16 * 1024);
#define GO(F) hipLaunchKernelGGL(kform<F>, dim3(nwg), dim3(512), 0, 0, buf, errs, dbg, rounds)
    if (form == 0) GO(0); else if (form == 1) GO(1); else if (form == 2) GO(2); else if (form == 3) GO(3); else GO(4);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    unsigned e[2]; float hd[32];
    hipMemcpy(e, errs, 8, hipMemcpyDeviceToHost); hipMemcpy(hd, dbg, 128, hipMemcpyDeviceToHost);
    printf("form %d (%s): %u wrong pieces of %llu\n", form, names[form], e[0], (unsigned long long)nwg * 512 * 16 * rounds);
    for (unsigned i = 0; i < (e[1] < 3 ? e[1] : 3); ++i)
      printf("   round %g wave %g tile %g lane %g: got (round %g wave %g tile %g lane %g)\n", hd[8*i], hd[8*i+1], hd[8*i+2], hd[8*i+3], hd[8*i+4],
             hd[8*i+5], hd[8*i+6], hd[8*i+7]);
  }
  return 0;
}
