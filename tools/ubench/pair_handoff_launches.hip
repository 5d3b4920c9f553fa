// Does the sc1 hand-off of pair_handoff.hip stay valid ACROSS LAUNCHES -- monotonic flag words read back at kernel entry, the
// same hand-off blocks re-used, and the workgroup -> XCD placement / the partner changing from launch to launch (stride 1, 8,
// 128, 2, 16, 64 in turn, other kernels of other grid sizes in between)?  k_tf256.hip's pair-split form depends on it.
//   EPOCH 0: the flag's value at entry is read with an sc1 load        EPOCH 1: with an agent-scope atomic (fetch_add 0)
//   INV   1: every wave runs buffer_inv sc1 + vmcnt(0) at kernel entry (agent acquire)
// Every 16-byte piece read is checked: (absolute round, writer, piece).  Build & run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/pair_handoff_launches.hip -o /tmp/phl && /tmp/phl
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

__global__ void filler(float* p, int n) {   // a kernel of another grid size between the hand-off launches
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0001f + 1.f;
}

template <int EPOCH, int INV>
__global__ __launch_bounds__(512) void kpair(float* buf, unsigned* flags, unsigned* errs, float* dbg, int rounds, int stride) {
  const unsigned nwg = gridDim.x, me = blockIdx.x, other = me ^ (unsigned)stride;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc(flags, 0, 0x7fffffff, 0x00020000);
  if (INV) asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
  unsigned r0;
  if (EPOCH == 0) r0 = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(fr, 0, 128 * me, 16 | (1 << 31));
  else r0 = __hip_atomic_fetch_add(flags + 32 * me, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  r0 = __builtin_amdgcn_readfirstlane(r0);
  unsigned bad = 0;
  for (int k = 1; k <= rounds; ++k) {
    const unsigned r = r0 + (unsigned)k;
    const unsigned mine = ((unsigned)(k & 1) * nwg + me) * 32768u + threadIdx.x * 16u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f4 v = f4{(float)r, (float)me, (float)(threadIdx.x + 512 * j), 1.f};
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i4, v), rs, mine + 8192u * j, 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x < 64) {
      __builtin_amdgcn_raw_buffer_store_b32((int)r, fr, 0, 128 * me, 16);
      for (;;) {
        const unsigned got = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(fr, 0, 128 * other, 16 | (1 << 31));
        if ((int)(got - r) >= 0) break;
        __builtin_amdgcn_s_sleep(2);
      }
    }
    __syncthreads();
    const unsigned theirs = ((unsigned)(k & 1) * nwg + other) * 32768u + threadIdx.x * 16u;
    f4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, theirs + 8192u * j, 0, 16 | (1 << 31)));
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (v[j][0] != (float)r || v[j][1] != (float)other || v[j][2] != (float)(threadIdx.x + 512 * j)) {
        if (!bad) {
          const unsigned slot = atomicAdd(errs + 1, 1u);
          if (slot < 8) { float* d = dbg + 8 * slot; d[0] = (float)r; d[1] = (float)me; d[2] = (float)other; d[3] = (float)k;
                          d[4] = v[j][0]; d[5] = v[j][1]; d[6] = v[j][2]; d[7] = (float)r0; }
        }
        ++bad;
      }
  }
  if (bad) atomicAdd(errs, bad);
}

int main() {
  const int nwg = 256, rounds = 7, launches = 600;
  float *buf, *dbg, *junk; unsigned *flags, *errs;
  hipMalloc(&buf, (size_t)2 * nwg * 32768); hipMalloc(&flags, nwg * 128); hipMalloc(&errs, 256); hipMalloc(&dbg, 256);
  hipMalloc(&junk, 1 << 24);
  hipMemset(junk, 0, 1 << 24);
  const int strides[6] = {1, 8, 128, 2, 16, 64};
  for (int cfg = 0; cfg < 4; ++cfg) {
    const int epoch = cfg & 1, inv = cfg >> 1;
    for (int vary = 0; vary < 2; ++vary) {
      hipMemset(flags, 0, nwg * 128); hipMemset(errs, 0, 256); hipMemset(buf, 0, (size_t)2 * nwg * 32768);
      for (int l = 0; l < launches; ++l) {
        const int stride = vary ? strides[l % 6] : 8;
        if (vary) hipLaunchKernelGGL(filler, dim3(37 + 13 * (l % 11)), dim3(256), 0, 0, junk, 1 << 20);
#define GO(E, I) hipLaunchKernelGGL((kpair<E, I>), dim3(nwg), dim3(512), 0, 0, buf, flags, errs, dbg, rounds, stride)
        if (cfg == 0) GO(0, 0); else if (cfg == 1) GO(1, 0); else if (cfg == 2) GO(0, 1); else GO(1, 1);
      }
      if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
      unsigned e[2]; float hd[64];
      hipMemcpy(e, errs, 8, hipMemcpyDeviceToHost); hipMemcpy(hd, dbg, 256, hipMemcpyDeviceToHost);
      printf("epoch read by %s, %s at entry, %s: %u mismatches of %u checks over %d launches x %d rounds\n", epoch ? "atomic" : "sc1 load",
             inv ? "buffer_inv sc1" : "no invalidate", vary ? "partner / placement varies, other kernels between" : "fixed partner (id ^ 8)", e[0],
             (unsigned)(nwg * 512 * 4) * rounds * launches, launches, rounds);
      for (unsigned i = 0; i < (e[1] < 4 ? e[1] : 4); ++i)
        printf("   expected round %g writer %g (me %g, local round %g, r0 %g): got round %g writer %g piece %g\n", hd[8*i], hd[8*i+2], hd[8*i+1],
               hd[8*i+3], hd[8*i+7], hd[8*i+4], hd[8*i+5], hd[8*i+6]);
    }
  }
  return 0;
}
