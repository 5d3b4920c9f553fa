// Can two workgroups hand 32 KB to each other inside a launch without the agent-scope fences that cost 68 us (grid_barrier.hip)?
// Per round: every workgroup stores 32 KB, waits for the stores (vmcnt 0), publishes the round number in its flag word, polls the
// partner's flag and reads the partner's 32 KB; every value read is checked against what the partner must have written in THIS round.
//   MODE 0: plain stores / loads + __threadfence() on both sides (the reference: must show 0 mismatches)
//   MODE 1: sc1 stores, sc1 flag, sc1 loads, no fence      MODE 2: sc0 sc1 (system scope) everywhere, no fence
//   MODE 3: as 1, plus `buffer_inv sc1` (invalidate, no write-back) on the reader's side once the flag is seen
// Partner = workgroup + 128 (same XCD if workgroups go round robin over the 8 XCDs) or + 1 (another XCD).
// Build & run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/ubench/pair_exchange.hip -o /tmp/px && /tmp/px
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

#define ST(BITS) asm volatile("global_store_dwordx4 %0, %1, off " BITS :: "v"(p), "v"(v) : "memory")
#define LD4(BITS)                                                                                                          \
  asm volatile("global_load_dwordx4 %0, %4, off " BITS "\n\tglobal_load_dwordx4 %1, %5, off " BITS "\n\t"                   \
               "global_load_dwordx4 %2, %6, off " BITS "\n\tglobal_load_dwordx4 %3, %7, off " BITS "\n\ts_waitcnt vmcnt(0)" \
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(q), "v"(q + 512), "v"(q + 1024), "v"(q + 1536) : "memory")

template <int MODE>
__global__ __launch_bounds__(512) void kpair(float* buf, unsigned* flags, unsigned* errs, unsigned long long* res, int rounds, int partner_off) {
  const unsigned nwg = gridDim.x, me = blockIdx.x, other = (me + partner_off) % nwg;
  unsigned long long t0, t1;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  unsigned bad = 0;
  for (int r = 1; r <= rounds; ++r) {
    f4* mine = reinterpret_cast<f4*>(buf) + ((size_t)(r & 1) * nwg + me) * 2048 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f4* p = mine + 512 * k;
      const f4 v = f4{(float)r, (float)me, (float)(threadIdx.x + 512 * k), 1.f};
      if (MODE == 0) *p = v; else if (MODE == 1 || MODE == 3) ST("sc1"); else ST("sc0 sc1");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 0) __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned* fp = flags + 32 * me;
      const unsigned* fq = flags + 32 * other;
      if (MODE == 0) {
        __hip_atomic_store(fp, (unsigned)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(fq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r) __builtin_amdgcn_s_sleep(1);
      } else {
        unsigned rr = (unsigned)r, got;
        if (MODE == 1 || MODE == 3) asm volatile("global_store_dword %0, %1, off sc1" :: "v"(fp), "v"(rr) : "memory");
        else asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(fp), "v"(rr) : "memory");
        do {
          if (MODE == 1 || MODE == 3) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(got) : "v"(fq) : "memory");
          else asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(got) : "v"(fq) : "memory");
          if (got < rr) __builtin_amdgcn_s_sleep(1);
        } while (got < rr);
      }
    }
    __syncthreads();
    if (MODE == 0) __threadfence();
    if (MODE == 3) asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
    const f4* q = reinterpret_cast<const f4*>(buf) + ((size_t)(r & 1) * nwg + other) * 2048 + threadIdx.x;
    f4 v[4];
    if (MODE == 0) { for (int k = 0; k < 4; ++k) v[k] = __builtin_nontemporal_load(q + 512 * k); }
    else if (MODE == 1 || MODE == 3) LD4("sc1");
    else LD4("sc0 sc1");
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (v[k][0] != (float)r || v[k][1] != (float)other) ++bad;        // stale or foreign data
  }
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (bad) atomicAdd(errs, bad);
  if (threadIdx.x == 0) res[me] = t1 - t0;
}

int main() {
  const int nwg = 256, rounds = 200;
  float* buf; unsigned *flags, *errs; unsigned long long* res;
  hipMalloc(&buf, (size_t)2 * nwg * 32768); hipMalloc(&flags, nwg * 128); hipMalloc(&errs, 256); hipMalloc(&res, nwg * 8);
  for (int mode = 0; mode < 4; ++mode)
    for (int off : {128, 1}) {
      hipMemset(flags, 0, nwg * 128); hipMemset(errs, 0, 256); hipMemset(buf, 0, (size_t)2 * nwg * 32768);
      if (mode == 0) hipLaunchKernelGGL(kpair<0>, dim3(nwg), dim3(512), 0, 0, buf, flags, errs, res, rounds, off);
      else if (mode == 1) hipLaunchKernelGGL(kpair<1>, dim3(nwg), dim3(512), 0, 0, buf, flags, errs, res, rounds, off);
      else if (mode == 2) hipLaunchKernelGGL(kpair<2>, dim3(nwg), dim3(512), 0, 0, buf, flags, errs, res, rounds, off);
      else hipLaunchKernelGGL(kpair<3>, dim3(nwg), dim3(512), 0, 0, buf, flags, errs, res, rounds, off);
      hipDeviceSynchronize();
      unsigned long long h[256]; unsigned e; hipMemcpy(h, res, nwg * 8, hipMemcpyDeviceToHost); hipMemcpy(&e, errs, 4, hipMemcpyDeviceToHost);
      double s = 0; for (int i = 0; i < nwg; ++i) s += h[i];
      printf("mode %d (%s), partner = workgroup + %3d: %6.2f us per round of 32 KB each way, %u mismatches of %u checks\n", mode,
             mode == 0 ? "plain accesses + __threadfence" : mode == 1 ? "sc1 accesses, no fence" : mode == 2 ? "sc0 sc1 accesses, no fence" : "sc1 accesses + buffer_inv sc1 at the reader", off,
             s / nwg / rounds / 100.0, e, (unsigned)(nwg * 512 * 4) * rounds);
    }
  return 0;
}
