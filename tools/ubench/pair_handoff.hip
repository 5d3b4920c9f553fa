// Hand-off of 32 KB each way between the two workgroups of a PAIR inside one launch, per round; every 16-byte piece read is
// checked against what the partner must have written in THIS round (round number, writer id, piece index).
//
// Supersedes pair_exchange.hip, whose "partner = workgroup + 1" case was a RING (me reads me + 1, me + 1 reads me + 2): a
// writer only waited for ITS reader's flag... of the wrong workgroup, so its round r + 2 data could overwrite round r data that
// was still being read -- the "3 % stale pieces cross-XCD" of round 2 were that protocol error (the fenced reference mode
// showed mismatches too), not incoherent L2s.  Here partners are symmetric (me ^ stride) for every stride.
//
//   MODE 0: plain stores / loads + __threadfence() both sides, release / acquire flag atomics (reference)
//   MODE 1: sc1 stores (16 B), every storing wave s_waitcnt vmcnt(0), workgroup barrier, ONE lane sc1 flag store; ONE lane
//           sc1 poll, workgroup barrier, sc1 loads (16 B)   -- MI355X_MICROARCH.md "Hand-offs measured with sc1 loads", row 1
//   MODE 2: as 1 + agent acquire (buffer_inv sc1, vmcnt(0)) by every wave behind the poll's barrier
//   MODE 4: as 1 through the compiler: __builtin_amdgcn_raw_buffer_store / load_b128 with aux = sc1 (waits and hazards tracked)
//   MODE 5: as 4 with the wave-uniform part of every address in the instruction's SCALAR offset (an SGPR rewritten between
//           consecutive buffer instructions) and only the lane's 16 bytes in the vector offset -- the form k_tf256.hip first used
//   MODE 6: as 4 with PLAIN stores (aux = 0: the lines stay in the XCD's L2) and sc1 loads (L1 bypassed, L2-served): coherent only
//           between workgroups of ONE XCD -- what a same-XCD fast path of the hand-off would cost (and what it does across XCDs)
//   MODE 3: plain stores, vmcnt(0), barrier, lane-0 agent release (buffer_wbl2 sc1, vmcnt(0)), sc1 flag; poll, agent acquire,
//           vmcnt(0), barrier, plain loads               -- the guide's "Valid forms", producer / consumer bullets
// stride 1: partners on different XCDs (workgroups are dealt round robin), 8 / 128: same XCD; the XCC ids are read back
// (HW_REG_XCC_ID) and the number of pairs whose ids differ is printed.  skew = 1: uneven load (a third of the workgroups
// sleep a pseudo-random time and stream 64 KB of unrelated memory every round); the consumer's addresses repeat every second
// round (double buffer), so stale L1 / L2 copies of them exist by construction.
// Build & run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/ubench/pair_handoff.hip -o /tmp/ph && /tmp/ph
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

// (s_nop 1 behind the store: hipcc's hazard recognizer does not see into inline asm, and a VALU write to the data registers of a
//  128-bit store within two wait states of it corrupts what is stored -- the first version of this probe showed that as 12 % wrong
//  piece indices in every sc1 mode, same XCD or not)
#define ST_SC1() asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory")
#define LD4_SC1()                                                                                                          \
  asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"                            \
               "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"           \
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(q), "v"(q + 512), "v"(q + 1024), "v"(q + 1536) : "memory")

template <int MODE>
__global__ __launch_bounds__(512) void kpair(float* buf, unsigned* flags, unsigned* errs, unsigned long long* res, unsigned* xcc,
                                             const float* junk, float* sink, float* dbg, int rounds, int stride, int skew) {
  const unsigned nwg = gridDim.x, me = blockIdx.x, other = me ^ (unsigned)stride;
  if (threadIdx.x == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc[me] = id & 15u;
  }
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 0x7fffffff, 0x00020000);
  unsigned long long t0, t1;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  unsigned bad = 0;
  float acc = 0.f;
  for (int r = 1; r <= rounds; ++r) {
    if (skew && ((me * 2654435761u + (unsigned)r * 40503u) >> 13) % 3u == 0u) {      // uneven load
      const unsigned n = ((me * 97u + (unsigned)r * 31u) & 15u) + 1u;
      for (unsigned k = 0; k < n; ++k) __builtin_amdgcn_s_sleep(32);
      const f4* jp = reinterpret_cast<const f4*>(junk) + ((size_t)((me * 7u + (unsigned)r) % nwg)) * 4096 + threadIdx.x;
      for (int k = 0; k < 8; ++k) { const f4 j = jp[512 * k]; acc += j[0] + j[3]; }
    }
    f4* mine = reinterpret_cast<f4*>(buf) + ((size_t)(r & 1) * nwg + me) * 2048 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f4* p = mine + 512 * k;
      const f4 v = f4{(float)r, (float)me, (float)(threadIdx.x + 512 * k), 1.f};
      if (MODE == 0 || MODE == 3) *p = v;
      else if (MODE == 4) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i4, v), rs, (unsigned)((const char*)p - (const char*)buf), 0, 16);
      else if (MODE == 6) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i4, v), rs, (unsigned)((const char*)p - (const char*)buf), 0, 0);
      else if (MODE == 5) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i4, v), rs, (threadIdx.x & 63u) * 16u,
                              __builtin_amdgcn_readfirstlane((unsigned)((const char*)p - (const char*)buf) - (threadIdx.x & 63u) * 16u), 16);
      else ST_SC1();
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
        if (MODE == 3) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        unsigned rr = (unsigned)r, got;
        asm volatile("global_store_dword %0, %1, off sc1" :: "v"(fp), "v"(rr) : "memory");
        do {
          asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(got) : "v"(fq) : "memory");
          if (got < rr) __builtin_amdgcn_s_sleep(1);
        } while (got < rr);
      }
    }
    if (MODE == 3) {       // every wave: one acquire behind the poll (the polling wave's own loads would not need the barrier)
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (MODE == 0) __threadfence();
    if (MODE == 2) asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
    const f4* q = reinterpret_cast<const f4*>(buf) + ((size_t)(r & 1) * nwg + other) * 2048 + threadIdx.x;
    f4 v[4];
    if (MODE == 0 || MODE == 3) { for (int k = 0; k < 4; ++k) v[k] = q[512 * k]; }
    else if (MODE == 4 || MODE == 6) { for (int k = 0; k < 4; ++k) v[k] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)((const char*)(q + 512 * k) - (const char*)buf), 0, 16)); }
    else if (MODE == 5) { for (int k = 0; k < 4; ++k) v[k] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (threadIdx.x & 63u) * 16u,
                              __builtin_amdgcn_readfirstlane((unsigned)((const char*)(q + 512 * k) - (const char*)buf) - (threadIdx.x & 63u) * 16u), 16)); }
    else LD4_SC1();
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (v[k][0] != (float)r || v[k][1] != (float)other || v[k][2] != (float)(threadIdx.x + 512 * k)) {
        if (!bad) {
          const unsigned slot = atomicAdd(errs + 1, 1u);
          if (slot < 16) { float* d = dbg + 8 * slot; d[0] = (float)r; d[1] = (float)me; d[2] = (float)threadIdx.x; d[3] = (float)k;
                           d[4] = v[k][0]; d[5] = v[k][1]; d[6] = v[k][2]; d[7] = v[k][3]; }
        }
        ++bad;
      }
  }
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (bad) atomicAdd(errs, bad);
  if (threadIdx.x == 0) res[me] = t1 - t0;
  if (acc == 12345.678f) sink[me] = acc;
}

int main(int argc, char** argv) {
  const int nwg = 256, rounds = 400;
  float *buf, *junk, *sink, *dbg; unsigned *flags, *errs, *xcc; unsigned long long* res;
  hipMalloc(&buf, (size_t)2 * nwg * 32768); hipMalloc(&flags, nwg * 128); hipMalloc(&errs, 256); hipMalloc(&res, nwg * 8);
  hipMalloc(&xcc, nwg * 4); hipMalloc(&junk, (size_t)nwg * 65536); hipMalloc(&sink, nwg * 4); hipMalloc(&dbg, 16 * 8 * 4);
  hipMemset(junk, 0, (size_t)nwg * 65536);
  const char* names[7] = {"plain + __threadfence both sides", "sc1 stores / flag / loads, no fence", "sc1 + buffer_inv sc1 at the reader",
                          "plain stores + agent release; agent acquire + plain loads", "sc1 through raw_buffer builtins (aux = sc1), no fence", "as 4, wave-uniform part of the address in the scalar offset",
                          "as 4 with PLAIN stores (same-XCD form: lines kept in the shared L2), sc1 flag and loads"};
  for (int skew = 0; skew < 2; ++skew)
    for (int mode = (argc > 1 ? atoi(argv[1]) : 0); mode < (argc > 2 ? atoi(argv[2]) : 7); ++mode)
      for (int stride : {1, 8, 128}) {
        hipMemset(flags, 0, nwg * 128); hipMemset(errs, 0, 256); hipMemset(buf, 0, (size_t)2 * nwg * 32768);
#define GO(M) hipLaunchKernelGGL(kpair<M>, dim3(nwg), dim3(512), 0, 0, buf, flags, errs, res, xcc, junk, sink, dbg, rounds, stride, skew)
        if (mode == 0) GO(0); else if (mode == 1) GO(1); else if (mode == 2) GO(2); else if (mode == 3) GO(3); else if (mode == 4) GO(4); else if (mode == 5) GO(5); else GO(6);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        unsigned long long h[256]; unsigned e, x[256];
        hipMemcpy(h, res, nwg * 8, hipMemcpyDeviceToHost); hipMemcpy(&e, errs, 4, hipMemcpyDeviceToHost); hipMemcpy(x, xcc, nwg * 4, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < nwg; ++i) s += h[i];
        if (e) { float hd[128]; hipMemcpy(hd, dbg, sizeof hd, hipMemcpyDeviceToHost);
                 for (int i = 0; i < 6; ++i) printf("   r %g me %g tid %g k %g: got (%g %g %g %g)\n", hd[8*i], hd[8*i+1], hd[8*i+2], hd[8*i+3], hd[8*i+4], hd[8*i+5], hd[8*i+6], hd[8*i+7]); }
        int differ = 0, modmatch = 0;
        for (int i = 0; i < nwg; ++i) { differ += x[i] != x[i ^ stride]; modmatch += (x[i] == x[i % 8]); }
        printf("skew %d mode %d (%s), partner = id ^ %3d: %7.2f us per round, %u mismatches of %u checks; %d of %d workgroups have a partner "
               "on another XCD; xcc(id) == xcc(id %% 8) for %d of %d\n", skew, mode, names[mode], stride, s / nwg / rounds / 100.0, e,
               (unsigned)(nwg * 512 * 4) * rounds, differ, nwg, modmatch, nwg);
      }
  return 0;
}
