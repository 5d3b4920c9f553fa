// Where does a launch of k_attn_ctx (csrc/k_attn.hip, the LDS-streamed form of round 4) spend its time?  The product kernel, compiled
// here with its timing-only switches (-DMDT_TUNING -DMDT_ABL_CTX_*: WRONG results), at configs[2]'s shapes.
//
//   for v in "" -DMDT_ABL_CTX_NOSM -DMDT_ABL_CTX_NODMA -DMDT_ABL_CTX_OCC1 -DMDT_ABL_CTX_NOS -DMDT_ABL_CTX_NOPV; do
//     hipcc -O3 -std=c++17 --offload-arch=gfx950 -DMDT_TUNING $v -Imoleculediffusiontransformer_amd/csrc -Iinclude \
//         tools/ubench/attn_ctx_probe.hip -o /tmp/attn_ctx_probe && /tmp/attn_ctx_probe; done
#include "../../moleculediffusiontransformer_amd/csrc/k_attn.hip"

#include <vector>

int main() {
  const int B = 4096, H = 8, Tk = 64, F = 128;
  for (int T : {4, 1}) {
    const size_t nq = (size_t)B * T * H * F, nc = (size_t)B * Tk * F;
    float *q, *c, *o;
    (void)hipMalloc(&q, nq * 4); (void)hipMalloc(&c, nc * 4); (void)hipMalloc(&o, nq * 4);
    std::vector<float> hq(nq), hc(nc);
    for (size_t i = 0; i < nq; ++i) hq[i] = 0.3f * (float)((i * 2654435761u >> 8) % 2001 - 1000) / 1000.f;
    for (size_t i = 0; i < nc; ++i) hc[i] = (float)((i * 40503u >> 4) % 2001 - 1000) / 1000.f;
    (void)hipMemcpy(q, hq.data(), nq * 4, hipMemcpyHostToDevice); (void)hipMemcpy(c, hc.data(), nc * 4, hipMemcpyHostToDevice);
    mdt::AttnArgs a{};
    a.q = q; a.k = c; a.out = o; a.batch = B; a.T = T; a.Tk = Tk; a.heads = H; a.ldq = F; a.ldkv = F; a.ldo = F; a.kv_bstride = Tk;
    a.scale = 0.125f;
#ifdef CTXP_SPLIT
    a.split_scores = 1;           // the default mode's form: split-bf16 scores
#endif
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) (void)mdt::launch_attn_ctx(a, 0);
    (void)hipEventRecord(e0, 0);
    const int reps = 50;
    for (int i = 0; i < reps; ++i) (void)mdt::launch_attn_ctx(a, 0);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)(nc + 2 * nq) * 4;
    printf("T=%d: %.1f us per launch, %.2f TB/s algorithmic\n", T, ms * 1000 / reps, bytes / (ms / reps * 1e-3) / 1e12);
#ifdef MDT_ABL_CTX_CLOCK
    static unsigned long long st[4 + 2048];
    (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(mdt::g_ctx_stamp), sizeof(st));
    {
      unsigned long long lo = ~0ull, hi = 0, smax = 0, emin = ~0ull;
      const int nwg = 512;
      for (int w = 0; w < nwg; ++w) {
        lo = std::min(lo, st[4 + 2 * w]); hi = std::max(hi, st[5 + 2 * w]);
        smax = std::max(smax, st[4 + 2 * w]); emin = std::min(emin, st[5 + 2 * w]);
      }
      double life = 0;
      for (int w = 0; w < nwg; ++w) life += (double)(st[5 + 2 * w] - st[4 + 2 * w]);
      printf("      all workgroups (100 MHz clock): first start -> last end %.2f us, last start +%.2f us, first end +%.2f us, mean lifetime %.2f us\n",
             (hi - lo) / 100.0, (smax - lo) / 100.0, (emin - lo) / 100.0, life / nwg / 100.0);
    }
    printf("      workgroup 0: %llu shader-clock ticks in %llu ticks of the 100 MHz clock = %.2f us -> %.3f GHz\n", st[0], st[1], st[1] / 100.0,
           st[0] / (st[1] * 10.0));
#endif
    (void)hipFree(q); (void)hipFree(c); (void)hipFree(o);
  }
  return 0;
}
