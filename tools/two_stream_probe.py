#!/usr/bin/env python3
"""Premise probe (GPU box): does ONE sample() call of B molecules finish sooner as TWO half-batches in flight on two HIP streams?

    python tools/two_stream_probe.py [B=1024] [timesteps=64] [calls=3]

At B = 1024 the narrow program's launches are either 128..256-workgroup ring kernels that take 75..230 us or latency-bound launches
(30 x k_rconv at ~11 us: 19 % of an evaluation, a ring kernel's prologue + drain ~6-8 us).  Two half-batches on two streams put one
half's latency-bound launches beside the other half's streamed phases.  Per-sample arithmetic does not depend on the batch, so the
result is the unsplit call's, bit for bit (checked here).  Two host threads, one per stream; each thread owns a model (engine)."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from moleculediffusiontransformer_amd import NoiseSource  # noqa: E402
from moleculediffusiontransformer_amd.synth import make_synth_model, synth_normal  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    calls = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    nsplit = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    dev = torch.device("cuda:0")
    seq = synth_normal("probe/seq", (B, 12))
    whole = make_synth_model("cfg1", dev)
    whole.kernel_choice = "narrow"
    parts = [make_synth_model("cfg1", dev) for _ in range(nsplit)]
    for p in parts:
        p.kernel_choice = "narrow"
    hb = B // nsplit

    def run_whole(k):
        return whole.sample(seq, dev, cond_scale=1.0, timesteps=T, noise=NoiseSource(seed=100 + k, sample0=0))

    outs = [None] * nsplit

    def run_part(i, k, stream):
        with torch.cuda.stream(stream):
            outs[i] = parts[i].sample(seq[i * hb:(i + 1) * hb], dev, cond_scale=1.0, timesteps=T,
                                      noise=NoiseSource(seed=100 + k, sample0=i * hb))
            stream.synchronize()

    streams = [torch.cuda.Stream(device=dev) for _ in range(nsplit)]

    def run_split(k):
        th = [threading.Thread(target=run_part, args=(i, k, streams[i])) for i in range(nsplit)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        return torch.cat(outs)

    run_whole(0)                            # compile, graphs
    for i in range(nsplit):                 # (graph captures one at a time: concurrent captures from two threads fail in torch)
        run_part(i, 0, streams[i])
    torch.cuda.synchronize()
    res = {}
    for name, fn in (("whole", run_whole), ("split", run_split), ("whole", run_whole), ("split", run_split)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(calls):
            o = fn(1 + k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / calls
        res.setdefault(name, []).append(dt)
        print(f"{name}: {1e3 * dt:8.2f} ms per call  {B / dt:8.1f} molecules/s", flush=True)
    a, b = run_whole(7), run_split(7)
    print("bitwise equal:", bool(torch.equal(a, b)), " max abs diff", float((a - b).abs().max()))
    print("handoff status", [p._engine.handoff_status() for p in parts], whole._engine.handoff_status())


if __name__ == "__main__":
    main()
