#!/usr/bin/env python3
"""Feasibility probe: do two independent half-batch sampling chains on two HIP streams overlap on one GPU?
Runs sample() for B=1024 on one stream, then 2 x B=512 from two host threads on two streams (separate model
instances, hence separate arenas / graphs)."""
import os
import sys
import threading
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import gpu_util
from gpu_util import make_model
from moleculediffusiontransformer_amd import NoiseSource
from moleculediffusiontransformer_amd.synth import synth_normal

dev = torch.device("cuda", 0)
gpu_util.DEV = "cuda:0"
T = int(os.environ.get("T", "16"))
models = [make_model("cfg1") for _ in range(2)]


def run(model, seq, stream, reps, sample0):
    with torch.cuda.stream(stream):
        for r in range(reps):
            model.sample(seq, dev, cond_scale=1.0, timesteps=T, clamp=False, noise=NoiseSource(seed=7 + r, sample0=sample0))


for nchain in (1, 2, 1, 2):
    B = 1024 // nchain
    seqs = [synth_normal(f"p/{c}", (B, 12)).to(dev) for c in range(nchain)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(nchain)]
    for c in range(nchain):           # warm-up / graph capture, one chain at a time
        run(models[c], seqs[c], streams[c], 1, c * B)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(models[c], seqs[c], streams[c], 3, c * B)) for c in range(nchain)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    evals = 2 * (T - 1) * 3
    print(f"chains={nchain} B/chain={B}: {1024 * 3 / dt * (T - 1) / 63:8.1f} mol/s-equivalent@64 steps, {dt / evals * 1e3:6.3f} ms per (full-batch) eval", flush=True)
