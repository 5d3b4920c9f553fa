#!/usr/bin/env python3
"""Stress of the in-launch hand-offs at MODEL level (GPU box), round 6: BASELINE configs[1] at B = 1024 -- every launch of the 256-channel
level pair-split, k_tf256 NSPLIT = 2 and k_res256 NSPLIT = 2 -- 64 timesteps (126 evaluations x 9 pair-split launches x 5..16
hand-offs each) with the counter-based noise, N calls with the SAME seed: every call must return the first call's bits, the status
word must stay 0.  Also with the partners on different XCDs (pair stride 1).   python tools/stress_pair_split.py [calls=20]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import DEV, make_model
from moleculediffusiontransformer_amd import NoiseSource, runtime as rt
from moleculediffusiontransformer_amd.synth import synth_normal

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad_total = 0
for stride in (8, 1):
    os.environ["MDT_PAIR_STRIDE"] = str(stride)
    m = make_model("cfg1")
    m.kernel_choice = "narrow"
    seq = synth_normal("stress/seq", (1024, 12))
    first, bad = None, 0
    for k in range(calls):
        out = m.sample(seq, DEV, cond_scale=1.0, timesteps=64, noise=NoiseSource(seed=99, sample0=0))
        if first is None:
            first = out
            ops = m._engine.c.programs["eval"]
            assert all(op.i[rt.F_NSPLIT] == 2 and op.i[rt.F_PAIR_STRIDE] == stride for op in ops if op.kind in (rt.OP_TF256, rt.OP_RES256))
        elif not torch.equal(out, first):
            bad += 1
    st = m._engine.handoff_status()
    print(f"pair stride {stride}: {bad} of {calls - 1} repeats differ from the first call, finite {bool(torch.isfinite(first).all())}, status word {st}", flush=True)
    bad_total += bad + (st != 0)
print("TOTAL_BAD", bad_total)
