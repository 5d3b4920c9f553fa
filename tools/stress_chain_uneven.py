#!/usr/bin/env python3
"""Uneven-load stress of the pair-split chain's per-wave hand-offs (GPU box; MI355X_MICROARCH.md: "test every hand-off under UNEVEN load,
checking every word").  The up-path chain of configs[1] (4 two-source blocks, 7 hand-offs per wave) at B = 1024 as a single op, launched
repeatedly while a second stream holds a varying number of compute units with mdt_test_occupy (the partners of many pairs then run at
different times: long waits, blocks re-used across launches with stale L2 / L1 lines around); every launch must reproduce the bits of
an undisturbed launch, the status word must stay 0.   MDT_TEST_HOOKS=1 python tools/stress_chain_uneven.py [launches=200]"""
import os
import sys
os.environ.setdefault("MDT_TEST_HOOKS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import ref
from helpers import synth_sd
from moleculediffusiontransformer_amd import runtime as rt
from moleculediffusiontransformer_amd.compiler import Ten, UNetCompiler
from moleculediffusiontransformer_amd.netspec import inverse_unet_config

n_launch = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = "cuda:0"
A = rt.SP_ACT
sd = {k[len("unet."):]: v for k, v in synth_sd("cfg1").items() if k.startswith("unet.")}
T, C, B = 4, 256, 1024
lib = rt.load_library()
bad_total = 0
for stride in (8, 1):
    blocks = [f"upsamples.0.blocks.{j}." for j in range(4)]
    comp = UNetCompiler(inverse_unet_config(16, 64, 128, 12), 64, 12, sd)
    comp.pair_stride = stride
    n = len(blocks)
    x, y = Ten(A, 0, T, C), Ten(A, T * C, T, C)
    skips = [Ten(A, 2 * T * C + (n - 1 - k) * T * C, T, C) for k in range(n)]
    comp.resnet_chain256(x, blocks, 2, skips, 2 ** -0.5, y, False, nsplit=2)
    op = comp.ops[0]
    op.p3 = ref(rt.SP_SHR, 0)
    nrb = (B * T + 31) // 32
    xflags = torch.zeros(64 + 64 * nrb, dtype=torch.int32, device=dev)
    xbuf = torch.empty(2 * nrb * 2 * 32 * 256, device=dev)
    W = comp.W.pack().to(dev)
    g = torch.Generator().manual_seed(5)
    act0 = torch.randn(B * (2 + n) * T * C, generator=g).to(dev)
    shr = (torch.randn(2 * C * n, generator=g) * 0.1).to(dev)
    prog = rt.Program([op])
    side = torch.cuda.Stream(device=dev)

    def launch(act):
        b = rt.MdtBindings()
        b.weights, b.act, b.shr = rt.ptr(W), rt.ptr(act), rt.ptr(shr)
        b.ext[3], b.ext[4] = rt.ptr(xflags), rt.ptr(xbuf)
        prog.run(b, B)

    with torch.cuda.device(dev):
        ref_act = act0.clone()
        launch(ref_act)
        torch.cuda.synchronize()
        want = ref_act[B * T * C: 2 * B * T * C].clone()
        bad = 0
        for k in range(n_launch):
            act = act0.clone()
            hold = (37 * k) % 200 + 8                   # 8 .. 207 compute units held for ~0.3 ms while the chain runs on the rest
            rt.check(lib.mdt_test_occupy(hold, 160 * 1024, 30000, side.cuda_stream))
            launch(act)
            torch.cuda.synchronize()
            if not torch.equal(act[B * T * C: 2 * B * T * C], want):
                bad += 1
        st = int(xflags[0])
    print(f"pair stride {stride}: {bad} of {n_launch} launches under uneven load differ from the undisturbed one, status word {st}", flush=True)
    bad_total += bad + (st != 0)
print("TOTAL_BAD", bad_total)
