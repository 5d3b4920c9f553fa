#!/usr/bin/env python3
"""Micro-benchmark of the row-stationary convolution (GPU box).  MDT_DBG=8 + MDT_BUILD_DEFS=-DMDT_STAMPS prints
the in-kernel clock stamps of wave 0 / workgroup 0."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import ref, rnd
from moleculediffusiontransformer_amd import runtime as rt
from moleculediffusiontransformer_amd.compiler import Ten, UNetCompiler
from moleculediffusiontransformer_amd.netspec import inverse_unet_config

A = rt.SP_ACT
dev = "cuda:0"
for (C, T, taps, gsize) in [(256, 4, 3, 32), (256, 4, 1, 0), (256, 4, 1, 8), (128, 16, 3, 16), (128, 16, 1, 0)]:
    for B in (32 // T if C == 256 else 64 // T, 1024):
        comp = UNetCompiler(inverse_unet_config(16, 64, 128, 12), 64, 12, {})
        g_off = comp.W.add("gb", torch.cat([torch.ones(C), torch.zeros(2 * C)]))
        x, out = Ten(A, 0, T, C), Ten(A, T * C, T, C)
        comp.rconv(x, rnd(C, C, taps, scale=(C * taps) ** -0.5), "w", out, taps=taps, bias_off=g_off + 2 * C,
                   gn=(g_off, g_off + C, gsize, 1e-5, True) if gsize else None)
        op = comp.ops[0]
        dbg = torch.zeros(256, device=dev)
        if os.environ.get("MDT_DBG", "0") == "8":
            op.p2 = ref(rt.SP_EXT0, 0)
        W = comp.W.pack().to(dev)
        act = torch.randn(B * 2 * T * C, device=dev)
        prog = rt.Program([op])
        b = rt.MdtBindings(); b.weights, b.act = rt.ptr(W), rt.ptr(act); b.ext[0] = rt.ptr(dbg)
        with torch.cuda.device(dev):
            for _ in range(3): prog.run(b, B)
            torch.cuda.synchronize()
            t = rt.EventTimer(1); t.start()
            for _ in range(20): prog.run(b, B)
            t.stop(); ms = t.collect()[0] / 20
        if os.environ.get("MDT_DBG", "0") == "8":
            st = dbg.cpu().view(torch.int64)[:60].tolist()
            print("   stamp deltas:", [st[k + 1] - st[k] for k in range(59) if st[k + 1] > 0])
        fl = 2.0 * B * T * C * C * taps
        print(f"C={C} T={T} taps={taps} gsize={gsize:2d} B={B:5d}: {ms * 1e3:7.1f} us  {fl / (ms * 1e-3) / 1e12:6.1f} TF", flush=True)
