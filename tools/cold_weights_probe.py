#!/usr/bin/env python3
"""How much of a fused sub-block launch is the COLD start of its weight stream?  (GPU box: python tools/cold_weights_probe.py)

One C = 256 self-attention sub-block (k_tblock32, head split, B = 1024: the most frequent launch of an evaluation) is timed
  warm:  the same launch 200 times back to back (its 2 MB of weights stay in every XCD's L2)
  cold:  round robin over NCOPY copies of the weights (each copy is out of the L2s, NCOPY * 2 MB against 8 x 4 MB of L2 and
         256 MB of Infinity Cache) -- what an evaluation does: 86 launches with different weights between two uses.
"""
import os
import sys
import ctypes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from gpu_util import rnd  # noqa: E402
from moleculediffusiontransformer_amd import runtime as rt  # noqa: E402
from moleculediffusiontransformer_amd.compiler import Ten, UNetCompiler  # noqa: E402
from moleculediffusiontransformer_amd.netspec import inverse_unet_config  # noqa: E402

A = rt.SP_ACT
dev = "cuda:0"
cfg = inverse_unet_config(16, 64, 128, 12)
C, T, B, mid, n_ctx, p = 256, 4, 1024, 512, 12, "blk."
sd = {p + "norm.weight": torch.ones(C), p + "norm.bias": torch.zeros(C), p + "norm_context.weight": torch.ones(C),
      p + "norm_context.bias": torch.zeros(C), p + "to_q.weight": rnd(mid, C, scale=C ** -0.5),
      p + "to_kv.weight": rnd(2 * mid, C, scale=C ** -0.5), p + "attention.to_out.weight": rnd(C, mid, scale=mid ** -0.5),
      p + "attention.to_out.bias": torch.zeros(C)}
for mode, name in ((rt.TB_SELF, "self-attention"),):
    comp = UNetCompiler(cfg, 64, n_ctx, sd)
    comp.tblock(Ten(A, 0, T, C), mode, p, None, variant=3)
    op0 = comp.ops[0]
    op0.out = rt.MdtRef(A, 0, T * C)
    W1 = comp.W.pack()
    n1 = W1.numel()
    for ncopy in (1, 8, 64):
        W = W1.repeat(ncopy).to(dev)
        act = torch.randn(B * (T * C + 2 * T * C), device=dev)
        ops = []
        for k in range(ncopy):
            o = rt.MdtOp()
            ctypes.memmove(ctypes.byref(o), ctypes.byref(op0), ctypes.sizeof(rt.MdtOp))
            o.w = rt.MdtRef(op0.w.space, 0, op0.w.off + k * n1)
            o.bias = rt.MdtRef(op0.bias.space, 0, op0.bias.off + k * n1)
            ops.append(o)
        reps = max(1, 192 // ncopy)
        prog = rt.Program(ops * reps)
        b = rt.MdtBindings()
        b.weights, b.act = rt.ptr(W), rt.ptr(act)
        with torch.cuda.device(dev):
            prog.run(b, B)
            torch.cuda.synchronize()
            t = rt.EventTimer(1)
            t.start()
            prog.run(b, B)
            t.stop()
            us = t.collect()[0] * 1e3 / (ncopy * reps)
        print(f"{name}: {ncopy:3d} weight copies ({ncopy * n1 * 4 / 1e6:6.1f} MB): {us:6.2f} us per launch", flush=True)
