#!/usr/bin/env python3
"""Micro-benchmark of the fused ResNet block kernel (GPU box): time against the batch (512 samples = one pass of the
persistent loop on 256 CUs), i.e. start-up cost and cost per pass."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import ref, rnd
from moleculediffusiontransformer_amd import runtime as rt
from moleculediffusiontransformer_amd.compiler import Ten, UNetCompiler
from moleculediffusiontransformer_amd.netspec import inverse_unet_config

A, dev = rt.SP_ACT, "cuda:0"
for cin, cout in ((16, 64), (64, 16)):
    p = "blk."
    sd = {p + "block1.groupnorm.weight": torch.ones(cin), p + "block1.groupnorm.bias": torch.zeros(cin),
          p + "block1.project.weight": rnd(cout, cin, 3, scale=(3 * cin) ** -0.5), p + "block1.project.bias": torch.zeros(cout),
          p + "block2.groupnorm.weight": torch.ones(cout), p + "block2.groupnorm.bias": torch.zeros(cout),
          p + "block2.project.weight": rnd(cout, cout, 3, scale=(3 * cout) ** -0.5), p + "block2.project.bias": torch.zeros(cout),
          p + "to_out.weight": rnd(cout, cin, 1, scale=cin ** -0.5), p + "to_out.bias": torch.zeros(cout)}
    comp = UNetCompiler(inverse_unet_config(16, 64, 128, 12), 64, 12, sd)
    comp.resnet(Ten(A, 0, 64, cin), p, cin, cout, 1, free_input=False)
    op = comp.ops[0]
    op.out = ref(A, 64 * cin)
    op.p3 = ref(rt.SP_SHR, 0)
    W = comp.W.pack().to(dev)
    shr = torch.zeros(2 * cout, device=dev)
    for B in (2, 512, 1024, 2048, 4096):
        act = torch.randn(B * 64 * (cin + cout), device=dev)
        prog = rt.Program([op])
        b = rt.MdtBindings(); b.weights, b.act, b.shr = rt.ptr(W), rt.ptr(act), rt.ptr(shr)
        with torch.cuda.device(dev):
            for _ in range(3): prog.run(b, B)
            torch.cuda.synchronize()
            t = rt.EventTimer(1); t.start()
            for _ in range(20): prog.run(b, B)
            t.stop(); ms = t.collect()[0] / 20
        print(f"cin={cin} cout={cout} B={B:5d}: {ms * 1e3:7.1f} us   {B * 64 * 4 * (cin + cout) / (ms * 1e-3) / 1e9:7.0f} GB/s", flush=True)
