#!/usr/bin/env python3
"""Micro-benchmark of the chained ResNet blocks of the 256-channel level (MDT_OP_RES256, csrc/k_res256.hip) on the GPU box:
the four chains of BASELINE configs[1] (down path: 3 blocks, bottleneck: 1 + 1, up path: 4 two-source blocks) as single ops.
MDT_DBG=8 with a library built with MDT_BUILD_DEFS=-DMDT_STAMPS prints the in-kernel clock stamps of wave 0 / workgroup 0
(source line: cycles since the previous stamp) for block 0 of each chain."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import ref
from helpers import synth_sd
from moleculediffusiontransformer_amd import runtime as rt
from moleculediffusiontransformer_amd.compiler import Ten, UNetCompiler
from moleculediffusiontransformer_amd.netspec import inverse_unet_config

A = rt.SP_ACT
dev = "cuda:0"
sd = {k[len("unet."):]: v for k, v in synth_sd("cfg1").items() if k.startswith("unet.")}
T, C = 4, 256
chains = [("down", 1, [f"downsamples.1.blocks.{j}." for j in range(3)]), ("bottleneck", 1, ["bottleneck.pre_block."]),
          ("up", 2, [f"upsamples.0.blocks.{j}." for j in range(4)])]
for B in (8, 1024, int(os.environ.get("MDT_BIG", "4096"))):
    for name, kind, blocks in chains:
        comp = UNetCompiler(inverse_unet_config(16, 64, 128, 12), 64, 12, sd)
        n = len(blocks)
        x, y = Ten(A, 0, T, C), Ten(A, T * C, T, C)
        base = 2 * T * C
        if kind == 1:
            skips = [Ten(A, base + k * T * C, T, C) for k in range(n)]
        else:
            skips = [Ten(A, base + (n - 1 - k) * T * C, T, C) for k in range(n)]
        nsplit = int(os.environ.get("MDT_NSPLIT", "1"))              # 2: the pair-split chain (round 6)
        comp.resnet_chain256(x, blocks, kind, skips, 2 ** -0.5, y, False, nsplit=nsplit)
        op = comp.ops[0]
        op.p3 = ref(rt.SP_SHR, 0)
        dbg = torch.zeros(1024, device=dev)
        nrb = (B * T + 31) // 32
        xflags = torch.zeros(64 + 64 * nrb, dtype=torch.int32, device=dev)
        xbuf = torch.empty(2 * nrb * 2 * 32 * 256, device=dev)
        if os.environ.get("MDT_DBG", "0") == "8":
            op.p2 = ref(rt.SP_EXT0, 0)
        W = comp.W.pack()
        W = W.to(dev)
        act = torch.randn(B * (2 + n) * T * C, device=dev)
        shr = torch.randn(2 * C * n, device=dev) * 0.1
        prog = rt.Program([op])
        b = rt.MdtBindings(); b.weights, b.act, b.shr = rt.ptr(W), rt.ptr(act), rt.ptr(shr); b.ext[0] = rt.ptr(dbg); b.ext[3], b.ext[4] = rt.ptr(xflags), rt.ptr(xbuf)
        with torch.cuda.device(dev):
            for _ in range(3): prog.run(b, B)
            torch.cuda.synchronize()
            t = rt.EventTimer(1); t.start()
            for _ in range(20): prog.run(b, B)
            t.stop(); ms = t.collect()[0] / 20
        nt = op.i[rt.F_NT]
        print(f"{name:10s} kind {kind} blocks {n} B={B:5d} nsplit {nsplit}: {ms * 1e3:7.1f} us  ({nt} tiles)  status {int(xflags[0])}", flush=True)
        if os.environ.get("MDT_DBG", "0") == "8":
            st = dbg.cpu().view(torch.int64)[:120].tolist()
            st = [v for v in st if v]
            mask = (1 << 48) - 1
            print("   stamps (line: cycles):", " ".join(f"{st[k + 1] >> 48}:{(st[k + 1] & mask) - (st[k] & mask)}" for k in range(len(st) - 1)))
            ld = dbg.cpu().view(torch.int64)[128:128 + 96].view(24, 4).tolist()
            if ld[0][0]:
                print("   loader wave 4, tiles 24..47 (wait for landing | barrier | issue | turn start to next turn start):",
                      " ".join(f"{r[1] - r[0]}|{r[2] - r[1]}|{r[3] - r[2]}|{(ld[j + 1][0] - r[0]) if j + 1 < 24 else 0}" for j, r in enumerate(ld)))
