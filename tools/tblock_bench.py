#!/usr/bin/env python3
"""Micro-benchmark of the fused transformer sub-block kernel (GPU box)."""
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
cfg = inverse_unet_config(16, 64, 128, 12)
dev = "cuda:0"
for (C, T) in [(128, 16), (256, 4)]:
    for mode, name in ((rt.TB_SELF, "SELF"), (rt.TB_CROSS, "CROSS"), (rt.TB_FF, "FF")):
        variant = int(os.environ.get('VARIANT', '0'))
        if variant >= 2 and C != 256:
            continue
        for B in ([int(v) for v in os.environ['BS'].split(',')] if os.environ.get('BS') else (64 * 16 // T // 16, 1024)):
            n_ctx, mid, p = 12, 512, "blk."
            sd = {p + "norm.weight": torch.ones(C), p + "norm.bias": torch.zeros(C), p + "norm_context.weight": torch.ones(C),
                  p + "norm_context.bias": torch.zeros(C), p + "to_q.weight": rnd(mid, C, scale=C ** -0.5),
                  p + "to_kv.weight": rnd(2 * mid, C, scale=C ** -0.5), p + "attention.to_out.weight": rnd(C, mid, scale=mid ** -0.5),
                  p + "attention.to_out.bias": torch.zeros(C), p + "0.weight": rnd(4 * C, C, scale=C ** -0.5), p + "0.bias": torch.zeros(4 * C),
                  p + "2.weight": rnd(C, 4 * C, scale=(4 * C) ** -0.5), p + "2.bias": torch.zeros(C)}
            comp = UNetCompiler(cfg, 64, n_ctx, sd)
            comp.tblock(Ten(A, 0, T, C), mode, p, 0 if mode == rt.TB_CROSS else None, variant=variant)
            op = comp.ops[0]
            if mode == rt.TB_CROSS:
                op.a2 = ref(A, T * C)
            dbg = torch.zeros(2 * (256 + 16 * 1024) + 64, device=dev)
            if os.environ.get('MDT_DBG', '0') == '8':
                op.p0 = ref(rt.SP_EXT0, 0)
            W = comp.W.pack().to(dev)
            if variant == 3:
                op.out = ref(A, T * C + n_ctx * 2 * mid)
            act = torch.randn(B * (T * C + n_ctx * 2 * mid + 2 * T * C), device=dev) * 0.1
            prog = rt.Program([op])
            b = rt.MdtBindings(); b.weights, b.act = rt.ptr(W), rt.ptr(act); b.ext[0] = rt.ptr(dbg)
            with torch.cuda.device(dev):
                for _ in range(3): prog.run(b, B)
                torch.cuda.synchronize()
                t = rt.EventTimer(1); t.start()
                for _ in range(20): prog.run(b, B)
                t.stop(); ms = t.collect()[0] / 20
            if os.environ.get('MDT_DBG', '0') == '8':
                raw = dbg.cpu().view(torch.int64)[:120].tolist()
                st = [v & 0xffffffffffff for v in raw]
                tags = [(v >> 48) & 0xffff for v in raw]
                d = [(tags[k + 1], st[k + 1] - st[k]) for k in range(119) if st[k + 1] > 0]
                print("   stamps (source line of the stamp : cycles since the previous one):", " ".join(f"{t}:{c}" for t, c in d[:70]),
                      " total", st[max(k for k in range(120) if st[k] > 0)] - st[0])
                ls = dbg.cpu().view(torch.int64)[128:248].tolist()
                if ls[0] > 0:
                    print("   loader (wait-landed, barrier, issue+loop) x tiles:", [(ls[3*k+1]-ls[3*k], ls[3*k+2]-ls[3*k+1], ls[3*k+3]-ls[3*k+2]) for k in range(12) if ls[3*k+3] > 0])
            if os.environ.get('MDT_DBG', '0') == '8':
                nwg = min(1024, ((B * T + 31) // 32) * 2)
                rt_ = dbg.cpu().view(torch.int64)[256:256 + 8 * nwg].view(-1, 8)
                rt_ = rt_[rt_[:, 0] > 0]
                if len(rt_):
                    t0 = int(rt_[:, 0].min())
                    us = (rt_ - t0).double() / 100.0       # us since the first workgroup entered
                    names = ["entry", "rows arrived", "normalised", "head 1", "head 2", "head 3", "head 4", "exit"]
                    print(f"   {len(rt_)} workgroups, 100 MHz clock, us since the first entry (min / mean / max over workgroups), then per-workgroup phase lengths:")
                    for k in range(8):
                        if (rt_[:, k] > 0).all():
                            d = us[:, k] - (us[:, k - 1] if k else 0)
                            print(f"     {names[k]:13s} at {us[:, k].min():6.2f} / {us[:, k].mean():6.2f} / {us[:, k].max():6.2f}   phase {d.min():5.2f} / {d.mean():5.2f} / {d.max():5.2f}")
                    cy = dbg.cpu().view(torch.int64)[256 + 8 * 1024:256 + 8 * 1024 + 8 * nwg].view(-1, 8)[:len(rt_)]
                    for k0, k1, nm in ((0, 7, "whole kernel"), (3, 6, "heads 2-4")):
                        mhz = (cy[:, k1] - cy[:, k0]).double() / (rt_[:, k1] - rt_[:, k0]).double() * 100.0
                        print(f"     shader clock over {nm}: {mhz.min():.0f} / {mhz.mean():.0f} / {mhz.max():.0f} MHz; cycles {float((cy[:, k1] - cy[:, k0]).double().mean()):.0f}")
                    tot = us[:, 7] - us[:, 0]
                    slow = int(tot.argmax()); fast = int(tot.argmin())
                    print(f"     slowest workgroup #{slow}: phases", [round(float(us[slow, k] - (us[slow, k - 1] if k else 0)), 2) for k in range(8)],
                          f"fastest #{fast}:", [round(float(us[fast, k] - (us[fast, k - 1] if k else 0)), 2) for k in range(8)])
            print(f"C={C} T={T} {name:5s} B={B:5d} blocks={(B * T + 63) // 64:4d}: {ms * 1e3:7.1f} us  weights {W.numel() * 4 / 1e6:.2f} MB", flush=True)
