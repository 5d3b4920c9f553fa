#!/usr/bin/env python3
"""Micro-benchmark of single GEMM ops through the C ABI (GPU box): back-to-back launches, HIP-event timed.
Shapes are the heavy hitters of the cfg-1 U-Net at B=1024.  MDT_TILE="<cfg>,<stages>" forces a kernel config."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from moleculediffusiontransformer_amd import runtime as rt  # noqa: E402

SHAPES_ALL = [  # (rows/sample R, B, cin, taps, N, pro)
    (4, 1024, 512, 1, 256, 0), (4, 1024, 256, 1, 512, 1), (4, 1024, 256, 1, 1024, 1), (16, 1024, 128, 1, 1024, 1),
    (16, 1024, 128, 1, 512, 1), (16, 1024, 512, 1, 128, 0), (4, 1024, 256, 3, 256, 2), (16, 1024, 128, 3, 128, 2),
    (16, 1024, 256, 1, 128, 0), (4, 1024, 256, 1, 512, 0),
    (4, 1024, 256, 3, 256, 0), (16, 1024, 128, 3, 128, 0), (4, 1024, 512, 3, 256, 0), (16, 1024, 256, 3, 128, 0),   # 10-13: resnet convs
]
SHAPES = [SHAPES_ALL[int(i)] for i in os.environ["SHAPES"].split(",")] if os.environ.get("SHAPES") else SHAPES_ALL


def main():
    dev = "cuda:0"
    reps = 50
    for (R, B, cin, taps, N, pro) in SHAPES:
        K = taps * cin
        M = B * R
        w = torch.randn(N, K) * K ** -0.5
        hi = w.to(torch.bfloat16)
        lo = (w - hi.float()).to(torch.bfloat16)
        G = 8
        weights = torch.cat([hi.view(-1).view(torch.float32), lo.view(-1).view(torch.float32), torch.randn(N),
                             torch.ones(cin), torch.zeros(cin), torch.zeros(2 * cin)]).to(dev)
        o_lo = N * K // 2
        o_b = 2 * o_lo
        act = torch.randn(B * (R * cin + 64 + R * N), device=dev)
        act[B * R * cin: B * (R * cin + 64)] = 1.0
        op = rt.MdtOp()
        op.kind = rt.OP_GEMM
        op.a, op.w, op.a2 = rt.MdtRef(rt.SP_ACT, 0, 0), rt.MdtRef(rt.SP_WEIGHT, 0, 0), rt.MdtRef(rt.SP_WEIGHT, 0, o_lo)
        op.bias, op.out = rt.MdtRef(rt.SP_WEIGHT, 0, o_b), rt.MdtRef(rt.SP_ACT, 0, R * cin + 64)
        op.p0, op.p1 = rt.MdtRef(rt.SP_WEIGHT, 0, o_b + N), rt.MdtRef(rt.SP_WEIGHT, 0, o_b + N + cin)
        op.p2, op.p3 = rt.MdtRef(rt.SP_ACT, 0, R * cin), rt.MdtRef(rt.SP_WEIGHT, 0, o_b + N + 2 * cin)
        i = op.i
        i[rt.G_R_OUT], i[rt.G_R_IN], i[rt.G_LDA], i[rt.G_CIN], i[rt.G_TAPS] = R, R, cin, cin, taps
        i[rt.G_T_STRIDE], i[rt.G_T_DJ], i[rt.G_T_OFF] = 1, (1 if taps > 1 else 0), -(taps // 2)
        i[rt.G_N], i[rt.G_LDC], i[rt.G_O_ROWS], i[rt.G_O_STRIDE] = N, N, R, 1
        i[rt.G_PRO], i[rt.G_GROUPS], i[rt.G_GSIZE], i[rt.G_PRO_SILU] = pro, G, cin // G, 1
        op.f[0] = 1e-5
        prog = rt.Program([op])
        b = rt.MdtBindings()
        b.weights, b.act = rt.ptr(weights), rt.ptr(act)
        with torch.cuda.device(dev):
            for _ in range(5):
                prog.run(b, B)
            torch.cuda.synchronize()
            t = rt.EventTimer(1)
            t.start()
            for _ in range(reps):
                prog.run(b, B)
            t.stop()
            ms = t.collect()[0] / reps
        fl = 2.0 * M * N * K
        print(f"M={M:6d} N={N:5d} K={K:5d} pro={pro}: {ms * 1e3:8.1f} us  {fl / (ms * 1e-3) / 1e12:7.1f} TF", flush=True)


if __name__ == "__main__":
    main()
