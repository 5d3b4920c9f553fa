#!/usr/bin/env python3
"""Micro-benchmark of the plain-bf16 GEMM forms on the deep-UNet (configs[4]) layer shapes through the C ABI (GPU box):
  reg   = MDT_G_WFMT 1: fp32 A, prologue in the kernel, register staging (k_gemm3<NPROD = 1>)
  dma   = MDT_OP_PREP16 + MDT_G_WFMT 2: bf16 A written once, both operands by LDS-DMA (k_gemm_b16); prep and GEMM timed apart
MDT_TILE16 = 0 / 1 / 2 / 3 forces the 256x256 / 256x128 / 128x128 / 256x256-on-four-waves tile.  BATCH=<B> (default 512)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from moleculediffusiontransformer_amd import runtime as rt  # noqa: E402

B = int(os.environ.get("BATCH", "512"))
SHAPES = [  # (rows / sample, cin, taps, N, prologue)
    (32, 512, 3, 512, 2), (32, 1024, 3, 512, 2), (32, 512, 1, 512, 1), (32, 512, 1, 1024, 1), (32, 1024, 1, 512, 0),
    (8, 1024, 3, 1024, 2), (8, 2048, 3, 1024, 2), (8, 1024, 1, 1024, 1), (8, 1024, 1, 2048, 1), (8, 2048, 1, 1024, 0),
    (8, 512, 1, 1024, 1), (8, 1024, 1, 512, 0), (128, 256, 3, 256, 2),
]


def time_prog(prog, b, reps=20):
    for _ in range(3):
        prog.run(b, B)
    torch.cuda.synchronize()
    t = rt.EventTimer(1)
    t.start()
    for _ in range(reps):
        prog.run(b, B)
    t.stop()
    return t.collect()[0] / reps


def main():
    dev = "cuda:0"
    for (R, cin, taps, N, pro) in SHAPES:
        K, M, G = taps * cin, B * R, 8
        w = (torch.randn(N, K) * K ** -0.5).to(torch.bfloat16)
        weights = torch.cat([w.view(-1).view(torch.float32), torch.randn(N), torch.ones(cin), torch.zeros(cin),
                             torch.zeros(2 * cin)]).to(dev)
        o_b = N * K // 2
        x_off, st_off, a16_off = 0, R * cin, R * cin + 64
        out_off = a16_off + R * cin // 2
        act = torch.randn(B * (out_off + R * N), device=dev)
        act[B * st_off: B * a16_off] = 1.0

        def ref(space, off):
            return rt.MdtRef(space, 0, off)

        def gemm(wfmt, a_off, prologue):
            op = rt.MdtOp()
            op.kind = rt.OP_GEMM
            op.a, op.w, op.bias, op.out = ref(rt.SP_ACT, a_off), ref(rt.SP_WEIGHT, 0), ref(rt.SP_WEIGHT, o_b), ref(rt.SP_ACT, out_off)
            op.p0, op.p1 = ref(rt.SP_WEIGHT, o_b + N), ref(rt.SP_WEIGHT, o_b + N + cin)
            op.p2, op.p3 = ref(rt.SP_ACT, st_off), ref(rt.SP_WEIGHT, o_b + N + 2 * cin)
            i = op.i
            i[rt.G_R_OUT], i[rt.G_R_IN], i[rt.G_LDA], i[rt.G_CIN], i[rt.G_TAPS] = R, R, cin, cin, taps
            i[rt.G_T_STRIDE], i[rt.G_T_DJ], i[rt.G_T_OFF] = 1, (1 if taps > 1 else 0), -(taps // 2)
            i[rt.G_N], i[rt.G_LDC], i[rt.G_O_ROWS], i[rt.G_O_STRIDE] = N, N, R, 1
            i[rt.G_PRO], i[rt.G_GROUPS], i[rt.G_GSIZE], i[rt.G_PRO_SILU] = prologue, G, cin // G, 1
            i[rt.G_WFMT] = wfmt
            op.f[0] = 1e-5
            return op

        pre = rt.MdtOp()
        pre.kind = rt.OP_PREP16
        pre.a, pre.out = ref(rt.SP_ACT, x_off), ref(rt.SP_ACT, a16_off)
        pre.p0, pre.p1 = ref(rt.SP_WEIGHT, o_b + N), ref(rt.SP_WEIGHT, o_b + N + cin)
        pre.p2, pre.p3 = ref(rt.SP_ACT, st_off), ref(rt.SP_WEIGHT, o_b + N + 2 * cin)
        pi = pre.i
        pi[rt.G_R_IN], pi[rt.G_LDA], pi[rt.G_CIN], pi[rt.G_PRO], pi[rt.G_GROUPS], pi[rt.G_GSIZE], pi[rt.G_PRO_SILU] = \
            R, cin, cin, pro, G, cin // G, 1
        pre.f[0] = 1e-5
        b = rt.MdtBindings()
        b.weights, b.act = rt.ptr(weights), rt.ptr(act)
        fl = 2.0 * M * N * K
        with torch.cuda.device(dev):
            t_reg = time_prog(rt.Program([gemm(1, x_off, pro)]), b)
            t_pre = time_prog(rt.Program([pre]), b)
            t_dma = time_prog(rt.Program([gemm(2, a16_off, 0)]), b)
        tf = lambda ms: fl / (ms * 1e-3) / 1e12   # noqa: E731
        print(f"M={M:6d} N={N:5d} K={K:5d} pro={pro}: reg {t_reg * 1e3:7.1f} us {tf(t_reg):6.0f} TF | prep {t_pre * 1e3:6.1f} us "
              f"({M * cin * 6 / (t_pre * 1e-3) / 1e12:4.1f} TB/s) + dma {t_dma * 1e3:7.1f} us {tf(t_dma):6.0f} TF "
              f"= {tf(t_pre + t_dma):6.0f} TF", flush=True)


if __name__ == "__main__":
    main()
