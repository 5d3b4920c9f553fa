#!/usr/bin/env python3
"""Per-op HIP-event timing of one U-Net evaluation (GPU box): python tools/op_profile.py <case> <B> [gemm_mode].
Writes gpurun_out/op_profile_<case>_b<B>.txt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from gpu_util import make_model  # noqa: E402
from moleculediffusiontransformer_amd import runtime as rt  # noqa: E402
from moleculediffusiontransformer_amd.synth import synth_normal  # noqa: E402


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    m = make_model(case)
    if len(sys.argv) > 3:
        m.gemm_mode = sys.argv[3]            # bf16x3 (default) | f32 | bf16
    n_ctx = m.unet.config.ctx_max_length
    eng = m.engine("cuda:0", n_ctx, B)
    eng.reserve(B)
    emb = m._embed(synth_normal("prof/seq", (B, n_ctx)), "cuda:0")
    eng.prepare_context(emb)
    eng.prepare_times(torch.tensor([0.1]))
    eng.select_time(0)
    eng.xin.normal_()
    ops = eng.c.programs["eval"]
    prog = eng.programs["eval"]
    bind = eng._bind(xin=eng.xin, out=eng.pred)
    best = None
    for rep in range(5):
        t = rt.EventTimer(len(ops))
        for i in range(len(ops)):
            t.start()
            prog.run(bind, B, 0, i, 1)
            t.stop()
        ms = t.collect()
        best = ms if best is None else [min(a, b) for a, b in zip(best, ms)]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    lines = []
    tot = {}
    for idx, (op, t) in enumerate(zip(ops, best)):
        i = op.i
        if op.kind == rt.OP_GEMM:
            M, N, K = B * i[rt.G_R_OUT], i[rt.G_N], i[rt.G_TAPS] * i[rt.G_CIN]
            fl = 2.0 * M * N * K
            desc = f"GEMM  M={M:6d} N={N:5d} K={K:5d} taps={i[rt.G_TAPS]} pro={i[rt.G_PRO]} act={i[rt.G_ACT]} res={int(op.res.space != 0)}"
            key = f"gemm M={M} N={N} K={K} pro={i[rt.G_PRO]}"
        elif op.kind == rt.OP_ATTN:
            fl = 4.0 * B * i[rt.A_T] * i[rt.A_TK] * 512
            desc = f"ATTN  T={i[rt.A_T]} Tk={i[rt.A_TK]}"
            key = desc
        elif op.kind == rt.OP_GN_STATS:
            fl = 0
            desc = f"GNST  rows={i[rt.N_ROWS]} ld={i[rt.N_LD]} G={i[rt.N_GROUPS]}"
            key = desc
        elif op.kind == rt.OP_GN_ACT:
            fl = 0
            desc = f"GNACT rows={i[rt.N_ROWS]} ld={i[rt.N_LD]} G={i[rt.N_GROUPS]}"
            key = desc
        elif op.kind == rt.OP_RCONV:
            fl = 2.0 * B * i[rt.R_T] * i[rt.R_C] * i[rt.R_C] * i[rt.R_TAPS]
            desc = f"RCONV T={i[rt.R_T]} C={i[rt.R_C]} taps={i[rt.R_TAPS]} gsize={i[rt.R_GSIZE]} res={int(op.res.space != 0)}"
            key = desc
        elif op.kind == rt.OP_TBLOCK:
            fl = 0
            desc = f"TBLK  mode={i[rt.B_MODE]} C={i[rt.B_C]} T={i[rt.B_T]}"
            key = desc
        else:
            fl = 0
            desc = f"kind{op.kind}"
            key = desc
        lines.append(f"{idx:4d} {t * 1e3:9.1f} us  {fl / (t * 1e-3) / 1e12 if t > 0 else 0:7.2f} TF  {desc}")
        n, tt, ff = tot.get(key, (0, 0.0, 0.0))
        tot[key] = (n + 1, tt + t, ff + fl)
    lines.append("")
    lines.append(f"total {sum(best):.3f} ms for {len(ops)} ops at B={B}")
    for key, (n, tt, ff) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{tt:8.3f} ms  x{n:3d}  {ff / (tt * 1e-3) / 1e12 if tt > 0 else 0:7.2f} TF  {key}")
    out = os.path.join(ROOT, "gpurun_out", f"op_profile_{case}_b{B}.txt")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[-40:]))


if __name__ == "__main__":
    main()
