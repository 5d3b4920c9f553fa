#!/usr/bin/env python3
"""Per-launch HIP-event times of one pair-split Transformer1d launch (MDT_OP_TF256, B = 1024, 4 blocks with cross-attention), N launches
in a row in ONE process: looks for sporadic slow launches (a partner workgroup scheduled late would show as a multi-millisecond launch).

    python tools/pair_launch_outliers.py [launches=3000] [pair_stride=8]
"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from gpu_util import ref  # noqa: E402
from test_gpu_ops import _transformer_sd  # noqa: E402
from moleculediffusiontransformer_amd import runtime as rt  # noqa: E402
from moleculediffusiontransformer_amd.compiler import Ten, UNetCompiler  # noqa: E402
from moleculediffusiontransformer_amd.netspec import inverse_unet_config  # noqa: E402

A = rt.SP_ACT
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
stride = int(sys.argv[2]) if len(sys.argv) > 2 else 8
NW = int(sys.argv[4]) if len(sys.argv) > 4 else 1          # copies of the packed weights (17 MB each) the launches rotate over
NCOPY = int(sys.argv[3]) if len(sys.argv) > 3 else 1      # activation arenas (x, y, hoisted K / V: 213 MB each) the launches rotate over:
                                                           # 1 = K / V stay in the 256 MB Infinity Cache, 8 = they come from HBM as in an evaluation
B, layers, cross, T, C, n_ctx, mid = 1024, 4, True, 4, 256, 12, 512
dev = "cuda:0"
cfg = inverse_unet_config(16, 64, 128, n_ctx)
sd = _transformer_sd("tf.", C, layers, cross)
lib = rt.load_library()
comp = UNetCompiler(cfg, 64, n_ctx, sd, tf256=False)
comp.transformer(Ten(A, 0, T, C), "tf.", C, layers, cross, free_input=False)
op = comp.ops[0]
op.out = ref(A, T * C)
op.a2 = ref(A, 2 * T * C)
nrb = (B * T + 31) // 32
Ws = [comp.W.pack().to(dev) for _ in range(NW)]
W = Ws[0]
acts = [torch.randn(B * (2 * T * C + layers * n_ctx * 2 * mid), device=dev) * 0.3 for _ in range(NCOPY)]
act = acts[0]
flags = torch.zeros(64 + 64 * nrb, dtype=torch.int32, device=dev)
xbuf = torch.zeros(2 * nrb * 2 * 32 * 256, device=dev)
def binding(a_, w_):
    b_ = rt.MdtBindings(); b_.weights, b_.act = rt.ptr(w_), rt.ptr(a_); b_.ext[3], b_.ext[4] = rt.ptr(flags), rt.ptr(xbuf)
    return b_
bs = [binding(acts[i % NCOPY], Ws[i % NW]) for i in range(max(NCOPY, NW) if NCOPY * NW > 1 else 1)]
b = bs[0]
prog = rt.Program([op])
lib.mdt_set_tuning(b"pair_stride", stride)
with torch.cuda.device(dev):
    for _ in range(5):
        prog.run(b, B)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    ev[0].record()
    for i in range(N):
        prog.run(bs[i % len(bs)], B)
        ev[i + 1].record()
    torch.cuda.synchronize()
ms = sorted((ev[i].elapsed_time(ev[i + 1]), i) for i in range(N))
med = ms[N // 2][0]
slow = [(round(t * 1e3), i) for t, i in ms if t > 2 * med]
print(f"pair stride {stride}, {NCOPY} arena(s), {NW} weight copies: {N} launches, median {med * 1e3:.1f} us, min {ms[0][0] * 1e3:.1f}, p99 {ms[int(N * 0.99)][0] * 1e3:.1f}, "
      f"max {ms[-1][0] * 1e3:.1f}; launches over 2 x median: {len(slow)} {slow[:12]}; status {int(flags[0])}")
lib.mdt_set_tuning(b"pair_stride", 0)
