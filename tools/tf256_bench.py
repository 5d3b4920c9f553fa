#!/usr/bin/env python3
"""One Transformer1d of the 256-channel level as MDT_OP_TF256 (GPU box): whole-workgroup form against the pair-split form, HIP-event
time per launch; with MDT_BUILD_DEFS=-DMDT_STAMPS also the in-kernel timeline of the two workgroups of row block 0.

    python tools/tf256_bench.py [B=1024] [layers=4] [cross=1]
"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from gpu_util import ref, rnd  # noqa: E402
from test_gpu_ops import _transformer_sd  # noqa: E402
from moleculediffusiontransformer_amd import runtime as rt  # noqa: E402
from moleculediffusiontransformer_amd.compiler import Ten, UNetCompiler  # noqa: E402
from moleculediffusiontransformer_amd.netspec import inverse_unet_config  # noqa: E402

A = rt.SP_ACT
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cross = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
T, C, n_ctx, mid = 4, 256, 12, 512
dev = "cuda:0"
cfg = inverse_unet_config(16, 64, 128, n_ctx)
sd = _transformer_sd("tf.", C, layers, cross)
stamps_on = "MDT_STAMPS" in os.environ.get("MDT_BUILD_DEFS", "")
lib = rt.load_library()
for form, stride in (("whole", 0), ("pair", 8), ("pair", 1)):
    comp = UNetCompiler(cfg, 64, n_ctx, sd, tf256=(form == "whole"))
    comp.transformer(Ten(A, 0, T, C), "tf.", C, layers, cross, free_input=False)
    op = comp.ops[0]
    op.out = ref(A, T * C)
    if cross:
        op.a2 = ref(A, 2 * T * C)
    nrb = (B * T + 31) // 32
    W = comp.W.pack().to(dev)
    act = torch.randn(B * (2 * T * C + layers * n_ctx * 2 * mid), device=dev) * 0.3
    flags = torch.zeros(64 + 64 * nrb + 4096, dtype=torch.int32, device=dev)
    xbuf = torch.zeros(2 * nrb * 2 * 32 * 256, device=dev)
    b = rt.MdtBindings(); b.weights, b.act = rt.ptr(W), rt.ptr(act); b.ext[3], b.ext[4] = rt.ptr(flags), rt.ptr(xbuf)
    prog = rt.Program([op])
    lib.mdt_set_tuning(b"pair_stride", stride)
    with torch.cuda.device(dev):
        for _ in range(3):
            prog.run(b, B)
        torch.cuda.synchronize()
        t = rt.EventTimer(1); t.start()
        for _ in range(20):
            prog.run(b, B)
        t.stop(); ms = t.collect()[0] / 20
    nsub = 1 + layers * (3 if cross else 2)
    print(f"TF256 {form:5s} stride {stride}: B={B} layers={layers} cross={int(cross)}: {ms * 1e3:8.1f} us per launch, {ms * 1e3 / nsub:6.2f} us per sub-block "
          f"({nsub} sub-blocks), weights {W.numel() * 4 / 1e6:.1f} MB, status {int(flags[0])}", flush=True)
    if stamps_on and form == "pair":
        for hh in range(2):
            raw = flags.cpu()[64 + 64 * nrb + 512 * hh: 64 + 64 * nrb + 512 * hh + 500].view(torch.int64).tolist()
            st = [(v >> 48) & 0xffff for v in raw if v], [v & 0xffffffffffff for v in raw if v]
            d = [(st[0][k + 1], st[1][k + 1] - st[1][k]) for k in range(len(st[0]) - 1)]
            print(f"  half {hh}: total {st[1][-1] - st[1][0]} cycles; (source line of the stamp : cycles since the previous stamp)")
            print("   " + " ".join(f"{ln}:{c}" for ln, c in d))
lib.mdt_set_tuning(b"pair_stride", 0)
