"""-m gpu: every kernel class through the C ABI against the CPU interpreter / closed-form torch expressions."""
import math

import pytest
import torch

from gpu_util import DEV, ref, rnd, run_both
from moleculediffusiontransformer_amd import runtime as rt

pytestmark = pytest.mark.gpu

W, A, S, E0 = rt.SP_WEIGHT, rt.SP_ACT, rt.SP_SHR, rt.SP_EXT0


@pytest.fixture(params=["bf16x3", "f32"])
def prod(request, monkeypatch):
    """Product type of the ring kernels under test: split-bf16 tiles (hi*hi + hi*lo + lo*hi on bf16 MFMAs) or fp32 fragment tiles
    with exact fp32 MFMA products (MDT_F_WF32 / MDT_R_WF32 / MDT_K_WF32).  Same ops, same interpreter, same closed forms: every
    UNetCompiler a test builds with the default mode is built in this one."""
    from moleculediffusiontransformer_amd import compiler
    orig = compiler.UNetCompiler.__init__

    def init(self, cfg, length, cond_len, sd, max_time_rows=1024, gemm_mode="bf16x3", fuse_blocks=True, tf256=False):
        orig(self, cfg, length, cond_len, sd, max_time_rows, request.param if gemm_mode == "bf16x3" else gemm_mode, fuse_blocks, tf256)
    monkeypatch.setattr(compiler.UNetCompiler, "__init__", init)
    return request.param


def gemm_op(**kw):
    op = rt.MdtOp()
    op.kind = rt.OP_GEMM
    for k in ("a", "w", "bias", "out", "res", "p0", "p1", "p2", "p3"):
        if k in kw:
            setattr(op, k, kw.pop(k))
    i = op.i
    i[rt.G_T_STRIDE], i[rt.G_O_STRIDE] = 1, 1
    names = dict(r_out=rt.G_R_OUT, r_in=rt.G_R_IN, lda=rt.G_LDA, cin=rt.G_CIN, taps=rt.G_TAPS, t_stride=rt.G_T_STRIDE,
                 t_dj=rt.G_T_DJ, t_off=rt.G_T_OFF, n=rt.G_N, ldc=rt.G_LDC, o_rows=rt.G_O_ROWS,
                 o_stride=rt.G_O_STRIDE, o_off=rt.G_O_OFF, ldr=rt.G_LDR, pro=rt.G_PRO, groups=rt.G_GROUPS,
                 gsize=rt.G_GSIZE, pro_silu=rt.G_PRO_SILU, act=rt.G_ACT, m_mode=rt.G_M_MODE, a_col=rt.G_A_COL,
                 o_col=rt.G_O_COL)
    op.f[0] = kw.pop("eps", 0.0)
    for k, v in kw.items():
        i[names[k]] = v
    return op


@pytest.mark.parametrize("B,R,cin,N,taps", [(3, 16, 128, 512, 1), (5, 64, 16, 64, 3), (2, 4, 512, 256, 3),
                                            (7, 16, 48, 32, 1), (1, 1, 80, 256, 1), (9, 64, 64, 16, 3)])
def test_gemm_plain_bias_gelu_residual(B, R, cin, N, taps):
    K = taps * cin
    weights = torch.cat([rnd(N, K, seed=1, scale=K ** -0.5).view(-1), rnd(N, seed=2)])
    act = torch.cat([rnd(B * R * cin, seed=3), torch.zeros(B * R * N), rnd(B * R * N, seed=4)])
    ops = [gemm_op(a=ref(A, 0), w=ref(W, 0), bias=ref(W, N * K), out=ref(A, R * cin), res=ref(A, R * cin + R * N),
                   r_out=R, r_in=R, lda=cin, cin=cin, taps=taps, t_dj=1 if taps > 1 else 0,
                   t_off=-(taps // 2), n=N, ldc=N, o_rows=R, ldr=N, act=0)]
    (ga, _, _), (ca, _, _) = run_both(ops, weights, act, torch.zeros(4), {}, B)
    assert (ga - ca).abs().max() < 2e-5
    ops[0].i[rt.G_ACT] = 1
    ops[0].res = ref(rt.SP_NONE)
    (ga, _, _), (ca, _, _) = run_both(ops, weights, act, torch.zeros(4), {}, B)
    assert (ga - ca).abs().max() < 2e-5


@pytest.mark.parametrize("B,R,cin,N", [(37, 1, 256, 1024), (70, 1, 256, 256), (129, 4, 128, 1024), (5, 4, 128, 64), (33, 16, 128, 192),
                                       (300, 1, 256, 64), (1100, 4, 128, 512), (2100, 1, 256, 1024),
                                       # enough row blocks to fill the chip alone: ONE workgroup per row block walks all 16 chunks
                                       (4200, 4, 128, 1024), (8200, 1, 256, 1024)])
@pytest.mark.parametrize("ln,res", [("plain", False), (False, True), ("plain", True), ("affine", False), ("affine", True)])
def test_row_stationary_projection_on_ring_tiles(B, R, cin, N, ln, res, prod):
    """k_proj (MDT_G_WFMT = 16): LayerNorm prologue (without affine -- what the compiler emits, gain / bias folded into W / bias -- and
    with gain / bias vectors), bias, residual IN PLACE on the output, row counts that are no multiple of the workgroup's rows, one to
    several workgroups per row block; against the interpreter (split-bf16 weights) and the exact fp32 product."""
    from moleculediffusiontransformer_amd.compiler import UNetCompiler
    w = rnd(N, cin, seed=1, scale=cin ** -0.5)
    f32 = prod == "f32"                  # MDT_G_WFMT = 17: fp32 fragment tiles, exact fp32 MFMA products
    tile = UNetCompiler._tile_f32 if f32 else UNetCompiler._tile
    tiles = [tile(w[64 * c: 64 * c + 64, 128 * h: 128 * h + 128]) for c in range(N // 64) for h in range(cin // 128)]
    wt = torch.cat(tiles)
    weights = torch.cat([wt, rnd(N, seed=2), 1 + 0.1 * rnd(cin, seed=5), 0.1 * rnd(cin, seed=6)])
    nb = wt.numel()
    x = rnd(B * R * cin, seed=3) * 1.3 + 0.2
    out0 = rnd(B * R * N, seed=4)
    act = torch.cat([x, out0])
    op = gemm_op(a=ref(A, 0), w=ref(W, 0), bias=ref(W, nb), out=ref(A, R * cin), r_out=R, r_in=R, lda=cin, cin=cin, taps=1, n=N,
                 ldc=N, o_rows=R, eps=1e-5)
    if ln:
        op.i[rt.G_PRO] = rt.PRO_LAYERNORM
        if ln == "affine":
            op.p0, op.p1 = ref(W, nb + N), ref(W, nb + N + cin)
    if res:
        op.res = ref(A, R * cin)
        op.i[rt.G_LDR] = N
    op.i[rt.G_WFMT] = 17 if f32 else 16
    (ga, _, _), (ca, _, _) = run_both([op], weights, act, torch.zeros(4), {}, B)
    assert torch.equal(ga[: B * R * cin], x)
    assert (ga - ca).abs().max() < (1e-5 if f32 else 5e-5)      # (summation order; the split planes carry ~2^-17 relative rounding)
    a = x.view(B * R, cin)
    if ln == "affine":
        a = torch.nn.functional.layer_norm(a, (cin,), weights[nb + N: nb + N + cin], weights[nb + N + cin:], 1e-5)
    elif ln:
        a = torch.nn.functional.layer_norm(a, (cin,), eps=1e-5)
    y = a.double() @ w.double().T + weights[nb: nb + N].double() + (out0.view(B * R, N).double() if res else 0)
    assert (ga[B * R * cin:].view(B * R, N).double() - y).abs().max() < (1e-5 if f32 else 1e-4)


@pytest.mark.parametrize("B,T,C", [(70, 1, 256), (2100, 1, 256), (4100, 1, 256), (129, 4, 128), (1100, 4, 128), (5, 16, 128), (37, 2, 256)])
@pytest.mark.parametrize("res", [False, True])
def test_row_stationary_conv_as_a_k1024_projection(B, T, C, res, prod):
    """MDT_R_KSRC: the 1024 / C channel blocks of one input tensor as the sources of k_rconv (the next block's rows requested while
    the current block's tiles run) = out = x W^T + b (+ res in place), K = 1024; against the interpreter and the exact product."""
    from moleculediffusiontransformer_amd.compiler import UNetCompiler
    K, ks = 1024, 1024 // C
    f32 = prod == "f32"
    tile = UNetCompiler._tile_f32 if f32 else UNetCompiler._tile
    w = rnd(C, K, seed=1, scale=K ** -0.5)
    tiles = [tile(w[64 * ch: 64 * ch + 64, s_ * C + 128 * kh: s_ * C + 128 * kh + 128])
             for s_ in range(ks) for kh in range(C // 128) for ch in range(C // 64)]
    wt = torch.cat(tiles)
    weights = torch.cat([wt, rnd(C, seed=2)])
    x = rnd(B * T * K, seed=3)
    out0 = rnd(B * T * C, seed=4)
    act = torch.cat([x, out0])
    op = rt.MdtOp()
    op.kind = rt.OP_RCONV
    op.a, op.out, op.w, op.bias = ref(A, 0), ref(A, T * K), ref(W, 0), ref(W, wt.numel())
    if res:
        op.res = ref(A, T * K)
    i = op.i
    i[rt.R_T], i[rt.R_C], i[rt.R_LDA], i[rt.R_LDC], i[rt.R_TAPS], i[rt.R_LDR] = T, C, K, C, 1, (C if res else 0)
    i[rt.R_FILM_LD], i[rt.R_WF32], i[rt.R_KSRC] = C, int(f32), ks
    op.f[0], op.f[1] = 1e-5, 1.0
    (ga, _, _), (ca, _, _) = run_both([op], weights, act, torch.zeros(4), {}, B)
    assert torch.equal(ga[: B * T * K], x)
    assert (ga - ca).abs().max() < 5e-5
    y = x.view(B * T, K).double() @ w.double().T + weights[wt.numel():].double() + (out0.view(B * T, C).double() if res else 0)
    assert (ga[B * T * K:].view(B * T, C).double() - y).abs().max() < (2e-5 if f32 else 1e-4)


@pytest.mark.parametrize("B,L,ci,co,f", [(37, 64, 64, 128, 4), (1030, 64, 64, 128, 4), (70, 16, 128, 256, 4), (300, 16, 64, 128, 4),
                                         (9, 4, 128, 256, 4), (2100, 4, 128, 256, 4), (5, 8, 128, 256, 4)])
def test_strided_convolution_in_patch_form(B, L, ci, co, f, prod):
    """DownsampleBlock1d's Conv1d(k = 2 f + 1, stride f, padding f) (modules.py:62-75) as k_rconv over PATCHES (f tokens = one row of
    f ci values; compiler.py::down_patch): 256 -> 128 channels with MDT_R_HALF_OUT, 512 -> 256 with the row's two halves as sources
    (MDT_R_KSRC = 2); one output token per sample keeps the centre tap only.  Against torch's strided convolution."""
    from moleculediffusiontransformer_amd.compiler import UNetCompiler
    f32 = prod == "f32"
    tile = UNetCompiler._tile_f32 if f32 else UNetCompiler._tile
    w = rnd(co, ci, 2 * f + 1, seed=1, scale=(ci * (2 * f + 1)) ** -0.5)
    bias = rnd(co, seed=2)
    cp, T = ci * f, L // f
    w3 = torch.zeros(co, cp, 3)
    for q in range(f):
        w3[:, q * ci: (q + 1) * ci, 0] = w[:, :, q]
        w3[:, q * ci: (q + 1) * ci, 1] = w[:, :, f + q]
    w3[:, :ci, 2] = w[:, :, 2 * f]
    half, C = cp == 256, 256
    ks = 0 if half else 2
    if half:
        w3 = torch.cat([w3, torch.zeros_like(w3)])
    taps = 3
    if T == 1:
        w3, taps = w3[:, :, 1:2], 1
    nsrc = max(ks, 1)
    tiles = [tile(w3[64 * ch: 64 * ch + 64, s_ * C + 128 * kh: s_ * C + 128 * kh + 128, tap])
             for s_ in range(nsrc) for tap in range(taps) for kh in range(2) for ch in range(4)]
    wt = torch.cat(tiles)
    weights = torch.cat([wt, bias])
    x = rnd(B * L * ci, seed=3)
    act = torch.cat([x, torch.zeros(B * T * co)])
    op = rt.MdtOp()
    op.kind = rt.OP_RCONV
    op.a, op.out, op.w, op.bias = ref(A, 0), ref(A, L * ci), ref(W, 0), ref(W, wt.numel())
    i = op.i
    i[rt.R_T], i[rt.R_C], i[rt.R_LDA], i[rt.R_LDC], i[rt.R_TAPS] = T, C, cp, co, taps
    i[rt.R_FILM_LD], i[rt.R_WF32], i[rt.R_KSRC], i[rt.R_HALF_OUT] = C, int(f32), ks, int(half)
    op.f[0], op.f[1] = 1e-5, 1.0
    (ga, _, _), (ca, _, _) = run_both([op], weights, act, torch.zeros(4), {}, B)
    assert torch.equal(ga[: B * L * ci], x)
    assert (ga - ca).abs().max() < 5e-5
    y = torch.nn.functional.conv1d(x.view(B, L, ci).transpose(1, 2).double(), w.double(), bias.double(), stride=f, padding=f)
    assert (ga[B * L * ci:].view(B, T, co).double() - y.transpose(1, 2)).abs().max() < (2e-5 if f32 else 1e-4)


@pytest.mark.parametrize("B,T,ci,co,f", [(37, 4, 256, 128, 4), (1030, 4, 256, 128, 4), (70, 16, 128, 64, 4), (1100, 16, 128, 64, 4),
                                         (9, 1, 256, 128, 4), (2100, 1, 256, 128, 4), (5, 8, 128, 64, 4)])
@pytest.mark.parametrize("res", [False, True])
def test_transposed_convolution_in_patch_form(B, T, ci, co, f, res, prod):
    """UpsampleBlock1d's ConvTranspose1d(k = 2 f, stride f, padding f / 2) (modules.py:74-81) as ONE k_rconv launch that writes
    patches (MDT_R_NB: f co = NB ci output channels per input token; compiler.py::up_patch), residual added in place;
    against torch's transposed convolution."""
    from moleculediffusiontransformer_amd.compiler import UNetCompiler
    f32 = prod == "f32"
    tile = UNetCompiler._tile_f32 if f32 else UNetCompiler._tile
    wt = rnd(ci, co, 2 * f, seed=1, scale=(2 * ci) ** -0.5)
    bias = rnd(co, seed=2)
    h, C, nb = f // 2, ci, f * co // ci
    w3 = torch.zeros(f * co, ci, 3)
    for j in range(f):
        rows = slice(j * co, (j + 1) * co)
        if j < h:
            w3[rows, :, 0], w3[rows, :, 1] = wt[:, :, j + h + f].T, wt[:, :, j + h].T
        else:
            w3[rows, :, 1], w3[rows, :, 2] = wt[:, :, j - h + f].T, wt[:, :, j - h].T
    taps = 3
    if T == 1:
        w3, taps = w3[:, :, 1:2], 1
    tiles = [tile(w3[b_ * C + 64 * ch: b_ * C + 64 * ch + 64, 128 * kh: 128 * kh + 128, tap])
             for b_ in range(nb) for tap in range(taps) for kh in range(C // 128) for ch in range(C // 64)]
    wtl = torch.cat(tiles)
    weights = torch.cat([wtl, bias.repeat(f)])
    x = rnd(B * T * ci, seed=3)
    out0 = rnd(B * T * f * co, seed=4)
    act = torch.cat([x, out0])
    op = rt.MdtOp()
    op.kind = rt.OP_RCONV
    op.a, op.out, op.w, op.bias = ref(A, 0), ref(A, T * ci), ref(W, 0), ref(W, wtl.numel())
    if res:
        op.res = ref(A, T * ci)
    i = op.i
    i[rt.R_T], i[rt.R_C], i[rt.R_LDA], i[rt.R_LDC], i[rt.R_TAPS], i[rt.R_LDR] = T, C, ci, f * co, taps, (f * co if res else 0)
    i[rt.R_FILM_LD], i[rt.R_WF32], i[rt.R_NB] = C, int(f32), nb
    op.f[0], op.f[1] = 1e-5, 1.0
    (ga, _, _), (ca, _, _) = run_both([op], weights, act, torch.zeros(4), {}, B)
    assert torch.equal(ga[: B * T * ci], x)
    assert (ga - ca).abs().max() < 5e-5
    y = torch.nn.functional.conv_transpose1d(x.view(B, T, ci).transpose(1, 2).double(), wt.double(), bias.double(), stride=f, padding=h)
    want = y.transpose(1, 2) + (out0.view(B, T * f, co).double() if res else 0)
    assert (ga[B * T * ci:].view(B, T * f, co).double() - want).abs().max() < (2e-5 if f32 else 1e-4)


def test_gemm_strided_conv_and_transposed_phases():
    B, Lin, cin, N, f = 3, 16, 64, 128, 4
    # Conv1d k=9 s=4 p=4 (modules.py:40-51)
    weights = torch.cat([rnd(N, 9 * cin, seed=1, scale=0.05).view(-1), rnd(N, seed=2)])
    act = torch.cat([rnd(B * Lin * cin, seed=3), torch.zeros(B * (Lin // f) * N)])
    ops = [gemm_op(a=ref(A, 0), w=ref(W, 0), bias=ref(W, N * 9 * cin), out=ref(A, Lin * cin), r_out=Lin // f,
                   r_in=Lin, lda=cin, cin=cin, taps=9, t_stride=f, t_dj=1, t_off=-f, n=N, ldc=N, o_rows=Lin // f)]
    (ga, _, _), (ca, _, _) = run_both(ops, weights, act, torch.zeros(4), {}, B)
    assert (ga - ca).abs().max() < 2e-5
    x = act[: B * Lin * cin].view(B, Lin, cin).transpose(1, 2)
    wt = weights[: N * 9 * cin].view(N, 9, cin).permute(0, 2, 1)
    y = torch.nn.functional.conv1d(x, wt, weights[N * 9 * cin:], stride=f, padding=f)
    assert (ga[B * Lin * cin:].view(B, Lin // f, N) - y.transpose(1, 2)).abs().max() < 2e-5
    # ConvTranspose1d k=8 s=4 p=2 as 4 phases (modules.py:74-81), with a residual on the output rows
    wt = rnd(cin, N, 2 * f, seed=5, scale=0.05)
    bias = rnd(N, seed=6)
    chunks, offs, o = [], [], 0
    for ph in range(f):
        wp = torch.stack((wt[:, :, ph], wt[:, :, ph + f]), 0).permute(2, 0, 1).contiguous().view(-1)
        chunks.append(wp)
        offs.append(o)
        o += wp.numel()
    weights = torch.cat(chunks + [bias])
    Lout = Lin * f
    act = torch.cat([rnd(B * Lin * cin, seed=7), torch.zeros(B * Lout * N), rnd(B * Lout * N, seed=8)])
    ops = []
    for ph in range(f):
        shift = 1 if ph < f // 2 else 0
        ops.append(gemm_op(a=ref(A, 0), w=ref(W, offs[ph]), bias=ref(W, o), out=ref(A, Lin * cin),
                           res=ref(A, Lin * cin + Lout * N), r_out=Lin, r_in=Lin, lda=cin, cin=cin, taps=2, t_dj=-1,
                           t_off=shift, n=N, ldc=N, o_rows=Lout, o_stride=f, o_off=f * shift + ph - f // 2, ldr=N))
    (ga, _, _), (ca, _, _) = run_both(ops, weights, act, torch.zeros(4), {}, B)
    assert (ga - ca).abs().max() < 2e-5
    x = act[: B * Lin * cin].view(B, Lin, cin).transpose(1, 2)
    y = torch.nn.functional.conv_transpose1d(x, wt, bias, stride=f, padding=f // 2)
    y = y.transpose(1, 2) + act[B * Lin * cin + B * Lout * N:].view(B, Lout, N)
    assert (ga[B * Lin * cin: B * Lin * cin + B * Lout * N].view(B, Lout, N) - y).abs().max() < 2e-5


@pytest.mark.parametrize("B,R,C,N", [(4, 16, 128, 512), (3, 12, 128, 1024), (2, 4, 256, 512), (130, 1, 64, 64)])
def test_gemm_layernorm_prologue(B, R, C, N):
    weights = torch.cat([rnd(N, C, seed=1, scale=C ** -0.5).view(-1), 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)])
    act = torch.cat([rnd(B * R * C, seed=4) * 2 + 0.5, torch.zeros(B * R * N)])
    ops = [gemm_op(a=ref(A, 0), w=ref(W, 0), out=ref(A, R * C), p0=ref(W, N * C), p1=ref(W, N * C + C), r_out=R,
                   r_in=R, lda=C, cin=C, taps=1, n=N, ldc=N, o_rows=R, pro=rt.PRO_LAYERNORM, eps=1e-5)]
    (ga, _, _), (ca, _, _) = run_both(ops, weights, act, torch.zeros(4), {}, B)
    assert (ga - ca).abs().max() < 3e-5
    x = act[: B * R * C].view(B * R, C)
    y = torch.nn.functional.layer_norm(x, (C,), weights[N * C: N * C + C], weights[N * C + C:], 1e-5) @ \
        weights[: N * C].view(N, C).T
    assert (ga[B * R * C:].view(B * R, N) - y).abs().max() < 3e-5


@pytest.mark.parametrize("B,R,C,G,N,film,silu,eps", [(3, 64, 64, 1, 64, False, True, 1e-5),
                                                      (2, 16, 128, 8, 128, True, True, 1e-5),
                                                      (5, 4, 512, 8, 256, True, True, 1e-5),
                                                      (3, 16, 128, 32, 128, False, False, 1e-6),
                                                      (2, 32, 32, 32, 32, False, False, 1e-6)])
def test_gn_stats_and_groupnorm_prologue(B, R, C, G, N, film, silu, eps):
    gs = C // G
    taps = 3 if silu else 1
    K = taps * C
    weights = torch.cat([rnd(N, K, seed=1, scale=K ** -0.5).view(-1), 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)])
    shr = 0.3 * rnd(2 * C, seed=9)
    xoff, stoff, ooff = 0, R * C, R * C + 64
    act = torch.zeros(B * (R * C + 64 + R * N))
    act[: B * R * C] = rnd(B * R * C, seed=4) * 1.5 + 0.3
    st = rt.MdtOp()
    st.kind = rt.OP_GN_STATS
    st.a, st.out = ref(A, xoff), ref(A, stoff)
    st.i[rt.N_ROWS], st.i[rt.N_LD], st.i[rt.N_GROUPS], st.i[rt.N_GSIZE] = R, C, G, gs
    st.f[0] = eps
    ops = [st, gemm_op(a=ref(A, xoff), w=ref(W, 0), out=ref(A, ooff), p0=ref(W, N * K), p1=ref(W, N * K + C),
                       p2=ref(A, stoff), p3=ref(S, 0) if film else ref(rt.SP_NONE), r_out=R, r_in=R, lda=C, cin=C,
                       taps=taps, t_dj=1 if taps > 1 else 0, t_off=-(taps // 2), n=N, ldc=N, o_rows=R,
                       pro=rt.PRO_GROUPNORM, groups=G, gsize=gs, pro_silu=int(silu))]
    # NB: stats are per-sample 2G floats at ACT offset stoff (scaled by B at run time) -> needs 2G <= 64
    (ga, _, _), (ca, _, _) = run_both(ops, weights, act, shr, {}, B)
    assert (ga - ca).abs().max() < 5e-5
    # independent check against torch's GroupNorm + conv
    x = act[: B * R * C].view(B, R, C).transpose(1, 2)
    h = torch.nn.functional.group_norm(x, G, weights[N * K: N * K + C], weights[N * K + C:], eps)
    if film:
        h = h * (shr[:C].view(1, C, 1) + 1) + shr[C:].view(1, C, 1)
    if silu:
        h = torch.nn.functional.silu(h)
    y = torch.nn.functional.conv1d(h, weights[: N * K].view(N, taps, C).permute(0, 2, 1), padding=taps // 2)
    assert (ga[B * ooff:].view(B, R, N) - y.transpose(1, 2)).abs().max() < 5e-5


@pytest.mark.parametrize("B,T,Tk,shared", [(3, 16, 16, False), (2, 16, 12, False), (5, 4, 4, False), (2, 4, 12, True),
                                           (2, 1, 64, False), (1, 32, 32, False), (3, 16, 12, True),
                                           # more than 64 queries / keys per sample: k_attn_long (online softmax over key chunks)
                                           (2, 256, 256, False), (1, 100, 70, False), (2, 256, 32, True), (1, 20, 300, False),
                                           (1, 1024, 1024, False)])
def test_attention(B, T, Tk, shared):
    H, D = 8, 64
    act = torch.cat([rnd(B * T * H * D, seed=1), rnd(B * Tk * 2 * H * D, seed=2), torch.zeros(B * T * H * D)])
    shr = rnd(Tk * 2 * H * D, seed=3)
    op = rt.MdtOp()
    op.kind = rt.OP_ATTN
    op.a, op.out = ref(A, 0), ref(A, T * H * D + Tk * 2 * H * D)
    op.a2 = ref(S, 0) if shared else ref(A, T * H * D)
    i = op.i
    i[rt.A_T], i[rt.A_TK], i[rt.A_HEADS], i[rt.A_LDQ], i[rt.A_LDKV], i[rt.A_LDO] = T, Tk, H, H * D, 2 * H * D, H * D
    i[rt.A_KV_BSTRIDE] = 0 if shared else Tk
    op.f[0] = D ** -0.5
    (ga, _, _), (ca, _, _) = run_both([op], torch.zeros(4), act, shr, {}, B)
    assert (ga - ca).abs().max() < 1e-5


@pytest.mark.parametrize("B,T,Tk,shared", [(3, 32, 32, False), (2, 8, 8, False), (3, 32, 12, True), (2, 8, 12, True), (1, 100, 70, False)])
@pytest.mark.parametrize("in16", [pytest.param(1, marks=pytest.mark.slow), 2, 3])      # default: k | v in bf16, and q | k | v
@pytest.mark.parametrize("out16", [0, 1])
def test_attention_with_bf16_operands(B, T, Tk, shared, in16, out16):
    """MDT_A_IN16 (plain-bf16 mode): q and / or k | v arrive as bf16 (written so by their GEMM), are widened exactly and contracted in
    fp32; the merged q | k | v tensor of a self-attention layer (one pitch, column offsets) when all three are bf16."""
    H, D = 8, 64
    merged = in16 == 3 and not shared and T == Tk
    qf, kvf = rnd(B * T * H * D, seed=1), rnd((1 if shared else B) * Tk * 2 * H * D, seed=2)

    def pack(x, half):
        return x.to(torch.bfloat16).view(torch.float32) if half else x
    if merged:
        qkv = torch.cat([qf.view(B * T, H * D), kvf.view(B * Tk, 2 * H * D)], 1)
        regions = [pack(qkv.reshape(-1), True)]
    else:
        regions = [pack(qf, in16 & 1)] + ([] if shared else [pack(kvf, in16 & 2)])
    n_in = sum(r.numel() for r in regions)
    n_out = B * T * H * D // (2 if out16 else 1)
    act = torch.cat(regions + [torch.zeros(n_out)])
    shr = pack(kvf, in16 & 2) if shared else torch.zeros(4)
    op = rt.MdtOp()
    op.kind = rt.OP_ATTN
    op.a, op.out = ref(A, 0), ref(A, n_in // B)
    i = op.i
    if merged:
        op.a2 = ref(A, 0)
        i[rt.A_LDQ], i[rt.A_LDKV], i[rt.A_KCOL] = 3 * H * D, 3 * H * D, H * D
    else:
        op.a2 = ref(S, 0) if shared else ref(A, regions[0].numel() // B)
        i[rt.A_LDQ], i[rt.A_LDKV] = H * D, 2 * H * D
    i[rt.A_T], i[rt.A_TK], i[rt.A_HEADS], i[rt.A_LDO] = T, Tk, H, H * D
    i[rt.A_KV_BSTRIDE] = 0 if shared else Tk
    i[rt.A_IN16], i[rt.A_OUT16] = in16, out16
    op.f[0] = D ** -0.5
    (ga, _, _), (ca, _, _) = run_both([op], torch.zeros(4), act, shr, {}, B)
    go, co = ga[n_in:], ca[n_in:]
    if out16:
        go, co = go.view(torch.bfloat16).float(), co.view(torch.bfloat16).float()
        assert (go - co).abs().max() < 2e-2 and ((go - co).abs() > 1e-5).float().mean() < 1e-3      # a rounding boundary, rarely
    else:
        assert (go - co).abs().max() < 1e-5
    # and against torch on the widened values
    q = (qf.to(torch.bfloat16).float() if in16 & 1 else qf).view(B, T, H, D).transpose(1, 2)
    kv = (kvf.to(torch.bfloat16).float() if in16 & 2 else kvf).view(-1, Tk, 2, H, D)
    k, v = kv[:, :, 0].transpose(1, 2), kv[:, :, 1].transpose(1, 2)
    o = ((q @ k.transpose(-1, -2) * D ** -0.5).softmax(-1) @ v).transpose(1, 2).reshape(-1)
    assert (go - (o.to(torch.bfloat16).float() if out16 else o)).abs().max() < (2e-2 if out16 else 1e-5)


@pytest.mark.parametrize("B,T,Tk,shared", [(3, 16, 64, False), (2, 32, 64, True), (5, 4, 40, False), (2, 1, 33, False),
                                           (1, 64, 64, False), (3, 8, 16, True), (2, 3, 20, False), (3, 5, 64, True),
                                           (2100, 1, 33, False), (1100, 4, 64, False), (650, 16, 20, True), (2500, 2, 9, False)])
@pytest.mark.parametrize("split", [0, 1])
def test_attention_on_normalised_context(B, T, Tk, shared, split):
    """MDT_OP_ATTN_CTX: softmax(q' c^T scale) c with keys = values = the context rows, against the interpreter and torch.
    The large batches give every wave of the (persistent) launch several work units: the stream of context chunks then runs
    across unit boundaries (query rows reloaded, previous unit stored from the accumulators, 1 / 2 / 3 / 4 chunks per unit)."""
    H, F_ = 8, 128
    R = T * H
    act = torch.cat([rnd(B * R * F_, seed=1) * 0.3, rnd(B * Tk * F_, seed=2), torch.zeros(B * R * F_)])
    shr = rnd(Tk * F_, seed=3)
    op = rt.MdtOp()
    op.kind = rt.OP_ATTN_CTX
    op.a, op.out = ref(A, 0), ref(A, R * F_ + Tk * F_)
    op.a2 = ref(S, 0) if shared else ref(A, R * F_)
    i = op.i
    i[rt.A_T], i[rt.A_TK], i[rt.A_HEADS], i[rt.A_LDQ], i[rt.A_LDKV], i[rt.A_LDO] = T, Tk, H, F_, F_, F_
    i[rt.A_KV_BSTRIDE] = 0 if shared else Tk
    i[rt.A_SPLIT] = split           # 1: the scores as split-bf16 products (2^-17 per product), 0: exact fp32 MFMA
    op.f[0] = 0.125
    (ga, _, _), (ca, _, _) = run_both([op], torch.zeros(4), act, shr, {}, B)
    assert (ga - ca).abs().max() < 1e-5
    q = act[: B * R * F_].view(B, R, F_).double()
    c = (shr.view(1, Tk, F_).expand(B, -1, -1) if shared else act[B * R * F_: B * (R + Tk) * F_].view(B, Tk, F_)).double()
    want = ((q @ c.transpose(1, 2)) * 0.125).softmax(-1) @ c
    assert (ga[B * (R + Tk) * F_:].view(B, R, F_).double() - want).abs().max() < 1e-5


def test_concat_patch_time_embed():
    B, R, Ca, Cb = 3, 16, 128, 128
    act = torch.cat([rnd(B * R * Ca, seed=1), rnd(B * R * Cb, seed=2), torch.zeros(B * R * (Ca + Cb))])
    op = rt.MdtOp()
    op.kind = rt.OP_CONCAT
    op.a, op.a2, op.out = ref(A, 0), ref(A, R * Ca), ref(A, R * (Ca + Cb))
    op.i[rt.C_ROWS], op.i[rt.C_CA], op.i[rt.C_CB] = R, Ca, Cb
    op.f[0] = 2 ** -0.5
    (ga, _, _), (ca, _, _) = run_both([op], torch.zeros(4), act, torch.zeros(4), {}, B)
    assert torch.equal(ga, ca)
    # Patcher / Unpatcher round trip
    L, C, p = 64, 16, 4
    act = torch.cat([rnd(B * L * C, seed=3), torch.zeros(B * L * C), torch.zeros(B * L * C)])
    fwd, inv = rt.MdtOp(), rt.MdtOp()
    for o, (src, dst, inverse) in ((fwd, (0, L * C, 0)), (inv, (L * C, 2 * L * C, 1))):
        o.kind = rt.OP_PATCH
        o.a, o.out = ref(A, src), ref(A, dst)
        o.i[rt.P_ROWS_IN], o.i[rt.P_C_IN], o.i[rt.P_PATCH], o.i[rt.P_INVERSE] = L, C, p, inverse
        o.i[rt.P_LD_IN], o.i[rt.P_LD_OUT] = (C * p, C) if inverse else (C, C * p)
    (ga, _, _), (ca, _, _) = run_both([fwd, inv], torch.zeros(4), act, torch.zeros(4), {}, B)
    assert torch.equal(ga, ca)
    assert torch.equal(ga[2 * B * L * C:], ga[: B * L * C])
    x = act[: B * L * C].view(B, L, C).transpose(1, 2)                      # b c (l p)
    y = x.reshape(B, C, L // p, p).permute(0, 1, 3, 2).reshape(B, C * p, L // p)   # 'b c (l p) -> b (c p) l'
    assert torch.equal(ga[B * L * C: 2 * B * L * C].view(B, L // p, C * p), y.transpose(1, 2))
    # LearnedPositionalEmbedding
    n, half, ld = 7, 32, 80
    shr = torch.cat([torch.linspace(-1.7, 0.55, n), torch.zeros(57), torch.zeros(n * ld)])
    weights = rnd(half, seed=5)
    t = rt.MdtOp()
    t.kind = rt.OP_TIME_EMBED
    t.a, t.w, t.out = ref(S, 0), ref(W, 0), ref(S, 64)
    t.i[rt.T_HALF], t.i[rt.T_LD] = half, ld
    (_, gs, _), (_, cs, _) = run_both([t], weights, torch.zeros(4), shr, {}, 1, n)
    assert (gs - cs).abs().max() < 2e-6


def test_row_copy_of_the_film_table():
    """mdt_copy_f32 (engine.select_time: the FiLM rows of one evaluation out of the per-call table): aligned rows take the 16-byte
    path, odd offsets / lengths the scalar one; bit-exact either way, nothing outside [dst, dst + n) is touched."""
    lib = rt.load_library()
    buf = rnd(20000, seed=3).to(DEV)
    for dst, src, n in ((0, 8192, 5120), (4, 10001, 4097), (3, 9000, 7), (16, 8192, 4)):
        b = buf.clone()
        want = b.clone()
        want[dst: dst + n] = want[src: src + n]
        with torch.cuda.device(DEV):
            rt.check(lib.mdt_copy_f32(b.data_ptr() + 4 * dst, b.data_ptr() + 4 * src, n, rt.current_stream()))
            torch.cuda.synchronize()
        assert torch.equal(b, want), (dst, src, n)
    assert lib.mdt_copy_f32(0, 0, 0, 0) == 0 and lib.mdt_copy_f32(0, buf.data_ptr(), 4, 0) != 0


def test_sampler_kernels_match_reference_arithmetic():
    lib = rt.load_library()
    B, C, L, Cp = 5, 22, 32, 32
    x, xm, nz = rnd(B, C, L, seed=1) * 3, rnd(B, C, L, seed=2) * 3, rnd(B, C, L, seed=3)
    pred = rnd(B, L, Cp, seed=4)
    c_skip, c_out, c_in, sigma, sigma_mid, dt_mid, dt_down, up = 0.31, 0.095, 3.3, 0.29, 0.21, -0.08, -0.11, 0.17
    gx, gxm, gnz, gp = (t.to(DEV).contiguous() for t in (x, xm, nz, pred))
    with torch.cuda.device(DEV):
        st = rt.current_stream()
        xin = torch.full((B, L, Cp), 7.0, device=DEV)
        rt.check(lib.mdt_precond_in(rt.ptr(gx), rt.ptr(xin), c_in, B, C, L, Cp, st))
        want = torch.zeros(B, L, Cp)
        want[:, :, :C] = (c_in * x).transpose(1, 2)
        assert torch.equal(xin.cpu(), want)
        D = torch.empty_like(gx)
        rt.check(lib.mdt_precond_out(rt.ptr(gx), rt.ptr(gp), rt.ptr(D), c_skip, c_out, B, C, L, Cp, 0, st))
        p = pred[:, :, :C].transpose(1, 2)
        den = (c_skip * x + c_out * p).clamp(-1.0, 1.0)
        assert torch.equal(D.cpu(), den)
        out_mid, xin_mid = torch.empty_like(gx), torch.empty_like(xin)
        rt.check(lib.mdt_adpm2_mid(rt.ptr(gx), rt.ptr(gp), rt.ptr(out_mid), rt.ptr(xin_mid), c_skip, c_out, sigma,
                                   dt_mid, c_in, B, C, L, Cp, 0, st))
        sg = torch.tensor(sigma)
        x_mid = x + ((x - den) / sg) * torch.tensor(dt_mid)
        assert torch.equal(out_mid.cpu(), x_mid)
        assert torch.equal(xin_mid.cpu()[:, :, :C], (torch.tensor(c_in) * x_mid).transpose(1, 2))
        assert float(xin_mid[:, :, C:].abs().max()) == 0.0
        x2 = gx.clone()
        rt.check(lib.mdt_adpm2_next(rt.ptr(x2), rt.ptr(gxm), rt.ptr(gp), rt.ptr(gnz), rt.ptr(xin_mid), c_skip, c_out,
                                    sigma_mid, dt_down, up, c_in, 0, 0, 0, B, C, L, Cp, 0, 0, st))
        den2 = (c_skip * xm + c_out * p).clamp(-1.0, 1.0)
        want = x + ((xm - den2) / torch.tensor(sigma_mid)) * torch.tensor(dt_down)
        want = want + nz * torch.tensor(up)
        assert torch.equal(x2.cpu(), want)
        assert torch.equal(xin_mid.cpu()[:, :, :C], (torch.tensor(c_in) * want).transpose(1, 2))
        # guidance mix, clamp, argmax, inpaint merge
        a, b = rnd(B, L, Cp, seed=5).to(DEV), rnd(B, L, Cp, seed=6).to(DEV)
        o = torch.empty_like(a)
        rt.check(lib.mdt_cfg_mix(rt.ptr(a), rt.ptr(b), rt.ptr(o), 7.5, a.numel(), st))
        assert torch.equal(o.cpu(), b.cpu() + (a.cpu() - b.cpu()) * 7.5)
        c = gx.clone()
        rt.check(lib.mdt_clamp(rt.ptr(c), -1.0, 1.0, c.numel(), st))
        assert torch.equal(c.cpu(), x.clamp(-1, 1))
        tok = torch.empty(B, L, dtype=torch.int32, device=DEV)
        rt.check(lib.mdt_argmax_tokens(rt.ptr(gx), rt.ptr(tok), B, C, L, st))
        assert torch.equal(tok.cpu().long(), x.permute(0, 2, 1).argmax(dim=2))
        mask = (rnd(B, C, L, seed=8) > 0)
        gm = mask.to(torch.uint8).to(DEV)
        y = gx.clone()
        rt.check(lib.mdt_inpaint_merge(rt.ptr(y), rt.ptr(gxm), rt.ptr(gm), rt.ptr(gnz), 0.7, 0, 0, 0, B, C, L, st))
        src_noisy = xm + torch.tensor(0.7) * nz
        assert torch.equal(y.cpu(), src_noisy * mask + x * ~mask)


def test_counter_based_noise_is_normal_and_sharding_independent():
    lib = rt.load_library()
    B, C, L = 64, 16, 64
    with torch.cuda.device(DEV):
        st = rt.current_stream()
        full = torch.empty(B, C, L, device=DEV)
        rt.check(lib.mdt_init_noise(rt.ptr(full), 0, 1.0, 1234, 3, 0, B, C, L, st))
        lo, hi = torch.empty(B // 2, C, L, device=DEV), torch.empty(B // 2, C, L, device=DEV)
        rt.check(lib.mdt_init_noise(rt.ptr(lo), 0, 1.0, 1234, 3, 0, B // 2, C, L, st))
        rt.check(lib.mdt_init_noise(rt.ptr(hi), 0, 1.0, 1234, 3, B // 2, B // 2, C, L, st))
        other = torch.empty(B, C, L, device=DEV)
        rt.check(lib.mdt_init_noise(rt.ptr(other), 0, 1.0, 1234, 4, 0, B, C, L, st))
        torch.cuda.synchronize()
    assert torch.equal(torch.cat([lo, hi]), full)
    assert not torch.equal(other, full)
    f = full.cpu().double()
    n = f.numel()
    assert abs(f.mean()) < 4 / math.sqrt(n) and abs(f.var() - 1) < 0.02
    assert abs((f ** 3).mean()) < 0.05 and abs((f ** 4).mean() - 3) < 0.1
    assert abs((f[:, :, 1:] * f[:, :, :-1]).mean()) < 0.01


def _split_planes(w):
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    return hi.contiguous().view(-1).view(torch.float32), lo.contiguous().view(-1).view(torch.float32)


@pytest.mark.parametrize("B,R,cin,N,taps,pro", [
    (64, 16, 128, 512, 1, rt.PRO_LAYERNORM),      # 128x128 tiles
    (1024, 16, 128, 128, 3, rt.PRO_GROUPNORM),    # 128x64 tiles, conv taps, GN+FiLM+SiLU
    (3, 16, 128, 96, 3, rt.PRO_GROUPNORM),        # 64x64 tiles, ragged N and M
    (40, 4, 512, 256, 1, rt.PRO_NONE),
    (5, 64, 64, 64, 3, rt.PRO_NONE),
    (33, 1, 256, 160, 1, rt.PRO_SILU),
    (40, 4, 256, 256, 1, rt.PRO_GROUPNORM),       # A-stationary kernel, GN prologue
    (1024, 4, 256, 1024, 1, rt.PRO_LAYERNORM),    # A-stationary, column range split over workgroups
    (700, 16, 128, 384, 1, rt.PRO_NONE),
])
def test_gemm_split_bf16(B, R, cin, N, taps, pro):
    """k_gemm3 (bf16x3) against the interpreter (fp32 activations x reconstructed hi+lo weights)."""
    K = taps * cin
    w = rnd(N, K, seed=1, scale=K ** -0.5)
    hi, lo = _split_planes(w)
    G = 8
    gs = cin // G
    weights = torch.cat([hi, lo, rnd(N, seed=2), 1 + 0.1 * rnd(cin, seed=3), 0.1 * rnd(cin, seed=4)])
    o_hi, o_lo, o_b, o_g, o_nb = 0, hi.numel(), 2 * hi.numel(), 2 * hi.numel() + N, 2 * hi.numel() + N + cin
    xoff, stoff, ooff, roff = 0, R * cin, R * cin + 64, R * cin + 64 + R * N
    act = torch.zeros(B * (roff + R * N))
    act[: B * R * cin] = rnd(B * R * cin, seed=5) * 1.3 + 0.2
    act[B * roff:] = rnd(B * R * N, seed=6)
    shr = 0.3 * rnd(2 * cin, seed=7)
    ops = []
    if pro == rt.PRO_GROUPNORM:
        st = rt.MdtOp()
        st.kind = rt.OP_GN_STATS
        st.a, st.out = ref(A, xoff), ref(A, stoff)
        st.i[rt.N_ROWS], st.i[rt.N_LD], st.i[rt.N_GROUPS], st.i[rt.N_GSIZE] = R, cin, G, gs
        st.f[0] = 1e-5
        ops.append(st)
    op = gemm_op(a=ref(A, xoff), w=ref(W, o_hi), bias=ref(W, o_b), out=ref(A, ooff), res=ref(A, roff),
                 p0=ref(W, o_g), p1=ref(W, o_nb), p2=ref(A, stoff), p3=ref(S, 0), r_out=R, r_in=R, lda=cin, cin=cin,
                 taps=taps, t_dj=1 if taps > 1 else 0, t_off=-(taps // 2), n=N, ldc=N, o_rows=R, ldr=N, pro=pro,
                 groups=G, gsize=gs, pro_silu=1, act=0, eps=1e-5)
    op.a2 = ref(W, o_lo)
    ops.append(op)
    (ga, _, _), (ca, _, _) = run_both(ops, weights, act, shr, {}, B)
    out_g, out_c = ga[B * ooff: B * roff], ca[B * ooff: B * roff]
    scale = out_c.abs().max().item()
    assert (out_g - out_c).abs().max() < 4e-5 * max(scale, 1.0), ((out_g - out_c).abs().max().item(), scale)


@pytest.mark.parametrize("B,R,cin,N,taps,pro", [
    (64, 32, 512, 512, 3, rt.PRO_GROUPNORM),      # 128x128 tiles with 64-deep chunks, conv taps (deep-UNet level 1)
    (96, 8, 1024, 2048, 1, rt.PRO_LAYERNORM),     # LayerNorm prologue over 1024 features
    (3, 16, 128, 96, 3, rt.PRO_GROUPNORM),        # 64x64 tiles, ragged N and M
    (40, 4, 512, 256, 1, rt.PRO_NONE),
    (5, 64, 32, 64, 3, rt.PRO_NONE),              # 32 channels: 32-deep chunks
    (33, 1, 256, 160, 1, rt.PRO_SILU),
    (700, 16, 128, 384, 1, rt.PRO_NONE),
])
def test_gemm_plain_bf16(B, R, cin, N, taps, pro):
    """The plain-bf16 mode of k_gemm3 (MDT_G_WFMT = 1: one bf16 weight plane, A rounded to bf16 after the prologue, fp32
    accumulation) against the interpreter doing the same roundings, and its distance from the unrounded fp32 product
    (the stated budget of the mode: 2^-8 relative per operand -> ~1e-2 of the output scale)."""
    K = taps * cin
    w = rnd(N, K, seed=1, scale=K ** -0.5)
    hi = w.to(torch.bfloat16).contiguous().view(-1).view(torch.float32)
    G = 8
    gs = cin // G
    weights = torch.cat([hi, rnd(N, seed=2), 1 + 0.1 * rnd(cin, seed=3), 0.1 * rnd(cin, seed=4)])
    o_b, o_g, o_nb = hi.numel(), hi.numel() + N, hi.numel() + N + cin
    xoff, stoff, ooff, roff = 0, R * cin, R * cin + 64, R * cin + 64 + R * N
    act = torch.zeros(B * (roff + R * N))
    act[: B * R * cin] = rnd(B * R * cin, seed=5) * 1.3 + 0.2
    act[B * roff:] = rnd(B * R * N, seed=6)
    shr = 0.3 * rnd(2 * cin, seed=7)
    ops = []
    if pro == rt.PRO_GROUPNORM:
        st = rt.MdtOp()
        st.kind = rt.OP_GN_STATS
        st.a, st.out = ref(A, xoff), ref(A, stoff)
        st.i[rt.N_ROWS], st.i[rt.N_LD], st.i[rt.N_GROUPS], st.i[rt.N_GSIZE] = R, cin, G, gs
        st.f[0] = 1e-5
        ops.append(st)
    op = gemm_op(a=ref(A, xoff), w=ref(W, 0), bias=ref(W, o_b), out=ref(A, ooff), res=ref(A, roff),
                 p0=ref(W, o_g), p1=ref(W, o_nb), p2=ref(A, stoff), p3=ref(S, 0), r_out=R, r_in=R, lda=cin, cin=cin,
                 taps=taps, t_dj=1 if taps > 1 else 0, t_off=-(taps // 2), n=N, ldc=N, o_rows=R, ldr=N, pro=pro,
                 groups=G, gsize=gs, pro_silu=1, act=0, eps=1e-5)
    op.i[rt.G_WFMT] = 1
    ops.append(op)
    (ga, _, _), (ca, _, _) = run_both(ops, weights, act, shr, {}, B)
    out_g, out_c = ga[B * ooff: B * roff], ca[B * ooff: B * roff]
    scale = max(out_c.abs().max().item(), 1.0)
    # same roundings on both sides; a prologue value that lands on a bf16 tie can round the other way (one bf16 ulp of one
    # operand: ~2^-8 |a w|), hence not bitwise
    assert (out_g - out_c).abs().max() < (2e-3 if pro != rt.PRO_NONE else 4e-5) * scale
    # distance from the fp32 product (weights unrounded, activations unrounded)
    full = torch.cat([w.view(-1), weights[o_b:]])
    op32 = gemm_op(a=ref(A, xoff), w=ref(W, 0), bias=ref(W, N * K), out=ref(A, ooff), res=ref(A, roff),
                   p0=ref(W, N * K + N), p1=ref(W, N * K + N + cin), p2=ref(A, stoff), p3=ref(S, 0), r_out=R, r_in=R, lda=cin,
                   cin=cin, taps=taps, t_dj=1 if taps > 1 else 0, t_off=-(taps // 2), n=N, ldc=N, o_rows=R, ldr=N, pro=pro,
                   groups=G, gsize=gs, pro_silu=1, act=0, eps=1e-5)
    from oracle.program_interp import Buffers, run_program
    cpu = Buffers(full, act.clone(), shr.clone(), {})
    run_program(ops[:-1] + [op32], cpu, B, 0)
    exact = cpu.act[B * ooff: B * roff]
    assert (out_g - exact).abs().max() < 2e-2 * scale


@pytest.mark.parametrize("tile", ["", "0", "1", "2"])
@pytest.mark.parametrize("B,R,cin,N,taps,pro,act", [
    (64, 32, 512, 512, 3, rt.PRO_GROUPNORM, 0),   # level-1 ResNet convolution of the deep U-Net
    (70, 8, 1024, 2048, 1, rt.PRO_LAYERNORM, 1),  # feed-forward up-projection + GELU, ragged M
    (33, 8, 1024, 1024, 3, rt.PRO_GROUPNORM, 0),
    (5, 32, 64, 96, 3, rt.PRO_SILU, 0),           # ragged N
    (3, 128, 256, 256, 1, rt.PRO_NONE, 0),
])
def test_gemm_bf16_streamed(B, R, cin, N, taps, pro, act, tile):
    """MDT_OP_PREP16 + bf16 x bf16 GEMM (k_prep16 / k_gemm_b16: both operands by LDS-DMA) against the interpreter doing the
    same roundings, in every tile configuration."""
    import os
    K = taps * cin
    w = rnd(N, K, seed=1, scale=K ** -0.5)
    hi = w.to(torch.bfloat16).contiguous().view(-1).view(torch.float32)
    G = 8
    gs = cin // G
    weights = torch.cat([hi, rnd(N, seed=2), 1 + 0.1 * rnd(cin, seed=3), 0.1 * rnd(cin, seed=4)])
    o_b, o_g, o_nb = hi.numel(), hi.numel() + N, hi.numel() + N + cin
    xoff, stoff, a16off = 0, R * cin, R * cin + 64
    ooff = a16off + R * cin // 2
    roff = ooff + R * N
    act_buf = torch.zeros(B * (roff + R * N))
    act_buf[: B * R * cin] = rnd(B * R * cin, seed=5) * 1.3 + 0.2
    act_buf[B * roff:] = rnd(B * R * N, seed=6)
    shr = 0.3 * rnd(2 * cin, seed=7)
    ops = []
    if pro == rt.PRO_GROUPNORM:
        st = rt.MdtOp()
        st.kind = rt.OP_GN_STATS
        st.a, st.out = ref(A, xoff), ref(A, stoff)
        st.i[rt.N_ROWS], st.i[rt.N_LD], st.i[rt.N_GROUPS], st.i[rt.N_GSIZE] = R, cin, G, gs
        st.f[0] = 1e-5
        ops.append(st)
    pre = rt.MdtOp()
    pre.kind = rt.OP_PREP16
    pre.a, pre.out, pre.p0, pre.p1, pre.p2, pre.p3 = ref(A, xoff), ref(A, a16off), ref(W, o_g), ref(W, o_nb), ref(A, stoff), ref(S, 0)
    pi = pre.i
    pi[rt.G_R_IN], pi[rt.G_LDA], pi[rt.G_CIN], pi[rt.G_PRO], pi[rt.G_GROUPS], pi[rt.G_GSIZE], pi[rt.G_PRO_SILU] = R, cin, cin, pro, G, gs, 1
    pre.f[0] = 1e-5
    ops.append(pre)
    op = gemm_op(a=ref(A, a16off), w=ref(W, 0), bias=ref(W, o_b), out=ref(A, ooff), res=ref(A, roff), r_out=R, r_in=R, lda=cin,
                 cin=cin, taps=taps, t_dj=1 if taps > 1 else 0, t_off=-(taps // 2), n=N, ldc=N, o_rows=R, ldr=N, act=act)
    op.i[rt.G_WFMT] = 2
    ops.append(op)
    if tile:
        rt.load_library().mdt_set_tuning(b"tile16", int(tile))
    try:
        (ga, _, _), (ca, _, _) = run_both(ops, weights, act_buf, shr, {}, B)
    finally:
        rt.load_library().mdt_set_tuning(b"tile16", -1)
    out_g, out_c = ga[B * ooff: B * roff], ca[B * ooff: B * roff]
    scale = max(out_c.abs().max().item(), 1.0)
    assert (out_g - out_c).abs().max() < (2e-3 if pro != rt.PRO_NONE else 4e-5) * scale
    a16_g = ga[B * a16off: B * ooff].view(torch.bfloat16).float()
    a16_c = ca[B * a16off: B * ooff].view(torch.bfloat16).float()
    assert (a16_g - a16_c).abs().max() <= 2.0 ** -7 * max(a16_c.abs().max().item(), 1.0)     # at most one bf16 ulp (ties)


def test_gemm_bf16_chain_with_bf16_intermediate():
    """Feed-forward shape of the plain-bf16 mode: PREP16 -> GEMM (GELU, bf16 OUTPUT: MDT_G_WFMT 6) -> GEMM (+ residual),
    against the interpreter."""
    B, R, C, Hd = 37, 8, 1024, 2048
    w1 = rnd(Hd, C, seed=1, scale=C ** -0.5).to(torch.bfloat16).contiguous().view(-1).view(torch.float32)
    w2 = rnd(C, Hd, seed=2, scale=Hd ** -0.5).to(torch.bfloat16).contiguous().view(-1).view(torch.float32)
    weights = torch.cat([w1, w2, rnd(Hd, seed=3), rnd(C, seed=4)])
    o_w2, o_b1, o_b2 = w1.numel(), w1.numel() + w2.numel(), w1.numel() + w2.numel() + Hd
    xoff, a16off, hoff = 0, R * C, R * C + R * C // 2
    ooff = hoff + R * Hd // 2
    act = torch.zeros(B * (ooff + R * C))
    act[: B * R * C] = rnd(B * R * C, seed=5)
    pre = rt.MdtOp()
    pre.kind = rt.OP_PREP16
    pre.a, pre.out = ref(A, xoff), ref(A, a16off)
    pre.i[rt.G_R_IN], pre.i[rt.G_LDA], pre.i[rt.G_CIN] = R, C, C
    g1 = gemm_op(a=ref(A, a16off), w=ref(W, 0), bias=ref(W, o_b1), out=ref(A, hoff), r_out=R, r_in=R, lda=C, cin=C, taps=1,
                 n=Hd, ldc=Hd, o_rows=R, act=1)
    g1.i[rt.G_WFMT] = 6
    g2 = gemm_op(a=ref(A, hoff), w=ref(W, o_w2), bias=ref(W, o_b2), out=ref(A, ooff), res=ref(A, xoff), r_out=R, r_in=R,
                 lda=Hd, cin=Hd, taps=1, n=C, ldc=C, o_rows=R, ldr=C)
    g2.i[rt.G_WFMT] = 2
    (ga, _, _), (ca, _, _) = run_both([pre, g1, g2], weights, act, torch.zeros(4), {}, B)
    h_g = ga[B * hoff: B * ooff].view(torch.bfloat16).float()
    h_c = ca[B * hoff: B * ooff].view(torch.bfloat16).float()
    assert (h_g - h_c).abs().max() <= 2.0 ** -7 * max(h_c.abs().max().item(), 1.0)      # one bf16 ulp (rounding ties)
    out_g, out_c = ga[B * ooff:], ca[B * ooff:]
    assert (out_g - out_c).abs().max() < 3e-3 * max(out_c.abs().max().item(), 1.0)


@pytest.mark.parametrize("tile,w16", [(-1, 1), (0, 1), (1, 1), (2, 1), (-1, 0), (0, 0)])
@pytest.mark.parametrize("B,R,C", [(37, 8, 1024), (5, 32, 512), (130, 4, 64)])
def test_bf16_residual_stream_ops(B, R, C, tile, w16):
    """Round 6, the bf16 RESIDUAL STREAM of the plain-bf16 mode's transformer blocks: MDT_OP_PREP16 with a bf16 input (WFMT 2: LayerNorm of
    the bf16 stream, statistics in fp32 on the widened values) -> GEMM -> GEMM with A, residual and output all bf16, IN PLACE on the
    stream (MDT_G_WFMT 38), against the interpreter.  Shapes: the 1024- and 512-channel levels of configs[4] and a minimal one."""
    Hd = 2 * C
    w1 = rnd(Hd, C, seed=1, scale=C ** -0.5).to(torch.bfloat16).contiguous().view(-1).view(torch.float32)
    w2 = rnd(C, Hd, seed=2, scale=Hd ** -0.5).to(torch.bfloat16).contiguous().view(-1).view(torch.float32)
    weights = torch.cat([w1, w2, rnd(Hd, seed=3), rnd(C, seed=4), 1 + 0.1 * rnd(C, seed=6), 0.1 * rnd(C, seed=7)])
    o_w2 = w1.numel()
    o_b1 = o_w2 + w2.numel()
    o_b2, o_g, o_be = o_b1 + Hd, o_b1 + Hd + C, o_b1 + Hd + 2 * C
    # per-sample arena (floats): [x16 (R C / 2) | ln16 (R C / 2) | h16 (R Hd / 2)]
    xoff, lnoff, hoff = 0, R * C // 2, R * C
    per = hoff + R * Hd // 2
    act = torch.zeros(B * per)
    x16 = (rnd(B * R * C, seed=5) * 1.5 + 0.3).to(torch.bfloat16)
    act[: B * R * C // 2] = x16.view(-1).view(torch.float32)
    pre = rt.MdtOp()
    pre.kind = rt.OP_PREP16
    pre.a, pre.out, pre.p0, pre.p1 = ref(A, xoff), ref(A, lnoff), ref(W, o_g), ref(W, o_be)
    pre.i[rt.G_R_IN], pre.i[rt.G_LDA], pre.i[rt.G_CIN], pre.i[rt.G_PRO], pre.i[rt.G_WFMT] = R, C, C, rt.PRO_LAYERNORM, 2
    pre.f[0] = 1e-5
    g1 = gemm_op(a=ref(A, lnoff), w=ref(W, 0), bias=ref(W, o_b1), out=ref(A, hoff), r_out=R, r_in=R, lda=C, cin=C, taps=1,
                 n=Hd, ldc=Hd, o_rows=R, act=1)
    g1.i[rt.G_WFMT] = 6
    g2 = gemm_op(a=ref(A, hoff), w=ref(W, o_w2), bias=ref(W, o_b2), out=ref(A, xoff), res=ref(A, xoff), r_out=R, r_in=R,
                 lda=Hd, cin=Hd, taps=1, n=C, ldc=C, o_rows=R, ldr=C)
    g2.i[rt.G_WFMT] = 38
    # every tile of k_gemm_b16 (256 x 256 / 256 x 128 / 128 x 128; ragged M, N below the tile) with the all-bf16 epilogue
    # (w16 = 1: 8 columns per lane, the residual requested under the main loop) and with the generic one (w16 = 0)
    lib = rt.load_library()
    lib.mdt_set_tuning(b"tile16", int(tile))
    lib.mdt_set_tuning(b"w16", int(w16))
    try:
        (ga, _, _), (ca, _, _) = run_both([pre, g1, g2], weights, act, torch.zeros(4), {}, B)
    finally:
        lib.mdt_set_tuning(b"tile16", -1)
        lib.mdt_set_tuning(b"w16", 1)
    n_x, n_h = B * R * C // 2, B * R * Hd // 2
    ln_g, ln_c = ga[n_x: 2 * n_x].view(torch.bfloat16).float(), ca[n_x: 2 * n_x].view(torch.bfloat16).float()
    assert (ln_g - ln_c).abs().max() <= 2.0 ** -7 * max(ln_c.abs().max().item(), 1.0)    # one bf16 ulp (rounding ties)
    out_g, out_c = ga[:n_x].view(torch.bfloat16).float(), ca[:n_x].view(torch.bfloat16).float()
    assert torch.isfinite(out_g).all() and not torch.equal(out_c, x16.float())
    # the stream was updated in place: bf16(x + W2 gelu(W1 LN(x) + b1) + b2); a differently rounded hidden value moves a sum by
    # less than a bf16 ulp of the result or two
    assert (out_g - out_c).abs().max() <= 2.0 ** -6 * max(out_c.abs().max().item(), 1.0)
    # closed form of the three ops on the bf16 inputs
    xf = x16.float().view(B, R, C)
    wl1, wl2 = w1.view(torch.bfloat16).float().view(Hd, C), w2.view(torch.bfloat16).float().view(C, Hd)
    ln = torch.nn.functional.layer_norm(xf, (C,), weights[o_g: o_g + C], weights[o_be: o_be + C], 1e-5).to(torch.bfloat16).float()
    h = torch.nn.functional.gelu(ln @ wl1.T + weights[o_b1: o_b1 + Hd]).to(torch.bfloat16).float()
    want = (xf + h @ wl2.T + weights[o_b2: o_b2 + C]).to(torch.bfloat16).float()
    assert (out_g.view(B, R, C) - want).abs().max() <= 2.0 ** -5 * max(want.abs().max().item(), 1.0)


@pytest.mark.parametrize("tile", [-1, 0, 1, 2])
@pytest.mark.parametrize("B,R,C,N", [(37, 8, 1024, 1536), (5, 32, 512, 512), (130, 4, 256, 768), (67, 2, 64, 128)])
def test_layernorm_folded_into_bf16_gemm(B, R, C, N, tile):
    """Round 6, MDT_G_WFMT 134: the projection reads the RAW bf16 residual stream (here just written in place by a WFMT 38 GEMM), gathers
    each row's mean / rstd from the A fragments it multiplies and applies the LayerNorm to its accumulators (column sums of the
    gain-folded weights in p0): against the interpreter and against LayerNorm + matmul in closed form.  Ragged M, every tile of
    k_gemm_b16, one chunk (C = 64) to sixteen."""
    Hd = C
    w1 = rnd(C, Hd, seed=1, scale=Hd ** -0.5).to(torch.bfloat16)                      # to_out-like: [C][Hd]
    gam, bet = 1 + 0.2 * rnd(C, seed=2), 0.2 * rnd(C, seed=3)
    wq = rnd(N, C, seed=4, scale=C ** -0.5)
    w2 = (wq * gam.unsqueeze(0)).to(torch.bfloat16)                                    # gain folded, as the compiler packs it
    csum = w2.float().sum(dim=1)
    bias2 = wq @ bet + 0.1 * rnd(N, seed=5)
    weights = torch.cat([w1.contiguous().view(-1).view(torch.float32), w2.contiguous().view(-1).view(torch.float32), rnd(C, seed=6), csum, bias2])
    o_w2 = w1.numel() // 2
    o_b1 = o_w2 + w2.numel() // 2
    o_cs, o_b2 = o_b1 + C, o_b1 + C + N
    # per-sample arena (floats): [x16 (R C / 2) | a16 (R Hd / 2) | q16 (R N / 2)]
    xoff, aoff = 0, R * C // 2
    qoff = aoff + R * Hd // 2
    per = qoff + R * N // 2
    act = torch.zeros(B * per)
    x16 = (rnd(B * R * C, seed=7) * 1.5 + 0.7).to(torch.bfloat16)                     # a mean well away from 0: the cancellation case
    a16 = rnd(B * R * Hd, seed=8).to(torch.bfloat16)
    act[: B * R * C // 2] = x16.view(-1).view(torch.float32)
    act[B * aoff: B * qoff] = a16.view(-1).view(torch.float32)
    g1 = gemm_op(a=ref(A, aoff), w=ref(W, 0), bias=ref(W, o_b1), out=ref(A, xoff), res=ref(A, xoff), r_out=R, r_in=R,
                 lda=Hd, cin=Hd, taps=1, n=C, ldc=C, o_rows=R, ldr=C)
    g1.i[rt.G_WFMT] = 38
    g2 = gemm_op(a=ref(A, xoff), w=ref(W, o_w2), bias=ref(W, o_b2), out=ref(A, qoff), p0=ref(W, o_cs), r_out=R, r_in=R,
                 lda=C, cin=C, taps=1, n=N, ldc=N, o_rows=R)
    g2.i[rt.G_WFMT] = 134
    g2.f[0] = 1e-5
    lib = rt.load_library()
    lib.mdt_set_tuning(b"tile16", int(tile))
    try:
        (ga, _, _), (ca, _, _) = run_both([g1, g2], weights, act, torch.zeros(4), {}, B)
    finally:
        lib.mdt_set_tuning(b"tile16", -1)
    n_x = B * R * C // 2
    xg, xc = ga[:n_x].view(torch.bfloat16).float(), ca[:n_x].view(torch.bfloat16).float()
    assert (xg - xc).abs().max() <= 2.0 ** -7 * max(xc.abs().max().item(), 1.0)         # the new stream: one bf16 ulp (rounding ties)
    qg, qc = ga[B * qoff:].view(torch.bfloat16).float(), ca[B * qoff:].view(torch.bfloat16).float()
    scale = max(qc.abs().max().item(), 1.0)
    assert torch.isfinite(qg).all() and (qg - qc).abs().max() <= 2.0 ** -6 * scale
    # closed form on the GPU's own stream: LayerNorm (gain, bias) then the fp32 projection
    want = torch.nn.functional.layer_norm(xg.view(B * R, C), (C,), gam, bet, 1e-5) @ wq.T + 0.1 * rnd(N, seed=5)
    assert (qg.view(B * R, N) - want).abs().max() <= 3e-2 * scale                        # bf16 weights and output against fp32 ones


@pytest.mark.parametrize("w16", [1, 0])
@pytest.mark.parametrize("fold", [False, True])
def test_bf16_gemm_bf16_output_into_a_column_block(fold, w16):
    """k_gemm_b16 with a bf16 output written into a COLUMN BLOCK of wider rows (MDT_G_O_COL, LDC > N: how q | k | v or two
    projections can share a tensor), with GELU, through the all-bf16 epilogue and the generic one; the columns beside the block stay
    untouched.  fold: the same with the LayerNorm folded into the GEMM (MDT_G_WFMT 134; always the all-bf16 epilogue)."""
    if fold and not w16:
        pytest.skip("a folded LayerNorm lives in the all-bf16 epilogue only")
    B, R, C, N, LDC, OCOL = 41, 8, 256, 128, 320, 64
    gam, bet = 1 + 0.2 * rnd(C, seed=2), 0.2 * rnd(C, seed=3)
    wq = rnd(N, C, seed=4, scale=C ** -0.5)
    w2 = ((wq * gam.unsqueeze(0)) if fold else wq).to(torch.bfloat16)
    csum = w2.float().sum(dim=1)
    bias = (wq @ bet if fold else torch.zeros(N)) + 0.1 * rnd(N, seed=5)
    weights = torch.cat([w2.contiguous().view(-1).view(torch.float32), csum, bias])
    o_cs = w2.numel() // 2
    o_b = o_cs + N
    # per-sample arena (floats): [x16 (R C / 2) | out16 (R LDC / 2)]
    ooff = R * C // 2
    act = torch.zeros(B * (ooff + R * LDC // 2))
    x16 = (rnd(B * R * C, seed=7) * 1.2 + 0.4).to(torch.bfloat16)
    act[: B * R * C // 2] = x16.view(-1).view(torch.float32)
    marker16 = torch.full((B * R * LDC,), 3.0).to(torch.bfloat16)
    act[B * ooff:] = marker16.view(-1).view(torch.float32)
    g = gemm_op(a=ref(A, 0), w=ref(W, 0), bias=ref(W, o_b), out=ref(A, ooff), r_out=R, r_in=R, lda=C, cin=C, taps=1, n=N, ldc=LDC,
                o_rows=R, o_col=OCOL, act=1)
    g.i[rt.G_WFMT] = 6
    if fold:
        g.p0 = ref(W, o_cs)
        g.i[rt.G_WFMT] = 134
        g.f[0] = 1e-5
    lib = rt.load_library()
    lib.mdt_set_tuning(b"w16", int(w16))
    try:
        (ga, _, _), (ca, _, _) = run_both([g], weights, act, torch.zeros(4), {}, B)
    finally:
        lib.mdt_set_tuning(b"w16", 1)
    og = ga[B * ooff:].view(torch.bfloat16).float().view(B * R, LDC)
    oc = ca[B * ooff:].view(torch.bfloat16).float().view(B * R, LDC)
    assert torch.equal(og[:, :OCOL], torch.full((B * R, OCOL), 3.0)) and torch.equal(og[:, OCOL + N:], torch.full((B * R, LDC - OCOL - N), 3.0))
    assert (og - oc).abs().max() <= 2.0 ** -7 * max(oc.abs().max().item(), 1.0)
    xin = x16.float().view(B * R, C)
    pre = (torch.nn.functional.layer_norm(xin, (C,), gam, bet, 1e-5) @ wq.T + 0.1 * rnd(N, seed=5)) if fold else (xin @ w2.float().T + bias)
    want = torch.nn.functional.gelu(pre)
    assert (og[:, OCOL: OCOL + N] - want).abs().max() <= (3e-2 if fold else 2.0 ** -7) * max(want.abs().max().item(), 1.0)


@pytest.mark.parametrize("tile", [-1, 0, 1, 2])
def test_bf16_gemm_single_chunk_with_bf16_residual(tile):
    """k_gemm_b16's all-bf16 epilogue with K = 64: ONE chunk, so the residual is requested before the loop instead of two chunks
    before its end (the hand-counted vmcnt of that case); in place, ragged M, against the interpreter and the closed form."""
    B, R, C = 83, 4, 64
    w = rnd(C, C, seed=1, scale=C ** -0.5).to(torch.bfloat16).contiguous().view(-1).view(torch.float32)
    weights = torch.cat([w, rnd(C, seed=2)])
    # per-sample arena (floats): [x16 (R C / 2) | a16 (R C / 2)]
    act = torch.zeros(B * R * C)
    x16 = (rnd(B * R * C, seed=3) * 1.5 + 0.3).to(torch.bfloat16)
    a16 = rnd(B * R * C, seed=4).to(torch.bfloat16)
    act[: B * R * C // 2] = x16.view(-1).view(torch.float32)
    act[B * R * C // 2:] = a16.view(-1).view(torch.float32)
    g = gemm_op(a=ref(A, R * C // 2), w=ref(W, 0), bias=ref(W, w.numel()), out=ref(A, 0), res=ref(A, 0), r_out=R, r_in=R,
                lda=C, cin=C, taps=1, n=C, ldc=C, o_rows=R, ldr=C)
    g.i[rt.G_WFMT] = 38
    lib = rt.load_library()
    lib.mdt_set_tuning(b"tile16", int(tile))
    try:
        (ga, _, _), (ca, _, _) = run_both([g], weights, act, torch.zeros(4), {}, B)
    finally:
        lib.mdt_set_tuning(b"tile16", -1)
    n_x = B * R * C // 2
    out_g, out_c = ga[:n_x].view(torch.bfloat16).float(), ca[:n_x].view(torch.bfloat16).float()
    assert (out_g - out_c).abs().max() <= 2.0 ** -7 * max(out_c.abs().max().item(), 1.0)
    want = (x16.float().view(-1, C) + a16.float().view(-1, C) @ w.view(torch.bfloat16).float().view(C, C).T + weights[w.numel():])
    assert (out_g.view(-1, C) - want.to(torch.bfloat16).float()).abs().max() <= 2.0 ** -7 * max(want.abs().max().item(), 1.0)
    assert torch.equal(ga[n_x:], act[n_x:])                      # the A operand is untouched


@pytest.mark.parametrize("variant", [0, 2, 3])
@pytest.mark.parametrize("mode", [rt.TB_FF, rt.TB_SELF, rt.TB_CROSS])
@pytest.mark.parametrize("C,T,B", [(128, 16, 5), (256, 4, 37), (128, 4, 16), (256, 16, 3), (128, 1, 70)])
@pytest.mark.parametrize("products", ["bf16x3", "f32"])
def test_fused_transformer_sub_block(mode, C, T, B, variant, products):
    """MDT_OP_TBLOCK (k_tblock_lw: variant 0, C = 128; k_tblock32: variants 2 / 3, C = 256; both with split-bf16 and -- MDT_B_WF32,
    k_tblock32 since round 6 -- exact-fp32 products) against the interpreter: LayerNorm folding, tile packing, DMA ring, MFMA
    operand chaining."""
    from moleculediffusiontransformer_amd.compiler import Ten, UNetCompiler
    from moleculediffusiontransformer_amd.netspec import inverse_unet_config
    cfg = inverse_unet_config(16, 64, 128, 12)
    n_ctx, mid = 12, 512
    if variant == 0 and (C != 128 or (mode == rt.TB_CROSS and (16 // T) * n_ctx > 16)):
        pytest.skip("variant 0 (64-row workgroups, loader waves) serves C = 128, cross blocks with at most 16 keys per 16 rows")
    if variant >= 2 and (C != 256 or (mode == rt.TB_CROSS and (16 // T) * n_ctx > 48)):
        pytest.skip("variant 2 (32-row workgroups) serves C = 256, cross blocks with at most 48 keys per 16 rows")
    p = "blk."
    sd = {p + "norm.weight": 1 + 0.2 * rnd(C, seed=1), p + "norm.bias": 0.2 * rnd(C, seed=2),
          p + "norm_context.weight": 1 + 0.2 * rnd(C, seed=3), p + "norm_context.bias": 0.2 * rnd(C, seed=4),
          p + "to_q.weight": rnd(mid, C, seed=5, scale=C ** -0.5), p + "to_kv.weight": rnd(2 * mid, C, seed=6, scale=C ** -0.5),
          p + "attention.to_out.weight": rnd(C, mid, seed=7, scale=mid ** -0.5), p + "attention.to_out.bias": 0.1 * rnd(C, seed=8),
          p + "0.weight": rnd(2 * C, C, seed=9, scale=C ** -0.5), p + "0.bias": 0.1 * rnd(2 * C, seed=10),
          p + "2.weight": rnd(C, 2 * C, seed=11, scale=(2 * C) ** -0.5), p + "2.bias": 0.1 * rnd(C, seed=12)}
    comp = UNetCompiler(cfg, 64, n_ctx, sd, gemm_mode=products)
    t = Ten(A, 0, T, C)
    comp.tblock(t, mode, p, 0 if mode == rt.TB_CROSS else None, variant=variant)
    assert comp.ops[0].i[rt.B_WF32] == int(products == "f32")
    op = comp.ops[0]
    kv_off = T * C
    if mode == rt.TB_CROSS:
        op.a2 = ref(A, kv_off)
    act = torch.cat([rnd(B * T * C, seed=13) * 1.5 + 0.3, rnd(B * n_ctx * 2 * mid, seed=14)])
    n_kv = act.numel()
    if variant == 3:                                         # partial-sum scratch behind the K/V rows
        op.out = ref(A, T * C + n_ctx * 2 * mid)
        act = torch.cat([act, torch.zeros(B * 2 * T * C)])
    (ga, _, _), (ca, _, _) = run_both([op], comp.W.pack(), act, torch.zeros(4), {}, B)
    xg, xc = ga[: B * T * C], ca[: B * T * C]
    assert torch.isfinite(xg).all()
    assert (xg - xc).abs().max() < 1e-4 * max(1.0, xc.abs().max().item()), (xg - xc).abs().max().item()
    assert torch.equal(ga[B * T * C: n_kv], ca[B * T * C: n_kv])      # K/V untouched
    if mode == rt.TB_CROSS:                                  # batch-invariant context (guidance pass): stride 0
        op.i[rt.B_KV_BSTRIDE] = 0
        op.a2 = ref(S, 0)
        shr = rnd(n_ctx * 2 * mid, seed=15)
        (ga, _, _), (ca, _, _) = run_both([op], comp.W.pack(), act, shr, {}, B)
        assert (ga[: B * T * C] - ca[: B * T * C]).abs().max() < 1e-4 * max(1.0, ca.abs().max().item())


@pytest.mark.parametrize("B,R,C,G,film,silu,eps", [(5, 64, 64, 1, False, True, 1e-5), (3, 16, 128, 8, True, True, 1e-5),
                                                    (4, 4, 512, 8, True, True, 1e-5), (6, 16, 128, 32, False, False, 1e-6),
                                                    (2, 4, 256, 32, False, False, 1e-6), (3, 64, 16, 1, False, True, 1e-5),
                                                    # deep-UNet samples (16 K - 32 K elements): the 1024-thread form
                                                    (3, 32, 512, 8, True, True, 1e-5), (2, 32, 1024, 8, False, True, 1e-5),
                                                    (2, 128, 256, 8, True, True, 1e-5), (3, 32, 512, 32, False, False, 1e-6),
                                                    (2, 8, 2048, 8, False, True, 1e-5)])
def test_gn_act(B, R, C, G, film, silu, eps):
    """k_gn_act (statistics + normalise + FiLM + SiLU in one pass) against torch's GroupNorm."""
    weights = torch.cat([1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)])
    shr = 0.3 * rnd(2 * C, seed=9)
    act = torch.cat([rnd(B * R * C, seed=4) * 1.5 + 0.3, torch.zeros(B * R * C)])
    op = rt.MdtOp()
    op.kind = rt.OP_GN_ACT
    op.a, op.out, op.p0, op.p1 = ref(A, 0), ref(A, R * C), ref(W, 0), ref(W, C)
    if film:
        op.p3 = ref(S, 0)
    op.i[rt.N_ROWS], op.i[rt.N_LD], op.i[rt.N_GROUPS], op.i[rt.N_GSIZE], op.i[rt.N_SILU] = R, C, G, C // G, int(silu)
    op.f[0] = eps
    (ga, _, _), (ca, _, _) = run_both([op], weights, act, shr, {}, B)
    assert (ga - ca).abs().max() < 2e-5
    x = act[: B * R * C].view(B, R, C).transpose(1, 2)
    h = torch.nn.functional.group_norm(x, G, weights[:C], weights[C:], eps)
    if film:
        h = h * (shr[:C].view(1, C, 1) + 1) + shr[C:].view(1, C, 1)
    if silu:
        h = torch.nn.functional.silu(h)
    assert (ga[B * R * C:].view(B, R, C) - h.transpose(1, 2)).abs().max() < 2e-5


@pytest.mark.parametrize("B,R,C,G", [(5, 32, 512, 8), (37, 8, 1024, 8), (9, 64, 256, 8), (3, 4, 64, 4), (2, 16, 1024, 8)])
def test_gn_act_on_two_sources_with_raw_copy(B, R, C, G):
    """MDT_OP_GN_ACT, round 6: the input is cat([a, SCALE2 * a2]) read from its two sources (the up path's ResnetBlock1d, plain-bf16
    mode), the normalised + SiLU output is bf16 and a raw bf16 copy of the input goes to p2 -- every kernel form (one workgroup per
    (sample, group) / per sample, 256 / 1024 threads) against the interpreter and torch's GroupNorm on the concatenated tensor."""
    ld = 2 * C
    weights = torch.cat([1 + 0.1 * rnd(ld, seed=2), 0.1 * rnd(ld, seed=3)])
    # per-sample arena (floats): [a (R C) | a2 (R C) | y16 (R ld / 2) | raw16 (R ld / 2)]
    aoff, boff, yoff = 0, R * C, 2 * R * C
    roff = yoff + R * ld // 2
    act = torch.zeros(B * (roff + R * ld // 2))
    xa, xb = rnd(B * R * C, seed=4) * 1.5 + 0.3, rnd(B * R * C, seed=5) * 0.8 - 0.2
    act[: B * R * C], act[B * boff: B * yoff] = xa, xb
    op = rt.MdtOp()
    op.kind = rt.OP_GN_ACT
    op.a, op.a2, op.out, op.p0, op.p1, op.p2 = ref(A, aoff), ref(A, boff), ref(A, yoff), ref(W, 0), ref(W, ld), ref(A, roff)
    i = op.i
    i[rt.N_ROWS], i[rt.N_LD], i[rt.N_GROUPS], i[rt.N_GSIZE], i[rt.N_SILU], i[rt.N_OUT16], i[rt.N_CA] = R, ld, G, ld // G, 1, 1, C
    op.f[0], op.f[1] = 1e-5, 0.7071
    (ga, _, _), (ca, _, _) = run_both([op], weights, act, torch.zeros(4), {}, B)
    assert torch.equal(ga[: B * yoff], act[: B * yoff])                                   # the sources are untouched
    yg, yc = ga[B * yoff: B * roff].view(torch.bfloat16).float(), ca[B * yoff: B * roff].view(torch.bfloat16).float()
    assert (yg - yc).abs().max() <= 2.0 ** -7 * max(yc.abs().max().item(), 1.0)          # one bf16 ulp (rounding ties)
    cat = torch.cat([xa.view(B, R, C), 0.7071 * xb.view(B, R, C)], dim=2)
    rg = ga[B * roff:].view(torch.bfloat16)
    assert torch.equal(rg.view(B, R, ld), cat.to(torch.bfloat16))                          # the raw copy: bit-exact bf16 of the input
    want = torch.nn.functional.silu(torch.nn.functional.group_norm(cat.transpose(1, 2), G, weights[:ld], weights[ld:], 1e-5)).transpose(1, 2)
    assert (yg.view(B, R, ld) - want).abs().max() <= 2.0 ** -7 * max(want.abs().max().item(), 1.0) + 2e-5


@pytest.mark.parametrize("C,T,B,taps,gsize,silu,in_scale2", [
    (128, 16, 70, 3, 32, True, 0.7071),    # ResnetBlock1d block1 on cat([x, skip / sqrt 2]) at the 128-channel level
    (256, 4, 37, 3, 64, True, 0.7071),
    (256, 4, 9, 1, 0, False, 0.7071),      # its 1x1 residual convolution on the raw concatenation
    (128, 16, 5, 1, 0, False, 0.5),
])
def test_row_stationary_conv_two_sources(C, T, B, taps, gsize, silu, in_scale2, prod):
    """k_rconv on a concatenated input that is never materialised: prologue + taps for source a, then for source b,
    into the same accumulators (second exchange barrier, fragment-set rotation between the sources)."""
    from moleculediffusiontransformer_amd.compiler import Ten, UNetCompiler
    from moleculediffusiontransformer_amd.netspec import inverse_unet_config
    comp = UNetCompiler(inverse_unet_config(16, 64, 128, 12), 64, 12, {})
    w = rnd(C, 2 * C, taps, seed=1, scale=(2 * C * taps) ** -0.5)
    gb = torch.cat([1 + 0.1 * rnd(2 * C, seed=2), 0.1 * rnd(2 * C, seed=3), 0.1 * rnd(C, seed=5)])   # gain | beta | conv bias
    g_off = comp.W.add("gb", gb)
    xa, out, xb = Ten(A, 0, T, C), Ten(A, T * C, T, C), Ten(A, 2 * T * C, T, C)
    comp.rconv(xa, w, "w", out, taps=taps, bias_off=g_off + 4 * C,
               gn=(g_off, g_off + 2 * C, gsize, 1e-5, silu) if gsize else None, x2=xb, in_scale2=in_scale2)
    op = comp.ops[0]
    act = torch.cat([rnd(B * T * C, seed=4) * 1.5 + 0.3, rnd(B * T * C, seed=6), rnd(B * T * C, seed=7) * 0.8 - 0.2])
    (ga, _, _), (ca, _, _) = run_both([op], comp.W.pack(), act, torch.zeros(4), {}, B)
    og, oc = ga[B * T * C: 2 * B * T * C], ca[B * T * C: 2 * B * T * C]
    scale = max(1.0, oc.abs().max().item())
    assert torch.isfinite(og).all() and (og - oc).abs().max() < 1e-4 * scale, (og - oc).abs().max().item()
    h = torch.cat([act[: B * T * C].view(B, T, C), in_scale2 * act[2 * B * T * C:].view(B, T, C)], dim=2).transpose(1, 2)
    if gsize:
        h = torch.nn.functional.group_norm(h, 2 * C // gsize, gb[: 2 * C], gb[2 * C: 4 * C], 1e-5)
        if silu:
            h = torch.nn.functional.silu(h)
    y = torch.nn.functional.conv1d(h, w, gb[4 * C:], padding=taps // 2).transpose(1, 2)
    assert (og.view(B, T, C) - y).abs().max() < 1e-4 * scale


@pytest.mark.parametrize("C,T,B,taps,gsize,film,silu,res,in_scale", [
    (128, 16, 5, 3, 16, True, True, "other", 1.0),     # ResnetBlock1d block2 at the 128-channel level (FiLM, residual)
    (128, 16, 70, 3, 32, False, True, None, 0.7071),   # half of a concatenated block1 input (skip scaling)
    (256, 4, 37, 3, 32, True, True, "other", 1.0),
    (256, 4, 9, 3, 64, False, True, "accumulate", 0.7071),   # second half: accumulates into the first half's output
    (256, 4, 16, 1, 8, False, False, None, 1.0),       # Transformer1d to_in: GroupNorm(32 groups, eps 1e-6) + 1x1 conv
    (128, 16, 3, 1, 4, False, False, None, 1.0),
    (128, 8, 6, 1, 0, False, False, "accumulate", 1.0),      # plain 1x1 conv (to_out), 8 tokens per sample
    (256, 1, 33, 3, 32, False, True, None, 1.0),       # one token per sample: both neighbours are padding
])
def test_row_stationary_conv(C, T, B, taps, gsize, film, silu, res, in_scale, prod):
    """k_rconv (GroupNorm + FiLM + SiLU prologue, DPP-shifted taps, loader-wave weight ring) against the interpreter
    and against torch's group_norm / conv1d."""
    from moleculediffusiontransformer_amd.compiler import Ten, UNetCompiler
    from moleculediffusiontransformer_amd.netspec import inverse_unet_config
    comp = UNetCompiler(inverse_unet_config(16, 64, 128, 12), 64, 12, {})
    w = rnd(C, C, taps, seed=1, scale=(C * taps) ** -0.5)
    gb = torch.cat([1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3), 0.1 * rnd(C, seed=5)])   # gain | beta | conv bias
    g_off = comp.W.add("gb", gb)
    x, out, other = Ten(A, 0, T, C), Ten(A, T * C, T, C), Ten(A, 2 * T * C, T, C)
    eps = 1e-6 if taps == 1 else 1e-5
    comp.rconv(x, w, "w", out, taps=taps, bias_off=None if res == "accumulate" else g_off + 2 * C,
               res={"other": other, "accumulate": out, None: None}[res],
               gn=(g_off, g_off + C, gsize, eps, silu) if gsize else None, in_scale=in_scale)
    op = comp.ops[0]
    if film:
        op.p3 = ref(S, 0)
    shr = 0.3 * rnd(2 * C, seed=9)
    act = torch.cat([rnd(B * T * C, seed=4) * 1.5 + 0.3, rnd(B * T * C, seed=6), rnd(B * T * C, seed=7)])
    (ga, _, _), (ca, _, _) = run_both([op], comp.W.pack(), act, shr, {}, B)
    og, oc = ga[B * T * C: 2 * B * T * C], ca[B * T * C: 2 * B * T * C]
    scale = max(1.0, oc.abs().max().item())
    assert torch.isfinite(og).all() and (og - oc).abs().max() < 1e-4 * scale, (og - oc).abs().max().item()
    assert torch.equal(ga[: B * T * C], ca[: B * T * C]) and torch.equal(ga[2 * B * T * C:], ca[2 * B * T * C:])
    # independent closed form
    h = (act[: B * T * C].view(B, T, C) * in_scale).transpose(1, 2)
    if gsize:
        h = torch.nn.functional.group_norm(h, C // gsize, gb[:C], gb[C: 2 * C], eps)
        if film:
            h = h * (shr[:C].view(1, C, 1) + 1) + shr[C:].view(1, C, 1)
        if silu:
            h = torch.nn.functional.silu(h)
    y = torch.nn.functional.conv1d(h, w, None if res == "accumulate" else gb[2 * C:], padding=taps // 2).transpose(1, 2)
    if res == "other":
        y = y + act[2 * B * T * C:].view(B, T, C)
    elif res == "accumulate":
        y = y + act[B * T * C: 2 * B * T * C].view(B, T, C)
    assert (og.view(B, T, C) - y).abs().max() < 1e-4 * scale


def _resblock_case(cin, cout, film):
    """One MDT_OP_RESBLOCK op with seeded weights + the closed form of the reference block (modules.py:145-205)."""
    from moleculediffusiontransformer_amd.compiler import Ten, UNetCompiler
    from moleculediffusiontransformer_amd.netspec import inverse_unet_config
    p = "blk."
    sd = {p + "block1.groupnorm.weight": 1 + 0.1 * rnd(cin, seed=1), p + "block1.groupnorm.bias": 0.1 * rnd(cin, seed=2),
          p + "block1.project.weight": rnd(cout, cin, 3, seed=3, scale=(3 * cin) ** -0.5),
          p + "block1.project.bias": 0.1 * rnd(cout, seed=4),
          p + "block2.groupnorm.weight": 1 + 0.1 * rnd(cout, seed=5), p + "block2.groupnorm.bias": 0.1 * rnd(cout, seed=6),
          p + "block2.project.weight": rnd(cout, cout, 3, seed=7, scale=(3 * cout) ** -0.5),
          p + "block2.project.bias": 0.1 * rnd(cout, seed=8),
          p + "to_out.weight": rnd(cout, cin, 1, seed=9, scale=cin ** -0.5), p + "to_out.bias": 0.1 * rnd(cout, seed=10)}
    comp = UNetCompiler(inverse_unet_config(16, 64, 128, 12), 64, 12, sd)
    cin_p, cout_p = (cin + 15) // 16 * 16, (cout + 15) // 16 * 16        # the op works on channel counts padded to 16
    x = Ten(A, 0, 64, cin_p, cin)
    assert comp.resblock_ok(64, cin, cout, 1, p)
    y = comp.resnet(x, p, cin, cout, 1, free_input=False)
    assert len(comp.ops) == 1 and comp.ops[0].kind == rt.OP_RESBLOCK
    op = comp.ops[0]
    op.out = ref(A, 64 * cin_p)
    op.p3 = ref(S, 0) if film else ref(0, 0)
    shr = torch.zeros(2 * cout_p)                      # [scale(cout_p) | shift(cout_p)], zero on the padding
    shr[:cout], shr[cout_p: cout_p + cout] = 0.3 * rnd(cout, seed=11), 0.3 * rnd(cout, seed=12)

    def closed_form(xin):                              # xin [B, 64, cin]
        F = torch.nn.functional
        xt = xin.transpose(1, 2)
        h = F.conv1d(F.silu(F.group_norm(xt, 1, sd[p + "block1.groupnorm.weight"], sd[p + "block1.groupnorm.bias"], 1e-5)),
                     sd[p + "block1.project.weight"], sd[p + "block1.project.bias"], padding=1)
        h = F.group_norm(h, 1, sd[p + "block2.groupnorm.weight"], sd[p + "block2.groupnorm.bias"], 1e-5)
        if film:
            h = h * (shr[:cout].view(1, cout, 1) + 1) + shr[cout_p: cout_p + cout].view(1, cout, 1)
        yt = F.conv1d(F.silu(h), sd[p + "block2.project.weight"], sd[p + "block2.project.bias"], padding=1)
        return (yt + F.conv1d(xt, sd[p + "to_out.weight"], sd[p + "to_out.bias"])).transpose(1, 2)
    return comp, op, shr, closed_form


@pytest.mark.parametrize("cin,cout", [(16, 64), (64, 16), (16, 16), (2, 16), (16, 1), (8, 16), (16, 8), (5, 50)])
@pytest.mark.parametrize("B,film", [(1, True), (2, False), (7, True), (1030, True)])
def test_fused_resnet_block(cin, cout, B, film, prod):
    """k_resblock (the Patcher / Unpatcher ResnetBlock1d in one launch; odd batches leave half a workgroup idle, 1030
    samples wrap the persistent loop) against the interpreter and against torch's group_norm / conv1d.  Channel counts that
    are not multiples of 16 (QMDiffusionForward: 2 -> 16 and 16 -> 1; AnalogDiffusionFull: 8 -> 16, 16 -> 8) run padded, with
    the GroupNorm statistics over the real channels only and exact zeros on the padding of the output."""
    comp, op, shr, closed_form = _resblock_case(cin, cout, film)
    cin_p, cout_p = (cin + 15) // 16 * 16, (cout + 15) // 16 * 16
    n_in, n_out = B * 64 * cin_p, B * 64 * cout_p
    xin = torch.zeros(B, 64, cin_p)
    xin[:, :, :cin] = (rnd(B * 64 * cin, seed=12) * 1.5 + 0.3).view(B, 64, cin)
    act = torch.cat([xin.view(-1), torch.full((n_out,), 7.0)])
    (ga, _, _), (ca, _, _) = run_both([op], comp.W.pack(), act, shr, {}, B)
    og, oc = ga[n_in:], ca[n_in:]
    scale = max(1.0, oc.abs().max().item())
    assert torch.isfinite(og).all() and (og - oc).abs().max() < 1e-4 * scale, (og - oc).abs().max().item()
    assert torch.equal(ga[:n_in], ca[:n_in])
    y = closed_form(xin[:, :, :cin])
    assert (og.view(B, 64, cout_p)[:, :, :cout] - y).abs().max() < 1e-4 * scale
    assert (og.view(B, 64, cout_p)[:, :, cout:] == 0).all()


@pytest.mark.parametrize("cin,cout,pin,pout", [(16, 64, 1, 4), (64, 16, 4, 1), (16, 16, 2, 2), (2, 16, 1, 2), (16, 1, 2, 1), (8, 16, 1, 4),
                                               (16, 8, 4, 1)])
@pytest.mark.parametrize("B", [1, 7, 1030])
def test_fused_resnet_block_with_folded_patch_rearranges(cin, cout, pin, pout, B, prod):
    """MDT_K_PATCH_IN / MDT_K_PATCH_OUT: the Unpatcher's / Patcher's `b (c p) l <-> b c (l p)` rearranges
    (modules.py:208-257) as the load / store pattern of k_resblock, op by op against the interpreter AND against the closed form
    with the rearrange applied by torch: odd B (half a workgroup idle, the B - 1 clamp of the two-samples-per-workgroup tail),
    1030 samples (the persistent loop wraps), p in {2, 4}, padded channel counts together with patching."""
    comp, op, shr, closed_form = _resblock_case(cin, cout, True)
    op.i[rt.K_PATCH_IN], op.i[rt.K_PATCH_OUT] = pin, pout
    cin_p, cout_p = (cin + 15) // 16 * 16, (cout + 15) // 16 * 16
    n_in, n_out = B * 64 * cin_p, B * 64 * cout_p
    x = torch.zeros(B, 64, cin_p)                        # the block's input in plain [T][C] form
    x[:, :, :cin] = (rnd(B * 64 * cin, seed=12) * 1.5 + 0.3).view(B, 64, cin)
    # PATCH_IN: a[l][c p + q] = x[l p + q][c]
    a = x.view(B, 64 // pin, pin, cin_p).permute(0, 1, 3, 2).reshape(B, 64, cin_p) if pin > 1 else x
    act = torch.cat([a.reshape(-1), torch.full((n_out,), 7.0)])
    (ga, _, _), (ca, _, _) = run_both([op], comp.W.pack(), act, shr, {}, B)
    og, oc = ga[n_in:], ca[n_in:]
    scale = max(1.0, oc.abs().max().item())
    assert torch.isfinite(og).all() and (og - oc).abs().max() < 1e-4 * scale, (og - oc).abs().max().item()
    assert torch.equal(ga[:n_in], ca[:n_in])
    y = closed_form(x[:, :, :cin])                       # [B, 64, cout]
    o = og.view(B, 64 // pout, cout_p, pout).permute(0, 1, 3, 2).reshape(B, 64, cout_p) if pout > 1 else og.view(B, 64, cout_p)
    assert (o[:, :, :cout] - y).abs().max() < 1e-4 * scale
    assert (o[:, :, cout:] == 0).all()


@pytest.mark.parametrize("mode,split,with_pin", [(rt.TB_SELF, True, False), (rt.TB_SELF, True, True), (rt.TB_CROSS, True, True),
                                                 (rt.TB_FF, False, True), (rt.TB_FF, False, False)])
@pytest.mark.parametrize("T,B", [(4, 37), (16, 3)])
def test_chained_split_sub_block(mode, split, with_pin, T, B, prod):
    """MDT_OP_TBLOCK variant 4 (k_tblock32): block input = x + p_in, head group 0 writes x_out = input + its partial
    + bias, head group 1 leaves its bare partial in p_out; no reduce launch, x itself is not touched."""
    from moleculediffusiontransformer_amd.compiler import Ten, UNetCompiler
    from moleculediffusiontransformer_amd.netspec import inverse_unet_config
    C, n_ctx, mid = 256, 12, 512
    if mode == rt.TB_CROSS and (16 // T) * n_ctx > 48:
        pytest.skip("more than 48 keys per 16 rows")
    p = "blk."
    sd = {p + "norm.weight": 1 + 0.2 * rnd(C, seed=1), p + "norm.bias": 0.2 * rnd(C, seed=2),
          p + "norm_context.weight": 1 + 0.2 * rnd(C, seed=3), p + "norm_context.bias": 0.2 * rnd(C, seed=4),
          p + "to_q.weight": rnd(mid, C, seed=5, scale=C ** -0.5), p + "to_kv.weight": rnd(2 * mid, C, seed=6, scale=C ** -0.5),
          p + "attention.to_out.weight": rnd(C, mid, seed=7, scale=mid ** -0.5), p + "attention.to_out.bias": 0.1 * rnd(C, seed=8),
          p + "0.weight": rnd(2 * C, C, seed=9, scale=C ** -0.5), p + "0.bias": 0.1 * rnd(2 * C, seed=10),
          p + "2.weight": rnd(C, 2 * C, seed=11, scale=(2 * C) ** -0.5), p + "2.bias": 0.1 * rnd(C, seed=12)}
    comp = UNetCompiler(inverse_unet_config(16, 64, 128, 12), 64, n_ctx, sd)
    n_x, n_kv = T * C, n_ctx * 2 * mid
    x, x_out, p_in, p_out = Ten(A, 0, T, C), Ten(A, n_x + n_kv, T, C), Ten(A, 2 * n_x + n_kv, T, C), Ten(A, 3 * n_x + n_kv, T, C)
    comp.tblock(x, mode, p, 0 if mode == rt.TB_CROSS else None, variant=4, x_out=x_out,
                p_in=p_in if with_pin else None, p_out=p_out if split else None)
    op = comp.ops[0]
    if mode == rt.TB_CROSS:
        op.a2 = ref(A, n_x)
    act = torch.cat([rnd(B * n_x, seed=13) * 1.5 + 0.3, rnd(B * n_kv, seed=14), torch.zeros(B * n_x),
                     0.5 * rnd(B * n_x, seed=15), torch.zeros(B * n_x)])
    (ga, _, _), (ca, _, _) = run_both([op], comp.W.pack(), act, torch.zeros(4), {}, B)
    lo = B * (n_x + n_kv)
    og, oc = ga[lo: lo + B * n_x], ca[lo: lo + B * n_x]
    assert torch.isfinite(og).all() and (og - oc).abs().max() < 1e-4 * max(1.0, oc.abs().max().item())
    assert not torch.equal(oc, torch.zeros_like(oc))
    if split:
        pg, pc = ga[lo + 2 * B * n_x:], ca[lo + 2 * B * n_x:]
        assert (pg - pc).abs().max() < 1e-4 * max(1.0, pc.abs().max().item()) and pc.abs().max() > 0
    assert torch.equal(ga[: lo], ca[: lo])              # x and K/V untouched
    assert torch.equal(ga[lo + B * n_x: lo + 2 * B * n_x], ca[lo + B * n_x: lo + 2 * B * n_x])   # p_in untouched


def _transformer_sd(p, C, layers, cross, ctx=128, mid=512, seed0=100):
    """Random Transformer1d parameters with the reference's key names (modules.py:469-524)."""
    k = [seed0]

    def r(*shape, scale=1.0):
        k[0] += 1
        return rnd(*shape, seed=k[0], scale=scale)
    sd = {p + "to_in.0.weight": 1 + 0.2 * r(C), p + "to_in.0.bias": 0.2 * r(C),
          p + "to_in.1.weight": r(C, C, 1, scale=C ** -0.5), p + "to_in.1.bias": 0.1 * r(C),
          p + "to_out.1.weight": r(C, C, 1, scale=C ** -0.5), p + "to_out.1.bias": 0.1 * r(C)}
    for li in range(layers):
        for name, cf in (("attention.", C),) + ((("cross_attention.", ctx),) if cross else ()):
            q = p + f"blocks.{li}." + name
            sd.update({q + "norm.weight": 1 + 0.2 * r(C), q + "norm.bias": 0.2 * r(C),
                       q + "norm_context.weight": 1 + 0.2 * r(cf), q + "norm_context.bias": 0.2 * r(cf),
                       q + "to_q.weight": r(mid, C, scale=C ** -0.5), q + "to_kv.weight": r(2 * mid, cf, scale=cf ** -0.5),
                       q + "attention.to_out.weight": r(C, mid, scale=mid ** -0.5), q + "attention.to_out.bias": 0.1 * r(C)})
        q = p + f"blocks.{li}.feed_forward."
        sd.update({q + "0.weight": r(2 * C, C, scale=C ** -0.5), q + "0.bias": 0.1 * r(2 * C),
                   q + "2.weight": r(C, 2 * C, scale=(2 * C) ** -0.5), q + "2.bias": 0.1 * r(C)})
    return sd


def _handoff_ext(B, T):
    """bindings.ext[3] / [4] of a pair-split MDT_OP_TF256: zeroed flag words, hand-off blocks (engine.py sizes them alike)."""
    nrb = (B * T + 31) // 32
    return {3: torch.zeros(64 + 64 * nrb), 4: torch.zeros(2 * nrb * 2 * 32 * 256)}


@pytest.mark.parametrize("form", ["whole", "pair8", pytest.param("pair1", marks=pytest.mark.slow)])    # (pair1 at model level: the pair-stride switch test)
@pytest.mark.parametrize("C,T,B,layers,cross,fixed", [
    (128, 16, 5, 2, False, False), (128, 16, 70, 4, True, False), (128, 4, 16, 2, False, False), (128, 16, 3, 2, True, True),
    (128, 8, 9, 1, False, False), (128, 2, 33, 1, False, False), (128, 16, 1030, 1, True, False),
    (256, 4, 37, 2, True, False), (256, 4, 5, 2, False, False), (256, 4, 9, 4, True, True), (256, 16, 3, 1, False, False),
    (256, 8, 11, 1, True, False), (256, 1, 70, 1, True, False), (256, 4, 1030, 1, True, False)])
def test_fused_transformer(C, T, B, layers, cross, fixed, form, prod):
    """MDT_OP_TF128 / MDT_OP_TF256 (k_tf128.hip, k_tf256.hip): a whole Transformer1d in one launch, against (i) the CPU
    interpreter of the op (tile order, K-column permutation to the accumulator layout, vector layout) and (ii) the
    reference's module arithmetic written out with torch ops (modules.py:469-524, :401-410, :350-364, :314-320).
    form: 'whole' = one workgroup per 32-row block; 'pair8' / 'pair1' = the pair-split form of the 256-channel level (two
    descriptor tables, hand-off between the two workgroups inside the launch) with the partners 8 workgroup ids apart (one XCD
    under the observed placement) or neighbours (different XCDs): the result must not depend on it."""
    from moleculediffusiontransformer_amd.compiler import Ten, UNetCompiler
    from moleculediffusiontransformer_amd.netspec import inverse_unet_config
    import torch.nn.functional as F
    if C == 128 and form != "whole":
        pytest.skip("the pair split exists for the 256-channel level only")
    n_ctx, mid, H = 12, 512, 8
    cfg = inverse_unet_config(16, 64, 128, n_ctx)
    p = "tf."
    sd = _transformer_sd(p, C, layers, cross)
    comp = UNetCompiler(cfg, 64, n_ctx, sd, tf256=(form == "whole"))
    comp.pair_stride = 1 if form == "pair1" else 8
    if not (comp.tf128_ok(C, T, layers, cross) or comp.tf256_ok(C, T, layers, cross)):
        pytest.skip("shape outside the fused transformers' envelope")
    x = Ten(A, 0, T, C)
    y = comp.transformer(x, p, C, layers, cross, free_input=False)
    assert [o.kind for o in comp.ops] == [rt.OP_TF128 if C == 128 else rt.OP_TF256]
    op = comp.ops[0]
    assert op.i[rt.F_WF32] == int(prod == "f32")
    assert op.i[rt.F_NSPLIT] == (2 if (C == 256 and form != "whole") else (1 if C == 256 else 0))
    ext = _handoff_ext(B, T) if form != "whole" else {}
    kv_floats = n_ctx * 2 * mid
    act_x = rnd(B * T * C, seed=13) * 1.5 + 0.3
    kv_all = rnd(layers * B * kv_floats, seed=14) if cross else torch.zeros(0)
    shr = rnd(layers * kv_floats, seed=15) if cross else torch.zeros(4)
    # per-sample arena: [x | y | K/V layer 0 | K/V layer 1 ...]
    op.out = ref(A, T * C)
    if cross:
        if fixed:
            op.a2 = ref(S, 0)
            op.i[rt.F_KV_BSTRIDE] = 0
        else:
            op.a2 = ref(A, 2 * T * C)
    act = torch.cat([act_x, torch.zeros(B * T * C), kv_all])
    (ga, _, ge), (ca, _, _) = run_both([op], comp.W.pack(), act, shr, ext, B)
    yg, yc = ga[B * T * C: 2 * B * T * C].view(B, T, C), ca[B * T * C: 2 * B * T * C].view(B, T, C)
    assert torch.isfinite(yg).all()
    if ext:
        flags = ge[3].view(torch.int32)
        nsub = 1 + layers * (3 if cross else 2)
        assert int(flags[0]) == 0, "a hand-off poll timed out"
        assert bool((flags[64::32] == nsub).all()), "every (row block, half) counts one hand-off per sub-block"
    tol = 2e-4 * max(1.0, yc.abs().max().item())
    assert (yg - yc).abs().max() < tol, (yg - yc).abs().max().item()
    assert torch.equal(ga[: B * T * C], act_x) and torch.equal(ga[2 * B * T * C:], kv_all)     # inputs untouched

    # (ii) the module arithmetic, independent of the packing
    def attn(q_, x_, ctx, kv=None):
        xn = F.layer_norm(x_, (C,), sd[q_ + "norm.weight"], sd[q_ + "norm.bias"], 1e-5)
        qq = (xn @ sd[q_ + "to_q.weight"].T).view(B, T, H, 64).transpose(1, 2)
        if kv is None:
            cn = F.layer_norm(ctx, (ctx.shape[-1],), sd[q_ + "norm_context.weight"], sd[q_ + "norm_context.bias"], 1e-5)
            kv = cn @ sd[q_ + "to_kv.weight"].T
        k_, v_ = kv.chunk(2, dim=-1)
        k_ = k_.reshape(B, -1, H, 64).transpose(1, 2)
        v_ = v_.reshape(B, -1, H, 64).transpose(1, 2)
        o = ((qq @ k_.transpose(-1, -2)) * 0.125).softmax(-1) @ v_
        return o.transpose(1, 2).reshape(B, T, mid) @ sd[q_ + "attention.to_out.weight"].T + sd[q_ + "attention.to_out.bias"]
    xt = act_x.view(B, T, C)
    h = F.group_norm(xt.transpose(1, 2), 32, sd[p + "to_in.0.weight"], sd[p + "to_in.0.bias"], 1e-6)
    h = F.conv1d(h, sd[p + "to_in.1.weight"], sd[p + "to_in.1.bias"]).transpose(1, 2)
    for li in range(layers):
        bp = p + f"blocks.{li}."
        h = h + attn(bp + "attention.", h, h)
        if cross:
            if fixed:
                kv = shr[li * kv_floats: (li + 1) * kv_floats].view(1, n_ctx, 2 * mid).expand(B, -1, -1)
            else:
                kv = kv_all[li * B * kv_floats: (li + 1) * B * kv_floats].view(B, n_ctx, 2 * mid)
            h = h + attn(bp + "cross_attention.", h, None, kv)
        f_ = bp + "feed_forward."
        h = h + F.gelu(h @ sd[f_ + "0.weight"].T + sd[f_ + "0.bias"]) @ sd[f_ + "2.weight"].T + sd[f_ + "2.bias"]
    want = F.conv1d(h.transpose(1, 2), sd[p + "to_out.1.weight"], sd[p + "to_out.1.bias"]).transpose(1, 2)
    assert (yg - want).abs().max() < tol, (yg - want).abs().max().item()
    # same launch again from the same buffers: the ring protocol has no race that a second run would expose differently
    (ga2, _, _), _ = run_both([op], comp.W.pack(), act, shr, ext, B)
    assert torch.equal(ga2, ga)


@pytest.mark.parametrize("T,B,layers,cross", [(4, 1024, 2, True), (4, 37, 1, True), (16, 9, 1, False)])
def test_pair_handoff_is_placement_independent_and_repeatable(T, B, layers, cross, prod):
    """The pair-split MDT_OP_TF256 hands 32 x 256 partial sums between two workgroups inside the launch (sc1 stores, drained,
    workgroup barrier, flag; poll, barrier, sc1 loads).  120 launches on ONE set of flag words (they
    count monotonically across launches), partners alternately on one XCD (ids 8 apart) and on different XCDs (neighbouring ids,
    forced through mdt_set_tuning): every launch returns the same bits, which agree with the CPU interpreter and, to rounding
    (different summation order of the heads), with the whole-workgroup form.  NOTE what this test CANNOT see: every launch
    computes the same values, so a piece served from another launch's block looks right -- the failure of the first form of the
    hand-off was found at model level (tests/test_gpu_parity.py::test_repeated_sampling_is_bitwise_stable)."""
    from moleculediffusiontransformer_amd.compiler import Ten, UNetCompiler
    from moleculediffusiontransformer_amd.netspec import inverse_unet_config
    from oracle.program_interp import Buffers, run_program
    n_ctx, mid, C = 12, 512, 256
    cfg = inverse_unet_config(16, 64, 128, n_ctx)
    p = "tf."
    sd = _transformer_sd(p, C, layers, cross)
    lib = rt.load_library()

    def build(whole):
        comp = UNetCompiler(cfg, 64, n_ctx, sd, tf256=whole)
        comp.transformer(Ten(A, 0, T, C), p, C, layers, cross, free_input=False)
        op = comp.ops[0]
        op.out = ref(A, T * C)
        if cross:
            op.a2 = ref(A, 2 * T * C)
        return op, comp.W.pack()
    kv_floats = n_ctx * 2 * mid
    act0 = torch.cat([rnd(B * T * C, seed=21) * 1.5 + 0.3, torch.zeros(B * T * C), rnd(layers * B * kv_floats, seed=22) if cross else torch.zeros(0)])
    op, weights = build(False)
    cpu = Buffers(weights.clone(), act0.clone(), torch.zeros(4), {})
    run_program([op], cpu, B, 0)
    want = cpu.act[B * T * C: 2 * B * T * C]
    gw, ga, gs = weights.to(DEV), act0.to(DEV), torch.zeros(4, device=DEV)
    ext = {k: v.to(DEV) for k, v in _handoff_ext(B, T).items()}
    b = rt.MdtBindings()
    b.weights, b.act, b.shr = rt.ptr(gw), rt.ptr(ga), rt.ptr(gs)
    b.ext[3], b.ext[4] = rt.ptr(ext[3]), rt.ptr(ext[4])
    prog = rt.Program([op])
    outs = []
    try:
        with torch.cuda.device(DEV):
            for rep in range(120):
                lib.mdt_set_tuning(b"pair_stride", 1 if rep % 2 else 8)
                ga[B * T * C: 2 * B * T * C].zero_()
                prog.run(b, B, 0)
                if rep < 4 or rep % 10 == 0:
                    outs.append(ga[B * T * C: 2 * B * T * C].clone())
                else:
                    assert torch.equal(ga[B * T * C: 2 * B * T * C], outs[0]), rep
            torch.cuda.synchronize()
    finally:
        lib.mdt_set_tuning(b"pair_stride", 0)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    flags = ext[3].view(torch.int32).cpu()
    nsub = 1 + layers * (3 if cross else 2)
    assert int(flags[0]) == 0 and bool((flags[64::32] == 120 * nsub).all())
    tol = 2e-4 * max(1.0, want.abs().max().item())
    assert (outs[0].cpu() - want).abs().max() < tol
    # the whole-workgroup form of the same transformer (other summation order of the heads' partial sums)
    opw, ww = build(True)
    (gaw, _, _), _ = run_both([opw], ww, act0, torch.zeros(4), {}, B)
    assert (gaw[B * T * C: 2 * B * T * C] - outs[0].cpu()).abs().max() < tol


def _resnet_sd(p, c, cin, seed0):
    r = lambda *sh, seed, scale=1.0: rnd(*sh, seed=seed0 + seed, scale=scale)   # noqa: E731
    sd = {p + "block1.groupnorm.weight": 1 + 0.2 * r(cin, seed=1), p + "block1.groupnorm.bias": 0.2 * r(cin, seed=2),
          p + "block1.project.weight": r(c, cin, 3, seed=3, scale=(3 * cin) ** -0.5), p + "block1.project.bias": 0.1 * r(c, seed=4),
          p + "block2.groupnorm.weight": 1 + 0.2 * r(c, seed=5), p + "block2.groupnorm.bias": 0.2 * r(c, seed=6),
          p + "block2.project.weight": r(c, c, 3, seed=7, scale=(3 * c) ** -0.5), p + "block2.project.bias": 0.1 * r(c, seed=8)}
    if cin != c:
        sd[p + "to_out.weight"] = r(c, cin, 1, seed=9, scale=cin ** -0.5)
        sd[p + "to_out.bias"] = 0.1 * r(c, seed=10)
    return sd


@pytest.mark.parametrize("kind,T,B,n_res,layers,cross", [
    (1, 16, 5, 3, 0, False),      # down path: the blocks alone, outputs stored as skips
    (1, 16, 70, 1, 1, True),      # ... and in front of a transformer with cross-attention
    (1, 4, 18, 2, 1, False),      # 4 tokens per sample: sample boundaries inside the 16-lane row
    (2, 16, 5, 4, 2, False),      # up path: cat([x, skip / sqrt 2]) blocks + the pre-transformer (configs[1])
    (2, 16, 1030, 1, 0, False),   # many workgroups, ragged last one
    (2, 8, 9, 2, 1, False),
    (2, 16, 33, 1, 1, True),
])
def test_resnet_blocks_inside_the_transformer_launch(kind, T, B, n_res, layers, cross, prod):
    """MDT_OP_TF128 with MDT_F_RES_KIND 1 / 2 (k_tf128.hip RES = 1 / 2): ResnetBlock1d blocks of the 128-channel level in front
    of the transformer in one launch, against (i) the CPU interpreter of the op and (ii) the reference's module arithmetic
    (modules.py:145-205: GroupNorm -> [FiLM] -> SiLU -> Conv1d(k = 3), twice, + to_out(x) | x; :828-829: cat with the scaled
    skip).  Kind 1 also stores every block's output; kind 2 reads its skips in reverse order."""
    from moleculediffusiontransformer_amd.compiler import Ten, UNetCompiler
    from moleculediffusiontransformer_amd.netspec import inverse_unet_config
    import torch.nn.functional as F
    C, n_ctx, mid, G = 128, 12, 512, 8
    cfg = inverse_unet_config(16, 64, 128, n_ctx)
    p = "tf."
    sd = _transformer_sd(p, C, max(layers, 1), cross)
    blocks = [f"res{k}." for k in range(n_res)]
    for k, bp in enumerate(blocks):
        sd.update(_resnet_sd(bp, C, C if kind == 1 else 2 * C, 100 * (k + 1)))
    comp = UNetCompiler(cfg, 64, n_ctx, sd)
    assert all(comp.res128_ok(bp, C, T, G, kind == 2) for bp in blocks) and comp.tf128_ok(C, T, layers, cross, kind, n_res)
    # per-sample arena: [x | y | skip 0 .. n_res-1 | K/V layers]
    x, y = Ten(A, 0, T, C), Ten(A, T * C, T, C)
    skips = [Ten(A, (2 + k) * T * C, T, C) for k in range(n_res)]
    if kind == 2:
        skips = skips[::-1]                  # consumed from the highest address downwards
    comp.transformer_fused128(x, p, C, layers, cross, False, res=(kind, blocks, G, skips, 2 ** -0.5), y=y)
    op = comp.ops[0]
    assert op.kind == rt.OP_TF128 and len(comp.ops) == 1
    film_off = 64
    op.p3 = ref(S, film_off)
    kv_floats = n_ctx * 2 * mid
    if cross:
        op.a2 = ref(A, (2 + n_res) * T * C)
    act_x = rnd(B * T * C, seed=13) * 1.5 + 0.3
    sk_in = rnd(n_res * B * T * C, seed=16) * 1.2 - 0.1 if kind == 2 else torch.zeros(n_res * B * T * C)
    kv_all = rnd(layers * B * kv_floats, seed=14) if cross else torch.zeros(0)
    act = torch.cat([act_x, torch.zeros(B * T * C), sk_in, kv_all])
    shr = torch.cat([torch.zeros(film_off), 0.3 * rnd(n_res * 2 * C, seed=15), torch.zeros(256)])
    (ga, _, _), (ca, _, _) = run_both([op], comp.W.pack(), act, shr, {}, B)
    n = B * T * C
    yg, yc = ga[n: 2 * n].view(B, T, C), ca[n: 2 * n].view(B, T, C)
    assert torch.isfinite(yg).all()
    tol = 2e-4 * max(1.0, yc.abs().max().item())
    assert (yg - yc).abs().max() < tol, (yg - yc).abs().max().item()
    assert torch.equal(ga[:n], act_x)
    if kind == 1:                            # every block's output stored as a skip tensor
        sg, sc = ga[2 * n: (2 + n_res) * n], ca[2 * n: (2 + n_res) * n]
        assert (sg - sc).abs().max() < tol
    else:
        assert torch.equal(ga[2 * n: (2 + n_res) * n], sk_in)
    # ---- the reference's arithmetic for the ResNet part (the transformer part is covered by test_fused_transformer) ----
    if layers == 0:
        h = act_x.view(B, T, C).transpose(1, 2).double()
        for k, bp in enumerate(blocks):
            fl = shr[film_off + k * 2 * C: film_off + (k + 1) * 2 * C].double()
            xin = h
            if kind == 2:
                sk = sk_in.view(n_res, B, T, C)[n_res - 1 - k].transpose(1, 2).double() * 2 ** -0.5
                xin = torch.cat([h, sk], dim=1)
            g = lambda key: sd[bp + key].double()   # noqa: E731
            t1 = F.conv1d(F.silu(F.group_norm(xin, G, g("block1.groupnorm.weight"), g("block1.groupnorm.bias"), 1e-5)),
                          g("block1.project.weight"), g("block1.project.bias"), padding=1)
            t2 = F.group_norm(t1, G, g("block2.groupnorm.weight"), g("block2.groupnorm.bias"), 1e-5)
            t2 = F.silu(t2 * (fl[:C].view(1, C, 1) + 1) + fl[C:].view(1, C, 1))
            t2 = F.conv1d(t2, g("block2.project.weight"), g("block2.project.bias"), padding=1)
            h = t2 + (F.conv1d(xin, g("to_out.weight"), g("to_out.bias")) if kind == 2 else xin)
            if kind == 1:
                assert (ga[(2 + k) * n: (3 + k) * n].view(B, T, C).double() - h.transpose(1, 2)).abs().max() < tol
        assert (yg.double() - h.transpose(1, 2)).abs().max() < tol


@pytest.mark.parametrize("kind,T,B,n_res", [
    (1, 4, 8, 3),        # down path of configs[1]: three blocks, every output stored as a skip
    (1, 4, 70, 1),       # bottleneck block; ragged last workgroup (280 rows)
    (1, 1, 37, 2),       # one token per sample (configs[2]): only the centre tap is streamed
    (1, 16, 3, 1),       # a whole 16-lane row per sample
    (2, 4, 8, 4),        # up path of configs[1]: four two-source blocks, skips consumed downwards
    (2, 4, 1030, 1),     # many workgroups, ragged last one
    (2, 1, 50, 2),
    (2, 8, 5, 2),
])
@pytest.mark.parametrize("form", ["whole", "pair8", "pair1"])
def test_resnet_chain_256(kind, T, B, n_res, form, prod, monkeypatch):
    """MDT_OP_RES256 (k_res256.hip): a chain of ResnetBlock1d blocks of a 256-channel level in one launch, against (i) the CPU
    interpreter of the op and (ii) the reference's module arithmetic (modules.py:145-205: GroupNorm -> [FiLM] -> SiLU -> Conv1d(k =
    3), twice, + to_out(x) | x; :828-829: cat with the scaled skip).  Kind 1 also stores every block's output; kind 2 reads its
    skips in reverse order.  Both product types.  form: 'whole' = one workgroup per 32-row block; 'pair8' / 'pair1' (round 6) = the
    PAIR-SPLIT chain (NSPLIT = 2: half hh streams output chunks 2 hh, 2 hh + 1 of every convolution, the pair hands its chunks to each
    other inside the launch) with the partners 8 workgroup ids apart or neighbours -- the result must not depend on the placement
    and equals the unsplit chain's BIT FOR BIT (every output channel is the same MFMA sequence)."""
    from moleculediffusiontransformer_amd.compiler import Ten, UNetCompiler
    from moleculediffusiontransformer_amd.netspec import inverse_unet_config
    import torch.nn.functional as F
    monkeypatch.setenv("MDT_RES256", "1")
    C, G = 256, 8
    cfg = inverse_unet_config(16, 64, 128, 12)
    blocks = [f"res{k}." for k in range(n_res)]
    sd = {}
    for k, bp in enumerate(blocks):
        sd.update(_resnet_sd(bp, C, C if kind == 1 else 2 * C, 100 * (k + 1)))
    comp = UNetCompiler(cfg, 64, 12, sd, gemm_mode="f32" if prod == "f32" else "bf16x3")
    assert all(comp.res256_ok(bp, C, T, G, kind == 2) for bp in blocks)
    # per-sample arena: [x | y | skip 0 .. n_res-1]
    x, y = Ten(A, 0, T, C), Ten(A, T * C, T, C)
    skips = [Ten(A, (2 + k) * T * C, T, C) for k in range(n_res)]
    if kind == 2:
        skips = skips[::-1]                  # consumed from the highest address downwards
    comp.pair_stride = 1 if form == "pair1" else 8
    comp.resnet_chain256(x, blocks, kind, skips, 2 ** -0.5, y, False, nsplit=1 if form == "whole" else 2)
    op = comp.ops[0]
    assert op.kind == rt.OP_RES256 and len(comp.ops) == 1 and op.i[rt.F_NPOST] == (1 if T == 1 else 3)
    assert op.i[rt.F_NSPLIT] == (0 if form == "whole" else 2)
    ext = {} if form == "whole" else _handoff_ext(B, T)
    film_off = 64
    op.p3 = ref(S, film_off)
    act_x = rnd(B * T * C, seed=13) * 1.5 + 0.3
    sk_in = rnd(n_res * B * T * C, seed=16) * 1.2 - 0.1 if kind == 2 else torch.zeros(n_res * B * T * C)
    act = torch.cat([act_x, torch.zeros(B * T * C), sk_in])
    shr = torch.cat([torch.zeros(film_off), 0.3 * rnd(n_res * 2 * C, seed=15), torch.zeros(256)])
    (ga, _, ge), (ca, _, _) = run_both([op], comp.W.pack(), act, shr, ext, B)
    n = B * T * C
    if form != "whole":
        assert int(ge[3].view(torch.int32)[0]) == 0                 # no hand-off poll timed out
        # the unsplit chain on the same inputs: bitwise equal
        comp1 = UNetCompiler(cfg, 64, 12, sd, gemm_mode="f32" if prod == "f32" else "bf16x3")
        comp1.resnet_chain256(x, blocks, kind, skips, 2 ** -0.5, y, False, nsplit=1)
        comp1.ops[0].p3 = ref(S, film_off)
        (gw, _, _), _ = run_both([comp1.ops[0]], comp1.W.pack(), act, shr, {}, B)
        assert torch.equal(ga, gw)
    yg, yc = ga[n: 2 * n].view(B, T, C), ca[n: 2 * n].view(B, T, C)
    assert torch.isfinite(yg).all()
    tol = 2e-4 * max(1.0, yc.abs().max().item())
    assert (yg - yc).abs().max() < tol, (yg - yc).abs().max().item()
    assert torch.equal(ga[:n], act_x)
    if kind == 1:                            # every block's output stored as a skip tensor
        sg, sc = ga[2 * n: (2 + n_res) * n], ca[2 * n: (2 + n_res) * n]
        assert (sg - sc).abs().max() < tol
    else:
        assert torch.equal(ga[2 * n: (2 + n_res) * n], sk_in)
    # repeated launches return the same bits
    (gb, _, _), _ = run_both([op], comp.W.pack(), act, shr, ext, B)
    assert torch.equal(ga, gb)
    # ---- the reference's arithmetic ----
    h = act_x.view(B, T, C).transpose(1, 2).double()
    for k, bp in enumerate(blocks):
        fl = shr[film_off + k * 2 * C: film_off + (k + 1) * 2 * C].double()
        xin = h
        if kind == 2:
            sk = sk_in.view(n_res, B, T, C)[n_res - 1 - k].transpose(1, 2).double() * 2 ** -0.5
            xin = torch.cat([h, sk], dim=1)
        g = lambda key: sd[bp + key].double()   # noqa: E731
        t1 = F.conv1d(F.silu(F.group_norm(xin, G, g("block1.groupnorm.weight"), g("block1.groupnorm.bias"), 1e-5)),
                      g("block1.project.weight"), g("block1.project.bias"), padding=1)
        t2 = F.group_norm(t1, G, g("block2.groupnorm.weight"), g("block2.groupnorm.bias"), 1e-5)
        t2 = F.silu(t2 * (fl[:C].view(1, C, 1) + 1) + fl[C:].view(1, C, 1))
        t2 = F.conv1d(t2, g("block2.project.weight"), g("block2.project.bias"), padding=1)
        h = t2 + (F.conv1d(xin, g("to_out.weight"), g("to_out.bias")) if kind == 2 else xin)
        if kind == 1:
            assert (ga[(2 + k) * n: (3 + k) * n].view(B, T, C).double() - h.transpose(1, 2)).abs().max() < tol
    assert (yg.double() - h.transpose(1, 2)).abs().max() < tol
