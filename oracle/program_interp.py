"""CPU interpreter of mdt_op programs (TEST INFRASTRUCTURE).

Executes the op list produced by moleculediffusiontransformer_amd/compiler.py with plain PyTorch CPU
ops over flat fp32 buffers, following the op semantics documented in include/mdt_hip.h.  It lets the
-m "not gpu" tests check the lowering (buffer offsets, tap/row mappings, weight packing, channel
padding) against oracle/unet_oracle.py without a GPU; the HIP kernels are then checked against the
same oracle on the GPU box.  Never imported by the product path.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from moleculediffusiontransformer_amd import runtime as rt


class Buffers:
    def __init__(self, weights: torch.Tensor, act: torch.Tensor, shr: torch.Tensor, ext: Dict[int, torch.Tensor]):
        self.weights, self.act, self.shr, self.ext = weights, act, shr, ext

    def view(self, ref, B: int, n: int) -> Optional[torch.Tensor]:
        if ref.space == rt.SP_NONE:
            return None
        if ref.space == rt.SP_WEIGHT:
            buf, off = self.weights, ref.off
        elif ref.space == rt.SP_ACT:
            buf, off = self.act, ref.off * B
        elif ref.space == rt.SP_SHR:
            buf, off = self.shr, ref.off
        else:
            buf, off = self.ext[ref.space - rt.SP_EXT0], ref.off
        assert off + n <= buf.numel(), (ref.space, off, n, buf.numel())
        return buf[off: off + n]


def _silu(x):
    return x / (1.0 + torch.exp(-x))


def run_program(ops, bufs: Buffers, B: int, n_shared_rows: int = 0) -> None:
    for op in ops:
        i, f = op.i, op.f
        if op.kind in (rt.OP_GEMM, rt.OP_PREP16):
            prep = op.kind == rt.OP_PREP16
            batches = B if (prep or i[rt.G_M_MODE] == 0) else (n_shared_rows if i[rt.G_M_MODE] == 1 else 1)
            r_out, r_in, lda, cin, taps = i[rt.G_R_OUT], i[rt.G_R_IN], i[rt.G_LDA], i[rt.G_CIN], i[rt.G_TAPS]
            n, ldc, o_rows = i[rt.G_N], i[rt.G_LDC], i[rt.G_O_ROWS]
            if i[rt.G_WFMT] & 2 and i[rt.G_WFMT] not in (16, 17):   # A already bf16 (GEMM: written by MDT_OP_PREP16 / a bf16 epilogue;
                # PREP16 with WFMT 2, round 6: the bf16 residual stream itself): lda / a_col in bf16 elements
                a = bufs.view(op.a, B, batches * r_in * lda // 2).view(torch.bfloat16).float().view(batches, r_in, lda)
            else:
                a = bufs.view(op.a, B, batches * r_in * lda).view(batches, r_in, lda)
            a = a[:, :, i[rt.G_A_COL]: i[rt.G_A_COL] + cin]
            pro = i[rt.G_PRO]
            if pro == rt.PRO_LAYERNORM and op.p0.space == rt.SP_NONE:      # (ring-tile projection: no affine, folded into W / bias)
                a = F.layer_norm(a, (cin,), eps=float(f[0]))
            elif pro == rt.PRO_LAYERNORM:
                g, b = bufs.view(op.p0, B, cin), bufs.view(op.p1, B, cin)
                a = F.layer_norm(a, (cin,), g, b, eps=float(f[0]))
            elif pro == rt.PRO_GROUPNORM:
                G, gs = i[rt.G_GROUPS], i[rt.G_GSIZE]
                g, b = bufs.view(op.p0, B, cin), bufs.view(op.p1, B, cin)
                st = bufs.view(op.p2, B, batches * G * 2).view(batches, G, 2)
                grp = torch.clamp(torch.arange(cin) // gs, max=G - 1)
                mean, rstd = st[:, grp, 0].unsqueeze(1), st[:, grp, 1].unsqueeze(1)
                a = (a - mean) * rstd * g + b
                if op.p3.space != rt.SP_NONE:
                    ss = bufs.view(op.p3, B, 2 * cin)
                    a = a * (ss[:cin] + 1.0) + ss[cin:]
                if i[rt.G_PRO_SILU]:
                    a = _silu(a)
            elif pro == rt.PRO_SILU:
                a = _silu(a)
            if prep:
                dst = bufs.view(op.out, B, batches * r_in * cin // 2)
                dst[:] = a.contiguous().to(torch.bfloat16).view(-1).view(torch.float32)
                continue
            nph = max(int(i[rt.G_PHASES]), 1)         # > 1: ConvTranspose1d phases sharing one op (include/mdt_hip.h)
            wbase = i[rt.G_WFMT] & 63          # (round 6: + 128 = the LayerNorm of the A rows folded into the GEMM, MDT_G_WFMT 134)
            if wbase in (1, 2, 6, 10, 38):          # plain bf16 products: one weight plane, A rounded to bf16 after the prologue
                half = nph * n * taps * cin // 2
                w_all = bufs.view(op.w, B, half).view(torch.bfloat16).float().view(nph, n, taps, cin)
                a = a.to(torch.bfloat16).float()
            elif i[rt.G_WFMT] in (16, 17):       # ring tiles (k_proj.hip): [64 features][128 k] hi | lo planes (17: fp32 fragments), order (chunk, K half)
                khn = cin // 128
                stream = bufs.view(op.w, B, (n // 64) * khn * 64 * 128)
                w_all = torch.zeros(n, cin)
                for c_ in range(n // 64):
                    for h_ in range(khn):
                        w_all[64 * c_: 64 * c_ + 64, 128 * h_: 128 * h_ + 128] = _untile(stream, c_ * khn + h_, 64, 128, i[rt.G_WFMT] == 17)
                w_all = w_all.view(1, n, 1, cin)
            elif op.a2.space != rt.SP_NONE:      # split-bf16 weights: two bf16 planes stored as raw bits
                half = nph * n * taps * cin // 2
                hi = bufs.view(op.w, B, half).view(torch.bfloat16).float()
                lo = bufs.view(op.a2, B, half).view(torch.bfloat16).float()
                w_all = (hi + lo).view(nph, n, taps, cin)
            else:
                w_all = bufs.view(op.w, B, nph * n * taps * cin).view(nph, n, taps, cin)
            r = torch.arange(r_out)
            out = None if wbase in (6, 38) else bufs.view(op.out, B, batches * o_rows * ldc).view(batches, o_rows, ldc)
            for ph in range(nph):
                w = w_all[ph]
                t_off, o_off = i[rt.G_T_OFF], i[rt.G_O_OFF]
                if nph > 1:
                    t_off = 1 if ph < nph // 2 else 0
                    o_off = nph * t_off + ph - nph // 2
                acc = torch.zeros(batches, r_out, n)
                for t in range(taps):
                    src = r * i[rt.G_T_STRIDE] + t * i[rt.G_T_DJ] + t_off
                    ok = (src >= 0) & (src < r_in)
                    rows = a[:, src.clamp(0, r_in - 1), :] * ok.view(1, -1, 1)
                    acc = acc + rows @ w[:, t, :].T
                if i[rt.G_WFMT] & 128:         # MDT_G_WFMT 134: LayerNorm of the raw bf16 A rows folded into the GEMM -- applied to the
                    # accumulators from the rows' mean / rstd and W's column sums (the statistics in fp32 on the bf16 values, as the kernel's)
                    assert taps == 1 and nph == 1 and pro == rt.PRO_NONE
                    mean = a.mean(dim=-1)
                    var = ((a * a).mean(dim=-1) - mean * mean).clamp(min=0.0)
                    rstd = 1.0 / torch.sqrt(var + float(f[0]))
                    acc = rstd.unsqueeze(-1) * (acc - mean.unsqueeze(-1) * bufs.view(op.p0, B, n))
                if op.bias.space != rt.SP_NONE:
                    acc = acc + bufs.view(op.bias, B, n)
                if i[rt.G_ACT] == 1:
                    acc = F.gelu(acc)
                orow = r * i[rt.G_O_STRIDE] + o_off
                if op.res.space != rt.SP_NONE:
                    ldr = i[rt.G_LDR]
                    if wbase == 38:            # bf16 residual stream (round 6): ldr in bf16 elements, widened exactly
                        res = bufs.view(op.res, B, batches * o_rows * ldr // 2).view(torch.bfloat16).float().view(batches, o_rows, ldr)
                    else:
                        res = bufs.view(op.res, B, batches * o_rows * ldr).view(batches, o_rows, ldr)
                    acc = acc + res[:, orow, :n]
                if wbase in (6, 38):           # bf16 output (ldc / o_col in bf16 elements; a column block of wider rows keeps the rest)
                    assert nph == 1 and r_out == o_rows
                    dst = bufs.view(op.out, B, batches * o_rows * ldc // 2)
                    rows16 = dst.view(torch.bfloat16).view(batches, o_rows, ldc).clone()
                    rows16[:, :, i[rt.G_O_COL]: i[rt.G_O_COL] + n] = acc.to(torch.bfloat16)
                    dst[:] = rows16.contiguous().view(-1).view(torch.float32)
                    continue
                out[:, orow, i[rt.G_O_COL]: i[rt.G_O_COL] + n] = acc
                if wbase == 10:                # ... and a bf16 copy of the fp32 output (the next GEMM's A operand)
                    bufs.view(op.p0, B, batches * o_rows * n // 2)[:] = acc.contiguous().to(torch.bfloat16).view(-1).view(torch.float32)
        elif op.kind == rt.OP_GN_STATS:
            rows, ld, G, gs = i[rt.N_ROWS], i[rt.N_LD], i[rt.N_GROUPS], i[rt.N_GSIZE]
            x = bufs.view(op.a, B, B * rows * ld).view(B, rows, ld)[:, :, : G * gs].reshape(B, rows, G, gs)
            mean = x.mean(dim=(1, 3))
            var = x.var(dim=(1, 3), unbiased=False)
            st = bufs.view(op.out, B, B * G * 2).view(B, G, 2)
            st[:, :, 0] = mean
            st[:, :, 1] = 1.0 / torch.sqrt(var + float(f[0]))
        elif op.kind == rt.OP_GN_ACT:
            rows, ld, G, gs = i[rt.N_ROWS], i[rt.N_LD], i[rt.N_GROUPS], i[rt.N_GSIZE]
            if op.a2.space != rt.SP_NONE:      # round 6: cat([a, SCALE2 * a2]) without the concatenated tensor (modules.py:828-829)
                ca = i[rt.N_CA]
                x = torch.cat([bufs.view(op.a, B, B * rows * ca).view(B, rows, ca),
                               bufs.view(op.a2, B, B * rows * (ld - ca)).view(B, rows, ld - ca) * float(f[1])], dim=2)
            else:
                x = bufs.view(op.a, B, B * rows * ld).view(B, rows, ld)
            if op.p2.space != rt.SP_NONE:      # ... and a raw bf16 copy of the input
                bufs.view(op.p2, B, B * rows * ld // 2)[:] = x.contiguous().to(torch.bfloat16).view(-1).view(torch.float32)
            y = F.group_norm(x.transpose(1, 2), G, bufs.view(op.p0, B, ld), bufs.view(op.p1, B, ld), float(f[0])).transpose(1, 2)
            if op.p3.space != rt.SP_NONE:
                ss = bufs.view(op.p3, B, 2 * ld)
                y = y * (ss[:ld] + 1.0) + ss[ld:]
            if i[rt.N_SILU]:
                y = _silu(y)
            if i[rt.N_OUT16]:
                bufs.view(op.out, B, B * rows * ld // 2)[:] = y.contiguous().to(torch.bfloat16).view(-1).view(torch.float32)
            else:
                bufs.view(op.out, B, B * rows * ld).view(B, rows, ld)[:] = y
        elif op.kind == rt.OP_RCONV:
            _rconv(op, bufs, B)
        elif op.kind == rt.OP_RESBLOCK:
            _resblock(op, bufs, B)
        elif op.kind == rt.OP_ATTN:
            T, Tk, H = i[rt.A_T], i[rt.A_TK], i[rt.A_HEADS]
            ldq, ldkv, ldo, bs = i[rt.A_LDQ], i[rt.A_LDKV], i[rt.A_LDO], i[rt.A_KV_BSTRIDE]
            D = 64
            qc, kc = i[rt.A_QCOL], i[rt.A_KCOL]
            in16 = i[rt.A_IN16]                 # mask: 1 = q is bf16, 2 = k | v are bf16 (pitches / columns in bf16 elements)

            def rows(ref, n, ld, half):
                if half:
                    return bufs.view(ref, B, n * ld // 2).view(torch.bfloat16).float().view(n, ld)
                return bufs.view(ref, B, n * ld).view(n, ld)
            q = rows(op.a, B * T, ldq, in16 & 1).view(B, T, ldq)[:, :, qc: qc + H * D].reshape(B, T, H, D).transpose(1, 2)
            if bs == 0:
                kv = rows(op.a2, Tk, ldkv, in16 & 2).view(1, Tk, ldkv).expand(B, -1, -1)
            else:
                assert bs == Tk
                kv = rows(op.a2, B * Tk, ldkv, in16 & 2).view(B, Tk, ldkv)
            k = kv[:, :, kc: kc + H * D].reshape(B, Tk, H, D).transpose(1, 2)
            v = kv[:, :, kc + H * D: kc + 2 * H * D].reshape(B, Tk, H, D).transpose(1, 2)
            sim = (q @ k.transpose(-1, -2)) * float(f[0])
            o = (sim.softmax(-1) @ v).transpose(1, 2).reshape(B, T, H * D)
            if i[rt.A_OUT16]:
                assert ldo == H * D
                bufs.view(op.out, B, B * T * ldo // 2)[:] = o.contiguous().to(torch.bfloat16).view(-1).view(torch.float32)
            else:
                out = bufs.view(op.out, B, B * T * ldo).view(B, T, ldo)
                out[:, :, : H * D] = o
        elif op.kind == rt.OP_CONCAT:
            rows, ca, cb = i[rt.C_ROWS], i[rt.C_CA], i[rt.C_CB]
            a = bufs.view(op.a, B, B * rows * ca).view(B * rows, ca)
            b = bufs.view(op.a2, B, B * rows * cb).view(B * rows, cb)
            out = bufs.view(op.out, B, B * rows * (ca + cb)).view(B * rows, ca + cb)
            out[:, :ca] = a
            out[:, ca:] = b * float(f[0])
        elif op.kind == rt.OP_PATCH:
            rl, cl, ld_in, ld_out, p, inv = (i[rt.P_ROWS_IN], i[rt.P_C_IN], i[rt.P_LD_IN], i[rt.P_LD_OUT],
                                             i[rt.P_PATCH], i[rt.P_INVERSE])
            if not inv:      # y[b, l, c*p + q] = x[b, l*p + q, c]
                x = bufs.view(op.a, B, B * rl * ld_in).view(B, rl // p, p, ld_in)[:, :, :, :cl]
                out = bufs.view(op.out, B, B * (rl // p) * ld_out).view(B, rl // p, ld_out)
                out[:, :, : cl * p] = x.permute(0, 1, 3, 2).reshape(B, rl // p, cl * p)
            else:            # x[b, l*p + q, c] = y[b, l, c*p + q]
                y = bufs.view(op.a, B, B * (rl // p) * ld_in).view(B, rl // p, ld_in)[:, :, : cl * p]
                out = bufs.view(op.out, B, B * rl * ld_out).view(B, rl // p, p, ld_out)
                out[:, :, :, :cl] = y.reshape(B, rl // p, cl, p).permute(0, 1, 3, 2)
        elif op.kind == rt.OP_TIME_EMBED:
            half, ld = i[rt.T_HALF], i[rt.T_LD]
            n = n_shared_rows
            t = bufs.view(op.a, B, n).view(n, 1)
            w = bufs.view(op.w, B, half).view(1, half)
            fr = t * w * 2 * math.pi
            out = bufs.view(op.out, B, n * ld).view(n, ld)
            out.zero_()
            out[:, 0:1] = t
            out[:, 1: 1 + half] = fr.sin()
            out[:, 1 + half: 1 + 2 * half] = fr.cos()
        elif op.kind == rt.OP_TBLOCK:
            _tblock(op, bufs, B)
        elif op.kind == rt.OP_ATTN_CTX:
            T, Tk, H, ldkv, bs = i[rt.A_T], i[rt.A_TK], i[rt.A_HEADS], i[rt.A_LDKV], i[rt.A_KV_BSTRIDE]
            q = bufs.view(op.a, B, B * T * H * 128).view(B, T * H, 128)
            if bs == 0:
                c = bufs.view(op.a2, B, Tk * ldkv).view(1, Tk, ldkv)[:, :, :128].expand(B, -1, -1)
            else:
                c = bufs.view(op.a2, B, B * Tk * ldkv).view(B, Tk, ldkv)[:, :, :128]
            pr = ((q @ c.transpose(1, 2)) * float(f[0])).softmax(-1)
            bufs.view(op.out, B, B * T * H * 128).view(B, T * H, 128)[:] = pr @ c
        elif op.kind == rt.OP_TF128:
            _tf128(op, bufs, B)
        elif op.kind == rt.OP_TF256:
            _tf256(op, bufs, B)
        elif op.kind == rt.OP_RES256:
            _res256(op, bufs, B)
        else:
            raise ValueError(f"unknown op kind {op.kind}")


_SLOT_PERM = [32 * (s >> 5) + 16 * ((s & 7) >> 2) + 4 * ((s >> 3) & 3) + (s & 3) for s in range(64)]


def _untile(stream: torch.Tensor, k: int, rows: int, cols: int, f32: bool = False) -> torch.Tensor:
    """Tile k of the weight stream -> fp32 [rows][cols].  Split-bf16 format: bf16 hi plane then lo plane (256*C bytes).
    f32 (MDT_F_WF32 / MDT_R_WF32 / MDT_K_WF32, include/mdt_hip.h): the same bytes hold fp32 MFMA fragments -- fragment
    (row tile rt, k-step st, half lo) at ((rt * (cols / 32) + st) * 2 + lo) * 256 floats, float r of lane i + 16 g =
    W[16 rt + i][32 st + 8 g + 4 lo + r]."""
    n = rows * cols                       # bf16 elements per plane; tile = 2 n bf16 = n floats
    if f32:
        raw = stream[k * n: (k + 1) * n].contiguous().view(rows // 16, cols // 32, 2, 4, 16, 4)   # [rt][st][lo][g][i][r]
        return raw.permute(0, 4, 1, 3, 2, 5).reshape(rows, cols).clone()
    raw = stream[k * n: (k + 1) * n].contiguous().view(torch.bfloat16)
    return (raw[:n].float() + raw[n:].float()).view(rows, cols)


def _rconv(op, bufs: Buffers, B: int) -> None:
    """MDT_OP_RCONV semantics (include/mdt_hip.h): GroupNorm + FiLM + SiLU + Conv1d(k = 1 | 3), C -> C channels,
    weights reconstructed from the packed tiles ([64 features][128 k], order tap / K half / feature chunk)."""
    i, f = op.i, op.f
    T, C, lda, ldc, ldr, taps, gs = (i[rt.R_T], i[rt.R_C], i[rt.R_LDA], i[rt.R_LDC], i[rt.R_LDR], i[rt.R_TAPS],
                                     i[rt.R_GSIZE])
    ksrc = i[rt.R_KSRC]
    if ksrc > 1:                                        # K = ksrc C projection: consecutive C-channel blocks of ONE tensor
        xa = bufs.view(op.a, B, B * T * lda).view(B, T, lda)
        srcs = [xa[:, :, s_ * C: (s_ + 1) * C] * float(f[1]) for s_ in range(ksrc)]      # every K block x IN_SCALE, as k_rconv does
    else:
        srcs = [(bufs.view(op.a, B, B * T * lda).view(B, T, lda)[:, :, :C] * float(f[1]))]
    if ksrc <= 1 and op.a2.space != rt.SP_NONE:         # second half of a concatenated input
        lda2 = i[rt.R_LDA2]
        srcs.append(bufs.view(op.a2, B, B * T * lda2).view(B, T, lda2)[:, :, :C] * float(f[2]))
    nsrc = len(srcs)
    ys = []
    for s_, x in enumerate(srcs):
        if gs > 0:
            gain = bufs.view(op.p0, B, nsrc * C)[s_ * C: (s_ + 1) * C]
            beta = bufs.view(op.p1, B, nsrc * C)[s_ * C: (s_ + 1) * C]
            y = F.group_norm(x.transpose(1, 2), C // gs, gain, beta, float(f[0])).transpose(1, 2)
            if op.p3.space != rt.SP_NONE:
                ss = bufs.view(op.p3, B, i[rt.R_FILM_LD] + C)
                y = y * (ss[:C] + 1.0) + ss[i[rt.R_FILM_LD]: i[rt.R_FILM_LD] + C]
            if i[rt.R_SILU]:
                y = _silu(y)
        else:
            y = x
        ys.append(y)
    y = torch.cat(ys, dim=2)
    nkh, nch = C // 128, C // 64
    nb = max(int(i[rt.R_NB]), 1)                        # NB x C output channels: NB convolutions of the same rows
    stream = bufs.view(op.w, B, nb * nsrc * taps * C * C)
    w = torch.empty(nb * C, nsrc * C, taps)
    k = 0
    for b_ in range(nb):
        for s_ in range(nsrc):
            for tap in range(taps):
                for kh in range(nkh):
                    for ch in range(nch):
                        w[b_ * C + 64 * ch: b_ * C + 64 * ch + 64, s_ * C + 128 * kh: s_ * C + 128 * kh + 128, tap] = \
                            _untile(stream, k, 64, 128, bool(i[rt.R_WF32]))
                        k += 1
    co = C // 2 if i[rt.R_HALF_OUT] else nb * C         # HALF_OUT: only output channels 0 .. C / 2 - 1 exist
    bias = bufs.view(op.bias, B, co) if op.bias.space != rt.SP_NONE else None
    o = F.conv1d(y.transpose(1, 2), w[:co], bias, padding=taps // 2).transpose(1, 2)
    if op.res.space != rt.SP_NONE:
        o = o + bufs.view(op.res, B, B * T * ldr).view(B, T, ldr)[:, :, :co]
    bufs.view(op.out, B, B * T * ldc).view(B, T, ldc)[:, :, :co] = o


def _resblock_steps(c: int, taps: int):
    """k-step enumeration of MDT_OP_RESBLOCK (csrc/k_resblock.hip): 32 (tap, channel) pairs per step, None = padding."""
    if taps == 3:
        if c == 64:
            return [[(s // 2, 32 * (s % 2) + j) for j in range(32)] for s in range(6)]
        return [[(j // 16, j % 16) for j in range(32)], [(2, j) if j < 16 else None for j in range(32)]]
    if c == 64:
        return [[(0, 32 * s + j) for j in range(32)] for s in range(2)]
    return [[(0, j) if j < 16 else None for j in range(32)]]


def _resblock(op, bufs: Buffers, B: int) -> None:
    """MDT_OP_RESBLOCK semantics (include/mdt_hip.h): the whole ResnetBlock1d (reference modules.py:145-205) with one
    GroupNorm group, weights reconstructed from the packed MFMA fragments."""
    i, f = op.i, op.f
    T, cin, cout, fld = i[rt.K_T], i[rt.K_CIN], i[rt.K_COUT], i[rt.K_FILM_LD]
    specs = [(cout, cin, 3), (cout, cout, 3), (cout, cin, 1)]
    nfrag = sum(len(_resblock_steps(c, taps)) * (n // 16) for n, c, taps in specs)
    stream = bufs.view(op.w, B, nfrag * 512)
    ws, k = [], 0
    for n, c, taps in specs:
        w = torch.zeros(n, c, taps)
        for step in _resblock_steps(c, taps):
            for r in range(n // 16):
                if i[rt.K_WF32]:
                    m = _untile(stream, k, 16, 32, True)
                else:
                    m = _untile(stream, k, 64, 8).view(4, 16, 8).permute(1, 0, 2).reshape(16, 32)   # lane 16 g + i -> [i][8 g + e]
                k += 1
                for j, tc in enumerate(step):
                    if tc is not None:
                        w[16 * r: 16 * r + 16, tc[1], tc[0]] = m[:, j]
        ws.append(w)
    vec = bufs.view(op.bias, B, 2 * cin + 4 * cout)
    g1, be1, b1 = vec[:cin], vec[cin: 2 * cin], vec[2 * cin: 2 * cin + cout]
    g2, be2, bo = (vec[2 * cin + cout: 2 * cin + 2 * cout], vec[2 * cin + 2 * cout: 2 * cin + 3 * cout],
                   vec[2 * cin + 3 * cout:])
    # cin / cout are padded channel counts; the block itself has ci / co channels (MDT_K_CIN_REAL / MDT_K_COUT_REAL, 0 = all)
    ci, co = (i[rt.K_CIN_REAL] or cin), (i[rt.K_COUT_REAL] or cout)
    pin, pout = i[rt.K_PATCH_IN], i[rt.K_PATCH_OUT]
    xin = bufs.view(op.a, B, B * T * cin)
    if pin > 1:          # still patched: a[l][c p + q] = x[l p + q][c]
        xin = xin.view(B, T // pin, cin, pin).permute(0, 1, 3, 2).reshape(B, T, cin)
    x = xin.view(B, T, cin).transpose(1, 2)[:, :ci]
    h = F.conv1d(_silu(F.group_norm(x, 1, g1[:ci], be1[:ci], float(f[0]))), ws[0][:co, :ci], b1[:co], padding=1)
    h = F.group_norm(h, 1, g2[:co], be2[:co], float(f[0]))
    if op.p3.space != rt.SP_NONE:
        ss = bufs.view(op.p3, B, fld + cout)
        h = h * (ss[:co, None] + 1.0) + ss[fld: fld + co, None]
    y = F.conv1d(_silu(h), ws[1][:co, :co], bo[:co], padding=1) + F.conv1d(x, ws[2][:co, :ci])
    o = torch.zeros(B, T, cout)
    o[:, :, :co] = y.transpose(1, 2)
    if pout > 1:         # written patched: out[l][c p + q] = y[l p + q][c]
        o = o.view(B, T // pout, pout, cout).permute(0, 1, 3, 2)
    bufs.view(op.out, B, B * T * cout)[:] = o.reshape(-1)


def _tblock(op, bufs: Buffers, B: int) -> None:
    """MDT_OP_TBLOCK semantics (include/mdt_hip.h) reconstructed from the packed tiles."""
    i, f = op.i, op.f
    mode, C, T, nchunk, nbias = i[rt.B_MODE], i[rt.B_C], i[rt.B_T], i[rt.B_NCHUNK], i[rt.B_NBIAS]
    x = bufs.view(op.a, B, B * T * C).view(B, T, C)
    chained = i[rt.B_VARIANT] == 4     # block input = a + res; head group 0 -> out (+ bias + input), group 1 -> p2
    if chained:
        x_in = x if op.res.space == rt.SP_NONE else x + bufs.view(op.res, B, B * T * C).view(B, T, C)
        x_out = bufs.view(op.out, B, B * T * C).view(B, T, C)
        split = op.p2.space != rt.SP_NONE
        x = x_in

    def finish(hidden, w_out, b_out):
        """hidden [B, T, mid] times w_out [C, mid]: in place, or split into the two head groups' partial sums."""
        if not chained:
            x.__iadd__(hidden @ w_out.T + b_out)
            return
        half = hidden.shape[-1] // 2
        if split:
            x_out[:] = x_in + hidden[..., :half] @ w_out[:, :half].T + b_out
            bufs.view(op.p2, B, B * T * C).view(B, T, C)[:] = hidden[..., half:] @ w_out[:, half:].T
        else:
            x_out[:] = x_in + hidden @ w_out.T + b_out
    tpc = 4 if mode == rt.TB_SELF else 2
    stream = bufs.view(op.w, B, (nchunk * tpc + (C // 64 if i[rt.B_POST] else 0)) * 64 * C)
    bias = bufs.view(op.bias, B, nbias)
    inv = torch.empty(64, dtype=torch.long)
    inv[torch.tensor(_SLOT_PERM)] = torch.arange(64)
    mid = 64 * nchunk

    def tile(k, rows, cols):
        if i[rt.B_VARIANT] not in (2, 3, 4):
            return _untile(stream, k, rows, cols, bool(i[rt.B_WF32]))      # (MDT_B_WF32: fp32 fragment tiles)
        # variant 2 (k_tblock32): every tile is stored as two 128-wide sub-tiles (K halves / output-row halves)
        f32 = bool(i[rt.B_WF32])                 # (round 6: fp32 fragment sub-tiles for the C = 256 variants too)
        if rows == 64:
            return torch.cat([_untile(stream, 2 * k, 64, 128, f32), _untile(stream, 2 * k + 1, 64, 128, f32)], dim=1)
        return torch.cat([_untile(stream, 2 * k, 128, 64, f32), _untile(stream, 2 * k + 1, 128, 64, f32)], dim=0)

    def proj(k0):          # stack of P tiles k0, k0 + tpc, ... -> [mid, C]
        return torch.cat([tile(h * tpc + k0, 64, C) for h in range(nchunk)])

    def outw():            # O tiles -> [C, mid] with the slot permutation undone
        return torch.cat([tile(h * tpc + tpc - 1, C, 64)[:, inv] for h in range(nchunk)], dim=1)

    if mode == rt.TB_FF:
        h = F.gelu(x @ proj(0).T + bias[:mid])
        if i[rt.B_POST]:
            # closing 1x1 convolution folded in: the W2 tiles hold Wout W2, C/64 extra output tiles hold Wout (natural k
            # order), the output bias holds Wout b2 + bout; no residual, result in `out`, x untouched
            wout = torch.cat([tile(nchunk * tpc + e, C, 64) for e in range(C // 64)], dim=1)
            bufs.view(op.out, B, B * T * C).view(B, T, C)[:] = h @ outw().T + x @ wout.T + bias[mid:]
            return
        finish(h, outw(), bias[mid:])
        return
    xn = F.layer_norm(x, (C,), None, None, eps=float(f[0]))
    H, D = nchunk, 64
    q = (xn @ proj(0).T + bias[:mid]).view(B, T, H, D).transpose(1, 2)
    if mode == rt.TB_SELF:
        k = (xn @ proj(1).T + bias[mid: 2 * mid]).view(B, T, H, D).transpose(1, 2)
        v = (xn @ proj(2).T + bias[2 * mid: 3 * mid]).view(B, T, H, D).transpose(1, 2)
        bo = bias[3 * mid:]
    else:
        Tk, bs, ldkv = i[rt.B_TK], i[rt.B_KV_BSTRIDE], i[rt.B_LDKV]
        if bs == 0:
            kv = bufs.view(op.a2, B, Tk * ldkv).view(1, Tk, ldkv).expand(B, -1, -1)
        else:
            kv = bufs.view(op.a2, B, B * Tk * ldkv).view(B, Tk, ldkv)
        if i[rt.B_KV2]:                 # dual guidance batch: the second half attends to the batch-invariant rows p1
            kvf = bufs.view(op.p1, B, Tk * ldkv).view(1, Tk, ldkv).expand(B - B // 2, -1, -1)
            kv = torch.cat([kv[: B // 2], kvf])
        k = kv[:, :, : H * D].reshape(B, Tk, H, D).transpose(1, 2)
        v = kv[:, :, H * D: 2 * H * D].reshape(B, Tk, H, D).transpose(1, 2)
        bo = bias[mid:]
    sim = (q @ k.transpose(-1, -2)) * float(f[1])
    o = (sim.softmax(-1) @ v).transpose(1, 2).reshape(B, T, H * D)
    finish(o, outw(), bo)


_ACC_PERM = [16 * (2 * (k >> 5) + ((k & 7) >> 2)) + 4 * ((k >> 3) & 3) + (k & 3) for k in range(128)]


def _tf128(op, bufs: Buffers, B: int) -> None:
    """MDT_OP_TF128 semantics (include/mdt_hip.h): a whole Transformer1d of a 128-channel level, reconstructed from the tile
    stream by following the tile descriptors (projection tiles: K columns in accumulator order; output tiles: slot order)."""
    i, f = op.i, op.f
    C, T, NT, nvec = i[rt.F_C], i[rt.F_T], i[rt.F_NT], i[rt.F_NVEC]
    Tk, bs, ldkv, H = i[rt.F_TK], i[rt.F_KV_BSTRIDE], i[rt.F_LDKV], i[rt.F_HEADS]
    nblocks, nff, npost, cross = i[rt.F_NBLOCKS], i[rt.F_NFF], i[rt.F_NPOST], bool(i[rt.F_CROSS])
    wf32 = bool(i[rt.F_WF32])          # fp32 fragment tiles (exact-fp32 products) instead of bf16 hi / lo planes
    desc = bufs.view(op.p0, B, NT).contiguous().view(torch.int32).tolist()
    nw = sum(1 for d in desc if (d & 3) < 2 and not (d >> 22))          # (aux bit 20: the skip rows of a ResNet block)
    stream = bufs.view(op.w, B, nw * 64 * C)
    vec = bufs.view(op.bias, B, nvec)
    inv_acc = torch.empty(128, dtype=torch.long)
    inv_acc[torch.tensor(_ACC_PERM)] = torch.arange(128)
    inv_slot = torch.empty(64, dtype=torch.long)
    inv_slot[torch.tensor(_SLOT_PERM)] = torch.arange(64)
    cur = {"t": 0, "v": 0}

    def P():                       # next projection tile -> [64, C] in natural K order
        d = desc[cur["t"]]
        cur["t"] += 1
        assert d & 3 == 0, "expected a projection tile"
        return _untile(stream, d >> 2, 64, C, wf32)[:, inv_acc]

    def O(natural_from: Optional[int] = None):     # next output tile -> [C, 64]
        d = desc[cur["t"]]
        cur["t"] += 1
        assert d & 3 == 1, "expected an output tile"
        t = _untile(stream, d >> 2, C, 64, wf32)
        if natural_from is None:
            return t[:, inv_slot]
        cols = torch.tensor(_ACC_PERM[natural_from: natural_from + 64]) - natural_from     # accumulator order inside the chunk
        inv = torch.empty(64, dtype=torch.long)
        inv[cols] = torch.arange(64)
        return t[:, inv]

    def KV(kind, layer, h):
        d = desc[cur["t"]]
        cur["t"] += 1
        assert d & 3 == kind and (d >> 2) == (layer << 4 | h), "K / V tile descriptor out of order"

    def V(n):
        out = vec[cur["v"]: cur["v"] + n]
        cur["v"] += n
        return out

    x = bufs.view(op.a, B, B * T * C).view(B, T, C)
    # ---- ResnetBlock1d blocks in front of the transformer (MDT_F_RES_KIND) ----
    res_kind, n_res = i[rt.F_RES_KIND], i[rt.F_N_RES]
    if res_kind:
        film = bufs.view(op.p3, B, i[rt.F_NFILM])
        eps_r, s_b = float(f[rt.FF_EPS_RES]), float(f[rt.FF_SKIP_SCALE])
        gs1, gs2 = (32 if i[rt.F_RES_PAIR1] else 16), (32 if i[rt.F_RES_PAIR2] else 16)

        def conv3_w():                       # 6 projection tiles (tap, output half) -> [C, C, 3]
            w = torch.zeros(C, C, 3)
            for tap in range(3):
                for half in range(C // 64):
                    w[64 * half: 64 * half + 64, :, tap] = P()
            return w

        def gn_silu(t, gsize, gam, bet, fl=None):      # t [B, T, c] -> silu(GroupNorm [* (scale + 1) + shift])
            cc = t.shape[2]
            y = F.group_norm(t.transpose(1, 2), cc // gsize, gam, bet, eps_r).transpose(1, 2)
            if fl is not None:
                y = y * (fl[:cc] + 1.0) + fl[cc:]
            return _silu(y)

        def conv3(t, w):                     # Conv1d(k = 3, padding 1) inside the sample
            return F.conv1d(t.transpose(1, 2), w, None, padding=1).transpose(1, 2)

        for rb in range(n_res):
            fl = film[rb * 2 * C: (rb + 1) * 2 * C]
            if res_kind == 1:
                w1, w2 = conv3_w(), conv3_w()
                g1, b1, bias1, g2, b2, bias2 = (V(C) for _ in range(6))
                h = conv3(gn_silu(x, gs1, g1, b1), w1) + bias1
                x = conv3(gn_silu(h, gs2, g2, b2, fl), w2) + bias2 + x
                sk = rt.MdtRef(op.res.space, 0, op.res.off + rb * T * C)
                bufs.view(sk, B, B * T * C).view(B, T, C)[:] = x
            else:
                def skip_desc():
                    d = desc[cur["t"]]
                    cur["t"] += 1
                    assert d & 3 == 0 and (d >> 2) == ((1 << 20) | rb), "expected the skip rows of this block"
                sk = rt.MdtRef(op.res.space, 0, op.res.off - rb * T * C)
                xb = bufs.view(sk, B, B * T * C).view(B, T, C) * s_b
                wra = torch.cat([P() for _ in range(C // 64)])
                skip_desc()
                wrb = torch.cat([P() for _ in range(C // 64)])
                w1a = conv3_w()
                skip_desc()
                w1b, w2 = conv3_w(), conv3_w()
                g1, b1, bias1, br, g2, b2, bias2 = V(2 * C), V(2 * C), V(C), V(C), V(C), V(C), V(C)
                xc = gn_silu(torch.cat([x, xb], dim=2), gs1, g1, b1)
                h = conv3(xc[:, :, :C], w1a) + conv3(xc[:, :, C:], w1b) + bias1
                r = x @ wra.T + xb @ wrb.T + br
                x = conv3(gn_silu(h, gs2, g2, b2, fl), w2) + bias2 + r
    if i[rt.F_HAS_IN]:
        xn = F.group_norm(x.transpose(1, 2), 32, None, None, eps=float(f[2])).transpose(1, 2)
        w = torch.cat([P() for _ in range(C // 64)])
        x = xn @ w.T + V(C)
    mid, D = 64 * H, 64
    for blk in range(nblocks):
        xn = F.layer_norm(x, (C,), None, None, eps=float(f[0]))
        wq, wk, wv, wo = [], [], [], []
        for h in range(H):
            wq.append(P()), wk.append(P()), wv.append(P()), wo.append(O())
        bq, bo = V(mid), V(C)
        q = (xn @ torch.cat(wq).T + bq).view(B, T, H, D).transpose(1, 2)
        k = (xn @ torch.cat(wk).T).view(B, T, H, D).transpose(1, 2)
        v = (xn @ torch.cat(wv).T).view(B, T, H, D).transpose(1, 2)
        o = (((q @ k.transpose(-1, -2)) * float(f[1])).softmax(-1) @ v).transpose(1, 2).reshape(B, T, mid)
        x = x + o @ torch.cat(wo, dim=1).T + bo
        if cross:
            xn = F.layer_norm(x, (C,), None, None, eps=float(f[0]))
            wq, wo = [], []
            for h in range(H):
                wq.append(P()), KV(2, blk, h), KV(3, blk, h), wo.append(O())
            bq, bo = V(mid), V(C)
            lstride = i[rt.F_KV_LSTRIDE]
            if bs == 0:
                base = bufs.view(op.a2, B, (blk + 1) * lstride)[blk * lstride:]
                kv = base[: Tk * ldkv].view(1, Tk, ldkv).expand(B, -1, -1)
            else:
                base = bufs.view(op.a2, B, (blk + 1) * lstride * B)[blk * lstride * B:]
                kv = base.view(B, Tk, ldkv)
            if i[rt.F_KV2]:
                kvf = bufs.view(op.p1, B, (blk + 1) * lstride)[blk * lstride:][: Tk * ldkv].view(1, Tk, ldkv)
                kv = torch.cat([kv[: B // 2], kvf.expand(B - B // 2, -1, -1)])
            q = (xn @ torch.cat(wq).T + bq).view(B, T, H, D).transpose(1, 2)
            k = kv[:, :, :mid].reshape(B, Tk, H, D).transpose(1, 2)
            v = kv[:, :, mid: 2 * mid].reshape(B, Tk, H, D).transpose(1, 2)
            o = (((q @ k.transpose(-1, -2)) * float(f[1])).softmax(-1) @ v).transpose(1, 2).reshape(B, T, mid)
            x = x + o @ torch.cat(wo, dim=1).T + bo
        w1, w2 = [], []
        for h in range(nff):
            w1.append(P()), w2.append(O())
        b1, b2 = V(64 * nff), V(C)
        hdn = F.gelu(x @ torch.cat(w1).T + b1)
        if blk == nblocks - 1 and npost:
            wout = torch.cat([O(natural_from=64 * e) for e in range(npost)], dim=1)
            x = hdn @ torch.cat(w2, dim=1).T + x @ wout.T + b2
        else:
            x = x + hdn @ torch.cat(w2, dim=1).T + b2
    assert cur["t"] == NT, (cur["t"], NT)
    bufs.view(op.out, B, B * T * C).view(B, T, C)[:] = x


def _tf256(op, bufs: Buffers, B: int) -> None:
    """MDT_OP_TF256 semantics (include/mdt_hip.h): as _tf128 for a 256-channel level; the stream holds 32 KB SUB-tiles (K
    halves of projection tiles, row halves of output tiles), scratch descriptors close every sub-block, vectors are 768
    floats per sub-block.  MDT_F_NSPLIT = 2: two descriptor tables, half hh listing heads / hidden chunks / to_in output
    chunks / folded to_out k chunks [hh n/2, (hh + 1) n/2) and two more scratch descriptors per sub-block (the hand-off)."""
    i, f = op.i, op.f
    C, T, NT, nvec = i[rt.F_C], i[rt.F_T], i[rt.F_NT], i[rt.F_NVEC]
    Tk, bs, ldkv, H = i[rt.F_TK], i[rt.F_KV_BSTRIDE], i[rt.F_LDKV], i[rt.F_HEADS]
    nblocks, nff, npost, cross = i[rt.F_NBLOCKS], i[rt.F_NFF], i[rt.F_NPOST], bool(i[rt.F_CROSS])
    nsplit = 2 if i[rt.F_NSPLIT] == 2 else 1
    wf32 = bool(i[rt.F_WF32])
    desc_all = bufs.view(op.p0, B, nsplit * NT).contiguous().view(torch.int32).tolist()
    tabs = [desc_all[h * NT: (h + 1) * NT] for h in range(nsplit)]
    nw = 1 + max(d >> 3 for d in desc_all if (d & 7) < 2)
    stream = bufs.view(op.w, B, nw * 64 * 128)
    vec = bufs.view(op.bias, B, nvec)
    accp = torch.tensor([16 * (2 * (k >> 5) + ((k & 7) >> 2)) + 4 * ((k >> 3) & 3) + (k & 3) for k in range(C)])
    inv_acc = torch.empty(C, dtype=torch.long)
    inv_acc[accp] = torch.arange(C)
    inv_slot = torch.empty(64, dtype=torch.long)
    inv_slot[torch.tensor(_SLOT_PERM)] = torch.arange(64)
    cur = {"t": [0] * nsplit, "sb": 0}

    def tab(index, count):         # which table lists item `index` of `count`
        return index * nsplit // count

    def take(kind, hh=0):
        d = tabs[hh][cur["t"][hh]]
        cur["t"][hh] += 1
        assert d & 7 == kind, (hh, cur["t"][hh] - 1, d & 7, kind)
        return d >> 3

    def P(hh=0):                   # projection tile = two K-half sub-tiles -> [64, C], natural K order
        a_, b_ = take(0, hh), take(0, hh)
        return torch.cat([_untile(stream, a_, 64, 128, wf32), _untile(stream, b_, 64, 128, wf32)], dim=1)[:, inv_acc]

    def O(hh=0, natural_from: Optional[int] = None):     # output tile = two row-half sub-tiles -> [C, 64]
        a_, b_ = take(1, hh), take(1, hh)
        t = torch.cat([_untile(stream, a_, 128, 64, wf32), _untile(stream, b_, 128, 64, wf32)], dim=0)
        if natural_from is None:
            return t[:, inv_slot]
        cols = accp[natural_from: natural_from + 64] - natural_from
        inv = torch.empty(64, dtype=torch.long)
        inv[cols] = torch.arange(64)
        return t[:, inv]

    def end_subblock(last=False):
        nxt = cur["sb"] + 1
        for hh in range(nsplit):
            t0 = cur["t"][hh]
            d = tabs[hh][t0]
            if last:
                assert d & 7 == 4
            else:
                assert d & 7 == 5 and (d >> 3) == (((768 * nxt) // 256) << 1 | (nxt & 1)), "vector descriptor of the next sub-block"
            n_scr = 2 * nsplit                                   # wave-pair exchange (+ the hand-off's two barriers)
            assert all(tabs[hh][t0 + k] & 7 == 4 for k in range(1, n_scr))
            cur["t"][hh] += n_scr
        cur["sb"] = nxt

    def V(off, n):
        base = 768 * cur["sb"]
        return vec[base + off: base + off + n]

    x = bufs.view(op.a, B, B * T * C).view(B, T, C)
    if i[rt.F_HAS_IN]:
        xn = F.group_norm(x.transpose(1, 2), 32, None, None, eps=float(f[2])).transpose(1, 2)
        w = torch.cat([P(tab(ch, C // 64)) for ch in range(C // 64)])
        x = xn @ w.T + V(0, C)
        end_subblock()
    mid, D = 64 * H, 64
    for blk in range(nblocks):
        xn = F.layer_norm(x, (C,), None, None, eps=float(f[0]))
        wq, wk, wv, wo = [], [], [], []
        for h in range(H):
            hh = tab(h, H)
            wq.append(P(hh)), wk.append(P(hh)), wv.append(P(hh)), wo.append(O(hh))
        bq, bo = V(0, mid), V(mid, C)
        q = (xn @ torch.cat(wq).T + bq).view(B, T, H, D).transpose(1, 2)
        k = (xn @ torch.cat(wk).T).view(B, T, H, D).transpose(1, 2)
        v = (xn @ torch.cat(wv).T).view(B, T, H, D).transpose(1, 2)
        o = (((q @ k.transpose(-1, -2)) * float(f[1])).softmax(-1) @ v).transpose(1, 2).reshape(B, T, mid)
        x = x + o @ torch.cat(wo, dim=1).T + bo
        end_subblock()
        if cross:
            xn = F.layer_norm(x, (C,), None, None, eps=float(f[0]))
            wq, wo = [], []
            for h in range(H):
                hh = tab(h, H)
                wq.append(P(hh))
                assert take(2, hh) == (blk << 4 | h) and take(3, hh) == (blk << 4 | h)
                wo.append(O(hh))
            bq, bo = V(0, mid), V(mid, C)
            lstride = i[rt.F_KV_LSTRIDE]
            if bs == 0:
                base = bufs.view(op.a2, B, (blk + 1) * lstride)[blk * lstride:]
                kv = base[: Tk * ldkv].view(1, Tk, ldkv).expand(B, -1, -1)
            else:
                base = bufs.view(op.a2, B, (blk + 1) * lstride * B)[blk * lstride * B:]
                kv = base.view(B, Tk, ldkv)
            if i[rt.F_KV2]:
                kvf = bufs.view(op.p1, B, (blk + 1) * lstride)[blk * lstride:][: Tk * ldkv].view(1, Tk, ldkv)
                kv = torch.cat([kv[: B // 2], kvf.expand(B - B // 2, -1, -1)])
            q = (xn @ torch.cat(wq).T + bq).view(B, T, H, D).transpose(1, 2)
            k = kv[:, :, :mid].reshape(B, Tk, H, D).transpose(1, 2)
            v = kv[:, :, mid: 2 * mid].reshape(B, Tk, H, D).transpose(1, 2)
            o = (((q @ k.transpose(-1, -2)) * float(f[1])).softmax(-1) @ v).transpose(1, 2).reshape(B, T, mid)
            x = x + o @ torch.cat(wo, dim=1).T + bo
            end_subblock()
        w1, w2 = [], []
        for h in range(nff):
            w1.append(P(tab(h, nff))), w2.append(O(tab(h, nff)))
        b1, b2 = V(0, 64 * nff), V(64 * nff, C)
        hdn = F.gelu(x @ torch.cat(w1).T + b1)
        last = blk == nblocks - 1
        if last and npost:
            wout = torch.cat([O(tab(e, npost // 2), natural_from=64 * e) for e in range(npost // 2)], dim=1)
            x = hdn @ torch.cat(w2, dim=1).T + x @ wout.T + b2
        else:
            x = x + hdn @ torch.cat(w2, dim=1).T + b2
        end_subblock(last=last)
    assert all(t == NT for t in cur["t"]), (cur["t"], NT)
    bufs.view(op.out, B, B * T * C).view(B, T, C)[:] = x


def _res256(op, bufs: Buffers, B: int) -> None:
    """MDT_OP_RES256 semantics (include/mdt_hip.h): a chain of ResnetBlock1d blocks (reference modules.py:145-205, :828-829) of a
    256-channel level, reconstructed from the sub-tile stream by following the tile descriptors: sub-tiles [64][128] in (tap, K half,
    chunk) order, rows 0..31 = output channels 32 ch .., rows 32..63 = channels 128 + 32 ch .., K columns in accumulator order."""
    i, f = op.i, op.f
    C, T, NT, taps = i[rt.F_C], i[rt.F_T], i[rt.F_NT], i[rt.F_NPOST]
    kind, n_res = i[rt.F_RES_KIND], i[rt.F_N_RES]
    assert C == 256 and kind in (1, 2) and taps in (1, 3)
    segs = bufs.view(op.p0, B, i[rt.F_HEADS]).contiguous().view(torch.int32).tolist()    # SEGMENTS: a weight descriptor = a run
    desc, nw = [], 0
    for d in segs:
        if d & 3 == 0:
            desc += [0 | ((nw + k) << 2) for k in range(d >> 2)]
            nw += d >> 2
        else:
            desc.append(d)
    assert len(desc) == NT, (len(desc), NT)
    # NSPLIT = 2 (pair-split form): ONE descriptor table walked by both halves; half hh streams only output chunks 2 hh, 2 hh + 1 of
    # every convolution -- (tap, K half, chunk of the half) order, half 1's sub-tiles NFF sub-tiles behind half 0's (the hand-offs
    # between the two workgroups take no tile of the stream)
    split = i[rt.F_NSPLIT] == 2
    assert not split or nw == i[rt.F_NFF], (nw, i[rt.F_NFF])
    stream = bufs.view(op.w, B, (2 if split else 1) * nw * 64 * 128)
    vec = bufs.view(op.bias, B, i[rt.F_NVEC])
    film = bufs.view(op.p3, B, i[rt.F_NFILM])
    acc = torch.tensor([16 * (2 * (k >> 5) + ((k & 7) >> 2)) + 4 * ((k >> 3) & 3) + (k & 3) for k in range(C)])
    inv_acc = torch.empty(C, dtype=torch.long)
    inv_acc[acc] = torch.arange(C)
    eps_r, s_b = float(f[rt.FF_EPS_RES]), float(f[rt.FF_SKIP_SCALE])
    wf32 = bool(i[rt.F_WF32])          # fp32 fragment sub-tiles (exact-fp32 products) instead of bf16 hi / lo planes
    cur = {"t": 0, "v": 0}

    def expect(kind_, aux=None):
        d = desc[cur["t"]]
        cur["t"] += 1
        assert d & 3 == kind_ and (aux is None or (d >> 2) == aux), (cur["t"] - 1, d, kind_, aux)
        return d >> 2

    def X(first_of_block=None):                 # an exchange tile: scratch, or scratch + the NEXT block's vectors at a block's first
        d = desc[cur["t"]]
        cur["t"] += 1
        if first_of_block is not None and first_of_block + 1 < n_res:
            assert d & 3 == 3 and (d >> 2) == first_of_block + 1, "a block's first exchange carries the next block's vectors"
        else:
            assert d & 3 == 2, (cur["t"] - 1, d)

    def conv_w(k):                               # k taps of sub-tiles -> [C, C, k] in natural channel order
        w = torch.zeros(C, C, k)
        for tap in range(k):
            for kh in range(2):
                for cl in range(2 if split else 4):
                    idx = expect(0)
                    for hh in range(2 if split else 1):
                        ch = 2 * hh + cl if split else cl
                        t = _untile(stream, idx + hh * nw, 64, 128, wf32)
                        rows = torch.cat([torch.arange(32 * ch, 32 * ch + 32), torch.arange(128 + 32 * ch, 128 + 32 * ch + 32)])
                        w[rows[:, None], acc[128 * kh: 128 * kh + 128][None, :], tap] = t
        return w

    def V(n):
        out = vec[cur["v"]: cur["v"] + n]
        cur["v"] += n
        return out

    def gn_silu(t, gsize, gam, bet, fl=None):
        cc = t.shape[2]
        y = F.group_norm(t.transpose(1, 2), cc // gsize, gam, bet, eps_r).transpose(1, 2)
        if fl is not None:
            y = y * (fl[:cc] + 1.0) + fl[cc:]
        return _silu(y)

    def conv(t, w):                              # Conv1d with `taps` live taps (one token per sample: the centre tap alone)
        if w.shape[2] == 1:
            return t @ w[:, :, 0].T
        return F.conv1d(t.transpose(1, 2), w, None, padding=1).transpose(1, 2)

    x = bufs.view(op.a, B, B * T * C).view(B, T, C).clone()
    expect(3, 0)                                 # the vector tile of block 0
    for rb in range(n_res):
        fl = film[rb * 2 * C: (rb + 1) * 2 * C]
        if kind == 1:
            X(rb)
            w1 = conv_w(taps)
            X()
            w2 = conv_w(taps)
            g1, b1, bias1, g2, b2, bias2 = (V(C) for _ in range(6))
            h = conv(gn_silu(x, 32, g1, b1), w1) + bias1
            x = conv(gn_silu(h, 32, g2, b2, fl), w2) + bias2 + x
            sk = rt.MdtRef(op.res.space, 0, op.res.off + rb * T * C)
            bufs.view(sk, B, B * T * C).view(B, T, C)[:] = x
        else:
            sk = rt.MdtRef(op.res.space, 0, op.res.off - rb * T * C)
            xb = bufs.view(sk, B, B * T * C).view(B, T, C) * s_b
            X(rb)
            w1a = conv_w(taps)
            X()
            wra = conv_w(1)
            expect(1, rb)
            X()
            wrb = conv_w(1)
            expect(1, rb)
            X()
            w1b = conv_w(taps)
            X()
            w2 = conv_w(taps)
            g1, b1, bias1, br, g2, b2, bias2 = V(2 * C), V(2 * C), V(C), V(C), V(C), V(C), V(C)
            xc = gn_silu(torch.cat([x, xb], dim=2), 64, g1, b1)
            h = conv(xc[:, :, :C], w1a) + conv(xc[:, :, C:], w1b) + bias1
            r = x @ wra[:, :, 0].T + xb @ wrb[:, :, 0].T + br
            x = conv(gn_silu(h, 32, g2, b2, fl), w2) + bias2 + r
    assert cur["t"] == NT and cur["v"] == vec.numel(), (cur, NT, vec.numel())
    bufs.view(op.out, B, B * T * C).view(B, T, C)[:] = x
