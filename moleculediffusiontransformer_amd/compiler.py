"""Lowers the 1-D conditional U-Net (UNet1d.forward, modules.py:1144-1180) to mdt_op programs.

Input: a UNetConfig, the sequence length, the conditioning length and a reference-format
state_dict (keys relative to the U-Net, e.g. ``downsamples.0.blocks.1.block1.project.weight``).
Output (CompiledUNet): one packed fp32 weight buffer plus five programs

  time        rows = all timesteps of a sampling call: LearnedPositionalEmbedding -> to_time -> to_mapping
              -> every ResnetBlock1d's MappingToScaleShift in ONE GEMM   (modules.py:545-566, :996-1010, :125-142)
  ctx         per call: cross-attention norm_context + to_kv on the conditioning embedding for every
              cross-attention layer (the context never changes inside DiffusionSampler.forward, diffusion.py:587)
  ctx_fixed   the same on UNetCFG1d's FixedEmbedding (batch-invariant; modules.py:1239, :1251)
  eval        one U-Net evaluation for the batch (conditional)
  eval_fixed  the same attending to the fixed embedding (classifier-free-guidance second pass)

Activations are token-major (B, L, Cp) with channels padded to a multiple of 16; every conv/linear is
an MDT_OP_GEMM with fused prologue/epilogue (see csrc/k_gemm.hip).  Buffers live at per-sample offsets
in one arena that is scaled by the batch size at run time, so a compiled model serves any batch size.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch

from . import runtime as rt
from .netspec import UNetConfig

EXT_XIN, EXT_CTX, EXT_OUT = 0, 1, 2      # bindings.ext slots used by the programs
EXT_XFLAGS, EXT_XBUF = 3, 4              # pair-split MDT_OP_TF256: hand-off flags / blocks (engine.py sizes them per batch)


def pad16(c: int) -> int:
    return (c + 15) // 16 * 16


@dataclass
class Ten:
    """A token-major tensor: `rows` per sample, `ld` floats per row, at `off` in `space`."""
    space: int
    off: int
    rows: int
    ld: int
    c: int = 0           # real (unpadded) channel count
    b16: bool = False    # bf16 elements (ld counts ELEMENTS; the allocation is rows * ld / 2 floats): plain-bf16 mode only

    def ref(self) -> rt.MdtRef:
        return rt.MdtRef(self.space, 0, self.off)


def _ref(space: int = 0, off: int = 0) -> rt.MdtRef:
    return rt.MdtRef(space, 0, off)


class _Arena:
    """First-fit allocator over per-sample float offsets (64-float granules)."""

    def __init__(self):
        self.free: List[Tuple[int, int]] = []   # (off, size)
        self.top = 0

    def alloc(self, n: int) -> int:
        n = (n + 63) // 64 * 64
        for i, (o, s) in enumerate(self.free):
            if s >= n:
                if s == n:
                    self.free.pop(i)
                else:
                    self.free[i] = (o + n, s - n)
                return o
        o = self.top
        self.top += n
        return o

    def release(self, off: int, n: int) -> None:
        n = (n + 63) // 64 * 64
        self.free.append((off, n))
        self.free.sort()
        merged: List[Tuple[int, int]] = []
        for o, s in self.free:
            if merged and merged[-1][0] + merged[-1][1] == o:
                merged[-1] = (merged[-1][0], merged[-1][1] + s)
            else:
                merged.append((o, s))
        self.free = merged


class _Weights:
    def __init__(self):
        self.chunks: List[torch.Tensor] = []
        self.n = 0
        self.index: Dict[str, int] = {}
        self.starts: Dict[int, torch.Tensor] = {}      # offset -> the tensor added there

    def read(self, off: int, n: int) -> torch.Tensor:
        """The first n floats of the tensor that was added at offset `off`."""
        t = self.starts[off]
        assert t.numel() >= n, (off, n, t.numel())
        return t[:n].clone()

    def add(self, name: str, t: torch.Tensor) -> int:
        t = t.detach().to(torch.float32).contiguous().reshape(-1).cpu()
        off = self.n
        pad = (-t.numel()) % 64
        self.chunks.append(t)
        self.starts[off] = t
        if pad:
            self.chunks.append(torch.zeros(pad))
        self.n += t.numel() + pad
        self.index[name] = off
        return off

    def pack(self) -> torch.Tensor:
        return torch.cat(self.chunks) if self.chunks else torch.zeros(0)


@dataclass
class CompiledUNet:
    cfg: UNetConfig
    length: int
    cond_len: int
    in_pad: int                      # padded in/out channels of the token-major U-Net input/output
    weights: torch.Tensor            # packed fp32 (CPU); moved to the device by the runtime
    programs: Dict[str, List[rt.MdtOp]]
    act_floats: int                  # per-sample arena size
    shr_floats: int                  # batch-invariant arena size for `max_time_rows`
    max_time_rows: int
    shr: Dict[str, int]              # named offsets in the shared arena
    ss_total: int                    # floats of one row of all (scale, shift) vectors
    n_cross: int
    flops_per_sample_eval: int       # 2*MACs of one conditional U-Net evaluation (dense contractions + attention)
    flops_ctx_per_sample: int
    gemm_mode: str = "bf16x3"
    weight_index: Dict[str, int] = field(default_factory=dict)
    # "eval_dual" (both guidance passes as one doubled batch): a cross-attention workgroup serves 64 / T (C = 128 ring kernel)
    # or 32 / T (C = 256) samples and picks conditional vs fixed K/V per WORKGROUP, so the number of samples must be a
    # multiple of the largest such group or a workgroup would straddle the two halves
    dual_multiple: int = 1
    tf256: bool = False              # compiled with the whole-transformer form of the 256-channel level (generative._wide)
    xchg_tokens: int = 0             # > 0: pair-split MDT_OP_TF256 ops present; tokens per sample of the largest (engine.py)


class UNetCompiler:
    def __init__(self, cfg: UNetConfig, length: int, cond_len: int, sd: Dict[str, torch.Tensor],
                 max_time_rows: int = 1024, gemm_mode: str = "bf16x3", fuse_blocks: bool = True, tf256: bool = False):
        self.fuse_blocks = fuse_blocks
        # (Round 6: the second-order switches MDT_FUSE_C256 / MDT_TB_SPLIT / MDT_FF_SPLIT / MDT_TB_CHAIN / MDT_FUSE_CROSS /
        #  MDT_CONVT_MERGE / MDT_GN_ACT -- fallbacks OF fallbacks that no default program reached and no test flipped -- are gone
        #  with their branches: the C = 256 sub-block launches always split heads / hidden chunks over two workgroups and chain
        #  their partial sums, the ConvTranspose phases of the GEMM form are one launch, GroupNorm-apply is k_gn_act where it fits.)
        self.use_rconv = os.environ.get("MDT_RCONV", "1") == "1"     # row-stationary convs (k_rconv) at C = 128 / 256
        self.use_resblock = os.environ.get("MDT_RESBLOCK", "1") == "1"   # Patcher / Unpatcher ResNets as ONE launch (k_resblock)
        self.t1_fold = os.environ.get("MDT_T1_FOLD", "1") == "1"   # self-attention over one token per sample as one folded GEMM
        self.ctx_split = os.environ.get("MDT_CTX_SPLIT", "1") == "1"   # k_attn_ctx: split-bf16 scores in the split-bf16 mode
        # Transformer1d's closing 1x1 convolution folded into its last feed-forward block (ring kernels only)
        self.fold_out = os.environ.get("MDT_FOLD_OUT", "1") == "1"
        self.rconv_two = os.environ.get("MDT_RCONV2", "k1") == "1"   # concatenated inputs as ONE two-source launch ...
        self.rconv_two_k1 = os.environ.get("MDT_RCONV2", "k1") in ("1", "k1")   # ... only their 1x1 residual convolution
        # (cross-attention sub-blocks fuse where their K/V rows stream through the loader-wave ring: k_tblock_lw: C = 128, at most
        #  16 context rows per 16 token rows; k_tblock32: C = 256, at most 48)
        # cross-attention layers that run layer by layer over MANY keys (QMDiffusionForward: 64): fold the key / value
        # projections into the query / output projections and attend to the normalised context itself
        # ResNet blocks of a 128-channel level inside the transformer launch that follows them (k_tf128 RES = 1 / 2)
        self.res128 = os.environ.get("MDT_RES128", "1") == "1"
        # MDT_RES256: the 256-channel level's ResNet blocks as chained launches (MDT_OP_RES256, csrc/k_res256.hip) -- auto (default):
        # the unsplit chain in the wide program (+2.1 ... +3.3 % same-box at B = 2048 / 8192 / guidance, round 5) and the PAIR-SPLIT
        # chain (round 6, res256_split()) in the narrow one; 1: the unsplit chain always; 0: never (one k_rconv launch per convolution);
        # whole: round 5's policy (unsplit chain in the wide program and on one-token levels, k_rconv launches at configs[1]'s
        # B = 1024 -- there the per-convolution launches, which split the output channels over two workgroups, were 0.4 % faster than
        # the unsplit chain on half the compute units, profiles/r5_res256_ab.txt); split: the pair-split chain always
        self.res256_mode = os.environ.get("MDT_RES256", "auto")
        self.patch_conv = os.environ.get("MDT_PATCH_CONV", "1") != "0"   # resampling convolutions in patch form on k_rconv
        self.fold_patch = os.environ.get("MDT_FOLD_PATCH", "1") != "0"   # Patcher / Unpatcher rearranges folded into k_resblock
        self.use_proj = os.environ.get("MDT_PROJ", "1") != "0"      # K = 128 / 256 projections on ring tiles (k_proj.hip)
        self.b16 = os.environ.get("MDT_B16", "1") == "1"
        # plain-bf16 mode: the TRANSFORMER blocks' residual stream as ONE bf16 tensor (round 6): residual, output and the next GEMM's A
        # operand at once -- no fp32 read + write of the stream per projection, no conversion pass in front of the feed-forward and
        # cross-attention GEMMs.  Priced on the oracle at 2e-4 .. 5e-4 of the final sample (tools/res16_experiment.py; the mode's bf16
        # operands cost 1.2e-3 .. 1.6e-3, its budget is 1e-2).  MDT_RES16=0: the fp32 stream of rounds 3-5.
        self.res16 = os.environ.get("MDT_RES16", "1") == "1"
        # ... and the LayerNorm in front of the attention projections FOLDED into the GEMM (second half of round 6, MDT_G_WFMT 134): the
        # projection reads the RAW stream, gathers the rows' statistics from the fragments it multiplies and scales / shifts its
        # accumulators per row -- no MDT_OP_PREP16 pass in front of it.  MDT_LNFOLD=0: the PREP16 pass.
        self.lnfold = os.environ.get("MDT_LNFOLD", "1") == "1"
        # ... and the up path's cat([x, skip]) never written in that mode: block1's GroupNorm pass reads both sources and leaves the raw
        # bf16 copy for the to_out convolution (MDT_OP_GN_ACT with a2 / p2).  MDT_CAT_FOLD=0: k_concat + a conversion pass.
        self.cat_fold = os.environ.get("MDT_CAT_FOLD", "1") == "1"
        self.qkv_merge = os.environ.get("MDT_QKV_MERGE", "1") == "1"   # ... and self-attention's q | k | v as one GEMM     # bf16 mode: regular layers as PREP16 + bf16 x bf16 GEMM
        self.fold_ctx = os.environ.get("MDT_FOLD_CTX", "1") == "1"
        self.has_chat = False                # some layer attends to the normalised context (ctx program emits it)
        self.tf128 = os.environ.get("MDT_TF128", "1") == "1"         # a whole C = 128 Transformer1d as ONE launch (k_tf128)
        # ... and a whole C = 256 one (k_tf256, 32-row workgroups, no head split).  Measured at B = 1024 (128 workgroups): 1.38 ms
        # for the five transformers against 1.25 ms as head-split launches (both bound by the per-CU weight stream); it wins
        # once the batch fills the chip without the split, so it is a per-batch choice (engine: program "eval_wide")
        self.tf256 = bool(tf256)
        # ... and where the batch does NOT fill the chip that way: the same launch with every row block's heads split over a
        # PAIR of workgroups that hand each other their partial sums inside the launch (k_tf256 NSPLIT = 2).
        # MDT_TF256_PAIR=0: one head-split launch per sub-block (k_tblock32), the form of rounds 1-2.
        self.tf256_pair = os.environ.get("MDT_TF256_PAIR", "1") == "1"
        self.pair_stride = int(os.environ.get("MDT_PAIR_STRIDE", "8"))
        self.xchg_tokens = 0                 # max tokens per sample over the pair-split ops (sizes the hand-off buffers)
        if gemm_mode not in ("f32", "bf16x3", "bf16"):
            raise ValueError("gemm_mode must be 'f32' (exact fp32 MFMA), 'bf16x3' (split-bf16 MFMA, fp32-class) or 'bf16' "
                             "(plain bf16 products, reduced precision: layer-by-layer GEMMs only)")
        self.gemm_mode = gemm_mode
        # Exact-fp32 mode on the SAME fused program as the default mode: the ring kernels (k_tf128, k_tf256, k_rconv,
        # k_resblock) take fp32 fragment tiles and form every product with v_mfma_f32_16x16x4_f32; everything else (resampling
        # convolutions, time / context programs) is the exact fp32 MFMA GEMM as before.  MDT_F32_FUSED=0: the layer-by-layer
        # program of rounds 1-3 (k_gemm + k_attn + k_gn_act), kept as the second exact implementation the tests compare with.
        self.wf32 = gemm_mode == "f32" and os.environ.get("MDT_F32_FUSED", "1") == "1"
        self.ring_mode = gemm_mode == "bf16x3" or self.wf32        # may the ring kernels be used at all?
        self._packed: Dict = {}
        self._zeros_off, self._zeros_len = 0, 0
        if cfg.channels % 16:
            raise ValueError("channels must be a multiple of 16")
        if length % (cfg.patch_size * _prod(cfg.factors)):
            raise ValueError("max_length must be divisible by patch_size * prod(factors)")
        if cfg.head_features != 64:
            raise ValueError("attention kernel is specialised for head_features == 64")
        if cfg.ctx_features % 16:
            raise ValueError("context features must be a multiple of 16")
        self.cfg, self.L, self.n_ctx, self.sd = cfg, length, cond_len, sd
        self.W = _Weights()
        self.arena = _Arena()
        self._ones_off: Dict[int, int] = {}
        self.ops: List[rt.MdtOp] = []
        self.flops = 0
        self.max_time_rows = max_time_rows
        self.kv_slots: List[int] = []        # ACT offsets of the hoisted K/V per cross-attention layer
        self.kv_fixed: List[int] = []        # SHR offsets for the fixed embedding
        self.cross_layers: List[str] = []    # key prefixes, in evaluation order
        self.ss_offsets: Dict[str, int] = {}
        self.ss_total = 0
        self.shr_top = 0
        self.shr: Dict[str, int] = {}

    # ------------------------------------------------------------------ helpers
    def _shr_alloc(self, name: str, n: int) -> int:
        off = self.shr_top
        self.shr_top += (n + 63) // 64 * 64
        self.shr[name] = off
        return off

    def _new(self, rows: int, ld: int, c: int = 0) -> Ten:
        return Ten(rt.SP_ACT, self.arena.alloc(rows * ld), rows, ld, c or ld)

    def _new16(self, rows: int, n: int) -> Ten:
        """A bf16 tensor (the A operand of a bf16 x bf16 GEMM)."""
        return Ten(rt.SP_ACT, self.arena.alloc(rows * n // 2), rows, n, n, True)

    def _free(self, t: Ten) -> None:
        if t.space == rt.SP_ACT:
            self.arena.release(t.off, t.rows * t.ld // (2 if t.b16 else 1))

    def b16_ok(self, cin: int) -> bool:
        """Plain-bf16 mode: is a regular layer with `cin` input channels lowered to the bf16 x bf16 GEMM?"""
        return self.gemm_mode == "bf16" and self.b16 and cin % 64 == 0

    def _ones(self, n: int) -> int:
        if n not in self._ones_off:
            self._ones_off[n] = self.W.add(f"ones{n}", torch.ones(n))
        return self._ones_off[n]

    def _zeros(self, n: int) -> int:
        if n > self._zeros_len:
            self._zeros_off = self.W.add(f"zeros{n}", torch.zeros(n))
            self._zeros_len = n
        return self._zeros_off

    def _vec(self, key: str, pad_to: int) -> int:
        v = self.sd[key]
        out = torch.zeros(pad_to)
        out[: v.numel()] = v
        return self.W.add(key, out)

    def _lin_w(self, key: str, n_pad: Optional[int] = None, k_pad: Optional[int] = None):
        w = self.sd[key]                       # [N, K]
        n, k = w.shape
        out = torch.zeros(n_pad or pad16(n), k_pad or pad16(k))
        out[:n, :k] = w
        return (key, out)

    def _conv_w(self, key: str, cin_pad: int, n_pad: int):
        w = self.sd[key]                       # [Cout, Cin, k] -> [Np][k][Cin_p]
        co, ci, k = w.shape
        out = torch.zeros(n_pad, k, cin_pad)
        out[:co, :, :ci] = w.permute(0, 2, 1)
        return (key, out.reshape(n_pad, k * cin_pad))

    def _pack_w(self, wt, cin: int):
        """Packs a GEMM weight [N][K] once.  Returns (offset, lo_offset or None): exact fp32, or the two bf16
        planes of the split-bf16 kernel (value rounded to nearest-even bf16, then its residual)."""
        name, w = wt
        split = self.gemm_mode == "bf16x3" and cin % 32 == 0
        plain = self.gemm_mode == "bf16" and cin % 32 == 0
        key = (name, split, plain)
        if key not in self._packed:
            if plain:        # one bf16 plane; the third element marks the format (MDT_G_WFMT = 1)
                hi = w.to(torch.bfloat16)
                self._packed[key] = (self.W.add(name + "/bf16", hi.contiguous().view(-1).view(torch.float32)), None, 1)
            elif split:
                hi = w.to(torch.bfloat16)
                lo = (w - hi.float()).to(torch.bfloat16)
                self._packed[key] = (self.W.add(name + "/bf16_hi", hi.contiguous().view(-1).view(torch.float32)),
                                     self.W.add(name + "/bf16_lo", lo.contiguous().view(-1).view(torch.float32)))
            else:
                self._packed[key] = (self.W.add(name, w), None)
        return self._packed[key]

    def _emit(self, op: rt.MdtOp) -> None:
        self.ops.append(op)

    def gemm(self, a: Ten, wt, n: int, out: Ten, *, cin: int, bias_off: Optional[int] = None,
             taps: int = 1, t_stride: int = 1, t_dj: int = 0, t_off: int = 0, r_out: Optional[int] = None,
             o_stride: int = 1, o_off: int = 0, res: Optional[Ten] = None, pro: int = rt.PRO_NONE,
             gain: Optional[int] = None, nbias: Optional[int] = None, stats: Optional[Ten] = None,
             film: Optional[rt.MdtRef] = None, groups: int = 0, gsize: int = 0, pro_silu: int = 0,
             act: int = 0, eps: float = 0.0, m_mode: int = 0, a_col: int = 0, o_col: int = 0,
             count_flops: bool = True, phases: int = 0, copy16: Optional[Ten] = None) -> None:
        r_out_ = a.rows if r_out is None else r_out
        a16 = None
        regular = (self.b16_ok(cin) and t_stride == 1 and phases <= 1 and o_stride == 1 and o_off == 0 and r_out_ == a.rows
                   and out.rows == r_out_ and m_mode == 0)
        assert not (a.b16 or out.b16) or (regular and (not a.b16 or (pro in (rt.PRO_NONE, rt.PRO_LAYERNORM) and a_col == 0))), "bf16 operand"
        lnf = None
        if (self.lnfold and regular and a.b16 and pro == rt.PRO_LAYERNORM and taps == 1 and t_off == 0 and out.b16 and res is None
                and copy16 is None and n % 8 == 0 and o_col % 8 == 0 and out.ld % 8 == 0):
            # LayerNorm folded around the GEMM (MDT_G_WFMT 134): W (g xn + b) = rstd ((W g) x - mean rowsum(W g)) + W b, fp64 on the host;
            # the row sums are taken over the bf16 values the MFMAs multiply
            key = (wt[0], "lnfold", gain, nbias, bias_off)
            if key not in self._packed:
                w = wt[1].double()
                gv, bv = self.W.read(gain, cin).double(), self.W.read(nbias, cin).double()
                b0 = self.W.read(bias_off, n).double() if bias_off is not None else torch.zeros(n, dtype=torch.float64)
                w2 = (w * gv.unsqueeze(0)).float()
                csum = w2.to(torch.bfloat16).double().sum(dim=1).float()
                self._packed[key] = ((wt[0] + "/lnfold", w2), self.W.add(wt[0] + "/lnfold.bias", (b0 + w @ bv).float()),
                                     self.W.add(wt[0] + "/lnfold.csum", csum))
            wt, bias_off, csum_off = self._packed[key]
            lnf = csum_off
            pro, gain, nbias = rt.PRO_NONE, None, None
        if regular and (not a.b16 or pro == rt.PRO_LAYERNORM):
            # plain-bf16 mode, regular layer: the prologue runs once per element in a pass of its own that writes the bf16 A
            # operand (MDT_OP_PREP16), the GEMM streams both operands by LDS-DMA (k_gemm_b16.hip)
            a16 = self._new16(a.rows, cin)
            pre = rt.MdtOp()
            pre.kind = rt.OP_PREP16
            pre.a, pre.out = a.ref(), a16.ref()
            if gain is not None:
                pre.p0 = _ref(rt.SP_WEIGHT, gain)
            if nbias is not None:
                pre.p1 = _ref(rt.SP_WEIGHT, nbias)
            if stats is not None:
                pre.p2 = stats.ref()
            if isinstance(film, tuple):
                pre._film = film
            elif film is not None:
                pre.p3 = film
            elif pro == rt.PRO_GROUPNORM:
                pre.p3 = _ref(rt.SP_WEIGHT, self._zeros(2 * cin))
            pi = pre.i
            pi[rt.G_R_IN], pi[rt.G_LDA], pi[rt.G_CIN], pi[rt.G_A_COL] = a.rows, a.ld, cin, a_col
            pi[rt.G_PRO], pi[rt.G_GROUPS], pi[rt.G_GSIZE], pi[rt.G_PRO_SILU] = pro, groups, gsize, pro_silu
            if a.b16:                            # LayerNorm of the bf16 residual stream (MDT_OP_PREP16 with a bf16 input)
                assert pro == rt.PRO_LAYERNORM and cin <= 1024 and a.ld % 8 == 0
                pi[rt.G_WFMT] = 2
            pre.f[0] = eps
            self._emit(pre)
            a = a16
            pro, gain, nbias, stats, film, groups, gsize, pro_silu, a_col = rt.PRO_NONE, None, None, None, None, 0, 0, 0, 0
        # K = 1024 -> C outputs (configs[2]'s output projections behind the 8 x 128 attention rows): k_rconv with the input's 1024 / C
        # channel blocks as sources accumulating into one output (MDT_R_KSRC)
        if (self.use_proj and self.use_rconv and self.ring_mode and self.fuse_blocks and (self.gemm_mode == "bf16x3" or self.wf32)
                and n in (128, 256) and cin == 1024 and taps == 1 and t_stride == 1 and t_off == 0 and phases <= 1 and o_stride == 1
                and o_off == 0 and r_out_ == a.rows and out.rows == r_out_ and act == 0 and pro == rt.PRO_NONE and m_mode == 0
                and not a.b16 and not out.b16 and copy16 is None and a_col == 0 and o_col == 0 and a.ld % 4 == 0 and out.ld % 4 == 0
                and a.rows in (1, 2, 4, 8, 16) and (res is None or res.ld % 4 == 0) and tuple(wt[1].shape) == (n, cin)):
            ksrc = cin // n
            key = (wt[0], "kblocks", self.wf32)
            if key not in self._packed:
                w = wt[1]
                tiles = [self._wtile(w[64 * ch: 64 * ch + 64, s_ * n + 128 * kh: s_ * n + 128 * kh + 128])
                         for s_ in range(ksrc) for kh in range(n // 128) for ch in range(n // 64)]
                self._packed[key] = (self.W.add(wt[0] + "/kblocks.tiles", torch.cat(tiles)), sum(t.numel() for t in tiles) * 4 // 1024)
            w_off, kb = self._packed[key]
            op = rt.MdtOp()
            op.kind = rt.OP_RCONV
            op.a, op.out, op.w = a.ref(), out.ref(), _ref(rt.SP_WEIGHT, w_off)
            op.i[rt.W_KB] = kb
            if bias_off is not None:
                op.bias = _ref(rt.SP_WEIGHT, bias_off)
            if res is not None:
                op.res = res.ref()
            i = op.i
            i[rt.R_T], i[rt.R_C], i[rt.R_LDA], i[rt.R_LDC], i[rt.R_TAPS] = a.rows, n, a.ld, out.ld, 1
            i[rt.R_LDR] = res.ld if res is not None else 0
            i[rt.R_FILM_LD], i[rt.R_WF32], i[rt.R_KSRC] = n, int(self.wf32), ksrc
            op.f[0], op.f[1] = 1e-5, 1.0
            self._emit(op)
            if count_flops:
                self.flops += 2 * a.rows * n * cin
            return
        op = rt.MdtOp()
        op.kind = rt.OP_GEMM
        # row-stationary projection on ring tiles (k_proj.hip, MDT_G_WFMT = 16): the K = 128 / 256 layers between the fused kernels
        ring = (self.use_proj and (self.gemm_mode == "bf16x3" or self.wf32) and cin in (128, 256) and n % 64 == 0 and n <= 2048 and taps == 1
                and t_stride == 1 and t_off == 0 and phases <= 1 and o_stride == 1 and o_off == 0 and r_out_ == a.rows
                and out.rows == r_out_ and act == 0 and pro in (rt.PRO_NONE, rt.PRO_LAYERNORM) and not a.b16 and not out.b16
                and copy16 is None and a.ld % 4 == 0 and a_col % 4 == 0 and o_col % 4 == 0 and out.ld % 4 == 0
                and (res is None or res.ld % 4 == 0) and tuple(wt[1].shape) == (n, cin))
        if ring:
            # LayerNorm's gain folds into the weights and its bias into the bias (W (g xn + b) = (W g) xn + W b, fp64 on the host): the
            # kernel normalises without per-channel vectors (48 float4 per lane ahead of its first barrier at K = 256 otherwise)
            fold = pro == rt.PRO_LAYERNORM
            key = (wt[0], "ring", self.wf32, gain if fold else None, nbias if fold else None, bias_off if fold else None)
            if key not in self._packed:
                w = wt[1].double()
                fb = None
                if fold:
                    gv = self.W.read(gain, cin).double()
                    bv = self.W.read(nbias, cin).double()
                    b0 = self.W.read(bias_off, n).double() if bias_off is not None else torch.zeros(n, dtype=torch.float64)
                    fb = self.W.add(wt[0] + "/ring.bias", (b0 + w @ bv).float())
                    w = w * gv.unsqueeze(0)
                w = w.float()
                tiles = [self._wtile(w[64 * c: 64 * c + 64, 128 * h: 128 * h + 128]) for c in range(n // 64) for h in range(cin // 128)]
                self._packed[key] = (self.W.add(wt[0] + "/ring.tiles", torch.cat(tiles)), None, 17 if self.wf32 else 16, fb)
            w_off, wlo_off, fmt16, fb = self._packed[key]
            wfmt = [fmt16]
            if fold:
                bias_off, gain, nbias = fb, None, None
        else:
            w_off, wlo_off, *wfmt = self._pack_w(wt, cin)
        op.a, op.w, op.out = a.ref(), _ref(rt.SP_WEIGHT, w_off), out.ref()
        op.i[rt.G_WFMT] = ((2 | (4 if out.b16 else 0) | (8 if copy16 is not None else 0)) if a.b16 else wfmt[0]) if wfmt else 0
        if res is not None and res.b16:      # the bf16 residual stream: A, residual and output all bf16 (MDT_G_WFMT 38)
            assert a.b16 and out.b16 and copy16 is None and o_col == 0
            op.i[rt.G_WFMT] = 38
        if lnf is not None:                  # LayerNorm of the raw bf16 A rows folded into the GEMM (MDT_G_WFMT 134)
            assert op.i[rt.G_WFMT] == 6
            op.i[rt.G_WFMT] = 134
        if copy16 is not None:               # a bf16 copy of the fp32 output, written by the epilogue (MDT_G_WFMT 10)
            assert a.b16 and copy16.b16 and not out.b16 and o_col == 0 and copy16.ld == n and copy16.rows == out.rows
            op.p0 = copy16.ref()
        if wlo_off is not None:
            op.a2 = _ref(rt.SP_WEIGHT, wlo_off)
        if bias_off is not None:
            op.bias = _ref(rt.SP_WEIGHT, bias_off)
        if res is not None:
            op.res = res.ref()
        if gain is not None:
            op.p0 = _ref(rt.SP_WEIGHT, gain)
        if nbias is not None:
            op.p1 = _ref(rt.SP_WEIGHT, nbias)
        if stats is not None:
            op.p2 = stats.ref()
        if lnf is not None:
            op.p0 = _ref(rt.SP_WEIGHT, lnf)
        if isinstance(film, tuple):      # ("ss", offset inside the shared scale/shift row): resolved in build()
            op._film = film
        elif film is not None:
            op.p3 = film
        elif pro == rt.PRO_GROUPNORM:    # no FiLM: scale = shift = 0 (x * (0 + 1) + 0 is exact)
            op.p3 = _ref(rt.SP_WEIGHT, self._zeros(2 * cin))
        r_out = a.rows if r_out is None else r_out
        i = op.i
        i[rt.G_R_OUT], i[rt.G_R_IN], i[rt.G_LDA], i[rt.G_CIN], i[rt.G_TAPS] = r_out, a.rows, a.ld, cin, taps
        i[rt.G_T_STRIDE], i[rt.G_T_DJ], i[rt.G_T_OFF] = t_stride, t_dj, t_off
        i[rt.G_N], i[rt.G_LDC], i[rt.G_O_ROWS], i[rt.G_O_STRIDE], i[rt.G_O_OFF] = n, out.ld, out.rows, o_stride, o_off
        i[rt.G_LDR] = res.ld if res is not None else 0
        i[rt.G_PRO], i[rt.G_GROUPS], i[rt.G_GSIZE], i[rt.G_PRO_SILU] = pro, groups, gsize, pro_silu
        i[rt.G_ACT], i[rt.G_M_MODE], i[rt.G_A_COL], i[rt.G_O_COL] = act, m_mode, a_col, o_col
        i[rt.G_PHASES] = phases
        op.f[0] = eps
        self._emit(op)
        if count_flops:
            self.flops += 2 * r_out * n * taps * cin * max(phases, 1)
        if a16 is not None:
            self._free(a16)

    def gn_stats(self, x: Ten, groups: int, gsize: int, eps: float) -> Ten:
        st = self._new(1, 2 * groups)
        op = rt.MdtOp()
        op.kind = rt.OP_GN_STATS
        op.a, op.out = x.ref(), st.ref()
        op.i[rt.N_ROWS], op.i[rt.N_LD], op.i[rt.N_GROUPS], op.i[rt.N_GSIZE] = x.rows, x.ld, groups, gsize
        op.f[0] = eps
        self._emit(op)
        return st

    @staticmethod
    def gn_act_ok(rows: int, ld: int, groups: int, gsize: int) -> bool:
        """Mirror of gn_act_eligible (csrc/k_norm.hip)."""
        if groups <= 0 or 256 % groups or gsize % 4 or groups * gsize != ld:
            return False
        tpg = 256 // groups
        return (rows * (gsize // 4) + tpg - 1) // tpg <= 32

    def gn_act(self, x: Ten, groups: int, gsize: int, eps: float, gain: int, nbias: int, silu: bool,
               film=None, x2: Optional[Ten] = None, scale2: float = 1.0, raw16: Optional[Ten] = None) -> Ten:
        """GroupNorm + FiLM + SiLU in one pass (MDT_OP_GN_ACT) -> new activated tensor (bf16 in the plain-bf16 mode: its
        only reader is the convolution GEMM).  x2 (round 6): the input is cat([x, scale2 * x2]) read from its two sources; raw16: the
        op also leaves a raw bf16 copy of its input."""
        ld = x.ld + (x2.ld if x2 is not None else 0)
        y = self._new16(x.rows, ld) if self.b16_ok(ld) else self._new(x.rows, ld, ld if x2 is not None else x.c)
        op = rt.MdtOp()
        op.kind = rt.OP_GN_ACT
        op.a, op.out = x.ref(), y.ref()
        op.p0, op.p1 = _ref(rt.SP_WEIGHT, gain), _ref(rt.SP_WEIGHT, nbias)
        if isinstance(film, tuple):
            op._film = film
        i = op.i
        i[rt.N_ROWS], i[rt.N_LD], i[rt.N_GROUPS], i[rt.N_GSIZE], i[rt.N_SILU] = x.rows, ld, groups, gsize, int(silu)
        i[rt.N_OUT16] = int(y.b16)
        if x2 is not None:
            assert x2.rows == x.rows and x.ld % gsize == 0 and not x.b16 and not x2.b16
            op.a2, i[rt.N_CA], op.f[1] = x2.ref(), x.ld, scale2
        if raw16 is not None:
            assert raw16.b16 and raw16.rows == x.rows and raw16.ld == ld
            op.p2 = raw16.ref()
        op.f[0] = eps
        self._emit(op)
        return y

    def attn(self, q: Ten, kv: rt.MdtRef, tk: int, kv_bstride: int, out: Ten, ldkv: Optional[int] = None, kcol: int = 0,
             kv16: bool = False) -> None:
        cfg = self.cfg
        op = rt.MdtOp()
        op.kind = rt.OP_ATTN
        op.a, op.out = q.ref(), out.ref()
        if isinstance(kv, tuple):        # ("kv", cross-attention layer index): resolved in build()
            op._kv = kv
            op.a2 = _ref(rt.SP_ACT, 0)
        else:
            op.a2 = kv
        i = op.i
        i[rt.A_T], i[rt.A_TK], i[rt.A_HEADS] = q.rows, tk, cfg.heads
        i[rt.A_LDQ], i[rt.A_LDKV], i[rt.A_LDO], i[rt.A_KV_BSTRIDE] = q.ld, ldkv or 2 * cfg.mid_features, out.ld, kv_bstride
        i[rt.A_OUT16], i[rt.A_KCOL] = int(out.b16), kcol
        i[rt.A_IN16] = (1 if q.b16 else 0) | (2 if kv16 else 0)     # plain-bf16 mode: q | k | v as their GEMM wrote them
        op.f[0] = float(cfg.head_features) ** -0.5
        self._emit(op)
        self.flops += 2 * 2 * q.rows * tk * cfg.mid_features

    # ------------------------------------------------------------------ row-stationary convolution (MDT_OP_RCONV)
    def rconv_ok(self, rows: int, c: int, taps: int, gsize: int) -> bool:
        """Mirror of rconv_supported (csrc/k_rconv.hip) plus the policy switches."""
        if not (self.use_rconv and self.ring_mode and self.fuse_blocks):
            return False
        return c in (128, 256) and 0 < rows <= 16 and 16 % rows == 0 and taps in (1, 3) and gsize in (0, 4, 8, 16, 32, 64)

    def rconv(self, x: Ten, w: torch.Tensor, name: str, out: Ten, *, taps: int, bias_off: Optional[int] = None,
              res: Optional[Ten] = None, gn=None, film=None, in_scale: float = 1.0, x2: Optional[Ten] = None,
              in_scale2: float = 1.0, ksrc: int = 0, half_out: bool = False, flops: Optional[int] = None, nb: int = 1) -> None:
        """out = bias + conv_k(silu(gn(in_scale * x) * (scale + 1) + shift)) (+ res); w is [C][C][taps] (a slice of
        the reference Conv1d weight), gn = (gain offset, bias offset, gsize, eps, silu) in the packed weights.
        With x2 the input is cat([in_scale * x, in_scale2 * x2]) and w is [C][2C][taps]; the GroupNorm vectors at the
        gain / bias offsets then hold 2C entries."""
        # ksrc = 2: x is [rows][2 C], its two C-channel blocks are the sources (MDT_R_KSRC); half_out: w is [C / 2][C][taps] and only
        # output channels 0 .. C / 2 - 1 exist (MDT_R_HALF_OUT; the tile stream keeps all four chunks of a (tap, K half), two unused)
        c = x.ld // ksrc if ksrc else x.ld
        nsrc = ksrc if ksrc else (2 if x2 is not None else 1)
        if half_out:
            w = torch.cat([w, torch.zeros_like(w)])
        if taps == 3 and x.rows == 1:
            # ONE token per sample (configs[2]'s 256-channel level): both neighbours of a k = 3 convolution are zero padding
            # (modules.py:105-112: padding = 1), their products are exactly 0 -- only the centre tap is streamed and multiplied
            w, taps = w[:, :, 1:2], 1
        # nb > 1: w is [nb C][C][taps] -- nb convolutions of the same rows, output channels block after block (MDT_R_NB)
        assert w.shape == (nb * c, nsrc * c, taps) and out.ld == (c // 2 if half_out else nb * c) and out.rows == x.rows, \
            (name, tuple(w.shape), c, taps)
        assert x2 is None or (x2.ld == c and x2.rows == x.rows and film is None)
        tiles = [self._wtile(w[b_ * c + 64 * ch: b_ * c + 64 * ch + 64, s * c + 128 * kh: s * c + 128 * kh + 128, tap])
                 for b_ in range(nb) for s in range(nsrc) for tap in range(taps) for kh in range(c // 128) for ch in range(c // 64)]
        op = rt.MdtOp()
        op.kind = rt.OP_RCONV
        op.a, op.out = x.ref(), out.ref()
        op.w = _ref(rt.SP_WEIGHT, self.W.add(name + "/rconv.tiles", torch.cat(tiles)))
        op.i[rt.W_KB] = sum(t.numel() for t in tiles) * 4 // 1024
        if bias_off is not None:
            op.bias = _ref(rt.SP_WEIGHT, bias_off)
        if res is not None:
            op.res = res.ref()
        i = op.i
        if x2 is not None:
            op.a2 = x2.ref()
            i[rt.R_LDA2] = x2.ld
            op.f[2] = in_scale2
        i[rt.R_T], i[rt.R_C], i[rt.R_LDA], i[rt.R_LDC], i[rt.R_TAPS] = x.rows, c, x.ld, out.ld, taps
        i[rt.R_LDR] = res.ld if res is not None else 0
        i[rt.R_FILM_LD], i[rt.R_WF32] = c, int(self.wf32)
        i[rt.R_KSRC], i[rt.R_HALF_OUT], i[rt.R_NB] = ksrc, int(half_out), (nb if nb > 1 else 0)
        op.f[0], op.f[1] = 1e-5, in_scale
        if gn is not None:
            gain, nbias, gsize, eps, silu = gn
            op.p0, op.p1 = _ref(rt.SP_WEIGHT, gain), _ref(rt.SP_WEIGHT, nbias)
            i[rt.R_GSIZE], i[rt.R_SILU] = gsize, int(silu)
            op.f[0] = eps
        if isinstance(film, tuple):
            op._film = film
        self._emit(op)
        # executed products (skipped padding is not counted; `flops`: the caller's count where the weights hold structural zeros)
        self.flops += flops if flops is not None else 2 * x.rows * c * c * taps * nsrc

    def down_patch_ok(self, x: Ten, ci: int, co: int, f: int) -> bool:
        """The strided convolution of a DownsampleBlock1d in PATCH form on the row-stationary kernel (down_patch)."""
        return (self.patch_conv and self.use_rconv and self.ring_mode and self.fuse_blocks and (self.gemm_mode == "bf16x3" or self.wf32)
                and x.ld == ci and x.rows % f == 0 and (x.rows // f) in (1, 2, 4, 8, 16)
                and ((ci * f == 256 and co == 128) or (ci * f == 512 and co == 256)))

    def down_patch(self, x: Ten, dp: str, ci: int, co: int, f: int, y: Ten) -> None:
        """Conv1d(ci -> co, kernel 2 f + 1, stride f, padding f) (modules.py:62-75) as a k = 3 convolution over PATCHES: f consecutive
        tokens are one row of f ci contiguous values, output token t reads patch t - 1 (taps 0 .. f - 1), patch t (taps f .. 2 f - 1)
        and the first token of patch t + 1 (tap 2 f); the zero padding of the patch convolution is the zero padding of the original.
        f ci = 256 -> 128 channels: the 256-channel kernel producing its lower half of the outputs; f ci = 512 -> 256: the two
        256-value halves of a patch as the kernel's two sources.  (The tiled GEMM ran these at 16 - 25 us, latency-bound.)"""
        w = self.sd[dp + "downsample.weight"].float()            # [co][ci][2 f + 1]
        cp = ci * f
        w3 = torch.zeros(co, cp, 3)
        for q in range(f):
            w3[:, q * ci: (q + 1) * ci, 0] = w[:, :, q]
            w3[:, q * ci: (q + 1) * ci, 1] = w[:, :, f + q]
        w3[:, :ci, 2] = w[:, :, 2 * f]
        xp = Ten(x.space, x.off, x.rows // f, cp, cp)
        bias = self._vec(dp + "downsample.bias", co)
        fl = 2 * xp.rows * co * ci * ((2 * f + 1) if xp.rows > 1 else f)      # the convolution's own products (as the GEMM lowering counts)
        if cp == 256:
            self.rconv(xp, w3, dp + "downsample.weight/patch", y, taps=3, bias_off=bias, half_out=True, flops=fl)
        else:
            self.rconv(xp, w3, dp + "downsample.weight/patch", y, taps=3, bias_off=bias, ksrc=2, flops=fl)

    def up_patch_ok(self, x: Ten, ci: int, co: int, f: int, res: Optional[Ten]) -> bool:
        """The ConvTranspose1d of an UpsampleBlock1d in PATCH form on the row-stationary kernel (up_patch)."""
        return (self.patch_conv and self.use_rconv and self.ring_mode and self.fuse_blocks and (self.gemm_mode == "bf16x3" or self.wf32)
                and f % 2 == 0 and x.ld == ci and ci in (128, 256) and (f * co) % ci == 0 and 1 < f * co // ci <= 8
                and x.rows in (1, 2, 4, 8, 16) and (res is None or res.ld == co))

    def up_patch(self, x: Ten, up: str, ci: int, co: int, f: int, y: Ten, res: Optional[Ten]) -> None:
        """ConvTranspose1d(ci -> co, kernel 2 f, stride f, padding f / 2) (modules.py:74-81) as a k = 3 convolution that writes
        PATCHES: input token t produces the f output tokens f t .. f t + f - 1 = one row of f co contiguous values.  With the phase
        weights W(ph, 0) = w[:, :, ph]^T, W(ph, 1) = w[:, :, ph + f]^T of the GEMM lowering: output token f t + j is
        W(j + f/2, 1) x[t - 1] + W(j + f/2, 0) x[t] for j < f/2 and W(j - f/2, 1) x[t] + W(j - f/2, 0) x[t + 1] otherwise; zero padding
        of the patch convolution = the rows the transposed convolution never reads.  f co = NB ci output channels: NB blocks of one
        MDT_OP_RCONV launch (the GEMM form ran the f phases as one latency-bound launch of 17 - 22 us)."""
        wt = self.sd[up + "upsample.weight"].float()             # [ci][co][2 f]
        w3 = torch.zeros(f * co, ci, 3)
        h = f // 2
        for j in range(f):
            rows = slice(j * co, (j + 1) * co)
            if j < h:
                w3[rows, :, 0] = wt[:, :, j + h + f].T
                w3[rows, :, 1] = wt[:, :, j + h].T
            else:
                w3[rows, :, 1] = wt[:, :, j - h + f].T
                w3[rows, :, 2] = wt[:, :, j - h].T
        bias = self.W.add(up + "upsample.bias/patch", self.sd[up + "upsample.bias"].float().repeat(f))
        yp = Ten(y.space, y.off, x.rows, f * co, f * co)
        rp = Ten(res.space, res.off, x.rows, f * co, f * co) if res is not None else None
        self.rconv(x, w3, up + "upsample.weight/patch", yp, taps=3, bias_off=bias, res=rp, nb=f * co // ci,
                   flops=2 * x.rows * f * co * 2 * ci)

    def _resnet_rconv(self, xa: Ten, xb: Optional[Ten], scale_b: float, p: str, c: int, groups: int,
                      free_input: bool) -> Ten:
        """ResnetBlock1d.forward (modules.py:193-205) on row-stationary convolutions; with xb the block input is
        cat([xa, scale_b * xb]) (UpsampleBlock1d.add_skip, modules.py:828-829) and is never materialised: every
        2C-channel convolution is two C-channel launches, the second accumulating into the first's output."""
        sd = self.sd
        cin = 2 * c if xb is not None else c
        gsize = cin // groups
        g1, b1 = self._vec(p + "block1.groupnorm.weight", cin), self._vec(p + "block1.groupnorm.bias", cin)
        w1 = sd[p + "block1.project.weight"]                          # [c, cin, 3]
        h = self._new(xa.rows, c)
        bias1 = self._vec(p + "block1.project.bias", c)
        if xb is None or self.rconv_two:
            # one launch, two sources (measured: 2520 molecules/s against 2545 for the two accumulating launches
            # below -- the in-kernel second prologue costs what the second launch costs -- so this is not the default)
            self.rconv(xa, w1, p + "block1.project.weight", h, taps=3, bias_off=bias1,
                       gn=(g1, b1, gsize, 1e-5, True), x2=xb, in_scale2=scale_b)
        else:
            self.rconv(xa, w1[:, :c], p + "block1.project.weight/a", h, taps=3, bias_off=bias1,
                       gn=(g1, b1, gsize, 1e-5, True))
            self.rconv(xb, w1[:, c:], p + "block1.project.weight/b", h, taps=3, res=h,
                       gn=(g1 + c, b1 + c, gsize, 1e-5, True), in_scale=scale_b)
        if xb is not None:
            wr = sd[p + "to_out.weight"]                              # [c, 2c, 1]
            r = self._new(xa.rows, c)
            br = self._vec(p + "to_out.bias", c)
            if self.rconv_two_k1:
                self.rconv(xa, wr, p + "to_out.weight", r, taps=1, bias_off=br, x2=xb, in_scale2=scale_b)
            else:
                self.rconv(xa, wr[:, :c], p + "to_out.weight/a", r, taps=1, bias_off=br)
                self.rconv(xb, wr[:, c:], p + "to_out.weight/b", r, taps=1, res=r, in_scale=scale_b)
        else:
            assert (p + "to_out.weight") not in sd
            r = xa
        ss_off = self.ss_total                                        # FiLM vectors inside the shared (scale | shift) row
        self.ss_offsets[p] = ss_off
        self.ss_total += 2 * c
        y = self._new(xa.rows, c)
        g2, b2 = self._vec(p + "block2.groupnorm.weight", c), self._vec(p + "block2.groupnorm.bias", c)
        self.rconv(h, sd[p + "block2.project.weight"], p + "block2.project.weight", y, taps=3,
                   bias_off=self._vec(p + "block2.project.bias", c), res=r, gn=(g2, b2, c // groups, 1e-5, True),
                   film=("ss", ss_off))
        self._free(h)
        if r is not xa:
            self._free(r)
        if free_input:
            self._free(xa)
            if xb is not None:
                self._free(xb)
        return y

    def resnet_cat(self, xa: Ten, xb: Ten, scale_b: float, p: str, c: int, groups: int) -> Ten:
        """ResnetBlock1d on cat([xa, scale_b * xb]) (2c -> c channels); frees both inputs."""
        if xa.ld == c and xb.ld == c and groups % 2 == 0 and self.rconv_ok(xa.rows, c, 3, 2 * c // groups):
            return self._resnet_rconv(xa, xb, scale_b, p, c, groups, True)
        if (self.cat_fold and self.gemm_mode == "bf16" and xa.ld == c and xb.ld == c and self.b16_ok(2 * c) and groups > 0
                and c % (2 * c // groups) == 0 and self.gn_act_ok(xa.rows, 2 * c, groups, 2 * c // groups)
                and (p + "to_out.weight") in self.sd and not self.resblock_ok(xa.rows, 2 * c, c, groups, p)):
            # plain-bf16 mode (round 6): the concatenated tensor is never written -- block1's GroupNorm pass reads the two sources and
            # leaves the raw bf16 copy that the to_out convolution multiplies (was: k_concat + a conversion pass, 81 / 174 us per block)
            return self.resnet(None, p, 2 * c, c, groups, cat=(xa, xb, scale_b))
        cat = self.concat(xa, xb, scale_b)
        self._free(xa)
        self._free(xb)
        return self.resnet(cat, p, 2 * c, c, groups)

    # ------------------------------------------------------------------ blocks
    # ------------------------------------------------------------------ whole ResNet block in one launch
    @staticmethod
    def _resblock_steps(c: int, taps: int):
        """k-steps of a convolution over c channels per tap as the kernel enumerates them (csrc/k_resblock.hip): every
        step is 32 (tap, channel) pairs, None where the step is padded."""
        if taps == 3:
            if c == 64:
                return [[(s // 2, 32 * (s % 2) + j) for j in range(32)] for s in range(6)]
            return [[(j // 16, j % 16) for j in range(32)], [(2, j) if j < 16 else None for j in range(32)]]
        if c == 64:
            return [[(0, 32 * s + j) for j in range(32)] for s in range(2)]
        return [[(0, j) if j < 16 else None for j in range(32)]]

    @classmethod
    def _resblock_frags(cls, w: torch.Tensor, f32: bool = False) -> List[torch.Tensor]:
        """Conv1d weight [cout][c][taps] -> MFMA A-operand fragments, order (step, 16-row tile): lane i + 16 g of a
        fragment holds W[16 rt + i][the step's pairs 8 g .. 8 g + 7]; bf16 hi plane then lo plane (1 KB each), or (f32) the
        fp32 values as two halves of 1 KB: pairs 8 g .. 8 g + 3, then 8 g + 4 .. 8 g + 7 (_tile_f32 of the [16][32] step)."""
        cout, c, taps = w.shape
        frags = []
        for step in cls._resblock_steps(c, taps):
            m = torch.zeros(cout, 32)
            for j, tc in enumerate(step):
                if tc is not None:
                    m[:, j] = w[:, tc[1], tc[0]]
            for r in range(cout // 16):
                if f32:
                    frags.append(cls._tile_f32(m[16 * r: 16 * r + 16]))
                else:
                    frags.append(cls._tile(m[16 * r: 16 * r + 16].reshape(16, 4, 8).permute(1, 0, 2).contiguous()))
        return frags

    def resblock_ok(self, rows: int, cin: int, cout: int, groups: int, p: str) -> bool:
        # (padded channel counts: QMDiffusionForward's Patcher has 2 real input channels, its Unpatcher 1 real output channel)
        return (self.use_resblock and self.ring_mode and self.fuse_blocks and groups == 1 and rows == 64
                and (pad16(cin), pad16(cout)) in ((16, 64), (64, 16), (16, 16)) and (p + "to_out.weight") in self.sd)

    def resblock(self, x: Ten, p: str, cin: int, cout: int, free_input: bool) -> Ten:
        """ResnetBlock1d with one GroupNorm group on the 64-token level as ONE launch (MDT_OP_RESBLOCK); cin / cout are the real
        channel counts, the op works on the padded ones (zero gains / biases / weights on the padding)."""
        sd = self.sd
        cin_p, cout_p = pad16(cin), pad16(cout)
        y = self._new(x.rows, cout_p, cout)

        def wpad(name, ci, co):
            w = sd[p + name].float()
            out = torch.zeros(pad16(co), pad16(ci), w.shape[2])
            out[:co, :ci] = w
            return out

        def vpad(v, c):
            out = torch.zeros(pad16(c))
            out[:c] = v.float()
            return out
        frags = (self._resblock_frags(wpad("block1.project.weight", cin, cout), self.wf32)
                 + self._resblock_frags(wpad("block2.project.weight", cout, cout), self.wf32)
                 + self._resblock_frags(wpad("to_out.weight", cin, cout), self.wf32))
        vec = torch.cat([vpad(sd[p + "block1.groupnorm.weight"], cin), vpad(sd[p + "block1.groupnorm.bias"], cin),
                         vpad(sd[p + "block1.project.bias"], cout),
                         vpad(sd[p + "block2.groupnorm.weight"], cout), vpad(sd[p + "block2.groupnorm.bias"], cout),
                         vpad(sd[p + "block2.project.bias"] + sd[p + "to_out.bias"], cout)])
        ss_off = self.ss_total                       # FiLM vectors of this block inside the shared (scale | shift) row
        self.ss_offsets[p] = ss_off
        self.ss_total += 2 * cout_p
        op = rt.MdtOp()
        op.kind = rt.OP_RESBLOCK
        op.a, op.out = x.ref(), y.ref()
        op.w = _ref(rt.SP_WEIGHT, self.W.add(p + "resblock.frags", torch.cat(frags)))
        op.bias = _ref(rt.SP_WEIGHT, self.W.add(p + "resblock.vec", vec))
        op._film = ("ss", ss_off)
        i = op.i
        i[rt.K_T], i[rt.K_CIN], i[rt.K_COUT], i[rt.K_FILM_LD], i[rt.K_WF32] = x.rows, cin_p, cout_p, cout_p, int(self.wf32)
        i[rt.K_CIN_REAL], i[rt.K_COUT_REAL] = cin, cout
        op.f[0] = 1e-5
        self._emit(op)
        self.flops += 2 * x.rows * (3 * cin_p * cout_p + 3 * cout_p * cout_p + cin_p * cout_p)
        if free_input:
            self._free(x)
        return y

    def resnet(self, x: Optional[Ten], p: str, cin: int, cout: int, groups: int, free_input: bool = True,
               cat: Optional[Tuple[Ten, Ten, float]] = None) -> Ten:
        """ResnetBlock1d.forward (modules.py:193-205); x has `cin` real channels.  cat = (xa, xb, scale) instead of x (resnet_cat, plain-bf16
        mode): the block's input is cat([xa, scale * xb]), read from its sources; both are freed."""
        cin_p, cout_p = pad16(cin), pad16(cout)
        if cat is not None:
            xa, xb, scale_b = cat
            g1, b1 = self._vec(p + "block1.groupnorm.weight", cin_p), self._vec(p + "block1.groupnorm.bias", cin_p)
            x16 = self._new16(xa.rows, cin_p)
            a1 = self.gn_act(xa, groups, cin // groups, 1e-5, g1, b1, True, x2=xb, scale2=scale_b, raw16=x16)
            self._free(xa)
            self._free(xb)
            h = self._new(x16.rows, cout_p, cout)
            self.gemm(a1, self._conv_w(p + "block1.project.weight", cin_p, cout_p), cout_p, h, cin=cin_p,
                      bias_off=self._vec(p + "block1.project.bias", cout_p), taps=3, t_dj=1, t_off=-1)
            self._free(a1)
            r = self._new(x16.rows, cout_p, cout)
            self.gemm(x16, self._conv_w(p + "to_out.weight", cin_p, cout_p), cout_p, r, cin=cin_p,
                      bias_off=self._vec(p + "to_out.bias", cout_p))
            self._free(x16)
            return self._resnet_tail(h, r, None, p, cout, cout_p, groups, False)
        assert x.ld == cin_p, (p, x.ld, cin_p)
        if self.resblock_ok(x.rows, cin, cout, groups, p):
            return self.resblock(x, p, cin, cout, free_input)
        if cin == cout and x.ld == cin and self.res256_ok(p, cin, x.rows, groups, False):
            y = self._new(x.rows, cin)                        # a chain of one block (its "skip" store is the output itself)
            return self.resnet_chain256(x, [p], 1, [y], 1.0, y, free_input)
        if cin == cout and x.ld == cin and (p + "to_out.weight") not in self.sd and groups > 0 and cin % groups == 0 \
                and self.rconv_ok(x.rows, cin, 3, cin // groups):
            return self._resnet_rconv(x, None, 1.0, p, cin, groups, free_input)
        g1, b1 = self._vec(p + "block1.groupnorm.weight", cin_p), self._vec(p + "block1.groupnorm.bias", cin_p)
        h = self._new(x.rows, cout_p, cout)
        w1 = self._conv_w(p + "block1.project.weight", cin_p, cout_p)
        bias1 = self._vec(p + "block1.project.bias", cout_p)
        if self.gn_act_ok(x.rows, cin_p, groups, cin // groups):
            # fused normalise + SiLU pass, then a plain conv GEMM (the prologue form recomputes the transform
            # 3 taps x N/64 column tiles times per element and is VALU-bound)
            a1 = self.gn_act(x, groups, cin // groups, 1e-5, g1, b1, True)
            self.gemm(a1, w1, cout_p, h, cin=cin_p, bias_off=bias1, taps=3, t_dj=1, t_off=-1)
            self._free(a1)
        else:
            st1 = self.gn_stats(x, groups, cin // groups, 1e-5)
            self.gemm(x, w1, cout_p, h, cin=cin_p, bias_off=bias1, taps=3, t_dj=1, t_off=-1,
                      pro=rt.PRO_GROUPNORM, gain=g1, nbias=b1, stats=st1, groups=groups, gsize=cin // groups,
                      pro_silu=1)
            self._free(st1)
        if (p + "to_out.weight") in self.sd:
            r = self._new(x.rows, cout_p, cout)
            self.gemm(x, self._conv_w(p + "to_out.weight", cin_p, cout_p), cout_p, r, cin=cin_p,
                      bias_off=self._vec(p + "to_out.bias", cout_p))
        else:
            r = x
        return self._resnet_tail(h, r, x, p, cout, cout_p, groups, free_input)

    def _resnet_tail(self, h: Ten, r: Ten, x: Optional[Ten], p: str, cout: int, cout_p: int, groups: int, free_input: bool) -> Ten:
        """block2 of a ResnetBlock1d (GroupNorm + FiLM + SiLU + conv) + the skip r; frees h, r (if it is not x) and x (free_input)."""
        # FiLM vectors of this block inside the shared (scale | shift) row
        ss_off = self.ss_total
        self.ss_offsets[p] = ss_off
        self.ss_total += 2 * cout_p
        y = self._new(h.rows, cout_p, cout)
        g2, b2 = self._vec(p + "block2.groupnorm.weight", cout_p), self._vec(p + "block2.groupnorm.bias", cout_p)
        w2 = self._conv_w(p + "block2.project.weight", cout_p, cout_p)
        bias2 = self._vec(p + "block2.project.bias", cout_p)
        if self.gn_act_ok(h.rows, cout_p, groups, cout // groups):
            a2 = self.gn_act(h, groups, cout // groups, 1e-5, g2, b2, True, film=("ss", ss_off))
            self.gemm(a2, w2, cout_p, y, cin=cout_p, bias_off=bias2, taps=3, t_dj=1, t_off=-1, res=r)
            self._free(a2)
        else:
            st2 = self.gn_stats(h, groups, cout // groups, 1e-5)
            self.gemm(h, w2, cout_p, y, cin=cout_p, bias_off=bias2, taps=3, t_dj=1, t_off=-1,
                      pro=rt.PRO_GROUPNORM, gain=g2, nbias=b2, stats=st2, groups=groups, gsize=cout // groups,
                      pro_silu=1, film=("ss", ss_off), res=r)
            self._free(st2)
        self._free(h)
        if r is not x:
            self._free(r)
        if free_input and x is not None:
            self._free(x)
        return y

    # ------------------------------------------------------------------ fused transformer sub-blocks
    # slot order of the output-projection K dimension: the kernel takes the B operand straight from the
    # accumulator registers of the previous MFMA, which hold feature d = 32 sp + 16 (e >> 2) + 4 g + (e & 3)
    # in k-slot sigma = 32 sp + 8 g + e (csrc/k_tblock_lw.hip)
    _SLOT_PERM = [32 * (s >> 5) + 16 * ((s & 7) >> 2) + 4 * ((s >> 3) & 3) + (s & 3) for s in range(64)]

    def can_fuse_transformer(self, c: int, rows: int, cross: bool) -> bool:
        # (exact-fp32 products: the C = 128 sub-block kernel has fp32-fragment instantiations since round 5, the C = 256 ones
        #  (k_tblock32) since round 6: MDT_B_WF32)
        if not (self.gemm_mode == "bf16x3" or self.wf32) or not self.fuse_blocks:
            return False
        if c not in (128, 256) or rows > 16 or 16 % rows or self.cfg.head_features != 64:
            return False
        if (c * self.cfg.ff_mult) % 64:
            return False
        # cross-attention with more context rows than any fused kernel takes (configs[2]: 64 keys per sample, 1..4 tokens)
        # runs layer by layer between the fused self-attention and feed-forward blocks
        return True

    @staticmethod
    def _tile(w: torch.Tensor) -> torch.Tensor:
        """[rows][cols] fp32 -> bf16 hi plane then lo plane, as raw bits viewed as float32."""
        hi = w.to(torch.bfloat16)
        lo = (w - hi.float()).to(torch.bfloat16)
        return torch.cat([hi.contiguous().view(-1), lo.contiguous().view(-1)]).view(torch.float32)

    @staticmethod
    def _tile_f32(w: torch.Tensor) -> torch.Tensor:
        """[rows][cols] fp32 (rows % 16 == 0, cols % 32 == 0, columns in k-slot order) -> the fp32 FRAGMENT tile of the exact-fp32
        ring kernels (MDT_F_WF32, include/mdt_hip.h): fragment (row tile rt, k-step st, half lo) is 1 KB at
        ((rt * (cols / 32) + st) * 2 + lo) * 1024, float r of lane i + 16 g in it = w[16 rt + i][32 st + 8 g + 4 lo + r].  Same
        size as the two bf16 planes of _tile, same weights, every value kept exactly."""
        rows, cols = w.shape
        assert rows % 16 == 0 and cols % 32 == 0, (rows, cols)
        return (w.float().reshape(rows // 16, 16, cols // 32, 4, 2, 4).permute(0, 2, 4, 3, 1, 5).contiguous().view(-1))

    def _wtile(self, w: torch.Tensor) -> torch.Tensor:
        """A ring-kernel weight tile in the format of this compilation's product type."""
        return self._tile_f32(w) if self.wf32 else self._tile(w)

    def tblock(self, t: Ten, mode: int, p: str, cross_index: Optional[int] = None, variant: int = 0,
               x_out: Optional[Ten] = None, p_in: Optional[Ten] = None, p_out: Optional[Ten] = None,
               post=None) -> None:
        """One fused sub-block: MDT_OP_TBLOCK (self-attention / cross-attention / feed-forward), in place on t, or
        (variant 4) from t + p_in into x_out with the second head group's partial sum left in p_out."""
        cfg, sd = self.cfg, self.sd
        c, rows = t.ld, t.rows
        perm = torch.tensor(self._SLOT_PERM)
        tiles: List[torch.Tensor] = []
        mats: List[torch.Tensor] = []                # the matrices behind `tiles` (the fp32 sub-tile packing below starts from them)

        def wtile(w: torch.Tensor) -> torch.Tensor:  # self._wtile that remembers its argument
            mats.append(w)
            return self._wtile(w)
        if variant in (2, 3, 4):
            assert c == 256, "variants 2 / 3 / 4 (32-row workgroups, sub-tile stream) serve C = 256"
            assert mode != rt.TB_CROSS or (16 // rows) * self.n_ctx <= 48, "at most 48 context rows per 16 token rows"
        if mode == rt.TB_FF:
            w1, b1 = sd[p + "0.weight"], sd[p + "0.bias"]          # [2C, C]
            w2, b2 = sd[p + "2.weight"], sd[p + "2.bias"]          # [C, 2C]
            nchunk = w1.shape[0] // 64
            if post is not None:
                # Transformer1d's closing Conv1d(k=1) (modules.py:524) folded in: y = Wout (x + W2 h + b2) + bout
                #   = (Wout W2) h + Wout x + (Wout b2 + bout); the kernel adds the Wout x term from its raw-x operands
                wout, bout = post[0].reshape(c, c).double(), post[1].double()
                w2, b2 = (wout @ w2.double()).float(), (wout @ b2.double() + bout).float()
            for h in range(nchunk):
                tiles += [wtile(w1[64 * h: 64 * h + 64]), wtile(w2[:, 64 * h: 64 * h + 64][:, perm])]
            if post is not None:
                assert variant in (0, 2, 4) and (x_out is None or (variant == 4 and p_out is None))
                tiles += [wtile(post[0].reshape(c, c)[:, 64 * e: 64 * e + 64]) for e in range(c // 64)]
                self.flops += 2 * rows * c * c
            bias = torch.cat([b1, b2])
            self.flops += 2 * 2 * rows * c * w1.shape[0]
        else:
            g_q, b_q = sd[p + "norm.weight"], sd[p + "norm.bias"]
            wq = sd[p + "to_q.weight"]                              # [mid, C]
            wo, bo = sd[p + "attention.to_out.weight"], sd[p + "attention.to_out.bias"]   # [C, mid]
            nchunk = wq.shape[0] // 64
            wq_f, bq_f = wq * g_q.unsqueeze(0), wq @ b_q           # LayerNorm affine folded: W (g*xn + b) = (W g) xn + W b
            if mode == rt.TB_SELF:
                g_c, b_c = sd[p + "norm_context.weight"], sd[p + "norm_context.bias"]
                wkv = sd[p + "to_kv.weight"]                        # [2 mid, C]
                wkv_f, bkv_f = wkv * g_c.unsqueeze(0), wkv @ b_c
                mid = wq.shape[0]
                for h in range(nchunk):
                    tiles += [wtile(wq_f[64 * h: 64 * h + 64]), wtile(wkv_f[64 * h: 64 * h + 64]),
                              wtile(wkv_f[mid + 64 * h: mid + 64 * h + 64]),
                              wtile(wo[:, 64 * h: 64 * h + 64][:, perm])]
                # k bias: softmax is invariant to a per-query constant (q . bk), drop it; v bias: sum_j p_j (v_j + bv)
                # = sum_j p_j v_j + bv, so it moves into the output bias.  The kernels then load only bq per head
                # (their own global loads queue behind the loader waves' DMA traffic: ~60 cycles of issue stall each).
                bo_eff = bo + wo @ bkv_f[mid:]
                bias = torch.cat([bq_f, torch.zeros(2 * mid), bo_eff])
                self.flops += 2 * rows * c * 3 * mid + 4 * rows * rows * mid + 2 * rows * mid * c
            else:
                for h in range(nchunk):
                    tiles += [wtile(wq_f[64 * h: 64 * h + 64]), wtile(wo[:, 64 * h: 64 * h + 64][:, perm])]
                bias = torch.cat([bq_f, bo])
                mid = wq.shape[0]
                self.flops += 2 * rows * c * mid + 4 * rows * self.n_ctx * mid + 2 * rows * mid * c
        if variant in (2, 3, 4):
            # k_tblock32 streams 32 KB sub-tiles in the C = 128 tile format: a [64][256] projection tile as its two
            # K halves, a [256][64] output tile as its two row halves (tiles alternate P ... P O per chunk)
            tpc = 4 if mode == rt.TB_SELF else 2
            sub: List[torch.Tensor] = []
            for k, tl in enumerate(tiles):
                out_tile = k % tpc == tpc - 1 or k >= nchunk * tpc      # output tiles (incl. a folded closing convolution's)
                if self.wf32:
                    # fp32 fragment sub-tiles (k_tf256.hip's format): [64][128] K halves, [128][64] row halves
                    m_ = mats[k]
                    for hm in ((m_[:128], m_[128:]) if out_tile else (m_[:, :128], m_[:, 128:])):
                        sub.append(self._tile_f32(hm.contiguous()))
                    continue
                raw = tl.view(torch.bfloat16)
                n = raw.numel() // 2
                if k % tpc == tpc - 1 or k >= nchunk * tpc:      # output tiles (incl. a folded closing convolution's)
                    hi, lo = raw[:n].view(c, 64), raw[n:].view(c, 64)
                    halves = [(hi[:128], lo[:128]), (hi[128:], lo[128:])]
                else:
                    hi, lo = raw[:n].view(64, c), raw[n:].view(64, c)
                    halves = [(hi[:, :128], lo[:, :128]), (hi[:, 128:], lo[:, 128:])]
                for hh, ll in halves:
                    sub.append(torch.cat([hh.contiguous().view(-1), ll.contiguous().view(-1)]).view(torch.float32))
            tiles = sub
        op = rt.MdtOp()
        op.kind = rt.OP_TBLOCK
        op.a = t.ref()
        op.w = _ref(rt.SP_WEIGHT, self.W.add(p + "tblock.tiles", torch.cat(tiles)))
        op.i[rt.W_KB] = sum(t.numel() for t in tiles) * 4 // 1024
        op.bias = _ref(rt.SP_WEIGHT, self.W.add(p + "tblock.bias", bias))
        i = op.i
        i[rt.B_MODE], i[rt.B_C], i[rt.B_T], i[rt.B_NCHUNK], i[rt.B_NBIAS] = mode, c, rows, nchunk, bias.numel()
        i[rt.B_TK], i[rt.B_KV_BSTRIDE], i[rt.B_LDKV], i[rt.B_HEADS] = self.n_ctx, self.n_ctx, 2 * cfg.mid_features, cfg.heads
        i[rt.B_VARIANT] = variant
        i[rt.B_WF32] = int(self.wf32)
        op.f[0], op.f[1] = 1e-5, float(cfg.head_features) ** -0.5
        if mode == rt.TB_CROSS:
            op._kv = ("kv", cross_index)
            op.a2 = _ref(rt.SP_ACT, 0)
        part = None
        if post is not None:
            op.out = post[2].ref()
            i[rt.B_POST] = (c // 64) * (2 if variant in (2, 4) else 1)     # extra (sub-)tiles
        if variant == 4:
            assert x_out is not None and x_out is not t and (p_out is None or nchunk % 2 == 0)
            op.out = x_out.ref()
            if p_in is not None:
                op.res = p_in.ref()
            if p_out is not None:
                op.p2 = p_out.ref()
        if variant == 3:                         # scratch for the two head groups' partial sums: [2][B rows][c]
            assert nchunk % 2 == 0
            part = self._new(rows, 2 * c)
            op.out = part.ref()
        self._emit(op)
        if part is not None:
            self._free(part)

    # k-slot order of a projection whose operand is the residual stream held in accumulator layout (csrc/k_tf128.hip): k-slot
    # 32 st + 8 g + e of the projection tile <-> feature 16 (2 st + (e >> 2)) + 4 g + (e & 3)
    _ACC_PERM = [16 * (2 * (k >> 5) + ((k & 7) >> 2)) + 4 * ((k >> 3) & 3) + (k & 3) for k in range(128)]

    def tf128_ok(self, c: int, rows: int, layers: int, cross: bool, res_kind: int = 0, n_res: int = 0) -> bool:
        if not (self.tf128 and self.ring_mode and self.fuse_blocks and self.fold_out):
            return False
        if c != 128 or rows > 16 or 16 % rows or self.cfg.head_features != 64 or (c * self.cfg.ff_mult) % 64:
            return False
        if layers < 1 and n_res < 1:
            return False
        if cross and (16 // rows) * self.n_ctx > 16:
            return False
        nfilm = (2 * c * n_res + 255) // 256 * 256
        return self._tf128_nvec(c, layers, cross, res_kind, n_res) + nfilm <= 8192      # 32 KB of LDS behind the 128 KB ring

    def _tf128_nvec(self, c: int, layers: int, cross: bool, res_kind: int = 0, n_res: int = 0) -> int:
        mid, hid = self.cfg.mid_features, c * self.cfg.ff_mult
        n = n_res * (6 * c if res_kind == 1 else 9 * c)
        if layers > 0:
            n += c + layers * ((mid + c) * (2 if cross else 1) + hid + c)
        return (n + 255) // 256 * 256

    def res128_ok(self, p: str, c: int, rows: int, groups: int, two_source: bool) -> bool:
        """Can the ResnetBlock1d at prefix p run inside a k_tf128 launch (csrc/k_tf128.hip, RES = 1 / 2)?  GroupNorm groups of 16
        or 32 channels (statistics on whole 16-channel accumulator tiles), C -> C (single source) or 2C -> C with to_out."""
        if not (self.res128 and self.tf128 and self.ring_mode and self.fuse_blocks and self.fold_out):
            return False
        if c != 128 or rows > 16 or 16 % rows or groups <= 0:
            return False
        cin = 2 * c if two_source else c
        if cin % groups or c % groups or cin // groups not in (16, 32) or c // groups not in (16, 32):
            return False
        w1 = self.sd.get(p + "block1.project.weight")
        if w1 is None or tuple(w1.shape) != (c, cin, 3) or tuple(self.sd[p + "block2.project.weight"].shape) != (c, c, 3):
            return False
        return ((p + "to_out.weight") in self.sd) == two_source

    def transformer_fused128(self, x: Ten, p: str, c: int, layers: int, cross: bool, free_input: bool, res=None,
                             y: Optional[Ten] = None) -> Ten:
        """Transformer1d.forward (modules.py:519-524) as ONE MDT_OP_TF128: to_in, every block's sub-blocks and to_out (folded
        into the last feed-forward block), the residual stream kept in registers.  Weight tiles in consumption order; every
        projection that consumes the residual stream has its K columns in accumulator order (_ACC_PERM).
        res = (kind, [block prefixes], groups, skip tensors, scale): ResnetBlock1d blocks of the level in front of the transformer
        in the same launch -- kind 1: x = Block(x), outputs stored to skips[k]; kind 2: x = Block(cat([x, scale * skips[k]]))
        (modules.py:193-205, :828-829); layers = 0: the blocks alone."""
        cfg, sd = self.cfg, self.sd
        rows, mid = x.rows, cfg.mid_features
        heads, nff = mid // 64, c * cfg.ff_mult // 64
        acc = torch.tensor(self._ACC_PERM)
        slot = torch.tensor(self._SLOT_PERM)
        tiles: List[torch.Tensor] = []
        desc: List[int] = []
        vec: List[torch.Tensor] = []

        def wtile(t: torch.Tensor, kind: int) -> None:            # kind 0: projection tile [64][128], 1: output tile [128][64]
            desc.append(kind | (len(tiles) << 2))
            tiles.append(self._wtile(t))

        def conv3_tiles(w: torch.Tensor) -> None:                 # [c][c][3] -> (tap, output half) projection tiles
            for tap in range(3):
                for half in range(c // 64):
                    wtile(w[64 * half: 64 * half + 64, :, tap][:, acc], 0)

        res_kind, res_blocks, res_skips, film0 = 0, [], [], None
        if res is not None:
            res_kind, res_blocks, groups, res_skips, scale_b = res
            assert len(res_skips) == len(res_blocks) and res_kind in (1, 2)
            for rb, bp in enumerate(res_blocks):
                ss_off = self.ss_total                              # FiLM vectors inside the shared (scale | shift) row
                self.ss_offsets[bp] = ss_off
                self.ss_total += 2 * c
                film0 = ss_off if film0 is None else film0
                w1, w2 = sd[bp + "block1.project.weight"].float(), sd[bp + "block2.project.weight"].float()
                g1, b1 = sd[bp + "block1.groupnorm.weight"].float(), sd[bp + "block1.groupnorm.bias"].float()
                g2, b2 = sd[bp + "block2.groupnorm.weight"].float(), sd[bp + "block2.groupnorm.bias"].float()
                bias1, bias2 = sd[bp + "block1.project.bias"].float(), sd[bp + "block2.project.bias"].float()
                if res_kind == 1:
                    conv3_tiles(w1)
                    conv3_tiles(w2)
                    vec += [g1, b1, bias1, g2, b2, bias2]
                    self.flops += 2 * rows * 3 * c * c * 2
                else:
                    wr, br = sd[bp + "to_out.weight"].float(), sd[bp + "to_out.bias"].float()      # [c, 2c, 1]
                    skip_desc = 0 | (((1 << 20) | rb) << 2)
                    for half in range(c // 64):
                        wtile(wr[64 * half: 64 * half + 64, :c, 0][:, acc], 0)
                    desc.append(skip_desc)
                    for half in range(c // 64):
                        wtile(wr[64 * half: 64 * half + 64, c:, 0][:, acc], 0)
                    conv3_tiles(w1[:, :c])
                    desc.append(skip_desc)
                    conv3_tiles(w1[:, c:])
                    conv3_tiles(w2)
                    vec += [g1, b1, bias1, br, g2, b2, bias2]
                    self.flops += 2 * rows * (3 * 2 * c * c + 3 * c * c + 2 * c * c)
        if layers > 0:
            # ---- to_in: GroupNorm(32, eps 1e-6) + Conv1d(k = 1); gain / bias folded: W (g xn + b) = (W g) xn + W b ----
            g_in, b_in = sd[p + "to_in.0.weight"].double(), sd[p + "to_in.0.bias"].double()
            w_in = sd[p + "to_in.1.weight"].reshape(c, c).double()
            w_in_f = (w_in * g_in.unsqueeze(0)).float()
            for half in range(c // 64):
                wtile(w_in_f[64 * half: 64 * half + 64][:, acc], 0)
            vec.append((w_in @ b_in + sd[p + "to_in.1.bias"].double()).float())
            self.flops += 2 * rows * c * c
        cross0 = len(self.cross_layers)
        for li in range(layers):
            bp = p + f"blocks.{li}."
            ap = bp + "attention."
            g_q, b_q = sd[ap + "norm.weight"], sd[ap + "norm.bias"]
            g_c, b_c = sd[ap + "norm_context.weight"], sd[ap + "norm_context.bias"]
            wq, wkv = sd[ap + "to_q.weight"], sd[ap + "to_kv.weight"]
            wo, bo = sd[ap + "attention.to_out.weight"], sd[ap + "attention.to_out.bias"]
            wq_f, bq_f = wq * g_q.unsqueeze(0), wq @ b_q
            wkv_f, bkv_f = wkv * g_c.unsqueeze(0), wkv @ b_c
            for h in range(heads):
                wtile(wq_f[64 * h: 64 * h + 64][:, acc], 0)
                wtile(wkv_f[64 * h: 64 * h + 64][:, acc], 0)
                wtile(wkv_f[mid + 64 * h: mid + 64 * h + 64][:, acc], 0)
                wtile(wo[:, 64 * h: 64 * h + 64][:, slot], 1)
            # k bias dropped (softmax-invariant), v bias folded into the output bias (see tblock())
            vec += [bq_f, bo + wo @ bkv_f[mid:]]
            self.flops += 2 * rows * c * 3 * mid + 4 * rows * rows * mid + 2 * rows * mid * c
            if cross:
                cp = bp + "cross_attention."
                self.cross_layers.append(cp)
                layer = len(self.cross_layers) - 1 - cross0
                g_q, b_q = sd[cp + "norm.weight"], sd[cp + "norm.bias"]
                wq = sd[cp + "to_q.weight"]
                wo, bo = sd[cp + "attention.to_out.weight"], sd[cp + "attention.to_out.bias"]
                wq_f, bq_f = wq * g_q.unsqueeze(0), wq @ b_q
                for h in range(heads):
                    wtile(wq_f[64 * h: 64 * h + 64][:, acc], 0)
                    desc.append(2 | ((layer << 4 | h) << 2))          # K rows of (layer, head)
                    desc.append(3 | ((layer << 4 | h) << 2))          # V rows
                    wtile(wo[:, 64 * h: 64 * h + 64][:, slot], 1)
                vec += [bq_f, bo]
                self.flops += 2 * rows * c * mid + 4 * rows * self.n_ctx * mid + 2 * rows * mid * c
            fp = bp + "feed_forward."
            w1, b1 = sd[fp + "0.weight"], sd[fp + "0.bias"]
            w2, b2 = sd[fp + "2.weight"], sd[fp + "2.bias"]
            last = li == layers - 1
            if last:
                # Transformer1d.to_out folded in: y = Wout (x + W2 h + b2) + bout = (Wout W2) h + Wout x + (Wout b2 + bout)
                wout, bout = sd[p + "to_out.1.weight"].reshape(c, c).double(), sd[p + "to_out.1.bias"].double()
                w2, b2 = (wout @ w2.double()).float(), (wout @ b2.double() + bout).float()
            for h in range(nff):
                wtile(w1[64 * h: 64 * h + 64][:, acc], 0)
                wtile(w2[:, 64 * h: 64 * h + 64][:, slot], 1)
            if last:
                wout_f = sd[p + "to_out.1.weight"].reshape(c, c)
                for e in range(c // 64):
                    wtile(wout_f[:, acc[64 * e: 64 * e + 64]], 1)
                self.flops += 2 * rows * c * c
            vec += [b1, b2]
            self.flops += 2 * 2 * rows * c * w1.shape[0]
        nvec = self._tf128_nvec(c, layers, cross, res_kind, len(res_blocks))
        v = torch.cat([t.float().reshape(-1) for t in vec])
        assert v.numel() <= nvec, (v.numel(), nvec)
        v = torch.cat([v, torch.zeros(nvec - v.numel())])
        if y is None:
            y = self._new(rows, c)
        op = rt.MdtOp()
        op.kind = rt.OP_TF128
        if res_kind:
            op.res = res_skips[0].ref()
            op._film = ("ss", film0)
            op.i[rt.F_RES_KIND], op.i[rt.F_N_RES] = res_kind, len(res_blocks)
            cin1 = (2 * c if res_kind == 2 else c)
            op.i[rt.F_RES_PAIR1], op.i[rt.F_RES_PAIR2] = int(cin1 // groups == 32), int(c // groups == 32)
            op.i[rt.F_NFILM] = (2 * c * len(res_blocks) + 255) // 256 * 256
            op.f[rt.FF_EPS_RES], op.f[rt.FF_SKIP_SCALE] = 1e-5, float(scale_b if res_kind == 2 else 1.0)
            for k in range(1, len(res_skips)):      # whole tensors apart, ascending (produced) / descending (consumed)
                step = rows * c if res_kind == 1 else -rows * c
                assert res_skips[k].off == res_skips[0].off + k * step and res_skips[k].space == rt.SP_ACT
        op.a, op.out = x.ref(), y.ref()
        wp = p or res_blocks[0]                    # (a launch of ResNet blocks alone is named after its first block)
        op.w = _ref(rt.SP_WEIGHT, self.W.add(wp + "tf128.tiles", torch.cat(tiles)))
        op.i[rt.W_KB] = min(sum(t.numel() for t in tiles) * 4 // 1024, 4096)   # (at most an L2's worth: the head of the stream)
        op.bias = _ref(rt.SP_WEIGHT, self.W.add(wp + "tf128.vec", v))
        op.p0 = _ref(rt.SP_WEIGHT, self.W.add(wp + "tf128.desc", torch.tensor(desc, dtype=torch.int32).view(torch.float32)))
        i = op.i
        i[rt.F_C], i[rt.F_T], i[rt.F_NT], i[rt.F_NVEC] = c, rows, len(desc), nvec
        i[rt.F_TK], i[rt.F_KV_BSTRIDE], i[rt.F_LDKV], i[rt.F_HEADS] = self.n_ctx, self.n_ctx, 2 * mid, heads
        i[rt.F_HAS_IN], i[rt.F_NBLOCKS], i[rt.F_NFF], i[rt.F_NPOST] = int(layers > 0), layers, nff, (c // 64 if layers > 0 else 0)
        i[rt.F_CROSS], i[rt.F_KV_LSTRIDE], i[rt.F_WF32] = int(cross), self.n_ctx * 2 * mid, int(self.wf32)
        op.f[0], op.f[1], op.f[2] = 1e-5, float(cfg.head_features) ** -0.5, 1e-6
        if cross:
            op._kv = ("kv", cross0)
            op.a2 = _ref(rt.SP_ACT, 0)
        self._emit(op)
        if free_input:
            self._free(x)
        return y

    def tf256_ok(self, c: int, rows: int, layers: int, cross: bool) -> bool:
        if not ((self.tf256 or self.tf256_pair) and self.ring_mode and self.fuse_blocks and self.fold_out):
            return False
        if c != 256 or rows > 16 or 16 % rows or self.cfg.head_features != 64 or layers < 1:
            return False
        if self.cfg.mid_features != 512 or c * self.cfg.ff_mult != 512:
            return False
        return not (cross and (16 // rows) * self.n_ctx > 48)

    def transformer_fused256(self, x: Ten, p: str, c: int, layers: int, cross: bool, free_input: bool,
                             y: Optional[Ten] = None) -> Ten:
        """Transformer1d.forward (modules.py:519-524) of a 256-channel level as ONE MDT_OP_TF256 (csrc/k_tf256.hip): 32 KB
        sub-tiles in consumption order, two scratch descriptors behind every sub-block (the wave pairs' partial sums meet
        there; the first one carries the next sub-block's vectors), K columns of residual-stream consumers in accumulator
        order.  Pair-split form (self.tf256 False): TWO descriptor tables over one tile pool -- half hh lists only its heads /
        hidden chunks / to_in output chunks / folded to_out k chunks -- and two more barrier-only descriptors per sub-block for
        the hand-off between the two workgroups."""
        cfg, sd = self.cfg, self.sd
        rows, mid = x.rows, cfg.mid_features
        heads, nff = mid // 64, c * cfg.ff_mult // 64
        nsplit = 1 if self.tf256 else 2
        acc = torch.tensor([16 * (2 * (k >> 5) + ((k & 7) >> 2)) + 4 * ((k >> 3) & 3) + (k & 3) for k in range(c)])
        slot = torch.tensor(self._SLOT_PERM)
        P, O, K, V, SCR, SCRV = 0, 1, 2, 3, 4, 5
        tiles: List[torch.Tensor] = []
        descs: List[List[int]] = [[] for _ in range(nsplit)]
        vecs: List[torch.Tensor] = []

        def half_of(index: int, count: int) -> int:          # which workgroup of the pair takes item `index` of `count`
            return index * nsplit // count

        def sub(t: torch.Tensor, kind: int, hh: int) -> None:
            descs[hh].append(kind | (len(tiles) << 3))
            tiles.append(self._wtile(t))

        def ptile(w: torch.Tensor, hh: int) -> None:         # [64][256] projection tile -> its two K halves
            wp = w[:, acc]
            sub(wp[:, :128], P, hh)
            sub(wp[:, 128:], P, hh)

        def otile(w: torch.Tensor, hh: int) -> None:         # [256][64] output tile -> its two row halves
            sub(w[:128], O, hh)
            sub(w[128:], O, hh)

        def end_subblock(v: List[torch.Tensor], last: bool = False) -> None:
            flat = torch.cat([t.float().reshape(-1) for t in v])
            assert flat.numel() <= 768
            vecs.append(torch.cat([flat, torch.zeros(768 - flat.numel())]))
            nxt = len(vecs)                                  # index of the next sub-block's vectors
            for d in descs:
                d.append(SCR if last else (SCRV | ((((768 * nxt) // 256) << 1 | (nxt & 1)) << 3)))
                d.append(SCR)
                if nsplit == 2:
                    d += [SCR, SCR]                          # the hand-off's two barriers

        g_in, b_in = sd[p + "to_in.0.weight"].double(), sd[p + "to_in.0.bias"].double()
        w_in = sd[p + "to_in.1.weight"].reshape(c, c).double()
        w_in_f = (w_in * g_in.unsqueeze(0)).float()
        for ch in range(c // 64):
            ptile(w_in_f[64 * ch: 64 * ch + 64], half_of(ch, c // 64))
        end_subblock([(w_in @ b_in + sd[p + "to_in.1.bias"].double()).float()])
        self.flops += 2 * rows * c * c
        cross0 = len(self.cross_layers)
        for li in range(layers):
            bp = p + f"blocks.{li}."
            ap = bp + "attention."
            g_q, b_q = sd[ap + "norm.weight"], sd[ap + "norm.bias"]
            g_c, b_c = sd[ap + "norm_context.weight"], sd[ap + "norm_context.bias"]
            wq, wkv = sd[ap + "to_q.weight"], sd[ap + "to_kv.weight"]
            wo, bo = sd[ap + "attention.to_out.weight"], sd[ap + "attention.to_out.bias"]
            wq_f, bq_f = wq * g_q.unsqueeze(0), wq @ b_q
            wkv_f, bkv_f = wkv * g_c.unsqueeze(0), wkv @ b_c
            for h in range(heads):
                hh = half_of(h, heads)
                ptile(wq_f[64 * h: 64 * h + 64], hh)
                ptile(wkv_f[64 * h: 64 * h + 64], hh)
                ptile(wkv_f[mid + 64 * h: mid + 64 * h + 64], hh)
                otile(wo[:, 64 * h: 64 * h + 64][:, slot], hh)
            end_subblock([bq_f, bo + wo @ bkv_f[mid:]])
            self.flops += 2 * rows * c * 3 * mid + 4 * rows * rows * mid + 2 * rows * mid * c
            if cross:
                cp = bp + "cross_attention."
                self.cross_layers.append(cp)
                layer = len(self.cross_layers) - 1 - cross0
                g_q, b_q = sd[cp + "norm.weight"], sd[cp + "norm.bias"]
                wq = sd[cp + "to_q.weight"]
                wo, bo = sd[cp + "attention.to_out.weight"], sd[cp + "attention.to_out.bias"]
                wq_f, bq_f = wq * g_q.unsqueeze(0), wq @ b_q
                for h in range(heads):
                    hh = half_of(h, heads)
                    ptile(wq_f[64 * h: 64 * h + 64], hh)
                    descs[hh].append(K | ((layer << 4 | h) << 3))
                    descs[hh].append(V | ((layer << 4 | h) << 3))
                    otile(wo[:, 64 * h: 64 * h + 64][:, slot], hh)
                end_subblock([bq_f, bo])
                self.flops += 2 * rows * c * mid + 4 * rows * self.n_ctx * mid + 2 * rows * mid * c
            fp = bp + "feed_forward."
            w1, b1 = sd[fp + "0.weight"], sd[fp + "0.bias"]
            w2, b2 = sd[fp + "2.weight"], sd[fp + "2.bias"]
            last = li == layers - 1
            if last:
                wout, bout = sd[p + "to_out.1.weight"].reshape(c, c).double(), sd[p + "to_out.1.bias"].double()
                w2, b2 = (wout @ w2.double()).float(), (wout @ b2.double() + bout).float()
            for h in range(nff):
                hh = half_of(h, nff)
                ptile(w1[64 * h: 64 * h + 64], hh)
                otile(w2[:, 64 * h: 64 * h + 64][:, slot], hh)
            if last:
                wout_f = sd[p + "to_out.1.weight"].reshape(c, c)
                for e in range(c // 64):
                    otile(wout_f[:, acc[64 * e: 64 * e + 64]], half_of(e, c // 64))
                self.flops += 2 * rows * c * c
            end_subblock([b1, b2], last=last)
            self.flops += 2 * 2 * rows * c * w1.shape[0]
        assert all(len(d) == len(descs[0]) for d in descs)
        desc = [v for d in descs for v in d]
        if y is None:
            y = self._new(rows, c)
        op = rt.MdtOp()
        op.kind = rt.OP_TF256
        op.a, op.out = x.ref(), y.ref()
        v = torch.cat(vecs)
        op.w = _ref(rt.SP_WEIGHT, self.W.add(p + "tf256.tiles", torch.cat(tiles)))
        op.i[rt.W_KB] = min(sum(t.numel() for t in tiles) * 4 // 1024, 4096)
        op.bias = _ref(rt.SP_WEIGHT, self.W.add(p + "tf256.vec", v))
        op.p0 = _ref(rt.SP_WEIGHT, self.W.add(p + "tf256.desc", torch.tensor(desc, dtype=torch.int32).view(torch.float32)))
        i = op.i
        i[rt.F_C], i[rt.F_T], i[rt.F_NT], i[rt.F_NVEC] = c, rows, len(descs[0]), v.numel()
        i[rt.F_NSPLIT], i[rt.F_PAIR_STRIDE] = nsplit, self.pair_stride
        if nsplit == 2:
            op.p2, op.p3 = _ref(rt.SP_EXT0 + EXT_XFLAGS, 0), _ref(rt.SP_EXT0 + EXT_XBUF, 0)
            self.xchg_tokens = max(self.xchg_tokens, rows)
        i[rt.F_TK], i[rt.F_KV_BSTRIDE], i[rt.F_LDKV], i[rt.F_HEADS] = self.n_ctx, self.n_ctx, 2 * mid, heads
        i[rt.F_HAS_IN], i[rt.F_NBLOCKS], i[rt.F_NFF], i[rt.F_NPOST] = 1, layers, nff, 2 * (c // 64)
        i[rt.F_CROSS], i[rt.F_KV_LSTRIDE], i[rt.F_WF32] = int(cross), self.n_ctx * 2 * mid, int(self.wf32)
        op.f[0], op.f[1], op.f[2] = 1e-5, float(cfg.head_features) ** -0.5, 1e-6
        if cross:
            op._kv = ("kv", cross0)
            op.a2 = _ref(rt.SP_ACT, 0)
        self._emit(op)
        if free_input:
            self._free(x)
        return y

    # ------------------------------------------------------------------ a chain of ResNet blocks of a 256-channel level
    def res256_ok(self, p: str, c: int, rows: int, groups: int, two_source: bool) -> bool:
        """Can the ResnetBlock1d at prefix p be a link of an MDT_OP_RES256 chain (csrc/k_res256.hip)?  256 channels, GroupNorm
        groups of 32 channels (64 on the 2C-channel input of an up-path block), C -> C or 2C -> C with to_out."""
        if not (self.use_rconv and self.ring_mode and self.fuse_blocks) or self.res256_mode == "0":
            return False
        if c != 256 or rows > 16 or 16 % rows or groups <= 0:
            return False
        if self.res256_mode not in ("1", "auto", "split", "whole"):
            return False
        if self.res256_mode == "whole" and not (self.tf256 or rows == 1):      # round 5's policy: the unsplit chain where it won
            return False
        cin = 2 * c if two_source else c
        if cin % groups or c % groups or cin // groups != (64 if two_source else 32) or c // groups != 32:
            return False
        w1 = self.sd.get(p + "block1.project.weight")
        if w1 is None or tuple(w1.shape) != (c, cin, 3) or tuple(self.sd[p + "block2.project.weight"].shape) != (c, c, 3):
            return False
        return ((p + "to_out.weight") in self.sd) == two_source

    def res256_split(self) -> bool:
        """Form of the MDT_OP_RES256 chains: pair-split (k_res256 NSPLIT = 2: two workgroups per 32-row block, half the weight stream
        each, hand-offs inside the launch) in the NARROW program -- batches whose 32-row blocks do not fill the chip, where the
        unsplit chain ran on half the compute units and one k_rconv launch per convolution (26 per evaluation at configs[1]) was
        the faster form through round 5 -- and the unsplit chain in the wide program.  MDT_RES256: auto (default) | split | whole
        (never split; in the narrow program: k_rconv launches, round 5's policy) | 1 (unsplit chain everywhere) | 0 (no chains)."""
        if self.res256_mode in ("1", "whole"):
            return False
        if self.res256_mode == "split":
            return True
        return not self.tf256 and self.tf256_pair

    def resnet_chain256(self, x: Ten, blocks: List[str], kind: int, skips: List[Ten], scale_b: float, y: Ten,
                        free_input: bool, nsplit: Optional[int] = None) -> Ten:
        """ResnetBlock1d.forward (modules.py:193-205) for every prefix in `blocks`, ONE MDT_OP_RES256 launch.  kind 1: x = Block(x),
        block rb's output also stored to skips[rb] (whole tensors apart, ascending); kind 2: x = Block(cat([x, scale_b * skips[rb]]))
        (modules.py:828-829; skips descending, in order of consumption).  Sub-tiles [64][128]: rows 0..31 = output channels 32 ch ..,
        rows 32..63 = channels 128 + 32 ch .. (a wave of the kernel owns a contiguous half of the channels); K columns in
        accumulator order; one token per sample: only the centre tap of the k = 3 convolutions is streamed.
        nsplit = 2 (round 6; default in the narrow program): the pair-split form -- every row block is served by a PAIR of workgroups,
        half hh streams output chunks 2 hh, 2 hh + 1 of every convolution (its sub-tiles `NFF` sub-tiles behind half 0's), the pair
        hands its chunks to each other inside the launch behind every completed accumulator set (wave by wave, no tile of the stream)."""
        sd, c, rows = self.sd, 256, x.rows
        taps = 1 if rows == 1 else 3
        if nsplit is None:
            nsplit = 2 if self.res256_split() else 1
        split = nsplit == 2
        tiles1: List[torch.Tensor] = []                            # half 1's sub-tiles (split form), appended behind half 0's
        acc = torch.tensor([16 * (2 * (k >> 5) + ((k & 7) >> 2)) + 4 * ((k >> 3) & 3) + (k & 3) for k in range(c)])
        W_, S_, X_, XV_ = 0, 1, 2, 3
        tiles: List[torch.Tensor] = []
        desc: List[int] = [XV_ | (0 << 2)]                        # the vectors of block 0 (descriptors = SEGMENTS: a weight
                                                                    # descriptor stands for a run of consecutive sub-tiles)
        vec: List[torch.Tensor] = []
        n = len(blocks)

        def conv_tiles(w: torch.Tensor) -> None:                   # [c][c][k] -> (tap, K half, chunk) sub-tiles
            k = w.shape[2]
            tp = [k // 2] if (k == 3 and taps == 1) else list(range(k))
            for tap in tp:
                wt = w[:, :, tap][:, acc]
                for kh in range(2):
                    for cl in range(2 if split else 4):
                        if desc and (desc[-1] & 3) == W_ and len(desc) > 1:
                            desc[-1] += 1 << 2                  # run-length: one descriptor per RUN of weight sub-tiles
                        else:
                            desc.append(W_ | (1 << 2))
                        for hh in range(2 if split else 1):      # split: ONE table, chunk 2 hh + cl in half hh's stream
                            ch = 2 * hh + cl if split else cl
                            rws = torch.cat([torch.arange(32 * ch, 32 * ch + 32), torch.arange(128 + 32 * ch, 128 + 32 * ch + 32)])
                            (tiles1 if hh else tiles).append(self._wtile(wt[rws][:, 128 * kh: 128 * kh + 128].contiguous()))

        film0 = None
        for rb, bp in enumerate(blocks):
            ss_off = self.ss_total                                  # FiLM vectors inside the shared (scale | shift) row
            self.ss_offsets[bp] = ss_off
            self.ss_total += 2 * c
            film0 = ss_off if film0 is None else film0
            w1, w2 = sd[bp + "block1.project.weight"].float(), sd[bp + "block2.project.weight"].float()
            g1, b1 = sd[bp + "block1.groupnorm.weight"].float(), sd[bp + "block1.groupnorm.bias"].float()
            g2, b2 = sd[bp + "block2.groupnorm.weight"].float(), sd[bp + "block2.groupnorm.bias"].float()
            bias1, bias2 = sd[bp + "block1.project.bias"].float(), sd[bp + "block2.project.bias"].float()
            first_x = XV_ | ((rb + 1) << 2) if rb + 1 < n else X_    # a block's first exchange carries the NEXT block's vectors
            if kind == 1:
                desc.append(first_x)
                conv_tiles(w1)
                desc.append(X_)
                conv_tiles(w2)
                vec += [g1, b1, bias1, g2, b2, bias2]
                self.flops += 2 * rows * taps * c * c * 2
            else:
                wr, br = sd[bp + "to_out.weight"].float(), sd[bp + "to_out.bias"].float()      # [c, 2c, 1]
                desc.append(first_x)
                conv_tiles(w1[:, :c])                               # block1 on x
                desc.append(X_)
                conv_tiles(wr[:, :c])                               # to_out on x
                desc += [S_ | (rb << 2), X_]
                conv_tiles(wr[:, c:])                               # to_out on the skip
                desc += [S_ | (rb << 2), X_]
                conv_tiles(w1[:, c:])                               # block1 on the skip
                desc.append(X_)
                conv_tiles(w2)
                vec += [g1, b1, bias1, br, g2, b2, bias2]
                self.flops += 2 * rows * (taps * 2 * c * c + taps * c * c + 2 * c * c)
        v = torch.cat([t.reshape(-1) for t in vec])
        assert v.numel() == n * (6 if kind == 1 else 9) * c
        for k in range(1, len(skips)):                              # whole tensors apart (ascending produced / descending consumed)
            step = rows * c if kind == 1 else -rows * c
            assert skips[k].off == skips[0].off + k * step and skips[k].space == rt.SP_ACT
        op = rt.MdtOp()
        op.kind = rt.OP_RES256
        op.a, op.out, op.res = x.ref(), y.ref(), skips[0].ref()
        op._film = ("ss", film0)
        n_half = len(tiles)
        tiles = tiles + tiles1
        op.w = _ref(rt.SP_WEIGHT, self.W.add(blocks[0] + f"res256.tiles{nsplit}", torch.cat(tiles)))
        op.i[rt.W_KB] = min(sum(t.numel() for t in tiles) * 4 // 1024, 4096)
        op.bias = _ref(rt.SP_WEIGHT, self.W.add(blocks[0] + "res256.vec", v))
        op.p0 = _ref(rt.SP_WEIGHT, self.W.add(blocks[0] + f"res256.desc{nsplit}", torch.tensor(desc, dtype=torch.int32).view(torch.float32)))
        i = op.i
        ntiles = sum((d >> 2) if (d & 3) == W_ else 1 for d in desc)
        i[rt.F_C], i[rt.F_T], i[rt.F_NT], i[rt.F_NVEC], i[rt.F_NPOST], i[rt.F_HEADS] = c, rows, ntiles, v.numel(), taps, len(desc)
        i[rt.F_RES_KIND], i[rt.F_N_RES], i[rt.F_NFILM], i[rt.F_WF32] = kind, n, 2 * c * n, int(self.wf32)
        if split:
            # hand-off flags / blocks: the engine's buffers of the pair-split MDT_OP_TF256 ops (same row blocks, same flag lines)
            i[rt.F_NSPLIT], i[rt.F_PAIR_STRIDE], i[rt.F_NFF] = 2, self.pair_stride, n_half
            op.a2, op.p1 = _ref(rt.SP_EXT0 + EXT_XFLAGS, 0), _ref(rt.SP_EXT0 + EXT_XBUF, 0)
            self.xchg_tokens = max(self.xchg_tokens, rows)
        op.f[rt.FF_EPS_RES], op.f[rt.FF_SKIP_SCALE] = 1e-5, float(scale_b if kind == 2 else 1.0)
        self._emit(op)
        if free_input:
            self._free(x)
        return y

    def attention_layer(self, t: Ten, p: str, cross_index: Optional[int], copy16: Optional[Ten] = None) -> None:
        """x = Attention(x[, context]) + x, in place on t (modules.py:401-410, :457-459).  copy16 (plain-bf16 mode): the output
        projection also writes a bf16 copy of the new x, the A operand of the feed-forward block that follows."""
        cfg = self.cfg
        c, mid = t.ld, cfg.mid_features
        if cross_index is None and self.b16_ok(c) and self.b16_ok(mid) and self.qkv_merge:
            # plain-bf16 mode: q | k | v as ONE GEMM over ONE normalised operand -- norm and norm_context differ only in their
            # gains / biases, which fold into the weights (W (g xn + b) = (W g) xn + W b): one LayerNorm pass, one launch
            sd = self.sd
            wq, wkv = sd[p + "to_q.weight"].double(), sd[p + "to_kv.weight"].double()
            gq, bq = sd[p + "norm.weight"].double(), sd[p + "norm.bias"].double()
            gc, bc = sd[p + "norm_context.weight"].double(), sd[p + "norm_context.bias"].double()
            w = torch.cat([wq * gq.unsqueeze(0), wkv * gc.unsqueeze(0)]).float()
            bias = torch.cat([wq @ bq, wkv @ bc]).float()
            qkv = self._new16(t.rows, 3 * mid)          # bf16 out of the GEMM's epilogue: half the bytes of the attention core
            self.gemm(t, (p + "qkv.folded", w), 3 * mid, qkv, cin=c, pro=rt.PRO_LAYERNORM,
                      gain=self._ones(c), nbias=self._zeros(c), eps=1e-5,
                      bias_off=self.W.add(p + "qkv.folded.bias", bias))
            ao = self._new16(t.rows, mid)
            self.attn(qkv, qkv.ref(), t.rows, t.rows, ao, ldkv=3 * mid, kcol=mid, kv16=True)
            self._free(qkv)
            self.gemm(ao, self._lin_w(p + "attention.to_out.weight"), c, t, cin=mid,
                      bias_off=self._vec(p + "attention.to_out.bias", c), res=t, copy16=copy16)
            self._free(ao)
            return
        b16 = self.b16_ok(c) and self.b16_ok(mid)
        q = self._new16(t.rows, mid) if b16 else self._new(t.rows, mid)
        self.gemm(t, self._lin_w(p + "to_q.weight"), mid, q, cin=c, pro=rt.PRO_LAYERNORM,
                  gain=self._vec(p + "norm.weight", c), nbias=self._vec(p + "norm.bias", c), eps=1e-5)
        ao = self._new16(t.rows, mid) if self.b16_ok(mid) else self._new(t.rows, mid)
        if cross_index is None:
            kv = self._new16(t.rows, 2 * mid) if b16 else self._new(t.rows, 2 * mid)
            self.gemm(t, self._lin_w(p + "to_kv.weight"), 2 * mid, kv, cin=c, pro=rt.PRO_LAYERNORM,
                      gain=self._vec(p + "norm_context.weight", c), nbias=self._vec(p + "norm_context.bias", c),
                      eps=1e-5)
            self.attn(q, kv.ref(), t.rows, t.rows, ao, kv16=b16)
            self._free(kv)
        else:
            self.attn(q, ("kv", cross_index), self.n_ctx, self.n_ctx, ao)
        self._free(q)
        self.gemm(ao, self._lin_w(p + "attention.to_out.weight"), c, t, cin=mid,
                  bias_off=self._vec(p + "attention.to_out.bias", c), res=t, copy16=copy16 if ao.b16 else None)
        self._free(ao)

    def attention_single_token(self, t: Ten, p: str) -> Ten:
        """Self-attention over ONE token per sample (configs[2]'s 256-channel level): the softmax of a single score is 1 for
        every head, so the block is x + Wo (Wv LN_ctx(x)) + bo (modules.py:401-410, :350-364 with n = 1) -- the query / key
        projections and the attention core drop out exactly.  One GEMM with the LayerNorm prologue and the residual instead of
        four projections, the core and a reduce launch: W' = Wo Wv diag(g_ctx), b' = Wo (Wv b_ctx) + bo, folded in fp64.
        Returns the new residual stream (the GEMM's column tiles all read the whole input row: not in place)."""
        sd, c, mid = self.sd, t.ld, self.cfg.mid_features
        wv = sd[p + "to_kv.weight"].double()[mid:]                                 # [mid, C]
        gc, bc = sd[p + "norm_context.weight"].double(), sd[p + "norm_context.bias"].double()
        wo, bo = sd[p + "attention.to_out.weight"].double(), sd[p + "attention.to_out.bias"].double()   # [C, mid]
        w = (wo @ (wv * gc.unsqueeze(0))).float()
        bias = (wo @ (wv @ bc) + bo).float()
        out = self._new(t.rows, c)
        self.gemm(t, (p + "single_token.folded", w), c, out, cin=c, pro=rt.PRO_LAYERNORM,
                  gain=self._ones(c), nbias=self._zeros(c), eps=1e-5,
                  bias_off=self.W.add(p + "single_token.folded.bias", bias), res=t)
        self._free(t)
        return out

    def fold_ok(self) -> bool:
        """Fold a layer-by-layer cross-attention onto the normalised context?  Worth it when the hoisted K / V rows are the
        launch's main traffic (many keys); with few keys the doubled projections cost more than they save."""
        return self.fold_ctx and self.n_ctx >= 32 and self.n_ctx <= 64 and self.cfg.ctx_features == 128

    def attention_layer_folded(self, t: Ten, p: str) -> None:
        """x = Attention(x, context) + x (modules.py:401-410, :350-364) WITHOUT materialised keys / values.  With
        c = (ctx - mean) / std, LN_ctx(ctx) = c g + b, k_h = c A_h + a_h and v_h = c B_h + b_h (A_h = diag(g) Wk_h^T, ...):
          S_h = q_h k_h^T = (q_h A_h^T) c^T + const per query  ->  q'_h = Mq_h LN(x),  Mq_h = diag(g) Wk_h^T Wq_h   [128 x C]
          out = sum_h (P_h v_h) Wo_h^T + bo = sum_h (P_h c) N_h^T + bo',  N_h = Wo_h Wv_h diag(g),  bo' = bo + Wo (Wv b)
        (the constant drops out of the softmax; rows of P sum to one).  Three ops: GEMM(LN) -> MDT_OP_ATTN_CTX -> GEMM."""
        cfg, sd = self.cfg, self.sd
        c, mid, F_ = t.ld, cfg.mid_features, cfg.ctx_features
        H = cfg.heads
        gq, bq = self._vec(p + "norm.weight", c), self._vec(p + "norm.bias", c)
        g_c, b_c = sd[p + "norm_context.weight"].double(), sd[p + "norm_context.bias"].double()
        wq = sd[p + "to_q.weight"].double()                                    # [mid, C]
        wkv = sd[p + "to_kv.weight"].double()                                  # [2 mid, F]
        wk, wv = wkv[:mid], wkv[mid:]
        wo, bo = sd[p + "attention.to_out.weight"].double(), sd[p + "attention.to_out.bias"].double()   # [C, mid]
        mq = torch.cat([(wk[64 * h: 64 * h + 64] * g_c.unsqueeze(0)).T @ wq[64 * h: 64 * h + 64] for h in range(H)])   # [H F, C]
        nn_ = torch.cat([wo[:, 64 * h: 64 * h + 64] @ (wv[64 * h: 64 * h + 64] * g_c.unsqueeze(0)) for h in range(H)], dim=1)  # [C, H F]
        bo2 = bo + wo @ (wv @ b_c)
        qf = self._new(t.rows, H * F_)
        self.gemm(t, (p + "folded.q", mq.float()), H * F_, qf, cin=c, pro=rt.PRO_LAYERNORM, gain=gq, nbias=bq, eps=1e-5)
        r = self._new(t.rows, H * F_)
        op = rt.MdtOp()
        op.kind = rt.OP_ATTN_CTX
        op.a, op.out = qf.ref(), r.ref()
        op._kv = ("chat", 0)
        op.a2 = _ref(rt.SP_ACT, 0)
        i = op.i
        i[rt.A_T], i[rt.A_TK], i[rt.A_HEADS] = t.rows, self.n_ctx, H
        i[rt.A_LDQ], i[rt.A_LDKV], i[rt.A_LDO], i[rt.A_KV_BSTRIDE] = F_, F_, F_, self.n_ctx
        i[rt.A_SPLIT] = int(self.gemm_mode == "bf16x3" and self.ctx_split)      # scores in the mode's own arithmetic
        op.f[0] = float(cfg.head_features) ** -0.5
        self._emit(op)
        self.flops += 2 * 2 * t.rows * H * self.n_ctx * F_
        self._free(qf)
        self.gemm(r, (p + "folded.out", nn_.float()), c, t, cin=H * F_, bias_off=self.W.add(p + "folded.out.bias", bo2.float()), res=t)
        self._free(r)
        self.has_chat = True

    def transformer(self, x: Ten, p: str, c: int, layers: int, cross: bool, free_input: bool = True,
                    y_out: Optional[Ten] = None) -> Ten:
        """Transformer1d.forward (modules.py:519-524).  y_out: where the result should be written (a tensor allocated by the
        caller, e.g. next to the level's other skip tensors); the lowerings that cannot place their output return their own."""
        cfg = self.cfg
        assert x.ld == c and c % 32 == 0
        if self.tf128_ok(c, x.rows, layers, cross):
            return self.transformer_fused128(x, p, c, layers, cross, free_input, y=y_out)
        if self.tf256_ok(c, x.rows, layers, cross):
            return self.transformer_fused256(x, p, c, layers, cross, free_input, y=y_out)
        # plain-bf16 mode (round 6, MDT_RES16): the blocks' residual stream as ONE bf16 tensor -- to_in writes it, every residual
        # projection reads and writes it (MDT_G_WFMT 38), the LayerNorm passes read it (MDT_OP_PREP16 with a bf16 input), the
        # feed-forward up-projection and to_out consume it as their A operand directly
        stream16 = (self.res16 and self.gemm_mode == "bf16" and self.b16_ok(c) and self.b16_ok(c * cfg.ff_mult)
                    and self.b16_ok(cfg.mid_features) and self.qkv_merge and c <= 1024 and not (cross and self.fold_ok())
                    and not (x.rows == 1 and self.t1_fold))
        t = self._new16(x.rows, c) if stream16 else self._new(x.rows, c)
        gi, bi = self._vec(p + "to_in.0.weight", c), self._vec(p + "to_in.0.bias", c)
        wi, bias_i = self._conv_w(p + "to_in.1.weight", c, c), self._vec(p + "to_in.1.bias", c)
        if self.rconv_ok(x.rows, c, 1, c // 32):
            self.rconv(x, self.sd[p + "to_in.1.weight"], p + "to_in.1.weight", t, taps=1, bias_off=bias_i,
                       gn=(gi, bi, c // 32, 1e-6, False))
        elif self.gn_act_ok(x.rows, c, 32, c // 32):
            xa = self.gn_act(x, 32, c // 32, 1e-6, gi, bi, False)
            self.gemm(xa, wi, c, t, cin=c, bias_off=bias_i)
            self._free(xa)
        else:
            st = self.gn_stats(x, 32, c // 32, 1e-6)
            self.gemm(x, wi, c, t, cin=c, bias_off=bias_i, pro=rt.PRO_GROUPNORM, gain=gi, nbias=bi,
                      stats=st, groups=32, gsize=c // 32, pro_silu=0)
            self._free(st)
        if free_input:
            self._free(x)
        fused = self.can_fuse_transformer(c, t.rows, cross)
        # Kernel variant, from measurements at B = 1024 (tools/tblock_bench.py, MI355X):
        #   C = 128 (16384 rows): 64-row workgroups          self 46 us (layer-by-layer 120), ff 16 (32)
        #   C = 256 ( 4096 rows): 16-row feature-split ones  self 41 us (73), ff 22 (34); the 64-row kernel would
        #                         fill only 64 CUs (self 70 us)
        #   C = 256, 32-row workgroups + loader waves (k_tblock32, self-attention / feed-forward only): the 16-row form
        #                         sits on the L2 -> LDS bandwidth roof (its weight stream is read by 256 workgroups)
        variant = 2 if c == 256 else 0
        split = 3 if variant == 2 else variant     # self / cross: two workgroups per row block
        keys16 = (16 // t.rows) * self.n_ctx                        # context rows per 16 token rows
        ring_x = (c == 128 and keys16 <= 16) or (variant == 2 and keys16 <= 48)
        # C = 256: the sub-blocks of a transformer hand the residual stream on as (x, second head group's partial)
        # through ping-pong buffers, so the head split needs no reduce launch (every sub-block must be a ring kernel)
        chain = fused and split == 3 and (not cross or ring_x)
        pend: Optional[Ten] = None
        y_fold: Optional[Ten] = None
        for i in range(layers):
            bp = p + f"blocks.{i}."
            if chain:
                steps = [(rt.TB_SELF, bp + "attention.", None, True)]
                if cross:
                    self.cross_layers.append(bp + "cross_attention.")
                    steps.append((rt.TB_CROSS, bp + "cross_attention.", len(self.cross_layers) - 1, True))
                # the hidden chunks of a feed-forward block split like heads, except in the transformer's last
                # block, whose output leaves the chain (and may carry the folded closing convolution)
                steps.append((rt.TB_FF, bp + "feed_forward.", None, i + 1 < layers))
                for mode, name, ci, two in steps:
                    last_ff = mode == rt.TB_FF and i == layers - 1 and self.fold_out
                    nxt = y_out if (last_ff and y_out is not None) else self._new(t.rows, c)
                    po = self._new(t.rows, c) if two else None
                    if mode == rt.TB_FF and i == layers - 1 and self.fold_out:
                        # the transformer's closing 1x1 convolution rides on the last feed-forward block (below)
                        self.tblock(t, mode, name, ci, variant=4, x_out=nxt, p_in=pend, p_out=None,
                                    post=(self.sd[p + "to_out.1.weight"], self.sd[p + "to_out.1.bias"], nxt))
                        y_fold = nxt
                    else:
                        self.tblock(t, mode, name, ci, variant=4, x_out=nxt, p_in=pend, p_out=po)
                    self._free(t)
                    if pend is not None:
                        self._free(pend)
                    t, pend = nxt, po
                continue
            if fused:
                if t.rows == 1 and self.t1_fold:
                    t = self.attention_single_token(t, bp + "attention.")
                else:
                    self.tblock(t, rt.TB_SELF, bp + "attention.", variant=split)
                if cross:
                    self.cross_layers.append(bp + "cross_attention.")
                    xv = split
                    if ring_x:
                        self.tblock(t, rt.TB_CROSS, bp + "cross_attention.", len(self.cross_layers) - 1,
                                    variant=xv)
                    elif self.fold_ok():
                        self.cross_layers.pop()
                        self.attention_layer_folded(t, bp + "cross_attention.")
                    else:
                        # q-GEMM + attention + out-GEMM (the pre-ring fused cross kernels measured slower than this:
                        # 78 us against ~62 us, their per-head K/V loads were not pipelined)
                        self.attention_layer(t, bp + "cross_attention.", len(self.cross_layers) - 1)
                ffv = split
                if i == layers - 1 and self.fold_out and ffv in (0, 2):
                    y_fold = y_out if y_out is not None else self._new(t.rows, c)
                    self.tblock(t, rt.TB_FF, bp + "feed_forward.", variant=ffv,
                                post=(self.sd[p + "to_out.1.weight"], self.sd[p + "to_out.1.bias"], y_fold))
                else:
                    self.tblock(t, rt.TB_FF, bp + "feed_forward.", variant=ffv)
                continue
            hid = c * cfg.ff_mult
            ff16 = self.b16_ok(c) and self.b16_ok(hid) and self.b16_ok(cfg.mid_features)
            folded = cross and self.fold_ok()
            # plain-bf16 mode: the attention block in front of the feed-forward block hands it x as bf16 (no conversion pass)
            t16 = self._new16(t.rows, c) if (ff16 and not folded and not t.b16) else None      # (a bf16 stream IS that operand)
            if t.rows == 1 and self.t1_fold and t16 is None:
                t = self.attention_single_token(t, bp + "attention.")
            else:
                self.attention_layer(t, bp + "attention.", None, copy16=None if cross else t16)
            if folded:
                self.attention_layer_folded(t, bp + "cross_attention.")
            elif cross:
                self.cross_layers.append(bp + "cross_attention.")
                self.attention_layer(t, bp + "cross_attention.", len(self.cross_layers) - 1, copy16=t16)
            h = self._new16(t.rows, hid) if (self.b16_ok(c) and self.b16_ok(hid)) else self._new(t.rows, hid)
            self.gemm(t16 if t16 is not None else t, self._lin_w(bp + "feed_forward.0.weight"), c * cfg.ff_mult, h, cin=c,
                      bias_off=self._vec(bp + "feed_forward.0.bias", c * cfg.ff_mult), act=1)
            self.gemm(h, self._lin_w(bp + "feed_forward.2.weight"), c, t, cin=c * cfg.ff_mult,
                      bias_off=self._vec(bp + "feed_forward.2.bias", c), res=t)
            self._free(h)
            if t16 is not None:
                self._free(t16)
        if y_fold is not None:                      # the last feed-forward block already produced to_out(t)
            if y_fold is not t:
                self._free(t)
            return y_fold
        y = y_out if y_out is not None else self._new(t.rows, c)
        if self.rconv_ok(t.rows, c, 1, 0):
            self.rconv(t, self.sd[p + "to_out.1.weight"], p + "to_out.1.weight", y, taps=1,
                       bias_off=self._vec(p + "to_out.1.bias", c))
        else:
            self.gemm(t, self._conv_w(p + "to_out.1.weight", c, c), c, y, cin=c, bias_off=self._vec(p + "to_out.1.bias", c))
        self._free(t)
        return y

    def concat(self, a: Ten, b: Ten, scale_b: float) -> Ten:
        out = self._new(a.rows, a.ld + b.ld)
        op = rt.MdtOp()
        op.kind = rt.OP_CONCAT
        op.a, op.a2, op.out = a.ref(), b.ref(), out.ref()
        op.i[rt.C_ROWS], op.i[rt.C_CA], op.i[rt.C_CB] = a.rows, a.ld, b.ld
        op.f[0] = scale_b
        self._emit(op)
        return out

    def patch(self, x: Ten, patch: int, inverse: bool) -> Ten:
        if inverse:                       # (L/p, C*p) -> (L, C)
            out = self._new(x.rows * patch, x.ld // patch)
            rows_long, c_long, ld_in, ld_out = out.rows, out.ld, x.ld, out.ld
        else:                             # (L, C) -> (L/p, C*p)
            out = self._new(x.rows // patch, x.ld * patch)
            rows_long, c_long, ld_in, ld_out = x.rows, x.ld, x.ld, out.ld
        op = rt.MdtOp()
        op.kind = rt.OP_PATCH
        op.a, op.out = x.ref(), out.ref()
        i = op.i
        i[rt.P_ROWS_IN], i[rt.P_C_IN], i[rt.P_LD_IN], i[rt.P_LD_OUT] = rows_long, c_long, ld_in, ld_out
        i[rt.P_PATCH], i[rt.P_INVERSE] = patch, int(inverse)
        self._emit(op)
        return out

    # ------------------------------------------------------------------ programs
    def build_eval(self) -> None:
        cfg, L = self.cfg, self.L
        ps, g = cfg.patch_size, cfg.resnet_groups
        cin, c0 = cfg.in_channels, cfg.level_channels(0)
        self.in_pad = pad16(cin)
        x = Ten(rt.SP_EXT0 + EXT_XIN, 0, L, self.in_pad, cin)
        x = self.resnet(x, "to_in.block.", cin, c0 // ps, 1, free_input=False)
        if ps > 1:
            if (c0 // ps) % 16:
                raise ValueError("channels // patch_size must be a multiple of 16")
            if self.fold_patch and self.ops[-1].kind == rt.OP_RESBLOCK and x.rows % ps == 0:
                # the Patcher's rearrange as the store pattern of the block that feeds it (MDT_K_PATCH_OUT): one launch less
                y = Ten(x.space, x.off, x.rows // ps, x.ld * ps, x.ld * ps)
                self.ops[-1].i[rt.K_PATCH_OUT] = ps
            else:
                y = self.patch(x, ps, inverse=False)
                self._free(x)
            x = y
        skips_list = [[x]]
        for i in range(cfg.num_layers):
            dp = f"downsamples.{i}."
            ci, co, f = cfg.level_channels(i), cfg.level_channels(i + 1), cfg.factors[i]
            y = self._new(x.rows // f, co)
            if self.down_patch_ok(x, ci, co, f):
                self.down_patch(x, dp, ci, co, f, y)
            elif x.rows == f:
                # ONE output token per sample (configs[2]'s last level): taps 0 .. f - 1 and 2 f of the strided convolution
                # (modules.py:62-75: kernel 2 f + 1, stride f, padding f) only ever see padding -- taps f .. 2 f - 1 on inputs
                # 0 .. f - 1 are the whole sum, exactly
                name, wfull = self._conv_w(dp + "downsample.weight", ci, co)
                wcut = wfull.view(co, 2 * f + 1, ci)[:, f: 2 * f].reshape(co, f * ci).contiguous()
                self.gemm(x, (name + "/live_taps", wcut), co, y, cin=ci, bias_off=self._vec(dp + "downsample.bias", co),
                          taps=f, t_stride=f, t_dj=1, t_off=0, r_out=1)
            else:
                self.gemm(x, self._conv_w(dp + "downsample.weight", ci, co), co, y, cin=ci,
                          bias_off=self._vec(dp + "downsample.bias", co), taps=2 * f + 1, t_stride=f, t_dj=1, t_off=-f,
                          r_out=x.rows // f)
            x = y                         # the block input stays alive as a skip of the previous level
            skips: List[Ten] = []
            x_is_skip = False
            if cfg.pre_transformer > 0:
                y = self.transformer(x, dp + "pre_transformer_block.", co, cfg.pre_transformer, False)
                skips.append(y)
                x, x_is_skip = y, True
            nb, nat = cfg.num_blocks[i], cfg.attentions[i]
            blocks = [dp + f"blocks.{j}." for j in range(nb)]
            res_ok = nb > 0 and all(self.res128_ok(bp, co, x.rows, g, False) for bp in blocks)
            together = res_ok and self.tf128_ok(co, x.rows, nat, nat > 0, 1, nb)
            alone = res_ok and not together and self.tf128_ok(co, x.rows, 0, False, 1, nb)
            if together or alone:
                # the level's ResNet blocks in ONE launch, with the transformer that follows them when its vectors fit the LDS
                # behind the ring as well; the outputs (all of them skips of the up path) are whole tensors apart in one
                # allocation, so this launch and the up path's address them by index
                n_out = nb + (1 if nat > 0 else 0)
                base = self.arena.alloc(n_out * x.rows * co)
                outs = [Ten(rt.SP_ACT, base + k * x.rows * co, x.rows, co, co) for k in range(n_out)]
                if together:
                    self.transformer_fused128(x, dp + "transformer.", co, nat, nat > 0, free_input=not x_is_skip,
                                              res=(1, blocks, g, outs[:nb], 1.0), y=outs[-1])
                else:
                    self.transformer_fused128(x, "", co, 0, False, free_input=not x_is_skip,
                                              res=(1, blocks, g, outs[:nb], 1.0), y=outs[nb - 1])
                    if nat > 0:
                        yy = self.transformer(outs[nb - 1], dp + "transformer.", co, nat, True, free_input=False, y_out=outs[nb])
                        if yy is not outs[nb]:       # (a lowering that places its own output: the up path then finds its skips
                            self._free(outs[nb])     # apart and runs its blocks as separate launches)
                            outs[nb] = yy
                skips += outs
                x = outs[-1]
            elif nb > 0 and x.ld == co and all(self.res256_ok(bp, co, x.rows, g, False) for bp in blocks):
                # 256-channel level: the blocks as ONE chained launch (MDT_OP_RES256); outputs whole tensors apart in one allocation
                # (with the transformer's behind them), so that the up path's chain addresses its skips by index
                n_out = nb + (1 if nat > 0 else 0)
                base = self.arena.alloc(n_out * x.rows * co)
                outs = [Ten(rt.SP_ACT, base + k * x.rows * co, x.rows, co, co) for k in range(n_out)]
                self.resnet_chain256(x, blocks, 1, outs[:nb], 1.0, outs[nb - 1], free_input=not x_is_skip)
                if nat > 0:
                    yy = self.transformer(outs[nb - 1], dp + "transformer.", co, nat, True, free_input=False, y_out=outs[nb])
                    if yy is not outs[nb]:
                        self._free(outs[nb])
                        outs[nb] = yy
                skips += outs
                x = outs[-1]
            else:
                for bp in blocks:
                    x = self.resnet(x, bp, co, co, g, free_input=not x_is_skip)
                    skips.append(x)
                    x_is_skip = True
                if nat > 0:
                    x = self.transformer(x, dp + "transformer.", co, nat, True, free_input=False)
                    skips.append(x)
            skips_list.append(skips)
        cb = cfg.level_channels(cfg.num_layers)
        keep = x in skips_list[-1]
        x = self.resnet(x, "bottleneck.pre_block.", cb, cb, g, free_input=not keep)
        if cfg.attentions[-1] > 0:
            x = self.transformer(x, "bottleneck.transformer.", cb, cfg.attentions[-1], True)
        x = self.resnet(x, "bottleneck.post_block.", cb, cb, g)
        for u, i in enumerate(reversed(range(cfg.num_layers))):
            up = f"upsamples.{u}."
            ci, co, f = cfg.level_channels(i + 1), cfg.level_channels(i), cfg.factors[i]
            skips = skips_list.pop()
            n_res = cfg.num_blocks[i] + (1 if cfg.attentions[i] else 0)
            blocks = [up + f"blocks.{j}." for j in range(n_res)]
            # the transformer that follows the blocks (it shares their launch when everything fits k_tf128)
            tfs = []
            if cfg.pre_transformer > 0:
                tfs.append((up + "pre_transformer_block.", cfg.pre_transformer, False))
            if cfg.attentions[i] > 0:
                tfs.append((up + "transformer.", cfg.attentions[i], True))
            cons = skips[len(skips) - n_res:][::-1] if 0 < n_res <= len(skips) else []       # in order of consumption
            nxt = tfs[0] if tfs else ("", 0, False)
            res_ok = (bool(cons) and x.ld == ci and all(self.res128_ok(bp, ci, x.rows, g, True) for bp in blocks)
                      and all(sk.space == rt.SP_ACT and sk.rows == x.rows and sk.ld == ci
                              and sk.off == cons[0].off - k * x.rows * ci for k, sk in enumerate(cons)))
            together = res_ok and nxt[1] > 0 and self.tf128_ok(ci, x.rows, nxt[1], nxt[2], 2, n_res)
            fused = together or (res_ok and self.tf128_ok(ci, x.rows, 0, False, 2, n_res))
            if fused:
                del skips[len(skips) - n_res:]
                if together:
                    x = self.transformer_fused128(x, nxt[0], ci, nxt[1], nxt[2], True, res=(2, blocks, g, cons, 2 ** -0.5))
                    tfs = tfs[1:]
                else:                     # the blocks alone (the transformer behind them is not a k_tf128 launch)
                    x = self.transformer_fused128(x, "", ci, 0, False, True, res=(2, blocks, g, cons, 2 ** -0.5))
                for sk in cons:
                    self._free(sk)
            elif (bool(cons) and x.ld == ci and all(self.res256_ok(bp, ci, x.rows, g, True) for bp in blocks)
                  and all(sk.space == rt.SP_ACT and sk.rows == x.rows and sk.ld == ci
                          and sk.off == cons[0].off - k * x.rows * ci for k, sk in enumerate(cons))):
                del skips[len(skips) - n_res:]
                x = self.resnet_chain256(x, blocks, 2, cons, 2 ** -0.5, self._new(x.rows, ci), True)
                for sk in cons:
                    self._free(sk)
            else:
                for bp in blocks:
                    sk = skips.pop()
                    x = self.resnet_cat(x, sk, 2 ** -0.5, bp, ci, g)
            for sk in skips:              # DownsampleBlock1d emits one more skip than is consumed (:702, :843-845)
                self._free(sk)
            for tp, tl, tc in tfs:
                x = self.transformer(x, tp, ci, tl, tc)
            # ConvTranspose1d k=2f s=f p=f/2 as f output phases of 2 taps each (modules.py:74-81)
            if f % 2:
                raise ValueError("odd upsample factors are not supported")
            wt = self.sd[up + "upsample.weight"]          # [Cin, Cout, 2f]
            bias = self._vec(up + "upsample.bias", co)
            y = self._new(x.rows * f, co)
            last = u == cfg.num_layers - 1
            res = skips_list[0][0] if last else None       # `x += skips_list.pop()` (modules.py:1176)
            if self.up_patch_ok(x, ci, co, f, res):
                self.up_patch(x, up, ci, co, f, y, res)
            else:
                # all f phases in ONE launch (grid.z = phase): each phase alone is a 64..256-workgroup GEMM that runs
                # at launch latency, and the phases are independent
                wall = torch.cat([torch.stack((wt[:, :, ph], wt[:, :, ph + f]), dim=0).permute(2, 0, 1).reshape(co, 2 * ci)
                                  for ph in range(f)])                                       # [f * Cout][2 * Cin]
                self.gemm(x, (f"{up}upsample.weight/phases", wall), co, y, cin=ci, bias_off=bias, taps=2, t_stride=1,
                          t_dj=-1, t_off=0, r_out=x.rows, o_stride=f, o_off=0, res=res, phases=f)
            self._free(x)
            x = y
        self._free(skips_list.pop()[0])
        fold_in = (ps > 1 and self.fold_patch and x.ld % ps == 0
                   and self.resblock_ok(x.rows * ps, c0 // ps, cin, 1, "to_out.block.") and x.ld // ps == pad16(c0 // ps))
        if fold_in:
            # ... and the Unpatcher's as the load pattern of the block behind it (MDT_K_PATCH_IN)
            x = Ten(x.space, x.off, x.rows * ps, x.ld // ps, x.ld // ps)
        elif ps > 1:
            y = self.patch(x, ps, inverse=True)
            self._free(x)
            x = y
        out = Ten(rt.SP_EXT0 + EXT_OUT, 0, L, self.in_pad, cin)
        y = self.resnet(x, "to_out.block.", c0 // ps, cin, 1)
        if fold_in:
            assert self.ops[-1].kind == rt.OP_RESBLOCK
            self.ops[-1].i[rt.K_PATCH_IN] = ps
        # the final resnet wrote into an arena buffer; retarget its last GEMM to the bound output tensor
        last_op = self.ops[-1]
        assert last_op.kind in (rt.OP_GEMM, rt.OP_RESBLOCK)
        last_op.out = out.ref()
        self._free(y)

    def build(self) -> CompiledUNet:
        cfg = self.cfg
        # K/V slots live at the bottom of the per-sample arena (written by `ctx`, read by `eval`)
        mid2 = 2 * cfg.mid_features
        self.build_eval()
        eval_ops = self.ops
        flops_eval = self.flops
        n_cross = len(self.cross_layers)
        kv_base = self.arena.top
        self.kv_slots = [kv_base + i * self.n_ctx * mid2 for i in range(n_cross)]
        act_floats = kv_base + n_cross * self.n_ctx * mid2
        chat_slot = act_floats                      # the normalised context c [n_ctx][ctx_features] per sample (folded layers)
        if self.has_chat:
            act_floats += (self.n_ctx * cfg.ctx_features + 63) // 64 * 64

        # ---- shared arena layout ----
        rows = self.max_time_rows
        mapf = cfg.mapping_features
        ldt = pad16(cfg.channels + 1)
        cn = self._shr_alloc("c_noise", rows)
        temb = self._shr_alloc("time_embed", rows * ldt)
        m1 = self._shr_alloc("map1", rows * mapf)
        m2 = self._shr_alloc("map2", rows * mapf)
        m3 = self._shr_alloc("map3", rows * mapf)
        ss_all = self._shr_alloc("ss_all", rows * self.ss_total)
        ss_cur = self._shr_alloc("ss_cur", self.ss_total)
        kvf = self._shr_alloc("kv_fixed", n_cross * self.n_ctx * mid2)
        self.kv_fixed = [kvf + i * self.n_ctx * mid2 for i in range(n_cross)]
        chat_fixed = self._shr_alloc("chat_fixed", self.n_ctx * cfg.ctx_features)

        # ---- resolve symbolic refs of the eval program; derive the fixed-embedding twin ----
        def resolve(ops, fixed: bool):
            out = []
            for op in ops:
                o = rt.MdtOp()
                C_memmove(o, op)
                if op.kind in (rt.OP_GEMM, rt.OP_GN_ACT, rt.OP_RCONV, rt.OP_RESBLOCK, rt.OP_PREP16, rt.OP_TF128, rt.OP_RES256) and isinstance(getattr(op, "_film", None), tuple):
                    o.p3 = _ref(rt.SP_SHR, ss_cur + op._film[1])
                if op.kind == rt.OP_ATTN_CTX:
                    if fixed:
                        o.a2 = _ref(rt.SP_SHR, chat_fixed)
                        o.i[rt.A_KV_BSTRIDE] = 0
                    else:
                        o.a2 = _ref(rt.SP_ACT, chat_slot)
                elif op.kind in (rt.OP_ATTN, rt.OP_TBLOCK, rt.OP_TF128, rt.OP_TF256) and isinstance(getattr(op, "_kv", None), tuple):
                    idx = op._kv[1]
                    slot = {rt.OP_ATTN: rt.A_KV_BSTRIDE, rt.OP_TBLOCK: rt.B_KV_BSTRIDE, rt.OP_TF128: rt.F_KV_BSTRIDE,
                            rt.OP_TF256: rt.F_KV_BSTRIDE}[op.kind]
                    if fixed:
                        o.a2 = _ref(rt.SP_SHR, self.kv_fixed[idx])
                        o.i[slot] = 0
                    else:
                        o.a2 = _ref(rt.SP_ACT, self.kv_slots[idx])
                out.append(o)
            return out

        programs = {"eval": resolve(eval_ops, False), "eval_fixed": resolve(eval_ops, True)}
        # Both passes of classifier-free guidance as ONE evaluation of a batch of 2B (UNetCFG1d.forward, modules.py:1248-1253:
        # the masked pass sees the same x and time, only the context differs): the first half of the samples attends to its
        # hoisted K/V, the second half to the FixedEmbedding's.  Needs every cross-attention block on a ring kernel (the
        # only ones that take the second K/V pointer); otherwise the engine falls back to two passes.
        cross = [op for op in eval_ops if isinstance(getattr(op, "_kv", None), tuple)]      # (folded layers: never "ring")
        ring = all(op.kind in (rt.OP_TF128, rt.OP_TF256) or
                   (op.kind == rt.OP_TBLOCK and (op.i[rt.B_VARIANT] >= 2 or (op.i[rt.B_VARIANT] == 0 and op.i[rt.B_C] == 128
                                                                         and (16 // op.i[rt.B_T]) * op.i[rt.B_TK] <= 16)))
                   for op in cross)
        dual_multiple = 1
        if cross and ring and os.environ.get("MDT_CFG_DUAL", "1") == "1":
            dual_multiple = max(64 // op.i[rt.F_T] if op.kind == rt.OP_TF128 else 32 // op.i[rt.F_T] if op.kind == rt.OP_TF256
                                else (32 if op.i[rt.B_VARIANT] >= 2 else 64) // op.i[rt.B_T] for op in cross)
            dual = resolve(eval_ops, False)
            for o, op in zip(dual, eval_ops):
                if isinstance(getattr(op, "_kv", None), tuple):
                    o.p1 = _ref(rt.SP_SHR, self.kv_fixed[op._kv[1]])
                    o.i[rt.F_KV2 if op.kind in (rt.OP_TF128, rt.OP_TF256) else rt.B_KV2] = 1
            programs["eval_dual"] = dual

        # ---- time program (m_mode 1) ----
        self.ops, self.flops = [], 0
        t_cn = Ten(rt.SP_SHR, cn, 1, 1)
        t_emb = Ten(rt.SP_SHR, temb, 1, ldt)
        op = rt.MdtOp()
        op.kind = rt.OP_TIME_EMBED
        op.a, op.w, op.out = t_cn.ref(), _ref(rt.SP_WEIGHT, self.W.add("to_time.0.0.weights", self.sd["to_time.0.0.weights"])), t_emb.ref()
        op.i[rt.T_HALF], op.i[rt.T_LD] = cfg.channels // 2, ldt
        self._emit(op)
        t1, t2, t3 = Ten(rt.SP_SHR, m1, 1, mapf), Ten(rt.SP_SHR, m2, 1, mapf), Ten(rt.SP_SHR, m3, 1, mapf)
        self.gemm(t_emb, self._lin_w("to_time.0.1.weight", mapf, ldt), mapf, t1, cin=ldt,
                  bias_off=self._vec("to_time.0.1.bias", mapf), act=1, m_mode=1)
        self.gemm(t1, self._lin_w("to_mapping.0.weight"), mapf, t2, cin=mapf,
                  bias_off=self._vec("to_mapping.0.bias", mapf), act=1, m_mode=1)
        self.gemm(t2, self._lin_w("to_mapping.2.weight"), mapf, t3, cin=mapf,
                  bias_off=self._vec("to_mapping.2.bias", mapf), act=1, m_mode=1)
        # all MappingToScaleShift linears as one GEMM: rows laid out [scale(Cp) | shift(Cp)] per block
        w_all = torch.zeros(self.ss_total, mapf)
        b_all = torch.zeros(self.ss_total)
        for p, off in self.ss_offsets.items():
            w = self.sd[p + "to_scale_shift.to_scale_shift.1.weight"]      # [2C, mapf]
            b = self.sd[p + "to_scale_shift.to_scale_shift.1.bias"]
            c = w.shape[0] // 2
            cp = pad16(c)
            w_all[off: off + c] = w[:c]
            w_all[off + cp: off + cp + c] = w[c:]
            b_all[off: off + c] = b[:c]
            b_all[off + cp: off + cp + c] = b[c:]
        t_ss = Ten(rt.SP_SHR, ss_all, 1, self.ss_total)
        self.gemm(t3, ("scale_shift_all.weight", w_all), self.ss_total, t_ss, cin=mapf,
                  bias_off=self.W.add("scale_shift_all.bias", b_all), pro=rt.PRO_SILU, m_mode=1)
        programs["time"] = self.ops

        # ---- context programs ----
        self.ops, self.flops = [], 0
        ctx = Ten(rt.SP_EXT0 + EXT_CTX, 0, self.n_ctx, cfg.ctx_features)
        fixed = Ten(rt.SP_WEIGHT, self.W.add("fixed_embedding.embedding.weight",
                                             self.sd["fixed_embedding.embedding.weight"][: self.n_ctx]),
                    self.n_ctx, cfg.ctx_features)
        ctx_ops, fixed_ops = [], []
        for idx, p in enumerate(self.cross_layers):
            w = self._lin_w(p + "to_kv.weight")
            gain = self._vec(p + "norm_context.weight", cfg.ctx_features)
            nb = self._vec(p + "norm_context.bias", cfg.ctx_features)
            self.ops = ctx_ops
            self.gemm(ctx, w, mid2, Ten(rt.SP_ACT, self.kv_slots[idx], self.n_ctx, mid2), cin=cfg.ctx_features,
                      pro=rt.PRO_LAYERNORM, gain=gain, nbias=nb, eps=1e-5)
            self.ops = fixed_ops
            self.gemm(fixed, w, mid2, Ten(rt.SP_SHR, self.kv_fixed[idx], self.n_ctx, mid2), cin=cfg.ctx_features,
                      pro=rt.PRO_LAYERNORM, gain=gain, nbias=nb, eps=1e-5, m_mode=2, count_flops=False)
        if self.has_chat:
            # c = LayerNorm(ctx) without affine, once per call, shared by every folded layer: identity GEMM with the
            # LayerNorm prologue (gain 1, bias 0)
            F_ = cfg.ctx_features
            eye = ("ctx_identity", torch.eye(F_))
            ones, zeros = self.W.add("ctx_ones", torch.ones(F_)), self._zeros(F_)
            self.ops = ctx_ops
            self.gemm(ctx, eye, F_, Ten(rt.SP_ACT, chat_slot, self.n_ctx, F_), cin=F_, pro=rt.PRO_LAYERNORM, gain=ones,
                      nbias=zeros, eps=1e-5)
            self.ops = fixed_ops
            self.gemm(fixed, eye, F_, Ten(rt.SP_SHR, chat_fixed, self.n_ctx, F_), cin=F_, pro=rt.PRO_LAYERNORM, gain=ones,
                      nbias=zeros, eps=1e-5, m_mode=2, count_flops=False)
        programs["ctx"], programs["ctx_fixed"] = ctx_ops, fixed_ops
        flops_ctx = self.flops

        return CompiledUNet(cfg=cfg, length=self.L, cond_len=self.n_ctx, in_pad=self.in_pad, weights=self.W.pack(),
                            programs=programs, act_floats=act_floats, shr_floats=self.shr_top,
                            max_time_rows=rows, shr=dict(self.shr), ss_total=self.ss_total, n_cross=n_cross,
                            flops_per_sample_eval=flops_eval, flops_ctx_per_sample=flops_ctx, gemm_mode=self.gemm_mode,
                            weight_index=dict(self.W.index), dual_multiple=dual_multiple, tf256=self.tf256,
                            xchg_tokens=self.xchg_tokens)


def C_memmove(dst: rt.MdtOp, src: rt.MdtOp) -> None:
    import ctypes
    ctypes.memmove(ctypes.byref(dst), ctypes.byref(src), ctypes.sizeof(rt.MdtOp))


def _prod(xs) -> int:
    r = 1
    for x in xs:
        r *= x
    return r


def compile_unet(cfg: UNetConfig, length: int, cond_len: int, sd: Dict[str, torch.Tensor],
                 max_time_rows: int = 1024, gemm_mode: str = "bf16x3", fuse_blocks: bool = True,
                 tf256: bool = False) -> CompiledUNet:
    """tf256: the 256-channel transformers as whole-transformer launches without the head split (k_tf256.hip; the better
    form once the batch fills the chip by itself) instead of one head-split launch per sub-block (k_tblock32.hip)."""
    return UNetCompiler(cfg, length, cond_len, sd, max_time_rows, gemm_mode, fuse_blocks, tf256).build()
