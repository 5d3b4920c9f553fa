"""Drop-in class surface: QMDiffusion (generative.py:718-914) and QMDiffusionForward (:31-225).

Same constructor keywords, attributes (.unet, .diffusion, .fc1, .GELUact, .p_enc_1d, .max_length,
.pred_dim, ...), state_dict key layout (the U-Net appears under unet.*, diffusion.net.* and
diffusion.diffusion.net.*, as in the reference) and sample()/inpaint() signatures.  The sampling
path runs on an MI355X through libmdt_hip.so; there is no CPU implementation behind these classes.
"""
from __future__ import annotations

import os
from typing import Callable, Optional

import torch
import torch.nn as nn

from . import ops, runtime as rt
from .compiler import compile_unet
from .diffusion import (ADPM2Sampler, DiffusionInpainter, DiffusionSampler, KarrasSchedule, LogNormalDistribution,
                        NoiseSource, run_adpm2, run_adpm2_inpaint, scale_weights)
from .engine import UNetEngine, _require_gpu
from .modules import PositionalEncoding1D, UNetCFG1d
from .netspec import forward_unet_config, inverse_unet_config

Tensor = torch.Tensor


class KDiffusion_mod(nn.Module):
    """diffusion.py:770-844: Karras preconditioning around the net (alias 'k')."""

    alias = "k"

    def __init__(self, net: nn.Module, *, sigma_distribution, sigma_data: float, dynamic_threshold: float = 0.0):
        super().__init__()
        if not 0.0 <= dynamic_threshold <= 1.0:
            raise ValueError("dynamic_threshold is a quantile: 0 (static clamp to [-1, 1]) ... 1")
        self.net = net
        self.sigma_data = sigma_data
        self.sigma_distribution = sigma_distribution
        self.dynamic_threshold = dynamic_threshold
        self._owner = None

    def denoise_fn(self, x_noisy: Tensor, sigmas: Optional[Tensor] = None, sigma=None, *, embedding: Tensor,
                   embedding_scale: float = 1.0) -> Tensor:
        """diffusion.py:798-814: a scalar ``sigma`` (the sampling case) or one sigma per sample (``sigmas``, the
        training-time form; samples sharing a sigma are evaluated together)."""
        if sigma is None and sigmas is None:
            raise AssertionError("Either sigma or sigmas must be provided")       # to_batch, diffusion.py:97
        if sigma is not None:
            return self._owner._denoise(x_noisy, sigma, embedding, embedding_scale)
        sig = torch.as_tensor(sigmas, dtype=torch.float32).flatten().cpu()
        if sig.numel() != x_noisy.shape[0]:
            raise AssertionError("sigmas must hold one value per sample")
        out = torch.empty_like(x_noisy, dtype=torch.float32)
        for v in torch.unique(sig):
            rows = (sig == v).nonzero().flatten().to(x_noisy.device)
            out[rows] = self._owner._denoise(x_noisy[rows], v, embedding[rows], embedding_scale)
        return out

    def forward(self, x: Tensor, noise: Optional[Tensor] = None, *, embedding: Tensor, **kwargs) -> Tensor:
        """Training loss of KDiffusion_mod.forward (diffusion.py:820-844) -- plain PyTorch with autograd (train.py);
        the MI355X kernels serve sampling only."""
        from .train import kdiffusion_loss
        return kdiffusion_loss(self._owner, x, noise, embedding, **kwargs)


class XDiffusion_x(nn.Module):
    """diffusion.py:706-767: picks the diffusion class by alias ('k' -> KDiffusion_mod) and exposes
    sample()/inpaint() with sampler and schedule objects."""

    def __init__(self, type: str, net: nn.Module, **kwargs):
        super().__init__()
        if type != "k":
            raise NotImplementedError(f"type='{type}': only the 'k' diffusion used by QMDiffusion* is built")
        self.net = net
        self.diffusion = KDiffusion_mod(net=net, **kwargs)

    def forward(self, *args, **kwargs):
        return self.diffusion(*args, **kwargs)

    def sample(self, noise, num_steps: int, sigma_schedule, sampler, clamp: bool, **kwargs) -> Tensor:
        """diffusion.py:724-741: builds a DiffusionSampler and calls it.  ``noise``: the initial draw (B, C, L) as in the
        reference, a NoiseSource, or None (drawn like generative.py:853)."""
        diffusion_sampler = DiffusionSampler(diffusion=self.diffusion, sampler=sampler, sigma_schedule=sigma_schedule,
                                             num_steps=num_steps, clamp=clamp)
        return diffusion_sampler(noise, **kwargs)

    def inpaint(self, sigma_schedule, sampler, inpaint, in_paint_mask, num_steps: int, num_resamples: int,
                **kwargs) -> Tensor:
        """diffusion.py:744-767: builds a DiffusionInpainter and calls it."""
        inpainter = DiffusionInpainter(diffusion=self.diffusion, sampler=sampler, sigma_schedule=sigma_schedule,
                                       num_steps=num_steps, num_resamples=num_resamples)
        return inpainter(inpaint, in_paint_mask, **kwargs)


class _FusedLoop:
    """What ADPM2Sampler.forward / inpaint call when the denoiser belongs to a QMDiffusion* model: the whole loop on
    the fused path (run_adpm2 / run_adpm2_inpaint) instead of one callback per evaluation."""

    def __init__(self, owner, kwargs: dict, extra: dict):
        unknown = set(kwargs) - {"embedding", "embedding_scale"}
        if unknown or "embedding" not in kwargs:
            raise TypeError(f"denoise_fn takes embedding= and embedding_scale= (got {sorted(kwargs)})")
        self.owner, self.kw, self.extra = owner, kwargs, extra

    def sample(self, noise, sampler, sigmas, num_steps):
        o, emb = self.owner, self.kw["embedding"]
        if emb.shape[0] == 0:                       # nothing to generate (the reference returns an empty tensor too)
            return torch.empty(0, o.pred_dim, o.max_length, device=emb.device)
        guided = self.kw.get("embedding_scale", 1.0) != 1.0
        eng = o.engine(emb.device, emb.shape[1], emb.shape[0] * (2 if guided else 1))
        ns = o._noise_source(noise, emb.shape[0], emb.device)
        x = self.extra
        sigma_data = o.diffusion.diffusion.sigma_data
        scale = self.kw.get("embedding_scale", 1.0)
        if ns.steps is None and x.get("trace") is None and x.get("timer") is None and type(sampler) is ADPM2Sampler:
            # the plain call (counter-based step noise, nothing to record): the whole loop as ONE custom op
            tok = x.get("tokens")
            init = None if ns.init is None else ns.init.to(device=emb.device, dtype=torch.float32)
            out, t = torch.ops.mdt.sample(emb, init, None, torch.as_tensor(sigmas, dtype=torch.float32).cpu(),
                                          ops.register_engine(eng), o.pred_dim, float(sampler.rho), float(sigma_data), float(scale),
                                          bool(x.get("clamp", False)), int(ns.seed or 0), int(ns.sample0), tok is not None,
                                          float(o.diffusion.diffusion.dynamic_threshold))
            if tok is not None:
                tok.copy_(t)
            return out
        return run_adpm2(eng, emb, o.pred_dim, num_steps, ns, sigmas, sampler, sigma_data, scale, bool(x.get("clamp", False)),
                         x.get("trace"), x.get("timer"), x.get("tokens"), float(o.diffusion.diffusion.dynamic_threshold))

    def inpaint(self, source, mask, sampler, sigmas, num_steps, num_resamples):
        o, emb = self.owner, self.kw["embedding"]
        guided = self.kw.get("embedding_scale", 1.0) != 1.0
        eng = o.engine(emb.device, emb.shape[1], emb.shape[0] * (2 if guided else 1))
        draw, seed = self.extra.get("draw"), self.extra.get("seed")
        if draw is None and seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        return run_adpm2_inpaint(eng, emb, source, mask, num_steps, num_resamples, draw, seed, sigmas, sampler,
                                 o.diffusion.diffusion.sigma_data, self.kw.get("embedding_scale", 1.0),
                                 dynamic_threshold=float(o.diffusion.diffusion.dynamic_threshold))


class _QMBase(nn.Module):
    _inverse = True

    def _init_common(self, max_length, channels, pred_dim, unet, context_embedding_max_length, unet_type,
                     pos_emb_fourier, pos_emb_fourier_add, text_embed_dim, embed_dim_position):
        self.unet_type = unet_type
        self.fc1 = nn.Linear(1, text_embed_dim)
        self.GELUact = nn.GELU()
        self.pos_emb_fourier = pos_emb_fourier
        self.pos_emb_fourier_add = pos_emb_fourier_add
        self._text_dim = text_embed_dim
        self._pos_dim = 0
        self._pos_add = bool(pos_emb_fourier and pos_emb_fourier_add)
        if pos_emb_fourier:
            if pos_emb_fourier_add:
                # x + p_enc_1d(x): the encoding returns its first text_embed_dim columns (transformer.py:3470), so any
                # text_embed_dim <= embed_dim_position works as in the reference; a larger one does not broadcast there either
                if text_embed_dim > embed_dim_position:
                    raise RuntimeError("pos_emb_fourier_add=True needs text_embed_dim <= embed_dim_position "
                                       "(x + p_enc_1d(x) does not broadcast otherwise, generative.py:846)")
            else:
                text_embed_dim = text_embed_dim + embed_dim_position
            self._pos_dim = embed_dim_position
            self.p_enc_1d = PositionalEncoding1D(embed_dim_position)
        self.max_length = max_length
        self.pred_dim = pred_dim
        if unet_type != "cfg":
            raise NotImplementedError("only unet_type='cfg' (UNetCFG1d) is on the MI355X sampling path")
        if unet is not None:
            if not isinstance(unet, UNetCFG1d):
                raise TypeError("unet must be a moleculediffusiontransformer_amd UNetCFG1d")
            self.unet = unet
        else:
            self.unet = UNetCFG1d(self._unet_config(pred_dim, channels, text_embed_dim, context_embedding_max_length))
        self.diffusion = XDiffusion_x(type="k", net=self.unet,
                                      sigma_distribution=LogNormalDistribution(mean=-1.2, std=1.2),
                                      sigma_data=0.1, dynamic_threshold=0.0)
        object.__setattr__(self.diffusion.diffusion, "_owner", self)
        self.unet._evaluator = self._unet_call
        # 'bf16x3': split-bf16 MFMA GEMMs (fp32-class accuracy, ~5x the fp32-MFMA rate); 'f32': exact fp32 MFMA
        self.gemm_mode = os.environ.get("MDT_GEMM", "bf16x3")
        # form of the 256-channel transformers: 'auto' (by the batch of each call), 'wide', 'narrow' (see _wide)
        self.kernel_choice = {"1": "wide", "0": "narrow"}.get(os.environ.get("MDT_TF256", "auto"), "auto")
        # A pair hand-off that times out (a partner workgroup was not resident: compute units held by another stream / process)
        # leaves garbage rows and a status word on the device.  By default every sampling call and every net() / denoise_fn()
        # evaluation waits for that word before it returns (one event wait per call) and raises RuntimeError.  A pipelined caller
        # that issues calls back to back may set this True (or MDT_DEFER_HANDOFF=1): the word is then looked at by the NEXT call,
        # or by engine.handoff_check(wait=True) when the caller synchronises anyway.
        self.defer_handoff_check = os.environ.get("MDT_DEFER_HANDOFF", "0") == "1"
        self._engine: Optional[UNetEngine] = None
        self._engines = {}               # wide (bool) -> UNetEngine, for the current parameter values
        self._engine_key = None
        self.sampler_stats = {}

    def _unet_config(self, pred_dim, channels, ctx_features, ctx_max_length):
        mk = inverse_unet_config if self._inverse else forward_unet_config
        return mk(pred_dim, channels, ctx_features, ctx_max_length)

    # ------------------------------------------------------------------ engine management
    def _param_key(self, device, n_ctx):
        # every call asks whether the compiled engine still belongs to the parameter VALUES (an optimiser step or load_state_dict
        # bumps _version, .to() moves the storage).  The Parameter objects themselves are fixed once the module is built, so the
        # walk over the module tree (766 parameters in ~400 modules: milliseconds, visible to net() / denoise_fn seam users at
        # small batches) happens once; per call only data_ptr / _version of the cached list are read.
        cached = self.__dict__.get("_plist")
        if cached is None or cached[0] is not self.unet:                 # (a caller may assign another U-Net module)
            cached = self.__dict__["_plist"] = (self.unet, list(self.unet.parameters()))
        ps = cached[1]
        return (str(device), n_ctx, self.gemm_mode, tuple((p.data_ptr(), p._version) for p in ps))

    def _wide(self, batch: Optional[int]) -> bool:
        """Which form of the 256-channel transformers to run (compiler.py: MDT_TF256).  With the heads split over workgroup PAIRS
        (k_tf256 NSPLIT = 2) a batch of up to 4096 rows at that level (1024 samples of 4 tokens) fills the chip;
        above, the whole-transformer launch without the split (measured against the split forms: +10 % at 2048, +7 % at 4096
        samples).  The two forms add the heads' partial sums in different orders, so they agree to
        rounding (1e-6 class), not bit for bit: ``kernel_choice`` = 'wide' / 'narrow' pins one of them for every batch size
        (what a sharded run does, distributed.sample_sharded / pin_kernel_choice); 'auto' decides per call."""
        if self.kernel_choice in ("wide", "narrow"):
            return self.kernel_choice == "wide"
        cfg = self.unet.config
        t256 = None
        length = self.max_length // cfg.patch_size
        for lvl in range(cfg.num_layers + 1):
            if cfg.level_channels(lvl) == 256:
                t256 = length
            if lvl < cfg.num_layers:
                length //= cfg.factors[lvl]
        # pair-split launches need both workgroups of a pair running at the same time: 2 workgroups per 32 rows, and the device
        # keeps rt.pair_capacity() of them resident at once (compute units x occupancy, asked from the device: 256 on an MI355X,
        # i.e. 4096 rows).  The narrow form is used while ONE launch holds the level; above, the whole-transformer form is faster
        # (a pinned 'narrow' stays correct at any batch: launch_tf256 splits it into launches that fit, csrc/k_tf256.hip)
        if not batch or t256 is None:
            return False
        return 2 * ((batch * t256 + 31) // 32) > rt.pair_capacity()

    def pin_kernel_choice(self, batch: Optional[int]) -> str:
        """Resolve 'auto' for a batch of ``batch`` U-Net rows (samples, doubled under guidance) and keep that choice for every
        later call, whatever its batch: per-sample results then do not depend on how a batch is split over calls or ranks.
        ``batch`` None releases the pin.  Returns the choice now in force."""
        if batch is None:
            self.kernel_choice = "auto"
        else:
            self.kernel_choice = "auto"
            self.kernel_choice = "wide" if self._wide(batch) else "narrow"
        return self.kernel_choice

    def engine(self, device, n_ctx: Optional[int] = None, batch: Optional[int] = None) -> UNetEngine:
        """Compiled-program engine for the current parameter values on `device` (rebuilt after an optimiser step or
        load_state_dict), `n_ctx` conditioning tokens and the kernel choice that fits `batch` U-Net rows (see _wide)."""
        device = torch.device(device)
        _require_gpu(device)
        n_ctx = self.unet.config.ctx_max_length if n_ctx is None else n_ctx
        if n_ctx > self.unet.config.ctx_max_length:
            raise AssertionError("Input sequence length must be <= max_length")   # FixedEmbedding, modules.py:1194
        wide = self._wide(batch)
        key = self._param_key(device, n_ctx)
        if self._engine_key != key:                  # parameters changed: every cached engine is stale
            self._engines = {}
            self._engine_key = key
        if wide not in self._engines:
            sd = {k: v.detach().float().cpu() for k, v in self.unet.state_dict().items()}
            compiled = compile_unet(self.unet.config, self.max_length, n_ctx, sd, gemm_mode=self.gemm_mode, tf256=wide)
            self._engines[wide] = UNetEngine(compiled, device)
        self._engine = self._engines[wide]
        self._engine.sync_handoff_check = not self.defer_handoff_check
        return self._engine

    # ------------------------------------------------------------------ conditioning prelude
    def _embed(self, sequences: Tensor, device) -> Tensor:
        """generative.py:838-850 / :149-161 on the device (torch.ops.mdt.cond_embed -> mdt_cond_embed)."""
        device = torch.device(device)
        _require_gpu(device)
        seq = sequences.detach().float().to(device)
        inv = self.p_enc_1d.inv_freq.detach().float().to(device) if self._pos_dim else None
        return torch.ops.mdt.cond_embed(seq, self.fc1.weight.detach().to(device), self.fc1.bias.detach().to(device), inv,
                                        self._pos_dim, self._pos_add)

    # ------------------------------------------------------------------ seams used by the wrappers above
    def _noise_source(self, noise, B, device, sample0=0):
        if isinstance(noise, NoiseSource):
            return noise
        # default: initial noise from the CPU global generator exactly as generative.py:853 does; the
        # per-step draws come from the counter-based device generator seeded from the same CPU generator
        init = torch.randn(B, self.pred_dim, self.max_length) if noise is None else noise
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        ns = NoiseSource(seed=seed, sample0=sample0)
        ns.init = init
        return ns

    def _fused_adapter(self, kwargs: dict, extra: dict) -> _FusedLoop:
        return _FusedLoop(self, kwargs, extra)

    def _unet_call(self, x: Tensor, time, embedding: Tensor, embedding_scale: float = 1.0) -> Tensor:
        """net(x, time, embedding=..., embedding_scale=...) (modules.py:1228-1255); x is (B, C, L) as in the reference,
        ``time`` a scalar or one value per row (rows sharing a time value are evaluated together)."""
        t = torch.as_tensor(time, dtype=torch.float32).flatten().cpu()
        B = x.shape[0]
        if t.numel() not in (1, B):
            raise ValueError(f"time must be a scalar or hold one value per sample (got {t.numel()} for batch {B})")
        if t.numel() > 1 and not bool((t == t[0]).all()):
            out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
            for v in torch.unique(t):
                rows = (t == v).nonzero().flatten().to(x.device)
                out[rows] = self._unet_call(x[rows], v, embedding[rows], embedding_scale)
            return out
        device = x.device
        _require_gpu(device)
        eng = self.engine(device, embedding.shape[1], x.shape[0])
        C = x.shape[1]
        with torch.no_grad():
            xin = torch.ops.mdt.precond_in(x, 1.0, eng.c.in_pad)
            pred = torch.ops.mdt.unet_eval(xin, embedding, float(t[0]), float(embedding_scale), ops.register_engine(eng))
            return pred[:, :, :C].transpose(1, 2).contiguous()

    def _denoise(self, x_noisy: Tensor, sigma, embedding: Tensor, embedding_scale: float = 1.0) -> Tensor:
        """KDiffusion_mod.denoise_fn for a scalar sigma (diffusion.py:798-814) as three ops: input scaling, the network,
        output mix + clip."""
        device = x_noisy.device
        _require_gpu(device)
        eng = self.engine(device, embedding.shape[1], x_noisy.shape[0])
        w = scale_weights(torch.as_tensor(sigma, dtype=torch.float32).cpu(), self.diffusion.diffusion.sigma_data)
        with torch.no_grad():
            xin = torch.ops.mdt.precond_in(x_noisy, w.c_in, eng.c.in_pad)
            pred = torch.ops.mdt.unet_eval(xin, embedding, w.c_noise, float(embedding_scale), ops.register_engine(eng))
            return torch.ops.mdt.precond_out(x_noisy, pred, w.c_skip, w.c_out, float(self.diffusion.diffusion.dynamic_threshold))

    # ------------------------------------------------------------------ public API (reference signatures)
    def forward(self, sequences, output):
        """Training loss (generative.py:812-833 / :120-143): conditioning prelude + KDiffusion_mod.forward, plain PyTorch
        with autograd on whatever device the parameters live on (train.py).  Not on the MI355X sampling path."""
        from .train import conditioning_embedding
        x = conditioning_embedding(self, sequences)
        return self.diffusion(output, embedding=x)

    def _do_sample(self, sequences, device, cond_scale, timesteps, clamp, noise=None, trace=None, timer=None, tokens=None):
        emb = self._embed(sequences, device)
        return self.diffusion.sample(num_steps=timesteps, sampler=ADPM2Sampler(rho=1),
                                     sigma_schedule=KarrasSchedule(sigma_min=0.001, sigma_max=9.0, rho=3.0),
                                     clamp=clamp, noise=noise, embedding=emb, embedding_scale=cond_scale,
                                     trace=trace, timer=timer, tokens=tokens)

    def sample_tokens(self, sequences, device, cond_scale=None, timesteps=100, clamp=False, *, noise=None,
                      return_sample: bool = False):
        """sample() followed by the decode step of the reference's callers (sample_loop_generative / generate_from_conditioning,
        generative.py:1212-1213, :1690-1691: ``permute(0, 2, 1)`` then ``argmax(dim=2)``), with the argmax taken inside the
        last sampler update: returns (B, max_length) int64 token ids on ``device`` (and the fp32 sample if asked)."""
        if cond_scale is None:
            cond_scale = 7.5 if self._inverse else 1.0
        B = sequences.shape[0]
        tok = torch.zeros(B, self.max_length, dtype=torch.int32, device=device)
        x = self._do_sample(sequences, device, cond_scale, timesteps, clamp, noise, tokens=tok if B else None)
        tok = tok.long()
        return (tok, x) if return_sample else tok

    def inpaint(self, sequences, device, cond_scale=7.5, timesteps=100, num_resamples=1, inpaint=None,
                in_paint_mask=None, *, draw=None, seed=None):
        emb = self._embed(sequences, device)
        return self.diffusion.inpaint(num_steps=timesteps, num_resamples=num_resamples, sampler=ADPM2Sampler(rho=1),
                                      sigma_schedule=KarrasSchedule(sigma_min=0.001, sigma_max=9.0, rho=3.0),
                                      inpaint=inpaint, in_paint_mask=in_paint_mask, embedding=emb,
                                      embedding_scale=cond_scale, draw=draw, seed=seed)


class QMDiffusion(_QMBase):
    """Generative inverse diffusion model (generative.py:718-914)."""

    _inverse = True

    def __init__(self, max_length=1024, channels=128, pred_dim=1, context_embedding_max_length=32, unet_type="cfg",
                 pos_emb_fourier=True, pos_emb_fourier_add=False, text_embed_dim=1024, embed_dim_position=64,
                 unet=None):
        super().__init__()
        print("Using unet type: ", unet_type)
        self._init_common(max_length, channels, pred_dim, unet, context_embedding_max_length, unet_type,
                          pos_emb_fourier, pos_emb_fourier_add, text_embed_dim, embed_dim_position)

    def sample(self, sequences, device, cond_scale=7.5, timesteps=100, clamp=False, *, noise=None, trace=None,
               timer=None):
        return self._do_sample(sequences, device, cond_scale, timesteps, clamp, noise, trace, timer)


class QMDiffusionForward(_QMBase):
    """Forward diffusion property predictor (generative.py:31-225); note `unet` is the 4th positional."""

    _inverse = False

    def __init__(self, max_length=1024, channels=128, pred_dim=1, unet=None, context_embedding_max_length=32,
                 unet_type="cfg", pos_emb_fourier=True, pos_emb_fourier_add=False, text_embed_dim=1024,
                 embed_dim_position=64):
        super().__init__()
        self._init_common(max_length, channels, pred_dim, unet, context_embedding_max_length, unet_type,
                          pos_emb_fourier, pos_emb_fourier_add, text_embed_dim, embed_dim_position)

    def sample(self, sequences, device, cond_scale=1.0, timesteps=100, clamp=False, *, noise=None, trace=None,
               timer=None):
        return self._do_sample(sequences, device, cond_scale, timesteps, clamp, noise, trace, timer)


# ----------------------------------------------------------------------------------------------------------------------
# the inverse -> forward validation chain of the reference's callers, kept on the device
# ----------------------------------------------------------------------------------------------------------------------
def tokens_to_forward_input(tokens: Tensor, max_length: int, X_norm_factor: float = 1.0) -> Tensor:
    """What the reference does between the two models with strings (generative.py:1229 reverse_tokenize ->
    predict_properties_from_SMILES :425-429 texts_to_sequences + pad_sequences(maxlen, padding='post',
    truncating='post') / X_norm_factor), restated on token ids: id 0 is "no character" (keras' sequences_to_texts skips
    it), so every row is compacted to its non-zero ids in order, truncated / zero-padded at the end to ``max_length`` and
    scaled.  Assumes every id 1..pred_dim-1 is in the tokenizer's vocabulary (as produced by the one-hot training data).
    Pinned bit for bit by tests/golden/token_chain.npz (the reference's own two functions over a restated keras tokenizer)."""
    tok = tokens.long()
    B, L = tok.shape
    keep = tok != 0
    order = torch.argsort((~keep).to(torch.int8), dim=1, stable=True)          # non-zero ids first, original order kept
    packed = torch.gather(tok * keep, 1, order)
    out = torch.zeros(B, max_length, dtype=torch.float64, device=tok.device)
    n = min(L, max_length)
    out[:, :n] = packed[:, :n].double()
    # the reference divides the int32 array by X_norm_factor in float64 (numpy) and torch.Tensor() rounds to fp32 (generative.py:428-429)
    return (out / X_norm_factor).float()


def predict_properties_from_tokens(model_forward: "QMDiffusionForward", tokens: Tensor, device, cond_scale: float = 1.0,
                                   timesteps: int = 100, clamp: bool = False, X_norm_factor: float = 1.0,
                                   context_embedding_max_length: int = 12, noise=None) -> Tensor:
    """predict_properties_from_SMILES (generative.py:404-451) for molecules given as token ids on the device: the forward
    model's sample() on the re-tokenised ids, first ``context_embedding_max_length`` positions = the (scaled) properties.
    Returns (B, context_embedding_max_length) on ``device`` (the caller applies scaler.inverse_transform)."""
    data = tokens_to_forward_input(tokens.to(device), model_forward.max_length, X_norm_factor)
    result = model_forward.sample(data, device, cond_scale=cond_scale, timesteps=timesteps, clamp=clamp, noise=noise)
    return result[:, 0, :context_embedding_max_length]


def generate_and_validate(model: "QMDiffusion", model_forward: "QMDiffusionForward", conditioning: Tensor, device,
                          cond_scale: float = 1.0, timesteps: int = 100, forward_timesteps: int = 100,
                          X_norm_factor: float = 1.0, noise=None, forward_noise=None):
    """generate_from_conditioning's core (generative.py:1685-1713) without leaving the GPU: sample molecules for the
    conditioning, decode them inside the last sampler update, re-predict their properties with the forward model.
    Returns (tokens (B, L) int64, predicted properties (B, n_cond))."""
    tokens = model.sample_tokens(conditioning, device, cond_scale=cond_scale, timesteps=timesteps, noise=noise)
    props = predict_properties_from_tokens(model_forward, tokens, device, cond_scale=1.0, timesteps=forward_timesteps,
                                           X_norm_factor=X_norm_factor,
                                           context_embedding_max_length=conditioning.shape[1], noise=forward_noise)
    return tokens, props
