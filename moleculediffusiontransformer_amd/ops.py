"""`torch.ops.mdt.*`: the path's native entry points as PyTorch custom ops (last row).

Each op is a thin body over the C ABI of libmdt_hip.so (include/mdt_hip.h): tensors in, tensors out, launched on the
CURRENT HIP stream of the tensors' device, no hidden state besides the packed-weight handle a model owns (`handle`, an
integer from register_engine()).  Errors follow the TORCH_CHECK convention: every failure is a Python RuntimeError
(non-HIP tensors included -- there is no CPU implementation behind these ops; only shape inference is registered for
fake / meta tensors).

    cond_embed          generative.py:838-850 (fc1 -> GELU, PositionalEncoding1D, cat)
    precond_in / _out   KDiffusion_mod.denoise_fn scaling + clip (diffusion.py:798-814)
    cfg_mix             UNetCFG1d.forward guidance mix (modules.py:1253)
    adpm2_mid / _next   the two halves of ADPM2Sampler.step (diffusion.py:502-515)
    adpm2_euler         one Euler move of the step for a caller-supplied denoiser
    argmax_tokens       decode step after the path (generative.py:1212-1213)
    unet_eval           UNetCFG1d.forward: net(x, time, embedding=, embedding_scale=) (modules.py:1228-1255)
    sample              DiffusionSampler.forward + ADPM2Sampler.forward, the whole loop (diffusion.py:577-591, :517-524)
    all_gather_samples  the one collective of a sharded call (RCCL all_gather_into_tensor)
"""
from __future__ import annotations

import weakref
from typing import Dict, Optional, Tuple

import torch
from torch.library import custom_op

from . import runtime as rt

Tensor = torch.Tensor

_ENGINES: "weakref.WeakValueDictionary[int, object]" = weakref.WeakValueDictionary()
_NEXT = [1]


def register_engine(engine) -> int:
    """Integer handle of a UNetEngine (packed weights + programs) for the ops that evaluate the network."""
    h = getattr(engine, "_op_handle", None)
    if h is None:
        h = _NEXT[0]
        _NEXT[0] += 1
        engine._op_handle = h
    _ENGINES[h] = engine
    return h


def _engine(handle: int):
    e = _ENGINES.get(handle)
    if e is None:
        raise RuntimeError(f"mdt: unknown or released engine handle {handle}")
    return e


def _hip(*tensors: Optional[Tensor]) -> torch.device:
    """Device guard: every tensor on ONE HIP device, fp32 unless stated; returns the device."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if t.device.type != "cuda":
            raise RuntimeError(f"mdt ops run on an AMD GPU through libmdt_hip.so; got a tensor on '{t.device}' "
                               "(there is no CPU implementation)")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"mdt: tensors on different devices ({dev} and {t.device})")
    if dev is None:
        raise RuntimeError("mdt: no tensor argument")
    return dev


def _f32c(t: Tensor) -> Tensor:
    return t.detach().to(torch.float32).contiguous()


# ----------------------------------------------------------------------------------------------------------------------
@custom_op("mdt::cond_embed", mutates_args=())
def cond_embed(seq: Tensor, fc1_w: Tensor, fc1_b: Tensor, inv_freq: Optional[Tensor], pos_dim: int, pos_add: bool = False) -> Tensor:
    """pos_add: the positional encoding is added to the fc1 features (pos_emb_fourier_add) instead of concatenated."""
    dev = _hip(seq, fc1_w, fc1_b, inv_freq)
    lib = rt.load_library()
    seq, w, b = _f32c(seq), _f32c(fc1_w).view(-1), _f32c(fc1_b)
    B, n = seq.shape
    D1 = b.numel()
    if pos_dim and (inv_freq is None or inv_freq.numel() * 2 != pos_dim):
        raise RuntimeError("mdt::cond_embed: inv_freq must hold pos_dim / 2 frequencies")
    if pos_add and D1 > pos_dim:
        raise RuntimeError("mdt::cond_embed: the additive form needs text_embed_dim <= embed_dim_position (the encoding has "
                           "embed_dim_position columns, of which the first text_embed_dim are added)")
    inv = _f32c(inv_freq) if pos_dim else w
    out = torch.empty(B, n, D1 if pos_add else D1 + pos_dim, device=dev)
    if B:
        with torch.cuda.device(dev):
            if pos_add:
                rt.check(lib.mdt_cond_embed_add(rt.ptr(seq), rt.ptr(w), rt.ptr(b), rt.ptr(inv), rt.ptr(out), B, n, D1, pos_dim,
                                                rt.current_stream()))
            else:
                rt.check(lib.mdt_cond_embed(rt.ptr(seq), rt.ptr(w), rt.ptr(b), rt.ptr(inv), rt.ptr(out), B, n, D1, pos_dim,
                                            rt.current_stream()))
    return out


@cond_embed.register_fake
def _(seq, fc1_w, fc1_b, inv_freq, pos_dim, pos_add=False):
    return seq.new_empty(seq.shape[0], seq.shape[1], fc1_b.numel() + (0 if pos_add else pos_dim), dtype=torch.float32)


@custom_op("mdt::precond_in", mutates_args=())
def precond_in(x: Tensor, c_in: float, Cp: int) -> Tensor:
    dev = _hip(x)
    lib = rt.load_library()
    x = _f32c(x)
    B, C, L = x.shape
    if Cp < C or Cp % 16:
        raise RuntimeError(f"mdt::precond_in: Cp={Cp} must be a multiple of 16 and >= C={C}")
    out = torch.zeros(B, L, Cp, device=dev)
    if B:
        with torch.cuda.device(dev):
            rt.check(lib.mdt_precond_in(rt.ptr(x), rt.ptr(out), float(c_in), B, C, L, Cp, rt.current_stream()))
    return out


@precond_in.register_fake
def _(x, c_in, Cp):
    return x.new_empty(x.shape[0], x.shape[2], Cp, dtype=torch.float32)


def _dyn_scale(lib, x: Tensor, pred: Tensor, c_skip: float, c_out: float, q: float) -> Optional[Tensor]:
    """clip()'s per-sample dynamic threshold (diffusion.py:78-85) for the denoise stage of the next kernel, or None for q == 0."""
    if not q:
        return None
    B, C, L = x.shape
    scale = torch.empty(B, device=x.device)
    rt.check(lib.mdt_dyn_scale(rt.ptr(x), rt.ptr(pred), rt.ptr(scale), float(c_skip), float(c_out), float(q), B, C, L,
                               pred.shape[2], rt.current_stream()))
    return scale


@custom_op("mdt::precond_out", mutates_args=())
def precond_out(x: Tensor, pred: Tensor, c_skip: float, c_out: float, dynamic_threshold: float = 0.0) -> Tensor:
    """D = clip(c_skip x + c_out pred, dynamic_threshold) (diffusion.py:811-814, :75-88)."""
    dev = _hip(x, pred)
    lib = rt.load_library()
    x, pred = _f32c(x), _f32c(pred)
    B, C, L = x.shape
    if pred.dim() != 3 or pred.shape[0] != B or pred.shape[1] != L or pred.shape[2] < C:
        raise RuntimeError(f"mdt::precond_out: pred {tuple(pred.shape)} is not token-major (B, L, Cp) for x {tuple(x.shape)}")
    out = torch.empty_like(x)
    if B:
        with torch.cuda.device(dev):
            ds = _dyn_scale(lib, x, pred, c_skip, c_out, dynamic_threshold)
            rt.check(lib.mdt_precond_out(rt.ptr(x), rt.ptr(pred), rt.ptr(out), float(c_skip), float(c_out), B, C, L,
                                         pred.shape[2], rt.ptr(ds), rt.current_stream()))
    return out


@precond_out.register_fake
def _(x, pred, c_skip, c_out, dynamic_threshold=0.0):
    return x.new_empty(x.shape, dtype=torch.float32)


@custom_op("mdt::cfg_mix", mutates_args=())
def cfg_mix(cond: Tensor, uncond: Tensor, scale: float) -> Tensor:
    dev = _hip(cond, uncond)
    lib = rt.load_library()
    cond, uncond = _f32c(cond), _f32c(uncond)
    if cond.shape != uncond.shape:
        raise RuntimeError("mdt::cfg_mix: shape mismatch")
    out = torch.empty_like(cond)
    if cond.numel():
        with torch.cuda.device(dev):
            rt.check(lib.mdt_cfg_mix(rt.ptr(cond), rt.ptr(uncond), rt.ptr(out), float(scale), cond.numel(), rt.current_stream()))
    return out


@cfg_mix.register_fake
def _(cond, uncond, scale):
    return cond.new_empty(cond.shape, dtype=torch.float32)


@custom_op("mdt::adpm2_mid", mutates_args=())
def adpm2_mid(x: Tensor, pred: Tensor, c_skip: float, c_out: float, sigma: float, dt_mid: float,
              c_in_mid: float, dynamic_threshold: float = 0.0) -> Tuple[Tensor, Tensor]:
    dev = _hip(x, pred)
    lib = rt.load_library()
    x, pred = _f32c(x), _f32c(pred)
    B, C, L = x.shape
    Cp = pred.shape[2]
    x_mid = torch.empty_like(x)
    xin = torch.zeros(B, L, Cp, device=dev)
    if B:
        with torch.cuda.device(dev):
            ds = _dyn_scale(lib, x, pred, c_skip, c_out, dynamic_threshold)
            rt.check(lib.mdt_adpm2_mid(rt.ptr(x), rt.ptr(pred), rt.ptr(x_mid), rt.ptr(xin), float(c_skip), float(c_out),
                                       float(sigma), float(dt_mid), float(c_in_mid), B, C, L, Cp, rt.ptr(ds), rt.current_stream()))
    return x_mid, xin


@adpm2_mid.register_fake
def _(x, pred, c_skip, c_out, sigma, dt_mid, c_in_mid, dynamic_threshold=0.0):
    return x.new_empty(x.shape, dtype=torch.float32), pred.new_empty(pred.shape, dtype=torch.float32)


@custom_op("mdt::adpm2_next", mutates_args=())
def adpm2_next(x: Tensor, x_mid: Tensor, pred: Tensor, noise: Optional[Tensor], c_skip: float, c_out: float,
               sigma_mid: float, dt_down: float, sigma_up: float, c_in_next: float, seed: int, step: int,
               sample0: int, dynamic_threshold: float = 0.0) -> Tuple[Tensor, Tensor]:
    """Returns (x_next, xin_next); noise None = counter-based generator keyed by (seed, step, sample0 + b)."""
    dev = _hip(x, x_mid, pred, noise)
    lib = rt.load_library()
    xn, x_mid, pred = _f32c(x).clone(), _f32c(x_mid), _f32c(pred)
    nz = None if noise is None else _f32c(noise)
    B, C, L = xn.shape
    Cp = pred.shape[2]
    xin = torch.zeros(B, L, Cp, device=dev)
    if B:
        with torch.cuda.device(dev):
            ds = _dyn_scale(lib, x_mid, pred, c_skip, c_out, dynamic_threshold)
            rt.check(lib.mdt_adpm2_next(rt.ptr(xn), rt.ptr(x_mid), rt.ptr(pred), rt.ptr(nz), rt.ptr(xin), float(c_skip),
                                        float(c_out), float(sigma_mid), float(dt_down), float(sigma_up), float(c_in_next),
                                        int(seed), int(step), int(sample0), B, C, L, Cp, 0, rt.ptr(ds), rt.current_stream()))
    return xn, xin


@adpm2_next.register_fake
def _(x, x_mid, pred, noise, c_skip, c_out, sigma_mid, dt_down, sigma_up, c_in_next, seed, step, sample0, dynamic_threshold=0.0):
    return x.new_empty(x.shape, dtype=torch.float32), pred.new_empty(pred.shape, dtype=torch.float32)


@custom_op("mdt::adpm2_euler", mutates_args=())
def adpm2_euler(x_base: Tensor, x_from: Tensor, denoised: Tensor, noise: Optional[Tensor], sigma: float, dt: float,
                sigma_up: float) -> Tensor:
    dev = _hip(x_base, x_from, denoised, noise)
    lib = rt.load_library()
    xb, xf, dn = _f32c(x_base), _f32c(x_from), _f32c(denoised)
    nz = None if noise is None else _f32c(noise)
    B, C, L = xb.shape
    out = torch.empty_like(xb)
    if B:
        with torch.cuda.device(dev):
            rt.check(lib.mdt_adpm2_euler(rt.ptr(xb), rt.ptr(xf), rt.ptr(dn), rt.ptr(nz), rt.ptr(out), float(sigma), float(dt),
                                         float(sigma_up), 0 if nz is None else 1, 0, 0, 0, B, C, L, rt.current_stream()))
    return out


@adpm2_euler.register_fake
def _(x_base, x_from, denoised, noise, sigma, dt, sigma_up):
    return x_base.new_empty(x_base.shape, dtype=torch.float32)


@custom_op("mdt::argmax_tokens", mutates_args=())
def argmax_tokens(x: Tensor) -> Tensor:
    dev = _hip(x)
    lib = rt.load_library()
    x = _f32c(x)
    B, C, L = x.shape
    tok = torch.zeros(B, L, dtype=torch.int32, device=dev)
    if B:
        with torch.cuda.device(dev):
            rt.check(lib.mdt_argmax_tokens(rt.ptr(x), rt.ptr(tok), B, C, L, rt.current_stream()))
    return tok


@argmax_tokens.register_fake
def _(x):
    return x.new_empty(x.shape[0], x.shape[2], dtype=torch.int32)


# ----------------------------------------------------------------------------------------------------------------------
# the network and the whole loop: `handle` names the model's compiled engine (packed weights, programs, HIP graphs)
# ----------------------------------------------------------------------------------------------------------------------
@custom_op("mdt::unet_eval", mutates_args=())
def unet_eval(xin: Tensor, embedding: Tensor, c_noise: float, embedding_scale: float, handle: int) -> Tensor:
    """xin (B, L, Cp) token-major network input, embedding (B, n, F); returns the prediction (B, L, Cp)."""
    dev = _hip(xin, embedding)
    lib = rt.load_library()
    eng = _engine(handle)
    if eng.device != dev:
        raise RuntimeError(f"mdt::unet_eval: engine lives on {eng.device}, tensors on {dev}")
    xin = _f32c(xin)
    B = xin.shape[0]
    if tuple(xin.shape[1:]) != (eng.c.length, eng.c.in_pad):
        raise RuntimeError(f"mdt::unet_eval: xin {tuple(xin.shape)} is not (B, {eng.c.length}, {eng.c.in_pad})")
    if B == 0:
        return torch.empty_like(xin)
    with torch.no_grad(), torch.cuda.device(dev):
        eng.reserve(B)
        eng.prepare_context(embedding)
        eng.prepare_times(torch.tensor([float(c_noise)]))
        eng.select_time(0)
        eng.xin.copy_(xin)
        eng.handoff_check()
        pred = eng.eval(False)
        if embedding_scale != 1.0:
            um = eng.eval(True)
            rt.check(lib.mdt_cfg_mix(rt.ptr(pred), rt.ptr(um), rt.ptr(pred), float(embedding_scale), pred.numel(),
                                     rt.current_stream()))
        out = pred.clone()
        eng.note_handoff()          # pair hand-off status: checked before the prediction is returned (engine.py)
        return out


@unet_eval.register_fake
def _(xin, embedding, c_noise, embedding_scale, handle):
    return xin.new_empty(xin.shape, dtype=torch.float32)


@custom_op("mdt::sample", mutates_args=())
def sample(embedding: Tensor, init_noise: Optional[Tensor], step_noise: Optional[Tensor], sigmas: Tensor, handle: int,
           pred_dim: int, rho: float, sigma_data: float, embedding_scale: float, clamp: bool, seed: int, sample0: int,
           want_tokens: bool, dynamic_threshold: float = 0.0) -> Tuple[Tensor, Tensor]:
    """The whole ADPM2 loop for an evaluated sigma schedule (num_steps + 1 values).  init_noise (B, C, L) / step_noise
    (num_steps - 1, B, C, L): explicit draws in the reference's call order; each one that is None comes from the counter-based
    generator keyed by (seed, draw index, sample0 + b) instead.  Returns (x (B, C, L), tokens (B, L) int32 or an empty tensor)."""
    from .diffusion import ADPM2Sampler, NoiseSource, run_adpm2
    dev = _hip(embedding, step_noise)
    eng = _engine(handle)
    if eng.device != dev:
        raise RuntimeError(f"mdt::sample: engine lives on {eng.device}, tensors on {dev}")
    B = embedding.shape[0]
    num_steps = sigmas.numel() - 1
    ns = NoiseSource(seed=int(seed), sample0=int(sample0))
    if init_noise is not None:
        ns.init = init_noise
    if step_noise is not None:
        if step_noise.shape[0] != max(num_steps - 1, 0):
            raise RuntimeError(f"mdt::sample: step_noise holds {step_noise.shape[0]} draws, the loop makes {num_steps - 1}")
        ns.steps = lambda i: step_noise[i]
    tok = torch.zeros(B, eng.c.length, dtype=torch.int32, device=dev) if want_tokens else None
    x = run_adpm2(eng, embedding, pred_dim, num_steps, ns, sigmas, ADPM2Sampler(rho=rho), float(sigma_data),
                  float(embedding_scale), bool(clamp), None, None, tok if B else None, float(dynamic_threshold))
    return x, (tok if want_tokens else torch.empty(0, dtype=torch.int32, device=dev))


@sample.register_fake
def _(embedding, init_noise, step_noise, sigmas, handle, pred_dim, rho, sigma_data, embedding_scale, clamp, seed, sample0,
      want_tokens, dynamic_threshold=0.0):
    eng = _engine(handle)
    B = embedding.shape[0]
    return (embedding.new_empty(B, pred_dim, eng.c.length, dtype=torch.float32),
            embedding.new_empty((B, eng.c.length) if want_tokens else (0,), dtype=torch.int32))


@custom_op("mdt::all_gather_samples", mutates_args=())
def all_gather_samples(local: Tensor, total: int) -> Tensor:
    """(b_r, ...) per rank -> (total, ...) on every rank over the default process group (RCCL on HIP tensors)."""
    from .distributed import all_gather_samples as gather
    return gather(local, int(total), force_collective=True)


@all_gather_samples.register_fake
def _(local, total):
    return local.new_empty((total,) + tuple(local.shape[1:]))
