"""Deterministic synthetic weights, a closed-form function of (key name, element index).

There is no network for checkpoints, and the reference's default init depends on
construction order and RNG state, so parity tests, bench.py and the golden
vectors all use this fill: it is regenerated bit-identically anywhere (integer
hashing only, no library RNG) and loaded into the real reference with
``load_state_dict`` when the golden vectors are made (tests/golden/make_golden.py).
Scales follow PyTorch's default init (uniform +-1/sqrt(fan_in); norm gains near 1).
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np
import torch


def _hash_uniform(name: str, n: int) -> np.ndarray:
    """n values in [0, 1) on a 2^-24 grid (exact in fp32), splitmix64 over (crc32(name), index)."""
    seed = np.uint64(zlib.crc32(name.encode("utf-8")))
    with np.errstate(over="ignore"):
        z = (seed << np.uint64(32)) + np.arange(n, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(40)).astype(np.float64) / float(1 << 24)


def synth_tensor(name: str, shape: Tuple[int, ...]) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    u = 2.0 * _hash_uniform(name, n) - 1.0                       # [-1, 1)
    if name.endswith("inv_freq"):                                 # PositionalEncoding1D buffer
        ch = 2 * shape[0]
        v = 1.0 / (10000 ** (torch.arange(0, ch, 2).float() / ch))
        return v
    if name.endswith(".weights") or "fixed_embedding" in name:    # randn-initialised in the reference
        v = u * np.sqrt(3.0)
    elif len(shape) >= 2:                                         # Linear / Conv / ConvTranspose weight
        fan_in = int(np.prod(shape[1:]))
        v = u / np.sqrt(fan_in)
    elif name.endswith(".weight"):                                # GroupNorm / LayerNorm gain
        v = 1.0 + 0.1 * u
    else:                                                         # every bias
        v = 0.1 * u
    return torch.from_numpy(v.astype(np.float32)).reshape(shape)


def synth_state_dict(keys_shapes: Iterable[Tuple[str, Tuple[int, ...]]],
                     alias_prefixes: Tuple[str, ...] = ("unet.", "diffusion.net.", "diffusion.diffusion.net.")
                     ) -> Dict[str, torch.Tensor]:
    """Fill every (key, shape).  Keys under the three aliases of the U-Net (the
    reference's state_dict carries the same storage under unet.*, diffusion.net.* and
    diffusion.diffusion.net.*) are generated from their canonical ``unet.`` name so the aliases agree."""
    out: Dict[str, torch.Tensor] = {}
    for key, shape in keys_shapes:
        canon = key
        for ap in alias_prefixes[1:]:
            if key.startswith(ap):
                canon = alias_prefixes[0] + key[len(ap):]
        out[key] = synth_tensor(canon, tuple(shape))
    return out


def synth_normal(name: str, shape: Tuple[int, ...]) -> torch.Tensor:
    """Deterministic N(0,1) fp32 noise keyed by name (Box-Muller over the hashed uniforms).
    Stands for torch.randn / torch.randn_like in parity mode so that the oracle, the golden
    generator and the HIP path consume identical noise without depending on a library RNG."""
    n = int(np.prod(shape))
    u1 = _hash_uniform(name + "/u1", n)
    u2 = _hash_uniform(name + "/u2", n)
    r = np.sqrt(-2.0 * np.log(1.0 - u1))          # 1-u1 in (0, 1]
    z = r * np.cos(2.0 * np.pi * u2)
    return torch.from_numpy(z.astype(np.float32)).reshape(shape)


def synth_uniform(name: str, shape: Tuple[int, ...]) -> torch.Tensor:
    n = int(np.prod(shape))
    return torch.from_numpy(_hash_uniform(name, n).astype(np.float32)).reshape(shape)


# BASELINE.json configurations (and the small parity-test models) by name: (kind, wrapper keyword arguments)
MODEL_CASES = {
    "cfg1": ("inverse", dict(max_length=64, pred_dim=16, channels=64, context_embedding_max_length=12)),
    "cfg3": ("forward", dict(max_length=64, pred_dim=1, channels=64, context_embedding_max_length=64)),
    "tiny": ("inverse", dict(max_length=32, pred_dim=16, channels=16, context_embedding_max_length=12)),
    "pd22": ("inverse", dict(max_length=32, pred_dim=22, channels=32, context_embedding_max_length=12)),
    # BASELINE.json configs[4] architecture (deep U-Net): channels=256, pred_dim=32, max_len=128
    "cfg5": ("inverse", dict(max_length=128, pred_dim=32, channels=256, context_embedding_max_length=12)),
    # the inverse model as trained in the reference's notebook (Inverse_Diffusion.ipynb:1587-1604; 90,965,554 parameters)
    "nb": ("inverse", dict(max_length=32, pred_dim=22, channels=128, context_embedding_max_length=12)),
    # an AnalogDiffusionSparse-shaped U-Net (graphmodel.py:266-283: patch_size 8, attentions [1, 1], no pre-transformer)
    "sparse": ("sparse", dict(max_length=128, pred_dim=3, channels=128, context_embedding_max_length=12)),
    # AnalogDiffusionFull (graphmodel.py:391-446: patch_size 4, num_blocks [3, 3]) with pos_emb_fourier_add=True
    "full": ("full", dict(max_length=64, pred_dim=8, channels=64, context_embedding_max_length=12)),
}


def make_synth_model(case: str, device=None):
    """QMDiffusion / QMDiffusionForward of a named configuration with the deterministic synthetic weights loaded
    (text_embed_dim=64 + embed_dim_position=64 = the 128 context features every notebook of the reference uses)."""
    from .generative import QMDiffusion, QMDiffusionForward
    kind, kw = MODEL_CASES[case]
    if kind == "full":
        from .graphmodel import AnalogDiffusionFull
        m = AnalogDiffusionFull(text_embed_dim=64, embed_dim_position=64, pos_emb_fourier_add=True, **kw)
    elif kind == "sparse":
        from .modules import UNetCFG1d
        from .netspec import sparse_unet_config
        unet = UNetCFG1d(sparse_unet_config(kw["pred_dim"], kw["channels"], 128, kw["context_embedding_max_length"]))
        m = QMDiffusion(text_embed_dim=64, embed_dim_position=64, unet=unet, **kw)
    else:
        cls = QMDiffusion if kind == "inverse" else QMDiffusionForward
        m = cls(text_embed_dim=64, embed_dim_position=64, **kw)
    m.load_state_dict(synth_state_dict([(k, tuple(v.shape)) for k, v in m.state_dict().items()]))
    return m if device is None else m.to(device)
