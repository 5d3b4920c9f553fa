"""Parameter containers of the 1-D conditional U-Net with the reference's attribute/key layout.

The reference builds its network from nn.Module classes whose forward() runs ATen ops
(modules.py:934-1255).  Here the module tree only OWNS the parameters, under exactly the same
names, so state_dict()/load_state_dict()/parameters()/.to() behave as on the reference
(checkpoint row); evaluation is the compiled op program on libmdt_hip.so.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from .netspec import Manifest, UNetConfig, unet_manifest


class ParamNode(nn.Module):
    """A named container: children and parameters are attached by dotted key."""

    def extra_repr(self) -> str:
        n = sum(p.numel() for p in self.parameters(recurse=False))
        return f"own_params={n}" if n else ""


def _default_init(manifest: Manifest) -> Dict[str, torch.Tensor]:
    """PyTorch's default initialisers by parameter role (nn.Linear / nn.Conv1d / nn.ConvTranspose1d:
    kaiming_uniform(a=sqrt(5)) = U(+-1/sqrt(fan_in)); norms: ones/zeros; nn.Embedding and
    LearnedPositionalEmbedding.weights: N(0, 1))."""
    shapes = dict(manifest)
    out: Dict[str, torch.Tensor] = {}
    for name, shape in manifest:
        if name.endswith(".weights") or name.endswith("embedding.weight"):
            t = torch.randn(shape)
        elif len(shape) >= 2:
            fan_in = shape[1] * math.prod(shape[2:])
            b = 1.0 / math.sqrt(fan_in)
            t = torch.empty(shape).uniform_(-b, b)
        elif name.endswith(".weight"):
            t = torch.ones(shape)
        else:
            wshape = shapes.get(name[: -len("bias")] + "weight")
            if wshape is not None and len(wshape) >= 2:
                fan_in = wshape[1] * math.prod(wshape[2:])
                b = 1.0 / math.sqrt(fan_in)
                t = torch.empty(shape).uniform_(-b, b)
            else:
                t = torch.zeros(shape)
        out[name] = t
    return out


def attach_parameters(root: nn.Module, manifest: Manifest, values: Optional[Dict[str, torch.Tensor]] = None) -> None:
    values = values or _default_init(manifest)
    for name, shape in manifest:
        node = root
        parts = name.split(".")
        for part in parts[:-1]:
            child = node._modules.get(part)
            if child is None:
                child = ParamNode()
                node.add_module(part, child)
            node = child
        node.register_parameter(parts[-1], nn.Parameter(values[name].reshape(shape).clone()))


class UNetCFG1d(ParamNode):
    """Parameter owner for UNetCFG1d (modules.py:1211-1255) built by XUNet1d(type='cfg').

    Calling it evaluates the network through the owning model's engine (see generative.py); the
    signature follows the reference: ``net(x, time, *, embedding, embedding_scale=1.0)``."""

    def __init__(self, config: UNetConfig):
        super().__init__()
        self.config = config
        attach_parameters(self, unet_manifest(config))
        self._evaluator = None      # set by the owning QMDiffusion*/KDiffusion_mod

    def forward(self, x, time, *, embedding, embedding_scale: float = 1.0, **kwargs):
        if kwargs:
            raise TypeError(f"unsupported arguments: {sorted(kwargs)}")
        if self._evaluator is None:
            raise RuntimeError("this U-Net is not attached to a QMDiffusion / QMDiffusionForward model")
        return self._evaluator(x, time, embedding, embedding_scale)


class PositionalEncoding1D(nn.Module):
    """Buffer owner for transformer.py:3444-3470 (``inv_freq`` appears in the state_dict)."""

    def __init__(self, channels: int):
        super().__init__()
        self.org_channels = channels
        channels = int(math.ceil(channels / 2) * 2)
        self.channels = channels
        inv_freq = 1.0 / (10000 ** (torch.arange(0, channels, 2).float() / channels))
        self.register_buffer("inv_freq", inv_freq)
