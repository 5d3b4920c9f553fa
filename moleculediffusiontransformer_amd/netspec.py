"""Structure of the 1-D conditional U-Net behind QMDiffusion / QMDiffusionForward.

Pure description (no compute): hyper-parameters, the ordered parameter manifest
with the reference's state_dict key names, and the block walk the program
compiler follows.  Reference: UNet1d.__init__ (modules.py:934-1099),
UNetCFG1d.__init__ (:1215-1226), wrapper ctors (generative.py:69-83, :761-776).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Sequence, Tuple


@dataclass(frozen=True)
class UNetConfig:
    in_channels: int
    channels: int
    patch_size: int
    multipliers: Tuple[int, ...] = (1, 2, 4)
    factors: Tuple[int, ...] = (4, 4)
    num_blocks: Tuple[int, ...] = (3, 3)
    attentions: Tuple[int, ...] = (4, 4)
    heads: int = 8
    head_features: int = 64
    ff_mult: int = 2
    pre_transformer: int = 0
    resnet_groups: int = 8
    ctx_features: int = 128          # context_embedding_features
    ctx_max_length: int = 12         # context_embedding_max_length (FixedEmbedding rows)

    @property
    def num_layers(self) -> int:
        return len(self.multipliers) - 1

    @property
    def mapping_features(self) -> int:
        return self.channels * 4      # context_features_multiplier = 4 (modules.py:953, :994)

    @property
    def mid_features(self) -> int:
        return self.heads * self.head_features

    def level_channels(self, i: int) -> int:
        return self.channels * self.multipliers[i]


def inverse_unet_config(pred_dim: int, channels: int, ctx_features: int,
                        ctx_max_length: int) -> UNetConfig:
    """QMDiffusion(unet_type='cfg'): generative.py:761-776."""
    return UNetConfig(in_channels=pred_dim, channels=channels, patch_size=1,
                      attentions=(4, 4), pre_transformer=2,
                      ctx_features=ctx_features, ctx_max_length=ctx_max_length)


def forward_unet_config(pred_dim: int, channels: int, ctx_features: int,
                        ctx_max_length: int) -> UNetConfig:
    """QMDiffusionForward(unet_type='cfg'): generative.py:69-83."""
    return UNetConfig(in_channels=pred_dim, channels=channels, patch_size=4,
                      attentions=(2, 2), pre_transformer=0,
                      ctx_features=ctx_features, ctx_max_length=ctx_max_length)


def sparse_unet_config(pred_dim: int, channels: int, ctx_features: int,
                       ctx_max_length: int, patch_size: int = 8, num_blocks: Tuple[int, ...] = (2, 2)) -> UNetConfig:
    """The U-Net of AnalogDiffusionSparse / AnalogDiffusionFull (graphmodel.py:266-283, :433-446; patch_size 8 / 4): the
    same stack with a patching stem, two (Sparse) / three (Full: num_blocks=(3, 3)) ResNet blocks and ONE transformer layer per
    level, no pre-transformer.  Pass it as
    ``QMDiffusion(unet=UNetCFG1d(sparse_unet_config(...)))``."""
    return UNetConfig(in_channels=pred_dim, channels=channels, patch_size=patch_size, num_blocks=tuple(num_blocks),
                      attentions=(1, 1), pre_transformer=0, ctx_features=ctx_features, ctx_max_length=ctx_max_length)


Manifest = List[Tuple[str, Tuple[int, ...]]]


def _resnet(p: str, cin: int, cout: int, mapf: int) -> Manifest:
    m: Manifest = [
        (p + "block1.groupnorm.weight", (cin,)), (p + "block1.groupnorm.bias", (cin,)),
        (p + "block1.project.weight", (cout, cin, 3)), (p + "block1.project.bias", (cout,)),
        (p + "to_scale_shift.to_scale_shift.1.weight", (2 * cout, mapf)),
        (p + "to_scale_shift.to_scale_shift.1.bias", (2 * cout,)),
        (p + "block2.groupnorm.weight", (cout,)), (p + "block2.groupnorm.bias", (cout,)),
        (p + "block2.project.weight", (cout, cout, 3)), (p + "block2.project.bias", (cout,)),
    ]
    if cin != cout:
        m += [(p + "to_out.weight", (cout, cin, 1)), (p + "to_out.bias", (cout,))]
    return m


def _attention(p: str, c: int, mid: int, ctx: int) -> Manifest:
    return [
        (p + "norm.weight", (c,)), (p + "norm.bias", (c,)),
        (p + "norm_context.weight", (ctx,)), (p + "norm_context.bias", (ctx,)),
        (p + "to_q.weight", (mid, c)), (p + "to_kv.weight", (2 * mid, ctx)),
        (p + "attention.to_out.weight", (c, mid)), (p + "attention.to_out.bias", (c,)),
    ]


def _transformer(p: str, c: int, layers: int, cfg: UNetConfig, cross: bool) -> Manifest:
    m: Manifest = [(p + "to_in.0.weight", (c,)), (p + "to_in.0.bias", (c,)),
                   (p + "to_in.1.weight", (c, c, 1)), (p + "to_in.1.bias", (c,))]
    for i in range(layers):
        bp = p + f"blocks.{i}."
        m += _attention(bp + "attention.", c, cfg.mid_features, c)
        if cross:
            m += _attention(bp + "cross_attention.", c, cfg.mid_features, cfg.ctx_features)
        m += [(bp + "feed_forward.0.weight", (c * cfg.ff_mult, c)),
              (bp + "feed_forward.0.bias", (c * cfg.ff_mult,)),
              (bp + "feed_forward.2.weight", (c, c * cfg.ff_mult)),
              (bp + "feed_forward.2.bias", (c,))]
    m += [(p + "to_out.1.weight", (c, c, 1)), (p + "to_out.1.bias", (c,))]
    return m


def unet_manifest(cfg: UNetConfig, prefix: str = "") -> Manifest:
    """Parameter names/shapes in the reference's registration order."""
    p = prefix
    mapf = cfg.mapping_features
    c0 = cfg.level_channels(0)
    m: Manifest = [
        (p + "to_mapping.0.weight", (mapf, mapf)), (p + "to_mapping.0.bias", (mapf,)),
        (p + "to_mapping.2.weight", (mapf, mapf)), (p + "to_mapping.2.bias", (mapf,)),
        (p + "to_time.0.0.weights", (cfg.channels // 2,)),
        (p + "to_time.0.1.weight", (mapf, cfg.channels + 1)), (p + "to_time.0.1.bias", (mapf,)),
    ]
    m += _resnet(p + "to_in.block.", cfg.in_channels, c0 // cfg.patch_size, mapf)
    for i in range(cfg.num_layers):
        dp = p + f"downsamples.{i}."
        cin, cout = cfg.level_channels(i), cfg.level_channels(i + 1)
        f = cfg.factors[i]
        if cfg.pre_transformer > 0:
            m += _transformer(dp + "pre_transformer_block.", cout, cfg.pre_transformer, cfg, False)
        m += [(dp + "downsample.weight", (cout, cin, 2 * f + 1)), (dp + "downsample.bias", (cout,))]
        for j in range(cfg.num_blocks[i]):
            m += _resnet(dp + f"blocks.{j}.", cout, cout, mapf)
        if cfg.attentions[i] > 0:
            m += _transformer(dp + "transformer.", cout, cfg.attentions[i], cfg, True)
    cb = cfg.level_channels(cfg.num_layers)
    bp = p + "bottleneck."
    m += _resnet(bp + "pre_block.", cb, cb, mapf)
    if cfg.attentions[-1] > 0:
        m += _transformer(bp + "transformer.", cb, cfg.attentions[-1], cfg, True)
    m += _resnet(bp + "post_block.", cb, cb, mapf)
    for u, i in enumerate(reversed(range(cfg.num_layers))):
        up = p + f"upsamples.{u}."
        cin, cout = cfg.level_channels(i + 1), cfg.level_channels(i)
        f = cfg.factors[i]
        if cfg.pre_transformer > 0:
            m += _transformer(up + "pre_transformer_block.", cin, cfg.pre_transformer, cfg, False)
        n_res = cfg.num_blocks[i] + (1 if cfg.attentions[i] else 0)
        for j in range(n_res):
            m += _resnet(up + f"blocks.{j}.", 2 * cin, cin, mapf)
        if cfg.attentions[i] > 0:
            m += _transformer(up + "transformer.", cin, cfg.attentions[i], cfg, True)
        m += [(up + "upsample.weight", (cin, cout, 2 * f)), (up + "upsample.bias", (cout,))]
    m += _resnet(p + "to_out.block.", c0 // cfg.patch_size, cfg.in_channels, mapf)
    m += [(p + "fixed_embedding.embedding.weight", (cfg.ctx_max_length, cfg.ctx_features))]
    return m
