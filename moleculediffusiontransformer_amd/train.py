"""Training loss of the class surface: QMDiffusion.forward / QMDiffusionForward.forward (generative.py:812-833,
:120-143) -> KDiffusion_mod.forward (diffusion.py:820-844) -> UNetCFG1d.forward (modules.py:1228-1255).

SURVEY §3.4: training is NOT on the MI355X hot path; it is kept as plain PyTorch with autograd on whatever device
the parameters live on, so that the reference's train loops (`loss = model(sequences, output); loss.backward()`)
run against these classes and the resulting state_dict feeds sample().  This module differentiates through the
network, which the inference kernels cannot; nothing here is called by sample()/inpaint(), and sample() never
falls back to it (it raises without an AMD GPU).

The network is evaluated on the parameter tree of modules.UNetCFG1d by dotted name (the reference's state_dict keys).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


class _Params:
    """Parameter lookup by the reference's key names relative to the U-Net root."""

    def __init__(self, unet):
        self.p = dict(unet.named_parameters())

    def __getitem__(self, key: str) -> Tensor:
        return self.p[key]

    def __contains__(self, key: str) -> bool:
        return key in self.p


def conditioning_embedding(model, sequences: Tensor) -> Tensor:
    """generative.py:815-828: fc1 -> GELU -> cat(PositionalEncoding1D) (transformer.py:3456-3470), with autograd."""
    x = sequences.float().unsqueeze(2).to(model.fc1.weight.device)
    x = model.GELUact(model.fc1(x))
    if model.pos_emb_fourier:
        inv_freq = model.p_enc_1d.inv_freq
        pos = torch.arange(x.shape[1], device=x.device, dtype=inv_freq.dtype)
        s = torch.einsum("i,j->ij", pos, inv_freq)
        # PositionalEncoding1D.forward returns emb[:, :, :orig_ch] with orig_ch = the INPUT's channel count (transformer.py:3460-3470)
        emb = torch.cat((s.sin(), s.cos()), dim=-1)[:, : x.shape[-1]].to(x.dtype)
        emb = emb.unsqueeze(0).expand(x.shape[0], -1, -1)
        x = x + emb if getattr(model, "pos_emb_fourier_add", False) else torch.cat((x, emb), dim=2)
    return x


def _conv_block(P, p, x, groups, scale_shift=None):
    """ConvBlock1d.forward (modules.py:114-122)."""
    x = F.group_norm(x, groups, P[p + "groupnorm.weight"], P[p + "groupnorm.bias"], eps=1e-5)
    if scale_shift is not None:
        x = x * (scale_shift[0] + 1) + scale_shift[1]
    return F.conv1d(F.silu(x), P[p + "project.weight"], P[p + "project.bias"], padding=1)


def _resnet(P, p, x, mapping, groups):
    """ResnetBlock1d.forward (modules.py:193-205)."""
    h = _conv_block(P, p + "block1.", x, groups)
    ss = F.linear(F.silu(mapping), P[p + "to_scale_shift.to_scale_shift.1.weight"], P[p + "to_scale_shift.to_scale_shift.1.bias"])
    h = _conv_block(P, p + "block2.", h, groups, ss.unsqueeze(-1).chunk(2, dim=1))
    if (p + "to_out.weight") in P:
        x = F.conv1d(x, P[p + "to_out.weight"], P[p + "to_out.bias"])
    return h + x


def _attention(P, p, x, context, heads):
    """Attention.forward + AttentionBase.forward (modules.py:401-410, :350-364)."""
    ctx = x if context is None else context
    xn = F.layer_norm(x, x.shape[-1:], P[p + "norm.weight"], P[p + "norm.bias"], eps=1e-5)
    cn = F.layer_norm(ctx, ctx.shape[-1:], P[p + "norm_context.weight"], P[p + "norm_context.bias"], eps=1e-5)
    q = F.linear(xn, P[p + "to_q.weight"])
    k, v = F.linear(cn, P[p + "to_kv.weight"]).chunk(2, dim=-1)
    b, n, _ = q.shape
    d = q.shape[-1] // heads
    q, k, v = (t.reshape(b, t.shape[1], heads, d).transpose(1, 2) for t in (q, k, v))
    attn = (torch.matmul(q, k.transpose(-1, -2)) * d ** -0.5).softmax(dim=-1)
    out = torch.matmul(attn, v).transpose(1, 2).reshape(b, n, heads * d)
    return F.linear(out, P[p + "attention.to_out.weight"], P[p + "attention.to_out.bias"])


def _transformer(P, p, x, context, heads):
    """Transformer1d.forward (modules.py:519-524) over TransformerBlock.forward (:456-461)."""
    x = F.group_norm(x, 32, P[p + "to_in.0.weight"], P[p + "to_in.0.bias"], eps=1e-6)
    x = F.conv1d(x, P[p + "to_in.1.weight"], P[p + "to_in.1.bias"]).transpose(1, 2)
    i = 0
    while (p + f"blocks.{i}.attention.to_q.weight") in P:
        bp = p + f"blocks.{i}."
        x = _attention(P, bp + "attention.", x, None, heads) + x
        if (bp + "cross_attention.to_q.weight") in P:
            assert context is not None, "You must provide a context when using context_features"
            x = _attention(P, bp + "cross_attention.", x, context, heads) + x
        h = F.gelu(F.linear(x, P[bp + "feed_forward.0.weight"], P[bp + "feed_forward.0.bias"]))
        x = F.linear(h, P[bp + "feed_forward.2.weight"], P[bp + "feed_forward.2.bias"]) + x
        i += 1
    return F.conv1d(x.transpose(1, 2), P[p + "to_out.1.weight"], P[p + "to_out.1.bias"])


def unet_forward(unet, x: Tensor, time: Tensor, embedding: Tensor) -> Tensor:
    """UNet1d.forward (modules.py:1144-1180) on the parameter tree of `unet` (modules.UNetCFG1d), differentiable."""
    cfg, P = unet.config, _Params(unet)
    heads, g, ps, nl = cfg.heads, cfg.resnet_groups, cfg.patch_size, cfg.num_layers
    t = time.unsqueeze(1)
    fr = t * P["to_time.0.0.weights"].unsqueeze(0) * 2 * math.pi
    m = F.gelu(F.linear(torch.cat((t, fr.sin(), fr.cos()), dim=-1), P["to_time.0.1.weight"], P["to_time.0.1.bias"]))
    m = F.gelu(F.linear(m, P["to_mapping.0.weight"], P["to_mapping.0.bias"]))
    mapping = F.gelu(F.linear(m, P["to_mapping.2.weight"], P["to_mapping.2.bias"]))

    x = _resnet(P, "to_in.block.", x, mapping, 1)
    if ps > 1:
        b, c, lp = x.shape
        x = x.view(b, c, lp // ps, ps).permute(0, 1, 3, 2).reshape(b, c * ps, lp // ps)
    skips_list = [x]
    for i in range(nl):
        dp, f = f"downsamples.{i}.", cfg.factors[i]
        x = F.conv1d(x, P[dp + "downsample.weight"], P[dp + "downsample.bias"], stride=f, padding=f)
        skips = []
        if cfg.pre_transformer > 0:
            x = _transformer(P, dp + "pre_transformer_block.", x, None, heads)
            skips.append(x)
        for j in range(cfg.num_blocks[i]):
            x = _resnet(P, dp + f"blocks.{j}.", x, mapping, g)
            skips.append(x)
        if cfg.attentions[i] > 0:
            x = _transformer(P, dp + "transformer.", x, embedding, heads)
            skips.append(x)
        skips_list.append(skips)
    x = _resnet(P, "bottleneck.pre_block.", x, mapping, g)
    if cfg.attentions[-1] > 0:
        x = _transformer(P, "bottleneck.transformer.", x, embedding, heads)
    x = _resnet(P, "bottleneck.post_block.", x, mapping, g)
    for u, i in enumerate(reversed(range(nl))):
        up, f = f"upsamples.{u}.", cfg.factors[i]
        skips = skips_list.pop()
        for j in range(cfg.num_blocks[i] + (1 if cfg.attentions[i] else 0)):
            x = _resnet(P, up + f"blocks.{j}.", torch.cat([x, skips.pop() * 2 ** -0.5], dim=1), mapping, g)
        if cfg.pre_transformer > 0:
            x = _transformer(P, up + "pre_transformer_block.", x, None, heads)
        if cfg.attentions[i] > 0:
            x = _transformer(P, up + "transformer.", x, embedding, heads)
        x = F.conv_transpose1d(x, P[up + "upsample.weight"], P[up + "upsample.bias"], stride=f,
                               padding=f // 2 + f % 2, output_padding=f % 2)
    x = x + skips_list.pop()
    if ps > 1:
        b, cp, l = x.shape
        x = x.view(b, cp // ps, ps, l).permute(0, 1, 3, 2).reshape(b, cp // ps, l * ps)
    return _resnet(P, "to_out.block.", x, mapping, 1)


def unet_cfg_forward(unet, x, time, embedding, embedding_scale: float = 1.0, embedding_mask_proba: float = 0.0):
    """UNetCFG1d.forward (modules.py:1228-1255) including the training-time random masking to the FixedEmbedding."""
    b, n = embedding.shape[0], embedding.shape[1]
    if n > unet.config.ctx_max_length:
        raise AssertionError("Input sequence length must be <= max_length")
    fixed = dict(unet.named_parameters())["fixed_embedding.embedding.weight"][:n].unsqueeze(0).expand(b, -1, -1)
    if embedding_mask_proba > 0.0:
        if embedding_mask_proba == 1:
            mask = torch.ones((b, 1, 1), dtype=torch.bool, device=embedding.device)
        else:
            mask = torch.bernoulli(torch.full((b, 1, 1), embedding_mask_proba, device=embedding.device)).to(torch.bool)
        embedding = torch.where(mask, fixed, embedding)
    if embedding_scale != 1.0:
        out = unet_forward(unet, x, time, embedding)
        out_masked = unet_forward(unet, x, time, fixed)
        return out_masked + (out - out_masked) * embedding_scale
    return unet_forward(unet, x, time, embedding)


def _clip(x: Tensor, dynamic_threshold: float = 0.0) -> Tensor:
    """clip(), diffusion.py:75-88, as the training objective sees it (denoise_fn :814): clamp to [-1, 1], or -- with
    ``dynamic_threshold`` > 0 -- clamp to the per-sample quantile of |x| (at least 1) and divide by it."""
    if dynamic_threshold == 0.0:
        return x.clamp(-1.0, 1.0)
    scale = torch.quantile(x.flatten(1).abs(), dynamic_threshold, dim=-1).clamp(min=1.0)
    scale = scale.view(-1, *((1,) * (x.ndim - 1)))
    return x.clamp(-scale, scale) / scale


def kdiffusion_loss(model, x: Tensor, noise: Optional[Tensor], embedding: Tensor, sigmas: Optional[Tensor] = None,
                    **kwargs) -> Tensor:
    """KDiffusion_mod.forward (diffusion.py:820-844): per-sample log-normal sigma, noised input, one denoise, weighted MSE.
    ``sigmas`` (testing aid) replaces the sigma_distribution draw."""
    kd = model.diffusion.diffusion
    sd = kd.sigma_data
    b, device = x.shape[0], x.device
    if sigmas is None:
        sigmas = kd.sigma_distribution(num_samples=b, device=device)
    sp = sigmas.view(-1, 1, 1)
    noise = torch.randn_like(x) if noise is None else noise
    x_noisy = x + sp * noise
    c_noise = torch.log(sigmas) * 0.25                                  # get_scale_weights, diffusion.py:789-796
    c_skip = (sd ** 2) / (sp ** 2 + sd ** 2)
    c_out = sp * sd * (sd ** 2 + sp ** 2) ** -0.5
    c_in = (sp ** 2 + sd ** 2) ** -0.5
    pred = unet_cfg_forward(model.unet, c_in * x_noisy, c_noise, embedding, **kwargs)
    x_denoised = _clip(c_skip * x_noisy + c_out * pred, float(kd.dynamic_threshold))
    losses = F.mse_loss(x_denoised, x, reduction="none").flatten(1).mean(dim=1)
    losses = losses * ((sigmas ** 2 + sd ** 2) * (sigmas * sd) ** -2)    # loss_weight, diffusion.py:816-818
    return losses.mean()
