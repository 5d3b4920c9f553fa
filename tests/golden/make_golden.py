"""Generate tests/golden/*.npz from the REAL reference (build container only).

    python tests/golden/make_golden.py

Imports /root/reference through oracle/ref_import.py, loads the deterministic
synthetic weights (moleculediffusiontransformer_amd/synth.py) into the reference
models with load_state_dict, injects deterministic noise in the reference's own
RNG call order (torch.randn once at generative.py:853/:164, torch.randn_like per
step at diffusion.py:514; inpaint order diffusion.py:535-547) and records inputs
and outputs.  Only data is stored: weights and noise are regenerated from their
names by the tests.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.ref_import import import_reference  # noqa: E402
from moleculediffusiontransformer_amd.synth import (synth_state_dict, synth_normal,  # noqa: E402
                                                    synth_uniform)

OUT = os.path.dirname(os.path.abspath(__file__))
MD = import_reference()


def build(kind, **kw):
    cls = MD.QMDiffusion if kind == "inverse" else MD.QMDiffusionForward
    m = cls(unet_type="cfg", pos_emb_fourier=True, pos_emb_fourier_add=False, **kw).eval()
    sd = synth_state_dict([(k, tuple(v.shape)) for k, v in m.state_dict().items()])
    m.load_state_dict(sd)
    return m


class NoiseInjector:
    """Replaces torch.randn / torch.randn_like by named deterministic draws, in call order."""

    def __init__(self, tag):
        self.tag, self.n = tag, 0

    def _draw(self, shape):
        t = synth_normal(f"{self.tag}/draw{self.n}", tuple(shape))
        self.n += 1
        return t

    def __enter__(self):
        self._r, self._rl = torch.randn, torch.randn_like
        torch.randn = lambda *s, **k: self._draw(s[0] if len(s) == 1 and not isinstance(s[0], int) else s)
        torch.randn_like = lambda x, **k: self._draw(x.shape)
        return self

    def __exit__(self, *a):
        torch.randn, torch.randn_like = self._r, self._rl


def embed(m, seq):
    with torch.no_grad():
        e = m.GELUact(m.fc1(seq.float().unsqueeze(2)))
        if getattr(m, "pos_emb_fourier_add", False):          # generative.py:844-846 / graphmodel.py:338-339
            return e + m.p_enc_1d(e)
        return torch.cat((e, m.p_enc_1d(e)), 2)


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name), **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v))
                                                    for k, v in arrs.items()})
    print("wrote", name, {k: tuple(np.asarray(v).shape) for k, v in arrs.items()})


def unet_case(tag, m, B, L, pd, cond_len, seq, hooks=()):
    x = synth_normal(f"{tag}/x", (B, pd, L))
    t = torch.tensor([0.3, -0.9, 0.05, -1.5][:B])
    rec = {}
    handles = []
    for hname in hooks:
        mod = m.unet.get_submodule(hname)

        def fn(mod_, args, kwargs, out, hname=hname):
            rec["in:" + hname] = args[0].detach().clone()
            rec["out:" + hname] = (out[0] if isinstance(out, tuple) else out).detach().clone()
        handles.append(mod.register_forward_hook(fn, with_kwargs=True))
    with torch.no_grad():
        emb = embed(m, seq)
        mapping = m.unet.get_mapping(t)
        y1 = m.unet(x, t, embedding=emb, embedding_scale=1.0)
    for h in handles:
        h.remove()
    with torch.no_grad():
        y75 = m.unet(x, t, embedding=emb, embedding_scale=7.5)
        den = m.diffusion.diffusion.denoise_fn(x * 2.5, sigma=torch.tensor(2.5), embedding=emb,
                                               embedding_scale=1.0)
    save(f"{tag}_unet.npz", seq=seq, x=x, t=t, emb=emb, mapping=mapping, y_scale1=y1, y_scale7p5=y75,
         denoise_sigma2p5=den, **rec)


def sample_case(tag, m, seq, T, cond_scale, want=()):
    inj = NoiseInjector(tag)
    rec = {}
    if want:
        smp = MD.ADPM2Sampler
        orig_step = smp.step
        cnt = {"i": 0}

        def step(self, x, fn, sigma, sigma_next):
            out = orig_step(self, x, fn, sigma, sigma_next)
            cnt["i"] += 1
            if cnt["i"] in want:
                rec[f"x_step{cnt['i']}"] = out.detach().clone()
            return out
        smp.step = step
    try:
        with inj:
            out = m.sample(seq, "cpu", cond_scale=cond_scale, timesteps=T, clamp=False)
    finally:
        if want:
            smp.step = orig_step
    assert inj.n == T, (inj.n, T)       # 1 initial draw + (T-1) step draws
    save(f"{tag}_sample.npz", seq=seq, timesteps=T, cond_scale=cond_scale, out=out, **rec)


def main():
    torch.set_num_threads(8)
    # ---- cfg-1: inverse c=64, pred_dim=16, L=64, cond_len=12 (README.md:98-134) ----
    m = build("inverse", max_length=64, pred_dim=16, channels=64, context_embedding_max_length=12,
              text_embed_dim=64, embed_dim_position=64)
    seq = synth_normal("cfg1/seq", (4, 12))
    unet_case("cfg1", m, 2, 64, 16, 12, seq[:2], hooks=(
        "to_in", "downsamples.0", "downsamples.1", "bottleneck", "upsamples.0", "upsamples.1",
        "downsamples.0.pre_transformer_block", "downsamples.0.blocks.0", "downsamples.0.transformer",
        "downsamples.0.transformer.blocks.0.attention", "downsamples.0.transformer.blocks.0.cross_attention",
        "upsamples.0.blocks.0", "to_out"))
    sample_case("cfg1_b4_t64", m, seq, 64, 1.0, want=(1, 2, 32, 63))
    sample_case("cfg1_b2_t12_cfg7p5", m, seq[:2], 12, 7.5)
    # Karras schedule + per-step ADPM2 scalars + preconditioning scalars (KATs of SURVEY §8a)
    ks = MD.KarrasSchedule(sigma_min=0.001, sigma_max=9.0, rho=3.0)
    smp = MD.ADPM2Sampler(rho=1)
    rows = {}
    for T in (64, 100, 12):
        sig = ks(T, "cpu")
        ups, downs, mids = [], [], []
        for i in range(T - 1):
            u, d, md = smp.get_sigmas(sig[i], sig[i + 1])
            ups.append(u), downs.append(d), mids.append(float(md))
        rows[f"sigmas_{T}"] = sig.numpy()
        rows[f"up_{T}"] = np.array(ups, dtype=np.float64)
        rows[f"down_{T}"] = np.array(downs, dtype=np.float64)
        rows[f"mid_{T}"] = np.array(mids, dtype=np.float32)
    kd = m.diffusion.diffusion
    sw = [kd.get_scale_weights(torch.full((4,), s)) for s in (9.0, 1.0, 0.001)]
    rows["scale_weights"] = np.array([[float(c.flatten()[0]) for c in w] for w in sw], dtype=np.float32)
    save("scalars.npz", **rows)

    # ---- cfg-3: forward predictor c=64, pred_dim=1, L=64, cond_len=64 (README.md:76-94) ----
    mf = build("forward", max_length=64, pred_dim=1, channels=64, context_embedding_max_length=64,
               text_embed_dim=64, embed_dim_position=64)
    seqf = synth_uniform("cfg3/seq", (2, 64))
    unet_case("cfg3", mf, 2, 64, 1, 64, seqf, hooks=("to_in", "downsamples.0", "downsamples.1", "bottleneck",
                                                      "upsamples.0", "upsamples.1"))
    sample_case("cfg3_b2_t10", mf, seqf, 10, 1.0)

    # ---- tiny inverse (channels=16) and a padded-channel inverse (pred_dim=22, L=32) ----
    mt = build("inverse", max_length=32, pred_dim=16, channels=16, context_embedding_max_length=12,
               text_embed_dim=64, embed_dim_position=64)
    seqt = synth_normal("tiny/seq", (3, 12))
    unet_case("tiny", mt, 3, 32, 16, 12, seqt)
    sample_case("tiny_b3_t8", mt, seqt, 8, 1.0, want=(1, 7))
    sample_case("tiny_b3_t8_cfg2", mt, seqt, 8, 2.0)
    mp = build("inverse", max_length=32, pred_dim=22, channels=32, context_embedding_max_length=12,
               text_embed_dim=64, embed_dim_position=64)
    unet_case("pd22", mp, 2, 32, 22, 12, seqt[:2])
    sample_case("pd22_b2_t6", mp, seqt[:2], 6, 1.0)

    # ---- inpainting on the tiny model (diffusion.py:526-549; generative.py:871-914) ----
    src = synth_uniform("tiny/inpaint_src", (3, 16, 32)) * 2 - 1
    mask = torch.zeros(3, 16, 32, dtype=torch.bool)
    mask[:, :, :12] = True
    inj = NoiseInjector("tiny_inpaint")
    with inj:
        out = mt.inpaint(seqt, "cpu", cond_scale=2.0, timesteps=6, num_resamples=2, inpaint=src,
                         in_paint_mask=mask)
    save("tiny_inpaint.npz", seq=seqt, src=src, mask=mask, out=out, ndraws=inj.n, timesteps=6,
         num_resamples=2, cond_scale=2.0)

    # ---- state_dict key layout KATs (SURVEY §5 checkpoint row) ----
    keys = {}
    for tag, mod in (("cfg1", m), ("cfg3", mf)):
        sd = mod.state_dict()
        keys[tag + "_keys"] = np.array(list(sd.keys()))
        keys[tag + "_numel"] = np.array([v.numel() for v in sd.values()])
        keys[tag + "_nparams"] = sum(p.numel() for p in mod.parameters())
    save("state_dict_keys.npz", **keys)


if __name__ == "__main__":
    main()
