"""Pins oracle/unet_oracle.py against golden vectors produced by the real reference."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from helpers import CASES, noise_fns, oracle_cfg, synth_sd, to_t
from oracle import unet_oracle as O

TOL = 2e-6   # oracle and reference issue the same ATen ops; observed difference is exactly 0


@pytest.mark.parametrize("case", ["cfg1", "cfg3", "tiny", "pd22", "nb", "sparse", "full"])
def test_unet_eval_matches_reference(case):
    g = load_golden(f"{case}_unet.npz")
    sd, cfg = synth_sd(case), oracle_cfg(case)
    with torch.no_grad():
        emb = O.cond_embed(sd, cfg, to_t(g["seq"]))
        assert (emb - to_t(g["emb"])).abs().max() <= TOL
        mp = O.time_mapping(sd, "unet.", to_t(g["t"]))
        assert (mp - to_t(g["mapping"])).abs().max() <= TOL
        taps = {}
        y = O.unet_forward(sd, cfg, to_t(g["x"]), to_t(g["t"]), emb, taps=taps)
        assert (y - to_t(g["y_scale1"])).abs().max() <= TOL
        y = O.unet_cfg_forward(sd, cfg, to_t(g["x"]), to_t(g["t"]), emb, 7.5)
        assert (y - to_t(g["y_scale7p5"])).abs().max() <= 4 * TOL
        d = O.denoise(sd, cfg, to_t(g["x"]) * 2.5, torch.tensor(2.5), emb, 1.0)
        assert (d - to_t(g["denoise_sigma2p5"])).abs().max() <= TOL
        for ours, theirs in (("to_in", "out:to_in"), ("down0", "out:downsamples.0"),
                             ("down1", "out:downsamples.1"), ("bottleneck", "out:bottleneck"),
                             ("up0", "out:upsamples.0"), ("up1", "out:upsamples.1")):
            if theirs in g:
                assert (taps[ours] - to_t(g[theirs])).abs().max() <= TOL, ours


@pytest.mark.parametrize("name,case,want", [
    ("cfg1_b4_t64", "cfg1", (1, 2, 32, 63)),
    ("cfg1_b2_t12_cfg7p5", "cfg1", ()),
    ("cfg3_b2_t10", "cfg3", ()),
    ("tiny_b3_t8", "tiny", (1, 7)),
    ("tiny_b3_t8_cfg2", "tiny", ()),
    ("pd22_b2_t6", "pd22", ()),
    ("nb_b2_t6", "nb", ()),
    ("nb_b2_t5_cfg2", "nb", ()),
    ("sparse_b2_t5", "sparse", ()),
    ("full_b2_t5", "full", ()),          # AnalogDiffusionFull, pos_emb_fourier_add=True (graphmodel.py:391-597)
    ("full_b2_t4_cfg3", "full", ()),
])
def test_sample_matches_reference(name, case, want):
    g = load_golden(f"{name}_sample.npz")
    sd, cfg = synth_sd(case), oracle_cfg(case)
    out_ref = to_t(g["out"])
    init, step = noise_fns(name, tuple(out_ref.shape))
    trace = {"want": want}
    out = O.sample(sd, cfg, to_t(g["seq"]), init, step, int(g["timesteps"]), float(g["cond_scale"]),
                   False, trace)
    assert out.shape == out_ref.shape and not out.requires_grad
    assert (out - out_ref).abs().max() <= TOL
    for s in want:
        assert (trace[s] - to_t(g[f"x_step{s}"])).abs().max() <= TOL


def test_inpaint_matches_reference():
    g = load_golden("tiny_inpaint.npz")
    sd, cfg = synth_sd("tiny"), oracle_cfg("tiny")
    n = {"i": 0}

    def draw(like):
        from moleculediffusiontransformer_amd.synth import synth_normal
        t = synth_normal(f"tiny_inpaint/draw{n['i']}", tuple(like.shape))
        n["i"] += 1
        return t
    with torch.no_grad():
        emb = O.cond_embed(sd, cfg, to_t(g["seq"]))
    src, mask = to_t(g["src"]), to_t(g["mask"])
    out = O.adpm2_inpaint(sd, cfg, src, mask, emb, int(g["timesteps"]), int(g["num_resamples"]), draw,
                          float(g["cond_scale"]))
    assert n["i"] == int(g["ndraws"])
    assert (out - to_t(g["out"])).abs().max() <= TOL
    assert torch.equal(out[mask], src[mask])      # kept region is returned bit-equal (diffusion.py:549)


def test_schedule_and_scalars_kats():
    g = load_golden("scalars.npz")
    for T in (64, 100, 12):
        sig = O.karras_sigmas(T)
        assert np.array_equal(sig.numpy(), g[f"sigmas_{T}"])
        for i in range(T - 1):
            up, down, mid = O.adpm2_sigmas(sig[i], sig[i + 1])
            assert up == g[f"up_{T}"][i] and down == g[f"down_{T}"][i]
            assert np.float32(float(mid)) == g[f"mid_{T}"][i]
    # SURVEY §8a known answers (measured on the reference)
    s64 = O.karras_sigmas(64)
    assert s64[1].item() == 8.598163604736328 and s64[62].item() == 0.0022702966816723347
    assert s64[63].item() == 0.0010000006295740604 and s64[64].item() == 0.0
    up, down, mid = O.adpm2_sigmas(s64[0], s64[1])
    assert up == 2.5405128444864213 and down == 8.214268844302843
    for row, s in zip(g["scale_weights"], (9.0, 1.0, 0.001)):
        w = O.scale_weights(torch.full((4,), s))
        assert np.array_equal(np.array([float(c.flatten()[0]) for c in w], dtype=np.float32), row)
    w = O.scale_weights(torch.full((4,), 9.0))
    assert float(w[0].flatten()[0]) == 0.000123441539471969
    assert float(w[3][0]) == 0.5493061542510986


def test_dynamic_thresholding_matches_reference():
    """clip() with dynamic_threshold > 0 (diffusion.py:75-88) and a sample with KDiffusion_mod.dynamic_threshold = 0.9: the
    oracle against vectors from the real reference (tests/golden/make_golden_r4.py dynthr)."""
    g = load_golden("dynthr.npz")
    x = to_t(g["x"])
    for q in (0.5, 0.9, 0.995, 1.0):
        assert torch.equal(O.clip(x, q), to_t(g[f"clip_q{q}"]))
    assert torch.equal(O.clip(x, 0.0), x.clamp(-1.0, 1.0))
    sd, cfg = synth_sd("tiny"), oracle_cfg("tiny")
    ref = to_t(g["sample_q0.9_t6"])
    init, step = noise_fns("tiny_dyn_t6", tuple(ref.shape))
    out = O.sample(sd, cfg, to_t(g["seq"]), init, step, 6, 1.0, False, None, 0.9)
    assert (out - ref).abs().max() <= TOL


@pytest.mark.parametrize("case", ["cfg1", "cfg3", "tiny", "pd22", "nb", "sparse", "full"])
def test_guided_denoise_matches_reference(case):
    """denoise_fn under embedding_scale = 7.5 (diffusion.py:798-814 over modules.py:1248-1253): what the sampler consumes of a
    guided evaluation (tests/golden/make_golden_r6.py guided_denoise)."""
    g, gd = load_golden(f"{case}_unet.npz"), load_golden("guided_denoise.npz")
    sd, cfg = synth_sd(case), oracle_cfg(case)
    with torch.no_grad():
        emb = O.cond_embed(sd, cfg, to_t(g["seq"]))
        d = O.denoise(sd, cfg, to_t(g["x"]) * 2.5, torch.tensor(2.5), emb, 7.5)
    assert (d - to_t(gd[f"{case}_denoise_sigma2p5_scale7p5"])).abs().max() <= TOL


def test_token_chain_matches_reference():
    """SURVEY 8 (f3), second half: ids -> reverse_tokenize -> texts_to_sequences -> pad_sequences(post, post) -> / X_norm_factor ->
    forward model, as the reference's own functions computed it (tests/golden/make_golden_r6.py token_chain: the keras
    tokenizer restated there, generative.py:1069-1078 and :404-451 themselves executed).  The host restatement is BIT-EQUAL on the
    forward input; the oracle's forward model reproduces the predicted properties."""
    from moleculediffusiontransformer_amd import tokens_to_forward_input
    g = load_golden("token_chain.npz")
    ids, L, xn = to_t(g["ids"]), int(g["max_length"]), float(g["X_norm_factor"])
    data = tokens_to_forward_input(ids, L, xn)
    assert data.dtype == torch.float32 and torch.equal(data, to_t(g["forward_input"]))
    # the strings in between are what the ids say: zeros dropped, order kept
    alphabet = "".join(g["alphabet"].tolist())
    assert ["".join(alphabet[i - 1] for i in row if i) for row in g["ids"].tolist()] == g["smiles"].tolist()
    sd, cfg = synth_sd("cfg3"), oracle_cfg("cfg3")
    T = int(g["timesteps"])
    init, step = noise_fns("r6_chain_t10", (ids.shape[0], 1, L))
    out = O.sample(sd, cfg, data, init, step, T, 1.0, False)
    assert (out[:, 0, :12] - to_t(g["result"])).abs().max() <= TOL


@pytest.mark.parametrize("name,case,tag,B,shape,T,uniform", [("cfg3_t100", "cfg3", "full3", 4096, (1, 64), 100, True),
                                                             ("cfg5_t16", "cfg5", "full5", 32, (32, 128), 16, False)])
def test_fullsize_probe_rows_match_reference(name, case, tag, B, shape, T, uniform):
    """tests/golden/fullsize_rows.npz (the reference's result for the probe rows of tests/test_gpu_fullsize.py's BASELINE-size runs,
    tests/golden/make_golden_r6.py fullsize_rows) against the oracle on the same named draws.  (cfg5_t256 -- 510 evaluations of the
    deep U-Net -- takes the oracle ~7 minutes of host time: it is checked on the GPU only.)"""
    from moleculediffusiontransformer_amd.synth import synth_normal, synth_uniform
    g = load_golden("fullsize_rows.npz")
    rows = torch.from_numpy(g[name + "_rows"])
    n_cond = CASES[case][1]["context_embedding_max_length"]
    seq = (synth_uniform if uniform else synth_normal)(f"{tag}/seq", (B, n_cond))[rows]
    init = synth_normal(f"{tag}/init", (B,) + shape)[rows]
    out = O.sample(synth_sd(case), oracle_cfg(case), seq, init, lambda i, x: synth_normal(f"{tag}/step{i}", (B,) + shape)[rows],
                   T, 1.0, False)
    assert (out - to_t(g[name])).abs().max() <= TOL
