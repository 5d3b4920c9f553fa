"""-m gpu: BASELINE.json's configurations at their FULL sizes (the golden fixtures and the oracle cover a few rows; per-sample
arithmetic does not depend on the batch around it, so those rows pin the whole batch).

Round 6: the probe rows of the long runs (configs[4] at 256 / 16 timesteps, configs[2] at 100) are compared with
tests/golden/fullsize_rows.npz -- what the REAL reference returns for those rows on the same named draws
(tests/golden/make_golden_r6.py fullsize_rows) -- instead of running the oracle on the GPU box's host at test time (396 s + 90 s +
2 x 30 s of the 951 s the suite took in round 5)."""
import pytest
import torch

from conftest import load_golden
from gpu_util import DEV, make_model
from helpers import noise_fns, oracle_cfg, synth_sd, to_t
from moleculediffusiontransformer_amd import NoiseSource, runtime as rt
from moleculediffusiontransformer_amd.synth import synth_normal, synth_uniform
from oracle import unet_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


def test_configs1_batch_1024_rows_of_the_golden_fixture():
    """configs[1]: inverse model, B = 1024, 64 steps.  Rows 0-3 carry the conditioning and the noise draws of the
    reference-generated fixture cfg1_b4_t64: they must match it (<= 1e-4) and equal a B = 4 run bit for bit."""
    g = load_golden("cfg1_b4_t64_sample.npz")
    m = make_model("cfg1")
    B, T = 1024, 64
    seq4, out4 = to_t(g["seq"]), to_t(g["out"])
    init4, step4 = noise_fns("cfg1_b4_t64", tuple(out4.shape))
    seq = torch.cat([seq4, synth_normal("full1/seq", (B - 4, 12))])
    init = torch.cat([init4, synth_normal("full1/init", (B - 4, 16, 64))])
    steps = lambda i: torch.cat([step4(i, init4), synth_normal(f"full1/step{i}", (B - 4, 16, 64))])     # noqa: E731
    out = m.sample(seq, DEV, cond_scale=1.0, timesteps=T, clamp=False, noise=NoiseSource(init=init, steps=steps))
    assert out.shape == (B, 16, 64) and torch.isfinite(out).all()
    assert (out[:4].cpu() - out4).abs().max() < TOL
    small = m.sample(seq4, DEV, cond_scale=1.0, timesteps=T, clamp=False,
                     noise=NoiseSource(init=init4, steps=lambda i: step4(i, init4)))
    assert torch.equal(out[:4], small)


def test_configs2_forward_model_batch_4096_100_steps():
    """configs[2]: QMDiffusionForward, B = 4096, 100 steps; four probe rows against the REFERENCE's result on identical noise
    (fullsize_rows.npz: cfg3_t100)."""
    m = make_model("cfg3")
    B, T = 4096, 100
    seq = synth_uniform("full3/seq", (B, 64))
    init = synth_normal("full3/init", (B, 1, 64))
    nz = [synth_normal(f"full3/step{i}", (B, 1, 64)) for i in range(T - 1)]
    out = m.sample(seq, DEV, cond_scale=1.0, timesteps=T, clamp=False, noise=NoiseSource(init=init, steps=lambda i: nz[i]))
    assert out.shape == (B, 1, 64) and torch.isfinite(out).all()
    g = load_golden("fullsize_rows.npz")
    rows = torch.from_numpy(g["cfg3_t100_rows"])
    assert rows.tolist() == [0, 1, 2047, 4095]
    assert (out.cpu()[rows] - to_t(g["cfg3_t100"])).abs().max() < TOL


def test_configs3_shard_of_8192_is_shard_invariant():
    """configs[3] (65,536 molecules over 8 GPUs) per-GPU shard: B = 8192, 64 steps, counter-based noise keyed by the global
    sample index.  One call == two calls of 4096 with sample0 = 0 / 4096, bit for bit (what the all-gather relies on)."""
    m = make_model("cfg1")
    B, T = 8192, 64
    seq = synth_normal("full4/seq", (B, 12))
    full = m.sample(seq, DEV, cond_scale=1.0, timesteps=T, noise=NoiseSource(seed=77, sample0=0))
    assert full.shape == (B, 16, 64) and torch.isfinite(full).all()
    lo = m.sample(seq[:4096], DEV, cond_scale=1.0, timesteps=T, noise=NoiseSource(seed=77, sample0=0))
    hi = m.sample(seq[4096:], DEV, cond_scale=1.0, timesteps=T, noise=NoiseSource(seed=77, sample0=4096))
    assert torch.equal(full[:4096], lo) and torch.equal(full[4096:], hi)


def test_configs4_deep_unet_batch_32_16_steps():
    """configs[4] architecture (channels 256, pred_dim 32, max_len 128) at B = 32, 16 steps (fp32-class products); two probe
    rows against the oracle."""
    m = make_model("cfg5")
    B, T = 32, 16
    seq = synth_normal("full5/seq", (B, 12))
    init = synth_normal("full5/init", (B, 32, 128))
    nz = [synth_normal(f"full5/step{i}", (B, 32, 128)) for i in range(T - 1)]
    out = m.sample(seq, DEV, cond_scale=1.0, timesteps=T, clamp=False, noise=NoiseSource(init=init, steps=lambda i: nz[i]))
    assert out.shape == (B, 32, 128) and torch.isfinite(out).all()
    g = load_golden("fullsize_rows.npz")
    rows = torch.from_numpy(g["cfg5_t16_rows"])
    assert rows.tolist() == [0, 31]
    assert (out.cpu()[rows] - to_t(g["cfg5_t16"])).abs().max() < TOL


def test_configs4_plain_bf16_mode():
    """configs[4] names bf16: the reduced-precision mode (gemm_mode 'bf16': one bf16 MFMA per product, bf16 GEMM operands,
    fp32 accumulation / residual stream / normalisation statistics) against the fp32 oracle on the same noise.  Stated budget
    of the mode: max-abs <= 1e-2 on samples of O(1) magnitude after 16 steps (measured 1.2e-3 - 1.6e-3), the decoded tokens
    (argmax over the 32 channels, generative.py:1212-1213) equal to the fp32 result wherever the fp32 argmax margin exceeds twice
    the deviation, on the oracle's probe rows and over the whole batch (measured: 99.6 % of all tokens; random-weight samples
    are nearly tied)."""
    m = make_model("cfg5")
    m.gemm_mode = "bf16"
    B, T = 32, 16
    seq = synth_normal("full5/seq", (B, 12))
    init = synth_normal("full5/init", (B, 32, 128))
    nz = [synth_normal(f"full5/step{i}", (B, 32, 128)) for i in range(T - 1)]
    out = m.sample(seq, DEV, cond_scale=1.0, timesteps=T, clamp=False, noise=NoiseSource(init=init, steps=lambda i: nz[i]))
    eng = m._engine
    assert eng.c.gemm_mode == "bf16" and any(op.kind == rt.OP_PREP16 for op in eng.c.programs["eval"])
    assert out.shape == (B, 32, 128) and torch.isfinite(out).all()
    g = load_golden("fullsize_rows.npz")
    rows, ref = torch.from_numpy(g["cfg5_t16_rows"]), to_t(g["cfg5_t16"])        # the reference's result for rows 0 / 31
    got = out.cpu()[rows]
    err = (got - ref).abs().max().item()
    agree = (got.argmax(1) == ref.argmax(1)).float().mean().item()
    print(f"bf16 mode: max-abs {err:.3e}, token agreement {agree:.4f}")
    top2r = ref.topk(2, dim=1).values
    flips_r = got.argmax(1) != ref.argmax(1)
    assert err < 1e-2 and agree >= 0.99
    assert not (flips_r & ((top2r[:, 0] - top2r[:, 1]) > 2 * err)).any()      # only nearly tied channels may flip (see below)
    # and against this build's fp32-class mode on all rows
    m.gemm_mode = "bf16x3"
    out3 = m.sample(seq, DEV, cond_scale=1.0, timesteps=T, clamp=False, noise=NoiseSource(init=init, steps=lambda i: nz[i]))
    err3 = (out - out3).abs().max().item()
    agree3 = (out.argmax(1) == out3.argmax(1)).float().mean().item()
    print(f"bf16 vs bf16x3, all {B} rows: max-abs {err3:.3e}, token agreement {agree3:.4f}")
    # random-weight samples have nearly tied channels: a token can only flip where the fp32-class margin (top-1 minus top-2)
    # is below twice the deviation; everywhere else the tokens must agree
    top2 = out3.topk(2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]
    flips = out.argmax(1) != out3.argmax(1)
    print(f"flipped tokens: {int(flips.sum())} of {flips.numel()}, largest margin among them {margin[flips].max().item() if flips.any() else 0:.3e}")
    assert err3 < 1e-2 and agree3 >= 0.99
    assert not (flips & (margin > 2 * err3)).any()


@pytest.fixture(scope="module")
def cfg5_256_steps():
    """configs[4] at its stated length (256 timesteps = 510 U-Net evaluations, diffusion.py:517-524), B = 8, explicit noise in
    the reference's call order; rows 0 and 7 as the REFERENCE computed them on identical noise (fullsize_rows.npz: cfg5_t256)."""
    B, T = 8, 256
    seq = synth_normal("full5long/seq", (B, 12))
    init = synth_normal("full5long/init", (B, 32, 128))
    nz = [synth_normal(f"full5long/step{i}", (B, 32, 128)) for i in range(T - 1)]
    g = load_golden("fullsize_rows.npz")
    rows, ref = torch.from_numpy(g["cfg5_t256_rows"]), to_t(g["cfg5_t256"])
    assert rows.tolist() == [0, 7]
    return B, T, seq, init, nz, rows, ref


@pytest.mark.parametrize("mode,budget", [("bf16x3", 1e-4), ("bf16", 1e-2)])
def test_configs4_at_256_steps(cfg5_256_steps, mode, budget):
    """VERDICT r2 configs_untested: the deep U-Net through ALL 256 timesteps, fp32-class products (<= 1e-4, the path's contract)
    and the reduced-precision mode BASELINE configs[4] names (<= 1e-2), against the reference on two probe rows."""
    B, T, seq, init, nz, rows, ref = cfg5_256_steps
    m = make_model("cfg5")
    m.gemm_mode = mode
    out = m.sample(seq, DEV, cond_scale=1.0, timesteps=T, clamp=False, noise=NoiseSource(init=init, steps=lambda i: nz[i]))
    assert out.shape == (B, 32, 128) and torch.isfinite(out).all()
    err = (out.cpu()[rows] - ref).abs().max().item()
    agree = (out.cpu()[rows].argmax(1) == ref.argmax(1)).float().mean().item()
    print(f"cfg5, 256 steps, {mode}: max-abs {err:.3e} vs the reference, token agreement {agree:.4f}")
    assert err < budget


def test_wide_path_at_batch_2048_against_the_oracle():
    """The whole-transformer form of the 256-channel level (k_tf256 without the pair split: the default above 1024 samples of 4 tokens)
    AT SIZE: B = 2048, 6 timesteps, four probe rows against the pinned oracle on identical noise (<= 1e-4)."""
    m = make_model("cfg1")
    B, T = 2048, 6
    seq = synth_normal("wide2048/seq", (B, 12))
    init = synth_normal("wide2048/init", (B, 16, 64))
    nz = [synth_normal(f"wide2048/step{i}", (B, 16, 64)) for i in range(T - 1)]
    out = m.sample(seq, DEV, cond_scale=1.0, timesteps=T, clamp=False, noise=NoiseSource(init=init, steps=lambda i: nz[i]))
    assert m._engine.c.tf256 and any(op.kind == rt.OP_TF256 and op.i[rt.F_NSPLIT] == 1 for op in m._engine.c.programs["eval"])
    rows = torch.tensor([0, 1023, 1024, 2047])
    ref = O.sample(synth_sd("cfg1"), oracle_cfg("cfg1"), seq[rows], init[rows], lambda i, x: nz[i][rows], T, 1.0, False)
    assert (out.cpu()[rows] - ref).abs().max() < TOL
    # the pair-split form on the same rows (what batches up to 1024 run): same answer to rounding, own oracle check
    m.kernel_choice = "narrow"
    out_n = m.sample(seq[:1024], DEV, cond_scale=1.0, timesteps=T, clamp=False,
                     noise=NoiseSource(init=init[:1024], steps=lambda i: nz[i][:1024]))
    assert not m._engine.c.tf256 and m._engine.handoff_status() == 0
    assert (out_n.cpu()[[0, 1023]] - ref[:2]).abs().max() < TOL


def test_reference_default_constructor_samples():
    """QMDiffusion() with NO arguments -- the reference's defaults (generative.py:720-736: max_length 1024, channels 128, pred_dim 1,
    32 conditioning tokens, text_embed_dim 1024 + embed_dim_position 64 = 1088 context features) -- builds, samples on the GPU and
    agrees with the oracle on identical noise (B = 1, 2 timesteps = 2 U-Net evaluations).  Rounds 1-3 refused this configuration
    at engine construction (sampler tile above 64 KiB of LDS, 256-token attention level)."""
    import torch
    from moleculediffusiontransformer_amd import NoiseSource, QMDiffusion
    from moleculediffusiontransformer_amd.netspec import inverse_unet_config, unet_manifest
    from moleculediffusiontransformer_amd.synth import synth_normal, synth_state_dict
    from oracle import unet_oracle as O
    m = QMDiffusion()
    assert m.max_length == 1024 and m.unet.config.ctx_features == 1088 and m.unet.config.ctx_max_length == 32
    m.load_state_dict(synth_state_dict([(k, tuple(v.shape)) for k, v in m.state_dict().items()]))
    m = m.to(DEV)
    keys = [("fc1.weight", (1024, 1)), ("fc1.bias", (1024,)), ("p_enc_1d.inv_freq", (32,))]
    keys += unet_manifest(inverse_unet_config(1, 128, 1088, 32), "unet.")
    sd = synth_state_dict(keys)
    cfg = O.inverse_config(1024, 128, 1, 32, text_embed_dim=1024, embed_dim_position=64)
    seq = synth_normal("default/seq", (1, 32))
    init = synth_normal("default/init", (1, 1, 1024))
    nz = [synth_normal("default/s0", (1, 1, 1024))]
    want = O.sample(sd, cfg, seq, init, lambda i, x: nz[i], 2, 1.0, False)
    out = m.sample(seq, DEV, cond_scale=1.0, timesteps=2, noise=NoiseSource(init=init, steps=lambda i: nz[i]))
    assert out.shape == (1, 1, 1024) and (out.cpu() - want).abs().max() < 1e-4
