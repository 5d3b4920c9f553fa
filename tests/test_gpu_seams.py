"""-m gpu: the class-surface seams around the fused loop (SURVEY section 8 b / f3): per-row time in net(x, time), per-sample
sigmas in denoise_fn, ADPM2Sampler.step / forward with a caller-supplied fn, DiffusionSampler as a callable object, the
decode step fused into the last update, the on-device inverse -> forward chain, the doubled guidance batch's guard."""
import pytest
import torch

from conftest import load_golden
from gpu_util import DEV, make_model
from helpers import noise_fns, to_t
from moleculediffusiontransformer_amd import (ADPM2Sampler, DiffusionSampler, KarrasSchedule, NoiseSource, QMDiffusion,
                                              generate_and_validate, predict_properties_from_tokens)
from moleculediffusiontransformer_amd.synth import synth_normal, synth_state_dict, synth_uniform

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["tiny", "cfg1"])
def test_net_takes_one_time_value_per_row(case):
    """modules.py:1228: net(x, time) with a (B,) time vector -- the golden batch uses a different time per row."""
    g = load_golden(f"{case}_unet.npz")
    m = make_model(case)
    emb = m._embed(to_t(g["seq"]), DEV)
    x, t = to_t(g["x"]).to(DEV), to_t(g["t"])
    y = m.unet(x, t, embedding=emb, embedding_scale=1.0)
    assert (y.cpu() - to_t(g["y_scale1"])).abs().max() < 5e-5
    # per-sample sigmas (the training-time form of denoise_fn, diffusion.py:798-808) == one call per row
    sig = torch.tensor([2.5, 0.7, 2.5, 0.05][: x.shape[0]])
    d = m.diffusion.diffusion.denoise_fn(x, sigmas=sig, embedding=emb)
    for b in range(x.shape[0]):
        one = m.diffusion.diffusion.denoise_fn(x[b:b + 1], sigma=sig[b], embedding=emb[b:b + 1])
        assert torch.equal(d[b:b + 1], one)


def test_sampler_step_with_a_caller_supplied_fn_equals_the_fused_loop():
    """ADPM2Sampler.forward(noise, fn, sigmas, num_steps) (diffusion.py:517-524) with an opaque fn runs step() per
    iteration on mdt_adpm2_euler; with the model's own denoiser bound by DiffusionSampler it takes the fused loop.  Same
    arithmetic in the same order: equal bit for bit, and equal to the reference's golden sample."""
    g = load_golden("tiny_b3_t8_sample.npz")
    m = make_model("tiny")
    seq, T = to_t(g["seq"]), int(g["timesteps"])
    init, step = noise_fns("tiny_b3_t8", tuple(g["out"].shape))
    fused = m.sample(seq, DEV, cond_scale=1.0, timesteps=T, noise=NoiseSource(init=init, steps=lambda i: step(i, init)))
    emb = m._embed(seq, DEV)
    kd = m.diffusion.diffusion
    calls = {"n": 0}

    def fn(x, sigma):                       # an opaque callable: no `fused` attribute
        calls["n"] += 1
        return kd.denoise_fn(x, sigma=sigma, embedding=emb, embedding_scale=1.0)

    s = ADPM2Sampler(rho=1)
    sigmas = KarrasSchedule(0.001, 9.0, 3.0)(T)
    x = (float(sigmas[0]) * init).to(DEV)
    for i in range(T - 1):
        x = s.step(x, fn, sigmas[i], sigmas[i + 1], noise=step(i, init))
    assert calls["n"] == 2 * (T - 1)
    assert torch.equal(x, fused)
    assert (x.cpu() - to_t(g["out"])).abs().max() < 1e-4
    # the object form of the reference: DiffusionSampler(diffusion, sampler=..., sigma_schedule=...)(noise, **kwargs)
    ds = DiffusionSampler(kd, sampler=s, sigma_schedule=KarrasSchedule(0.001, 9.0, 3.0), num_steps=T, clamp=False)
    y = ds(NoiseSource(init=init, steps=lambda i: step(i, init)), embedding=emb, embedding_scale=1.0)
    assert torch.equal(y, fused)
    # forward() with an opaque fn draws its step noise from torch's device generator, like diffusion.py:514
    torch.manual_seed(5)
    a = s(init.to(DEV), fn=fn, sigmas=sigmas, num_steps=4)
    torch.manual_seed(5)
    b = s(init.to(DEV), fn=fn, sigmas=sigmas, num_steps=4)
    assert torch.equal(a, b) and torch.isfinite(a).all()


@pytest.mark.parametrize("name,case", [("tiny_b3_t8", "tiny"), ("cfg1_b2_t12_cfg7p5", "cfg1"), ("pd22_b2_t6", "pd22")])
def test_tokens_decoded_in_the_last_update_are_bit_exact(name, case):
    """generative.py:1212-1213: permute(0, 2, 1) -> argmax(dim=2) of the fp32 sample == the ids written by the last
    mdt_adpm2_next, through the Python API; also on the clamped sample (ties at +-1: first maximum)."""
    g = load_golden(f"{name}_sample.npz")
    m = make_model(case)
    seq, T, cs = to_t(g["seq"]), int(g["timesteps"]), float(g["cond_scale"])
    init, step = noise_fns(name, tuple(g["out"].shape))
    ns = lambda: NoiseSource(init=init, steps=lambda i: step(i, init))     # noqa: E731
    tok, x = m.sample_tokens(seq, DEV, cond_scale=cs, timesteps=T, noise=ns(), return_sample=True)
    assert tok.dtype == torch.int64 and tok.shape == (x.shape[0], x.shape[2]) and tok.device.type == "cuda"
    assert torch.equal(tok, torch.argmax(torch.permute(x, (0, 2, 1)), dim=2))
    assert torch.equal(tok.cpu(), torch.argmax(torch.permute(to_t(g["out"]), (0, 2, 1)), dim=2))
    tokc, xc = m.sample_tokens(seq, DEV, cond_scale=cs, timesteps=T, clamp=True, noise=ns(), return_sample=True)
    assert float(xc.abs().max()) <= 1.0 and torch.equal(tokc, torch.argmax(torch.permute(xc, (0, 2, 1)), dim=2))


def test_inverse_to_forward_chain_stays_on_the_device():
    """generate_from_conditioning's core (generative.py:1685-1713): sample -> argmax -> re-tokenise -> forward model."""
    inv, fwd = make_model("tiny"), make_model("cfg3")
    cond = synth_normal("chain/cond", (4, 12))
    tokens, props = generate_and_validate(inv, fwd, cond, DEV, cond_scale=1.0, timesteps=5, forward_timesteps=4,
                                          X_norm_factor=16.0, noise=NoiseSource(seed=3), forward_noise=NoiseSource(seed=4))
    assert tokens.shape == (4, 32) and props.shape == (4, 12) and props.device.type == "cuda"
    assert torch.isfinite(props).all()
    # the same through the two public calls
    again = predict_properties_from_tokens(fwd, tokens, DEV, timesteps=4, X_norm_factor=16.0, noise=NoiseSource(seed=4))
    assert torch.equal(again, props)


def test_token_chain_matches_the_reference_fixture():
    """SURVEY 8 (f3), second half, PINNED: token ids -> forward input -> forward model -> properties against
    tests/golden/token_chain.npz, which the reference's own reverse_tokenize / predict_properties_from_SMILES
    (generative.py:1069-1078, :404-451) produced over a restated keras tokenizer (tests/golden/make_golden_r6.py)."""
    from moleculediffusiontransformer_amd import tokens_to_forward_input
    g = load_golden("token_chain.npz")
    ids, L, xn, T = to_t(g["ids"]), int(g["max_length"]), float(g["X_norm_factor"]), int(g["timesteps"])
    data = tokens_to_forward_input(ids.to(DEV), L, xn)
    assert torch.equal(data.cpu(), to_t(g["forward_input"]))               # integer / byte work: bit-exact
    fwd = make_model("cfg3")
    init, step = noise_fns("r6_chain_t10", (ids.shape[0], 1, L))
    props = predict_properties_from_tokens(fwd, ids, DEV, cond_scale=1.0, timesteps=T, X_norm_factor=xn,
                                           context_embedding_max_length=12,
                                           noise=NoiseSource(init=init, steps=lambda i: step(i, init)))
    assert props.shape == (5, 12) and (props.cpu() - to_t(g["result"])).abs().max() < 1e-4


def test_handoff_status_of_captured_evaluations_is_reported_as_such(monkeypatch):
    """an evaluation noted inside a caller's stream capture cannot be waited for; the engine remembers the capture and
    the next look (handoff_check(wait=True) or the next plain call) reads the word and says that a replay may be the culprit."""
    m = make_model("cfg1")
    m.kernel_choice = "narrow"                           # the pair-split program: the one with in-launch hand-offs
    g = load_golden("cfg1_unet.npz")
    emb = m._embed(to_t(g["seq"]), DEV)
    x, t = to_t(g["x"]).to(DEV), to_t(g["t"])
    y0 = m.unet(x[:1], t[:1], embedding=emb[:1], embedding_scale=1.0)
    eng = m._engine
    assert eng.xflags is not None and eng.handoff_status() == 0 and not getattr(eng, "_xstat_captured", False)
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: True)
    eng.note_handoff()                                   # what an evaluation inside a capture does: nothing but remember
    monkeypatch.undo()
    assert eng._xstat_captured and not getattr(eng, "_xstat_pending", False)
    eng.handoff_check(wait=True)                         # replays were fine: no error
    y1 = m.unet(x[:1], t[:1], embedding=emb[:1], embedding_scale=1.0)
    assert torch.equal(y0, y1)
    eng.xflags[0] = 1                                    # a poll of some replay ran into its time-out
    with pytest.raises(RuntimeError, match="replay"):
        eng.handoff_check(wait=True)
    assert eng.handoff_status() == 0                     # cleared by the report


def test_doubled_guidance_batch_never_straddles_a_workgroup():
    """max_length=32, channels=64, 6 conditioning tokens: the C = 256 level has 2 tokens per sample, i.e. 16 samples per
    32-row cross-attention workgroup.  B = 16 runs both guidance passes as one doubled batch; B = 24 would put conditional
    and unconditional samples into one workgroup and must take the two-pass form.  Both equal the two-pass result."""
    m = QMDiffusion(max_length=32, pred_dim=16, channels=64, context_embedding_max_length=6, text_embed_dim=64,
                    embed_dim_position=64)
    m.load_state_dict(synth_state_dict([(k, tuple(v.shape)) for k, v in m.state_dict().items()]))
    m = m.to(DEV)
    eng = m.engine(DEV, 6)
    if not eng.has_dual:
        pytest.skip("no doubled-batch program for this configuration")
    assert eng.c.dual_multiple == 16
    seq = synth_normal("dual16/seq", (32, 6))
    run = lambda s: m.sample(s, DEV, cond_scale=3.0, timesteps=4, noise=NoiseSource(seed=21, sample0=0)).cpu()   # noqa: E731
    d32, d16, d24 = run(seq), run(seq[:16]), run(seq[:24])
    prog = eng.programs.pop("eval_dual")
    try:
        two = run(seq)
    finally:
        eng.programs["eval_dual"] = prog
    assert torch.equal(d32, two) and torch.equal(d16, two[:16]) and torch.equal(d24, two[:24])
    eng.reserve(48)
    with pytest.raises(ValueError, match="straddle"):
        eng.eval(dual=True)


def test_torch_ops_namespace_matches_the_class_surface():
    """SURVEY section 8(b), last row: torch.ops.mdt.* (moleculediffusiontransformer_amd/ops.py) over the C ABI.  The prelude,
    one denoise_fn as three ops, one ADPM2 step as ops and the whole loop as ONE op reproduce the reference's golden
    vectors / the class surface bit for bit; errors are RuntimeErrors."""
    import moleculediffusiontransformer_amd.ops as ops
    from moleculediffusiontransformer_amd.diffusion import adpm2_plan
    g = load_golden("tiny_b3_t8_sample.npz")
    m = make_model("tiny")
    seq, T = to_t(g["seq"]), int(g["timesteps"])
    init, step = noise_fns("tiny_b3_t8", tuple(g["out"].shape))
    dev = torch.device(DEV)
    emb = torch.ops.mdt.cond_embed(seq.to(dev), m.fc1.weight.detach(), m.fc1.bias.detach(), m.p_enc_1d.inv_freq.to(dev), 64)
    assert torch.equal(emb, m._embed(seq, DEV))
    eng = m.engine(dev, emb.shape[1], emb.shape[0])
    h = ops.register_engine(eng)
    sigmas, steps = adpm2_plan(T, KarrasSchedule(0.001, 9.0, 3.0), ADPM2Sampler(rho=1), 0.1)
    # the loop, op by op (diffusion.py:502-524 + :798-814)
    x = (float(sigmas[0]) * init).to(dev)
    xin = torch.ops.mdt.precond_in(x, steps[0].w.c_in, eng.c.in_pad)
    for i, s in enumerate(steps):
        pred = torch.ops.mdt.unet_eval(xin, emb, s.w.c_noise, 1.0, h)
        if i == 0:      # denoise_fn == precond_in -> unet_eval -> precond_out
            d = torch.ops.mdt.precond_out(x, pred, s.w.c_skip, s.w.c_out)
            assert torch.equal(d, m.diffusion.diffusion.denoise_fn(x, sigma=sigmas[0], embedding=emb))
        x_mid, xin = torch.ops.mdt.adpm2_mid(x, pred, s.w.c_skip, s.w.c_out, s.sigma, s.dt_mid, s.w_mid.c_in)
        pred = torch.ops.mdt.unet_eval(xin, emb, s.w_mid.c_noise, 1.0, h)
        c_next = steps[i + 1].w.c_in if i + 1 < len(steps) else 0.0
        x, xin = torch.ops.mdt.adpm2_next(x, x_mid, pred, step(i, init).to(dev), s.w_mid.c_skip, s.w_mid.c_out, s.sigma_mid,
                                          s.dt_down, s.sigma_up, c_next, 0, i + 1, 0)
    fused = m.sample(seq, DEV, cond_scale=1.0, timesteps=T, noise=NoiseSource(init=init, steps=lambda i: step(i, init)))
    assert torch.equal(x, fused)
    assert (x.cpu() - to_t(g["out"])).abs().max() < 1e-4
    assert torch.equal(torch.ops.mdt.argmax_tokens(x).long(), torch.argmax(torch.permute(x, (0, 2, 1)), dim=2))
    # the whole loop as one op, explicit draws in the reference's order
    nz = torch.stack([step(i, init) for i in range(T - 1)]).to(dev)
    y, tok = torch.ops.mdt.sample(emb, init.to(dev), nz, sigmas, h, m.pred_dim, 1.0, 0.1, 1.0, False, 0, 0, True)
    assert torch.equal(y, fused) and torch.equal(tok.long(), torch.argmax(torch.permute(y, (0, 2, 1)), dim=2))
    # counter-based noise: the op is what the plain class call runs
    a = m.sample(seq, DEV, cond_scale=1.0, timesteps=T, noise=NoiseSource(seed=11, sample0=5))
    b, _ = torch.ops.mdt.sample(emb, None, None, sigmas, h, m.pred_dim, 1.0, 0.1, 1.0, False, 11, 5, False)
    assert torch.equal(a, b)
    with pytest.raises(RuntimeError):
        torch.ops.mdt.unet_eval(xin[:, :8], emb, 0.0, 1.0, h)           # wrong shape
    with pytest.raises(RuntimeError):
        torch.ops.mdt.precond_in(x.cpu(), 1.0, 16)                       # not a HIP tensor


def test_additive_prelude_with_a_narrower_text_embedding():
    """pos_emb_fourier_add with text_embed_dim (32) < embed_dim_position (64): mdt_cond_embed_add adds the first 32 columns of
    the 64-column encoding, as the reference's x + p_enc_1d(x) does (transformer.py:3456-3470) -- against the
    embedding recorded from the real reference."""
    from conftest import load_golden
    from moleculediffusiontransformer_amd.graphmodel import AnalogDiffusionSparse
    a = load_golden("add_embed_d32.npz")
    m = AnalogDiffusionSparse(max_length=16, channels=32, pred_dim=3, context_embedding_max_length=12, pos_emb_fourier=True,
                              pos_emb_fourier_add=True, text_embed_dim=32, embed_dim_position=64)
    with torch.no_grad():
        m.fc1.weight.copy_(torch.from_numpy(a["fc1_w"]))
        m.fc1.bias.copy_(torch.from_numpy(a["fc1_b"]))
    m = m.to(DEV)
    emb = m._embed(torch.from_numpy(a["seq"]), DEV)
    assert emb.shape == (2, 12, 32) and (emb.cpu() - torch.from_numpy(a["emb"])).abs().max() < 1e-6


def test_dynamic_thresholding_on_the_gpu_path():
    """clip() with dynamic_threshold > 0 (diffusion.py:75-88; VERDICT r3 'envelope holes'): mdt_dyn_scale (per-sample quantile of
    |x_denoised| by an in-LDS sort, torch.quantile's interpolation) feeding the denoise stage of mdt_precond_out / mdt_adpm2_mid /
    mdt_adpm2_next -- against clip() vectors and a 6-step sample recorded from the real reference with
    KDiffusion_mod.dynamic_threshold = 0.9."""
    g = load_golden("dynthr.npz")
    x = to_t(g["x"]).to(DEV)
    B, C, L = x.shape
    pred = torch.zeros(B, L, 16, device=DEV)
    for q in (0.5, 0.9, 0.995, 1.0):        # c_skip = 1, c_out = 0: x_denoised = x
        got = torch.ops.mdt.precond_out(x, pred, 1.0, 0.0, q)
        assert (got.cpu() - to_t(g[f"clip_q{q}"])).abs().max() < 1e-6, q
    m = make_model("tiny")
    assert m.diffusion.diffusion.dynamic_threshold == 0.0
    m.diffusion.diffusion.dynamic_threshold = 0.9
    ref = to_t(g["sample_q0.9_t6"])
    init, step = noise_fns("tiny_dyn_t6", tuple(ref.shape))
    out = m.sample(to_t(g["seq"]), DEV, cond_scale=1.0, timesteps=6, clamp=False,
                   noise=NoiseSource(init=init, steps=lambda i: step(i, init)))
    assert (out.cpu() - ref).abs().max() < 1e-4
    # the per-step path (net -> denoise_fn) applies the same clip
    emb = m._embed(to_t(g["seq"]), DEV)
    d = m.diffusion.diffusion.denoise_fn(init.to(DEV) * 2.5, sigma=torch.tensor(2.5), embedding=emb, embedding_scale=1.0)
    from helpers import oracle_cfg, synth_sd
    from oracle import unet_oracle as O
    want = O.denoise(synth_sd("tiny"), oracle_cfg("tiny"), init * 2.5, torch.tensor(2.5), emb.cpu(), 1.0, dynamic_threshold=0.9)
    assert (d.cpu() - want).abs().max() < 5e-5
    with pytest.raises(ValueError, match="quantile"):
        from moleculediffusiontransformer_amd.generative import KDiffusion_mod
        KDiffusion_mod(net=m.unet, sigma_distribution=None, sigma_data=0.1, dynamic_threshold=1.5)


def test_default_max_length_1024_runs_and_matches_the_oracle():
    """The reference constructors' default max_length = 1024 (generative.py:720-736; VERDICT r3 'envelope holes'): the sampler
    kernels stage a 1024 x 17 float tile (68 KiB, above the 64 KiB default LDS limit of a launch) and the first attention level
    has 256 tokens per sample (k_attn_long: online softmax over key chunks).  A narrow model (channels 32) so that the CPU
    oracle finishes in seconds; 3 timesteps, B = 2, against the oracle on identical noise, plus the cross-attention level with
    a 32-token context (the default context_embedding_max_length)."""
    from helpers import oracle_cfg  # noqa: F401
    from moleculediffusiontransformer_amd.netspec import inverse_unet_config, unet_manifest
    from oracle import unet_oracle as O
    kw = dict(max_length=1024, pred_dim=1, channels=32, context_embedding_max_length=32)
    m = QMDiffusion(text_embed_dim=64, embed_dim_position=64, **kw)
    m.load_state_dict(synth_state_dict([(k, tuple(v.shape)) for k, v in m.state_dict().items()]))
    m = m.to(DEV)
    keys = [("fc1.weight", (64, 1)), ("fc1.bias", (64,)), ("p_enc_1d.inv_freq", (32,))]
    keys += unet_manifest(inverse_unet_config(1, 32, 128, 32), "unet.")
    sd = synth_state_dict(keys)
    cfg = O.inverse_config(1024, 32, 1, 32)
    B, T = 2, 3
    seq = synth_normal("l1024/seq", (B, 32))
    init = synth_normal("l1024/init", (B, 1, 1024))
    nz = [synth_normal(f"l1024/s{i}", (B, 1, 1024)) for i in range(T - 1)]
    want = O.sample(sd, cfg, seq, init, lambda i, x: nz[i], T, 1.0, False)
    out = m.sample(seq, DEV, cond_scale=1.0, timesteps=T, noise=NoiseSource(init=init, steps=lambda i: nz[i]))
    assert out.shape == (B, 1, 1024)
    assert (out.cpu() - want).abs().max() < 1e-4
    from moleculediffusiontransformer_amd import runtime as rt
    ev = m._engine.c.programs["eval"]
    assert any(op.kind == rt.OP_ATTN and op.i[rt.A_T] == 256 for op in ev)          # the long-sequence kernel was on the path
