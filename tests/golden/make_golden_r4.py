"""Round-4 fixtures from the REAL reference (build container only):

    python tests/golden/make_golden_r4.py          (writes all four files)

  analog_forward.npz   what AnalogDiffusionSparse.forward / AnalogDiffusionFull.forward (graphmodel.py:316-353 / :497-545) hand to
                       self.diffusion -- the training target sliced / padded from the packed `output` rows and the conditioning
                       embedding -- with predict_neighbors on and off (Full is NOT the Sparse recipe)
  add_embed_d32.npz    the additive conditioning prelude with text_embed_dim (32) < embed_dim_position (64): the encoding's first
                       32 columns are added (transformer.py:3456-3470)

  dynthr.npz           clip() with dynamic_threshold > 0 (diffusion.py:75-88) on heavy-tailed inputs, and a 6-step sample of the
                       tiny inverse model with KDiffusion_mod.dynamic_threshold = 0.9 (every class hard-codes 0.0; the attribute is
                       read at denoise time, diffusion.py:814)

  train_loss.npz       QMDiffusion.forward (generative.py:812-833 -> KDiffusion_mod.forward, diffusion.py:820-844) of the tiny
                       inverse model with KDiffusion_mod.dynamic_threshold = 0.0 and 0.9 on fixed sigmas / noise: the training
                       objective applies clip() inside denoise_fn (:814), so it depends on the threshold

Only inputs and outputs are stored.  Every array is a function of named synthetic draws (synth.py), so re-running the script
reproduces the files bit for bit.
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (imports the reference)
from moleculediffusiontransformer_amd.synth import synth_normal  # noqa: E402


class _Rec(torch.nn.Module):
    def forward(self, output, embedding=None):
        self.output, self.embedding = output.detach().clone(), embedding.detach().clone()
        return torch.zeros(())


def main():
    import MoleculeDiffusion.graphmodel as gm  # type: ignore
    from MoleculeDiffusion.graphmodel import AnalogDiffusionFull, AnalogDiffusionSparse  # type: ignore
    # graphmodel.py:320 reads a module global `max_neighbors` that the file itself never defines (the reference's notebooks set
    # it); 5 = the neighbour rows of the packed graphs there, and what moleculediffusiontransformer_amd.graphmodel uses
    gm.max_neighbors = 5
    out = {}
    seq = synth_normal("r4/analog/seq", (2, 12))
    packed = synth_normal("r4/analog/packed", (2, 9, 10))          # node numbers | xyz | 5 neighbour rows, 10 positions
    out["seq"], out["packed"] = seq.numpy(), packed.numpy()
    for pn in (False, True):
        sp = AnalogDiffusionSparse(max_length=16, channels=32, pred_dim=8 if pn else 3, context_embedding_max_length=12,
                                   unet_type="cfg", text_embed_dim=64, embed_dim_position=64, predict_neighbors=pn)
        with torch.no_grad():       # the recorded embedding goes through fc1: synthetic weights, not the constructor's random ones
            sp.fc1.weight.copy_(synth_normal("r4/analog/fc1_w", tuple(sp.fc1.weight.shape)))
            sp.fc1.bias.copy_(synth_normal("r4/analog/fc1_b", tuple(sp.fc1.bias.shape)))
        sp.diffusion = _Rec()
        sp.forward(seq, packed)
        out[f"sparse_pn{int(pn)}_target"] = sp.diffusion.output.numpy()
        out[f"sparse_pn{int(pn)}_emb"] = sp.diffusion.embedding.numpy()
        fu = AnalogDiffusionFull(max_length=16, channels=32, pred_dim=8, context_embedding_max_length=12, unet_type="cfg",
                                 text_embed_dim=64, embed_dim_position=64, predict_neighbors=pn)
        fu.diffusion = _Rec()
        fu.forward(seq, packed)
        out[f"full_pn{int(pn)}_target"] = fu.diffusion.output.numpy()
    G.save("analog_forward.npz", **out)

    m = AnalogDiffusionSparse(max_length=16, channels=32, pred_dim=3, context_embedding_max_length=12, unet_type="cfg",
                              pos_emb_fourier=True, pos_emb_fourier_add=True, text_embed_dim=32, embed_dim_position=64)
    w, b = synth_normal("r4/add/fc1_w", (32, 1)), synth_normal("r4/add/fc1_b", (32,))
    with torch.no_grad():
        m.fc1.weight.copy_(w)
        m.fc1.bias.copy_(b)
    m.diffusion = _Rec()
    m.forward(seq, packed)
    G.save("add_embed_d32.npz", seq=seq.numpy(), fc1_w=w.numpy(), fc1_b=b.numpy(), emb=m.diffusion.embedding.numpy())


def dynthr():
    import MoleculeDiffusion.diffusion as MD  # type: ignore
    out = {}
    x = synth_normal("r4/dyn/x", (3, 16, 64)) * torch.tensor([0.5, 2.0, 6.0]).view(3, 1, 1)
    out["x"] = x.numpy()
    for q in (0.5, 0.9, 0.995, 1.0):
        out[f"clip_q{q}"] = MD.clip(x.clone(), dynamic_threshold=q).numpy()
    m = G.build("inverse", max_length=32, pred_dim=16, channels=16, context_embedding_max_length=12, text_embed_dim=64,
                embed_dim_position=64)
    m.diffusion.diffusion.dynamic_threshold = 0.9
    seq = synth_normal("tiny/seq", (3, 12))
    inj = G.NoiseInjector("tiny_dyn_t6")
    with inj:
        y = m.sample(seq, "cpu", cond_scale=1.0, timesteps=6, clamp=False)
    out["seq"], out["sample_q0.9_t6"] = seq.numpy(), y.detach().numpy()
    G.save("dynthr.npz", **out)


def train_loss():
    out = {}
    m = G.build("inverse", max_length=32, pred_dim=16, channels=16, context_embedding_max_length=12, text_embed_dim=64,
                embed_dim_position=64)
    B = 3
    seq = synth_normal("tiny/seq", (B, 12))
    x0 = synth_normal("r4/train/x0", (B, 16, 32)) * 1.5          # heavy enough that the 0.9 quantile of |x_denoised| exceeds 1
    noise = synth_normal("r4/train/noise", (B, 16, 32))
    sigmas = (-1.2 + 1.2 * synth_normal("r4/train/sig", (B,))).exp()
    kd = m.diffusion.diffusion
    kd.sigma_distribution = lambda num_samples, device: sigmas.clone()       # the draw of diffusion.py:824, fixed
    out.update(seq=seq.numpy(), x0=x0.numpy(), noise=noise.numpy(), sigmas=sigmas.numpy())
    for q in (0.0, 0.9):
        kd.dynamic_threshold = q
        inj = G.NoiseInjector("unused")
        inj._draw = lambda shape: noise.clone()                              # randn_like of diffusion.py:828
        with inj, torch.no_grad():
            out[f"loss_q{q}"] = m(seq, x0).numpy()
    G.save("train_loss.npz", **out)


if __name__ == "__main__":
    only = [a for a in sys.argv[1:] if a in ("main", "dynthr", "train_loss")]
    for name in (only or ["main", "dynthr", "train_loss"]):
        {"main": main, "dynthr": dynthr, "train_loss": train_loss}[name]()
