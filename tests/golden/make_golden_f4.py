"""Generate the SURVEY section 8 (f4) fixtures from the REAL reference (build container only):

    python tests/golden/make_golden_f4.py

  nb_*      the inverse model as trained in the reference's notebook (Inverse_Diffusion.ipynb:1587-1604: channels=128,
            pred_dim=22, max_length=32 -> levels with 128 / 256 / 512 channels; 90,965,554 parameters)
  sparse_*  AnalogDiffusionSparse(unet_type='cfg') (graphmodel.py:225-390: patch_size 8, num_blocks [2, 2], attentions
            [1, 1], no pre-transformer), max_length=128, pred_dim=3

  full_*    AnalogDiffusionFull(unet_type='cfg', pos_emb_fourier_add=True) (graphmodel.py:391-597: patch_size 4, num_blocks
            [3, 3]), max_length=64, pred_dim=8, channels=64

Same recipe as make_golden.py (whose helpers it reuses): synthetic weights by key name, deterministic noise injected in the
reference's RNG call order; only inputs and outputs are stored.
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (imports the reference)
from moleculediffusiontransformer_amd.synth import synth_normal, synth_state_dict  # noqa: E402


def main():
    torch.set_num_threads(8)
    nb = G.build("inverse", max_length=32, pred_dim=22, channels=128, context_embedding_max_length=12,
                 text_embed_dim=64, embed_dim_position=64)
    assert sum(p.numel() for p in nb.parameters()) == 90965554
    seq = synth_normal("nb/seq", (2, 12))
    G.unet_case("nb", nb, 2, 32, 22, 12, seq)
    G.sample_case("nb_b2_t6", nb, seq, 6, 1.0)
    G.sample_case("nb_b2_t5_cfg2", nb, seq, 5, 2.0)

    from MoleculeDiffusion.graphmodel import AnalogDiffusionSparse  # type: ignore
    sp = AnalogDiffusionSparse(max_length=128, channels=128, pred_dim=3, context_embedding_max_length=12,
                               unet_type="cfg", pos_emb_fourier=True, pos_emb_fourier_add=False, text_embed_dim=64,
                               embed_dim_position=64).eval()
    sp.load_state_dict(synth_state_dict([(k, tuple(v.shape)) for k, v in sp.state_dict().items()]))
    if not hasattr(sp, "pred_dim"):
        sp.pred_dim = 3
    seqs = synth_normal("sparse/seq", (2, 12))
    G.unet_case("sparse", sp, 2, 128, 3, 12, seqs)
    G.sample_case("sparse_b2_t5", sp, seqs, 5, 1.0)
    G.save("sparse_keys.npz", keys=list(sp.state_dict().keys()), nparams=sum(p.numel() for p in sp.parameters()))

    # full_*: AnalogDiffusionFull(unet_type='cfg') (graphmodel.py:391-597: patch_size 4, num_blocks [3, 3]) with
    # pos_emb_fourier_add=True (the positional encoding ADDED to the fc1 features, text_embed_dim == embed_dim_position)
    from MoleculeDiffusion.graphmodel import AnalogDiffusionFull  # type: ignore
    fu = AnalogDiffusionFull(max_length=64, channels=64, pred_dim=8, context_embedding_max_length=12, unet_type="cfg",
                             pos_emb_fourier=True, pos_emb_fourier_add=True, text_embed_dim=64, embed_dim_position=64).eval()
    fu.load_state_dict(synth_state_dict([(k, tuple(v.shape)) for k, v in fu.state_dict().items()]))
    seqf = synth_normal("full/seq", (2, 12))
    G.unet_case("full", fu, 2, 64, 8, 12, seqf)
    G.sample_case("full_b2_t5", fu, seqf, 5, 1.0)
    G.sample_case("full_b2_t4_cfg3", fu, seqf, 4, 3.0)
    G.save("full_keys.npz", keys=list(fu.state_dict().keys()), nparams=sum(p.numel() for p in fu.parameters()))


if __name__ == "__main__":
    main()
