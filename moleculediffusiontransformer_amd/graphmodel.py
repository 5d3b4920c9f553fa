"""AnalogDiffusionSparse / AnalogDiffusionFull (graphmodel.py:225-390 / :391-597 of the reference): the same inverse sampling
path as QMDiffusion behind a patching U-Net (patch_size 8 / 4, num_blocks [2, 2] / [3, 3], ONE transformer layer per level, no
pre-transformer).  Constructor keywords, attributes, state_dict layout and the sample() signature are the reference's; sample()
runs on the MI355X path, forward() is the training loss on the reference's packed `output` rows
(node numbers | xyz | neighbours), plain PyTorch as for the other classes.
"""
from __future__ import annotations

import torch

from .generative import _QMBase
from .netspec import sparse_unet_config

max_neighbors = 5        # graphmodel.py's module-level constant used by forward() to slice `output`


def pad_sequence(output_xyz: torch.Tensor, max_length: int) -> torch.Tensor:
    """graphmodel.py:220-223: zero-pad (B, C, n) on the right to (B, C, max_length).  As there, n > max_length is an error
    (the slice assignment does not fit); unlike there the result stays on the input's device and dtype (the reference
    allocates torch.zeros on the CPU whatever the input)."""
    B, C, n = output_xyz.shape
    if n > max_length:
        raise RuntimeError(f"The expanded size of the tensor ({max_length}) must match the existing size ({n}) at non-singleton "
                           "dimension 2: sequences longer than max_length cannot be padded (graphmodel.py:222)")
    out = torch.zeros(B, C, max_length, dtype=output_xyz.dtype, device=output_xyz.device)
    out[:, :, :n] = output_xyz
    return out


class _AnalogBase(_QMBase):
    _inverse = True
    _patch, _blocks = 8, (2, 2)

    def __init__(self, max_length=1024, channels=128, pred_dim=1, context_embedding_max_length=32, unet_type="cfg",
                 pos_emb_fourier=True, pos_emb_fourier_add=False, text_embed_dim=1024, embed_dim_position=64,
                 predict_neighbors=False):
        super().__init__()
        self.predict_neighbors = predict_neighbors
        print("Using unet type: ", unet_type)
        self._init_common(max_length, channels, pred_dim, None, context_embedding_max_length, unet_type, pos_emb_fourier,
                          pos_emb_fourier_add, text_embed_dim, embed_dim_position)

    def _unet_config(self, pred_dim, channels, ctx_features, ctx_max_length):
        return sparse_unet_config(pred_dim, channels, ctx_features, ctx_max_length, patch_size=self._patch,
                                  num_blocks=self._blocks)

    def forward(self, sequences, output):
        """AnalogDiffusionSparse.forward (graphmodel.py:316-353): xyz rows (and the max_neighbors neighbour rows with
        predict_neighbors) padded to max_length, then the diffusion loss on the conditioning embedding."""
        from .train import conditioning_embedding
        xyz = pad_sequence(output[:, 1:4, :], self.max_length)
        if self.predict_neighbors:
            xyz = torch.cat((xyz, pad_sequence(output[:, 4:4 + max_neighbors, :], self.max_length)), 1)
        return self.diffusion(xyz, embedding=conditioning_embedding(self, sequences))

    def sample(self, sequences, device, cond_scale=7.5, timesteps=100, clamp=False, *, noise=None, trace=None, timer=None):
        return self._do_sample(sequences, device, cond_scale, timesteps, clamp, noise, trace, timer)


class AnalogDiffusionSparse(_AnalogBase):
    """graphmodel.py:225-390."""
    _patch, _blocks = 8, (2, 2)


class AnalogDiffusionFull(_AnalogBase):
    """graphmodel.py:391-597 (predict_neighbors defaults to True there)."""
    _patch, _blocks = 4, (3, 3)

    def __init__(self, max_length=1024, channels=128, pred_dim=1, context_embedding_max_length=32, unet_type="cfg",
                 pos_emb_fourier=True, pos_emb_fourier_add=False, text_embed_dim=1024, embed_dim_position=64,
                 predict_neighbors=True):
        super().__init__(max_length, channels, pred_dim, context_embedding_max_length, unet_type, pos_emb_fourier,
                         pos_emb_fourier_add, text_embed_dim, embed_dim_position, predict_neighbors)

    def forward(self, sequences, output):
        """AnalogDiffusionFull.forward (graphmodel.py:497-545) -- NOT the Sparse recipe: no padding; the neighbour rows are
        output[:, 4:4 + max_length, :] (max_length rows, not max_neighbors); with predict_neighbors the target is
        cat(xyz rows, neighbour rows), without it the packed `output` goes to the diffusion as it is (node-number row
        included), exactly as the reference does."""
        from .train import conditioning_embedding
        if self.predict_neighbors:
            output = torch.cat((output[:, 1:4, :], output[:, 4:4 + self.max_length, :]), 1)
        return self.diffusion(output, embedding=conditioning_embedding(self, sequences))
