"""MI355X-native inverse-diffusion sampling path of MoleculeDiffusionTransformer.

Drop-in classes (same names and call signatures as the reference's MoleculeDiffusion package):

    from moleculediffusiontransformer_amd import QMDiffusion, QMDiffusionForward

The sampling hot path (QMDiffusion.sample -> ADPM2 sampler -> 1-D conditional U-Net) runs in
hand-written gfx950 kernels (csrc/, C ABI in include/mdt_hip.h).
"""
from .diffusion import (ADPM2Sampler, DiffusionInpainter, DiffusionSampler, KarrasSchedule,  # noqa: F401
                        LogNormalDistribution, NoiseSource, Sampler)
from .generative import (KDiffusion_mod, QMDiffusion, QMDiffusionForward, XDiffusion_x,  # noqa: F401
                         generate_and_validate, predict_properties_from_tokens, tokens_to_forward_input)
from .graphmodel import AnalogDiffusionFull, AnalogDiffusionSparse  # noqa: F401
from .modules import PositionalEncoding1D, UNetCFG1d  # noqa: F401
from .netspec import UNetConfig, forward_unet_config, inverse_unet_config  # noqa: F401

__version__ = "0.1.0"


def count_parameters(model) -> None:
    """utils.py:16-25 of the reference."""
    total = sum(p.numel() for p in model.parameters())
    trainable = sum(p.numel() for p in model.parameters() if p.requires_grad)
    print("-" * 100)
    print("Total parameters: ", total, " trainable parameters: ", trainable)
    print("-" * 100)
