"""Shared builders for tests: model configs, synthetic state_dicts, deterministic noise."""
import functools

import torch

from moleculediffusiontransformer_amd.netspec import (forward_unet_config, inverse_unet_config, sparse_unet_config,
                                                      unet_manifest)
from moleculediffusiontransformer_amd.synth import MODEL_CASES, synth_normal, synth_state_dict
from oracle import unet_oracle as O

# name -> (kind, wrapper kwargs)
CASES = MODEL_CASES


def oracle_cfg(case):
    kind, kw = CASES[case]
    f = {"inverse": O.inverse_config, "forward": O.forward_config, "sparse": O.sparse_config, "full": O.full_config}[kind]
    return f(kw["max_length"], kw["channels"], kw["pred_dim"], kw["context_embedding_max_length"])


@functools.lru_cache(maxsize=None)
def synth_sd(case):
    """Reference-format state_dict (canonical 'unet.' prefix + fc1 + p_enc_1d) with synthetic weights."""
    kind, kw = CASES[case]
    if kind == "full":
        ucfg = sparse_unet_config(kw["pred_dim"], kw["channels"], 64, kw["context_embedding_max_length"], patch_size=4, num_blocks=(3, 3))
    else:
        mk = {"inverse": inverse_unet_config, "forward": forward_unet_config, "sparse": sparse_unet_config}[kind]
        ucfg = mk(kw["pred_dim"], kw["channels"], 128, kw["context_embedding_max_length"])
    keys = [("fc1.weight", (64, 1)), ("fc1.bias", (64,)), ("p_enc_1d.inv_freq", (32,))]
    keys += unet_manifest(ucfg, "unet.")
    return synth_state_dict(keys)


def noise_fns(tag, shape):
    """(init_noise, step_noise(i, x)) reproducing tests/golden/make_golden.py's NoiseInjector order."""
    init = synth_normal(f"{tag}/draw0", shape)
    return init, (lambda i, x: synth_normal(f"{tag}/draw{i + 1}", tuple(x.shape)))


def to_t(a):
    return torch.from_numpy(a)
