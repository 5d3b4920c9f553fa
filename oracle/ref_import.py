"""Import the read-only reference (/root/reference) in the build container.

TEST INFRASTRUCTURE ONLY.  Used by tests/golden/make_golden.py to generate the
golden vectors that pin oracle/unet_oracle.py.  The reference never travels to
the GPU box; nothing on the product path, in bench.py or in the -m gpu tests
imports this file.

The reference package imports tensorflow, rdkit, seaborn, torchvision and
torch_geometric at import time (MoleculeDiffusion/__init__.py,
generative.py:16-24, transformer.py:10) although the diffusion sampling path
uses none of them.  None is installed here, so inert module stubs are
registered first.  tqdm.notebook is replaced by a pass-through because the
real one needs ipywidgets (diffusion.py:15, :522).
"""
import importlib.machinery
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "MoleculeDiffusion"))


class _Dummy:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self

    def __getattr__(self, name):
        return _Dummy()


def _stub(name: str) -> types.ModuleType:
    mod = types.ModuleType(name)
    mod.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
    mod.__path__ = []  # behaves as a package

    def _getattr(attr):
        if attr.startswith("__"):
            raise AttributeError(attr)
        return _Dummy

    mod.__getattr__ = _getattr  # type: ignore[attr-defined]
    return mod


_STUBS = [
    "torchvision", "torchvision.transforms",
    "tensorflow", "tensorflow.keras", "tensorflow.keras.preprocessing",
    "tensorflow.keras.preprocessing.text", "tensorflow.keras.preprocessing.sequence",
    "seaborn",
    "rdkit", "rdkit.Chem", "rdkit.Chem.Draw", "rdkit.Chem.Draw.IPythonConsole",
    "rdkit.Chem.rdDepictor", "rdkit.Chem.rdFMCS", "rdkit.Chem.Draw.rdDepictor",
    "rdkit.DataStructs", "rdkit.Chem.AllChem", "rdkit.Chem.Descriptors",
    "torch_geometric", "torch_geometric.nn", "torch_geometric.utils",
    "torch_geometric.data", "torch_geometric.loader",
]


def import_reference():
    """Returns the reference's MoleculeDiffusion package (QMDiffusion, ...)."""
    if not reference_available():
        raise RuntimeError("reference tree not present (expected only in the build container)")
    import torch
    import einops

    # Pre-warm einops' backend cache on real torch types before a fake
    # `tensorflow` appears in sys.modules (otherwise einops probes it).
    einops.rearrange(torch.zeros(2, 2), "a b -> b a")
    einops.rearrange(torch.nn.Parameter(torch.zeros(2, 2)), "a b -> b a")

    for name in _STUBS:
        if name not in sys.modules:
            sys.modules[name] = _stub(name)
            parent, _, child = name.rpartition(".")
            if parent and parent in sys.modules:
                setattr(sys.modules[parent], child, sys.modules[name])

    nb = types.ModuleType("tqdm.notebook")
    nb.__spec__ = importlib.machinery.ModuleSpec("tqdm.notebook", loader=None)
    nb.tqdm = lambda it=None, *a, **k: it
    nb.trange = lambda *a, **k: range(*a)
    import tqdm  # noqa: F401
    sys.modules["tqdm.notebook"] = nb

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import MoleculeDiffusion  # type: ignore

    return MoleculeDiffusion
