"""Builds libmdt_hip.so (hipcc, --offload-arch=gfx950) in-tree, next to the sources.

hipcc cross-compiles without a GPU, so this runs in the build container; the built .so is
git-ignored but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
INCLUDE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
# MDT_LIB_TAG (tuning only): a second library next to the shipped one -- libmdt_hip_<tag>.so with its own objects and stamp --
# so that A/B builds (MDT_BUILD_DEFS=...) are made once in the build container and travel to the GPU box side by side.
_TAG = os.environ.get("MDT_LIB_TAG", "")
LIB = os.path.join(CSRC, f"libmdt_hip{'_' + _TAG if _TAG else ''}.so")
STAMP = os.path.join(CSRC, f".build_stamp{'_' + _TAG if _TAG else ''}")

# (source, extra flags).  k_elem keeps the reference's separate fp32 mul/add rounding.
SOURCES = [
    ("k_gemm.hip", []),
    ("k_gemm_bf16x3.hip", []),
    ("k_gemm_b16.hip", []),
    ("k_gemm_as.hip", []),
    # VGPR-form MFMA: keeps the persistent accumulators out of the AGPR shuttle (v_accvgpr_write + s_nop per MFMA)
    ("k_tblock_lw.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]),
    ("k_tblock32.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]),
    ("k_tf128.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]),
    ("k_tf128_f32.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]),      # the same kernels with exact fp32 MFMA products
    ("k_tf256.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]),
    ("k_tf256_f32.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]),
    ("k_rconv.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]),
    ("k_rconv_f32.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]),
    ("k_proj.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]),
    ("k_res256.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]),
    ("k_resblock.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]),
    ("k_norm.hip", []),
    ("k_attn.hip", []),
    ("k_elem.hip", ["-ffp-contract=off"]),
    ("mdt_api.cpp", ["-x", "hip"]),
]
COMMON = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libmdt_hip.so cannot be built")


def _digest() -> str:
    h = hashlib.sha256()
    paths = [os.path.join(CSRC, n) for n in sorted(os.listdir(CSRC))] + [os.path.join(INCLUDE, "mdt_hip.h")]
    for p in paths:
        if p.endswith((".hip", ".cpp", ".h")):
            h.update(os.path.basename(p).encode())
            with open(p, "rb") as f:
                h.update(f.read())
    h.update(repr((SOURCES, COMMON, os.environ.get('MDT_BUILD_DEFS', ''))).encode())
    return h.hexdigest()


def library_path() -> str:
    return LIB


def is_fresh() -> bool:
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return False
    with open(STAMP) as f:
        return f.read().strip() == _digest()


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile every translation unit for gfx950 and link libmdt_hip.so.  Returns its path."""
    if not force and is_fresh():
        return LIB
    # one builder at a time: under torchrun every rank may find the library missing at the same moment
    import fcntl
    with open(os.path.join(CSRC, ".build_lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and is_fresh():          # another process built it while this one waited
                return LIB
            return _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose: bool) -> str:
    hipcc = _hipcc()

    def compile_one(item):
        src, extra = item
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + (f".{_TAG}.o" if _TAG else ".o"))
        defs = os.environ.get("MDT_BUILD_DEFS", "").split()       # tuning builds only, e.g. -DMDT_STAMPS
        cmd = [hipcc, *COMMON, *extra, *defs, "-I", INCLUDE, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(5, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(STAMP, "w") as f:
        f.write(_digest())
    return LIB


if __name__ == "__main__":
    import sys
    print(build_library(force="--force" in sys.argv, verbose=True))
