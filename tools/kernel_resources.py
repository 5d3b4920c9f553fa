#!/usr/bin/env python3
"""Per-kernel register / scratch table of one csrc/*.hip translation unit, from hipcc's resource remarks.

    python tools/kernel_resources.py k_tf256 [k_tf128 ...]

Used by tests/test_host_logic.py (scratch limits of the ring kernels) and when tuning: a ring kernel that spills places
scratch loads next to in-flight inline-asm ds_reads.
"""
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "moleculediffusiontransformer_amd", "csrc")
FIELDS = {"sgprs": r"TotalSGPRs: (\d+)", "vgprs": r"VGPRs: (\d+)", "agprs": r"AGPRs: (\d+)",
          "scratch": r"ScratchSize \[bytes/lane\]: (\d+)", "sgpr_spill": r"SGPRs Spill: (\d+)",
          "vgpr_spill": r"VGPRs Spill: (\d+)", "lds": r"LDS Size \[bytes/block\]: (\d+)", "occupancy": r"Occupancy \[waves/SIMD\]: (\d+)"}


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def flags_of(name: str):
    """Exactly the flags build.py compiles this source with."""
    sys.path.insert(0, ROOT)
    from moleculediffusiontransformer_amd import build as b
    for src, extra in b.SOURCES:
        if os.path.splitext(src)[0] == name:
            return src, [*b.COMMON, *extra]
    raise KeyError(name)


def resources(name: str, defs=()):
    """[(demangled-ish kernel name, {field: int})] for every kernel of csrc/<name>.hip."""
    src, flags = flags_of(name)
    r = subprocess.run([hipcc(), *flags, *defs, "-I", os.path.join(ROOT, "include"), "-c", os.path.join(CSRC, src),
                        "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-3000:])
    out = []
    for blk in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
        fn = blk.split()[0]
        vals = {}
        for k, pat in FIELDS.items():
            m = re.search(pat, blk)
            vals[k] = int(m.group(1)) if m else -1
        out.append((fn, vals))
    return out


def _parse_remarks(stderr: str):
    out = []
    for blk in re.split(r"remark: [^\n]*Function Name: ", stderr)[1:]:
        fn = blk.split()[0]
        vals = {}
        for k, pat in FIELDS.items():
            m = re.search(pat, blk)
            vals[k] = int(m.group(1)) if m else -1
        out.append((fn, vals))
    return out


def resources_and_assembly(name: str, defs=()):
    """(resources(name), the device assembly) from ONE device-only hipcc run: what tests/test_host_logic.py's fixture needs for the
    scratch limits and for tools/isa_lint.py (two full compiles per unit before: the CPU suite's longest step)."""
    src, flags = flags_of(name)
    flags = [f for f in flags if f != "-fPIC"]
    r = subprocess.run([hipcc(), *flags, *defs, "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                        "-Rpass-analysis=kernel-resource-usage", os.path.join(CSRC, src), "-o", "-"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-3000:])
    return _parse_remarks(r.stderr), r.stdout


def demangle(fn: str) -> str:
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", fn], capture_output=True, text=True).stdout.strip() or fn
    except Exception:
        return fn


if __name__ == "__main__":
    for name in sys.argv[1:]:
        for fn, v in resources(name):
            print(f"{demangle(fn)[:70]:70s} vgpr {v['vgprs']:3d} agpr {v['agprs']:3d} sgpr {v['sgprs']:3d} scratch {v['scratch']:4d} "
                  f"vspill {v['vgpr_spill']:3d} sspill {v['sgpr_spill']:3d} lds {v['lds']:6d}")
