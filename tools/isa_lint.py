#!/usr/bin/env python3
"""Static check of the ring kernels' ISA: no instruction touches the destination registers of a `ds_read` that may still
be in flight.

Why: the fused kernels issue their weight-fragment `ds_read_b128`s from inline asm, one between the MFMAs, and wait for them
with hand-counted `s_waitcnt lgkmcnt(N)`.  hipcc believes an asm statement's outputs are valid when the statement ends, so it
is free to place a register copy (phi elimination at a loop back-edge), a spill store or a re-use of such a register between
the read and the wait that covers it -- silently wrong data, seen once in tools/experiments.  Tests that
compare results would catch it only if the copy happens to land on a live path; this lint reads the compiler's assembly for
EVERY kernel of a translation unit and checks the property inside every basic block.

Model (gfx950): LDS instructions return in issue order and count in lgkmcnt together with scalar memory reads, which may
return out of order.  Inside every basic block the lint keeps, for each `ds_read` issued in that block, the number of LDS
operations issued after it; `s_waitcnt lgkmcnt(n)` leaves at most n LGKM operations outstanding, hence at most the n youngest
LDS operations: it retires every read with at least n younger LDS operations (n = 0 retires everything; 16 younger LDS
operations retire a read by themselves, the counter has 4 bits).  Any instruction that reads or writes a VGPR of an outstanding read's destination is a
violation (a later `ds_read` into the same registers is not: the later one lands later).  The counting is EXACT inside a
block -- no false positives -- and that is where a phi copy or a spill of a just-issued read sits (the bottom of the block
that issued it).  Reads that stay in flight ACROSS a branch are not followed: the kernels guard both their prefetches and
their waits with correlated run-time flags (has_next / more), which a path-insensitive analysis turns into hundreds of
infeasible hazards; those hand-offs stay covered by the repeated-launch bit-equality tests (tests/test_gpu_ops.py).

Round 4: the same property for VECTOR-MEMORY loads (`global_load_* / buffer_load_* / scratch_load_* / flat_load_*` with a
VGPR destination; LDS-DMA forms have none).  vmcnt counts every vector-memory operation of the wave (loads, stores, LDS-DMA,
atomics) and they retire in issue order, which is also what the kernels' hand-counted `s_waitcnt vmcnt(N)` rely on: a load is
retired by `vmcnt(n)` once at least n vector-memory operations were issued behind it (6-bit counter: 64 younger operations
retire it by themselves).  Compiler-placed loads satisfy this by construction; the check is there for inline-asm loads whose
landing register the compiler believes valid at once (`k_tf256`'s in-launch L2 prefetch was one; it is an LDS-DMA
into a sink now and has no landing register at all).

    python tools/isa_lint.py k_tf256 [k_tf128 ...]       # exit status 1 on any violation
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources as kr  # noqa: E402

VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
LABEL = re.compile(r"^([.\w$]+):")
LGKM = re.compile(r"lgkmcnt\((\d+)\)")
VMCNT = re.compile(r"vmcnt\((\d+)\)")
VMEM_PREFIX = ("global_", "buffer_", "scratch_", "flat_", "tbuffer_")
VCAP = 64
SMEM_PREFIX = ("s_load_", "s_buffer_load_", "s_memtime", "s_memrealtime", "s_scratch_load", "s_atc_probe", "s_dcache")
CAP = 16


def assembly(name: str, defs=()) -> str:
    src, flags = kr.flags_of(name)
    flags = [f for f in flags if f != "-fPIC"]
    r = subprocess.run([kr.hipcc(), *flags, *defs, "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                        os.path.join(kr.CSRC, src), "-o", "-"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-3000:])
    return r.stdout


def vregs(text: str):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def split_kernels(asm: str):
    """{kernel symbol: [instruction / label lines]} from hipcc -S output."""
    kernels, cur, name = {}, None, None
    for line in asm.splitlines():
        s = line.strip()
        m = re.match(r"^(\w+):\s*(;.*)?$", line)
        if m and line.startswith("_Z") and cur is None:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if s.startswith(".Lfunc_end"):
                kernels[name] = cur
                cur = None
                continue
            if not s or s.startswith(";") or (s.startswith(".") and not LABEL.match(s)):
                continue
            cur.append(s.split(";")[0].strip() if not s.startswith(";") else s)
    return kernels


def blocks_of(lines):
    """Basic blocks: list of (label | None, [instructions]); edges by label / fallthrough."""
    blocks, cur, label = [], [], None
    for s in lines:
        m = LABEL.match(s)
        if m:
            if cur or label is not None:
                blocks.append((label, cur))
            label, cur = m.group(1), []
            continue
        if not s:
            continue
        cur.append(s)
        op = s.split()[0]
        if op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            blocks.append((label, cur))
            label, cur = None, []
    if cur or label is not None:
        blocks.append((label, cur))
    return blocks


def lint_kernel(lines):
    """Returns a list of violation strings for one kernel (exact counting inside each basic block, see the module docstring)."""
    out = []
    for label, ins in blocks_of(lines):
        pend = {}            # position of the ds_read -> [dest vgprs, text, younger LGKM ops, younger LDS ops]
        pendv = {}           # position of a vector-memory load -> [dest vgprs, text, younger vector-memory ops]
        smem = False
        for pos, s in enumerate(ins):
            parts = s.split(None, 1)
            op, rest = parts[0], (parts[1] if len(parts) > 1 else "")
            is_read = op.startswith(("ds_read", "ds_load"))
            touched = vregs(rest)
            is_vmem = op.startswith(VMEM_PREFIX)
            is_vload = is_vmem and "_load_" in op and "_lds_" not in op and not re.search(r"\blds\b", rest)
            if pendv and touched:
                for rid in list(pendv):
                    hit = pendv[rid][0] & touched
                    if not hit:
                        continue
                    if is_vload and hit <= vregs(rest.split(",")[0]) and not (pendv[rid][0] & vregs(",".join(rest.split(",")[1:]))):
                        del pendv[rid]         # a later load into the same registers: vector-memory loads return in order
                        continue
                    out.append(f"[{label}] '{s}' touches v{sorted(hit)} of in-flight '{pendv[rid][1]}' "
                               f"({pendv[rid][2]} vector-memory op(s) issued since)")
            if op == "s_waitcnt":
                m = VMCNT.search(rest)
                if m:
                    n = int(m.group(1))
                    pendv = {} if n == 0 else {r: v for r, v in pendv.items() if v[2] < n}
            if is_vmem:
                for v in pendv.values():
                    v[2] += 1
                pendv = {r: v for r, v in pendv.items() if v[2] < VCAP}
                if is_vload:
                    pendv[pos] = [frozenset(vregs(rest.split(",")[0])), s, 0]
            if pend and touched:
                for rid in list(pend):
                    hit = pend[rid][0] & touched
                    if not hit:
                        continue
                    if is_read and hit <= vregs(rest.split(",")[0]) and not (pend[rid][0] & vregs(",".join(rest.split(",")[1:]))):
                        del pend[rid]          # a later read into the same registers: LDS returns in order, the later one wins
                        continue
                    out.append(f"[{label}] '{s}' touches v{sorted(hit)} of in-flight '{pend[rid][1]}' "
                               f"({pend[rid][2]} LGKM op(s) issued since)")
            if op == "s_waitcnt":
                m = LGKM.search(rest)
                if m:
                    n = int(m.group(1))
                    if n == 0:
                        pend, smem = {}, False
                    else:
                        # at most n LGKM operations outstanding => at most n LDS operations, the n youngest (LDS returns in
                        # order; a scalar load that is still out only takes one of the n places)
                        pend = {r: v for r, v in pend.items() if v[3] < n}
                continue
            is_lds = op.startswith("ds_")
            is_smem = op.startswith(SMEM_PREFIX)
            if is_lds or is_smem or op.startswith("s_sendmsg"):
                for v in pend.values():
                    v[2] += 1
                    v[3] += 1 if is_lds else 0
                # lgkmcnt is a 4-bit counter: with 16 LDS operations issued behind a read the counter has wrapped through it,
                # and LDS operations return in order -- the read has landed whatever scalar loads are outstanding
                pend = {r: v for r, v in pend.items() if v[3] < CAP}
                smem = smem or is_smem
                if is_read:
                    pend[pos] = [frozenset(vregs(rest.split(",")[0])), s, 0, 0]
    return sorted(set(out))


def lint_unit(name: str, defs=(), asm=None):
    """{kernel: [violations]} for csrc/<name>.hip; also the number of ds_reads seen (so that an empty parse cannot pass).  asm: the
    unit's device assembly if the caller has it already (kernel_resources.resources_and_assembly)."""
    kernels = split_kernels(asm if asm is not None else assembly(name, defs))
    if not kernels:
        raise RuntimeError(f"no kernels found in the assembly of {name}")
    report, n_reads = {}, 0
    for kn, lines in kernels.items():
        n_reads += sum(1 for s in lines if s.startswith(("ds_read", "ds_load")))      # (vector-memory loads are checked too)
        report[kn] = lint_kernel(lines)
    return report, n_reads


if __name__ == "__main__":
    bad = 0
    for unit in sys.argv[1:]:
        report, n = lint_unit(unit)
        for kn, v in report.items():
            print(f"{unit}: {kr.demangle(kn)[:80]}: {'ok' if not v else str(len(v)) + ' violation(s)'}")
            for line in v[:10]:
                print("    " + line)
            bad += len(v)
        print(f"{unit}: {n} ds_read instructions checked")
    sys.exit(1 if bad else 0)
