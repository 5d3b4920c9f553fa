"""Folds the rocprofv3 outputs of tools/profile_round.sh into the small csv summaries kept under profiles/.

usage: python tools/pmc_summary.py gpurun_out/prof_r1 r1 [destination directory, default profiles/]

FETCH_SIZE is reported in KB and, on gfx950, counts exactly half of the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM section) -> corrected = 2 x raw.  WRITE_SIZE (KB) is exact.
"""
import collections
import csv
import glob
import os
import shutil
import sys


def short(name: str) -> str:
    return name.split("(")[0].replace("void ", "").strip()


def counters(d: str, counter: str):
    tot, n = collections.Counter(), collections.Counter()
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            tot[k] += float(r["Counter_Value"])
            n[k] += 1
    return tot, n


def main():
    src, tag = sys.argv[1], sys.argv[2]
    dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    os.makedirs(dst, exist_ok=True)
    what = open(os.path.join(src, "args.txt")).read().strip() if os.path.exists(os.path.join(src, "args.txt")) else \
        "bench.py --batch 1024 --timesteps 4"
    # stamp: the digest of the kernel sources the profiled library was built from (moleculediffusiontransformer_amd/build.py)
    # and the commit, so that bench.py can refuse a summary of other code
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from moleculediffusiontransformer_amd import build as _b
    head = os.environ.get("MDT_GIT_HEAD", "")          # no .git on the GPU box: tools/profile_round.sh callers pass it in
    stamp = f"# csrc_digest: {_b._digest()}" + (f"  git: {head}" if head else "") + "\n"
    for kind in ("kernel_stats", "domain_stats"):
        fs = glob.glob(os.path.join(src, "trace", "**", f"*_{kind}.csv"), recursive=True)
        if fs:
            shutil.copy(fs[0], os.path.join(dst, f"{tag}_{kind}.csv"))
    for a, b in (("bench.json", f"{tag}_bench.json"), ("bench_under_rocprof.json", f"{tag}_bench_under_rocprof.json")):
        if os.path.exists(os.path.join(src, a)):
            shutil.copy(os.path.join(src, a), os.path.join(dst, b))
    fr, fn = counters(os.path.join(src, "pmc_fetch"), "FETCH_SIZE")
    wr, wn = counters(os.path.join(src, "pmc_write"), "WRITE_SIZE")
    with open(os.path.join(dst, f"{tag}_pmc_hbm_traffic.csv"), "w") as f:
        f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), " + what + "\n")
        f.write(stamp)
        f.write("# FETCH_SIZE is in KB and reads exactly 1/2 of the bytes of wide coalesced reads on gfx950 "
                "(MI355X_MICROARCH.md, HBM): corrected = 2 x raw\n")
        f.write("kernel,dispatches,fetch_raw_KB_per_dispatch,fetch_corrected_MB_per_dispatch,"
                "write_MB_per_dispatch,hbm_MB_per_dispatch\n")
        rows = []
        for k in fr:
            d = fn[k]
            raw = fr[k] / d
            w = (wr.get(k, 0.0) / max(wn.get(k, 1), 1)) / 1024.0
            rows.append((k, d, raw, 2 * raw / 1024.0, w))
        rows.sort(key=lambda r: -(r[3] + r[4]) * r[1])
        for k, d, raw, fc, w in rows:
            f.write(f"\"{k}\",{d},{raw:.1f},{fc:.2f},{w:.2f},{fc + w:.2f}\n")
    # SQ pass: where the waves spend their cycles, MFMA pipe occupancy, LDS bank conflicts (per kernel, per dispatch)
    names = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES",
             "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE"]
    vals = {n: counters(os.path.join(src, "pmc_sq"), n) for n in names}
    if vals["SQ_WAVE_CYCLES"][0]:
        with open(os.path.join(dst, f"{tag}_pmc_sq.csv"), "w") as f:
            f.write("# rocprofv3 --pmc " + " ".join(names) + " (one pass, --kernel-trace only), " + what + "\n")
            f.write(stamp)
            f.write("# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves; WAIT_ANY = parked on "
                    "s_waitcnt / s_barrier, WAIT_INST_ANY = issue stall, ACTIVE_INST_ANY = issuing; MFMA busy in cycles; "
                    "GRBM_GUI_ACTIVE summed over the 8 XCDs (per-launch cycles = value / 8)\n")
            f.write("# mfma_util = MFMA_BUSY_CYCLES / (kernel duration of the same pass x 2.4 GHz nominal x 256 CUs x 4 SIMDs); "
                    "fractions are over ALL waves of a workgroup: the ring kernels' 4 loader waves (of 8) are parked by design\n")
            f.write("# lds_active = SQ_LDS_IDX_ACTIVE / (duration x 2.4 GHz x 256 CUs): share of the time the CUs' LDS arrays are "
                    "busy (fragment ds_reads + LDS-DMA writes; 256 B/clk/CU for ds_read_b128 on gfx950 -- far from a limit, DESIGN 3.5)\n")
            f.write("kernel,dispatches,parked_frac,issue_stall_frac,issuing_frac,mfma_busy_cycles_per_dispatch,"
                    "avg_duration_us,mfma_util,lds_conflict_frac,lds_active\n")
            dur, dn = collections.Counter(), collections.Counter()
            for tf in glob.glob(os.path.join(src, "pmc_sq", "**", "*_kernel_trace.csv"), recursive=True):
                for r in csv.DictReader(open(tf)):
                    k = short(r["Kernel_Name"])
                    dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                    dn[k] += 1
            tot, n = vals["SQ_WAVE_CYCLES"]
            rows = []
            for k in tot:
                wc = tot[k]
                if wc <= 0:
                    continue
                d = n[k]
                g = lambda name: vals[name][0].get(k, 0.0)
                us = dur[k] / max(dn[k], 1) / 1e3
                gui = us * 2400.0                          # cycles at the nominal clock
                mf = g("SQ_VALU_MFMA_BUSY_CYCLES") / d
                lds = g("SQ_LDS_IDX_ACTIVE")
                rows.append((wc, f"\"{k}\",{d},{g('SQ_WAIT_ANY') / wc:.3f},{g('SQ_WAIT_INST_ANY') / wc:.3f},"
                                 f"{g('SQ_ACTIVE_INST_ANY') / wc:.3f},{mf:.0f},{us:.1f},"
                                 f"{mf / (gui * 1024) if gui else 0:.4f},{g('SQ_LDS_BANK_CONFLICT') / lds if lds else 0:.4f},"
                                 f"{lds / d / (gui * 256) if gui else 0:.4f}\n"))
            for _, line in sorted(rows, key=lambda r: -r[0]):
                f.write(line)
    print("profiles written for", tag)


if __name__ == "__main__":
    main()
