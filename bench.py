#!/usr/bin/env python3
"""Headline benchmark: molecules/sec at 64 diffusion steps (QM9-shaped inverse model) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

A "step" is one full QMDiffusion.sample() call over one batch: BASELINE.json configs[1] = inverse model
channels=64, pred_dim=16, max_len=64, cond_len=12, batch 1024 per GPU, 64 timesteps (126 U-Net
evaluations + 63 ADPM2 updates), fp32, cond_scale=1.0, synthetic weights/conditioning, inputs resident in
HBM, on-device counter-based noise.  `--batch 8192` is the per-GPU shard of configs[3] (65,536 over 8 GPUs).

N > 1: one rank per GPU over RCCL.  Either the driver launches the ranks itself (torch.distributed.run sets
RANK / WORLD_SIZE) or this script does: `python bench.py --gpus N` without RANK in the environment starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process before anything touches a
GPU, relays rank 0's JSON line and exits with the child's return code.  Every rank samples its own batch (weak
scaling, no collective in the step loop) and the generated samples are all-gathered once per call over RCCL.

Prints ONE JSON line (rank 0) with the contract fields plus
  "roofline":     the dominant kernel class of one U-Net evaluation, HIP-event timed per launch
  "exact_f32":    the same workload and the same fused program with exact fp32 MFMA products (MDT_GEMM=f32), 5 steps, with
                  its own roofline block against the fp32 MFMA peak                                      (N = 1)
  "cpu_baseline": the CPU oracle (PyTorch restatement pinned to the reference) on a bounded sample       (N = 1)
  "multi_gpu":    ranks seen by the all-gather and the bitwise shard-invariance check                    (N > 1)
"""
import argparse
import contextlib
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA (not the 2:1-sparse headline)
HBM_PEAK_GBS = 8000.0
REF_FLOPS_PER_SAMPLE_EVAL = 455.3e6   # reference op graph (incl. per-eval cross-attn K/V + time mapping)
REF_OPGRAPH_BYTES_PER_SAMPLE_EVAL = 13.21e6

WORKLOADS = {
    "cfg1": "QMDiffusion inverse sample(): channels=64 pred_dim=16 max_len=64 cond_len=12, ",
    "cfg3": "QMDiffusionForward sample(): channels=64 pred_dim=1 max_len=64 cond_len=64, ",
    "cfg5": "QMDiffusion inverse sample(): channels=256 pred_dim=32 max_len=128 cond_len=12, ",
    "nb": "QMDiffusion inverse sample() of the reference notebook's model: channels=128 pred_dim=22 max_len=32 cond_len=12, ",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1024, help="molecules per GPU per step (8192 = configs[3]'s shard)")
    ap.add_argument("--timesteps", type=int, default=64)
    ap.add_argument("--workload", default="cfg1", choices=sorted(WORKLOADS),
                    help="cfg1 = BASELINE configs[1] (the headline metric, default); cfg3 = QMDiffusionForward "
                         "(configs[2]: --batch 4096 --timesteps 100); cfg5 = deep U-Net architecture of configs[4] "
                         "(channels 256, max_len 128).  Other workloads are informational: no CPU baseline / parity leg")
    ap.add_argument("--cond-scale", type=float, default=1.0, help="classifier-free guidance scale (2 U-Net passes if != 1)")
    ap.add_argument("--gemm-mode", default=None, choices=("bf16x3", "f32", "bf16"),
                    help="GEMM products: bf16x3 = split-bf16, fp32-class (default); f32 = exact fp32 MFMA; bf16 = plain bf16 "
                         "products, the reduced-precision mode BASELINE configs[4] names (layer-by-layer GEMM program)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true")
    ap.add_argument("--no-exact-f32", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--master-port", type=int, default=29511, help="rendezvous port when this script launches its own ranks")
    return ap.parse_args()


def self_launch(a) -> int:
    """--gpus N > 1 without a rank environment: start the ranks as a child torch.distributed.run (this process has not
    imported torch or touched a GPU), stream the child's output through, return its exit code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(a.master_port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    child = subprocess.Popen(cmd, env=env, cwd=ROOT)
    try:
        return child.wait()
    except KeyboardInterrupt:
        child.terminate()
        return child.wait()


def op_flops(op, rt, B):
    """Algorithmic FLOPs (2*MAC, fp32-equivalent) of one op of the eval program for batch B."""
    i = op.i
    if op.kind == rt.OP_GEMM:
        return 2.0 * B * i[rt.G_R_OUT] * i[rt.G_N] * i[rt.G_TAPS] * i[rt.G_CIN] * max(i[rt.G_PHASES], 1)
    if op.kind == rt.OP_ATTN:
        return 4.0 * B * i[rt.A_T] * i[rt.A_TK] * 64 * i[rt.A_HEADS]
    if op.kind == rt.OP_RCONV:
        return 2.0 * B * i[rt.R_T] * i[rt.R_C] * i[rt.R_C] * i[rt.R_TAPS] * (2 if op.a2.space else 1)
    if op.kind == rt.OP_RESBLOCK:
        cin, cout = i[rt.K_CIN], i[rt.K_COUT]
        return 2.0 * B * i[rt.K_T] * (3 * cin * cout + 3 * cout * cout + cin * cout)
    if op.kind == rt.OP_TBLOCK:
        c, t, nch, tk = i[rt.B_C], i[rt.B_T], i[rt.B_NCHUNK], i[rt.B_TK]
        mid = 64 * nch
        post = 2.0 * B * t * c * c if i[rt.B_POST] else 0.0
        if i[rt.B_MODE] == rt.TB_FF:
            return 4.0 * B * t * c * mid + post
        if i[rt.B_MODE] == rt.TB_SELF:
            return B * (2.0 * t * c * 3 * mid + 4.0 * t * t * mid + 2.0 * t * mid * c)
        return B * (2.0 * t * c * mid + 4.0 * t * tk * mid + 2.0 * t * mid * c)
    if op.kind in (rt.OP_TF128, rt.OP_TF256):       # a whole Transformer1d: to_in + blocks (+ folded to_out)
        c, t, tk = i[rt.F_C], i[rt.F_T], i[rt.F_TK]
        mid, hid = 64 * i[rt.F_HEADS], 64 * i[rt.F_NFF]
        blk = (2.0 * t * c * 3 * mid + 4.0 * t * t * mid + 2.0 * t * mid * c) + 4.0 * t * c * hid
        if i[rt.F_CROSS]:
            blk += 2.0 * t * c * mid + 4.0 * t * tk * mid + 2.0 * t * mid * c
        return B * (i[rt.F_NBLOCKS] * blk + (2.0 * t * c * c if i[rt.F_HAS_IN] else 0.0) + (2.0 * t * c * c if i[rt.F_NPOST] else 0.0))
    if op.kind == rt.OP_RES256:                       # a chain of ResnetBlock1d blocks at C = 256 (NPOST = live taps of the k = 3 convolutions)
        c, t, taps, n = 256, i[rt.F_T], i[rt.F_NPOST], i[rt.F_N_RES]
        per_block = 2 * taps * c * c if i[rt.F_RES_KIND] == 1 else (taps * 2 * c * c + taps * c * c + 2 * c * c)
        return 2.0 * B * t * n * per_block
    return 0.0


# the fused transformer kernels (one sub-block per launch: k_tblock32 / k_tblock_lw; a whole Transformer1d per launch:
# k_tf128 / k_tf256) are ONE kernel class for the roofline: same operator, same MFMA / LDS-ring structure
KERNEL_CLASS = {7: "k_tblock", 11: "k_tblock", 12: "k_tblock"}


def kernel_breakdown(eng, rt, B):
    """HIP-event time of every op of ONE U-Net evaluation (plain launches, one interval per launch),
    grouped by kernel class.  Returns dict class -> (launches, total_ms, flops)."""
    prog = eng.programs["eval"]
    ops = eng.c.programs["eval"]
    bind = eng._bind(xin=eng.xin, out=eng.pred)
    best = None
    for rep in range(3):
        timer = rt.EventTimer(len(ops))
        for i in range(len(ops)):
            timer.start()
            prog.run(bind, B, 0, i, 1)
            timer.stop()
        ms = timer.collect()
        if best is None or sum(ms) < sum(best):
            best = ms
    out = {}
    for op, t in zip(ops, best):
        k = KERNEL_CLASS.get(op.kind) or rt.OP_NAMES.get(op.kind, f"kind{op.kind}")
        n, tot, fl = out.get(k, (0, 0.0, 0.0))
        out[k] = (n + 1, tot + t, fl + op_flops(op, rt, B))
    return out


def pmc_summary(variant=""):
    """Rows of the newest committed PMC traffic summary (profiles/r*_pmc_hbm_traffic.csv: FETCH_SIZE x2 gfx950
    correction + WRITE_SIZE, separate rocprofv3 passes of THIS workload at batch 1024): (kernel name, launches in the
    profiled run, MB per launch)."""
    import glob
    import re
    # the headline workload's summary of the newest round: profiles/r<N>_pmc_hbm_traffic.csv (NOT r<N>_<other config>_pmc_...)
    # variant "f32": r<N>_f32_pmc_hbm_traffic.csv, the same workload in the exact-fp32 mode
    mid = (variant + "_") if variant else ""
    files = [f for f in glob.glob(os.path.join(ROOT, "profiles", f"r*_{mid}pmc_hbm_traffic.csv"))
             if re.fullmatch(rf"r\d+_{mid}pmc_hbm_traffic\.csv", os.path.basename(f))]
    files.sort(key=lambda f: int(re.match(r"r(\d+)_", os.path.basename(f)).group(1)))
    if not files:
        return None, []
    rows = []
    digest = None
    for line in open(files[-1]):
        if line.startswith("# csrc_digest:"):
            digest = line.split()[2]
        if line.startswith("#") or line.startswith("kernel,"):
            continue
        name, rest = line.rsplit(",", 5)[0].strip('"'), line.strip().rsplit(",", 5)[1:]
        rows.append((name, int(rest[0]), float(rest[4])))
    PMC_DIGEST[os.path.basename(files[-1])] = digest
    return os.path.basename(files[-1]), rows


PMC_DIGEST = {}


def pmc_is_current(fname):
    """(bool, reason): does the committed PMC summary stem from THIS library's kernel sources?  tools/pmc_summary.py stamps
    every summary with build._digest() of the sources it profiled; a summary of other code is not quoted."""
    from moleculediffusiontransformer_amd import build as b
    want, have = b._digest(), PMC_DIGEST.get(fname)
    if have is None:
        return False, f"profiles/{fname} carries no csrc_digest stamp (recorded before round 6): not quoted"
    if have != want:
        return False, f"profiles/{fname} was recorded on other kernel sources (csrc digest {have[:12]}, this library {want[:12]}): not quoted"
    return True, ""


def pmc_traffic(kernel_class, variant=""):
    """HBM bytes per launch of one kernel class from the committed PMC summary, or (None, None)."""
    fname, rows = pmc_summary(variant)
    prefixes = ("mdt::k_tblock", "mdt::k_tf128", "mdt::k_tf256") if kernel_class == "k_tblock" else ("mdt::" + kernel_class,)
    n = sum(r[1] for r in rows if r[0].startswith(prefixes))
    if not n:
        return None, "no PMC summary of this kernel class under profiles/"
    ok, why = pmc_is_current(fname)
    if not ok:
        return None, why
    mb = sum(r[1] * r[2] for r in rows if r[0].startswith(prefixes))
    return round(mb / n * 1e6), ("bytes per launch; profiles/" + fname + " (rocprofv3 PMC passes of this workload on THESE kernel "
                                 "sources -- csrc digest checked --, recorded earlier, not collected live)")


def cpu_baseline_leg(torch, model, device, evals):
    """the reference path restated on PyTorch CPU ops (oracle/unet_oracle.py, pinned bit-exact to the
    reference by tests/golden) at B = 4 and B = 256, fp32, no_grad; thread count picked by a short sweep; one warm-up +
    median of 3; bounded to a few timesteps and scaled to the 126 evaluations of a 64-step call.  Also the checker of this
    run: the same bounded B = 256 sample goes through the HIP path on identical noise."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_cfg, synth_sd
    from oracle import unet_oracle as O
    from moleculediffusiontransformer_amd import NoiseSource
    from moleculediffusiontransformer_amd.synth import synth_normal
    sd, cfg = synth_sd("cfg1"), oracle_cfg("cfg1")

    def run(cb, ct, tag):
        cseq = synth_normal(f"bench/cpu/seq{tag}", (cb, 12))
        init = synth_normal(f"bench/cpu/init{tag}", (cb, 16, 64))
        nz = [synth_normal(f"bench/cpu/step{tag}/{i}", (cb, 16, 64)) for i in range(ct - 1)]
        c0 = time.perf_counter()
        out = O.sample(sd, cfg, cseq, init, lambda i, x: nz[i], ct, 1.0, False)
        return time.perf_counter() - c0, (cseq, init, nz, out)

    ncpu = os.cpu_count() or 1
    cands = sorted({n for n in (4, 8, 16, 32, 64) if n <= ncpu})
    sweep = {}
    for n in cands:                                   # sweep to the END (4 .. 64 threads): one 2-timestep call (2 evaluations) at
        torch.set_num_threads(n)                      # B = 256, warm-up + one timed; the fastest point is the baseline (more than
        run(256, 2, "sweep")                          # 64 threads only oversubscribe these small GEMMs: 256 threads took 370 s)
        sweep[n] = round(run(256, 2, "sweep")[0], 3)
    threads = min(sweep, key=sweep.get)
    torch.set_num_threads(threads)
    points = {}
    keep = None
    for cb, ct in ((4, 12), (256, 4)):
        run(cb, 2, "warm")
        times = []
        for rep in range(3):
            dt, data = run(cb, ct, "")
            times.append(dt)
            keep = data
        med = sorted(times)[1]
        per_eval = med / (2 * (ct - 1))
        points[cb] = {"timesteps_timed": ct, "evals_timed": 2 * (ct - 1), "median_s": round(med, 3),
                      "ms_per_eval": round(1e3 * per_eval, 2), "molecules_per_s_at_64_steps": round(cb / (per_eval * evals), 3)}
    cseq, init, nz, cpu_out = keep                    # the B = 256 sample, last repetition
    hip_out = model.sample(cseq, device, cond_scale=1.0, timesteps=len(nz) + 1, clamp=False,
                           noise=NoiseSource(init=init, steps=lambda i: nz[i])).cpu()
    parity = {"max_abs_vs_cpu_reference_path": float((hip_out - cpu_out).abs().max()), "tolerance": 1e-4,
              "sample": f"batch 256, {len(nz) + 1} timesteps, identical noise"}
    base = {"value": points[256]["molecules_per_s_at_64_steps"], "unit": "molecules/s", "cores": threads, "kind": "port",
            "sample": "oracle/unet_oracle.py (PyTorch CPU fp32, bit-exact to the reference on the golden vectors): batch 256, "
                      f"{points[256]['timesteps_timed']} timesteps = {points[256]['evals_timed']} U-Net evals, warm-up + median of 3, "
                      f"scaled to {evals} evals; batch 4 (BASELINE configs[0]) in `points`",
            "host_cpus": ncpu, "thread_sweep_s_per_2_evals_b256": sweep, "points": points}
    return base, parity


class PowerSampler(threading.Thread):
    """Socket power and the SMU's shader-clock reading of THIS rank's GPU during the timed region (amdgpu hwmon sysfs,
    world-readable: a file read every 50 ms on a host thread).  Context for the roofline, not a metric: the fused kernels
    run the package at ~1.2 kW of its 1.4 kW cap, and in-kernel clock stamps (profiles/r2_launch_timeline.txt)
    show the shader clock the waves actually see dropping to 1.8-2.0 GHz in the streamed phases."""

    def __init__(self, device_index):
        super().__init__(daemon=True)
        self.dir, self.stop_flag, self.w, self.mhz = None, threading.Event(), [], []
        try:
            import glob
            import torch
            pr = torch.cuda.get_device_properties(device_index)
            bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            for d in glob.glob("/sys/class/drm/card*/device"):
                if os.path.basename(os.path.realpath(d)) == bdf:
                    hw = glob.glob(os.path.join(d, "hwmon", "hwmon*"))
                    if hw and os.path.exists(os.path.join(hw[0], "power1_input")):
                        self.dir = hw[0]
        except Exception:
            self.dir = None

    def _read(self, name):
        with open(os.path.join(self.dir, name)) as f:
            return float(f.read().strip())

    def run(self):
        while self.dir and not self.stop_flag.is_set():
            try:
                self.w.append(self._read("power1_input") * 1e-6)
                self.mhz.append(self._read("freq1_input") * 1e-6)
            except Exception:
                break
            self.stop_flag.wait(0.05)

    def summary(self):
        self.stop_flag.set()
        if not self.w:
            return None
        cap = None
        try:
            cap = self._read("power1_cap") * 1e-6
        except Exception:
            pass
        return {"socket_w_avg": round(sum(self.w) / len(self.w), 1), "socket_w_max": round(max(self.w), 1), "cap_w": cap,
                "sclk_mhz_smu_avg": round(sum(self.mhz) / len(self.mhz), 0), "samples": len(self.w),
                "source": "amdgpu hwmon power1_input / freq1_input of this GPU, sampled every 50 ms over the timed steps"}


def main():
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(self_launch(a))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the sampling path has no CPU fallback")
    # test hook (one-GPU boxes): MDT_BENCH_SHARE_GPU=1 puts every rank on cuda:0 over gloo to exercise the N > 1 logic
    share = os.environ.get("MDT_BENCH_SHARE_GPU", "0") == "1"
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # a rank environment (torch.distributed.run, or RANK=0 WORLD_SIZE=1 by hand) brings the process group up even for one
    # rank: the RCCL all-gather of the samples then runs at N = 1 too (smoke test of the collective path on a one-GPU box)
    grouped = world > 1 or "RANK" in os.environ
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(a.master_port))
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from moleculediffusiontransformer_amd import NoiseSource, runtime as rt
    from moleculediffusiontransformer_amd.distributed import all_gather_samples
    from moleculediffusiontransformer_amd.synth import make_synth_model, synth_normal

    with contextlib.redirect_stdout(sys.stderr):     # the class prints "Using unet type" like the reference does
        model = make_synth_model(a.workload, device)  # synthetic weights (no network for checkpoints)
    if a.gemm_mode:
        model.gemm_mode = a.gemm_mode
    B, T = a.batch, a.timesteps
    n_cond = model.unet.config.ctx_max_length
    seq_of = lambda r: synth_normal(f"bench/seq/rank{r}", (B, n_cond))     # noqa: E731
    seq = seq_of(rank).to(device)
    if a.workload != "cfg1" or a.cond_scale != 1.0:
        a.no_cpu_baseline = a.no_exact_f32 = a.no_other_configs = True
    evals = 2 * (T - 1)
    eval_timer = rt.EventTimer(evals * (a.steps + a.warmup) + 8)

    phase_ev = []          # N > 1: (sample start, sample end = gather start, gather end) HIP events of every timed step, this rank

    def one_step(step_idx, timed):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if (grouped and timed) else None
        if ev:
            ev[0].record()
        out = model.sample(seq, device, cond_scale=a.cond_scale, timesteps=T, clamp=False,
                           noise=NoiseSource(seed=1234 + step_idx, sample0=rank * B),
                           timer=eval_timer if timed else None)
        if grouped:
            if ev:
                ev[1].record()
            out = all_gather_samples(out, world * B, force_collective=True)
            if ev:
                ev[2].record()
                phase_ev.append(ev)
        return out

    for w in range(a.warmup):
        one_step(w, False)

    def fence():
        torch.cuda.synchronize(device)
        if grouped:
            dist.barrier()
            torch.cuda.synchronize(device)

    fence()
    sampler = PowerSampler(device.index if device.index is not None else 0) if rank == 0 else None
    if sampler:
        sampler.start()
    t0 = time.perf_counter()
    for k in range(a.steps):
        out = one_step(a.warmup + k, True)
    fence()
    elapsed = time.perf_counter() - t0
    power = sampler.summary() if sampler else None
    per_rank = None
    if grouped:
        # per rank: wall time of its K steps, GPU time inside sample() (compute, no collective) and inside the all-gather (which
        # includes waiting for the slowest rank), gathered to every rank (a K-independent 3 x N table; outside the timed region)
        smp = sum(e[0].elapsed_time(e[1]) for e in phase_ev) * 1e-3
        gth = sum(e[1].elapsed_time(e[2]) for e in phase_ev) * 1e-3
        mine = torch.tensor([elapsed, smp, gth], dtype=torch.float64, device="cpu" if share else device)
        table = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(table, mine)
        per_rank = [[float(v) for v in t_] for t_ in table]
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(out).all()

    eval_ms = eval_timer.collect()
    result = None
    if rank == 0:
        eng = model._engine                               # the engine the timed calls ran on (kernel choice by batch)
        mols = world * B * a.steps
        value = mols / elapsed
        flops_exec = eng.c.flops_per_sample_eval           # executed per sample per eval (K/V + time mapping hoisted)
        avg_eval_ms = sum(eval_ms) / len(eval_ms)
        split = eng.c.gemm_mode == "bf16x3"
        plain = eng.c.gemm_mode == "bf16"
        # split-bf16 path: every fp32 product is 3 bf16 MFMAs, so the executed matrix-core work is 3x the
        # algorithmic FLOPs and the roof is the dense bf16 MFMA peak; plain bf16: one MFMA per product, same roof;
        # exact path: fp32 MFMA peak.
        peak = BF16_MFMA_PEAK_TFLOPS if (split or plain) else FP32_MFMA_PEAK_TFLOPS
        mult = 3.0 if split else 1.0
        roof = {"bound": "mfma", "unit": "TFLOP/s", "peak": peak, "traffic": None,
                "mfma_dtype": "bf16 (3 MFMAs per fp32 product, fp32 accumulate)" if split else
                              ("bf16 (one MFMA per product, fp32 accumulate)" if plain else "f32")}
        extra = {}
        if not a.no_breakdown:
            bd = kernel_breakdown(eng, rt, B)
            dom = max((k for k in bd if bd[k][2] > 0), key=lambda k: bd[k][1])
            n_dom, ms_dom, fl_dom = bd[dom]
            alg = fl_dom / (ms_dom * 1e-3) / 1e12
            roof.update({"kernel": dom + (" = fused transformer kernels k_tblock32 / k_tf128 / k_tf256" if dom == "k_tblock" else "")
                                   + " (all launches of the dominant kernel class in one U-Net eval)",
                         "achieved": round(alg * mult, 2), "frac": round(alg * mult / peak, 4),
                         "algorithmic_tflops_fp32_equiv": round(alg, 2), "launches_per_eval": n_dom,
                         "avg_launch_us": round(1e3 * ms_dom / n_dom, 2), "flops_per_launch_avg": fl_dom / n_dom})
            if dom == "k_tblock":
                # the contract's roofline is the MFMA fraction; what binds this kernel class is not a memory level
                roof["binding_resource"] = "the weight stream where it enters the CU, then the in-launch hand-off: LDS-DMA runs through the 64 B/clk/CU " \
                                           "vector-memory path (a 32 KB sub-tile = 512 cycles at best, 555 measured with nothing else running) and a " \
                                           "32-row workgroup of the 256-channel level consumes one in 650-700 cycles (384 of them MFMA); round-3 ablations " \
                                           "of a pair-split launch (DESIGN.md 3.5): no wait for tiles to land -9 %, no workgroup barriers at all -18 %, " \
                                           "no waits on fragment reads -3.6 %, broadcast fragment reads -4.4 % of the class, a deeper ring +4.5 % (slower); " \
                                           "in-kernel stamps: 74 % streamed phases, 15 % hand-off between the two workgroups of a row block (2.35 us per " \
                                           "round), 5 % LayerNorm, 4 % wave-pair LDS exchange; the shader clock is stretched to 1.75-1.95 GHz in the streamed " \
                                           "phases; HBM ~23 % of peak over the whole evaluation"
            if a.workload == "cfg1" and B == 1024:     # the committed PMC summary is of exactly this workload
                roof["traffic"], roof["traffic_source"] = pmc_traffic(dom)
            extra["eval_breakdown_ms"] = {k: {"launches": n, "ms": round(t, 4),
                                              "algorithmic_tflops": round(f / (t * 1e-3) / 1e12, 2) if f else None}
                                          for k, (n, t, f) in sorted(bd.items())}
        else:
            alg = flops_exec * B / (avg_eval_ms * 1e-3) / 1e12
            roof.update({"kernel": "whole U-Net eval", "achieved": round(alg * mult, 2),
                         "frac": round(alg * mult / peak, 4)})
        med_eval = sorted(eval_ms)[len(eval_ms) // 2]
        slow = [(i, t) for i, t in enumerate(eval_ms) if t > 1.25 * med_eval]
        ue = {"ms_avg_graph_replay": round(avg_eval_ms, 4), "ms_min_graph_replay": round(min(eval_ms), 4),
              "ms_max_graph_replay": round(max(eval_ms), 4), "ms_median_graph_replay": round(med_eval, 4),
              # evaluations more than 25 % over the median, as (index in the timed run, evaluation of its call 0..evals-1, ms):
              # an index that repeats at the same evaluation of every call is the path's own; a scattered one is the box's
              "outliers_over_1p25_median": {"count": len(slow), "first": [(i, i % evals, round(t, 3)) for i, t in slow[:8]]},
              "evals_timed": len(eval_ms),
              "launches": len(eng.c.programs["eval"]),
              "flops_per_sample_executed": flops_exec,
              "tflops_executed_fp32_equiv": round(flops_exec * B / (avg_eval_ms * 1e-3) / 1e12, 2)}
        if split or plain:
            ue["bf16_mfma_fraction_executed"] = round(mult * flops_exec * B / (avg_eval_ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4)
        else:
            ue["fp32_mfma_fraction_executed"] = round(flops_exec * B / (avg_eval_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)
        if a.workload == "cfg1":
            # NOT a roofline: the byte count of the reference's UNFUSED op graph (SURVEY §8d, 13.21 MB / sample / eval)
            # divided by this path's time, as a fraction of the HBM peak.  The fused kernels move far fewer bytes.
            ue["reference_unfused_opgraph_bytes_per_s_over_hbm_peak"] = round(
                REF_OPGRAPH_BYTES_PER_SAMPLE_EVAL * B / (avg_eval_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            if B == 1024:
                fname, rows = pmc_summary()
                # kernels of the EVAL program only (not the hoisted time / context programs' k_gemm_as / k_gemm / k_time_embed,
                # the sampler's own kernels or torch's fills), evaluations counted from a kernel that runs once per evaluation
                eval_kernels = ("mdt::k_tf128", "mdt::k_tf256", "mdt::k_tblock", "mdt::k_tb_reduce", "mdt::k_rconv", "mdt::k_gemm3",
                                "mdt::k_resblock", "mdt::k_attn", "mdt::k_gn_", "mdt::k_concat", "mdt::k_patch")
                evals_profiled = sum(n for nm, n, mb in rows if nm.startswith("mdt::k_resblock<16, 64>")) or None
                if rows and evals_profiled and pmc_is_current(fname)[0]:
                    unet_mb = sum(n * mb for nm, n, mb in rows if nm.startswith(eval_kernels))
                    per_eval = unet_mb / evals_profiled
                    ue["hbm_measured_mb_per_eval"] = round(per_eval, 1)
                    ue["hbm_measured_frac"] = round(per_eval * 1e6 / (avg_eval_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                    ue["hbm_measured_source"] = f"profiles/{fname}: PMC bytes (FETCH_SIZE x2 + WRITE_SIZE) of the eval program's kernels " \
                                                f"over the {evals_profiled} evaluations of the profiled run (counted from k_resblock<16, 64>, " \
                                                f"one launch per evaluation), over THIS run's evaluation time"
        extra["unet_eval"] = ue
        extra["power"] = power
        result = {
            "metric": "molecules/sec @64 diffusion steps (QM9 max_len=64)" if a.workload == "cfg1"
                      else f"molecules/sec @{T} diffusion steps ({a.workload})", "value": round(value, 2),
            "unit": "molecules/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * elapsed / a.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 storage/accumulate; GEMM products as split-bf16 (bf16x3) MFMA" if split else
                     ("f32 storage/accumulate; GEMM products in plain bf16 (reduced precision, see DESIGN.md)" if plain else "f32"),
            "data": "synthetic",
            "config": {"workload": WORKLOADS[a.workload]
                                   + f"batch={B}/GPU, {T} timesteps ({evals} U-Net evals), cond_scale={a.cond_scale}, " + ("bf16 products" if plain else "fp32"),
                       "global_batch": world * B, "timesteps": T, "parallelism": f"batch-shard x{world}"},
            "roofline": roof,
        }
        result.update(extra)

        if grouped:
            # the gathered tensor holds every rank's rows; counter-based noise is keyed by the GLOBAL sample index, so the
            # last rank's first rows must equal, bit for bit, a 1-rank run of those global indices with the same seed
            probe = min(64, B)
            r_last = world - 1
            # same kernels as the timed run (the 256-channel transformers' form depends on the batch in 'auto'; the two
            # forms agree to rounding only): pin the model to the choice the timed engine was compiled with
            model.kernel_choice = "wide" if eng.c.tf256 else "narrow"
            alone = model.sample(seq_of(r_last)[:probe].to(device), device, cond_scale=a.cond_scale, timesteps=T, clamp=False,
                                 noise=NoiseSource(seed=1234 + a.warmup + a.steps - 1, sample0=r_last * B))
            rows = out[r_last * B: r_last * B + probe]
            rates = [B * a.steps / r_[1] for r_ in per_rank]          # molecules/s of each rank's sample() calls alone
            gms = [1e3 * r_[2] / a.steps for r_ in per_rank]
            result["multi_gpu"] = {
                "backend": "gloo (test hook: ranks share cuda:0)" if share else "nccl (RCCL over xGMI)",
                "rccl_ranks_seen": dist.get_world_size(), "gathered_rows": int(out.shape[0]),
                "collectives_per_step": 1, "all_gather_bytes_per_rank": int(B * out.shape[1] * out.shape[2] * 4),
                # the ONE collective of a step, timed apart from the compute (HIP events on the rank's stream; a rank's figure
                # includes its wait for the slowest rank to arrive)
                "all_gather_ms_per_step": {"min_over_ranks": round(min(gms), 3), "max_over_ranks": round(max(gms), 3),
                                           "share_of_step_time_max": round(max(gms) * a.steps / (1e3 * elapsed), 4)},
                "sample_ms_per_step_per_rank": [round(1e3 * r_[1] / a.steps, 2) for r_ in per_rank],
                "per_rank_molecules_per_s": {"min": round(min(rates), 1), "max": round(max(rates), 1),
                                             "ranks": [round(v, 1) for v in rates]},
                # what ONE of these GPUs delivers without the collective: compare with the N = 1 line (BENCH) -- value / N below it
                # is what the all-gather and the slowest rank cost
                "n1_equivalent_value": round(sum(rates) / len(rates), 1),
                "value_per_gpu": round(value / world, 1),
                "shard_invariance": {"rank": r_last, "rows": probe, "bitwise_equal_to_1_rank_run": bool(torch.equal(rows, alone))}}

        if not a.no_exact_f32 and world == 1:
            # the strict-fp32 number: the SAME workload and the SAME fused op program with exact fp32 products -- the ring kernels
            # take fp32 fragment tiles and issue v_mfma_f32_16x16x4_f32 (MDT_F_WF32), the reference's arithmetic
            with contextlib.redirect_stdout(sys.stderr):
                m32 = make_synth_model(a.workload, device)
            m32.gemm_mode = "f32"
            f32_steps = 5
            m32.sample(seq, device, cond_scale=1.0, timesteps=4, clamp=False, noise=NoiseSource(seed=7, sample0=0))  # compile + graphs
            torch.cuda.synchronize(device)
            t32 = rt.EventTimer(evals * f32_steps + 8)
            c0 = time.perf_counter()
            for k in range(f32_steps):
                o32 = m32.sample(seq, device, cond_scale=1.0, timesteps=T, clamp=False, noise=NoiseSource(seed=1234 + k, sample0=0),
                                 timer=t32)
            torch.cuda.synchronize(device)
            dt32 = time.perf_counter() - c0
            ref_step = model.sample(seq, device, cond_scale=1.0, timesteps=T, clamp=False,
                                    noise=NoiseSource(seed=1234 + f32_steps - 1, sample0=0))
            e32 = m32._engine
            ev32 = t32.collect()
            avg32 = sum(ev32) / len(ev32)
            bd32 = kernel_breakdown(e32, rt, B)
            dom32 = max((k for k in bd32 if bd32[k][2] > 0), key=lambda k: bd32[k][1])
            n32, ms32, fl32 = bd32[dom32]
            alg32 = fl32 / (ms32 * 1e-3) / 1e12
            f32_traffic = pmc_traffic(dom32, variant="f32") if B == 1024 else (None, None)
            result["exact_f32"] = {
                "value": round(B * f32_steps / dt32, 2), "unit": "molecules/s", "steps": f32_steps,
                "ms_per_step": round(1e3 * dt32 / f32_steps, 2), "dtype": "f32 (every product an exact fp32 MFMA, v_mfma_f32_16x16x4_f32)",
                "program": f"{len(e32.c.programs['eval'])} launches per evaluation: the fused program of the default mode with fp32 "
                           "fragment tiles (MDT_F_WF32), not a layer-by-layer fallback",
                "max_abs_vs_default_mode_same_noise": float((o32 - ref_step).abs().max()),
                "roofline": {"bound": "mfma", "unit": "TFLOP/s", "peak": FP32_MFMA_PEAK_TFLOPS, "mfma_dtype": "f32",
                             "kernel": dom32 + " (all launches of the dominant kernel class in one U-Net eval)",
                             "achieved": round(alg32, 2), "frac": round(alg32 / FP32_MFMA_PEAK_TFLOPS, 4),
                             "launches_per_eval": n32, "avg_launch_us": round(1e3 * ms32 / n32, 2),
                             "traffic": f32_traffic[0], "traffic_source": f32_traffic[1]},
                "unet_eval": {"ms_avg_graph_replay": round(avg32, 4), "evals_timed": len(ev32),
                              "tflops_executed": round(flops_exec * B / (avg32 * 1e-3) / 1e12, 2),
                              "fp32_mfma_fraction_executed": round(flops_exec * B / (avg32 * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)},
                "eval_breakdown_ms": {k: {"launches": n, "ms": round(t, 4)} for k, (n, t, f) in sorted(bd32.items())}}
            del m32

        if not a.no_other_configs and world == 1 and a.workload == "cfg1" and a.cond_scale == 1.0:
            # the other single-GPU configurations of BASELINE.json with the same binary, few steps each (informational:
            # the headline `value` above is configs[1])
            def quick(tag, case, batch, tsteps, cscale, nsteps=3, gemm_mode=None, warm_timesteps=None):
                """One BASELINE configuration with the same binary: >= 3 timed sample() calls and the roofline fraction of ITS
                dominant kernel class (HIP-event time of every launch of one evaluation at this batch, as the headline's)."""
                with contextlib.redirect_stdout(sys.stderr):
                    mm = model if (case == "cfg1" and not gemm_mode) else make_synth_model(case, device)
                if gemm_mode:
                    mm.gemm_mode = gemm_mode
                sq = synth_normal(f"bench/other/{tag}", (batch, mm.unet.config.ctx_max_length)).to(device)
                mm.sample(sq, device, cond_scale=cscale, timesteps=warm_timesteps or tsteps, noise=NoiseSource(seed=5, sample0=0))
                torch.cuda.synchronize(device)
                c0 = time.perf_counter()
                for k in range(nsteps):
                    o = mm.sample(sq, device, cond_scale=cscale, timesteps=tsteps, noise=NoiseSource(seed=6 + k, sample0=0))
                torch.cuda.synchronize(device)
                dt = (time.perf_counter() - c0) / nsteps
                assert torch.isfinite(o).all()
                r = {"molecules_per_s": round(batch / dt, 1), "ms_per_step": round(1e3 * dt, 2), "batch": batch,
                     "timesteps": tsteps, "cond_scale": cscale, "steps": nsteps}
                e = mm._engine
                mode = e.c.gemm_mode
                pk = FP32_MFMA_PEAK_TFLOPS if mode == "f32" else BF16_MFMA_PEAK_TFLOPS
                mu = 3.0 if mode == "bf16x3" else 1.0
                if mode != "bf16x3":
                    r["gemm_mode"] = mode
                # whole sample() call (sampler updates, conditioning prelude and hoisted programs included) over the U-Net's
                # executed FLOPs (guidance: two passes per evaluation); split-bf16: three MFMAs per product
                fl = e.c.flops_per_sample_eval * batch * 2 * (tsteps - 1) * (2 if cscale != 1.0 else 1)
                r["tflops_executed"] = round(mu * fl / dt / 1e12, 1)
                r["mfma_fraction_whole_call"] = round(mu * fl / dt / 1e12 / pk, 4)
                try:
                    eb = e.B                                  # (guidance: the engine runs the doubled batch)
                    bd_ = kernel_breakdown(e, rt, eb)
                    dom_ = max((k for k in bd_ if bd_[k][2] > 0), key=lambda k: bd_[k][1])
                    n_, ms_, fl_ = bd_[dom_]
                    tot_ = sum(v[1] for v in bd_.values())
                    r["roofline"] = {"bound": "mfma", "unit": "TFLOP/s", "peak": pk, "kernel": dom_, "launches_per_eval": n_,
                                     "avg_launch_us": round(1e3 * ms_ / n_, 2), "achieved": round(mu * fl_ / (ms_ * 1e-3) / 1e12, 2),
                                     "frac": round(mu * fl_ / (ms_ * 1e-3) / 1e12 / pk, 4),
                                     "share_of_eval_time": round(ms_ / tot_, 3), "launches_in_eval": sum(v[0] for v in bd_.values())}
                except Exception as ex:                     # never lose the line over a diagnostic
                    r["roofline"] = {"error": repr(ex)}
                if mm is not model:
                    del mm
                return r
            def sampler_update_rate(batch):
                """The genuinely HBM-bound kernel class: the two halves of the ADPM2 update at the configs[3] shard
                size, HIP-event timed over 50 launches each; algorithmic bytes = every tensor read or written once."""
                lib = rt.load_library()
                C_, L_, Cp_ = 16, 64, 16
                x = torch.randn(batch, C_, L_, device=device)
                xm, pred, xin = torch.empty_like(x), torch.randn(batch, L_, Cp_, device=device), torch.empty(batch, L_, Cp_, device=device)
                st = rt.current_stream()
                res = {}
                for name, nbuf, call in (
                        ("k_adpm2_mid", 4, lambda: lib.mdt_adpm2_mid(rt.ptr(x), rt.ptr(pred), rt.ptr(xm), rt.ptr(xin), 0.5, 0.5, 1.0, -0.1,
                                                                     0.7, batch, C_, L_, Cp_, 0, st)),
                        ("k_adpm2_next", 5, lambda: lib.mdt_adpm2_next(rt.ptr(x), rt.ptr(xm), rt.ptr(pred), 0, rt.ptr(xin), 0.5, 0.5, 1.0,
                                                                       -0.1, 0.01, 0.7, 9, 1, 0, batch, C_, L_, Cp_, 0, 0, st))):
                    for _ in range(5):
                        rt.check(call())
                    tm = rt.EventTimer(1)
                    tm.start()
                    for _ in range(50):
                        rt.check(call())
                    tm.stop()
                    us = tm.collect()[0] * 1e3 / 50
                    nbytes = nbuf * batch * C_ * L_ * 4
                    res[name] = {"us_per_launch": round(us, 2), "algorithmic_mb": round(nbytes / 1e6, 1),
                                 "gb_per_s": round(nbytes / (us * 1e-6) / 1e9, 0),
                                 "hbm_peak_frac": round(nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 3)}
                res["batch"] = batch
                return res
            result["other_configs"] = {
                "configs[2] QMDiffusionForward": quick("cfg3", "cfg3", 4096, 100, 1.0),
                "configs[3] per-GPU shard (batch 8192)": quick("shard", "cfg1", 8192, 64, 1.0),
                "configs[1] with guidance (cond_scale 7.5)": quick("cfg", "cfg1", 1024, 64, 7.5),
                "configs[4] architecture (channels 256, fp32-class products)": quick("cfg5", "cfg5", 128, 16, 1.0),
                "configs[4] architecture in its bf16 mode (plain bf16 products, bf16 GEMM operands)":
                    quick("cfg5b", "cfg5", 1024, 16, 1.0, gemm_mode="bf16"),
                "configs[4] in its bf16 mode at its stated length (256 timesteps = 510 U-Net evaluations)":
                    # batch 2048 per GPU: the layer-by-layer bf16 program is HBM-bound by its fp32 residual stream and its GEMMs
                    # reach their rate from M = 16384 rows on the 1024-channel level (tools/cfg4_probe.py: 110 / 140 / 152 / 153
                    # molecules/s at batch 512 / 1024 / 2048 / 4096); 3 timed calls of ~14 s each
                    quick("cfg5c", "cfg5", 2048, 256, 1.0, gemm_mode="bf16", warm_timesteps=4),
                # strict-fp32 numbers beyond configs[1] (exact fp32 MFMA products; `exact_f32` above is configs[1])
                "configs[2] QMDiffusionForward, exact fp32 products": quick("cfg3f", "cfg3", 4096, 100, 1.0, gemm_mode="f32"),
                "configs[3] per-GPU shard (batch 8192), exact fp32 products": quick("shardf", "cfg1", 8192, 64, 1.0, gemm_mode="f32"),
                "sampler update kernels at the configs[3] shard size (HBM-bound class)": sampler_update_rate(8192),
            }

        if not a.no_cpu_baseline and world == 1:
            result["cpu_baseline"], result["parity"] = cpu_baseline_leg(torch, model, device, evals)
        print(json.dumps(result), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
