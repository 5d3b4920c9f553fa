#!/usr/bin/env python3
"""Headline benchmark: molecules/sec at 64 diffusion steps (QM9-shaped inverse model) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one full QMDiffusion.sample() call over one batch: BASELINE.json configs[1] = inverse model
channels=64, pred_dim=16, max_len=64, cond_len=12, batch 1024 per GPU, 64 timesteps (126 U-Net
evaluations + 63 ADPM2 updates), fp32, cond_scale=1.0, synthetic weights/conditioning, inputs resident in
HBM, on-device counter-based noise.  For N > 1 the driver launches one rank per GPU through
torch.distributed.run; every rank samples its own 1024 molecules (weak scaling, no collective in the
loop) and the generated samples are all-gathered once per call over RCCL.

Prints ONE JSON line (rank 0) with the contract fields plus
  "roofline":     the dominant kernel class (k_gemm: fp32-MFMA implicit GEMM), HIP-event timed per launch
  "cpu_baseline": the CPU oracle (PyTorch restatement pinned to the reference) on a bounded sample
"""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA (not the 2:1-sparse headline)
HBM_PEAK_GBS = 8000.0
REF_FLOPS_PER_SAMPLE_EVAL = 455.3e6   # reference op graph (incl. per-eval cross-attn K/V + time mapping)
REF_OPGRAPH_BYTES_PER_SAMPLE_EVAL = 13.21e6


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1024, help="molecules per GPU per step")
    ap.add_argument("--timesteps", type=int, default=64)
    ap.add_argument("--workload", default="cfg1", choices=["cfg1", "cfg3", "cfg5"],
                    help="cfg1 = BASELINE configs[1] (the headline metric, default); cfg3 = QMDiffusionForward "
                         "(configs[2]: --batch 4096 --timesteps 100); cfg5 = deep U-Net architecture of configs[4] "
                         "(channels 256, max_len 128).  Other workloads are informational: no CPU baseline / parity leg")
    ap.add_argument("--cond-scale", type=float, default=1.0, help="classifier-free guidance scale (2 U-Net passes if != 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true")
    return ap.parse_args()


def op_flops(op, rt, B):
    """Algorithmic FLOPs (2*MAC, fp32-equivalent) of one op of the eval program for batch B."""
    i = op.i
    if op.kind == rt.OP_GEMM:
        return 2.0 * B * i[rt.G_R_OUT] * i[rt.G_N] * i[rt.G_TAPS] * i[rt.G_CIN]
    if op.kind == rt.OP_ATTN:
        return 4.0 * B * i[rt.A_T] * i[rt.A_TK] * 64 * i[rt.A_HEADS]
    if op.kind == rt.OP_RCONV:
        return 2.0 * B * i[rt.R_T] * i[rt.R_C] * i[rt.R_C] * i[rt.R_TAPS]
    if op.kind == rt.OP_RESBLOCK:
        cin, cout = i[rt.K_CIN], i[rt.K_COUT]
        return 2.0 * B * i[rt.K_T] * (3 * cin * cout + 3 * cout * cout + cin * cout)
    if op.kind == rt.OP_TBLOCK:
        c, t, nch, tk = i[rt.B_C], i[rt.B_T], i[rt.B_NCHUNK], i[rt.B_TK]
        mid = 64 * nch
        if i[rt.B_MODE] == rt.TB_FF:
            return 4.0 * B * t * c * mid
        if i[rt.B_MODE] == rt.TB_SELF:
            return B * (2.0 * t * c * 3 * mid + 4.0 * t * t * mid + 2.0 * t * mid * c)
        return B * (2.0 * t * c * mid + 4.0 * t * tk * mid + 2.0 * t * mid * c)
    return 0.0


def kernel_breakdown(model, eng, torch, rt, B):
    """HIP-event time of every op of ONE U-Net evaluation (plain launches, one interval per launch),
    grouped by kernel class.  Returns dict class -> (launches, total_ms, flops)."""
    prog = eng.programs["eval"]
    ops = eng.c.programs["eval"]
    bind = eng._bind(xin=eng.xin, out=eng.pred)
    names = {rt.OP_GEMM: "k_gemm", rt.OP_GN_STATS: "k_gn_stats", rt.OP_ATTN: "k_attn", rt.OP_CONCAT: "k_concat",
             rt.OP_PATCH: "k_patch", rt.OP_TBLOCK: "k_tblock", rt.OP_GN_ACT: "k_gn_act", rt.OP_RCONV: "k_rconv",
             rt.OP_RESBLOCK: "k_resblock"}
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
        k = names[op.kind]
        n, tot, fl = out.get(k, (0, 0.0, 0.0))
        out[k] = (n + 1, tot + t, fl + op_flops(op, rt, B))
    return out


def pmc_traffic(kernel_class):
    """HBM bytes per launch of one kernel class, from the committed PMC summary of this same workload
    (profiles/r*_pmc_hbm_traffic.csv: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate passes)."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles",
                                          "r*_pmc_hbm_traffic.csv")))
    if not files:
        return None, None
    n, mb = 0, 0.0
    for line in open(files[-1]):
        if line.startswith("#") or line.startswith("kernel,"):
            continue
        name, rest = line.rsplit(",", 5)[0].strip('"'), line.strip().rsplit(",", 5)[1:]
        if name.startswith("mdt::" + kernel_class):
            n += int(rest[0])
            mb += int(rest[0]) * float(rest[4])
    if n == 0:
        return None, None
    return round(mb / n * 1e6), ("bytes per launch; profiles/" + os.path.basename(files[-1]) +
                                 " (rocprofv3 PMC passes of this workload, recorded earlier, not collected live)")


def main():
    a = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the sampling path has no CPU fallback")
    # test hook (one-GPU boxes): MDT_BENCH_SHARE_GPU=1 puts every rank on cuda:0 over gloo to exercise the N > 1 logic
    share = os.environ.get("MDT_BENCH_SHARE_GPU", "0") == "1"
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    from gpu_util import make_model
    from moleculediffusiontransformer_amd import NoiseSource, runtime as rt
    from moleculediffusiontransformer_amd.distributed import all_gather_samples
    from moleculediffusiontransformer_amd.synth import synth_normal

    import gpu_util
    gpu_util.DEV = str(device)
    with contextlib.redirect_stdout(sys.stderr):     # the class prints "Using unet type" like the reference does
        model = make_model(a.workload)               # cfg1: inverse c=64, pred_dim=16, L=64, cond_len=12; synthetic weights
    B, T = a.batch, a.timesteps
    n_cond = model.unet.config.ctx_max_length
    seq = synth_normal(f"bench/seq/rank{rank}", (B, n_cond)).to(device)
    if a.workload != "cfg1" or a.cond_scale != 1.0:
        a.no_cpu_baseline = True
    evals = 2 * (T - 1)
    eval_timer = rt.EventTimer(evals * (a.steps + a.warmup) + 8)

    def one_step(step_idx, timed):
        out = model.sample(seq, device, cond_scale=a.cond_scale, timesteps=T, clamp=False,
                           noise=NoiseSource(seed=1234 + step_idx, sample0=rank * B),
                           timer=eval_timer if timed else None)
        if world > 1:
            out = all_gather_samples(out, world * B)
        return out

    for w in range(a.warmup):
        one_step(w, False)

    def fence():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    fence()
    t0 = time.perf_counter()
    for k in range(a.steps):
        out = one_step(a.warmup + k, True)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(out).all()

    eval_ms = eval_timer.collect()
    result = None
    if rank == 0:
        eng = model.engine(device, n_cond)
        mols = world * B * a.steps
        value = mols / elapsed
        flops_exec = eng.c.flops_per_sample_eval           # executed per sample per eval (K/V + time mapping hoisted)
        avg_eval_ms = sum(eval_ms) / len(eval_ms)
        split = eng.c.gemm_mode == "bf16x3"
        # split-bf16 path: every fp32 product is 3 bf16 MFMAs, so the executed matrix-core work is 3x the
        # algorithmic FLOPs and the roof is the dense bf16 MFMA peak; exact path: fp32 MFMA peak.
        peak = BF16_MFMA_PEAK_TFLOPS if split else FP32_MFMA_PEAK_TFLOPS
        mult = 3.0 if split else 1.0
        roof = {"bound": "mfma", "unit": "TFLOP/s", "peak": peak, "traffic": None,
                "mfma_dtype": "bf16 (3 MFMAs per fp32 product, fp32 accumulate)" if split else "f32"}
        extra = {}
        if not a.no_breakdown:
            bd = kernel_breakdown(model, eng, torch, rt, B)
            dom = max((k for k in bd if bd[k][2] > 0), key=lambda k: bd[k][1])
            n_dom, ms_dom, fl_dom = bd[dom]
            alg = fl_dom / (ms_dom * 1e-3) / 1e12
            roof.update({"kernel": dom + " (all instantiations of the dominant kernel class in one U-Net eval)",
                         "achieved": round(alg * mult, 2), "frac": round(alg * mult / peak, 4),
                         "algorithmic_tflops_fp32_equiv": round(alg, 2), "launches_per_eval": n_dom,
                         "avg_launch_us": round(1e3 * ms_dom / n_dom, 2), "flops_per_launch_avg": fl_dom / n_dom})
            if a.workload == "cfg1" and B == 1024:     # the committed PMC summary is of exactly this workload
                roof["traffic"], roof["traffic_source"] = pmc_traffic(dom)
            extra["eval_breakdown_ms"] = {k: {"launches": n, "ms": round(t, 4),
                                              "algorithmic_tflops": round(f / (t * 1e-3) / 1e12, 2) if f else None}
                                          for k, (n, t, f) in sorted(bd.items())}
        else:
            alg = flops_exec * B / (avg_eval_ms * 1e-3) / 1e12
            roof.update({"kernel": "whole U-Net eval", "achieved": round(alg * mult, 2),
                         "frac": round(alg * mult / peak, 4)})
        extra["unet_eval"] = {
            "ms_avg_graph_replay": round(avg_eval_ms, 4), "evals_timed": len(eval_ms),
            "flops_per_sample_executed": flops_exec,
            "tflops_executed": round(flops_exec * B / (avg_eval_ms * 1e-3) / 1e12, 2),
            "mfma_fraction_executed": round(flops_exec * B / (avg_eval_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
            "mfma_fraction_reference_opgraph": round(REF_FLOPS_PER_SAMPLE_EVAL * B / (avg_eval_ms * 1e-3) / 1e12
                                                     / FP32_MFMA_PEAK_TFLOPS, 4),
            "hbm_fraction_reference_opgraph": round(REF_OPGRAPH_BYTES_PER_SAMPLE_EVAL * B / (avg_eval_ms * 1e-3) / 1e9
                                                    / HBM_PEAK_GBS, 4),
        }
        if a.workload != "cfg1":
            for k in ("mfma_fraction_reference_opgraph", "hbm_fraction_reference_opgraph"):
                extra["unet_eval"].pop(k, None)
        result = {
            "metric": "molecules/sec @64 diffusion steps (QM9 max_len=64)" if a.workload == "cfg1"
                      else f"molecules/sec @{T} diffusion steps ({a.workload})", "value": round(value, 2),
            "unit": "molecules/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * elapsed / a.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 storage/accumulate; GEMM products as split-bf16 (bf16x3) MFMA" if split else "f32",
            "data": "synthetic",
            "config": {"workload": {"cfg1": "QMDiffusion inverse sample(): channels=64 pred_dim=16 max_len=64 cond_len=12, ",
                                    "cfg3": "QMDiffusionForward sample(): channels=64 pred_dim=1 max_len=64 cond_len=64, ",
                                    "cfg5": "QMDiffusion inverse sample(): channels=256 pred_dim=32 max_len=128 cond_len=12, "}[a.workload]
                                   + f"batch={B}/GPU, {T} timesteps ({evals} U-Net evals), cond_scale={a.cond_scale}, fp32",
                       "global_batch": world * B, "timesteps": T, "parallelism": f"batch-shard x{world}"},
            "roofline": roof,
        }
        result.update(extra)

        if not a.no_cpu_baseline and world == 1:
            from helpers import oracle_cfg, synth_sd
            from oracle import unet_oracle as O
            cb, ct = 128, 8                                   # bounded sample: 128 molecules, 8 timesteps = 14 evals
            sd, cfg = synth_sd("cfg1"), oracle_cfg("cfg1")
            cseq = synth_normal("bench/cpu/seq", (cb, 12))
            init = synth_normal("bench/cpu/init", (cb, 16, 64))
            nz = [synth_normal(f"bench/cpu/step{i}", (cb, 16, 64)) for i in range(ct - 1)]
            O.sample(sd, cfg, cseq[:4], init[:4], lambda i, x: nz[i][:4], 3, 1.0, False)   # warm-up
            c0 = time.perf_counter()
            cpu_out = O.sample(sd, cfg, cseq, init, lambda i, x: nz[i], ct, 1.0, False)
            cdt = time.perf_counter() - c0
            # the same bounded sample through the HIP path on the identical noise: the checker of this run
            hip_out = model.sample(cseq, device, cond_scale=1.0, timesteps=ct, clamp=False,
                                   noise=NoiseSource(init=init, steps=lambda i: nz[i])).cpu()
            result["parity"] = {"max_abs_vs_cpu_reference_path": float((hip_out - cpu_out).abs().max()),
                                "tolerance": 1e-4, "sample": f"batch {cb}, {ct} timesteps, identical noise"}
            per_eval = cdt / (2 * (ct - 1))
            result["cpu_baseline"] = {
                "value": round(cb / (per_eval * evals), 3), "unit": "molecules/s", "cores": torch.get_num_threads(),
                "kind": "port",
                "sample": f"oracle/unet_oracle.py (PyTorch CPU fp32, bit-exact to the reference): batch {cb}, {ct} "
                          f"timesteps = {2 * (ct - 1)} U-Net evals in {cdt:.2f} s, scaled to {evals} evals"}
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
