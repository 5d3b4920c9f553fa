#!/usr/bin/env python3
"""configs[4] (deep U-Net, plain-bf16 mode) throughput against the batch and the forced GEMM tile (GPU box):
   python tools/cfg4_probe.py [timesteps] [batches ...]          MDT_TILE16 = 0 / 1 / 2 forces 256x256 / 256x128 / 128x128."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from moleculediffusiontransformer_amd.synth import make_synth_model, synth_normal  # noqa: E402
from moleculediffusiontransformer_amd.diffusion import NoiseSource  # noqa: E402


def main():
    ts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    batches = [int(a) for a in sys.argv[2:]] or [512, 1024, 2048, 4096]
    dev = "cuda:0"
    m = make_synth_model("cfg5", dev)
    m.gemm_mode = "bf16"
    for B in batches:
        sq = synth_normal("probe/cfg4", (B, m.unet.config.ctx_max_length)).to(dev)
        m.sample(sq, dev, cond_scale=1.0, timesteps=4, noise=NoiseSource(seed=5, sample0=0))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 2
        for k in range(n):
            m.sample(sq, dev, cond_scale=1.0, timesteps=ts, noise=NoiseSource(seed=6 + k, sample0=0))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        evals = 2 * (ts - 1)
        print(f"B={B:5d} timesteps={ts}: {1e3 * dt:8.1f} ms/call  {1e3 * dt / evals:7.3f} ms/eval  "
              f"{B / (dt / evals * 510):8.1f} molecules/s at 256 steps (510 evaluations)", flush=True)


if __name__ == "__main__":
    main()
