#!/bin/bash
# Same-box A/B of k_gemm_b16's all-bf16 epilogue (w16 = 1: 8 columns per lane, the tile's bf16 residual requested under the main
# loop; round 6) against the generic epilogue (0): configs[4] architecture, bf16 mode, B = 2048, 16 timesteps, alternating.
# The switch is the run-time hook mdt_set_tuning("w16", v), so both legs run the same library.
Q="--workload cfg5 --gemm-mode bf16 --batch 2048 --timesteps 16 --no-breakdown --steps 2 --warmup 1"
for i in 1 2; do
  for m in 0 1; do
    python - $m $Q 2>/dev/null <<'PY' | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('w16=$m', d['value'], d['unet_eval']['ms_avg_graph_replay'], d['unet_eval']['launches'])"
import runpy, sys
from moleculediffusiontransformer_amd import runtime as rt
rt.load_library().mdt_set_tuning(b"w16", int(sys.argv[1]))
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path("bench.py", run_name="__main__")
PY
  done
done
