#!/bin/bash
# Same-box A/B of the up path.s cat([x, skip]) read from its two sources by the GroupNorm pass (MDT_CAT_FOLD=1, round 6) against k_concat + a conversion pass (0):
# configs[4] architecture, bf16 mode, B = 2048, 16 timesteps (30 evaluations per call), alternating
Q="--workload cfg5 --gemm-mode bf16 --batch 2048 --timesteps 16 --no-breakdown --steps 2 --warmup 1"
for i in 1 2; do
  for m in 0 1; do
    MDT_CAT_FOLD=$m python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MDT_CAT_FOLD=$m', d['value'], d['unet_eval']['ms_avg_graph_replay'], d['unet_eval']['launches'])"
  done
done
