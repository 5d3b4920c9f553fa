#!/bin/bash
# Same-box A/B of the pair-split ResNet chains (MDT_RES256=auto, round 6) against round 5's policy (MDT_RES256=whole: 26 k_rconv launches in
# the narrow program), alternating, BASELINE configs[1]: molecules/s, evaluation ms (graph replay), launches per evaluation
Q="--no-cpu-baseline --no-exact-f32 --no-other-configs --no-breakdown --steps 5 --warmup 2"
for i in 1 2 3; do
  for m in whole auto; do
    MDT_RES256=$m python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', d['value'], d['unet_eval']['ms_avg_graph_replay'], d['unet_eval']['launches'])"
  done
done
