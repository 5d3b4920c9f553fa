#!/bin/bash
# Same-box A/B of two builds of the library on configs[4] (bf16 mode, B = 2048, 16 timesteps), alternating:
#   tools/ab_lib.sh <tag>      compares libmdt_hip_<tag>.so (built earlier with MDT_LIB_TAG=<tag>, loaded as it is) with the current one
TAG=$1
Q="--workload cfg5 --gemm-mode bf16 --batch 2048 --timesteps 16 --no-breakdown --steps 2 --warmup 1"
for i in 1 2; do
  MDT_LIB_TAG=$TAG MDT_NO_BUILD=1 python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$TAG', d['value'], d['unet_eval']['ms_avg_graph_replay'], d['unet_eval']['launches'])"
  python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=current', d['value'], d['unet_eval']['ms_avg_graph_replay'], d['unet_eval']['launches'])"
done
