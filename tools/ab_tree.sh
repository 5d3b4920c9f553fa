#!/bin/bash
# Same-box A/B of two TREES (library + compiler + engine): the baseline exported with `git archive <rev>` into .ab_base/ and built there
# in the build container, against the working tree.  Headline workload (configs[1], B = 1024, 64 steps), alternating.
#   tools/ab_tree.sh [bench args ...]
Q="${@:---no-breakdown --no-cpu-baseline --no-exact-f32 --no-other-configs --steps 3 --warmup 1}"
one() { (cd $1 && python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', d['value'], d['unet_eval']['ms_avg_graph_replay'], d['unet_eval']['launches'])"); }
for i in 1 2 3; do
  one .ab_base base
  one . current
done
