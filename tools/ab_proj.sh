for rep in 1 2; do
for v in 1 0; do
  echo "== MDT_PROJ=$v"
  MDT_PROJ=$v python bench.py --workload cfg3 --batch 4096 --timesteps 100 --no-breakdown --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg3 B4096', d['value'], d['unet_eval']['ms_avg_graph_replay'], d['unet_eval']['launches'])"
done
done
