for rep in 1 2; do
for v in 1 0; do
  echo "== MDT_PATCH_CONV=$v"
  MDT_PATCH_CONV=$v python bench.py --no-cpu-baseline --no-exact-f32 --no-other-configs --no-breakdown --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('headline', d['value'], d['unet_eval']['ms_avg_graph_replay'], d['unet_eval']['launches'])"
  MDT_PATCH_CONV=$v python bench.py --workload cfg3 --batch 4096 --timesteps 100 --no-breakdown --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg3 B4096', d['value'], d['unet_eval']['ms_avg_graph_replay'], d['unet_eval']['launches'])"
done
done
