for v in 1 0; do
  echo "== MDT_RES256=$v"
  MDT_RES256=$v python bench.py --workload cfg3 --batch 4096 --timesteps 100 --no-breakdown --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg3 B4096', d['value'], d['unet_eval']['ms_avg_graph_replay'], d['unet_eval']['launches'])"
  MDT_RES256=$v python bench.py --cond-scale 7.5 --no-breakdown --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('guided', d['value'], d['unet_eval']['ms_avg_graph_replay'], d['unet_eval']['launches'])"
  MDT_RES256=$v python bench.py --batch 8192 --no-cpu-baseline --no-exact-f32 --no-other-configs --no-breakdown --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('shard8192', d['value'], d['unet_eval']['ms_avg_graph_replay'], d['unet_eval']['launches'])"
  MDT_RES256=$v python bench.py --batch 2048 --no-cpu-baseline --no-exact-f32 --no-other-configs --no-breakdown --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('B2048', d['value'], d['unet_eval']['ms_avg_graph_replay'], d['unet_eval']['launches'])"
done
