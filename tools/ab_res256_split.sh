mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x -k "test_sample_matches or test_wide_batch or test_repeated or configs1 or configs2 or kernel_choice_pin or handoff or co_residency or chunks or chained_resnet" -p no:cacheprovider 2>&1 | tail -6
Q="--no-cpu-baseline --no-exact-f32 --no-other-configs --steps 5 --warmup 2"
for i in 1 2; do
  for m in whole auto; do
    MDT_RES256=$m python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', d['value'], d['unet_eval']['ms_avg_graph_replay'], d['unet_eval']['launches'], d['eval_breakdown_ms'])"
  done
done
for m in whole auto; do
  MDT_RES256=$m python bench.py $Q --workload cfg3 --batch 4096 --timesteps 100 --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3 $m', d['value'], d['unet_eval']['ms_avg_graph_replay'], d['unet_eval']['launches'])"
done
