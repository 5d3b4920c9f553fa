#!/bin/bash
# Round profile recipe (run on the GPU box from the repo root):
#   bash tools/profile_round.sh r3                                   # the headline workload (BASELINE configs[1], B = 1024)
#   BENCH_ARGS="--workload cfg3 --batch 4096" TRACE_TS=100 bash tools/profile_round.sh r3_cfg3      # another configuration
# 1. plain bench.py (headline only: with the CPU baseline leg)         -> gpurun_out/prof_<tag>/bench.json
# 2. rocprofv3 --output-format csv --kernel-trace --stats of the same command   -> .../trace (kernel_stats.csv, domain_stats.csv)
# 3. PMC passes, kernel-trace only, on a 4-timestep run: FETCH_SIZE, WRITE_SIZE (HBM traffic), SQ wave-state /
#    MFMA-busy / LDS-conflict counters -> .../pmc_fetch, .../pmc_write, .../pmc_sq
# tools/pmc_summary.py then folds 2+3 into the csv files committed under profiles/.
tag=${1:-r3}
args=${BENCH_ARGS:-}
trace_ts=${TRACE_TS:-64}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
quiet="--no-cpu-baseline --no-exact-f32 --no-other-configs"
echo "bench.py $args (kernel trace: --timesteps $trace_ts; PMC passes: --timesteps 4 --steps 1 --warmup 1)" > "$out/args.txt"
if [ -z "$args" ]; then
  python3 bench.py > "$out/bench.json" 2> "$out/bench.err"
fi
cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d "$out/trace" -o runc -- python3 "$root/bench.py" $args --timesteps $trace_ts $quiet \
    > "$out/bench_under_rocprof.json" 2> "$out/trace.log"
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d "$out/pmc_fetch" -o runc -- python3 "$root/bench.py" $args \
    $quiet --no-breakdown --timesteps 4 --steps 1 --warmup 1 > "$out/pmc_fetch.json" 2> "$out/pmc_fetch.log"
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d "$out/pmc_write" -o runc -- python3 "$root/bench.py" $args \
    $quiet --no-breakdown --timesteps 4 --steps 1 --warmup 1 > "$out/pmc_write.json" 2> "$out/pmc_write.log"
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d "$out/pmc_sq" -o runc -- python3 "$root/bench.py" $args \
    $quiet --no-breakdown --timesteps 4 --steps 1 --warmup 1 > "$out/pmc_sq.json" 2> "$out/pmc_sq.log"
cd "$root"
# the per-kernel csv files of the passes are large: keep only the folded summaries (written to gpurun_out/prof_<tag>/summary,
# copied to profiles/ where that is tracked:  python3 tools/pmc_summary.py gpurun_out/prof_$tag $tag)
python3 tools/pmc_summary.py "$out" "$tag" "$out/summary" > "$out/summary.log" 2>&1
find "$out" -name "*_counter_collection.csv" -delete; find "$out" -name "*_kernel_trace.csv" -delete
