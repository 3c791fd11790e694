#!/bin/bash
# Everything under profiles/ that comes from a GPU box, in one go (run from the repo root on an MI355X):
#   bash benchmarks/collect_profiles.sh <tag>      -> gpurun_out/<tag>/...
# bench line, rocprofv3 kernel stats of the same command, the two PMC passes (+ hbm_traffic.json),
# the secondary configs with their kernel stats, and the parity report.
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python bench.py > "$OUT/bench_line.json" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python bench.py --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof_stats.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> "$OUT/rocprof_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> "$OUT/rocprof_write.err"
python benchmarks/kernel_trace_summary.py "$OUT/stats" > "$OUT/bench_path_kernel_trace_summary.json" 2> "$OUT/kernel_trace_summary.err"
python benchmarks/hbm_traffic.py "$OUT" > "$OUT/hbm_traffic.json" 2> "$OUT/hbm_traffic.err"
# (the 1.25e7-ray shard of the multi-GPU runs has its own entry in profiles/hbm_traffic.json: scratch/pmc_125.sh)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/configs_stats" -- python benchmarks/run_configs.py > "$OUT/run_configs.jsonl" 2> "$OUT/run_configs.err"
python tests/parity_report.py > "$OUT/parity_report.txt" 2> "$OUT/parity_report.err"
# keep only the summaries (the raw traces are large)
find "$OUT" -name "*kernel_stats.csv" | while read f; do cp "$f" "$OUT/$(basename $(dirname $(dirname "$f")))_kernel_stats.csv"; done
find "$OUT" -name "*counter_collection.csv" | while read f; do head -200 "$f" > "$OUT/$(basename $(dirname $(dirname "$f")))_counter_head.csv"; done
rm -rf "$OUT/stats" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/configs_stats"
ls -la "$OUT"
