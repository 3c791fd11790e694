export TMPDIR=/tmp
OUT=gpurun_out/pmc125
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python bench.py --rays 12500000 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> "$OUT/fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python bench.py --rays 12500000 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> "$OUT/write.err"
python benchmarks/hbm_traffic.py "$OUT" k_trace_iso 12492497 12 > "$OUT/hbm_traffic_125.json" 2> "$OUT/traffic.err"
cat "$OUT/hbm_traffic_125.json" | head -20; tail -2 "$OUT/traffic.err"
rm -rf "$OUT/pmc_fetch" "$OUT/pmc_write"
