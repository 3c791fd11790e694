#!/bin/bash
# What the figures of DESIGN.md section 5 come from, in one go (run from the repo root on an MI355X, ~6 minutes):
#   bash benchmarks/collect_profiles.sh <tag>      -> gpurun_out/<tag>/...   (copy what is to be judged into profiles/)
# bench.py's default run measures every configuration and takes the HBM traffic / FP64 instruction counters itself
# (rocprofv3 PMC passes over its --pmc-inner mode: separate passes, kernel trace only).  Here: that line (the FIRST GPU
# process of the box: what the driver's run is) + its detail file; the same command under rocprofv3 --kernel-trace
# --stats (the per-kernel averages the line's kernel_ms must agree with), summarised per configuration; fresh-process
# repeats; the 1e8-ray bundle alone; the N > 1 code path with one rank through RCCL; parity and ISA reports.
set -u
TAG=${1:-r06}
O=gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python bench.py --detail $O/bench_detail.json > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc=$?" > $O/rc.txt
STEPS=50
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps $STEPS --warmup 10 --no-cpu-baseline --traffic none --no-scaling-point --no-e2e --detail $O/bench_detail_under_rocprof.json > $O/bench_line_under_rocprof.json 2> $O/stats.err
# (k_trace_iso<MODE, VEC_IN, VEC_OUT, SHAPES, MOMENTS, UNI, IMG>: the double Gauss runs with the uniform first segment, the
#  reference's benchmark workload -- a divergent bundle -- with k0 / E0 arrays)
for cfg in "doublegauss:k_trace_iso<0, true, true, 0, false, true," "benchmark:k_trace_iso<0, true, true, 0, false, false," "asphere:k_trace_iso<0, true, true, 1," "xypoly:k_trace_iso<0, true, true, 2," "aniso:k_trace_general<0, false," "aniso_biaxial:k_trace_general<0, true," "plugin_propagate:k_propagate_rows<" "plugin_interact:k_interact_iso_rows<"; do
  python benchmarks/kernel_trace_summary.py $O/stats "${cfg#*:}" $STEPS > $O/${cfg%%:*}_kernel_trace_summary.json 2>> $O/stats.err
done
find $O/stats -name "*kernel_stats.csv" | head -1 | while read f; do cp "$f" $O/bench_kernel_stats.csv; done
for i in 1 2 3 4 5; do python bench.py --headline-only --no-cpu-baseline --traffic none --detail /tmp/fresh_detail.json > $O/fresh_$i.json 2> /dev/null; done
python - "$O" <<'PY' > $O/five_fresh_processes.json
import json, sys
rows = []
for i in range(1, 6):
    try:
        rows.append(json.loads(open("%s/fresh_%d.json" % (sys.argv[1], i)).read().strip().splitlines()[-1]))
    except Exception as exc:
        rows.append({"error": str(exc)})
print(json.dumps([r if "error" in r else
                  {"ms_per_step": r["ms_per_step"], "kernel_ms": r["roofline"]["kernel_ms"], "frac": r["roofline"]["frac"],
                   "placement": r["config"]["placement"]} for r in rows], indent=1))
PY
python bench.py --rays 100000000 --steps 3 --warmup 1 --headline-only --no-cpu-baseline --traffic none --detail /tmp/d.json > $O/bench_1e8_rays_line.json 2> $O/bench_1e8.err
# the N > 1 protocol as the driver runs it (--scaling strong: the 1e8-ray bundle, 5 wavelengths), one rank through RCCL
python bench.py --force-multi --steps 20 --warmup 5 --detail $O/bench_force_multi_strong_detail.json > $O/bench_force_multi_strong_1e8.json 2> $O/bench_force_multi_strong.err
python tests/parity_report.py > $O/parity_report.txt 2> $O/parity_report.err
python benchmarks/isa_report.py > $O/isa_report.json 2> /dev/null
# keep only the summaries (the raw traces are large)
find $O -name "*counter_collection.csv" -delete
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
find $O -name "*.db" -delete
rm -rf $O/stats
rm -f $O/fresh_?.json
cat $O/rc.txt; ls $O
