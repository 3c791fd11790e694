#!/bin/bash
# Everything under profiles/ that comes from a GPU box, in one go (run from the repo root on an MI355X):
#   bash benchmarks/collect_profiles.sh <tag>      -> gpurun_out/<tag>/...
# Per bench config: the plain bench line, rocprofv3 kernel stats of the same command, PMC passes in SEPARATE
# runs (kernel trace only -- never combined with sys / hip / hsa tracing): HBM traffic (FETCH_SIZE, WRITE_SIZE),
# for the crystal march also the FP64 instruction counters.  benchmarks/pmc_summary.py turns the counter files
# into the per-launch figures that go into profiles/hbm_traffic.json and profiles/fp64_flops.json.
set -u
TAG=${1:-r02}
O=gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
B="python bench.py --no-cpu-baseline --steps 20 --warmup 5"
python bench.py > $O/dg_bench_plain.json 2> $O/dg_bench_plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dg_stats -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/dg_stats_bench.json 2> $O/dg_stats.err
python benchmarks/kernel_trace_summary.py $O/dg_stats k_trace_iso 50 > $O/dg_kernel_trace_summary.json 2>> $O/dg_stats.err
for cfg in doublegauss asphere aniso; do
  kernel=k_trace_iso; [ $cfg = aniso ] && kernel=k_trace_general
  python bench.py --config $cfg > $O/${cfg}_bench_plain.json 2> $O/${cfg}_bench_plain.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${cfg}_stats -- $B --config $cfg > $O/${cfg}_stats_bench.json 2> $O/${cfg}_stats.err
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${cfg}_pmc/fetch -- $B --config $cfg > /dev/null 2> $O/${cfg}_fetch.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${cfg}_pmc/write -- $B --config $cfg > /dev/null 2> $O/${cfg}_write.err
  if [ $cfg = aniso ]; then
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/${cfg}_pmc/f64 -- $B --config $cfg > /dev/null 2> $O/${cfg}_f64.err
    rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/${cfg}_pmc/sq1 -- $B --config $cfg > /dev/null 2> $O/${cfg}_sq1.err
    rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $O/${cfg}_pmc/sq2 -- $B --config $cfg > /dev/null 2> $O/${cfg}_sq2.err
  fi
  python benchmarks/pmc_summary.py $O/${cfg}_pmc $kernel > $O/${cfg}_pmc_counters.json 2>> $O/${cfg}_fetch.err
done
python bench.py --force-multi --steps 20 --warmup 5 > $O/bench_force_multi.json 2> $O/bench_force_multi.err
python benchmarks/ab_variants.py > $O/ab_variants.json 2> $O/ab_variants.err
python benchmarks/ab_crystal.py > $O/ab_crystal.json 2> $O/ab_crystal.err
python benchmarks/ab_shapes.py > $O/ab_shapes.json 2> $O/ab_shapes.err
python benchmarks/dropin_call_time.py > $O/dropin.json 2> $O/dropin.err
python tests/parity_report.py > $O/parity_report.txt 2> $O/parity_report.err
# keep only the summaries (the raw traces are large)
find $O -name "*_kernel_stats.csv" | while read f; do cp "$f" $O/$(basename $(dirname $(dirname $f)))_kernel_stats.csv; done
find $O -name "*counter_collection.csv" -size +2M -delete
find $O -name "*kernel_trace.csv" -size +2M -delete
find $O -name "*.db" -delete
ls -la $O
