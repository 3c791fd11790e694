#!/bin/bash
# First contact with a multi-GPU node: the N = 1, 2, 4, 8 launches of the contract, each with its watchdog, for both
# per-step exchanges -- the image-plane gather (`auto`: the faster of in-place RCCL all-gather and direct peer writes, by
# a 3-step probe at start-up) and the statistics all-reduce alone -- one compact JSON line per run.
#   bash benchmarks/scale_preflight.sh [out dir] [max N] [extra bench.py arguments ...]
# e.g. on an 8-GPU node:      bash benchmarks/scale_preflight.sh gpurun_out/scale 8
#      dry run on ONE GPU:    bash benchmarks/scale_preflight.sh /tmp/scale 4 --backend gloo --rays-total 4000000 --steps 3 --warmup 1
# (gloo: every rank on cuda:0, gather staged through host memory -- plumbing only).  Every line carries rccl_world,
# ok_per_rank, ms_trace / ms_gather / ms_total (config) and the same scaling_point{rays, ms, frac, ok, value} shape the
# N = 1 default line has, so "N = 1 of the curve agrees with BENCH" can be checked by reading two numbers.
set -u
OUT=${1:-gpurun_out/scale_preflight}
MAXN=${2:-8}
shift 2 2> /dev/null || true
mkdir -p "$OUT"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export HSA_ENABLE_IPC_MODE_LEGACY=0
SUMMARY=$OUT/summary.jsonl
: > "$SUMMARY"
PORT=29600
for N in 1 2 4 8; do
  [ "$N" -gt "$MAXN" ] && break
  for EX in auto stats; do
    PORT=$((PORT + 1))
    if [ "$N" -eq 1 ]; then
      # the N = 1 point of the curve is the driver's plain `python bench.py --gpus 1`; beside it the multi-rank program
      # with one rank (RCCL world 1), which is what N > 1 runs per rank
      [ "$EX" = "auto" ] && python "$ROOT/bench.py" --gpus 1 --headline-only --no-cpu-baseline --traffic none --detail "$OUT/n1_default_detail.json" > "$OUT/n1_default.json" 2> "$OUT/n1_default.err"
      MASTER_PORT=$PORT PRT_BENCH_WATCHDOG=${PRT_BENCH_WATCHDOG:-600} python "$ROOT/bench.py" --force-multi --exchange $EX --detail "$OUT/n1_${EX}_detail.json" "$@" > "$OUT/n1_$EX.json" 2> "$OUT/n1_$EX.err"
    else
      MASTER_PORT=$PORT PRT_BENCH_WATCHDOG=${PRT_BENCH_WATCHDOG:-600} python "$ROOT/bench.py" --gpus $N --exchange $EX --detail "$OUT/n${N}_${EX}_detail.json" "$@" > "$OUT/n${N}_$EX.json" 2> "$OUT/n${N}_$EX.err"
    fi
    echo "rc=$?" >> "$OUT/n${N}_$EX.err"
    python - "$OUT/n${N}_$EX.json" $N $EX >> "$SUMMARY" <<'PY'
import json, sys
(path, n, ex) = sys.argv[1:4]
try:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    c = d.get("config", {})
    print(json.dumps({"n_gpus": int(n), "exchange_asked": ex, "exchange": c.get("exchange"), "probe_ms": c.get("exchange_probe_ms"),
                      "rccl_world": c.get("rccl_world"), "value": d.get("value"), "ms_total": c.get("ms_total"),
                      "ms_trace": c.get("ms_trace"), "ms_gather": c.get("ms_gather"),
                      "value_without_gather": c.get("value_without_gather"),
                      "ok_per_rank": (d.get("verified") or {}).get("ok_per_rank"), "scaling_point": d.get("scaling_point"),
                      "error": d.get("error")}))
except Exception as exc:
    print(json.dumps({"n_gpus": int(n), "exchange_asked": ex, "error": "no line: %s" % exc}))
PY
  done
done
cat "$SUMMARY"
