#!/bin/bash
# sample fclk / mclk / sclk / power while the bench kernel runs
python bench.py --no-cpu-baseline --steps 6000 --warmup 10 > /tmp/bench_out.json 2>/dev/null &
BP=$!
sleep 6.5
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "fclk|mclk|sclk|Power" | tr '\n' ' '; echo
  sleep 0.4
done
wait $BP
python -c "import json; d=json.loads(open('/tmp/bench_out.json').readline()); print('kernel_ms', d['roofline']['kernel_ms'], 'ms_per_step', d['ms_per_step'])"
rocm-smi --showuniqueid 2>/dev/null | grep Unique
rocm-smi --showclkfrq 2>/dev/null | grep -E "fclk|mclk|^GPU|\*" | head -30
