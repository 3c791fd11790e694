#!/bin/bash
# interleaved A/B of libprt builds on the same box: default vs every library given
for rep in 1 2; do
  python bench.py --no-cpu-baseline --steps 100 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('default ', d['roofline']['kernel_ms'], d['ms_per_step'])"
  for lib in "$@"; do
    PRT_LIBRARY=$lib python bench.py --no-cpu-baseline --steps 100 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$lib ', d['roofline']['kernel_ms'], d['ms_per_step'])"
  done
done
