#!/bin/bash
# packed mask flags (49 B per record) vs two mask arrays (50 B), interleaved on one box
for rep in 1 2 3; do
  python bench.py --no-cpu-baseline --steps 100 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('packed   ', d['roofline']['kernel_ms'], d['value'], d['roofline']['achieved'])"
  python bench.py --no-cpu-baseline --steps 100 --two-mask-arrays | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('two masks', d['roofline']['kernel_ms'], d['value'], d['roofline']['achieved'])"
done
