#!/bin/bash
# A/B of the plugin-granular sweep: traversal direction x placement (each variant in a process of its own)
O=gpurun_out/${1:-r06d}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for rep in 1 2; do
for v in "0 arena" "1 arena" "2 arena" "0 torch" "1 torch"; do
  set -- $v
  PRT_ROWS_PINGPONG=$1 python bench.py --configs plugin --placement $2 --steps 20 --warmup 5 --no-cpu-baseline --traffic none --detail /tmp/ab_detail.json 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pingpong $1 placement $2 rep $rep: ms %.4f frac %.4f ok %s' % (d['ms_per_step'], d['roofline']['frac'], d['verified']['ok']))"
done; done | tee $O/ab_plugin.txt
