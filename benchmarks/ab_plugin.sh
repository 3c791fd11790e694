#!/bin/bash
# A/B of the plugin-granular sweep (DeviceSystem.propagate + interact per surface): arena-placed arrays (every kernel
# reads two kinds of HBM and writes the third) against torch-allocated ones, each variant in a process of its own.
#   bash benchmarks/ab_plugin.sh <tag>   -> gpurun_out/<tag>/ab_plugin.txt
# (round 6 also ran it with a traversal-direction switch, PRT_ROWS_PINGPONG, that has been removed since:
#  profiles/r06d_ab_plugin_traversal_and_placement.txt)
O=gpurun_out/${1:-r06d}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for rep in 1 2; do
for placement in arena torch; do
  python bench.py --configs plugin --placement $placement --steps 20 --warmup 5 --no-cpu-baseline --traffic none --detail /tmp/ab_detail.json 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('placement $placement rep $rep: ms %.4f frac %.4f ok %s' % (d['ms_per_step'], d['roofline']['frac'], d['verified']['ok']))"
done; done | tee $O/ab_plugin.txt
