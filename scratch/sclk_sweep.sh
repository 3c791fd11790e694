# does the path-mode march depend on the shader clock?  (root on the box: rocm-smi perf determinism caps sclk)
export TMPDIR=/tmp
rocm-smi --showperflevel --showsclkrange 2>&1 | grep -v "^=\|^$" | head -8
for clk in 0 2100 1800 1500 1200; do
  if [ "$clk" != "0" ]; then rocm-smi --setperfdeterminism $clk 2>&1 | grep -i "success\|fail\|error\|denied" | head -2; fi
  echo "== sclk cap $clk"
  python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('path  ms %.4f  frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))"
  ./scratch/exp_rpt | head -1
done
rocm-smi --resetperfdeterminism 2>&1 | grep -i "success\|fail" | head -2
