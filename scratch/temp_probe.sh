rocm-smi --showtemp --showpower 2>&1 | grep -i "temp\|power" | head -8
python bench.py --steps 200 --warmup 10 --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms %.4f kernel %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))" &
BP=$!
sleep 9
for i in 1 2 3; do rocm-smi --showtemp --showpower --showclocks 2>&1 | grep -i "temp\|Socket Power\|mclk\|fclk\|sclk" | tr '\n' ';' | cut -c1-600; echo; sleep 0.5; done
wait $BP
amd-smi metric -t -p 2>/dev/null | head -40
