#!/bin/bash
O=gpurun_out/r02d; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -s -k "arena" > $O/pytest_arena.log 2>&1
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
for i in 1 2 3 4 5; do timeout 300 python bench.py --no-cpu-baseline > $O/bench_arena_$i.json 2> $O/bench_arena_$i.err; done
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --placement torch > $O/bench_torch_$i.json 2> $O/bench_torch_$i.err; done
timeout 300 python scratch/r02_place_exp.py > $O/place_exp.json 2> $O/place_exp.err
timeout 300 python scratch/r02_dropin_time.py > $O/dropin.json 2> $O/dropin.err
timeout 300 python bench.py --config asphere --no-cpu-baseline > $O/bench_asphere.json 2> $O/bench_asphere.err
timeout 300 python bench.py --config aniso --no-cpu-baseline > $O/bench_aniso.json 2> $O/bench_aniso.err
timeout 300 python bench.py --force-multi --steps 10 --warmup 3 > $O/bench_force_multi.json 2> $O/bench_force_multi.err

timeout 400 python bench.py > $O/bench_full.json 2> $O/bench_full.err
echo done > $O/done.txt
