#!/bin/bash
# What round 6 could not run (the GPU pool was closed to this repository): in this order, each step bounded.
#   bash benchmarks/first_gpu_contact.sh [tag]      -> gpurun_out/<tag>/...
# 1. collect_profiles.sh: the default bench line as the first GPU process + rocprofv3 kernel trace + fresh-process repeats
#    (expect aniso_biaxial near 0.15-0.16 ms instead of r06b's 0.227: DESIGN.md 8; a tenth record `surface_step`)
# 2. the first-contact tests (fused surface step through engine.py, rotated-tensor identity, scale preflight dry run)
# 3. the fused surface step against the pair of calls, with its own PMC pass
# 4. the whole -m gpu suite, smoke()
TAG=${1:-first_contact}
O=gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 1200 bash benchmarks/collect_profiles.sh "$TAG" > "$O/collect.log" 2>&1; echo "collect rc=$?"
head -c 2600 "$O/bench_line.json"; echo
timeout 900 python -m pytest tests/test_gpu_zz_first_contact.py -x -q -m gpu > "$O/pytest_first_contact.log" 2>&1; echo "first-contact tests rc=$?"; tail -15 "$O/pytest_first_contact.log"
timeout 400 python bench.py --configs surface_step,plugin --steps 20 --warmup 5 --cpu-budget 0.5 --detail "$O/surface_step_detail.json" > "$O/surface_step_line.json" 2> "$O/surface_step.err"; echo "surface_step bench rc=$?"
head -c 2500 "$O/surface_step_line.json"; echo
timeout 2700 python -m pytest tests -x -q -m gpu > "$O/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -8 "$O/pytest_gpu.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1; echo "smoke rc=$?"; tail -2 "$O/smoke.log"
