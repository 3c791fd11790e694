export TMPDIR=/tmp
mkdir -p gpurun_out/valu
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/valu/a -- python benchmarks/valu_profile.py run > gpurun_out/valu/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/valu/b -- python benchmarks/valu_profile.py run > gpurun_out/valu/b.log 2>&1
tail -3 gpurun_out/valu/a.log gpurun_out/valu/b.log
python benchmarks/valu_profile.py report gpurun_out/valu > gpurun_out/valu/valu_profile.json
find gpurun_out/valu -name "*counter_collection.csv" | while read f; do head -60 "$f" > gpurun_out/valu/$(basename $(dirname $(dirname $f)))_counter_head.csv; done
find gpurun_out/valu -name "*.csv" -size +1M -delete
cat gpurun_out/valu/valu_profile.json | head -150
