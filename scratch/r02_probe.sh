#!/bin/bash
# round 2, call 1: system info + VMM placement probe + first-allocation bench
O=gpurun_out/r02a; mkdir -p $O
{
  echo "== rocm-smi partitions"; rocm-smi --showmemorypartition --showcomputepartition 2>&1 | tail -12
  echo "== sysfs"; for f in /sys/class/drm/card*/device/current_memory_partition /sys/class/drm/card*/device/current_compute_partition /sys/class/drm/card*/device/available_memory_partition /sys/class/drm/card*/device/mem_info_vram_total /sys/class/drm/card*/device/mem_info_vram_used; do echo "$f: $(cat $f 2>&1)"; done
  echo "== kfd mem banks"; for f in /sys/class/kfd/kfd/topology/nodes/*/mem_banks/*/properties; do echo "$f"; cat $f 2>&1; done
  echo "== debugfs"; ls /sys/kernel/debug 2>&1 | head; ls /sys/kernel/debug/dri 2>&1 | head; mount | grep -i debug
  mount -t debugfs none /sys/kernel/debug 2>&1 | head -2; ls /sys/kernel/debug/dri 2>&1 | head
  for d in /sys/kernel/debug/dri/*; do echo "$d"; ls $d 2>&1 | tr '\n' ' ' | head -c 1500; echo; done
  echo "== rocminfo pools"; rocminfo 2>&1 | grep -i -A6 'pool' | head -60
} > $O/sysinfo.txt 2>&1
timeout 420 ./benchmarks/vmm_placement_probe 224 2 > $O/vmm_probe.txt 2>&1; echo "probe rc $?" >> $O/sysinfo.txt
for d in /sys/kernel/debug/dri/*; do [ -r $d/amdgpu_vram_mm ] && head -c 20000 $d/amdgpu_vram_mm > $O/vram_mm_$(basename $d).txt 2>&1; done
for i in 1 2 3; do timeout 300 python bench.py --placement-candidates 1 --steps 20 --warmup 10 --no-cpu-baseline > $O/bench_first_alloc_$i.json 2> $O/bench_first_alloc_$i.err; done
echo done >> $O/sysinfo.txt
