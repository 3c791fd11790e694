#!/usr/bin/env python
"""What the compiler made of the march kernels, from the gfx950 assembly (no GPU needed):

    python benchmarks/isa_report.py > profiles/<tag>_isa_report.json

Per instantiation of k_trace_iso / k_trace_general / k_propagate_rows / k_interact_iso_rows / k_surface_step_rows: VGPRs, scratch bytes per lane, waves per SIMD, LDS bytes
per block (hipcc -Rpass-analysis=kernel-resource-usage), and from the instruction stream
  flat_memory_ops        flat_load / flat_store (a pointer whose address space the compiler could not see:
                         per-lane vector memory traffic + lgkmcnt AND vmcnt waits)
  vmcnt_waits_in_loops   s_waitcnt vmcnt(n) inside a loop.  gfx950 counts loads and stores in ONE in-order
                         counter, so such a wait behind the first surface waits for path stores
  scalar_base_stores / vector_address_stores   global_store with an SGPR base + 32-bit lane offset vs a
                         64-bit address per lane
DESIGN.md 4 "One counter for loads and stores" is the story behind these columns; tests/test_kernel_isa.py
pins the figures of the kernels the BASELINE configs run.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pyrate_amd import build as prt_build

# the marches and the per-surface kernels of big bundles (k_propagate_rows / k_interact_iso_rows)
MARCH = ("_Z11k_trace_iso", "_Z15k_trace_general", "_Z16k_propagate_rows", "_Z19k_interact_iso_rows",
         "_Z19k_surface_step_rows")


def template_args(mangled):
    """'_Z11k_trace_isoILi0ELb1ELb1ELi0ELb0ELb0EEv...' -> 'k_trace_iso<0,1,1,0,0,0>'"""
    m = re.match(r"_Z\d+(k_[a-z_]+)I((?:L[ib]\d+E)+)E", mangled)
    if not m:
        return mangled
    return "%s<%s>" % (m.group(1), ",".join(re.findall(r"L[ib](\d+)E", m.group(2))))


def report(hipcc=None):
    hipcc = hipcc or prt_build.find_hipcc()
    if hipcc is None:
        raise RuntimeError("hipcc not found")
    flags = [f for f in prt_build.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "prt.s")
        cmd = [hipcc] + flags + ["--cuda-device-only", "-S", os.path.join(prt_build.CSRC, "prt.hip"), "-o", asm,
                                 "-Rpass-analysis=kernel-resource-usage"]
        res = subprocess.run(cmd, capture_output=True, text=True, cwd=prt_build.CSRC)
        if res.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + res.stderr[-2000:])
        with open(asm) as f:
            lines = f.read().splitlines()
    kernels = {}
    for blk in res.stderr.split("Function Name: ")[1:]:
        name = blk.split()[0]
        if not name.startswith(MARCH):
            continue

        def num(pat):
            m = re.search(pat, blk)
            return int(m.group(1)) if m else None
        kernels[name] = {"kernel": template_args(name), "vgprs": num(r" VGPRs: (\d+)"),
                         "scratch_bytes_per_lane": num(r"ScratchSize \[bytes/lane\]: (\d+)"),
                         "waves_per_simd": num(r"Occupancy \[waves/SIMD\]: (\d+)"),
                         "lds_bytes_per_block": num(r"LDS Size \[bytes/block\]: (\d+)")}
    cur = None
    for ln in lines:
        m = re.match(r"^(_Z\d+k_[A-Za-z0-9_]+):", ln)
        if m:
            cur = kernels.get(m.group(1))
            if cur is not None:
                cur.update({"flat_memory_ops": 0, "vmcnt_waits_in_loops": 0, "scalar_base_stores": 0,
                            "vector_address_stores": 0, "scalar_loads": 0, "instructions": 0})
                in_loop = False
            continue
        if cur is None:
            continue
        if ln.startswith(".Lfunc_end"):
            cur = None
            continue
        if re.match(r"^\.LBB\d+_\d+:", ln):
            in_loop = False          # the block's loop comment (if any) follows on the next lines
        if "in Loop:" in ln or "Loop Header" in ln:
            in_loop = True
        t = ln.strip()
        if not t or t.startswith((";", ".")):
            continue
        cur["instructions"] += 1
        if t.startswith(("flat_load", "flat_store")):
            cur["flat_memory_ops"] += 1
        elif t.startswith("s_waitcnt") and "vmcnt" in t and in_loop:
            cur["vmcnt_waits_in_loops"] += 1
        elif t.startswith("global_store"):
            cur["scalar_base_stores" if re.search(r", s\[\d+:\d+\]", t) else "vector_address_stores"] += 1
        elif t.startswith("s_load"):
            cur["scalar_loads"] += 1
    return {"hipcc": prt_build.hipcc_version(hipcc), "flags": flags,
            "kernels": sorted(kernels.values(), key=lambda k: k["kernel"])}


if __name__ == "__main__":
    json.dump(report(), sys.stdout, indent=1)
    print()
