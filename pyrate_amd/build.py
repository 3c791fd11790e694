"""Build the in-tree HIP library: hipcc --offload-arch=gfx950 -> pyrate_amd/csrc/libprt.so."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["prt.hip"]
HEADERS = ["prt_kernels.h", "prt_device.h", "prt_aniso.h", os.path.join("..", "..", "include", "prt.h")]
OUT = os.path.join(CSRC, "libprt.so")
# -ffp-contract=on: FMA contraction decided per source expression (the device default, "fast", lets the
# optimiser contract across statements, and it did so differently for the two rays a march thread
# owns: 37 % of the double Gauss rays differed by up to 2 ulp depending on whether their global index was
# even or odd).  With "on" a ray's result does not depend on its position in the bundle, so a sharded
# trace equals the unsharded one bit for bit whatever the shard boundaries; no measurable cost
# (scratch/slot_probe.py, scratch/ab_same_buffers.py).
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=on",
               "-Wall", "-Wno-unused-function"]


def find_hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_all(force=False, verbose=True):
    """Compile libprt.so for gfx950 (cross-compiles without a GPU)."""
    if not force and not needs_build():
        return OUT
    hipcc = find_hipcc()
    if hipcc is None:
        if os.path.exists(OUT):
            return OUT          # prebuilt .so travelled with the snapshot
        raise RuntimeError("hipcc not found and %s is not built" % OUT)
    cmd = [hipcc] + HIPCC_FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    return OUT


if __name__ == "__main__":
    build_all(force=True)
