"""Build the in-tree HIP library: hipcc --offload-arch=gfx950 -> pyrate_amd/csrc/libprt.so."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["prt.hip"]
HEADERS = ["prt_kernels.h", "prt_device.h", "prt_aniso.h", os.path.join("..", "..", "include", "prt.h")]
OUT = os.path.join(CSRC, "libprt.so")
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
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
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
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
