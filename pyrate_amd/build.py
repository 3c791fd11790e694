"""Build the in-tree HIP library: hipcc --offload-arch=gfx950 -> pyrate_amd/csrc/libprt.so."""
import hashlib
import json
import os
import shutil
import subprocess
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["prt.hip"]
HEADERS = ["prt_kernels.h", "prt_device.h", "prt_aniso.h", "prt_aniso_cplx.h", "prt_placed.h", os.path.join("..", "..", "include", "prt.h")]
OUT = os.path.join(CSRC, "libprt.so")
INFO = os.path.join(CSRC, "libprt.build.json")     # written by the build, travels with the .so
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
    """Compile libprt.so for gfx950 (cross-compiles without a GPU).  ``force`` rebuilds whenever a
    compiler is there (what ``__graft_entry__.build()`` asks for: the build check must compile, not
    find a file); otherwise only when a source is newer than the library."""
    hipcc = find_hipcc()
    if hipcc is None:
        if os.path.exists(OUT):
            return OUT          # prebuilt .so travelled with the snapshot, no compiler on this box
        raise RuntimeError("hipcc not found and %s is not built" % OUT)
    if not force and not needs_build():
        return OUT
    cmd = [hipcc] + HIPCC_FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    with open(INFO, "w") as f:
        json.dump({"hipcc": hipcc_version(hipcc), "flags": HIPCC_FLAGS, "sha256_16": library_hash(),
                   "built_unix": int(time.time())}, f)
    return OUT


def hipcc_version(hipcc=None):
    hipcc = hipcc or find_hipcc()
    if hipcc is None:
        return None
    out = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout
    for line in out.splitlines():
        if line.startswith("HIP version"):
            return line.split(":", 1)[1].strip()
    return out.splitlines()[0].strip() if out else None


def library_hash(path=None):
    """first 16 hex digits of the SHA-256 of libprt.so (None if it is not built); ``path``: another build
    of the library"""
    path = path or OUT
    if not os.path.exists(path):
        return None
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for block in iter(lambda: f.read(1 << 20), b""):
            h.update(block)
    return h.hexdigest()[:16]


def build_info(loaded_path=None):
    """what bench.py puts into its JSON line: hash of the library that is loaded (``loaded_path``: the file
    the process really mapped, if it is not the in-tree one -- PRT_LIBRARY), and the compiler that produced
    it (from the side file the build writes; None if the library came from elsewhere)"""
    info = {"libprt_sha256_16": library_hash(loaded_path), "hipcc": None}
    if loaded_path and os.path.abspath(loaded_path) != os.path.abspath(OUT):
        info["library"] = os.path.relpath(loaded_path, os.path.dirname(HERE))
    try:
        with open(INFO) as f:
            rec = json.load(f)
        if rec.get("sha256_16") == info["libprt_sha256_16"]:
            info["hipcc"] = rec.get("hipcc")
            info["built_unix"] = rec.get("built_unix")
    except (OSError, ValueError):
        pass
    return info


if __name__ == "__main__":
    build_all(force=True)
