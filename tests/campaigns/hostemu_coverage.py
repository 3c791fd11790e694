"""One-off (not collected by pytest): LINE COVERAGE of libprt's sources by the host-build tests.  Builds
tests/hostemu/hostemu_prt.cpp with clang++ --coverage, runs tests/test_hostemu.py + tests/test_hostemu_campaigns.py on that
library, reads the counts with gcov -- which kernel code the CPU-only suite executes, and which it does not.  This is how
round 6 found that the crystal solver's adjugate-flux route was dead for every tensor that is symmetric only up to rounding.

    python tests/campaigns/hostemu_coverage.py [workdir] > profiles/<tag>_hostemu_line_coverage.txt      (about 2 minutes)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hostemu          # noqa: E402

work = sys.argv[1] if len(sys.argv) > 1 else "/tmp/prt_hostemu_cov"
os.makedirs(work, exist_ok=True)
for f in os.listdir(work):
    if f.endswith((".gcda", ".gcno", ".gcov")):
        os.remove(os.path.join(work, f))
lib = os.path.join(work, "libprt_hostemu_cov.so")
subprocess.run([hostemu.find_clang(), "-x", "c++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "--coverage", "-I" + hostemu.HERE,
                "-ffp-contract=on", "-fno-math-errno", "-Wno-unknown-attributes", "-Wno-unused-function",
                os.path.join(hostemu.HERE, "hostemu_prt.cpp"), "-o", lib], check=True, cwd=work)
env = dict(os.environ, PRT_HOSTEMU_LIBRARY=lib)
r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_hostemu.py"),
                    os.path.join(ROOT, "tests", "test_hostemu_campaigns.py"), "-q", "-p", "no:cacheprovider",
                    "-k", "not sanitizers and not overrun and not stand_in"], env=env, cwd=work, capture_output=True, text=True)
print("# line coverage of pyrate_amd/csrc by the host-build tests (tests/campaigns/hostemu_coverage.py; clang --coverage + gcov)")
print("# tests:", r.stdout.strip().splitlines()[-1])
gcda = [f for f in os.listdir(work) if f.endswith(".gcda")][0]
subprocess.run(["gcov", gcda], cwd=work, capture_output=True, text=True)
for name in ("prt_device.h", "prt_aniso.h", "prt_aniso_cplx.h", "prt_kernels.h", "prt.hip", "prt_placed.h"):
    (hit, miss) = (0, [])
    for line in open(os.path.join(work, name + ".gcov"), errors="replace"):
        m = re.match(r"\s*(#####|=====|\d+\*?|-):\s*(\d+):", line)
        if not m or m.group(1) == "-":
            continue
        if m.group(1) in ("#####", "====="):
            miss.append(int(m.group(2)))
        else:
            hit += 1
    miss = sorted(set(miss))
    ranges = []
    for n in miss:
        if ranges and n <= ranges[-1][1] + 2:
            ranges[-1][1] = n
        else:
            ranges.append([n, n])
    total = hit + len(miss)
    print("%-18s %5.1f %% of %4d lines" % (name, 100.0 * hit / max(total, 1), total)
          + ("   (the placement arena: no virtual-memory API on the host)" if name == "prt_placed.h" else ""))
    if name != "prt_placed.h":
        print("    not executed: " + (", ".join("%d" % a if a == b else "%d-%d" % (a, b) for (a, b) in ranges) or "-"))
