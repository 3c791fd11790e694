"""package power / sclk while (a) a pure streaming write, (b) the compute-only image-mode march and
(c) the path-mode march run back to back for ~4 s each"""
import subprocess, sys, time, threading, torch
sys.path.insert(0, '.')
from pyrate_amd import engine, systems, _lib
dev = torch.device("cuda", 0)
sysd = engine.DeviceSystem(systems.double_gauss_records(), 0)
(x0, k0, e0, _) = systems.double_gauss_bundle_device(10_000_000, dev)
n = x0.shape[1]
bp = sysd.alloc_outputs(n, _lib.MODE_PATH)
bi = sysd.alloc_outputs(n, _lib.MODE_IMAGE)
big = torch.empty(6 * 1024 ** 3 // 8, dtype=torch.float64, device=dev)

def smi():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    sclk = [l for l in out.splitlines() if "sclk" in l][0].split("(")[1].split(")")[0]
    pw = [l for l in out.splitlines() if "Power (W)" in l][0].split(":")[-1].strip()
    return sclk, pw

def run(label, fn, seconds=4.0):
    samples = []
    stop = [False]
    def sampler():
        time.sleep(1.5)
        while not stop[0]:
            samples.append(smi())
            time.sleep(0.3)
    th = threading.Thread(target=sampler); th.start()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); it = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(50): fn()
        torch.cuda.synchronize(); it += 50
    dt = (time.perf_counter() - t0) / it * 1e3
    stop[0] = True; th.join()
    print("%-28s %.4f ms/launch   samples (sclk, W): %s" % (label, dt, samples[:6]))

for rep in range(2):
    run("fill_ 6 GiB (pure write)", lambda: big.fill_(1.0))
    run("image mode (compute only)", lambda: sysd.trace_into(x0, k0, bi, e0))
    run("path mode", lambda: sysd.trace_into(x0, k0, bp, e0))
    time.sleep(2.0)
