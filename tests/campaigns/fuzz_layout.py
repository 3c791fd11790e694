"""one-off sweep of buffer layouts through the C ABI: ray counts 1..700, row pitches (tight, +1, +7,
recommended), base pointers offset by 8 bytes (defeats the 16-B vector path), packed flags or not,
path / image mode -- all must equal the tight reference launch bit for bit"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch, ctypes
from pyrate_amd import engine, systems, _lib
dev = torch.device("cuda", 0)
lib = _lib.load()
recs = systems.double_gauss_records()
sysd = engine.DeviceSystem(recs, 0)
S = len(recs)
rng = np.random.RandomState(5)
bad = []; ncase = 0

def run(n, in_pitch, out_pitch, in_off, out_off, mode, flags, X, K, E):
    def alloc(rows, pitch, off, dtype=torch.float64):
        buf = torch.full((rows * pitch + off + 4,), float("nan") if dtype == torch.float64 else 7, dtype=dtype, device=dev)
        return buf, buf[off:]
    (xb, xv) = alloc(3, in_pitch, in_off); (kb, kv) = alloc(3, in_pitch, in_off); (eb, ev) = alloc(3, in_pitch, in_off)
    for (v, a) in ((xv, X), (kv, K), (ev, E)):
        v[:3 * in_pitch].view(3, in_pitch)[:, :n] = a
    rows = S if mode == 0 else 1
    (hb, hv) = alloc(3 * rows, out_pitch, out_off); (ob, ov) = alloc(3 * rows, out_pitch, out_off)
    (vb, vv) = alloc(rows, out_pitch, 2 * out_off, torch.uint8); (wb, wv) = alloc(rows, out_pitch, 2 * out_off, torch.uint8)
    rc = lib.prt_trace(sysd._h, n, in_pitch, xv.data_ptr(), kv.data_ptr(), ev.data_ptr(), None, mode | (2 if flags else 0),
                       out_pitch, hv.data_ptr(), ov.data_ptr(), vv.data_ptr(), None if flags else wv.data_ptr(), None)
    assert rc == 0, _lib.load().prt_last_error()
    torch.cuda.synchronize()
    xh = hv[:3 * rows * out_pitch].view(rows, 3, out_pitch)[:, :, :n].cpu().numpy()
    ko = ov[:3 * rows * out_pitch].view(rows, 3, out_pitch)[:, :, :n].cpu().numpy()
    va = vv[:rows * out_pitch].view(rows, out_pitch)[:, :n].cpu().numpy()
    wa = wv[:rows * out_pitch].view(rows, out_pitch)[:, :n].cpu().numpy()
    if flags:
        (va, wa) = (va & 1, va >> 1)
    return xh, ko, va, wa

for n in list(range(1, 40)) + [63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 699, 700]:
    (o, k, e0) = systems.double_gauss_bundle(max(n * 2, 50), rpup=9.0, field_deg=3.0)
    sel = rng.permutation(o.shape[1])[:n]
    if len(sel) < n: continue
    (X, K, E) = [torch.from_numpy(np.ascontiguousarray(a[:, sel])).to(dev) for a in (o, k, e0)]
    ref = {m: run(n, n, n, 0, 0, m, False, X, K, E) for m in (0, 1)}
    rp = engine.recommended_pitch(n)
    for (in_pitch, out_pitch, in_off, out_off, flags) in [(n, n, 1, 0, False), (n, n, 0, 1, False), (n + 1, n + 7, 0, 0, False),
                                                          (rp, rp, 0, 0, False), (rp, rp, 0, 0, True), (n, n, 0, 0, True),
                                                          (n + 3, rp, 1, 1, True), (rp, n + 1, 0, 0, True)]:
        for mode in (0, 1):
            ncase += 1
            got = run(n, in_pitch, out_pitch, in_off, out_off, mode, flags, X, K, E)
            for (a, b, name) in zip(got, ref[mode], ("x", "k", "valid", "valid_out")):
                same = np.array_equal(a.view(np.int64) if a.dtype == np.float64 else a, b.view(np.int64) if b.dtype == np.float64 else b)
                if not same:
                    bad.append((n, in_pitch, out_pitch, in_off, out_off, flags, mode, name)); break
print("layout cases:", ncase, " failures:", len(bad))
for b in bad[:20]: print(b)
