"""headline march vs row pitch: the 36 + 36 + 12 output rows are `pitch` elements apart; does the HBM channel /
bank mapping prefer some pitches?  Same process, same inputs, arena-placed outputs for every pitch."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyrate_amd import engine, systems, _lib
dev = torch.device("cuda", 0)
sysd = engine.DeviceSystem(systems.double_gauss_records(), 0)
(x0, k0, e0d, n) = systems.double_gauss_bundle_device(10000000, dev)
pitch_in = x0.stride(0)
st = engine._stream_handle(dev); P = engine._ptr
base = engine.recommended_pitch(n)
out = {"n": n, "base_pitch": base, "ms": {}}
offsets = [0] + [k * 131072 for k in range(1, 25)] + [1048576 + j * 32768 for j in (1, 2, 3, 5, 7)] + [0]
def run(pitch):
    bufs = sysd.alloc_outputs(n, packed_flags=True, pitch=pitch)
    ms = ctypes.c_double()
    res = []
    for rep in range(3):
        rc = sysd.lib.prt_trace_timed(sysd._h, n, pitch_in, P(x0), P(k0), P(e0d), None, engine._mode_word(bufs), bufs["pitch"], P(bufs["x_hit"]), P(bufs["k_out"]), P(bufs["valid"]), None, st, 20, ctypes.byref(ms))
        assert rc == 0
        res.append(round(ms.value, 4))
    kinds = bufs["placement"].get("kinds")
    del bufs
    return res, kinds
run(base)
for rnd in range(2):
    for off in offsets:
        (res, kinds) = run(base + off)
        out["ms"].setdefault(str(off), []).append({"ms": res, "kinds": kinds})
print(json.dumps(out))
