import sys, torch
sys.path.insert(0, '.')
from pyrate_amd import distributed as pdist, engine, systems, _lib
dev = torch.device("cuda", 0)
sysd = engine.DeviceSystem(systems.double_gauss_records(), 0)
(x0, k0, e0d, n) = systems.double_gauss_bundle_device(10000000, dev, field_deg=2.0)
whole = sysd.trace(x0, k0, e0d, mode=_lib.MODE_IMAGE, packed_flags=True)
def same(a, b):
    return torch.equal(a.contiguous().view(torch.int64), b.contiguous().view(torch.int64))
for r in range(8):
    (lo, hi) = pdist.shard_range(n, r, 8)
    (xs, ks, es, total) = systems.double_gauss_bundle_device(10000000, dev, field_deg=2.0, lo=lo, hi=hi)
    part = sysd.trace(xs, ks, es, mode=_lib.MODE_IMAGE, packed_flags=True)
    part2 = sysd.trace(x0[:, lo:hi].contiguous(), k0[:, lo:hi].contiguous(), e0d[:, lo:hi].contiguous(), mode=_lib.MODE_IMAGE, packed_flags=True)
    wx = whole.x_hit[0][:, lo:hi]
    d = (part.x_hit[0] - wx).abs()
    print(r, lo, hi, "inputs same:", same(xs, x0[:, lo:hi]), same(ks, k0[:, lo:hi]), same(es, e0d[:, lo:hi]),
          "| x same:", same(part.x_hit[0], wx), "sliced-input x same:", same(part2.x_hit[0], wx),
          "max|dx| %.3e" % float(d.max()), "n diff", int((d.max(dim=0).values > 0).sum()),
          "k same:", same(part.k_out[0], whole.k_out[0][:, lo:hi]), "v same:", torch.equal(part.valid_out[0], whole.valid_out[0][lo:hi]))
