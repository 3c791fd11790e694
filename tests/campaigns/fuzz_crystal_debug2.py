import sys, math
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
np.set_printoptions(precision=12, linewidth=220)
src = open('tests/campaigns/fuzz_crystal_stress.py').read()
exec(src[:src.index("bad = []; ntot = 0")])
body = src[src.index("    rng = np.random.RandomState(9000 + seed)"):src.index("    try:\n        with np.errstate")]
seed = 240
exec("if True:\n" + body)
from pyrate_amd.surface_table import classify_eps
with np.errstate(all="ignore"):
    out = oracle.trace(recs, x0, k0, e0)
sysd = engine.DeviceSystem(recs, 0)
res = sysd.trace(*[engine.to_device_rays(a, dev, pitched=False) for a in (x0, k0, e0)])
s = 2
rec = recs[s]
eps = np.asarray(rec["material"]["eps_re"])
print("class", classify_eps(rec["material"]["eps_re"], rec["material"]["eps_im"]), "interaction", rec["interaction"])
xh = out[s]["x_hit"]            # (3, n_in)
kin_glob = np.real(out[s-1]["k_out"])
n_in = xh.shape[1]
# normal in material frame like surface_step
Bs = np.asarray(rec["B_shape"]); gs = np.asarray(rec["g_shape"]); Bm = np.asarray(rec["B_mat"])
p = oracle.g2l_points(Bs, gs, xh)
nl = oracle.shape_normal(rec["shape"], p[0], p[1])
nm = Bm.T.dot(Bs.dot(nl))
k1 = Bm.T.dot(kin_glob)
kd = res.k_out[s].cpu().numpy(); ko = np.real(out[s]["k_out"])
err = np.abs(kd - ko).max(axis=0)
idx = np.where(err > 1e-6)[0]
print("mismatching outputs:", len(idx), "of", kd.shape[1], idx[:10])
for j in idx[:3]:
    i = j % n_in
    kpa = k1[:, i] - np.dot(k1[:, i], nm[:, i]) * nm[:, i]
    (xi, ev) = oracle.aniso_xi_efield(nm[:, i], kpa, eps)
    k4 = kpa[None, :] + xi[:, None] * nm[:, i][None, :]
    sn = np.array([np.sum(oracle.poynting_norm(k4[q], ev[q]) * nm[:, i]) for q in range(4)])
    print(" out slot", j, "(ray", i, "branch", j // n_in, ")  kpa^2", np.dot(kpa, kpa))
    print("  oracle xi", xi, "\n  S.n", sn, " order", sn.argsort())
    print("  k oracle", ko[:, j], " k hip", kd[:, j])
