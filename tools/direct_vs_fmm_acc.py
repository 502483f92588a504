#!/usr/bin/env python3
"""Where the direct and the FMM accelerations of one state differ most (GPU box): python tools/direct_vs_fmm_acc.py [n]"""
import sys, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from coulomb_oscillators_amd import Engine, EVAL_DIRECT, EVAL_FMM_KDTREE
from bench import gaussian_ball, coulomb_params
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
buf = gaussian_ball(n, 7); par = coulomb_params(n)
prm = torch.from_numpy(par).cuda()
acc = {}
for kind, name, opts in [(EVAL_DIRECT, "direct", {}), (EVAL_FMM_KDTREE, "fmm10", dict(fmm_order=10, unsort=1)), (EVAL_FMM_KDTREE, "fmm6", dict(fmm_order=6, unsort=1))]:
    e = Engine(sync=0, **opts)
    d = torch.from_numpy(buf.copy()).cuda()
    a = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
    if kind == EVAL_DIRECT: e.direct(d[0], a, n, prm)
    else: e.fmm_cart3_kdtree(d, a, n, prm)
    assert torch.equal(d[0].cpu(), torch.from_numpy(buf[0])), "state permuted"
    acc[name] = a.cpu().numpy().astype(np.float64)
    e.close()
x = buf[0].astype(np.float64)
# fp64 reference for the worst particles
def exact(i):
    dd = x[i] - x
    r2 = (dd * dd).sum(1); r2[i] = np.inf
    return float(par[0]) * (dd / r2[:, None] ** 1.5).sum(0), np.sqrt(r2.min())
mag = np.linalg.norm(acc["direct"], axis=1)
for name in ("fmm10", "fmm6"):
    err = np.linalg.norm(acc["direct"] - acc[name], axis=1) / (mag + mag.mean())
    order = np.argsort(-err)[:5]
    print(name, "particles with err > 1e-4:", int((err > 1e-4).sum()), "> 1e-5:", int((err > 1e-5).sum()))
    for i in order:
        ex, rmin = exact(i)
        print("  i=%6d err=%.3e |a_direct|=%.4e |a_%s|=%.4e |a_fp64|=%.4e  |direct-fp64|=%.3e |%s-fp64|=%.3e  nearest=%.3e" % (
            i, err[i], mag[i], name, np.linalg.norm(acc[name][i]), np.linalg.norm(ex), np.linalg.norm(acc["direct"][i] - ex), name, np.linalg.norm(acc[name][i] - ex), rmin))
