#!/usr/bin/env python3
"""The direct and the kd-tree FMM evaluator side by side through nbco_integrate (GPU box): accelerations on one state, then the
energies of both runs every `every` steps.  python tools/direct_vs_fmm.py [n] [steps] [every]"""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from coulomb_oscillators_amd import Engine, EVAL_DIRECT, EVAL_FMM_KDTREE, INTEG_LEAPFROG
from bench import gaussian_ball, coulomb_params
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
every = int(sys.argv[3]) if len(sys.argv) > 3 else 100
buf = gaussian_ball(n, 7); par = coulomb_params(n)
prm = torch.from_numpy(par).cuda()
runs = []
for kind, name, opts in [(EVAL_DIRECT, "direct", {}), (EVAL_FMM_KDTREE, "fmm10", dict(fmm_order=10, unsort=1))]:
    e = Engine(sync=0, **opts)
    d = torch.from_numpy(buf.copy()).cuda()
    e.compute_force(kind, d, n, prm)
    runs.append((name, kind, e, d))
a0, a1 = runs[0][3][2], runs[1][3][2]
mag = a0.norm(dim=1)
print("acc: max |direct - fmm10| / (|a| + mean|a|) = %.3e" % float(((a0 - a1).norm(dim=1) / (mag + mag.mean())).max()), flush=True)
for k in range(0, steps + 1, every):
    row = []
    for name, kind, e, d in runs:
        E = e.energy(d, n, prm)
        row.append("%s E=%.6e (parts %s)" % (name, sum(E), " ".join("%.4e" % x for x in E)))
    x0, x1 = runs[0][3][0], runs[1][3][0]
    print("step %5d  %s | %s | max|dx| = %.3e" % (k, row[0], row[1], float((x0 - x1).abs().max())), flush=True)
    if k == steps:
        break
    for name, kind, e, d in runs:
        for _ in range(every):
            e.integrate(INTEG_LEAPFROG, kind, d, n, prm, 5e-4)
