#!/usr/bin/env python3
"""Energy drift of the soak run's N = 65 536 case with the DIRECT evaluator (no tree, no expansions): what the integration
alone does to the energy of that realisation.  python tools/soak_direct.py [steps]"""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from coulomb_oscillators_amd import Engine, EVAL_DIRECT, EVAL_FMM_KDTREE, INTEG_LEAPFROG
from bench import gaussian_ball, coulomb_params
n, steps = 65536, int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for kind, name, opts in [(EVAL_DIRECT, "direct", {}), (EVAL_FMM_KDTREE, "fmm p=6", dict(fmm_order=6)), (EVAL_FMM_KDTREE, "fmm p=10", dict(fmm_order=10))]:
    buf = gaussian_ball(n, 7); par = coulomb_params(n)
    d = torch.from_numpy(buf).cuda(); prm = torch.from_numpy(par).cuda()
    e = Engine(unsort=0, sync=0, **opts)
    e.compute_force(kind, d, n, prm)
    E0 = sum(e.energy(d, n, prm))
    t0 = time.time()
    for k in range(steps):
        e.integrate(INTEG_LEAPFROG, kind, d, n, prm, 5e-4)
    torch.cuda.synchronize()
    E1 = sum(e.energy(d, n, prm))
    print(f"{name}: n={n} steps={steps} E0={E0:.9e} E1={E1:.9e} rel drift={(E1 - E0) / abs(E0):.2e} ({1e3 * (time.time() - t0) / steps:.3f} ms/step)", flush=True)
    e.close()
