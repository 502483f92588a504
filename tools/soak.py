#!/usr/bin/env python3
"""Long leapfrog runs on the GPU (finite state, energy drift, list growth): python tools/soak.py
N = 65 536: 3000 steps with a rebuild every step and with tree_steps = 8 (total energy before / after, O(N^2) diagnostic);
N = 1M: 1500 steps (the interaction lists grow ~60x on the way); PEFRL at N = 65 536."""
import sys, time, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from coulomb_oscillators_amd import Engine, EVAL_FMM_KDTREE, INTEG_LEAPFROG, INTEG_PEFRL
from bench import gaussian_ball, coulomb_params
for n, steps, ts, scheme in [(65536, 3000, 1, INTEG_LEAPFROG), (65536, 3000, 8, INTEG_LEAPFROG), (1 << 20, 1500, 1, INTEG_LEAPFROG), (65536, 600, 1, INTEG_PEFRL)]:
    buf = gaussian_ball(n, 7); par = coulomb_params(n)
    d = torch.from_numpy(buf).cuda(); prm = torch.from_numpy(par).cuda()
    e = Engine(fmm_order=6, unsort=0, tree_steps=ts, sync=0)
    e.compute_force(EVAL_FMM_KDTREE, d, n, prm)
    E0 = e.energy(d, n, prm) if n <= 65536 else None
    t0 = time.time()
    for k in range(steps):
        e.integrate(scheme, EVAL_FMM_KDTREE, d, n, prm, 5e-4)
    torch.cuda.synchronize(); dt = time.time() - t0
    ok = bool(torch.isfinite(d).all())
    E1 = e.energy(d, n, prm) if n <= 65536 else None
    info = e.kd_info()
    msg = f"n={n} steps={steps} tree_steps={ts} scheme={scheme}: finite={ok} {1e3*dt/steps:.3f} ms/step build_mode={info.build_mode}"
    if E0 is not None:
        t0_, t1_ = sum(E0), sum(E1)
        msg += f" E0={t0_:.9e} E1={t1_:.9e} rel drift={(t1_-t0_)/abs(t0_):.2e}"
    print(msg, flush=True)
    e.close()
