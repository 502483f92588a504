#!/usr/bin/env python3
"""Long runs of the warm median select against the cold one (GPU box): python tools/soak_warm.py
Two engines -- NBCO_SEL_WARM=1 and =0 -- integrate the same system with nbco_integrate_steps; their states must stay identical
bit for bit (same trees), whatever the cloud does on the way (it focuses around step 2000 at dt = 5e-4: lists grow 60x).
Prints the warm build / miss counters."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coulomb_oscillators_amd import Engine, EVAL_FMM_KDTREE, INTEG_LEAPFROG
from bench import gaussian_ball, coulomb_params


def engine(warm, **opts):
    os.environ["NBCO_SEL_WARM"] = "1" if warm else "0"
    try:
        return Engine(**opts)
    finally:
        del os.environ["NBCO_SEL_WARM"]


bad = 0
for n, steps, chunk, ts, dt, p in [(65536, 3000, 100, 1, 5e-4, 4), (300000, 2400, 100, 1, 5e-4, 4), (1 << 20, 2400, 200, 1, 5e-4, 6), (1 << 20, 1600, 200, 8, 5e-4, 6),
                                   (200000, 400, 20, 1, 2e-2, 3), (150000, 900, 50, 3, 5e-3, 5)]:
    buf = gaussian_ball(n, 3); par = coulomb_params(n)
    prm = torch.from_numpy(par).cuda()
    eng = [engine(w, fmm_order=p, unsort=0, tree_steps=ts, sync=0) for w in (False, True)]
    st = [torch.from_numpy(buf.copy()).cuda() for _ in eng]
    for e, d in zip(eng, st):
        e.compute_force(EVAL_FMM_KDTREE, d, n, prm)
    t0 = time.time()
    same = True
    for k in range(0, steps, chunk):
        for e, d in zip(eng, st):
            e.integrate_steps(INTEG_LEAPFROG, EVAL_FMM_KDTREE, d, n, prm, dt, chunk)
        torch.cuda.synchronize()
        if not torch.equal(st[0], st[1]):
            same = False
            print(f"  DIFFERENT after step {k + chunk}", flush=True)
            break
    info = eng[1].kd_info()
    ok = same and bool(torch.isfinite(st[1]).all())
    bad += not ok
    print(f"n={n} steps={steps} tree_steps={ts} dt={dt} p={p}: identical={same} finite={bool(torch.isfinite(st[1]).all())} warm_builds={info.warm_builds} "
          f"warm_misses={info.warm_misses} build_mode={info.build_mode} (cold engine build_mode={eng[0].kd_info().build_mode}) {time.time() - t0:.1f} s", flush=True)
    for e in eng:
        e.close()
print("FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
