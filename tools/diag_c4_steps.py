"""diagnostics: a few leapfrog steps of G kd-domains in lockstep (domains re-cut before every evaluation, capped LET exchange)
against the single-GPU run of the same system
    python tools/diag_c4_steps.py <log2 n> <G> <steps>"""
import sys

import numpy as np
import torch

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
from oracle.pyoracle import Oracle
from test_gpu_dist import loopback, make_state
from coulomb_oscillators_amd import EVAL_FMM_KDTREE, INTEG_LEAPFROG, Engine


def main():
    n, G, steps = 1 << int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    dt = 5e-4
    o = Oracle(np.float32)
    pos, vel = make_state(o, n, "reference")
    par = torch.from_numpy(o.params(n)).cuda()
    opts = dict(fmm_order=6, unsort=0, tree_steps=1)
    e1 = Engine(**opts)
    d = torch.cat([torch.from_numpy(pos).reshape(-1), torch.from_numpy(vel).reshape(-1), torch.zeros(3 * n)]).cuda()
    e1.compute_force(EVAL_FMM_KDTREE, d, n, par)
    for _ in range(steps):
        e1.integrate(INTEG_LEAPFROG, EVAL_FMM_KDTREE, d, n, par, dt)
    torch.cuda.synchronize()
    i1 = e1.kd_info()
    print("single: warm builds %d misses %d" % (i1.warm_builds, i1.warm_misses))
    ref = d.clone().view(3, n, 3)
    e1.close()
    world = loopback(n, G, pos, vel, **opts)
    nl = n // G
    world.force(par, elastic=True, let=True)
    for k in range(steps):
        for r in world.runs:
            r.eng.step(r.vel, r.acc, 0.5 * dt, nl); r.eng.step(r.pos, r.vel, dt, nl)
        world.partition([r.pos for r in world.runs], [r.vel for r in world.runs])
        world.force(par, elastic=True, let=True, capped=True)
        for r in world.runs:
            r.eng.step(r.vel, r.acc, 0.5 * dt, nl)
    torch.cuda.synchronize()
    got = torch.cat([r.buf.view(3, -1, 3) for r in world.runs], dim=1).reshape(3, n, 3)
    for name, k in (("positions", 0), ("velocities", 1), ("accelerations", 2)):
        print("%s differ in %d of %d rows" % (name, int((got[k] != ref[k]).any(dim=1).sum()), n))
    print("capped evaluations %d, repeated %d; warm builds / misses per domain %s" % (world.let_capped_evals, world.let_redos,
          [(int(r.eng.kd_info().warm_builds), int(r.eng.kd_info().warm_misses)) for r in world.runs]))


if __name__ == "__main__":
    main()
