"""diagnostics: which ingredient makes two lockstep worlds differ (see tools/soak_dist.py)
    python tools/diag_soak_dist.py [steps]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coulomb_oscillators_amd import Engine, LoopbackWorld
from bench import gaussian_ball, coulomb_params


def world(G, n, pos, vel, warm, gather_partition, **opts):
    os.environ["NBCO_SEL_WARM"] = "1" if warm else "0"
    try:
        engines = [Engine(**opts) for _ in range(G)]
    finally:
        del os.environ["NBCO_SEL_WARM"]
    w = LoopbackWorld(engines, n, gather_partition=gather_partition)
    nl = n // G
    w.partition([torch.from_numpy(pos[r * nl:(r + 1) * nl]).cuda() for r in range(G)], [torch.from_numpy(vel[r * nl:(r + 1) * nl]).cuda() for r in range(G)])
    return w


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    G, n, rebalance, dt, p = 4, 1 << 18, 8, 5e-4, 4
    buf = gaussian_ball(n, 5); par = torch.from_numpy(coulomb_params(n)).cuda()
    opts = dict(fmm_order=p, unsort=0, tree_steps=1, p2p_mutual=0, list_factor=8, list_grow=1)
    nl = n // G
    cfgs = {"base (gather exch, gathered part, cold)": (False, True, False),
            "base again": (False, True, False),
            "base a third time": (False, True, False),
            "dpart, gather exch, cold": (False, None, False),
            "all (soak's world a)": (True, None, True)}
    worlds = {k: (world(G, n, buf[0], buf[1], v[0], v[1], **opts), v[2]) for k, v in cfgs.items()}
    for w, let in worlds.values():
        w.force(par, elastic=True, let=let)
    ref = None
    for k in range(steps):
        states = {}
        for name, (w, let) in worlds.items():
            for r in w.runs:
                r.eng.step(r.vel, r.acc, 0.5 * dt, nl); r.eng.step(r.pos, r.vel, dt, nl)
            if (k + 1) % rebalance == 0:
                w.partition([r.pos for r in w.runs], [r.vel for r in w.runs])
            w.force(par, elastic=True, let=let)
            for r in w.runs:
                r.eng.step(r.vel, r.acc, 0.5 * dt, nl)
            states[name] = torch.cat([r.buf for r in w.runs])
        base = states["base (gather exch, gathered part, cold)"]
        line = []
        for name, s in states.items():
            if name.startswith("base"):
                continue
            d = int((s.view(3, -1, 3) != base.view(3, -1, 3)).any(dim=2).sum())
            line.append("%s: %d rows" % (name, d))
        print("step %2d  " % (k + 1) + " | ".join(line), flush=True)


if __name__ == "__main__":
    main()
