#!/usr/bin/env python3
"""Long sharded runs in lockstep on one card (GPU box): python tools/soak_dist.py
Two worlds of G kd-domains integrate the same system: (a) LET exchange + distributed re-partition + warm select, (b) all-gather
exchange + gathered partition + cold select.  Both cut their domains every `rebalance` evaluations and must stay identical bit
for bit through list growth, warm-select misses, restarted evaluations and particles changing owner."""
import os, sys, time, numpy as np, torch
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


bad = 0
CASES = [(4, 1 << 18, 600, 8, 5e-4, 4, 0), (8, 1 << 19, 300, 16, 5e-4, 6, 0), (2, 1 << 17, 400, 5, 5e-3, 3, 0), (4, 1 << 18, 200, 8, 5e-4, 5, 1)]
if "--long" in sys.argv:   # (frequent cuts, many domains, a long run: what the order of a domain after the cut has to survive)
    CASES = [(8, 1 << 20, 400, 8, 5e-4, 6, 0), (4, 1 << 18, 1500, 4, 2e-3, 4, 0), (16, 1 << 19, 200, 4, 1e-3, 3, 0), (4, 1 << 18, 600, 1, 5e-4, 4, 0),
             (2, 1 << 20, 300, 8, 5e-4, 6, 0)]
for G, n, steps, rebalance, dt, p, mutual in CASES:
    buf = gaussian_ball(n, 5); par = torch.from_numpy(coulomb_params(n)).cuda()
    opts = dict(fmm_order=p, unsort=0, tree_steps=1, p2p_mutual=mutual, list_factor=8, list_grow=1)
    wa = world(G, n, buf[0], buf[1], True, None, **opts)
    wb = world(G, n, buf[0], buf[1], False, True, **opts)
    nl = n // G
    t0 = time.time()
    same = True
    for w, let in ((wa, True), (wb, False)):
        w.force(par, elastic=True, let=let)
    for k in range(steps):
        for w, let in ((wa, True), (wb, False)):
            for r in w.runs:
                r.eng.step(r.vel, r.acc, 0.5 * dt, nl); r.eng.step(r.pos, r.vel, dt, nl)
            if (k + 1) % rebalance == 0:
                w.partition([r.pos for r in w.runs], [r.vel for r in w.runs])
            w.force(par, elastic=True, let=let)
            for r in w.runs:
                r.eng.step(r.vel, r.acc, 0.5 * dt, nl)
        if (k + 1) % 50 == 0:
            for r in wa.runs:
                r.eng.dist_let_check()
            sa = torch.cat([r.buf for r in wa.runs]); sb = torch.cat([r.buf for r in wb.runs])
            if not torch.equal(sa, sb):
                same = False
                print(f"  DIFFERENT after step {k + 1}", flush=True)
                break
    fin = bool(torch.isfinite(torch.cat([r.buf for r in wa.runs])).all())
    info = [r.eng.kd_info() for r in wa.runs]
    bad += not (same and fin)
    print(f"G={G} n={n} steps={steps} rebalance={rebalance} dt={dt} p={p} mutual={mutual}: identical={same} finite={fin} "
          f"warm builds/misses={sum(i.warm_builds for i in info)}/{sum(i.warm_misses for i in info)} "
          f"LET bytes/eval/rank={np.mean([r.exchange_bytes() for r in wa.runs]):.3g} vs all-gather {wb.runs[0].allgather_bytes():.3g}; "
          f"partition bytes/rank={np.mean([r.partition_bytes for r in wa.runs]):.3g} vs {wb.runs[0].partition_bytes:.3g}  {time.time() - t0:.0f} s", flush=True)
    for w in (wa, wb):
        for r in w.runs:
            r.eng.close()
print("FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
