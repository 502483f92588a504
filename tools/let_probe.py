"""Exchange volume and per-stage cost of the LET exchange, measured on ONE card: G kd-domains in lockstep (LoopbackWorld).

    python tools/let_probe.py [--gpus 8] [--particles 1048576 (per domain)] [--order 6] [--evals 4]

Prints one JSON line: bytes received per rank and evaluation with the all-gather and with the LET exchange, and the wall time of
an evaluation round in both forms (all G domains share the card, so times are sums over domains, not a scaling figure)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=8)
    ap.add_argument("--particles", type=int, default=1 << 20)
    ap.add_argument("--order", type=int, default=6)
    ap.add_argument("--evals", type=int, default=4)
    ap.add_argument("--mutual", type=int, default=0)
    a = ap.parse_args()
    import numpy as np
    import torch
    from coulomb_oscillators_amd import Engine, LoopbackWorld
    from bench import gaussian_ball, coulomb_params
    G, nl = a.gpus, a.particles
    n = G * nl
    buf = gaussian_ball(n)
    pos, vel = buf[0], buf[1]
    par = torch.from_numpy(coulomb_params(n)).cuda()
    out = {"domains": G, "n_per_domain": nl, "n_system": n, "order": a.order}
    for let in (False, True):
        world = LoopbackWorld([Engine(fmm_order=a.order, unsort=0, tree_steps=1, p2p_mutual=a.mutual) for _ in range(G)], n)
        world.partition([torch.from_numpy(pos[r * nl:(r + 1) * nl]).cuda() for r in range(G)], [torch.from_numpy(vel[r * nl:(r + 1) * nl]).cuda() for r in range(G)])
        world.force(par, let=let)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.evals):
            world.force(par, let=let)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.evals
        key = "let" if let else "allgather"
        out[key + "_ms_per_round_all_domains"] = 1e3 * dt
        if let:
            for r in world.runs:
                r.eng.dist_let_check()
            b = [r.exchange_bytes() for r in world.runs]
            out["let_bytes_per_eval_per_gpu"] = {"mean": float(np.mean(b)), "max": int(max(b)), "min": int(min(b))}
            out["allgather_bytes_per_eval_per_gpu"] = world.runs[0].allgather_bytes()
            out["reduction"] = world.runs[0].allgather_bytes() / float(np.mean(b))
        del world
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
