#!/usr/bin/env python3
"""Is the host ahead of the GPU?  K leapfrog steps in one nbco_integrate_steps call: the time until the call returns (the host
has enqueued every launch; each evaluation ends with one look at the traversal's flags, i.e. the host cannot be more than one
evaluation ahead) against the time until the stream has drained, and the host's own share (time spent outside that wait).
    python tools/host_ahead.py [tree_steps] [n]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coulomb_oscillators_amd import Engine, EVAL_FMM_KDTREE, INTEG_LEAPFROG
from bench import gaussian_ball, coulomb_params

ts = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
buf = gaussian_ball(n, 5)
d = torch.from_numpy(buf.copy()).cuda(); prm = torch.from_numpy(coulomb_params(n)).cuda()
eng = Engine(fmm_order=6, unsort=0, tree_steps=ts, m2l_first=1 if ts > 1 else 0)
eng.compute_force(EVAL_FMM_KDTREE, d, n, prm)
eng.integrate_steps(INTEG_LEAPFROG, EVAL_FMM_KDTREE, d, n, prm, 5e-4, 16)
torch.cuda.synchronize()
for rep in range(3):
    K = 32
    w0 = eng.host_wait_s() if hasattr(eng, "host_wait_s") else 0.0
    t0 = time.perf_counter()
    eng.integrate_steps(INTEG_LEAPFROG, EVAL_FMM_KDTREE, d, n, prm, 5e-4, K)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    w1 = eng.host_wait_s() if hasattr(eng, "host_wait_s") else 0.0
    print(f"tree_steps={ts} n={n}: call returns after {1e3 * (t1 - t0) / K:.4f} ms/step, drained after {1e3 * (t2 - t0) / K:.4f} ms/step, "
          f"host waited for flags {1e3 * (w1 - w0) / K:.4f} ms/step -> host busy {1e3 * ((t1 - t0) - (w1 - w0)) / K:.4f} ms/step", flush=True)
