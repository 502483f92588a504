"""The drop-in binary against the library at the same settings over the same stretch of the simulation (diagnostics):
`nbco3 -n N -p 6 -iters K` prints its loop time from iteration 9 on; the library is driven here through the ABI from the same initial
state with the same options (tree_steps 8, m2l_first) and timed over the same iterations.   python tools/cli_vs_lib.py [n] [iters]"""
import os
import re
import subprocess
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from coulomb_oscillators_amd import Engine, EVAL_FMM_KDTREE, INTEG_LEAPFROG  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 49
    exe = os.path.join(ROOT, "coulomb_oscillators_amd", "host", "nbco3")
    for rep in range(3):
        with tempfile.TemporaryDirectory() as tmp:
            r = subprocess.run([exe, "-n", str(n), "-p", "6", "-iters", str(iters), "-steps", "100000", "-o", tmp], capture_output=True, text=True, timeout=900,
                               env=dict(os.environ, NBCO_HOST_TIMING="1"))
        m = re.search(r"Steady loop time: ([0-9.eE+-]+) s, (\d+) iterations", r.stdout)
        print("cli  rep %d: steady %.4f ms/step over %s iterations   %s" % (rep, 1e3 * float(m.group(1)) / int(m.group(2)), m.group(2), r.stderr.strip()[-200:].replace("\n", " | ")))
    for rep in range(3):
        eng = Engine(fmm_order=6, unsort=0, tree_steps=8, m2l_first=1, sync=0)
        d = torch.from_numpy(bench.gaussian_ball(n)).cuda()
        prm = torch.from_numpy(bench.coulomb_params(n)).cuda()
        eng.compute_force(EVAL_FMM_KDTREE, d, n, prm)
        eng.integrate_steps(INTEG_LEAPFROG, EVAL_FMM_KDTREE, d, n, prm, 5e-4, 1)
        eng.integrate_steps(INTEG_LEAPFROG, EVAL_FMM_KDTREE, d, n, prm, 5e-4, 8)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.integrate_steps(INTEG_LEAPFROG, EVAL_FMM_KDTREE, d, n, prm, 5e-4, iters + 1 - 9)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        info = eng.kd_info()
        print("lib  rep %d: %.4f ms/step over %d iterations, warm builds %d misses %d" % (rep, 1e3 * dt / (iters + 1 - 9), iters + 1 - 9, info.warm_builds, info.warm_misses))
        eng.close()


if __name__ == "__main__":
    main()
