"""Distribution of the near-field list lengths per target leaf late in the benchmark's simulation (diagnostics):
    python tools/late_lists.py [steps, default 1000]
runs `steps` leapfrog steps of the N = 1M reference ball (tree_steps = 8), evaluates once with a rebuild and prints how the
P2P / M2L entries are spread over their targets."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import bench
from coulomb_oscillators_amd import EVAL_FMM_KDTREE, INTEG_LEAPFROG, Engine


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    n = 1 << 20
    eng = Engine(fmm_order=6, unsort=0, tree_steps=8, sync=0)
    d = torch.from_numpy(bench.gaussian_ball(n)).cuda()
    prm = torch.from_numpy(bench.coulomb_params(n)).cuda()
    eng.compute_force(EVAL_FMM_KDTREE, d, n, prm)
    eng.integrate_steps(INTEG_LEAPFROG, EVAL_FMM_KDTREE, d, n, prm, 5e-4, steps)
    eng.set(tree_steps=1)
    eng.compute_force(EVAL_FMM_KDTREE, d, n, prm)
    torch.cuda.synchronize()
    info = eng.kd_info()
    print("after %d steps: L %d, p2p pairs %d, m2l pairs %d, directed p2p interactions %d" % (steps, info.L, info.p2p_pairs, info.m2l_pairs, info.directed_p2p))
    for name in ("p2p", "m2l"):
        pr = eng.kd_array(name)
        cnt = np.bincount(np.concatenate([pr[:, 0], pr[:, 1]]), minlength=info.ntot)
        cnt = cnt[cnt > 0]
        srt = np.sort(cnt)[::-1]
        cum = np.cumsum(srt) / srt.sum()
        print("%s: %d targets with entries, mean %.1f, median %d, max %d; the 10 longest: %s" % (name, len(cnt), cnt.mean(), np.median(cnt), srt[0], srt[:10].tolist()))
        for frac in (0.25, 0.5, 0.75, 0.9):
            k = int(np.searchsorted(cum, frac)) + 1
            print("   %2d %% of the entries belong to the %d longest targets (length >= %d)" % (round(100 * frac), k, srt[k - 1]))
    mult = eng.kd_array("mult")
    lb, rb = eng.kd_array("lbound"), eng.kd_array("rbound")
    leaf0 = (1 << info.L) - 1
    diag = np.sqrt(((rb[leaf0:] - lb[leaf0:]) ** 2).sum(1))
    print("leaf box diagonals: median %.3g, 99 %% %.3g, max %.3g" % (np.median(diag), np.quantile(diag, 0.99), diag.max()))


if __name__ == "__main__":
    main()
