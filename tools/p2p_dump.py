"""Writes the inputs of the near-field launch of one real evaluation (N particles of the reference's Gaussian ball, p = 6, a few
leapfrog steps in) for tools/p2p_lab.hip:   NBCO_P2P_DUMP=/tmp/p2p.bin python3 tools/p2p_dump.py [n] [steps]
(the library's diagnostics hook, csrc/k_fmm_kd.hip: p2p_dump, rewrites the file at every evaluation: the last one stays)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coulomb_oscillators_amd import Engine, EVAL_FMM_KDTREE, INTEG_LEAPFROG  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    assert os.environ.get("NBCO_P2P_DUMP"), "set NBCO_P2P_DUMP=<file>"
    import bench
    eng = Engine(fmm_order=6, unsort=0, sync=0)
    buf = torch.from_numpy(bench.gaussian_ball(n)).cuda()
    par = torch.from_numpy(bench.coulomb_params(n)).cuda()
    eng.compute_force(EVAL_FMM_KDTREE, buf, n, par)
    for _ in range(steps):
        eng.integrate(INTEG_LEAPFROG, EVAL_FMM_KDTREE, buf, n, par, 5e-4)
    torch.cuda.synchronize()
    info = eng.kd_info()
    print("dumped to", os.environ["NBCO_P2P_DUMP"], "n", n, "directed pairs", info.directed_p2p, "p2p pairs", info.p2p_pairs)


if __name__ == "__main__":
    main()
