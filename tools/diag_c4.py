"""diagnostics: where a lockstep run of G kd-domains differs from the single-GPU evaluation of the same system
    python tools/diag_c4.py <log2 n> <G> [kind] [engine option=value ...]"""
import sys

import numpy as np
import torch

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
from oracle.pyoracle import Oracle
from test_gpu_dist import loopback, make_state, single_gpu


def main():
    n, G = 1 << int(sys.argv[1]), int(sys.argv[2])
    kind = sys.argv[3] if len(sys.argv) > 3 else "reference"
    extra = dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in sys.argv[4:])
    o = Oracle(np.float32)
    pos, vel = make_state(o, n, kind)
    par = torch.from_numpy(o.params(n)).cuda()
    opts = dict(fmm_order=6, unsort=0, tree_steps=1)
    opts.update(extra)
    e1, ref = single_gpu(n, pos, vel, par, **opts)
    info = e1.kd_info()
    print("single: L", info.L, "build_mode", info.build_mode, "pairs", info.p2p_pairs, info.m2l_pairs)
    ref = ref.clone()
    e1.close()
    world = loopback(n, G, pos, vel, **opts)
    world.force(par, elastic=False, let=True)
    torch.cuda.synchronize()
    got = torch.cat([r.buf.view(3, -1, 3) for r in world.runs], dim=1).reshape(3, n, 3)
    ref = ref.view(3, n, 3)
    dp = (got[0] != ref[0]).any(dim=1)
    print("positions differ in %d of %d rows" % (int(dp.sum()), n))
    if int(dp.sum()):
        idx = torch.nonzero(dp).flatten()
        print("first / last differing row", int(idx[0]), int(idx[-1]))
        leaf = (idx * (1 << info.L)) // n
        print("leaves touched:", int(torch.unique(leaf).numel()), "domains touched:", torch.unique(idx // (n // G)).tolist())
        # same particle sets per domain?
        nl = n // G
        for g in range(G):
            a, b = got[0, g * nl:(g + 1) * nl].view(torch.int32).to(torch.int64), ref[0, g * nl:(g + 1) * nl].view(torch.int32).to(torch.int64)
            same = int(a.sum()) == int(b.sum()) and int((a * a).sum()) == int((b * b).sum())
            print("  domain %d: same particle set %s, differing rows %d" % (g, same, int(dp[g * nl:(g + 1) * nl].sum())))
        # per leaf: same set?
        lf = torch.unique(leaf)[:5].tolist()
        for l in lf:
            s, e = (l * n) >> info.L, ((l + 1) * n) >> info.L
            A = got[0, s:e].cpu().numpy(); B = ref[0, s:e].cpu().numpy()
            sa = A[np.lexsort(A.T[::-1])]; sb = B[np.lexsort(B.T[::-1])]
            print("  leaf %d rows [%d, %d): same set %s" % (l, s, e, np.array_equal(sa, sb)))
    da = (got[2] != ref[2]).any(dim=1)
    print("accelerations differ in %d rows; max rel diff %.3g" % (int(da.sum()), float(((got[2] - ref[2]).norm(dim=1) / ref[2].norm(dim=1).clamp_min(1e-30)).max())))
    print("warm/cold info of domains:", [(int(r.eng.kd_info().build_mode), int(r.eng.kd_info().p2p_pairs)) for r in world.runs])


if __name__ == "__main__":
    main()
