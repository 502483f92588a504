"""diagnostics: distributed re-partition against the gathered one (cold builds, all-gather exchange), stage by stage
    python tools/diag_dpart.py [log2 n] [G] [seed]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coulomb_oscillators_amd import Engine, LoopbackWorld
from bench import gaussian_ball, coulomb_params


def rows(t, nl):
    a = np.concatenate([t[:3 * nl].cpu().numpy().reshape(nl, 3), t[3 * nl:6 * nl].cpu().numpy().reshape(nl, 3)], axis=1)
    return a[np.lexsort(a.T[::-1])]


def main():
    n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 18)
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    buf = gaussian_ball(n, seed); par = torch.from_numpy(coulomb_params(n)).cuda()
    nl = n // G
    os.environ["NBCO_SEL_WARM"] = "0"
    ws = []
    for gp in (True, None, None):
        engines = [Engine(fmm_order=4, unsort=0, tree_steps=1, list_factor=8, list_grow=1) for _ in range(G)]
        w = LoopbackWorld(engines, n, gather_partition=gp)
        w.partition([torch.from_numpy(buf[0][r * nl:(r + 1) * nl]).cuda() for r in range(G)], [torch.from_numpy(buf[1][r * nl:(r + 1) * nl]).cuda() for r in range(G)])
        ws.append(w)
    torch.cuda.synchronize()
    names = ["gathered", "dpart 1", "dpart 2"]
    for k in (1, 2):
        same_sets = [np.array_equal(rows(ws[0].runs[g].buf, nl), rows(ws[k].runs[g].buf, nl)) for g in range(G)]
        print("after the partition: %s vs gathered: same particle set per domain %s" % (names[k], same_sets))
    for w in ws:
        w.force(par, elastic=False, let=False)
    torch.cuda.synchronize()
    for k in (1, 2):
        a = torch.cat([r.buf.view(3, -1, 3) for r in ws[0].runs], dim=1)
        b = torch.cat([r.buf.view(3, -1, 3) for r in ws[k].runs], dim=1)
        dp = (a[0] != b[0]).any(dim=1)
        da = (a[2] != b[2]).any(dim=1)
        print("after one evaluation: %s vs gathered: positions differ in %d rows, accelerations in %d" % (names[k], int(dp.sum()), int(da.sum())))
        if int(dp.sum()):
            idx = torch.nonzero(dp).flatten()
            print("   rows %d .. %d, domains %s" % (int(idx[0]), int(idx[-1]), torch.unique(idx // nl).tolist()))
    # first differing nodes of the trees the evaluations left in domain 3's engine
    g = G - 1
    arrs = {}
    for k, w in enumerate(ws[:2]):
        e = w.runs[g].eng
        arrs[k] = {nm: e.kd_array(nm) for nm in ("splitdim", "lbound", "rbound", "index", "mult", "center")}
        print(names[k], "domain", g, "engine: L", e.kd_info().L, "ntot", e.kd_info().ntot)
    for nm in arrs[0]:
        a, b = arrs[0][nm], arrs[1][nm]
        if a.shape != b.shape:
            print(nm, "shapes differ", a.shape, b.shape); continue
        d = np.nonzero((a != b).reshape(a.shape[0], -1).any(axis=1))[0]
        if len(d):
            lv = np.floor(np.log2(d + 1)).astype(int)
            print("%s: %d nodes differ, first %s (levels %s)" % (nm, len(d), d[:6].tolist(), lv[:6].tolist()))
            i = int(d[0])
            print("    node %d: gathered %s   dpart %s" % (i, a[i].tolist() if a.ndim > 1 else a[i], b[i].tolist() if b.ndim > 1 else b[i]))
    # the first two local levels of that domain on the CPU (exact medians by sorting)
    P = ws[0].runs[g].buf[:3 * nl].cpu().numpy().reshape(nl, 3)      # (tree order of world 0; only the SET matters)
    Pd = ws[1].runs[g].buf[:3 * nl].cpu().numpy().reshape(nl, 3)
    print("domain %d: same set in both worlds after the evaluation: %s" % (g, np.array_equal(P[np.lexsort(P.T[::-1])], Pd[np.lexsort(Pd.T[::-1])])))
    lb0, rb0 = arrs[0]["lbound"][0], arrs[0]["rbound"][0]
    a0 = int(np.argmax(rb0 - lb0))
    print("local root box", lb0.tolist(), rb0.tolist(), "axis", a0, "(tree says", int(arrs[0]["splitdim"][0]), int(arrs[1]["splitdim"][0]), ")")
    o0 = np.argsort(P[:, a0], kind="stable")
    left = P[o0[: nl // 2]]
    print("  level 0: left max %.9g right min %.9g | trees: gathered rb[1] %.9g lb[2] %.9g  dpart rb[1] %.9g lb[2] %.9g" % (
        left[:, a0].max(), P[o0[nl // 2:], a0].min(), arrs[0]["rbound"][1][a0], arrs[0]["lbound"][2][a0], arrs[1]["rbound"][1][a0], arrs[1]["lbound"][2][a0]))
    piv = left[:, a0].max()
    for k in (0, 1):
        Pk = ws[k].runs[g].buf[:3 * nl].cpu().numpy().reshape(nl, 3)
        tw = np.nonzero(Pk[:, a0] == piv)[0]
        print("  %s: particles with the pivot's key: %s" % (names[k], [(int(i), "left" if i < nl // 2 else "right", Pk[i].tolist()) for i in tw]))
    lb1, rb1 = arrs[0]["lbound"][1].copy(), arrs[0]["rbound"][1].copy()
    a1 = int(np.argmax(rb1 - lb1))
    o1 = np.argsort(left[:, a1], kind="stable")
    print("  level 1 (left child): box extents %s axis %d (trees say %d %d): left max %.9g right min %.9g | trees: gathered rb[3] %.9g lb[4] %.9g  dpart rb[3] %.9g lb[4] %.9g" % (
        (rb1 - lb1).tolist(), a1, int(arrs[0]["splitdim"][1]), int(arrs[1]["splitdim"][1]), left[o1[nl // 4 - 1], a1], left[o1[nl // 4], a1],
        arrs[0]["rbound"][3][a1], arrs[0]["lbound"][4][a1], arrs[1]["rbound"][3][a1], arrs[1]["lbound"][4][a1]))
    # top trees
    for k, w in enumerate(ws):
        i0 = w.runs[0].eng.kd_info()
        sd = w.runs[0].eng.kd_array("splitdim")[:7]
        lb = w.runs[0].eng.kd_array("lbound")[:7]
        rb = w.runs[0].eng.kd_array("rbound")[:7]
        print(names[k], "top split axes", sd.tolist(), "root box", lb[0].tolist(), rb[0].tolist())
        print("    level-2 boxes lo", lb[3:7].tolist())
        print("    level-2 boxes hi", rb[3:7].tolist())


if __name__ == "__main__":
    main()
