"""The warm median select (k_kdselect.hip: one histogram pass per level around the previous build's pivots).  Bar: the tree -- and
with it every number downstream -- is the one the cold two-pass select builds, bit for bit; a window that misses the median costs
a repeated evaluation, never a different result."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _engine(warm, **opts):
    from coulomb_oscillators_amd import Engine
    old = os.environ.get("NBCO_SEL_WARM")
    os.environ["NBCO_SEL_WARM"] = "1" if warm else "0"
    try:
        return Engine(**opts)
    finally:
        if old is None:
            del os.environ["NBCO_SEL_WARM"]
        else:
            os.environ["NBCO_SEL_WARM"] = old


def _state(oracle32, n):
    import torch
    buf = oracle32.init_reference(n)
    return torch.from_numpy(buf.copy()).cuda(), torch.from_numpy(oracle32.params(n)).cuda()


@pytest.mark.parametrize("n,p,tree_steps,unsort", [(200000, 4, 1, 0), (1 << 20, 6, 1, 0), (150000, 3, 3, 0), (100000, 4, 1, 1), (20000, 5, 1, 0)])
def test_warm_builds_equal_cold_builds(oracle32, n, p, tree_steps, unsort):
    import torch
    from coulomb_oscillators_amd import EVAL_FMM_KDTREE, INTEG_LEAPFROG
    steps, dt = 9, 5e-4
    out, trees, infos = [], [], []
    for warm in (False, True):
        e = _engine(warm, fmm_order=p, unsort=unsort, tree_steps=tree_steps)
        d, prm = _state(oracle32, n)
        e.compute_force(EVAL_FMM_KDTREE, d, n, prm)
        for _ in range(steps):
            e.integrate(INTEG_LEAPFROG, EVAL_FMM_KDTREE, d, n, prm, dt)
        torch.cuda.synchronize()
        out.append(d.clone())
        trees.append({k: e.kd_array(k).copy() for k in ("lbound", "rbound", "splitdim", "index", "center")})
        infos.append(e.kd_info())
        e.close()
    assert torch.equal(out[0], out[1])
    for k in trees[0]:
        np.testing.assert_array_equal(trees[0][k], trees[1][k], err_msg=k)
    assert infos[0].warm_builds == 0
    if n > 8192:      # (smaller systems have no selection levels)
        assert infos[1].warm_builds >= steps // tree_steps and infos[1].warm_misses == 0
        assert infos[1].build_mode == 0


def test_a_missed_window_repeats_the_evaluation_cold(oracle32):
    """positions stretched by 30 % between two evaluations: the previous pivots are nowhere near the medians, every window misses,
    the evaluation is repeated with the two-pass select and gives the cold result; the next builds are warm again"""
    import torch
    from coulomb_oscillators_amd import EVAL_FMM_KDTREE
    n, p = 300000, 4
    res = []
    for warm in (False, True):
        e = _engine(warm, fmm_order=p, unsort=0)
        d, prm = _state(oracle32, n)
        e.compute_force(EVAL_FMM_KDTREE, d, n, prm)
        e.compute_force(EVAL_FMM_KDTREE, d, n, prm)
        d[0].mul_(1.3)
        e.compute_force(EVAL_FMM_KDTREE, d, n, prm)
        i1 = e.kd_info()
        e.compute_force(EVAL_FMM_KDTREE, d, n, prm)
        torch.cuda.synchronize()
        i2 = e.kd_info()
        res.append((d.clone(), i1, i2))
        e.close()
    assert torch.equal(res[0][0], res[1][0])
    _, i1, i2 = res[1]
    assert i1.warm_misses == 1 and i2.warm_misses == 1 and i2.warm_builds == i1.warm_builds + 1 and i2.build_mode == 0


def test_repeated_misses_send_it_into_a_cool_down(oracle32):
    """a miss costs a whole evaluation: two misses within 32 warm builds and the next 128 builds are cold (jumps of 50 % are
    beyond any window -- the situation of a cloud that keeps changing its shape); results stay the cold ones throughout"""
    import torch
    from coulomb_oscillators_amd import EVAL_FMM_KDTREE
    n = 100000
    e = _engine(True, fmm_order=3, unsort=0)
    c = _engine(False, fmm_order=3, unsort=0)
    d, prm = _state(oracle32, n)
    d2 = d.clone()
    e.compute_force(EVAL_FMM_KDTREE, d, n, prm)
    c.compute_force(EVAL_FMM_KDTREE, d2, n, prm)
    for k in range(8):
        for x in (d, d2):
            x[0].mul_(1.5 if k % 2 == 0 else 1 / 1.5)
        e.compute_force(EVAL_FMM_KDTREE, d, n, prm)
        c.compute_force(EVAL_FMM_KDTREE, d2, n, prm)
    torch.cuda.synchronize()
    info = e.kd_info()
    assert info.warm_misses == 2 and info.warm_builds == 2 and info.build_mode == 0
    assert torch.equal(d, d2)


@pytest.mark.parametrize("tree_steps", [3, 8])
def test_list_growth_keeps_the_rebuild_schedule(oracle32, tree_steps):
    """an engine that has to grow its lists in the middle of a run (list_factor = 1) and one that never has to stay identical: the
    repeated evaluation follows the rebuild schedule instead of forcing a rebuild"""
    import torch
    from coulomb_oscillators_amd import Engine, EVAL_FMM_KDTREE, INTEG_LEAPFROG
    n, p = 65536, 4
    out = []
    for lf in (48, 1):
        e = Engine(fmm_order=p, unsort=0, tree_steps=tree_steps, list_factor=lf, list_grow=1)
        d, prm = _state(oracle32, n)
        e.compute_force(EVAL_FMM_KDTREE, d, n, prm)
        e.integrate_steps(INTEG_LEAPFROG, EVAL_FMM_KDTREE, d, n, prm, 5e-4, 13)
        for _ in range(4):
            e.integrate(INTEG_LEAPFROG, EVAL_FMM_KDTREE, d, n, prm, 5e-4)
        torch.cuda.synchronize()
        out.append(d.clone())
        e.close()
    assert torch.equal(out[0], out[1])
