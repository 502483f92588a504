"""nbco_integrate_steps: `steps` steps in one call.  Bar: the final state equals `steps` calls of nbco_integrate BIT FOR BIT -- the
fused pass between two force evaluations (kd_turnaround_kernel: tree order, elastic term, two half kicks, drift, next build's
prologue) performs the same operations with the same roundings as the four kernels it replaces."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _state(oracle32, n):
    import torch
    buf = oracle32.init_reference(n)
    return torch.from_numpy(buf.copy()).cuda(), torch.from_numpy(oracle32.params(n)).cuda()


@pytest.mark.parametrize("n,p,tree_steps,elastic,steps", [(20000, 4, 1, True, 5), (20000, 4, 1, False, 3), (65536, 6, 3, True, 8), (65536, 5, 8, True, 11),
                                                          (100000, 3, 2, True, 4), (5000, 6, 1, True, 4)])
def test_fused_leapfrog_equals_step_by_step(oracle32, n, p, tree_steps, elastic, steps):
    import torch
    from coulomb_oscillators_amd import Engine, EVAL_FMM_KDTREE, INTEG_LEAPFROG
    dt = 5e-4
    out = []
    for fused in (False, True):
        e = Engine(fmm_order=p, unsort=0, tree_steps=tree_steps)
        d, prm = _state(oracle32, n)
        e.compute_force(EVAL_FMM_KDTREE, d, n, prm, elastic=elastic)
        if fused:
            e.integrate_steps(INTEG_LEAPFROG, EVAL_FMM_KDTREE, d, n, prm, dt, steps, elastic=elastic)
            # and the engine is in a state from which step-by-step calls carry on identically
            e.integrate(INTEG_LEAPFROG, EVAL_FMM_KDTREE, d, n, prm, dt, elastic=elastic)
            e.integrate_steps(INTEG_LEAPFROG, EVAL_FMM_KDTREE, d, n, prm, dt, 2, elastic=elastic)
        else:
            for _ in range(steps + 3):
                e.integrate(INTEG_LEAPFROG, EVAL_FMM_KDTREE, d, n, prm, dt, elastic=elastic)
        torch.cuda.synchronize()
        assert torch.isfinite(d).all()
        out.append(d.clone())
        e.close()
    assert torch.equal(out[0][0], out[1][0]), "positions differ"
    assert torch.equal(out[0][1], out[1][1]), "velocities differ"
    assert torch.equal(out[0][2], out[1][2]), "accelerations differ"


@pytest.mark.parametrize("scheme_name,kind_name,opts", [("INTEG_PEFRL", "EVAL_FMM_KDTREE", dict(unsort=0)), ("INTEG_LEAPFROG", "EVAL_FMM_KDTREE", dict(unsort=1)),
                                                        ("INTEG_LEAPFROG", "EVAL_DIRECT", dict()), ("INTEG_LEAPFROG", "EVAL_FMM_KDTREE", dict(unsort=0, track_order=1)),
                                                        ("INTEG_FORESTRUTH", "EVAL_FMM_TRACELESS", dict())])
def test_everything_else_loops(oracle32, scheme_name, kind_name, opts):
    import torch
    import coulomb_oscillators_amd as co
    scheme, kind = getattr(co, scheme_name), getattr(co, kind_name)
    n, steps, dt = 8192, 3, 5e-4
    out = []
    for fused in (False, True):
        e = co.Engine(fmm_order=4, **opts)
        d, prm = _state(oracle32, n)
        e.compute_force(kind, d, n, prm)
        if fused:
            e.integrate_steps(scheme, kind, d, n, prm, dt, steps)
        else:
            for _ in range(steps):
                e.integrate(scheme, kind, d, n, prm, dt)
        torch.cuda.synchronize()
        out.append(d.clone())
        e.close()
    assert torch.equal(out[0], out[1])


def test_one_step_and_zero_steps(oracle32):
    import torch
    from coulomb_oscillators_amd import Engine, EVAL_FMM_KDTREE, INTEG_LEAPFROG
    n = 8192
    e = Engine(fmm_order=4, unsort=0)
    d, prm = _state(oracle32, n)
    e.compute_force(EVAL_FMM_KDTREE, d, n, prm)
    before = d.clone()
    e.integrate_steps(INTEG_LEAPFROG, EVAL_FMM_KDTREE, d, n, prm, 5e-4, 0)
    torch.cuda.synchronize()
    assert torch.equal(d, before)
    e.integrate_steps(INTEG_LEAPFROG, EVAL_FMM_KDTREE, d, n, prm, 5e-4, 1)
    e2 = Engine(fmm_order=4, unsort=0)
    d2 = before.clone()
    e2.compute_force(EVAL_FMM_KDTREE, d2, n, prm)   # (same tree state as e had)
    d2.copy_(before)
    e2.integrate(INTEG_LEAPFROG, EVAL_FMM_KDTREE, d2, n, prm, 5e-4)
    torch.cuda.synchronize()
    assert torch.equal(d, d2)
