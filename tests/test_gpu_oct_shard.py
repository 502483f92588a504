"""GPU tests of the multi-GPU form of the uniform-octree evaluators (slabs of the sorted cell keys) on ONE card:
G contexts in lockstep (LoopbackSlabs), each evaluating its slab through nbco_fmm_oct_shard.  Bar: positions / velocities in
cell order and every acceleration equal the single-GPU nbco_fmm_traceless / nbco_fmm_symmetric BIT FOR BIT."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _state(oracle32, n, kind):
    if kind == "reference":
        buf = oracle32.init_reference(n)
        return np.ascontiguousarray(buf[0]), np.ascontiguousarray(buf[1])
    rng = np.random.default_rng(77 + n)
    if kind == "uniform":
        return rng.random((n, 3), dtype=np.float32), rng.standard_normal((n, 3)).astype(np.float32)
    a = rng.standard_normal((n // 3, 3)).astype(np.float32) * 0.05
    b = rng.standard_normal((n - n // 3, 3)).astype(np.float32) + np.float32(1.5)
    return np.concatenate([a, b]), rng.standard_normal((n, 3)).astype(np.float32)


@pytest.mark.parametrize("n,G,p,kind,sym,f64", [(30000, 2, 4, "uniform", False, False), (50000, 4, 6, "reference", False, False), (65536, 8, 3, "clumps", False, False),
                                                (40000, 3, 5, "uniform", True, False), (200000, 8, 6, "reference", False, False), (30000, 4, 10, "uniform", False, True),
                                                (20000, 5, 2, "clumps", True, False)])
def test_slabs_equal_single_gpu(oracle32, n, G, p, kind, sym, f64):
    import torch
    from coulomb_oscillators_amd import Engine, LoopbackSlabs
    pos, vel = _state(oracle32, n, kind)
    par = torch.from_numpy(oracle32.params(n)).cuda()
    opts = dict(fmm_order=p, far_fp64=int(f64))
    e1 = Engine(**opts)
    ref = torch.cat([torch.from_numpy(pos).reshape(-1), torch.from_numpy(vel).reshape(-1), torch.zeros(3 * n)]).cuda()
    (e1.fmm_cart3 if sym else e1.fmm_cart3_traceless)(ref, ref[6 * n:], n, par)
    world = LoopbackSlabs([Engine(**opts) for _ in range(G)], n, symmetric=sym)
    world.set_state(torch.from_numpy(pos).cuda(), torch.from_numpy(vel).cuda())
    for r in world.runs:
        r.acc.fill_(float("nan"))        # whatever a rank does not receive would stay NaN
    world.force(par, elastic=False)
    torch.cuda.synchronize()
    b = world.runs[0].bounds
    assert b[0] == 0 and b[-1] == n and all(x <= y for x, y in zip(b[:-1], b[1:]))
    sizes = np.diff(b)
    if kind == "uniform":
        assert sizes.max() < 2.0 * n / G       # (clustered inputs put many particles into few x-layers: coarser balance)
    for r in world.runs:
        assert torch.equal(r.buf, ref), "rank %d differs from the single-GPU evaluation" % r.rank
        assert r.exchange_bytes() == (G - 1) * 12 * int(sizes.max())


def test_slab_leapfrog_matches_single_gpu(oracle32):
    import torch
    from coulomb_oscillators_amd import Engine, LoopbackSlabs, EVAL_FMM_TRACELESS, INTEG_LEAPFROG
    n, G, p, steps, dt = 40000, 4, 4, 4, 1e-3
    pos, vel = _state(oracle32, n, "reference")
    par = torch.from_numpy(oracle32.params(n)).cuda()
    e1 = Engine(fmm_order=p)
    ref = torch.cat([torch.from_numpy(pos).reshape(-1), torch.from_numpy(vel).reshape(-1), torch.zeros(3 * n)]).cuda()

    def f1():
        e1.fmm_cart3_traceless(ref, ref[6 * n:], n, par)
        e1.add_elastic(ref[:3 * n], ref[6 * n:], n, par[3:])
    f1()
    for _ in range(steps):
        e1.step(ref[3 * n:6 * n], ref[6 * n:], 0.5 * dt, n); e1.step(ref[:3 * n], ref[3 * n:6 * n], dt, n); f1(); e1.step(ref[3 * n:6 * n], ref[6 * n:], 0.5 * dt, n)
    world = LoopbackSlabs([Engine(fmm_order=p) for _ in range(G)], n)
    world.set_state(torch.from_numpy(pos).cuda(), torch.from_numpy(vel).cuda())
    world.force(par)
    for _ in range(steps):
        for r in world.runs:
            r.eng.step(r.vel, r.acc, 0.5 * dt, n); r.eng.step(r.pos, r.vel, dt, n)
        world.force(par)
        for r in world.runs:
            r.eng.step(r.vel, r.acc, 0.5 * dt, n)
    torch.cuda.synchronize()
    for r in world.runs:
        assert torch.equal(r.buf, ref)


def test_argument_errors(engine):
    import torch
    from coulomb_oscillators_amd import EngineError
    n = 4096
    buf = torch.zeros(9 * n, device="cuda")
    for world, rank in ((0, 0), (4, 4), (65, 0), (2, -1)):
        with pytest.raises(EngineError):
            engine.fmm_oct_shard(buf, buf[6 * n:], n, None, world, rank)
