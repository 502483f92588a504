"""GPU parity tests of the kd-tree FMM evaluator (through the C ABI) against the CPU oracle.

Bar (north_star): tree integers and interaction lists bit-exact, per-particle forces within 1e-5
relative (regularised, see nbutil.force_err) of the oracle on the same inputs."""
import json
import os

import numpy as np
import pytest

from nbutil import canon_pairs, directed_pairs, force_err

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def dev(x):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def run_gpu(engine, buf, par, n, **opts):
    import torch
    engine.set(**opts)
    d = dev(buf[:2])
    a = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
    engine.fmm_cart3_kdtree(d, a, n, dev(par))
    return d.cpu().numpy(), a.cpu().numpy()


@pytest.mark.parametrize("n,p", [(4096, 6), (4096, 3), (5000, 4), (30001, 5), (1000, 2), (300, 1)])
def test_tree_and_lists_bit_exact(engine, oracle32, n, p):
    """index / mult / splitdim / bounds / centres and the P2P / M2L lists (as sets) equal the oracle's."""
    o = oracle32
    buf = o.init_reference(n)
    par = o.params(n)
    o.fmm_kd(buf[:2], par, p=p, threads=4, unsort=True)
    offM, offL = p * (p + 1) * (p + 2) // 6, (p + 1) ** 2
    want = o.kd_tree(offM=offM, offL=offL)
    run_gpu(engine, buf, par, n, fmm_order=p, unsort=1)
    info = engine.kd_info()
    assert (info.L, info.ntot) == (want["L"], want["ntot"])
    for name in ("index", "mult", "splitdim"):
        np.testing.assert_array_equal(engine.kd_array(name), want[name], err_msg=name)
    for name in ("lbound", "rbound", "center"):
        np.testing.assert_array_equal(engine.kd_array(name), want[name], err_msg=name)
    np.testing.assert_array_equal(engine.kd_array("unsort"), o.kd_unsort(n))
    for name in ("p2p", "m2l"):
        np.testing.assert_array_equal(canon_pairs(engine.kd_array(name)), canon_pairs(want[name]), err_msg=name)
    assert info.directed_p2p == directed_pairs(want["mult"], want["p2p"], want["L"])
    # expansions: multipoles to rounding, locals relative to the largest component of their order
    mp = engine.kd_array("mpole")
    mscale = np.abs(want["mpole"]).max(axis=0, keepdims=True).clip(1e-30)
    assert (np.abs(mp - want["mpole"]) / mscale).max() < 2e-5
    lo = engine.kd_array("local")
    scale = np.abs(want["local"]).max(axis=0, keepdims=True).clip(1e-30)
    assert (np.abs(lo - want["local"]) / scale).max() < 2e-5


@pytest.mark.parametrize("n,p", [(4096, 1), (4096, 2), (4096, 4), (4096, 6), (4096, 8), (4096, 10), (5000, 6), (30001, 3),
                                 (1000, 5), (65, 2)])
def test_accelerations_match_oracle(engine, oracle32, n, p):
    o = oracle32
    buf = o.init_reference(n)
    par = o.params(n)
    _, a_ref = o.fmm_kd(buf[:2], par, p=p, threads=4, unsort=True)
    pv, a = run_gpu(engine, buf, par, n, fmm_order=p, unsort=1)
    np.testing.assert_array_equal(pv, buf[:2])          # b_unsort: positions / velocities untouched
    assert force_err(a, a_ref) < 1e-5
    # and the FMM itself converges to the direct sum at the rate the reference shows
    ref = o.direct3(buf[0], par, threads=4)
    assert abs(o.mean_relerr(a, ref) - o.mean_relerr(a_ref, ref)) <= 0.02 * o.mean_relerr(a_ref, ref) + 2e-6


@pytest.mark.parametrize("radius,n,p", [(2.0, 20000, 3), (3.0, 8000, 2)])
def test_long_interaction_lists(engine, oracle32, radius, n, p):
    """a wide opening radius gives per-target lists of hundreds of entries (614 / 1189 at most here, against 254 on the
    BASELINE run): every size class of the per-target list sort is exercised, lists stay bit-exact, forces within 1e-5"""
    o = oracle32
    buf = o.init_reference(n)
    par = o.params(n)
    _, a_ref = o.fmm_kd(buf[:2], par, p=p, threads=8, unsort=True, radius=radius)
    want = o.kd_tree()
    beg = (1 << want["L"]) - 1
    pairs = np.asarray(want["p2p"]).reshape(-1, 2)
    per_target = np.bincount(np.concatenate([pairs[:, 0], pairs[:, 1]]) - beg, minlength=1 << want["L"]) + 1
    assert per_target.max() > 512 and (per_target <= 64).any() and ((per_target > 256) & (per_target <= 512)).any()
    _, a = run_gpu(engine, buf, par, n, fmm_order=p, unsort=1, tree_radius=radius)
    for name in ("p2p", "m2l"):
        np.testing.assert_array_equal(canon_pairs(engine.kd_array(name)), canon_pairs(want[name]), err_msg=name)
    assert engine.kd_info().directed_p2p == directed_pairs(want["mult"], want["p2p"], want["L"])
    assert force_err(a, a_ref) < 1e-5
    engine.set(tree_radius=1.0)


@pytest.mark.parametrize("mutual", [0, 1])
def test_long_ranges_take_their_own_kernel_when_the_lists_are_long(oracle32, mutual):
    """Late in a run a few stretched leaves are partners of most of the tree.  When the PREVIOUS evaluation's list averaged more
    than 48 entries per target, the per-target sort leaves ranges above 512 entries to list_longsort_kernel (a workgroup per long
    range, places from a bitmap of the range's sources).  Here: a wide opening radius and three far-away particles (their
    leaves pair with everything); the first evaluation of a context sorts the long ranges itself (the radix path), the second
    one hands them over -- same lists, same accelerations, bit for bit, with both near-field kernels."""
    import torch
    from coulomb_oscillators_amd import EVAL_FMM_KDTREE, Engine
    o = oracle32
    n, p = 65536, 3
    buf = o.init_reference(n)
    buf[0, :3] = np.array([[40.0, 0, 0], [0, -55.0, 0], [0, 0, 70.0]], dtype=np.float32) * np.abs(buf[0]).max()
    par = torch.from_numpy(o.params(n)).cuda()
    eng = Engine(fmm_order=p, unsort=0, tree_steps=1, tree_radius=2.0, p2p_mutual=mutual)
    d = torch.from_numpy(buf.copy()).cuda()
    out = []
    for k in range(3):
        eng.compute_force(EVAL_FMM_KDTREE, d, n, par)
        torch.cuda.synchronize()
        info = eng.kd_info()
        out.append((d.clone(), canon_pairs(eng.kd_array("p2p")), info.long_lists, info.directed_p2p))
        pairs = eng.kd_array("p2p")
    beg = (1 << info.L) - 1
    per_target = np.bincount(np.concatenate([pairs[:, 0], pairs[:, 1]]) - beg, minlength=1 << info.L) + 1
    assert per_target.max() > 2000 and per_target.mean() > 48 and 11 <= info.L + 1 <= 16, (per_target.max(), per_target.mean(), info.L)
    assert [x[2] for x in out] == [0, 1, 1]          # the first evaluation has no previous list to go by
    assert torch.isfinite(out[0][0]).all()
    for k in (1, 2):
        assert torch.equal(out[k][0], out[0][0])      # positions (tree order), velocities, accelerations
        np.testing.assert_array_equal(out[k][1], out[0][1])
        assert out[k][3] == out[0][3]
    eng.close()


def test_reference_test_mode_error_table_on_gpu(engine, oracle32):
    """The GPU evaluator reproduces the reference's recorded `-test` error table (main3.cu:790-811)."""
    with open(os.path.join(GOLD, "reference_recorded.json")) as f:
        rec = json.load(f)["test_mode_relerr"]
    o = oracle32
    n = rec["n"]
    buf = o.init_reference(n, test_mode=True)
    par = o.params(n)
    ref = o.direct3(buf[0], par, threads=4)
    for p, want in enumerate(rec["values"], start=1):
        _, a = run_gpu(engine, buf, par, n, fmm_order=p, unsort=1)
        got = o.mean_relerr(a, ref)
        assert abs(got - want) <= 1e-3 * want, (p, got, want)


def test_tree_order_output_and_velocity_permutation(engine, oracle32):
    """b_unsort = false (simulation mode): p, v come back in tree order, a in tree order (fmm_cart3_kdtree.cuh:1755-1760)."""
    o = oracle32
    n, p = 5000, 4
    buf = o.init_reference(n)
    par = o.params(n)
    pv_ref, a_ref = o.fmm_kd(buf[:2], par, p=p, threads=4, unsort=False)
    pv, a = run_gpu(engine, buf, par, n, fmm_order=p, unsort=0)
    np.testing.assert_array_equal(pv, pv_ref)
    assert force_err(a, a_ref) < 1e-5


def test_options_radius_coll_eps(engine, oracle32):
    o = oracle32
    n, p = 4096, 4
    buf = o.init_reference(n)
    par = o.params(n)
    for kw in (dict(radius=2.0), dict(coll=False), dict(eps2=1e-8), dict(dens_inhom=4.0)):
        okw = dict(p=p, threads=4, unsort=True)
        okw.update(kw)
        _, a_ref = o.fmm_kd(buf[:2], par, **okw)
        gkw = dict(fmm_order=p, unsort=1, tree_radius=kw.get("radius", 1.0), coll=int(kw.get("coll", True)),
                   eps2=kw.get("eps2", 1e-18), dens_inhom=kw.get("dens_inhom", 1.0))
        _, a = run_gpu(engine, buf, par, n, **gkw)
        assert force_err(a, a_ref) < 1e-5, kw
        t = o.kd_tree()
        np.testing.assert_array_equal(canon_pairs(engine.kd_array("m2l")), canon_pairs(t["m2l"]))
    engine.set(tree_radius=1.0, coll=1, eps2=1e-18, dens_inhom=1.0)


def test_m2l_first_variant_is_consistent(engine, oracle32):
    """GPU-reference traversal order (admissibility before the leaf test): different lists,
    same physics to FMM accuracy."""
    o = oracle32
    n, p = 4096, 6
    buf = o.init_reference(n)
    par = o.params(n)
    ref = o.direct3(buf[0], par, threads=4)
    _, a0 = run_gpu(engine, buf, par, n, fmm_order=p, unsort=1, m2l_first=0)
    n_p2p0 = engine.kd_info().p2p_pairs
    _, a1 = run_gpu(engine, buf, par, n, fmm_order=p, unsort=1, m2l_first=1)
    assert engine.kd_info().p2p_pairs <= n_p2p0
    assert o.mean_relerr(a1, ref) < 3 * o.mean_relerr(a0, ref) + 1e-5
    engine.set(m2l_first=0)


def test_leapfrog_with_fmm_matches_oracle(engine, oracle32):
    """nbco3's simulation loop (main3.cu:832-846): precompute + leapfrog steps, tree order kept."""
    from coulomb_oscillators_amd import EVAL_FMM_KDTREE, INTEG_LEAPFROG
    from oracle import pyoracle as po
    o = oracle32
    n, p = 4096, 5
    buf = o.init_reference(n)
    par = o.params(n)
    engine.set(fmm_order=p, unsort=0)
    d, prm = dev(buf), dev(par)
    o.compute_force(po.KIND_FMM_KD, buf, par, p=p, unsort=False, threads=4)
    engine.compute_force(EVAL_FMM_KDTREE, d, n, prm)
    for _ in range(3):
        o.integrate(po.SCHEME_LEAPFROG, po.KIND_FMM_KD, buf, par, 5e-4, p=p, unsort=False, threads=4)
        engine.integrate(INTEG_LEAPFROG, EVAL_FMM_KDTREE, d, n, prm, 5e-4)
    got = d.cpu().numpy()
    # particle order is the tree order of the last rebuild on both sides; match by sorting on (x,y,z)
    ka = np.lexsort((got[0][:, 2], got[0][:, 1], got[0][:, 0]))
    kb = np.lexsort((buf[0][:, 2], buf[0][:, 1], buf[0][:, 0]))
    assert np.abs(got[0][ka] - buf[0][kb]).max() <= 2e-6 * np.abs(buf[0]).max()
    assert np.abs(got[1][ka] - buf[1][kb]).max() <= 2e-5 * np.abs(buf[1]).max()
    assert force_err(got[2][ka], buf[2][kb]) < 5e-5


def test_tree_reuse_between_rebuilds(engine, oracle32):
    """tree_steps > 1 (the reference's GPU behaviour, fmm_cart3_kdtree.cuh:1619): topology reused,
    accuracy stays at FMM level after a few steps (-test2, main3.cu:812-831)."""
    from coulomb_oscillators_amd import EVAL_FMM_KDTREE, INTEG_LEAPFROG
    o = oracle32
    n, p = 4096, 6
    buf = o.init_reference(n)
    par = o.params(n)
    engine.set(fmm_order=p, unsort=0, tree_steps=8)
    d, prm = dev(buf), dev(par)
    engine.compute_force(EVAL_FMM_KDTREE, d, n, prm)
    assert engine.kd_info().rebuilt == 1
    for _ in range(4):
        engine.integrate(INTEG_LEAPFROG, EVAL_FMM_KDTREE, d, n, prm, 5e-4)
        assert engine.kd_info().rebuilt == 0
    got = d.cpu().numpy()
    ref = o.direct3(got[0], par, threads=4)
    k = np.array(par[3:6], dtype=np.float32)
    err = o.mean_relerr(got[2] + got[0] * k, ref)       # remove the elastic term again
    assert err < 5e-3
    engine.set(tree_steps=1)


def test_million_particles_properties(engine, oracle32):
    """BASELINE config 3 size (N = 1048576, p = 6): list sizes against the reference's recorded values
    (ties in fp32 coordinates make the last digits non-canonical), sampled force accuracy."""
    with open(os.path.join(GOLD, "reference_recorded.json")) as f:
        rec = json.load(f)["gaussian_p6_lists"]["1048576"]
    import torch
    o = oracle32
    n, p = 1048576, 6
    buf = o.init_reference(n)
    par = o.params(n)
    engine.set(fmm_order=p, unsort=1)
    d = dev(buf[:2])
    a = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
    engine.fmm_cart3_kdtree(d, a, n, dev(par))
    info = engine.kd_info()
    assert info.L == rec["L"]
    assert abs(info.p2p_pairs - rec["p2p"]) <= 1e-4 * rec["p2p"]
    assert abs(info.m2l_pairs - rec["m2l"]) <= 1e-4 * rec["m2l"]
    assert abs(info.directed_p2p - rec["pairs"]) <= 1e-4 * rec["pairs"]
    mult = engine.kd_array("mult")
    assert np.all(mult[(1 << info.L) - 1:] == 32)
    # sampled exact forces in fp64
    a_h = a.cpu().numpy().astype(np.float64)
    rows = np.random.default_rng(3).choice(n, 48, replace=False)
    pos64 = buf[0].astype(np.float64)
    want = np.empty((len(rows), 3))
    for k, i in enumerate(rows):
        dd = pos64[i] - pos64
        r2 = (dd ** 2).sum(1) + 1e-18
        want[k] = (dd / r2[:, None] ** 1.5).sum(0) * float(par[0])
    rel = np.linalg.norm(a_h[rows] - want, axis=1) / (np.linalg.norm(want, axis=1) + np.linalg.norm(want, axis=1).mean())
    assert np.median(rel) < 6e-3 and rel.max() < 5e-2
    # determinism: a second evaluation gives bit-identical accelerations (no float atomics)
    a2 = torch.zeros_like(a)
    engine.fmm_cart3_kdtree(d, a2, n, dev(par))
    assert torch.equal(a, a2)


@pytest.mark.parametrize("n,p,quant", [(262144, 6, 0), (100000, 4, 0), (65536, 5, 2e-5), (65536, 3, 4e-4)])
def test_large_tree_bit_exact_selection_build(engine, oracle32, n, p, quant):
    """Sizes where the top levels are built by median selection (k_kdselect.hip) and the rest in LDS: the whole
    tree, the permutation and the lists still equal the oracle's stable-sort build bit for bit.  `quant` snaps the
    coordinates to a grid: thousands of exactly tied keys exercise the tie resolver (2e-5: a few ties per pivot)
    and its overflow fallback to the sorting build (4e-4: hundreds of ties per pivot)."""
    o = oracle32
    buf = o.init_reference(n)
    if quant:
        buf[0] = (np.round(buf[0] / quant) * quant).astype(np.float32)
    par = o.params(n)
    _, a_ref = o.fmm_kd(buf[:2], par, p=p, threads=8, unsort=True)
    want = o.kd_tree()
    _, a = run_gpu(engine, buf, par, n, fmm_order=p, unsort=1)
    info = engine.kd_info()
    assert (info.L, info.ntot) == (want["L"], want["ntot"])
    for name in ("index", "mult", "splitdim", "lbound", "rbound", "center"):
        np.testing.assert_array_equal(engine.kd_array(name), want[name], err_msg=name)
    np.testing.assert_array_equal(engine.kd_array("unsort"), o.kd_unsort(n))
    for name in ("p2p", "m2l"):
        np.testing.assert_array_equal(canon_pairs(engine.kd_array(name)), canon_pairs(want[name]), err_msg=name)
    assert force_err(a, a_ref) < 1e-5
    # smooth inputs and a few ties per pivot stay on the fast path (two radix passes + exact resolution of the pivot's
    # bucket); hundreds of ties per pivot end in the stable-sort chain
    assert info.build_mode == (2 if quant == 4e-4 else 0)


@pytest.mark.parametrize("n,p,dens,levels", [(1 << 18, 2, 2.0, 17), (1 << 19, 2, 4.0, 19)])
def test_deep_trees(engine, oracle32, n, p, dens, levels):
    """trees of 17 and 19 levels (what N = 4M and N = 16M give at p = 6, here reached with the reference's -i option and
    one-particle leaves): every per-level launcher has to cope with more than 16 levels"""
    o = oracle32
    buf = o.init_reference(n)
    par = o.params(n)
    _, a_ref = o.fmm_kd(buf[:2], par, p=p, threads=8, unsort=True, dens_inhom=dens)
    want = o.kd_tree()
    assert want["L"] == levels
    _, a = run_gpu(engine, buf, par, n, fmm_order=p, unsort=1, dens_inhom=dens)
    info = engine.kd_info()
    assert (info.L, info.ntot) == (want["L"], want["ntot"])
    for name in ("index", "mult", "splitdim", "lbound", "rbound", "center"):
        np.testing.assert_array_equal(engine.kd_array(name), want[name], err_msg=name)
    for name in ("p2p", "m2l"):
        np.testing.assert_array_equal(canon_pairs(engine.kd_array(name)), canon_pairs(want[name]), err_msg=name)
    assert force_err(a, a_ref) < 1e-5
    engine.set(dens_inhom=1.0)


@pytest.mark.parametrize("n", [1, 2, 3, 7, 33, 100, 513, 4097, 8191, 8193, 12289])
def test_edge_sizes(oracle32, n):
    """sizes around every switch of the build (one-node trees, the 4096-particle LDS slice, the 8192-particle
    workgroup switch of the selection levels), fresh evaluation and tree reuse"""
    import torch
    from coulomb_oscillators_amd import Engine, EVAL_FMM_KDTREE
    o = oracle32
    for p in (2, 6):
        buf = o.init_reference(n)
        if n == 1:    # the reference's centred / rms-normalised init is 0/0 for a single particle
            buf[:] = 0
            buf[0, 0] = (0.1, -0.2, 0.3)
        par = o.params(n)
        _, a_ref = o.fmm_kd(buf[:2], par, p=p, threads=4, unsort=True)
        _, a = run_gpu(Engine(), buf, par, n, fmm_order=p, unsort=1)
        assert force_err(a, a_ref) < 1e-5
        e = Engine(fmm_order=p, unsort=0, tree_steps=3)
        d = dev(buf)
        for _ in range(4):
            e.compute_force(EVAL_FMM_KDTREE, d, n, dev(par))
        torch.cuda.synchronize()
        assert torch.isfinite(d).all()


def test_list_capacity_overflow_is_reported_and_recoverable(oracle32):
    """a traversal that runs out of list capacity must come back as NBCO_ERR_CAPACITY (everything queued behind it runs on
    a consistent empty state), leave the caller's positions / velocities untouched, and the context must keep working"""
    import torch
    from coulomb_oscillators_amd import Engine, EngineError
    o = oracle32
    n, p = 65536, 6
    buf = o.init_reference(n)
    par = o.params(n)
    e = Engine(fmm_order=p, unsort=0, list_factor=1, list_grow=0)
    d = dev(buf[:2])
    before = d.clone()
    a = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
    with pytest.raises(EngineError, match="list capacity"):
        e.fmm_cart3_kdtree(d, a, n, dev(par))
    torch.cuda.synchronize()
    assert torch.equal(d, before)
    _, want = o.fmm_kd(buf[:2], par, p=p, threads=4, unsort=False)
    # with list_grow (the default) the same context doubles its lists until the evaluation fits
    e.set(list_grow=1)
    e.fmm_cart3_kdtree(d, a, n, dev(par))
    torch.cuda.synchronize()
    assert force_err(a.cpu().numpy(), want) < 1e-5
    # ... and a sufficient list_factor needs no growth
    d2 = dev(buf[:2])
    e.set(list_factor=48)
    e.fmm_cart3_kdtree(d2, a, n, dev(par))
    torch.cuda.synchronize()
    assert force_err(a.cpu().numpy(), want) < 1e-5


@pytest.mark.parametrize("n,p,inhom,halves", [(4096, 6, 1.0, 1), (30001, 5, 1.0, 1), (65536, 6, 1.0, 1), (3000, 5, 1.3, 1), (46000, 6, 1.0, 2),
                                              (100000, 6, 1.0, 2), (65536, 8, 1.0, 2), (32768, 10, 1.0, 4), (24000, 9, 1.0, 4), (20000, 9, 1.0, 0),
                                              (5000, 6, 1.0, 0), (4096, 4, 1.0, 0)])
def test_mutual_near_field_matches_oracle_and_the_one_directional_kernel(engine, oracle32, n, p, inhom, halves):
    """opts.p2p_mutual: every leaf pair evaluated once, the force applied to both leaves (the reference GPU kernel's Newton-III
    form, fmm_cart3_kdtree.cuh:874-959).  Leaves are taken as 1, 2 or 4 halves of up to 32 particles (0: the leaf size fills the
    16-lane rows too badly, the one-directional kernel runs); forces stay within 1e-5 of the oracle, within 2e-6 of the
    one-directional kernel, and are bit-reproducible from run to run (no atomics)."""
    o = oracle32
    buf = o.init_reference(n)
    par = o.params(n)
    _, a_ref = o.fmm_kd(buf[:2], par, p=p, threads=4, unsort=True, dens_inhom=inhom)
    _, a_one = run_gpu(engine, buf, par, n, fmm_order=p, unsort=1, dens_inhom=inhom, p2p_mutual=0)
    info = engine.kd_info()
    assert info.p2p_halves == 0
    _, a_mut = run_gpu(engine, buf, par, n, fmm_order=p, unsort=1, dens_inhom=inhom, p2p_mutual=1)
    _, a_mut2 = run_gpu(engine, buf, par, n, fmm_order=p, unsort=1, dens_inhom=inhom, p2p_mutual=1)
    assert force_err(a_mut, a_ref) < 1e-5
    assert force_err(a_mut, a_one) < 2e-6
    np.testing.assert_array_equal(a_mut, a_mut2)
    assert engine.kd_info().directed_p2p == info.directed_p2p
    assert engine.kd_info().p2p_halves == halves, (engine.kd_info().p2p_halves, info.mlt_max)


def test_mutual_near_field_tree_order_and_reuse(engine, oracle32):
    """unsort = 0 with tree reuse: several leapfrog steps with the mutual kernel stay within rounding of the one-directional run"""
    import torch
    from coulomb_oscillators_amd import EVAL_FMM_KDTREE, INTEG_LEAPFROG
    n, p = 32768, 6
    buf = oracle32.init_reference(n)
    par = dev(oracle32.params(n))
    out = []
    for mutual in (0, 1):
        # (switching the kernel drops the tree: a reused tree belongs to the state it was built on)
        engine.set(fmm_order=p, unsort=0, tree_steps=8, p2p_mutual=mutual)   # one build, seven evaluations on it: the particle order stays comparable
        d = dev(buf.copy())
        engine.compute_force(EVAL_FMM_KDTREE, d, n, par)
        for _ in range(6):
            engine.integrate(INTEG_LEAPFROG, EVAL_FMM_KDTREE, d, n, par, 5e-4)
        out.append(d.cpu().numpy())
    assert np.isfinite(out[1]).all()
    # same particle order: both runs build their tree once, from identical positions
    np.testing.assert_allclose(out[1][0], out[0][0], rtol=0, atol=1e-6 * np.abs(out[0][0]).max())
    assert force_err(out[1][2], out[0][2]) < 1e-5


def test_three_pass_select_is_reached_and_succeeds(engine, oracle32):
    """build_mode 1 as a SUCCESSFUL mode (not a stop on the way to the sorting build): 300 particles whose split coordinates are
    distinct fp32 values packed into 1e-10 of a unit box sit around the root's median.  Two radix passes over the box-linear
    key (22 bits across the box) leave all of them in the pivot's bucket -- more candidates than the resolver takes -- so the
    context escalates to three passes over the ordered float bits, which tell them apart exactly; there is no exact tie, so the
    sorting build is never needed.  The tree is the oracle's, bit for bit, in that mode too."""
    o = oracle32
    n, p = 65536, 4
    rng = np.random.default_rng(5)
    pos = (rng.random((n, 3), dtype=np.float32) - np.float32(0.5)) * np.array([1.0, 0.8, 0.8], dtype=np.float32)
    k = 300
    half = (n - k) // 2
    x = np.sort(np.abs(pos[:, 0]) + np.float32(1e-3))          # strictly away from the cluster
    pos[:half, 0] = -x[:half]
    pos[half:n - k, 0] = x[half:n - k]
    pos[n - k:, 0] = (np.arange(k, dtype=np.float64) * 3e-13).astype(np.float32)       # distinct floats in [0, 9e-11]
    assert len(np.unique(pos[n - k:, 0])) == k
    pos = pos[rng.permutation(n)]
    buf = np.zeros((3, n, 3), dtype=np.float32)
    buf[0] = pos
    par = o.params(n)
    _, a_ref = o.fmm_kd(buf[:2], par, p=p, threads=8, unsort=True)
    want = o.kd_tree()
    _, a = run_gpu(engine, buf, par, n, fmm_order=p, unsort=1)
    info = engine.kd_info()
    assert info.build_mode == 1, info.build_mode
    assert (info.L, info.ntot) == (want["L"], want["ntot"])
    for name in ("index", "mult", "splitdim", "lbound", "rbound", "center"):
        np.testing.assert_array_equal(engine.kd_array(name), want[name], err_msg=name)
    np.testing.assert_array_equal(engine.kd_array("unsort"), o.kd_unsort(n))
    for name in ("p2p", "m2l"):
        np.testing.assert_array_equal(canon_pairs(engine.kd_array(name)), canon_pairs(want[name]), err_msg=name)
    assert force_err(a, a_ref) < 1e-5
    # and it stays there: a second evaluation builds with three passes from the start, same tree
    _, a2 = run_gpu(engine, buf, par, n, fmm_order=p, unsort=1)
    assert engine.kd_info().build_mode == 1
    np.testing.assert_array_equal(a, a2)


def test_eight_million_particles_properties(oracle32):
    """N = 2^23 (the sizes BASELINE config 4 is made of): the level-0 and level-1 nodes are above 2^22 particles, where the median
    selection takes its third radix pass (k_kdselect.hip: kd_select_level) in the DEFAULT build mode.  The oracle takes minutes
    here, so the checks are the ones that do not need it: closed forms of index / mult, the permutation, the split property of
    every node of the top levels (nothing left of a cut lies right of it), leaves inside their boxes, build_mode 0 with no tie
    flag, fp64 direct sums for a sample of particles, and -- with a rebuild every step for three leapfrog steps -- warm builds
    following the three-pass one with bit-reproducible results."""
    import torch
    from coulomb_oscillators_amd import Engine, EVAL_FMM_KDTREE, INTEG_LEAPFROG
    o = oracle32
    n, p = 1 << 23, 6
    buf = o.init_reference(n)
    par = o.params(n)
    eng = Engine(fmm_order=p, unsort=0, tree_steps=1, sync=0)
    d = dev(buf)
    prm = dev(par)
    eng.compute_force(EVAL_FMM_KDTREE, d, n, prm)
    torch.cuda.synchronize()
    info = eng.kd_info()
    L = info.L
    assert info.build_mode == 0 and info.n == n and L == 18 and info.ntot == (1 << (L + 1)) - 1
    mult, index, sd = eng.kd_array("mult"), eng.kd_array("index"), eng.kd_array("splitdim")
    lb, rb = eng.kd_array("lbound"), eng.kd_array("rbound")
    for l in (0, 1, 2, 5, L):
        beg, cnt = (1 << l) - 1, 1 << l
        np.testing.assert_array_equal(mult[beg:beg + cnt], np.full(cnt, n >> l))
        np.testing.assert_array_equal(index[beg:beg + cnt], np.arange(cnt, dtype=np.int64) * (n >> l))
    perm = eng.kd_array("unsort")
    assert np.array_equal(np.sort(perm), np.arange(n))
    # state is in tree order: node j of level l owns rows [index, index + mult)
    pos = d[0].cpu().numpy()
    np.testing.assert_array_equal(pos, buf[0][perm])
    for l in range(0, 6):
        for j in range(1 << l):
            node = (1 << l) - 1 + j
            a0, m = int(index[node]), int(mult[node])
            ax = int(sd[node])
            left, right = pos[a0:a0 + m // 2, ax], pos[a0 + m // 2:a0 + m, ax]
            assert left.max() <= right.min(), (l, j)
            assert np.all(pos[a0:a0 + m].min(0) >= lb[node]) and np.all(pos[a0:a0 + m].max(0) <= rb[node])
    leaf0 = (1 << L) - 1
    P = pos.reshape(1 << L, 32, 3)
    assert np.all(P.min(1) >= lb[leaf0:]) and np.all(P.max(1) <= rb[leaf0:])
    # sampled fp64 direct sums (on the GPU: 2^23 sources per sample)
    acc = d[2].cpu().numpy().astype(np.float64)
    P64 = d[0].double()
    rows = np.random.default_rng(4).choice(n, 32, replace=False)
    want = np.empty((len(rows), 3))
    for k, i in enumerate(rows):
        dd = P64[int(i)] - P64
        w = ((dd * dd).sum(1) + 1e-18) ** -1.5
        w[int(i)] = 0
        want[k] = (dd * w[:, None]).sum(0).cpu().numpy() * float(par[0])
    # d[2] holds Coulomb + elastic (nbco_force): take the elastic term off again
    coul = acc[rows] + np.asarray(par[3:6], dtype=np.float64) * pos[rows].astype(np.float64)
    rel = np.linalg.norm(coul - want, axis=1) / (np.linalg.norm(want, axis=1) + np.linalg.norm(want, axis=1).mean())
    assert np.median(rel) < 6e-3 and rel.max() < 5e-2, (np.median(rel), rel.max())
    del P64
    # three steps with a rebuild each: the builds behind the first one are warm; same trajectory from a second context
    start = dev(buf)
    finals = []
    for rep in range(2):
        e2 = Engine(fmm_order=p, unsort=0, tree_steps=1, sync=0)
        s = start.clone()
        e2.compute_force(EVAL_FMM_KDTREE, s, n, prm)
        e2.integrate_steps(INTEG_LEAPFROG, EVAL_FMM_KDTREE, s, n, prm, 5e-4, 3)
        torch.cuda.synchronize()
        i2 = e2.kd_info()
        # (at this size the first warm windows -- sized for ~100 elements around the predicted pivot -- can miss: the evaluation is
        # repeated with the cold select and the window widened, nothing escalates; the trajectory is the cold one either way)
        assert i2.build_mode == 0 and i2.warm_builds >= 2 and i2.warm_misses <= i2.warm_builds, (i2.warm_builds, i2.warm_misses)
        assert bool(torch.isfinite(s).all())
        finals.append(s)
        e2.close()
    assert torch.equal(finals[0], finals[1])
    eng.close()


# ---- fp64 far field for the kd-tree evaluator (BASELINE config 5: "fp64 far-field / fp32 P2P"; reference: the -DSCAL=double build,
#      constants.cuh:22-24, operators fmm_cart_base3.cuh:1181-1208) -------------------------------------------------------------
@pytest.mark.parametrize("n,p,kind", [(8192, 6, "gauss"), (8192, 10, "gauss"), (20000, 8, "cube"), (4096, 3, "cube")])
def test_far_fp64_kdtree_against_both_oracles(engine, oracle32, oracle64, n, p, kind):
    """opts.far_fp64 with nbco_fmm_kdtree: the tree, the permutation and the lists are the fp32 ones, bit for bit (the geometry does
    not change with the option); the multipole / local tuples are doubles and follow the REAL = double oracle to the rounding of
    the fp32 centres; the accelerations are within 1e-5 of the double oracle wherever its lists equal the fp32 ones (they differ
    only when a node pair sits within fp32 rounding of the opening criterion), never further from it than the all-fp32
    evaluation, and within 1e-5 of the fp32 oracle."""
    o32, o64 = oracle32, oracle64
    buf = o32.init_reference(n, test_mode=(kind == "cube"))
    par = o32.params(n)
    _, want32 = o32.fmm_kd(buf[:2], par, p=p, threads=8, unsort=True)
    tree32 = o32.kd_tree()
    offM, offL = p * (p + 1) * (p + 2) // 6, (p + 1) ** 2
    _, want64 = o64.fmm_kd(buf[:2].astype(np.float64), par.astype(np.float64), p=p, threads=8, unsort=True)
    tree64 = o64.kd_tree(offM=offM, offL=offL)
    _, got32 = run_gpu(engine, buf, par, n, fmm_order=p, unsort=1, far_fp64=0)
    assert engine.kd_info().real_bytes == 4
    _, got = run_gpu(engine, buf, par, n, fmm_order=p, unsort=1, far_fp64=1)
    info = engine.kd_info()
    assert info.real_bytes == 8 and (info.L, info.ntot) == (tree32["L"], tree32["ntot"])
    for name in ("index", "mult", "splitdim", "lbound", "rbound", "center"):
        np.testing.assert_array_equal(engine.kd_array(name), tree32[name], err_msg=name)
    for name in ("p2p", "m2l"):
        np.testing.assert_array_equal(canon_pairs(engine.kd_array(name)), canon_pairs(tree32[name]), err_msg=name)
    mp, lc = engine.kd_array("mpole"), engine.kd_array("local")
    assert mp.dtype == np.float64 and lc.dtype == np.float64
    assert np.isfinite(got).all() and np.isfinite(mp).all() and np.isfinite(lc).all()
    assert force_err(got, want32) < 1e-5
    same_lists = all(np.array_equal(canon_pairs(tree64[k]), canon_pairs(tree32[k])) for k in ("p2p", "m2l"))
    if same_lists:
        e64, e32 = force_err(got, want64), force_err(got32, want64)
        assert e64 < 1e-5 and e64 < 1.5 * e32 + 2e-7
        # the tuples differ from the double oracle's by the rounding of the fp32 centres (6e-8 of a coordinate that is ~100 leaf
        # sizes from the origin), which a term of order k sees k times: 2e-5 up to order 6, 1e-4 at order 10
        tol = 2e-5 if p <= 6 else 1e-4
        w = tree64["mpole"]
        sc = np.abs(w).max(axis=0, keepdims=True).clip(1e-300)
        assert (np.abs(mp - w) / sc).max() < tol
        w = tree64["local"]
        sc = np.abs(w).max(axis=0, keepdims=True).clip(1e-300)
        assert (np.abs(lc - w) / sc).max() < tol
    engine.set(far_fp64=0)


def test_far_fp64_kdtree_keeps_order_10_in_range_at_a_million_particles(oracle32):
    """p = 10 at N = 2^20 on the BASELINE ball: the fp32 far field leaves its range (r^-11 19!! ~ 1e40, SURVEY N8: the kd-tree
    evaluator returns NaN / inf there, as the reference does), the fp64 far field stays finite, follows the fp64 direct sum to the
    truncation level of the default opening radius, is bit-reproducible, and feeds nbco_energy_fmm."""
    import torch
    from coulomb_oscillators_amd import Engine, EVAL_FMM_KDTREE
    o = oracle32
    n, p = 1 << 20, 10
    buf = o.init_reference(n)
    par = o.params(n)
    eng = Engine(fmm_order=p, unsort=0, far_fp64=1, sync=0)
    d = dev(buf)
    prm = dev(par)
    eng.fmm_cart3_kdtree(d, d[2], n, prm)
    torch.cuda.synchronize()
    assert eng.kd_info().real_bytes == 8
    a = d[2].clone()
    assert bool(torch.isfinite(a).all())
    rows = np.random.default_rng(8).choice(n, 32, replace=False)
    P64 = d[0].double()
    want = np.empty((len(rows), 3))
    for k, i in enumerate(rows):
        dd = P64[int(i)] - P64
        w = ((dd * dd).sum(1) + 1e-18) ** -1.5
        w[int(i)] = 0
        want[k] = (dd * w[:, None]).sum(0).cpu().numpy() * float(par[0])
    got = a.cpu().numpy().astype(np.float64)[rows]
    rel = np.linalg.norm(got - want, axis=1) / (np.linalg.norm(want, axis=1) + np.linalg.norm(want, axis=1).mean())
    assert np.median(rel) < 2e-3 and rel.max() < 3e-2, (np.median(rel), rel.max())     # p = 10 at the default radius (p = 6: 6e-3)
    e = eng.energy_fmm(d, n, prm)
    assert np.isfinite(e).all() and e[2] > 0
    d2 = dev(buf)
    eng2 = Engine(fmm_order=p, unsort=0, far_fp64=1, sync=0)
    eng2.fmm_cart3_kdtree(d2, d2[2], n, prm)
    torch.cuda.synchronize()
    assert torch.equal(d2[2], a)
    # the all-fp32 far field at this order and size: out of range
    eng2.set(far_fp64=0)
    d3 = dev(buf)
    eng2.fmm_cart3_kdtree(d3, d3[2], n, prm)
    torch.cuda.synchronize()
    assert not bool(torch.isfinite(d3[2]).all())
    eng.close(); eng2.close()
