"""CPU tests: pin the oracle against the reference outputs recorded in BASELINE.md / SURVEY.md
(tests/golden/reference_recorded.json) and against closed-form invariants."""
import json
import os

import numpy as np
import pytest

from nbutil import force_err

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def recorded():
    with open(os.path.join(GOLD, "reference_recorded.json")) as f:
        return json.load(f)


def test_reference_test_mode_error_table(oracle32, recorded):
    """`nbco3 -cpu -n 4096 -test`: the oracle reproduces all ten recorded mean relative errors."""
    o = oracle32
    rec = recorded["test_mode_relerr"]
    n = rec["n"]
    buf = o.init_reference(n, test_mode=True)
    par = o.params(n)
    ref = o.direct3(buf[0], par, threads=4)
    for p, want in enumerate(rec["values"], start=1):
        _, a = o.fmm_kd(buf[:2], par, p=p, threads=4, unsort=True)
        got = o.mean_relerr(a, ref)
        # recorded values carry 4 significant digits
        assert abs(got - want) <= 6e-4 * want, (p, got, want)


@pytest.mark.parametrize("n", [4096, 32768])
def test_reference_interaction_lists(oracle32, recorded, n):
    """List sizes and directed pair counts recorded from the reference's CPU traversal (exact)."""
    o = oracle32
    rec = recorded["gaussian_p6_lists"][str(n)]
    buf = o.init_reference(n)
    _, _ = o.fmm_kd(buf[:2], o.params(n), p=6, threads=4, unsort=False)
    t = o.kd_tree()
    assert t["L"] == rec["L"]
    assert len(t["p2p"]) == rec["p2p"]
    assert len(t["m2l"]) == rec["m2l"]
    mult = t["mult"].astype(np.int64)
    pairs = int((2 * mult[t["p2p"][:, 0]] * mult[t["p2p"][:, 1]]).sum() + (mult[(1 << t["L"]) - 1:] ** 2).sum())
    assert pairs == rec["pairs"]


def test_kd_ranges_and_mult(oracle32):
    """index[j] = ceil(n i / 2^l) (fmm_cart3_kdtree.cuh:117-118), mult = differences (appel.cuh:184-197)."""
    o = oracle32
    for n in (4096, 5000, 30001):
        buf = o.init_reference(n)
        o.fmm_kd(buf[:2], o.params(n), p=4, threads=2, unsort=True)
        t = o.kd_tree()
        L = t["L"]
        assert L == o.lib.oracle_kd_levels(n, 4, 1.0)
        for l in range(1, L + 1):
            m = 1 << l
            want = np.array([0 if i == 0 else (n * i - 1) // m + 1 for i in range(m)])
            np.testing.assert_array_equal(t["index"][m - 1:2 * m - 1], want)
        leaves = t["mult"][(1 << L) - 1:]
        assert leaves.sum() == n and leaves.max() - leaves.min() <= 1
        assert t["mult"][0] == n


def test_direct_variants_agree_with_fp64(oracle32, oracle64):
    o32, o64 = oracle32, oracle64
    n = 2048
    buf = o32.init_reference(n)
    par32 = o32.params(n)
    a3 = o32.direct3(buf[0], par32)
    a2 = o32.direct2(buf[0], par32, threads=3)
    a64 = o64.direct3(buf[0].astype(np.float64), o64.params(n))
    assert force_err(a3, a64) < 1e-6
    assert force_err(a2, a64) < 2e-5
    # the self term contributes exactly zero (direct.cuh:160, d = 0)
    one = o32.direct3(buf[0][:1], par32)
    assert np.all(one == 0)


def test_fmm_converges_to_direct(oracle32):
    o = oracle32
    n = 4096
    buf = o.init_reference(n)
    par = o.params(n)
    ref = o.direct3(buf[0], par, threads=4)
    errs = []
    for p in (2, 4, 6, 8, 10):
        _, a = o.fmm_kd(buf[:2], par, p=p, threads=1, unsort=True)
        errs.append(o.mean_relerr(a, ref))
    assert all(errs[i + 1] < errs[i] for i in range(len(errs) - 1)), errs
    assert errs[2] < 3e-3 and errs[-1] < 1e-4, errs     # 1.36e-3 (p=6), 2.8e-5 (p=10) on another ball


def test_fmm_thread_count_only_changes_rounding(oracle32):
    o = oracle32
    n = 4096
    buf = o.init_reference(n)
    par = o.params(n)
    _, a1 = o.fmm_kd(buf[:2], par, p=5, threads=1, unsort=True)
    _, a1b = o.fmm_kd(buf[:2], par, p=5, threads=1, unsort=True)
    _, a8 = o.fmm_kd(buf[:2], par, p=5, threads=8, unsort=True)
    np.testing.assert_array_equal(a1, a1b)              # single thread is bit-reproducible
    assert np.abs(a8 - a1).max() <= 2e-5 * np.abs(a1).max()


def test_unsort_false_permutes_velocities_consistently(oracle32):
    o = oracle32
    n = 1000
    buf = o.init_reference(n)
    par = o.params(n)
    pv, a = o.fmm_kd(buf[:2], par, p=3, threads=1, unsort=False)
    perm = o.kd_unsort(n)
    np.testing.assert_array_equal(pv[0], buf[0][perm])
    np.testing.assert_array_equal(pv[1], buf[1][perm])
    _, a_u = o.fmm_kd(buf[:2], par, p=3, threads=1, unsort=True)
    np.testing.assert_array_equal(a, a_u[perm])


def test_octree_traceless_accuracy(oracle32):
    """octree-traceless mean error 2.6e-4 (p=6) / 1.3e-5 (p=10) at N=4096."""
    o = oracle32
    n = 4096
    buf = o.init_reference(n)
    par = o.params(n)
    for p, bound in ((6, 8e-4), (10, 6e-5)):
        pv, a = o.fmm_oct_traceless(buf[:2], par, p=p, threads=4)
        ref = o.direct3(pv[0], par, threads=4)
        assert o.mean_relerr(a, ref) < bound


def test_integrators_order_and_energy(oracle64):
    """integrator.cuh:32-167: energy error of Euler / leapfrog / Forest-Ruth / PEFRL shrinks with
    the scheme's order when dt is halved (smooth potential: EPS2 = 1e-4)."""
    from oracle import pyoracle as po
    o = oracle64
    n, eps2 = 64, 1e-4
    base = o.init_reference(n)
    par = o.params(n)

    def drift(scheme, dt):
        buf = base.copy()
        o.compute_force(po.KIND_DIRECT3, buf, par, eps2=eps2)
        e0 = o.energy(buf, par, eps2=eps2).sum()
        for _ in range(int(round(2.0 / dt))):
            o.integrate(scheme, po.KIND_DIRECT3, buf, par, dt, eps2=eps2)
        return abs(o.energy(buf, par, eps2=eps2).sum() - e0) / abs(e0)

    for scheme, min_ratio in ((po.SCHEME_EULER, 1.5), (po.SCHEME_LEAPFROG, 3.2), (po.SCHEME_FR, 12.0), (po.SCHEME_PEFRL, 12.0)):
        e1, e2 = drift(scheme, 0.2), drift(scheme, 0.1)
        assert e1 / e2 > min_ratio, (scheme, e1, e2)
    # pre_symplectic_euler is F K D: one step equals compute_force + K + D
    a = base.copy(); b = base.copy()
    o.integrate(po.SCHEME_PRE_EULER, po.KIND_DIRECT3, a, par, 0.01, eps2=eps2)
    o.compute_force(po.KIND_DIRECT3, b, par, eps2=eps2)
    o.step(b[1], b[2], 0.01); o.step(b[0], b[1], 0.01)
    np.testing.assert_array_equal(a, b)


def test_init_reference_statistics(oracle32):
    """main3.cu:71-137: centred, RMS-normalised Gaussian ball."""
    o = oracle32
    n = 4096
    buf = o.init_reference(n)
    for arr, sig in ((buf[0], (0.003, 0.001, 0.01)), (buf[1], (0.003 * 1.095, 0.001, 0.01))):
        assert np.abs(arr.mean(axis=0)).max() < 1e-7
        np.testing.assert_allclose(np.sqrt((arr.astype(np.float64) ** 2).mean(axis=0)), sig, rtol=1e-4)
