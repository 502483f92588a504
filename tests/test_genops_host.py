"""CPU tests of the GENERATED far-field operators -- the text the gfx950 kernels compile (csrc/fmm_ops_gen.inc from gen_ops.py,
csrc/m2l_gen.inc from gen_m2l.py), built for the host by csrc/genops_host.cpp (libnbco_genops_host.so, g++) -- against the
oracle's operators, orders 1..10, fp64 (to rounding) and fp32.  No GPU involved: this is what tells a generator bug from a
kernel bug before anything runs on the device."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBPATH = os.path.join(ROOT, "coulomb_oscillators_amd", "libnbco_genops_host.so")

sym_off = lambda n: n * (n + 1) * (n + 2) // 6
tl_off = lambda n: n * n


@pytest.fixture(scope="module")
def gen(engine_lib):
    if not os.path.exists(LIBPATH):
        pytest.fail("libnbco_genops_host.so is not built (make -C coulomb_oscillators_amd/csrc)")
    return C.CDLL(LIBPATH)


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def oracle_ops(o):
    L, r, P = o.lib, o.creal, C.c_void_p
    L.oracle_op_p2m.argtypes = [P, C.c_int, P, C.c_int, P]
    L.oracle_op_m2m.argtypes = [P, P, C.c_int, P]
    L.oracle_op_m2l.argtypes = [P, P, C.c_int, P, r]
    L.oracle_op_l2l.argtypes = [P, P, C.c_int, P]
    L.oracle_op_l2p.argtypes = [P, P, C.c_int, P]
    L.oracle_op_p2m_tl.argtypes = [P, C.c_int, P, C.c_int, P]
    L.oracle_op_m2m_tl.argtypes = [P, P, C.c_int, P]
    return L


def rel(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max() / (np.abs(np.asarray(b, dtype=np.float64)).max() + 1e-300))


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("p", range(1, 11))
def test_generated_kd_operators_match_the_oracle(gen, oracle32, oracle64, p, dt):
    o = oracle64 if dt == np.float64 else oracle32
    L = oracle_ops(o)
    sfx = "f64" if dt == np.float64 else "f32"
    tol = 2e-12 if dt == np.float64 else 3e-5
    fn = lambda name: getattr(gen, "nbco_genop_%s_%s" % (name, sfx))
    rng = np.random.default_rng(500 + p)
    pts = (rng.standard_normal((11, 3)) * 0.25).astype(dt)
    c = pts.mean(axis=0).astype(dt)
    offM, offL = sym_off(p), tl_off(p + 1)
    # P2M: orders 0 (count) and 1 (zero) by the kernel's convention, 2..p-1 by the generated body
    M = np.zeros(max(offM, 1), dtype=dt)
    assert fn("p2m")(p, ptr(pts), len(pts), ptr(c), ptr(M)) == 0
    Mo = np.zeros(max(offM, 1), dtype=dt)
    L.oracle_op_p2m(ptr(Mo), p, ptr(pts), len(pts), ptr(c))
    Mo[0] = len(pts)
    assert M[0] == len(pts) and (offM < 4 or not M[1:4].any())
    if p > 2:
        assert rel(M[4:], Mo[4:]) < tol
    # M2M: a child's tuple shifted to the parent's centre
    d = (rng.standard_normal(3) * 0.2).astype(dt)
    Mp = np.zeros(max(offM, 1), dtype=dt)
    assert fn("m2m")(p, ptr(Mo), ptr(d), ptr(Mp)) == 0
    Mpo = np.zeros(max(offM, 1), dtype=dt)
    L.oracle_op_m2m(ptr(Mpo), ptr(Mo), p, ptr(d))
    if p > 2:
        assert rel(Mp[4:], Mpo[4:]) < tol
    # M2L: one source node seen from a target centre
    dd = np.array([1.4, -0.8, 1.1], dtype=dt) + (rng.standard_normal(3) * 0.1).astype(dt)
    Lg, Lo = np.zeros(offL, dtype=dt), np.zeros(offL, dtype=dt)
    if dt == np.float32:
        gen.nbco_genop_m2l_f32.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    else:
        gen.nbco_genop_m2l_f64.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
    assert fn("m2l")(p, ptr(Mo), ptr(dd), 0.0, ptr(Lg)) == 0
    L.oracle_op_m2l(ptr(Lo), ptr(Mo), p, ptr(dd), 0.0)
    # per order, relative to the largest component of that order
    for n in range(1, p + 1):
        sl = slice(tl_off(n), tl_off(n + 1))
        assert rel(Lg[sl], Lo[sl]) < tol * (1 if dt == np.float64 else 3), (n, Lg[sl], Lo[sl])
    # L2L and L2P on that local expansion
    d2 = (rng.standard_normal(3) * 0.15).astype(dt)
    Og, Oo = np.zeros(offL, dtype=dt), np.zeros(offL, dtype=dt)
    assert fn("l2l")(p, ptr(Lo), ptr(d2), ptr(Og)) == 0
    L.oracle_op_l2l(ptr(Oo), ptr(Lo), p, ptr(d2))
    for n in range(1, p + 1):
        sl = slice(tl_off(n), tl_off(n + 1))
        assert rel(Og[sl], Oo[sl]) < tol * (1 if dt == np.float64 else 3), n
    fg, fo = np.zeros(3, dtype=dt), np.zeros(3, dtype=dt)
    assert fn("l2p")(p, ptr(Lo), ptr(d2), ptr(fg)) == 0
    L.oracle_op_l2p(ptr(fo), ptr(Lo), p, ptr(d2))
    assert rel(fg, fo) < tol * (1 if dt == np.float64 else 3)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("p", range(2, 11))
def test_generated_octree_operators_match_the_oracle(gen, oracle32, oracle64, p, dt):
    """traceless P2M (orders 2..p) and M2M of the octree evaluator (fmm_cart3_traceless.cuh:61-168)"""
    # the yardstick is the fp64 oracle on the same (fp32-valued) inputs: the high orders are sums with cancellation, and two fp32
    # evaluations of them agree no better than either agrees with the exact value
    L = oracle_ops(oracle64)
    sfx = "f64" if dt == np.float64 else "f32"
    tol = 2e-12 if dt == np.float64 else 2e-4
    rng = np.random.default_rng(600 + p)
    pts = (rng.standard_normal((9, 3)) * 0.3).astype(dt)
    c = pts.mean(axis=0).astype(dt)
    off = tl_off(p + 1)
    A, Ao = np.zeros(off, dtype=dt), np.zeros(off, dtype=np.float64)
    assert getattr(gen, "nbco_genop_p2m_tl_" + sfx)(p, ptr(pts), len(pts), ptr(c), ptr(A)) == 0
    pts64, c64 = pts.astype(np.float64), c.astype(np.float64)
    L.oracle_op_p2m_tl(ptr(Ao), p, ptr(pts64), len(pts), ptr(c64))
    for n in range(2, p + 1):
        sl = slice(tl_off(n), tl_off(n + 1))
        assert rel(A[sl], Ao[sl]) < tol, n
    Ao[0] = len(pts)
    Ao[1:4] = 0
    d = (rng.standard_normal(3) * 0.2).astype(dt)
    B, Bo = np.zeros(off, dtype=dt), np.zeros(off, dtype=np.float64)
    Ain = Ao.astype(dt)
    assert getattr(gen, "nbco_genop_m2m_tl_" + sfx)(p, ptr(Ain), ptr(d), ptr(B)) == 0
    d64, Ain64 = d.astype(np.float64), Ain.astype(np.float64)
    L.oracle_op_m2m_tl(ptr(Bo), ptr(Ain64), p, ptr(d64))
    for n in range(2, p + 1):
        sl = slice(tl_off(n), tl_off(n + 1))
        assert rel(B[sl], Bo[sl]) < tol * 3, n
