"""CPU tests of the flattened FMM operator tables (csrc/fmm_tables.cpp, exported through
nbco_debug_table): a small numpy interpreter applies each table exactly the way the HIP kernels do
and the result is compared with the oracle's restatement of the reference operators
(fmm_cart_base3.cuh P2M :908, M2M :1042, gradient :698, M2L :1181, L2L :1348/:1365, L2P :1511/:1531)."""
import ctypes as C

import numpy as np
import pytest

ORDERS = [1, 2, 3, 4, 6, 8, 10]


class Tab:
    def __init__(self, lib_path, order):
        self.lib = C.CDLL(lib_path)
        self.lib.nbco_debug_table.argtypes = [C.c_int, C.c_char_p, C.c_void_p, C.c_longlong, C.POINTER(C.c_longlong)]
        self.P = order
        self.offM = order * (order + 1) * (order + 2) // 6
        self.offL = (order + 1) ** 2
        self.nfull = (order + 1) * (order + 2) * (order + 3) // 6

    def get(self, name, dtype):
        cnt = C.c_longlong()
        assert self.lib.nbco_debug_table(self.P, name.encode(), None, 0, C.byref(cnt)) == 0, name
        out = np.zeros(cnt.value, dtype=dtype)
        if cnt.value:
            assert self.lib.nbco_debug_table(self.P, name.encode(), out.ctypes.data_as(C.c_void_p), cnt.value, C.byref(cnt)) == 0
        return out

    def monomials(self, d):
        rec = self.get("mono_rec", np.uint32)
        D = np.zeros(self.nfull, dtype=np.float64)
        D[0] = 1.0
        for i in range(1, self.nfull):
            D[i] = D[rec[i] & 0xFFFF] * d[(rec[i] >> 16) & 3]
        return D

    def refine(self, F):
        st = self.get("rf_start", np.int32)
        dst, a, b = self.get("rf_dst", np.uint32), self.get("rf_a", np.uint32), self.get("rf_b", np.uint32)
        for z in range(2, self.P + 1):
            for e in range(st[z], st[z + 1]):
                F[dst[e]] = -(F[a[e]] + F[b[e]])
        return F

    def csr(self, name):
        return self.get(name + "_start", np.int32), self.get(name + "_idx", np.uint32), self.get(name + "_coef", np.float32)


@pytest.fixture(scope="module")
def rng():
    return np.random.default_rng(11)


@pytest.mark.parametrize("P", ORDERS)
def test_p2m_and_m2m_tables(engine_lib, oracle64, rng, P):
    o, t = oracle64, Tab(engine_lib, P)
    if t.offM == 0:
        return
    pts = rng.standard_normal((9, 3)) * 0.3
    c = pts.mean(axis=0)
    want = np.zeros(max(t.offM, 4))
    o.lib.oracle_op_p2m(o.ptr(want), P, o.ptr(pts), len(pts), o.ptr(c))
    coef = t.get("p2m_coef", np.float32).astype(np.float64)
    got = np.zeros(t.offM)
    for q in pts:
        got += coef[:t.offM] * t.monomials(q - c)[:t.offM]
    np.testing.assert_allclose(got, want[:t.offM], rtol=2e-6, atol=1e-12)

    # M2M: shift a child expansion (order 0 = charge count, dipole = 0) by d
    child = want[:t.offM].copy()
    child[0] = len(pts)
    d = np.array([0.21, -0.13, 0.34])
    wantp = np.zeros(max(t.offM, 4))
    o.lib.oracle_op_m2m(o.ptr(wantp), o.ptr(child), P, o.ptr(d))
    st, idx, cf = t.csr("m2m")
    D = t.monomials(d)
    gotp = np.zeros(t.offM)
    for out in range(t.offM):
        for e in range(st[out], st[out + 1]):
            gotp[out] += float(cf[e]) * D[idx[e] & 0xFFFF] * child[idx[e] >> 16]
    np.testing.assert_allclose(gotp, wantp[:t.offM], rtol=2e-6, atol=1e-12)


def apply_m2l(t, M, dvec, eps2):
    r = np.sqrt((dvec ** 2).sum() + eps2)
    u = dvec / r
    rinv = 1.0 / r
    gst, gexp, gcf = t.get("gp_start", np.int32), t.get("gp_exp", np.uint32), t.get("gp_coef", np.float32)
    tl2full = t.get("tl2full", np.int32)
    F = np.zeros(t.nfull)
    for e in range(1, t.offL):
        s = 0.0
        for k in range(gst[e], gst[e + 1]):
            ex, ey, ez = gexp[k] & 0xFF, (gexp[k] >> 8) & 0xFF, (gexp[k] >> 16) & 0xFF
            s += float(gcf[k]) * u[0] ** int(ex) * u[1] ** int(ey) * u[2] ** int(ez)
        F[tl2full[e]] = s
    t.refine(F)
    m_order = t.get("m_order", np.int32)
    Ms = M * rinv ** m_order[:len(M)]
    st, idx, cf = t.csr("m2l")
    tl_order = t.get("tl_order", np.int32)
    L = np.zeros(t.offL)
    for out in range(1, t.offL):
        acc = 0.0
        for e in range(st[out], st[out + 1]):
            acc += float(cf[e]) * Ms[idx[e] & 0xFFFF] * F[idx[e] >> 16]
        L[out] = acc * rinv ** (tl_order[out] + 1)
    return L, F, r


@pytest.mark.parametrize("P", ORDERS)
def test_gradient_and_m2l_tables(engine_lib, oracle64, rng, P):
    o, t = oracle64, Tab(engine_lib, P)
    M = np.zeros(max(t.offM, 1))
    # physically scaled multipoles: order-k moments of a source cluster of size ~0.25 (r ~ 1.4)
    m_order = t.get("m_order", np.int32)
    import math
    kfact = np.array([math.factorial(int(k)) for k in m_order[:len(M)]], dtype=np.float64)
    M[:] = rng.standard_normal(len(M)) * 0.25 ** m_order[:len(M)] / kfact      # P2M carries 1/k!
    M[0] = 7.0
    if t.offM >= 4:
        M[1:4] = 0.0            # no dipole about the centre of charge
    dvec = np.array([0.7, -1.1, 0.45])
    L, F, r = apply_m2l(t, M[:max(t.offM, 1)], dvec, 1e-18)
    # gradient tensors: oracle returns c * grad^n(1/r) in the full layout of order n
    u = dvec / r
    for n in range(1, P + 1):
        g = np.zeros((n + 1) * (n + 2) // 2)
        o.lib.oracle_op_gradient(o.ptr(g), n, o.ptr(u), r, 1.0)
        off = n * (n + 1) * (n + 2) // 6
        # fp32 table coefficients: cancellation in the high-order harmonic polynomials costs up to ~1e-4 of the largest component at order 10
        np.testing.assert_allclose(F[off:off + len(g)] * r ** (-n - 1), g, rtol=3e-6, atol=2e-4 * np.abs(g).max())
    want = np.zeros(t.offL)
    Mfull = np.zeros(max(t.offM, 4))
    Mfull[:len(M)] = M
    o.lib.oracle_op_m2l(o.ptr(want), o.ptr(Mfull), P, o.ptr(dvec), 1e-18)
    np.testing.assert_allclose(L[1:], want[1:], rtol=2e-5, atol=1e-6 * np.abs(want).max())


@pytest.mark.parametrize("P", ORDERS)
def test_l2l_and_l2p_tables(engine_lib, oracle64, rng, P):
    o, t = oracle64, Tab(engine_lib, P)
    # a physically traceless local expansion: produce it with the M2L operator itself
    M = np.zeros(max(t.offM, 4))
    M[0] = 5.0
    if t.offM > 4:
        M[4:t.offM] = rng.standard_normal(t.offM - 4) * 0.05
    Lp = np.zeros(t.offL)
    o.lib.oracle_op_m2l(o.ptr(Lp), o.ptr(M), P, o.ptr(np.array([1.3, 0.4, -0.9])), 1e-18)
    tl2full = t.get("tl2full", np.int32)
    F = np.zeros(t.nfull)
    F[tl2full[1:]] = Lp[1:]
    t.refine(F)
    # L2L
    d = np.array([0.11, -0.07, 0.05])
    want = np.zeros(t.offL)
    o.lib.oracle_op_l2l(o.ptr(want), o.ptr(Lp), P, o.ptr(d))
    st, idx, cf = t.csr("l2l")
    D = t.monomials(d)
    got = np.zeros(t.offL)
    for out in range(1, t.offL):
        for e in range(st[out], st[out + 1]):
            got[out] += float(cf[e]) * F[idx[e] & 0xFFFF] * D[idx[e] >> 16]
    np.testing.assert_allclose(got[1:], want[1:], rtol=3e-6, atol=2e-9 * np.abs(want).max())
    # L2P
    wantf = np.zeros(3)
    o.lib.oracle_op_l2p(o.ptr(wantf), o.ptr(Lp), P, o.ptr(d))
    lcf, lidx = t.get("l2p_coef", np.float32), t.get("l2p_idx", np.uint32)
    f = np.zeros(3)
    for k in range(t.offM if t.offM > 0 else 0):
        c = float(lcf[k]) * D[k]
        f[0] -= c * F[lidx[k] & 0x3FF]
        f[1] -= c * F[(lidx[k] >> 10) & 0x3FF]
        f[2] -= c * F[(lidx[k] >> 20) & 0x3FF]
    np.testing.assert_allclose(f, wantf, rtol=3e-6, atol=2e-9 * np.abs(wantf).max())
