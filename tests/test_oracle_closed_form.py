"""CPU tests: the oracle's operators against an INDEPENDENT first-principles implementation.

Nothing here follows the oracle's (or the reference's) loops.  The yardstick is built from Taylor's theorem alone:

  * partial derivatives of f(x) = 1/|x| by the polynomial recurrence  d/dx_a [P / r^(2n+1)] = (r^2 dP/dx_a - (2n+1) x_a P) / r^(2n+3)
    with exact integer polynomial coefficients;
  * the potential of unit charges at c_s + delta_j seen from r:  Phi(r) = sum_K  M[K] |K|!/K!  d^K f (r - c_s),
    M[K] = (-1)^|K| / |K|!  sum_j delta_j^K                      (the multipole normalisation of fmm_cart_base3.cuh:908-918);
  * a local expansion about c stores  L_n[X] = d^X Phi(c) / n!   (|X| = n), so  Phi(c + rho) = sum_X  n!/X!  L_n[X] rho^X,
    a = -grad Phi, and re-expanding about c' = c + d gives  L'_n[X] = 1/n! sum_Y (n+|Y|)!/Y!  L_{n+|Y|}[X+Y] d^Y.

Tensor storage as in SURVEY 8 (fmm_cart_base3.cuh:180-241): symmetric rank-n component (x, y, z) at
[n(n+1) - (n-z)(n-z+1)]/2 + n - x, tuples at n(n+1)(n+2)/6; traceless tensors keep z in {0, 1} at (z+1) n - x, tuples at n^2.

This does not pin the oracle to the reference (only reference-held vectors could: there are none); it removes
"the oracle and the kernels share a misreading of the operator algebra" as a failure mode, for p = 1..10.
"""
import ctypes as C
import itertools
import math

import numpy as np
import pytest


# ---- storage conventions ---------------------------------------------------------------------------------------------------
def sym_idx(x, z, n):
    return (n * (n + 1) - (n - z) * (n - z + 1)) // 2 + n - x


def sym_off(n):
    return n * (n + 1) * (n + 2) // 6


def tl_idx(x, z, n):
    return (z + 1) * n - x


def tl_off(n):
    return n * n


def multi_indices(n):
    return [(x, n - x - z, z) for z in range(n + 1) for x in range(n - z, -1, -1)]


def fact_multi(K):
    return math.factorial(K[0]) * math.factorial(K[1]) * math.factorial(K[2])


# ---- derivatives of 1/r by polynomial recurrence (exact integer coefficients) ------------------------------------------------
class InverseDistanceDerivatives:
    def __init__(self, maxorder):
        self.P = {(0, 0, 0): {(0, 0, 0): 1}}
        for n in range(maxorder):
            for a in multi_indices(n):
                for ax in range(3):
                    b = tuple(a[i] + (1 if i == ax else 0) for i in range(3))
                    if b not in self.P:
                        self.P[b] = self._step(self.P[a], n, ax)

    @staticmethod
    def _step(P, n, ax):
        out = {}

        def add(e, c):
            if c:
                out[e] = out.get(e, 0) + c
        for e, c in P.items():
            if e[ax] > 0:                                   # r^2 * dP/dx_a
                de = tuple(e[i] - (1 if i == ax else 0) for i in range(3))
                for q in range(3):
                    add(tuple(de[i] + (2 if i == q else 0) for i in range(3)), c * e[ax])
            add(tuple(e[i] + (1 if i == ax else 0) for i in range(3)), -(2 * n + 1) * c)   # -(2n+1) x_a P
        return {e: c for e, c in out.items() if c}

    def __call__(self, alpha, d):
        n = sum(alpha)
        r2 = float(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
        s = 0.0
        for e, c in self.P[tuple(alpha)].items():
            s += c * d[0] ** e[0] * d[1] ** e[1] * d[2] ** e[2]
        return s / r2 ** (n + 0.5)


PMAX = 10
DERIV = InverseDistanceDerivatives(PMAX + 1)


@pytest.fixture(scope="module")
def o64(oracle64):
    L = oracle64.lib
    P, r = C.c_void_p, C.c_double
    L.oracle_op_gradient.argtypes = [P, C.c_int, P, r, r]
    L.oracle_op_p2m.argtypes = [P, C.c_int, P, C.c_int, P]
    L.oracle_op_m2m.argtypes = [P, P, C.c_int, P]
    L.oracle_op_m2l.argtypes = [P, P, C.c_int, P, r]
    L.oracle_op_l2l.argtypes = [P, P, C.c_int, P]
    L.oracle_op_l2p.argtypes = [P, P, C.c_int, P]
    return oracle64


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def multipoles_from_definition(pts, c, p):
    """symmetric tuple, orders 0..p-1"""
    M = np.zeros(sym_off(p))
    for q in range(p):
        for (x, y, z) in multi_indices(q):
            d = pts - c
            M[sym_off(q) + sym_idx(x, z, q)] = (-1) ** q / math.factorial(q) * np.sum(d[:, 0] ** x * d[:, 1] ** y * d[:, 2] ** z)
    return M


def locals_from_sources(src, c, p):
    """exact Taylor coefficients L_n[X] = d^X Phi(c) / n! of Phi(r) = sum_j 1/|r - s_j|, as a dict over all multi-indices"""
    Lf = {}
    for n in range(p + 1):
        for X in multi_indices(n):
            Lf[X] = sum(DERIV(X, c - s) for s in src) / math.factorial(n)
    return Lf


def store_traceless(Lf, p):
    out = np.zeros(tl_off(p + 1))
    for n in range(p + 1):
        for z in range(min(1, n) + 1):
            for x in range(n - z, -1, -1):
                out[tl_off(n) + tl_idx(x, z, n)] = Lf[(x, n - x - z, z)]
    return out


def rel(a, b):
    scale = np.abs(b).max() + 1e-300
    return float(np.abs(a - b).max() / scale)


# ---- 1. gradient of 1/r: every component, including those rebuilt from the trace condition -----------------------------------
@pytest.mark.parametrize("n", range(0, PMAX + 1))
def test_gradient_components(o64, n):
    rng = np.random.default_rng(100 + n)
    for _ in range(3):
        d = rng.standard_normal(3)
        r = float(np.linalg.norm(d))
        g = np.zeros((n + 1) * (n + 2) // 2)
        u = np.ascontiguousarray(d / r)
        o64.lib.oracle_op_gradient(ptr(g), n, ptr(u), r, 1.0)
        want = np.array([DERIV(a, d) for a in multi_indices(n)])
        for a in multi_indices(n):
            assert sym_idx(a[0], a[2], n) == multi_indices(n).index(a)      # the layout formula itself
        assert rel(g, want) < 1e-11


# ---- 2. P2M is the monomial moment; 3. M2M is exact: moments about the new centre ---------------------------------------------
@pytest.mark.parametrize("p", range(1, PMAX + 1))
def test_p2m_and_m2m_against_moment_definition(o64, p):
    rng = np.random.default_rng(200 + p)
    pts = rng.standard_normal((9, 3)) * 0.3
    c_old, c_new = rng.standard_normal(3) * 0.1, rng.standard_normal(3) * 0.1
    want_old, want_new = multipoles_from_definition(pts, c_old, p), multipoles_from_definition(pts, c_new, p)
    M = np.zeros(sym_off(p))
    o64.lib.oracle_op_p2m(ptr(M), p, ptr(np.ascontiguousarray(pts)), len(pts), ptr(c_old))
    if p > 2:
        assert rel(M[sym_off(2):], want_old[sym_off(2):]) < 1e-12            # the operator fills orders 2..p-1
    # shift: orders 0 and 1 of the input come from the definition (the drivers keep them apart), 2..p-1 from the operator
    Min = want_old.copy()
    Mout = np.zeros(sym_off(p))
    d = np.ascontiguousarray(c_new - c_old)
    o64.lib.oracle_op_m2m(ptr(Mout), ptr(Min), p, ptr(d))
    if p > 2:
        assert rel(Mout[sym_off(2):], want_new[sym_off(2):]) < 1e-11


# ---- 4. M2L: Taylor coefficients of the multipole-expanded potential at the target centre -------------------------------------
@pytest.mark.parametrize("p", range(1, PMAX + 1))
def test_m2l_against_taylor_coefficients(o64, p):
    rng = np.random.default_rng(300 + p)
    pts = rng.standard_normal((7, 3)) * 0.2
    c_s = pts.mean(axis=0)                                  # centre of charge: the dipole vanishes, as in the kd-tree flavour
    c_t = c_s + np.array([1.3, -0.7, 0.9]) + rng.standard_normal(3) * 0.1
    M = multipoles_from_definition(pts, c_s, p)
    if p > 1:
        assert np.abs(M[1:4]).max() < 1e-15
        M[1:4] = 0.0
    d = c_t - c_s
    L = np.zeros(tl_off(p + 1))
    o64.lib.oracle_op_m2l(ptr(L), ptr(M), p, ptr(np.ascontiguousarray(d)), 0.0)
    want = np.zeros_like(L)
    for n in range(1, p + 1):                               # order-0 local is never formed (minm = 1)
        for z in range(2):
            for x in range(n - z, -1, -1):
                X = (x, n - x - z, z)
                s = 0.0
                for k in range(0, p):                       # multipole orders 0..p-1, total order m = n + k <= p, no dipole
                    if n + k > p or k == 1:
                        continue
                    for K in multi_indices(k):
                        s += M[sym_off(k) + sym_idx(K[0], K[2], k)] * math.factorial(k) / fact_multi(K) * DERIV(tuple(X[i] + K[i] for i in range(3)), d)
                want[tl_off(n) + tl_idx(x, z, n)] = s / math.factorial(n)
    assert rel(L[1:], want[1:]) < 1e-10
    # and the expansion means what it should: with every order present it converges to the true field of the sources
    if p >= 8:
        true_L = store_traceless(locals_from_sources(pts, c_t, p), p)
        assert rel(L[1:4], true_L[1:4]) < 5e-3             # order-1 local = -field/1: truncation error ~ (0.3/1.7)^p


# ---- 5. L2L: re-expansion about the child's centre; 6. L2P: minus the gradient of the local polynomial -------------------------
@pytest.mark.parametrize("p", range(1, PMAX + 1))
def test_l2l_and_l2p_against_taylor_reexpansion(o64, p):
    rng = np.random.default_rng(400 + p)
    src = rng.standard_normal((5, 3)) * 0.3 + np.array([2.0, 1.0, -1.5])
    c = rng.standard_normal(3) * 0.1
    Lf = locals_from_sources(src, c, p)
    Lp = store_traceless(Lf, p)
    d = rng.standard_normal(3) * 0.2
    Lc = np.zeros_like(Lp)
    o64.lib.oracle_op_l2l(ptr(Lc), ptr(Lp), p, ptr(np.ascontiguousarray(d)))
    want = np.zeros_like(Lp)
    for n in range(1, p + 1):
        for z in range(2):
            for x in range(n - z, -1, -1):
                X = (x, n - x - z, z)
                s = 0.0
                for k in range(0, p - n + 1):
                    for Y in multi_indices(k):
                        s += math.factorial(n + k) / fact_multi(Y) * Lf[tuple(X[i] + Y[i] for i in range(3))] * d[0] ** Y[0] * d[1] ** Y[1] * d[2] ** Y[2]
                want[tl_off(n) + tl_idx(x, z, n)] = s / math.factorial(n)
    assert rel(Lc[1:], want[1:]) < 1e-10
    # L2P at rho: a = -grad sum_X n!/X! L_n[X] rho^X  (orders 1..p)
    rho = rng.standard_normal(3) * 0.15
    f = np.zeros(3)
    o64.lib.oracle_op_l2p(ptr(f), ptr(Lp), p, ptr(np.ascontiguousarray(rho)))
    g = np.zeros(3)
    for n in range(1, p + 1):
        for X in multi_indices(n):
            coef = math.factorial(n) / fact_multi(X) * Lf[X]
            for a in range(3):
                if X[a] == 0:
                    continue
                e = [X[i] - (1 if i == a else 0) for i in range(3)]
                g[a] += coef * X[a] * rho[0] ** e[0] * rho[1] ** e[1] * rho[2] ** e[2]
    assert rel(f, -g) < 1e-11
    # physical meaning: the truncated expansion approaches the true acceleration sum_j (x - s_j)/|x - s_j|^3 as p grows
    xq = c + rho
    true = sum((xq - s) / np.linalg.norm(xq - s) ** 3 for s in src)
    assert rel(f, true) < 3.0 * (np.linalg.norm(rho) / 1.9) ** p + 1e-12


# ---- 7. opening criterion and particle ranges on small trees: an independent traversal over the oracle's own node arrays -------
def independent_lists(t, p, radius):
    """dual traversal written from the description of fmm_cart3_kdtree.cuh:401-414,569-611 (leaf-leaf first -> P2P; self pair ->
    three child pairs; admissible -> M2L; else split the leaf's partner, or the node whose box diagonal is not the smaller)"""
    ntot, mult, cen, lb, rb = t["ntot"], t["mult"], t["center"], t["lbound"], t["rbound"]
    size = ((rb - lb) ** 2).sum(axis=1)
    leaf = lambda i: 2 * i + 1 >= ntot
    p2p, m2l = [], []
    todo = [(0, 0)]
    while todo:
        a, b = todo.pop()
        if leaf(a) and leaf(b):
            if a != b:
                p2p.append((a, b))
        elif a == b:
            todo += [(2 * a + 1, 2 * a + 1), (2 * a + 1, 2 * a + 2), (2 * a + 2, 2 * a + 2)]
        else:
            M = (max(mult[a], mult[b]) / mult[0]) ** (1.0 / (3 * p + 6))
            dist2 = ((cen[b] - cen[a]) ** 2).sum()
            if (radius * M) ** 2 * max(size[a], size[b]) < dist2:
                m2l.append((a, b))
            elif leaf(a) or (not leaf(b) and size[a] <= size[b]):
                todo += [(a, 2 * b + 1), (a, 2 * b + 2)]
            else:
                todo += [(2 * a + 1, b), (2 * a + 2, b)]
    return p2p, m2l


@pytest.mark.parametrize("n,p,radius", [(64, 2, 1.0), (200, 3, 1.0), (777, 3, 2.0), (1500, 4, 1.0), (4096, 6, 1.0), (5000, 5, 3.0)])
def test_admissibility_and_ranges_on_small_trees(oracle64, n, p, radius):
    o = oracle64
    rng = np.random.default_rng(n)
    pos = rng.standard_normal((n, 3)) * np.array([1.0, 0.4, 2.5])
    pv = np.stack([pos, np.zeros_like(pos)])
    par = np.array([1.0 / n, 0, 0, 1, 1, 1])
    o.fmm_kd(pv, par, p=p, radius=radius, threads=1, unsort=True)
    t = o.kd_tree()
    L = t["L"]
    # particle ranges ceil(n i / 2^l) and their differences
    for l in range(L + 1):
        m = 1 << l
        want = np.array([-((-n * i) // m) for i in range(m)])
        np.testing.assert_array_equal(t["index"][m - 1:2 * m - 1], want)
        np.testing.assert_array_equal(t["mult"][m - 1:2 * m - 1], np.diff(np.append(want, n)))
    p2p, m2l = independent_lists(t, p, radius)
    canon = lambda prs: sorted((min(a, b), max(a, b)) for a, b in prs)
    assert canon(p2p) == canon(map(tuple, t["p2p"]))
    assert canon(m2l) == canon(map(tuple, t["m2l"]))
    # every particle pair is covered exactly once: by a leaf pair, by an admissible node pair, or inside one leaf
    leaves = t["mult"][(1 << L) - 1:].astype(np.int64)
    covered = (leaves * (leaves - 1) // 2).sum() + sum(int(t["mult"][a]) * int(t["mult"][b]) for a, b in p2p + m2l)
    assert covered == n * (n - 1) // 2
