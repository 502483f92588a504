#!/usr/bin/env python3
"""gen_ops.py -- emits fmm_ops_gen.inc: straight-line, register-resident P2M / M2M / L2L / L2P bodies for
orders 1..PMAX (companion of gen_m2l.py).

Reference operators: fmm_cart_base3.cuh P2M :908-918, M2M :1042-1076, L2L :1348-1363, L2P :1511-1529
(csrc/genops_host.cpp compiles the generated text for the host; tests/test_genops_host.py checks it against the oracle).  With the
normalisations

    D~[K] = d^K / K!            M~[X] = M[X] |X|! / X!            L~[X] = L[X] |X|!      (X! = x! y! z!)

every operator is a coefficient-free correlation / convolution, i.e. plain v_fma_f32 chains:

    P2M   M~_q[X]  = (-1)^q sum_particles D~[X]
    M2M   M~'[X]   = sum_{K <= X} D~[K] M~[X-K]                  (d = new centre - old centre)
    L2L   L~'_n[X] = sum_{m >= n} sum_{|K| = m-n} D~[K] L~_m[X+K] (d = child - parent)
    L2P   a_c      = - sum_{K} D~[K] L~_{|K|+1}[e_c + K]          (d = particle - leaf centre)

Arrays in HBM keep the reference's normalisation; the scale factors are literals at load / store.
Usage: gen_ops.py <out.inc> [PMAX]
"""
import sys
from math import factorial as fact


def sym_off(p):
    return p * (p + 1) * (p + 2) // 6


def sym_idx(x, z, n):
    return (n * (n + 1) - (n - z) * (n - z + 1)) // 2 + n - x


def tl_off(p):
    return p * p


def tl_idx(x, z, n):
    return (z + 1) * n - x


def comps(n):
    """(x, y, z) of every component of a rank-n symmetric tensor in storage order"""
    return [(x, n - x - z, z) for z in range(n + 1) for x in range(n - z, -1, -1)]


def lit(v):
    v = float(v)
    return "T(%d)" % int(v) if v == int(v) and abs(v) < 2 ** 24 else "T(%s)" % repr(v)


def full(x, y, z):
    n = x + y + z
    return sym_off(n) + sym_idx(x, z, n)


def emit_monomials(w, pmax_order):
    """D{i} = d^K / K! for all |K| <= pmax_order (needs dx, dy, dz)"""
    for a in "xyz":
        for k in range(2, pmax_order + 1):
            w("\tconst T d%s_%d = d%s * %s;" % (a, k, a, lit(1.0 / k)))
    w("\tconst T D0 = T(1);")
    for n in range(1, pmax_order + 1):
        for (x, y, z) in comps(n):
            i = full(x, y, z)
            if x > 0:
                par, a, k = full(x - 1, y, z), "x", x
            elif y > 0:
                par, a, k = full(x, y - 1, z), "y", y
            else:
                par, a, k = full(x, y, z - 1), "z", z
            fac = "d%s" % a if k == 1 else "d%s_%d" % (a, k)
            w("\tconst T D%d = %s;" % (i, fac) if par == 0 else "\tconst T D%d = D%d * %s;" % (i, par, fac))


def emit_expand_local(w, P, src):
    """F{i} = L~ in the full layout from the traceless tuple `src` (orders 1..P)"""
    for n in range(1, P + 1):
        g = {}
        for z in range(0, min(1, n) + 1):
            for x in range(n - z, -1, -1):
                y = n - x - z
                w("\tconst T F%d = %s[%d] * %s;" % (full(x, y, z), src, tl_off(n) + tl_idx(x, z, n), lit(fact(n))))
        for z in range(2, n + 1):
            for x in range(n - z, -1, -1):
                y = n - x - z
                w("\tconst T F%d = -(F%d + F%d);" % (full(x, y, z), full(x + 2, y, z - 2), full(x, y + 2, z - 2)))


def odfact(n):
    """n!! for odd n (and 1 for n <= 0)"""
    r = 1
    while n > 1:
        r *= n
        n -= 2
    return r


def coeff13(n, m):
    return (-1) ** m * odfact(2 * (n - m) - 1)


def coeff2(n, m):
    return fact(n) // (2 ** m * fact(m) * fact(n - 2 * m))


def harmonic_terms(n, x, z):
    """TL(d^n)[x, y, z] * (2n-1)!! for z in {0, 1} as a list of (coefficient, (ex, ey, ez), k): coefficient * d^e * (r^2)^k
    (fmm_cart_base3.cuh:711-727 / 830-846 / 931-947 written for the un-normalised vector d)"""
    y = n - x - z
    terms = []
    for k1 in range(x // 2 + 1):
        for k2 in range(y // 2 + 1):
            c = coeff13(n, k1 + k2) * coeff2(x, k1) * coeff2(y, k2)
            terms.append((c, (x - 2 * k1, y - 2 * k2, z), k1 + k2))
    return terms


def emit_r2_powers(w, kmax):
    w("\tconst T R2_0 = T(1);")
    w("\tconst T R2_1 = nb_fma(dx, dx, nb_fma(dy, dy, dz * dz));")
    for k in range(2, kmax + 1):
        w("\tconst T R2_%d = R2_%d * R2_1;" % (k, k - 1))


def harmonic_expr(n, x, z, scale):
    """expression for scale * TL-harmonic component (x, z) of order n in terms of D{i} (= d^K / K!) and R2_k"""
    by_k = {}
    for (c, e, k) in harmonic_terms(n, x, z):
        by_k.setdefault(k, []).append((c * fact(e[0]) * fact(e[1]) * fact(e[2]), e))
    expr = None
    for k in sorted(by_k):
        inner = None
        for (c, e) in by_k[k]:
            t = "%s * D%d" % (lit(c * scale), full(*e))
            inner = t if inner is None else "nb_fma(%s, D%d, %s)" % (lit(c * scale), full(*e), inner)
        t = inner if k == 0 else "(%s) * R2_%d" % (inner, k)
        expr = t if expr is None else ("nb_fma(%s, R2_%d, %s)" % (inner, k, expr) if False else "(%s + %s)" % (expr, t))
    return expr


def gen_oct(P, out):
    """traceless-multipole operators of the uniform-octree evaluator (fmm_cart3_traceless.cuh):
    P2M p2m_traceless_acc3 (fmm_cart_base3.cuh:920-949), M2M m2m_traceless_acc3 (:1078-1109)"""
    w = out.append
    offL = tl_off(P + 1)
    # ---------------------------------------------------------------- P2M (traceless), orders 2..P
    w("template <typename T> struct FmmOctOps<%d, T>" % P)
    w("{")
    w("static __device__ __forceinline__ void p2m_tl_accum(T dx, T dy, T dz, T (&A)[%d])" % offL)
    w("{")
    if P >= 2:
        emit_monomials(w, P)
        emit_r2_powers(w, P // 2)
        for q in range(2, P + 1):
            C = (-1) ** q / fact(q) / odfact(2 * q - 1)
            for z in range(0, 2):
                for x in range(q - z, -1, -1):
                    w("\tA[%d] += %s;" % (tl_off(q) + tl_idx(x, z, q), harmonic_expr(q, x, z, C)))
    else:
        w("\t(void)dx; (void)dy; (void)dz; (void)A;")
    w("}")
    # ---------------------------------------------------------------- M2M (traceless), orders 2..P, one child
    w("static __device__ __forceinline__ void m2m_tl_accum(const T *__restrict__ Mc, T dx, T dy, T dz, T (&A)[%d])" % offL)
    w("{")
    if P >= 2:
        emit_monomials(w, P)
        emit_r2_powers(w, P // 2)
        # TP{m}_{i}: full symmetric layout of tracelesspow_m(d) / m!
        w("\tconst T TP%d = T(1);" % full(0, 0, 0))
        for m in range(1, P + 1):
            C = 1.0 / odfact(2 * m - 1) / fact(m)
            for z in range(0, min(1, m) + 1):
                for x in range(m - z, -1, -1):
                    w("\tconst T TP%d = %s;" % (full(x, m - x - z, z), harmonic_expr(m, x, z, C)))
            for z in range(2, m + 1):
                for x in range(m - z, -1, -1):
                    y = m - x - z
                    w("\tconst T TP%d = -(TP%d + TP%d);" % (full(x, y, z), full(x + 2, y, z - 2), full(x, y + 2, z - 2)))
        # F{i}: full layout of the child's multipoles (orders 0, 2..P-... all used orders), from the traceless tuple
        for k in range(0, P + 1):
            if k == 1:
                continue
            for z in range(0, min(1, k) + 1):
                for x in range(k - z, -1, -1):
                    w("\tconst T F%d = Mc[%d];" % (full(x, k - x - z, z), tl_off(k) + tl_idx(x, z, k)))
            for z in range(2, k + 1):
                for x in range(k - z, -1, -1):
                    y = k - x - z
                    w("\tconst T F%d = -(F%d + F%d);" % (full(x, y, z), full(x + 2, y, z - 2), full(x, y + 2, z - 2)))
        for n in range(2, P + 1):
            for z in range(0, 2):
                for x in range(n - z, -1, -1):
                    y = n - x - z
                    expr = "A[%d]" % (tl_off(n) + tl_idx(x, z, n))
                    for m in range(0, n + 1):
                        if n - m == 1:
                            continue   # the child's dipole is identically zero (expansion about the centre of charge)
                        for k1 in range(0, min(x, m) + 1):
                            for k3 in range(max(0, m - k1 - y), min(z, m - k1) + 1):
                                k2 = m - k1 - k3
                                expr = "nb_fma(TP%d, F%d, %s)" % (full(k1, k2, k3), full(x - k1, y - k2, z - k3), expr)
                    w("\tA[%d] = %s;" % (tl_off(n) + tl_idx(x, z, n), expr))
    else:
        w("\t(void)Mc; (void)dx; (void)dy; (void)dz; (void)A;")
    w("}")
    w("};")
    w("")


def gen(P, out):
    w = out.append
    offM = sym_off(P)
    offL = tl_off(P + 1)
    # ---------------------------------------------------------------- P2M
    w("template <typename T> struct FmmOps<%d, T>" % P)
    w("{")
    w("static __device__ __forceinline__ void p2m_accum(T dx, T dy, T dz, T (&A)[%d])" % max(offM, 1))
    w("{")
    if P >= 3:
        emit_monomials(w, P - 1)
        for q in range(2, P):
            for (x, y, z) in comps(q):
                i = full(x, y, z)
                w("\tA[%d] += D%d;" % (i, i))
    else:
        w("\t(void)dx; (void)dy; (void)dz; (void)A;")
    w("}")
    w("static __device__ __forceinline__ void p2m_store(const T (&A)[%d], T *__restrict__ M)" % max(offM, 1))
    w("{")
    for q in range(2, P):
        for (x, y, z) in comps(q):
            i = full(x, y, z)
            w("\tM[%d] = A[%d] * %s;" % (i, i, lit((-1) ** q * fact(x) * fact(y) * fact(z) / fact(q))))
    if P < 3:
        w("\t(void)A; (void)M;")
    w("}")
    # ---------------------------------------------------------------- M2M
    w("static __device__ __forceinline__ void m2m_accum(const T *__restrict__ Mc, T dx, T dy, T dz, T (&A)[%d])" % max(offM, 1))
    w("{")
    if P >= 3:
        emit_monomials(w, P - 1)
        for k in range(0, P):
            if k == 1:
                continue
            for (x, y, z) in comps(k):
                i = full(x, y, z)
                c = fact(k) / (fact(x) * fact(y) * fact(z))
                w("\tconst T T%d = Mc[%d]%s;" % (i, i, "" if c == 1 else " * " + lit(c)))
        for n in range(2, P):
            for (x, y, z) in comps(n):
                o = full(x, y, z)
                expr = "A[%d]" % o
                for k1 in range(x + 1):
                    for k2 in range(y + 1):
                        for k3 in range(z + 1):
                            m = k1 + k2 + k3
                            if n - m == 1:
                                continue
                            expr = "nb_fma(D%d, T%d, %s)" % (full(k1, k2, k3), full(x - k1, y - k2, z - k3), expr)
                w("\tA[%d] = %s;" % (o, expr))
    else:
        w("\t(void)Mc; (void)dx; (void)dy; (void)dz; (void)A;")
    w("}")
    w("static __device__ __forceinline__ void m2m_store(const T (&A)[%d], T *__restrict__ M)" % max(offM, 1))
    w("{")
    for n in range(2, P):
        for (x, y, z) in comps(n):
            o = full(x, y, z)
            w("\tM[%d] = A[%d] * %s;" % (o, o, lit(fact(x) * fact(y) * fact(z) / fact(n))))
    if P < 3:
        w("\t(void)A; (void)M;")
    w("}")
    # ---------------------------------------------------------------- L2L
    w("static __device__ __forceinline__ void l2l_body(const T (&Lp)[%d], T dx, T dy, T dz, T (&O)[%d])" % (offL, offL))
    w("{")
    emit_monomials(w, P - 1)
    emit_expand_local(w, P, "Lp")
    w("\tO[0] = T(0);")
    for n in range(1, P + 1):
        for z in range(0, min(1, n) + 1):
            for x in range(n - z, -1, -1):
                y = n - x - z
                expr = None
                for m in range(n, P + 1):
                    k = m - n
                    for (kx, ky, kz) in comps(k):
                        t = "D%d * F%d" % (full(kx, ky, kz), full(x + kx, y + ky, z + kz)) if full(kx, ky, kz) != 0 else "F%d" % full(x, y, z)
                        if expr is None:
                            expr = t
                        else:
                            expr = "nb_fma(D%d, F%d, %s)" % (full(kx, ky, kz), full(x + kx, y + ky, z + kz), expr)
                w("\tO[%d] = (%s) * %s;" % (tl_off(n) + tl_idx(x, z, n), expr, lit(1.0 / fact(n))))
    w("}")
    # ---------------------------------------------------------------- L2P
    w("static __device__ __forceinline__ void l2p_body(const T (&Lp)[%d], T dx, T dy, T dz, T &fx, T &fy, T &fz)" % offL)
    w("{")
    emit_monomials(w, P - 1)
    emit_expand_local(w, P, "Lp")
    ex = ey = ez = None
    for q in range(0, P):
        for (kx, ky, kz) in comps(q):
            d = full(kx, ky, kz)
            ix, iy, iz = full(kx + 1, ky, kz), full(kx, ky + 1, kz), full(kx, ky, kz + 1)
            if d == 0:
                ex, ey, ez = "F%d" % ix, "F%d" % iy, "F%d" % iz
            else:
                ex = "nb_fma(D%d, F%d, %s)" % (d, ix, ex)
                ey = "nb_fma(D%d, F%d, %s)" % (d, iy, ey)
                ez = "nb_fma(D%d, F%d, %s)" % (d, iz, ez)
        # flush per order to keep expressions short
        w("\tconst T ex%d = %s;" % (q, ex))
        w("\tconst T ey%d = %s;" % (q, ey))
        w("\tconst T ez%d = %s;" % (q, ez))
        ex, ey, ez = "ex%d" % q, "ey%d" % q, "ez%d" % q
    w("\tfx = -%s; fy = -%s; fz = -%s;" % (ex, ey, ez))
    w("}")
    w("};")
    w("")


def main():
    path = sys.argv[1]
    pmax = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    out = ["// GENERATED by gen_ops.py -- do not edit.  Straight-line P2M / M2M / L2L / L2P bodies, orders 1..%d." % pmax, ""]
    for P in range(1, pmax + 1):
        gen(P, out)
        gen_oct(P, out)
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
