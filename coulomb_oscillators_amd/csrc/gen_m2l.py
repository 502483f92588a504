#!/usr/bin/env python3
"""gen_m2l.py -- emits m2l_gen.inc: straight-line, register-resident M2L bodies for orders 1..PMAX.

The reference's M2L recomputes index arithmetic and trinomials per term inside run-time loops
(fmm_cart_base3.cuh:1181-1208, :378-426, mymath.cuh:215-231).  For gfx950 every order gets a fully
flattened body -- scalar variables only, literal coefficients -- so that the 2m+1 gradient components,
the traceless refinement, and the sum_k M_k . G_(n+k) contractions become plain v_fma_f32 chains with
no LDS, no tables and no indexing:

    L_n[x,y,z] += (r^-(n+1) / n!) * sum_{m=n..P, k=m-n, k != 1, k <= P-1} sum_{|kappa|=k} k!/(kx!ky!kz!)
                  * (M_k[kappa] r^-k) * G^_m[(x,y,z)+kappa],      z in {0,1}
    G^_m[x,y,z]  = (-1)^m uz^z sum_{k1<=x/2,k2<=y/2} (-1)^(k1+k2) (2m-2(k1+k2)-1)!! c2(x,k1) c2(y,k2)
                   ux^(x-2k1) uy^(y-2k2),  z in {0,1};  G^[x,y,z>=2] = -G^[x+2,y,z-2] - G^[x,y+2,z-2]

(csrc/genops_host.cpp compiles the generated text for the host; tests/test_genops_host.py checks it against the oracle).
Usage: gen_m2l.py <out.inc> [PMAX]
"""
import sys
from math import factorial as fact


def odfact(n):
    r = 1
    while n > 1:
        r *= n
        n -= 2
    return r


def coeff2(a, k):
    return fact(a) / (2 ** k * fact(k) * fact(a - 2 * k))


def sym_off(p):
    return p * (p + 1) * (p + 2) // 6


def sym_idx(x, z, n):
    return (n * (n + 1) - (n - z) * (n - z + 1)) // 2 + n - x


def tl_off(p):
    return p * p


def tl_idx(x, z, n):
    return (z + 1) * n - x


def comp_xyz(i, n):
    """inverse of sym_idx for order n"""
    for z in range(n + 1):
        for x in range(n - z, -1, -1):
            if sym_idx(x, z, n) == i:
                return x, n - x - z, z
    raise ValueError


def lit(v):
    v = float(v)
    return "T(%d)" % int(v) if v == int(v) and abs(v) < 2 ** 24 else "T(%s)" % repr(v)


def gen_body(P, out):
    offM = sym_off(P)
    nout = tl_off(P + 1) - 1
    w = out.append
    w("template <typename T> struct M2LBody<%d, T>" % P)
    w("{")
    w("static __device__ __forceinline__ void run(const T *__restrict__ Mp, T ux, T uy, T uz, T rinv, T (&L)[%d])" % max(nout, 1))
    w("{")
    # powers of the unit vector and of 1/r
    for a in "xy":
        for e in range(2, P + 1):
            w("\tconst T u%s%d = u%s%d * u%s;" % (a, e, a, e - 1, a) if e > 2 else "\tconst T u%s2 = u%s * u%s;" % (a, a, a))
    w("\tconst T r1 = rinv;")
    for e in range(2, P + 2):
        w("\tconst T r%d = r%d * rinv;" % (e, e - 1))

    def upow(a, e):
        if e == 0:
            return None
        return "u%s" % a if e == 1 else "u%s%d" % (a, e)

    # scaled source multipoles (orders 0, 2..P-1; the dipole about the centre of charge is zero)
    for k in range(0, P):
        if k == 1:
            continue
        for i in range((k + 1) * (k + 2) // 2):
            idx = sym_off(k) + i
            # pre-multiplied by r^-k and by the trinomial weight k!/(kx!ky!kz!) of its component
            x, y, z = comp_xyz(i, k)
            tri = fact(k) // (fact(x) * fact(y) * fact(z))
            if k == 0:
                w("\tconst T M%d = Mp[%d];" % (idx, idx))
            elif tri == 1:
                w("\tconst T M%d = Mp[%d] * r%d;" % (idx, idx, k))
            else:
                w("\tconst T M%d = Mp[%d] * (r%d * %s);" % (idx, idx, k, lit(tri)))
    accs = {}
    declared = set()
    for n in range(1, P + 1):
        for i in range(2 * n + 1):
            accs[(n, i)] = []
    for m in range(1, P + 1):
        w("\t// ---- order-%d gradient tensor" % m)
        g = {}
        # independent components
        for z in range(0, min(1, m) + 1):
            for x in range(m - z, -1, -1):
                y = m - x - z
                terms = []
                for k1 in range(x // 2 + 1):
                    for k2 in range(y // 2 + 1):
                        j = k1 + k2
                        c = (-1) ** m * (-1) ** j * odfact(2 * (m - j) - 1) * coeff2(x, k1) * coeff2(y, k2)
                        f = [s for s in (upow("x", x - 2 * k1), upow("y", y - 2 * k2)) if s]
                        terms.append((c, f))
                name = "g%d_%d" % (m, sym_idx(x, z, m))
                expr = None
                for c, f in terms:
                    mono = " * ".join(f) if f else None
                    if expr is None:
                        expr = "%s * %s" % (lit(c), mono) if mono else lit(c)
                    else:
                        expr = "nb_fma(%s, %s, %s)" % (lit(c), mono, expr) if mono else "(%s + %s)" % (expr, lit(c))
                if z == 1:
                    expr = "(%s) * uz" % expr
                w("\tconst T %s = %s;" % (name, expr))
                g[(x, z)] = name
        for z in range(2, m + 1):
            for x in range(m - z, -1, -1):
                name = "g%d_%d" % (m, sym_idx(x, z, m))
                w("\tconst T %s = -(%s + %s);" % (name, g[(x + 2, z - 2)], g[(x, z - 2)]))
                g[(x, z)] = name
        # contractions that use G_m: outputs n = m-k
        for n in range(1, m + 1):
            k = m - n
            if k == 1 or k > P - 1:
                continue
            for z in range(0, min(1, n) + 1):
                for x in range(n - z, -1, -1):
                    o = tl_idx(x, z, n)
                    for kz in range(k + 1):
                        for kx in range(k - kz + 1):
                            midx = sym_off(k) + sym_idx(kx, kz, k)
                            accs[(n, o)].append(("M%d" % midx, g[(x + kx, z + kz)]))
            # emit the partial sums of this (n, m) block right away to keep live ranges short
            for z in range(0, min(1, n) + 1):
                for x in range(n - z, -1, -1):
                    o = tl_idx(x, z, n)
                    terms = accs[(n, o)]
                    var = "a%d_%d" % (n, o)
                    first = var not in declared
                    declared.add(var)
                    expr = var if not first else None
                    for ms, gs in terms:
                        expr = "%s * %s" % (ms, gs) if expr is None else "nb_fma(%s, %s, %s)" % (ms, gs, expr)
                    if first:
                        w("\tT %s = %s;" % (var, expr))
                    else:
                        w("\t%s = %s;" % (var, expr))
                    accs[(n, o)] = []
    for n in range(1, P + 1):
        for i in range(2 * n + 1):
            w("\tL[%d] = a%d_%d * (r%d * %s);" % (tl_off(n) + i - 1, n, i, n + 1, lit(1.0 / fact(n))))
    w("}")
    w("};")
    w("")


def main():
    path = sys.argv[1]
    pmax = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    out = ["// GENERATED by gen_m2l.py -- do not edit.  Straight-line M2L bodies, orders 1..%d." % pmax, ""]
    for P in range(1, pmax + 1):
        gen_body(P, out)
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
