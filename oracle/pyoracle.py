"""ctypes binding of the CPU oracle (oracle/nbco_oracle.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (coulomb_oscillators_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

REF_SEED = 5351550349027530206      # main3.cu:662
REF_DISCARD = 1248                  # main3.cu:663
SIGMA_X = (0.003, 0.001, 0.01)      # main3.cu:244
OMEGA0 = (1.095, 1.0, 1.0)          # main3.cu:241
XI = 2e-6                           # main3.cu:240

KIND_DIRECT3, KIND_FMM_KD, KIND_FMM_OCT, KIND_DIRECT2, KIND_FMM_OCT_SYM = 0, 1, 2, 3, 4
SCHEME_EULER, SCHEME_PRE_EULER, SCHEME_LEAPFROG, SCHEME_FR, SCHEME_PEFRL = 0, 1, 2, 3, 4


def build(force=False):
    """Compile both oracle libraries with oracle/Makefile (gcc only, seconds)."""
    libs = [os.path.join(_HERE, n) for n in ("liboracle_f32.so", "liboracle_f64.so")]
    src = os.path.join(_HERE, "nbco_oracle.cpp")
    stale = force or any((not os.path.exists(l)) or os.path.getmtime(l) < os.path.getmtime(src) for l in libs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return libs


class Oracle:
    def __init__(self, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        build()
        name = "liboracle_f32.so" if self.dtype == np.float32 else "liboracle_f64.so"
        self.lib = C.CDLL(os.path.join(_HERE, name))
        self.creal = C.c_float if self.dtype == np.float32 else C.c_double
        r = self.creal

        class Opts(C.Structure):
            _fields_ = [("p", C.c_int), ("radius", r), ("eps2", r), ("coll", C.c_int),
                        ("unsort", C.c_int), ("dens_inhom", r), ("threads", C.c_int)]
        self.Opts = Opts
        L = self.lib
        P = C.c_void_p
        L.oracle_real_bytes.restype = C.c_int
        assert L.oracle_real_bytes() == self.dtype.itemsize
        L.oracle_direct2.argtypes = [P, P, C.c_int, P, r, C.c_int]
        L.oracle_direct3.argtypes = [P, P, C.c_int, P, r, C.c_int]
        L.oracle_step.argtypes = [P, P, r, C.c_longlong]
        L.oracle_add_elastic.argtypes = [P, P, C.c_int, P]
        L.oracle_rescale.argtypes = [P, C.c_int, r]
        L.oracle_fmm_kd.argtypes = [P, P, C.c_int, P, C.POINTER(Opts)]
        L.oracle_fmm_kd.restype = C.c_int
        L.oracle_fmm_oct_traceless.argtypes = [P, P, C.c_int, P, C.POINTER(Opts)]
        L.oracle_fmm_oct_traceless.restype = C.c_int
        L.oracle_fmm_oct_symmetric.argtypes = [P, P, C.c_int, P, C.POINTER(Opts)]
        L.oracle_fmm_oct_symmetric.restype = C.c_int
        L.oracle_kd_levels.argtypes = [C.c_int, C.c_int, r]
        L.oracle_oct_levels.argtypes = [C.c_int, C.c_int, r]
        L.oracle_kd_list_size.argtypes = [C.c_int]
        L.oracle_kd_list_size.restype = C.c_longlong
        L.oracle_kd_get_ints.argtypes = [C.c_int, P]
        L.oracle_kd_get_reals.argtypes = [C.c_int, P]
        L.oracle_oct_get_ints.argtypes = [C.c_int, P]
        L.oracle_oct_get_reals.argtypes = [C.c_int, P]
        L.oracle_oct_get_reals.restype = None
        L.oracle_compute_force.argtypes = [C.c_int, P, C.c_int, P, C.POINTER(Opts), C.c_int]
        L.oracle_integrate.argtypes = [C.c_int, C.c_int, P, C.c_int, P, C.c_longdouble, C.c_longdouble,
                                       C.POINTER(Opts), C.c_int]
        L.oracle_init_reference.argtypes = [P, C.c_int, P, P, C.c_int, C.c_ulonglong, C.c_ulonglong]
        L.oracle_mean_relerr.argtypes = [P, P, C.c_int]
        L.oracle_mean_relerr.restype = r
        L.oracle_minmax.argtypes = [P, C.c_int, P]
        L.oracle_pow_sum.argtypes = [P, C.c_int, C.c_int, P]
        L.oracle_energy.argtypes = [P, C.c_int, P, r, C.c_int, P]
        L.oracle_op_gradient.argtypes = [P, C.c_int, P, r, r]
        L.oracle_op_p2m.argtypes = [P, C.c_int, P, C.c_int, P]
        L.oracle_op_m2m.argtypes = [P, P, C.c_int, P]
        L.oracle_op_m2l.argtypes = [P, P, C.c_int, P, r]
        L.oracle_op_l2l.argtypes = [P, P, C.c_int, P]
        L.oracle_op_l2p.argtypes = [P, P, C.c_int, P]

    # ---- helpers -------------------------------------------------------------------------
    def arr(self, x):
        return np.ascontiguousarray(x, dtype=self.dtype)

    @staticmethod
    def ptr(a):
        return a.ctypes.data_as(C.c_void_p) if a is not None else None

    def opts(self, p=3, radius=1.0, eps2=1e-18, coll=True, unsort=True, dens_inhom=1.0, threads=1):
        return self.Opts(int(p), radius, eps2, int(coll), int(unsort), dens_inhom, int(threads))

    def params(self, n, xi=XI, omega0=OMEGA0):
        """par[] of main3.cu:685-692, arithmetic in SCAL."""
        t = self.dtype.type
        om = [t(w) for w in omega0]
        return np.array([t(xi) / t(n), 0, 0, om[0] * om[0], om[1] * om[1], om[2] * om[2]], dtype=self.dtype)

    # ---- initial conditions ---------------------------------------------------------------
    def init_reference(self, n, sigma_x=SIGMA_X, sigma_u=None, test_mode=False,
                       seed=REF_SEED, discard=REF_DISCARD):
        """[pos|vel|acc] buffer exactly as main3.cu:655-666 builds it (acc zero-filled here)."""
        t = self.dtype.type
        sx = np.array([t(s) for s in sigma_x], dtype=self.dtype)
        if sigma_u is None:
            su = np.array([t(w) for w in OMEGA0], dtype=self.dtype) * sx     # u = omega0 * x, main3.cu:245
        else:
            su = np.array([t(s) for s in sigma_u], dtype=self.dtype)
        buf = np.zeros(9 * n, dtype=self.dtype)
        self.lib.oracle_init_reference(self.ptr(buf), n, self.ptr(sx), self.ptr(su), int(test_mode), seed, discard)
        return buf.reshape(3, n, 3)

    # ---- evaluators -----------------------------------------------------------------------
    def direct2(self, pos, param=None, eps2=1e-18, threads=1):
        pos = self.arr(pos); a = np.empty_like(pos)
        self.lib.oracle_direct2(self.ptr(pos), self.ptr(a), len(pos), self.ptr(param), eps2, threads)
        return a

    def direct3(self, pos, param=None, eps2=1e-18, threads=1):
        pos = self.arr(pos); a = np.empty_like(pos)
        self.lib.oracle_direct3(self.ptr(pos), self.ptr(a), len(pos), self.ptr(param), eps2, threads)
        return a

    def fmm_kd(self, posvel, param, **kw):
        """posvel: (2, n, 3) array [pos|vel]; returns (posvel_out, acc) after the evaluation."""
        pv = self.arr(posvel).copy(); n = pv.shape[1]
        a = np.zeros((n, 3), dtype=self.dtype)
        o = self.opts(**kw)
        rc = self.lib.oracle_fmm_kd(self.ptr(pv), self.ptr(a), n, self.ptr(param), C.byref(o))
        assert rc == 0, rc
        return pv, a

    def fmm_oct_traceless(self, posvel, param, **kw):
        pv = self.arr(posvel).copy(); n = pv.shape[1]
        a = np.zeros((n, 3), dtype=self.dtype)
        o = self.opts(**kw)
        rc = self.lib.oracle_fmm_oct_traceless(self.ptr(pv), self.ptr(a), n, self.ptr(param), C.byref(o))
        assert rc == 0, rc
        return pv, a

    def fmm_oct_symmetric(self, posvel, param, **kw):
        """fmm_cart3_cpu (fmm_cart3_symmetric.cuh:582-716): the octree evaluator with symmetric multipoles of orders 0..p"""
        pv = self.arr(posvel).copy(); n = pv.shape[1]
        a = np.zeros((n, 3), dtype=self.dtype)
        o = self.opts(**kw)
        rc = self.lib.oracle_fmm_oct_symmetric(self.ptr(pv), self.ptr(a), n, self.ptr(param), C.byref(o))
        assert rc == 0, rc
        return pv, a

    def kd_tree(self, offM=None, offL=None):
        """Arrays of the tree built by the last fmm_kd call."""
        L = self.lib
        ntot = L.oracle_kd_ntot()
        out = {"L": L.oracle_kd_L(), "ntot": ntot}
        for i, k in enumerate(["mult", "index", "splitdim"]):
            a = np.empty(ntot, dtype=np.int32); L.oracle_kd_get_ints(i, self.ptr(a)); out[k] = a
        for i, k in enumerate(["center", "lbound", "rbound"]):
            a = np.empty((ntot, 3), dtype=self.dtype); L.oracle_kd_get_reals(i, self.ptr(a)); out[k] = a
        for which, k in ((4, "p2p"), (5, "m2l")):
            cnt = L.oracle_kd_list_size(which - 4)
            a = np.empty((cnt, 2), dtype=np.int32)
            if cnt:
                L.oracle_kd_get_ints(which, self.ptr(a))
            out[k] = a
        if offM is not None:
            a = np.empty((ntot, offM), dtype=self.dtype); L.oracle_kd_get_reals(3, self.ptr(a)); out["mpole"] = a
        if offL is not None:
            a = np.empty((ntot, offL), dtype=self.dtype); L.oracle_kd_get_reals(4, self.ptr(a)); out["local"] = a
        return out

    def kd_unsort(self, n):
        a = np.empty(n, dtype=np.int32); self.lib.oracle_kd_get_ints(3, self.ptr(a)); return a

    def oct_tree(self, n):
        L = self.lib
        ntot = L.oracle_oct_ntot()
        out = {"L": L.oracle_oct_L(), "ntot": ntot}
        for i, (k, m) in enumerate([("mult", ntot), ("index", ntot), ("keys", n), ("perm", n)]):
            a = np.empty(m, dtype=np.int32); L.oracle_oct_get_ints(i, self.ptr(a)); out[k] = a
        return out

    def oct_expansions(self, p, symmetric=False):
        """centre / mpole / local arrays of the last octree call (order p); symmetric: multipole tuples of (p+1)(p+2)(p+3)/6 reals"""
        L = self.lib
        ntot, off = L.oracle_oct_ntot(), (p + 1) ** 2
        offm = (p + 1) * (p + 2) * (p + 3) // 6 if symmetric else off
        out = {}
        for i, (k, shape) in enumerate([("center", (ntot, 3)), ("mpole", (ntot, offm)), ("local", (ntot, off))]):
            a = np.empty(shape, dtype=self.dtype); L.oracle_oct_get_reals(i, self.ptr(a)); out[k] = a
        return out

    def compute_force(self, kind, buf, param, elastic=True, **kw):
        o = self.opts(**kw)
        self.lib.oracle_compute_force(kind, self.ptr(buf), buf.shape[1], self.ptr(param), C.byref(o), int(elastic))

    def integrate(self, scheme, kind, buf, param, dt, scale=1.0, elastic=True, **kw):
        o = self.opts(**kw)
        self.lib.oracle_integrate(scheme, kind, self.ptr(buf), buf.shape[1], self.ptr(param),
                                  dt, scale, C.byref(o), int(elastic))

    # ---- small kernels --------------------------------------------------------------------
    def step(self, b, a, ds):
        self.lib.oracle_step(self.ptr(b), self.ptr(a), ds, b.size)

    def add_elastic(self, p, a, k=None):
        self.lib.oracle_add_elastic(self.ptr(p), self.ptr(a), len(p), self.ptr(k))

    def rescale(self, a, c):
        self.lib.oracle_rescale(self.ptr(a), len(a), c)

    def mean_relerr(self, x, ref):
        x = self.arr(x); ref = self.arr(ref)
        return float(self.lib.oracle_mean_relerr(self.ptr(x), self.ptr(ref), len(x)))

    def minmax(self, p):
        p = self.arr(p); out = np.empty(6, dtype=self.dtype)
        self.lib.oracle_minmax(self.ptr(p), len(p), self.ptr(out))
        return out.reshape(2, 3)

    def pow_sum(self, x, expo):
        x = self.arr(x); out = np.empty(3, dtype=np.float64)
        self.lib.oracle_pow_sum(self.ptr(x), len(x), expo, self.ptr(out))
        return out

    def energy(self, buf, param, eps2=1e-18, threads=1):
        buf = self.arr(buf); out = np.empty(3, dtype=np.float64)
        self.lib.oracle_energy(self.ptr(buf), buf.shape[1], self.ptr(param), eps2, threads, self.ptr(out))
        return out
