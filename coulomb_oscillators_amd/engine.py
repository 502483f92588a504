"""ctypes binding of libnbco_hip.so (include/nbco.h).

Host-side mirror of the reference's function-pointer interface for this path:

    evaluator   void f(VEC *p, VEC *a, int n, const SCAL *param)      direct.cuh:233, fmm_cart3_kdtree.cuh:1478
    step        void step(VEC *b, const VEC *a, SCAL ds, int n)       kernel.cuh:100
    integrator  void leapfrog(f, SCAL *buf, int n, param, dt, step_func, scale)   integrator.cuh:68

Tensors are PyTorch CUDA(=HIP) tensors; only their device pointers cross the C ABI.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("NBCO_LIB") or os.path.join(_HERE, "libnbco_hip.so")   # NBCO_LIB: A/B builds of the same ABI (diagnostics)

EVAL_DIRECT, EVAL_DIRECT_KAHAN, EVAL_FMM_KDTREE, EVAL_FMM_TRACELESS, EVAL_FMM_SYMMETRIC = 0, 1, 2, 3, 4
INTEG_EULER, INTEG_PRE_EULER, INTEG_LEAPFROG, INTEG_FORESTRUTH, INTEG_PEFRL = 0, 1, 2, 3, 4
PHASES = ["build", "p2m_m2m", "traverse", "lists", "p2p", "m2l", "l2l", "l2p", "finish", "direct", "axpy"]

KD_FIELDS = {"mult": 0, "index": 1, "splitdim": 2, "center": 3, "lbound": 4, "rbound": 5,
             "mpole": 6, "local": 7, "p2p": 8, "m2l": 9, "unsort": 10, "order": 11}


class EngineError(RuntimeError):
    """status is the NBCO_ERR_* code of include/nbco.h when the error came from the library (None otherwise)"""

    def __init__(self, msg, status=None):
        super().__init__(msg)
        self.status = status


ERR_UNSUPPORTED = 4   # NBCO_ERR_UNSUPPORTED


class Opts(C.Structure):
    _fields_ = [("fmm_order", C.c_int), ("tree_radius", C.c_float), ("eps2", C.c_float), ("coll", C.c_int),
                ("unsort", C.c_int), ("dens_inhom", C.c_float), ("tree_L", C.c_int), ("tree_steps", C.c_int),
                ("m2l_first", C.c_int), ("sync", C.c_int), ("list_factor", C.c_int), ("list_grow", C.c_int), ("far_fp64", C.c_int),
                ("p2p_mutual", C.c_int), ("track_order", C.c_int), ("stream", C.c_void_p)]


class KdInfo(C.Structure):
    _fields_ = [("L", C.c_int), ("ntot", C.c_int), ("order", C.c_int), ("mlt_max", C.c_int), ("n", C.c_longlong),
                ("p2p_pairs", C.c_longlong), ("m2l_pairs", C.c_longlong), ("directed_p2p", C.c_longlong),
                ("rebuilt", C.c_int), ("build_mode", C.c_int), ("p2p_halves", C.c_int), ("warm_builds", C.c_longlong), ("warm_misses", C.c_longlong),
                ("real_bytes", C.c_int), ("long_lists", C.c_int)]


class OctInfo(C.Structure):
    _fields_ = [("L", C.c_int), ("ntot", C.c_int), ("order", C.c_int), ("tpl", C.c_int), ("n", C.c_longlong),
                ("m2l_entries", C.c_longlong), ("p2p_groups", C.c_longlong), ("p2p_desc", C.c_longlong),
                ("p2p_chunks", C.c_longlong), ("real_bytes", C.c_int), ("mpole_reals", C.c_int)]


OCT_FIELDS = {"mult": 0, "index": 1, "center4": 2, "mpole": 3, "local": 4, "keys": 5, "perm": 6}


class DistLayout(C.Structure):
    _fields_ = [("world", C.c_int), ("rank", C.c_int), ("d", C.c_int), ("L", C.c_int), ("L_local", C.c_int),
                ("ntot_local", C.c_int), ("order", C.c_int), ("n_global", C.c_longlong), ("n_local", C.c_longlong),
                ("nodes_bytes", C.c_longlong), ("pos_bytes", C.c_longlong), ("csz_bytes", C.c_longlong),
                ("mpole_bytes", C.c_longlong), ("let_node_bytes", C.c_longlong), ("let_counts", C.c_int)]


class DistStep(C.Structure):
    """nbco_dist_step: the collective a distributed re-partition asks its caller to run next (include/nbco.h)"""
    _fields_ = [("op", C.c_int), ("row_bytes", C.c_int), ("send_off", C.c_longlong), ("recv_off", C.c_longlong), ("count", C.c_longlong),
                ("rows_send", C.c_longlong * 64), ("rows_recv", C.c_longlong * 64)]


COLL_DONE, COLL_ALLREDUCE_MIN_I32, COLL_ALLREDUCE_SUM_I32, COLL_ALLGATHER, COLL_ALLTOALL = range(5)


def lib_path():
    return _LIB


def build_library(force=False):
    """Compile csrc/*.hip for gfx950 into libnbco_hip.so (hipcc cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", csrc, "-s", "-j8"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    return _LIB


_lib = None


def _load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        raise EngineError("libnbco_hip.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                          "there is no CPU fallback")
    L = C.CDLL(_LIB)
    P, LL, I, F, D = C.c_void_p, C.c_longlong, C.c_int, C.c_float, C.c_double
    sig = {
        "nbco_opts_default": [C.POINTER(Opts)],
        "nbco_create": [C.POINTER(P), C.POINTER(Opts)],
        "nbco_destroy": [P],
        "nbco_set_opts": [P, C.POINTER(Opts)],
        "nbco_get_opts": [P, C.POINTER(Opts)],
        "nbco_sync": [P],
        "nbco_step": [P, P, P, F, LL],
        "nbco_add_elastic": [P, P, P, LL, P],
        "nbco_elastic": [P, P, P, LL, P],
        "nbco_rescale": [P, P, LL, P],
        "nbco_gather": [P, P, P, P, LL],
        "nbco_gather_inverse": [P, P, P, P, LL],
        "nbco_copy": [P, P, P, LL],
        "nbco_direct": [P, P, P, LL, P],
        "nbco_direct3": [P, P, P, LL, P],
        "nbco_fmm_kdtree": [P, P, P, LL, P],
        "nbco_fmm_traceless": [P, P, P, LL, P],
        "nbco_fmm_symmetric": [P, P, P, LL, P],
        "nbco_fmm_oct_shard": [P, P, P, LL, P, I, I, I, C.POINTER(LL)],
        "nbco_force": [P, I, P, LL, P, I],
        "nbco_integrate": [P, I, I, P, LL, P, D, D, I],
        "nbco_integrate_steps": [P, I, I, P, LL, P, D, D, I, I],
        "nbco_minmax": [P, P, LL, P],
        "nbco_mean_relerr": [P, P, P, LL, C.POINTER(F)],
        "nbco_pow_sum": [P, P, I, LL, C.POINTER(D)],
        "nbco_energy": [P, P, LL, P, C.POINTER(D)],
        "nbco_energy_fmm": [P, P, LL, P, C.POINTER(D)],
        "nbco_kd_get_info": [P, C.POINTER(KdInfo)],
        "nbco_kd_copy": [P, I, P, LL],
        "nbco_oct_get_info": [P, C.POINTER(OctInfo)],
        "nbco_oct_copy": [P, I, P, LL],
        "nbco_dist_layout_query": [P, LL, I, I, C.POINTER(DistLayout)],
        "nbco_dist_partition": [P, P, LL, I, I, P],
        "nbco_dist_local": [P, P, LL, P, P],
        "nbco_dist_local_build": [P, P, LL, P],
        "nbco_dist_local_upward": [P, P, LL, P],
        "nbco_dist_finish": [P, P, P, P, P, P],
        "nbco_dist_local_geom": [P, P, LL, P, P],
        "nbco_dist_local_mpole": [P, P, LL, P],
        "nbco_dist_finish_traverse": [P, P, P],
        "nbco_dist_finish_rest": [P, P, P, P, P],
        "nbco_dist_repartition_workspace": [P, LL, I, C.POINTER(LL)],
        "nbco_dist_repartition_begin": [P, P, LL, I, I, P, LL, C.POINTER(DistStep)],
        "nbco_dist_repartition_next": [P, C.POINTER(DistStep)],
        "nbco_dist_let_local_geom": [P, P, LL, P],
        "nbco_dist_let_local_mpole": [P, P, LL],
        "nbco_dist_let_select": [P, P, P],
        "nbco_dist_let_pack": [P, P, P, P],
        "nbco_dist_let_finish": [P, P, P, P, P, P, P],
        "nbco_dist_let_check": [P],
        "nbco_dist_let_pack_capped": [P, P, P, P],
        "nbco_dist_let_finish_capped": [P, P, P, P, P, P, P],
        "nbco_dist_let_settle": [P, I],
        "nbco_dist_turnaround": [P, P, LL, P, D, D, I],
        "nbco_aux_stream": [P, C.POINTER(C.c_void_p)],
        "nbco_debug_violations": [P, C.POINTER(LL)],
        "nbco_profile_enable": [P, I],
        "nbco_profile_reset": [P],
        "nbco_profile_get": [P, I, C.POINTER(D), C.POINTER(LL)],
    }
    for name, args in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = I
    L.nbco_last_error.argtypes = [P]
    L.nbco_last_error.restype = C.c_char_p
    _lib = L
    return L


def default_opts(**kw):
    o = Opts()
    _load().nbco_opts_default(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class Engine:
    """One nbco context bound to the current torch HIP device and stream."""

    def __init__(self, stream=None, **opts):
        import torch
        self.lib = _load()
        if not torch.cuda.is_available():
            raise EngineError("no HIP device visible to PyTorch; the engine has no CPU fallback")
        self.torch = torch
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        o = default_opts(**opts)
        o.stream = stream
        self.ctx = C.c_void_p()
        rc = self.lib.nbco_create(C.byref(self.ctx), C.byref(o))
        if rc != 0:
            raise EngineError("nbco_create failed with status %d" % rc)

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.nbco_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise EngineError("nbco status %d: %s" % (rc, self.lib.nbco_last_error(self.ctx).decode()), status=rc)

    # ---- options --------------------------------------------------------------------------------
    def opts(self):
        o = Opts()
        self._chk(self.lib.nbco_get_opts(self.ctx, C.byref(o)))
        return o

    def set(self, **kw):
        o = self.opts()
        for k, v in kw.items():
            setattr(o, k, v)
        self._chk(self.lib.nbco_set_opts(self.ctx, C.byref(o)))

    def sync(self):
        self._chk(self.lib.nbco_sync(self.ctx))

    # ---- basic kernels --------------------------------------------------------------------------
    def step(self, b, a, ds, n=None):
        n = b.numel() // 3 if n is None else n
        self._chk(self.lib.nbco_step(self.ctx, _ptr(b), _ptr(a), ds, n))

    def add_elastic(self, p, a, n, k=None):
        self._chk(self.lib.nbco_add_elastic(self.ctx, _ptr(p), _ptr(a), n, _ptr(k)))

    def elastic(self, p, a, n, k=None):
        self._chk(self.lib.nbco_elastic(self.ctx, _ptr(p), _ptr(a), n, _ptr(k)))

    def rescale(self, a, n, param):
        self._chk(self.lib.nbco_rescale(self.ctx, _ptr(a), n, _ptr(param)))

    def gather(self, dst, src, idx, n):
        self._chk(self.lib.nbco_gather(self.ctx, _ptr(dst), _ptr(src), _ptr(idx), n))

    def gather_inverse(self, dst, src, idx, n):
        self._chk(self.lib.nbco_gather_inverse(self.ctx, _ptr(dst), _ptr(src), _ptr(idx), n))

    def copy(self, dst, src, n):
        self._chk(self.lib.nbco_copy(self.ctx, _ptr(dst), _ptr(src), n))

    # ---- evaluators: f(p, a, n, param) -----------------------------------------------------------
    def direct(self, p, a, n, param=None):
        self._chk(self.lib.nbco_direct(self.ctx, _ptr(p), _ptr(a), n, _ptr(param)))

    def direct3(self, p, a, n, param=None):
        self._chk(self.lib.nbco_direct3(self.ctx, _ptr(p), _ptr(a), n, _ptr(param)))

    def fmm_cart3_kdtree(self, p, a, n, param=None):
        self._chk(self.lib.nbco_fmm_kdtree(self.ctx, _ptr(p), _ptr(a), n, _ptr(param)))

    def fmm_cart3_traceless(self, p, a, n, param=None):
        self._chk(self.lib.nbco_fmm_traceless(self.ctx, _ptr(p), _ptr(a), n, _ptr(param)))

    def fmm_cart3(self, p, a, n, param=None):
        """the uniform-octree evaluator with symmetric multipoles (fmm_cart3_symmetric.cuh:413)"""
        self._chk(self.lib.nbco_fmm_symmetric(self.ctx, _ptr(p), _ptr(a), n, _ptr(param)))

    def fmm_oct_shard(self, p, a, n, param, world, rank, symmetric=False):
        """this rank's slab of the uniform-octree evaluation (nbco_fmm_oct_shard); returns the particle boundaries of all slabs"""
        b = (C.c_longlong * (world + 1))()
        self._chk(self.lib.nbco_fmm_oct_shard(self.ctx, _ptr(p), _ptr(a), n, _ptr(param), int(symmetric), world, rank, b))
        return list(b)

    def compute_force(self, kind, buf, n, param, elastic=True):
        self._chk(self.lib.nbco_force(self.ctx, kind, _ptr(buf), n, _ptr(param), int(elastic)))

    def integrate(self, scheme, kind, buf, n, param, dt, scale=1.0, elastic=True):
        self._chk(self.lib.nbco_integrate(self.ctx, scheme, kind, _ptr(buf), n, _ptr(param), dt, scale, int(elastic)))

    def integrate_steps(self, scheme, kind, buf, n, param, dt, steps, scale=1.0, elastic=True):
        """`steps` steps in one call (nbco_integrate_steps): same final state as `steps` calls of integrate()"""
        self._chk(self.lib.nbco_integrate_steps(self.ctx, scheme, kind, _ptr(buf), n, _ptr(param), dt, scale, int(elastic), int(steps)))

    # ---- reductions -----------------------------------------------------------------------------
    def minmax(self, p, n):
        out = self.torch.empty(6, dtype=self.torch.float32, device=p.device)
        self._chk(self.lib.nbco_minmax(self.ctx, _ptr(p), n, _ptr(out)))
        return out.view(2, 3)

    def mean_relerr(self, x, ref, n):
        out = C.c_float()
        self._chk(self.lib.nbco_mean_relerr(self.ctx, _ptr(x), _ptr(ref), n, C.byref(out)))
        return out.value

    def pow_sum(self, x, expo, n):
        out = (C.c_double * 3)()
        self._chk(self.lib.nbco_pow_sum(self.ctx, _ptr(x), expo, n, out))
        return list(out)

    def energy(self, buf, n, param):
        out = (C.c_double * 3)()
        self._chk(self.lib.nbco_energy(self.ctx, _ptr(buf), n, _ptr(param), out))
        return list(out)

    def energy_fmm(self, buf, n, param):
        """{kinetic, elastic, coulomb} with the Coulomb part from the lists of the last kd-tree evaluation (O(N log N))"""
        out = (C.c_double * 3)()
        self._chk(self.lib.nbco_energy_fmm(self.ctx, _ptr(buf), n, _ptr(param), out))
        return list(out)

    # ---- multi-GPU kd-domain sharding (see dist.py for the orchestration) ---------------------------
    def dist_layout(self, n_global, world, rank):
        lay = DistLayout()
        self._chk(self.lib.nbco_dist_layout_query(self.ctx, n_global, world, rank, C.byref(lay)))
        return lay

    def dist_partition(self, state_all, n_global, world, rank, state_local):
        self._chk(self.lib.nbco_dist_partition(self.ctx, _ptr(state_all), n_global, world, rank, _ptr(state_local)))

    def dist_local(self, buf_local, n_local, nodes_send, pos_send):
        self._chk(self.lib.nbco_dist_local(self.ctx, _ptr(buf_local), n_local, _ptr(nodes_send), _ptr(pos_send)))

    def dist_local_build(self, buf_local, n_local, pos_send):
        self._chk(self.lib.nbco_dist_local_build(self.ctx, _ptr(buf_local), n_local, _ptr(pos_send)))

    def dist_local_upward(self, buf_local, n_local, nodes_send):
        self._chk(self.lib.nbco_dist_local_upward(self.ctx, _ptr(buf_local), n_local, _ptr(nodes_send)))

    # the same evaluation with the node block in two all-gathers (traversal records first, multipoles later)
    def dist_local_geom(self, buf_local, n_local, pos_send, csz_send):
        self._chk(self.lib.nbco_dist_local_geom(self.ctx, _ptr(buf_local), n_local, _ptr(pos_send), _ptr(csz_send)))

    def dist_local_mpole(self, buf_local, n_local, mpole_send):
        self._chk(self.lib.nbco_dist_local_mpole(self.ctx, _ptr(buf_local), n_local, _ptr(mpole_send)))

    def dist_finish_traverse(self, csz_all, pos_all):
        self._chk(self.lib.nbco_dist_finish_traverse(self.ctx, _ptr(csz_all), _ptr(pos_all)))

    def dist_finish_rest(self, mpole_all, buf_local, a_local, param=None):
        self._chk(self.lib.nbco_dist_finish_rest(self.ctx, _ptr(mpole_all), _ptr(buf_local), _ptr(a_local), _ptr(param)))

    # the re-partition without gathering the state (include/nbco.h: nbco_dist_repartition_*)
    def dist_repartition_workspace(self, n_global, world):
        b = C.c_longlong()
        self._chk(self.lib.nbco_dist_repartition_workspace(self.ctx, n_global, world, C.byref(b)))
        return int(b.value)

    def dist_repartition_begin(self, state_local, n_global, world, rank, work):
        st = DistStep()
        self._chk(self.lib.nbco_dist_repartition_begin(self.ctx, _ptr(state_local), n_global, world, rank, _ptr(work), work.numel() * work.element_size(),
                                                       C.byref(st)))
        return st

    def dist_repartition_next(self):
        st = DistStep()
        self._chk(self.lib.nbco_dist_repartition_next(self.ctx, C.byref(st)))
        return st

    # the same evaluation with the LET exchange (include/nbco.h: nbco_dist_let_*)
    def dist_let_local_geom(self, buf_local, n_local, csz_send):
        self._chk(self.lib.nbco_dist_let_local_geom(self.ctx, _ptr(buf_local), n_local, _ptr(csz_send)))

    def dist_let_local_mpole(self, buf_local, n_local):
        self._chk(self.lib.nbco_dist_let_local_mpole(self.ctx, _ptr(buf_local), n_local))

    def dist_let_select(self, csz_all, counts_send):
        self._chk(self.lib.nbco_dist_let_select(self.ctx, _ptr(csz_all), _ptr(counts_send)))

    def dist_let_pack(self, counts_all_host, pos_send, mpole_send):
        self._chk(self.lib.nbco_dist_let_pack(self.ctx, _ptr(counts_all_host), _ptr(pos_send), _ptr(mpole_send)))

    def dist_let_finish(self, counts_all_host, pos_recv, mpole_recv, buf_local, a_local, param=None):
        self._chk(self.lib.nbco_dist_let_finish(self.ctx, _ptr(counts_all_host), _ptr(pos_recv), _ptr(mpole_recv), _ptr(buf_local), _ptr(a_local),
                                                _ptr(param)))

    def dist_turnaround(self, buf_local, n_local, param, dt, scale=1.0, elastic=True):
        """one pass between two force evaluations of a sharded leapfrog run (nbco_dist_turnaround)"""
        self._chk(self.lib.nbco_dist_turnaround(self.ctx, _ptr(buf_local), n_local, _ptr(param), dt, scale, int(elastic)))

    def dist_let_check(self):
        self._chk(self.lib.nbco_dist_let_check(self.ctx))

    # the exchange with segments sized ahead of the counts (nbco_dist_let_pack_capped): caps_* are host int64 tensors of 2 world values
    def dist_let_pack_capped(self, caps_out_host, pos_send, mpole_send):
        self._chk(self.lib.nbco_dist_let_pack_capped(self.ctx, _ptr(caps_out_host), _ptr(pos_send), _ptr(mpole_send)))

    def dist_let_finish_capped(self, caps_in_host, pos_recv, mpole_recv, buf_local, a_local, param=None):
        self._chk(self.lib.nbco_dist_let_finish_capped(self.ctx, _ptr(caps_in_host), _ptr(pos_recv), _ptr(mpole_recv), _ptr(buf_local), _ptr(a_local),
                                                       _ptr(param)))

    def dist_let_settle(self, ok):
        self._chk(self.lib.nbco_dist_let_settle(self.ctx, int(bool(ok))))

    def aux_stream(self):
        """raw hipStream_t of the context's second stream (see nbco_aux_stream)"""
        out = C.c_void_p()
        self._chk(self.lib.nbco_aux_stream(self.ctx, C.byref(out)))
        return out.value or 0

    def dist_finish(self, nodes_all, pos_all, buf_local, a_local, param=None):
        self._chk(self.lib.nbco_dist_finish(self.ctx, _ptr(nodes_all), _ptr(pos_all), _ptr(buf_local), _ptr(a_local), _ptr(param)))

    # ---- kd-tree introspection ------------------------------------------------------------------
    def kd_info(self):
        info = KdInfo()
        self._chk(self.lib.nbco_kd_get_info(self.ctx, C.byref(info)))
        return info

    def kd_array(self, name):
        import numpy as np
        info = self.kd_info()
        p = info.order
        offM, offL = p * (p + 1) * (p + 2) // 6, (p + 1) ** 2
        real = np.float64 if info.real_bytes == 8 else np.float32      # doubles after an evaluation with far_fp64
        shapes = {"mult": ((info.ntot,), np.int32), "index": ((info.ntot,), np.int32), "splitdim": ((info.ntot,), np.int32),
                  "center": ((info.ntot, 3), np.float32), "lbound": ((info.ntot, 3), np.float32),
                  "rbound": ((info.ntot, 3), np.float32), "mpole": ((info.ntot, offM), real),
                  "local": ((info.ntot, offL), real), "p2p": ((info.p2p_pairs, 2), np.int32),
                  "m2l": ((info.m2l_pairs, 2), np.int32), "unsort": ((info.n,), np.int32), "order": ((info.n,), np.int32)}
        shape, dt = shapes[name]
        out = np.empty(shape, dtype=dt)
        if out.size:
            self._chk(self.lib.nbco_kd_copy(self.ctx, KD_FIELDS[name], out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    # ---- octree introspection -----------------------------------------------------------------------
    def oct_info(self):
        info = OctInfo()
        self._chk(self.lib.nbco_oct_get_info(self.ctx, C.byref(info)))
        return info

    def oct_array(self, name):
        import numpy as np
        info = self.oct_info()
        off = (info.order + 1) ** 2
        real = np.float64 if info.real_bytes == 8 else np.float32
        shapes = {"mult": ((info.ntot,), np.int32), "index": ((info.ntot,), np.int32), "center4": ((info.ntot, 4), np.float32),
                  "mpole": ((info.ntot, info.mpole_reals), real), "local": ((info.ntot, off), real),
                  "keys": ((info.n,), np.uint32), "perm": ((info.n,), np.uint32)}
        shape, dt = shapes[name]
        out = np.empty(shape, dtype=dt)
        self._chk(self.lib.nbco_oct_copy(self.ctx, OCT_FIELDS[name], out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    # ---- checked build (libnbco_hip_checked.so, selected with NBCO_LIB) ------------------------------------
    def violations(self):
        """per-site counts of list-derived indices that failed their range check on the device (checked build only)"""
        out = (C.c_longlong * 8)()
        self._chk(self.lib.nbco_debug_violations(self.ctx, out))
        return list(out)

    # ---- profiling ------------------------------------------------------------------------------
    def profile(self, phases=True):
        """phases: True = all, False = off, or an iterable of phase names (see PHASES)."""
        if phases is True:
            mask = -1
        elif not phases:
            mask = 0
        else:
            mask = 0
            for name in phases:
                mask |= 1 << PHASES.index(name)
        self._chk(self.lib.nbco_profile_enable(self.ctx, mask))

    def profile_reset(self):
        self._chk(self.lib.nbco_profile_reset(self.ctx))

    def profile_get(self):
        out = {}
        for i, name in enumerate(PHASES):
            ms, cnt = C.c_double(), C.c_longlong()
            self._chk(self.lib.nbco_profile_get(self.ctx, i, C.byref(ms), C.byref(cnt)))
            out[name] = (ms.value, cnt.value)
        return out
