"""CPU tests of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports
exactly the entry points include/nbco.h declares.  No compute call is made without a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "nbco.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nbco_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_the_reference_interface():
    syms = declared_symbols()
    for need in ("nbco_direct", "nbco_direct3", "nbco_fmm_kdtree", "nbco_fmm_traceless", "nbco_step", "nbco_add_elastic",
                 "nbco_rescale", "nbco_integrate", "nbco_force", "nbco_minmax", "nbco_mean_relerr", "nbco_pow_sum"):
        assert need in syms


def test_library_exports_every_declared_symbol(engine_lib):
    lib = ctypes.CDLL(engine_lib)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_opts_default_matches_reference_globals(engine_lib):
    """constants.cuh:36-52: EPS2 = 1e-18, fmm_order = 3, tree_radius = 1, coll, b_unsort, dens_inhom = 1."""
    from coulomb_oscillators_amd import default_opts
    o = default_opts()
    assert o.fmm_order == 3 and o.tree_radius == 1.0 and o.coll == 1 and o.unsort == 1
    assert abs(o.eps2 - 1e-18) < 1e-24 and o.dens_inhom == 1.0 and o.tree_L == 0 and o.tree_steps == 1


def test_engine_fails_loudly_without_gpu(engine_lib):
    import torch
    from coulomb_oscillators_amd import Engine, EngineError
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(EngineError):
        Engine()
    # the C entry point itself also refuses (no CPU fallback)
    lib = ctypes.CDLL(engine_lib)
    ctx = ctypes.c_void_p()
    assert lib.nbco_create(ctypes.byref(ctx), None) != 0


def test_init_gaussian_is_the_reference_stream(engine_lib, oracle32):
    """nbco_init_gaussian (host only): initGA / initU of main3.cu:94-137 over mt19937_64(5351550349027530206), discard 1248 --
    bit-identical to the oracle's restatement, centred, RMS exactly sigma."""
    import numpy as np
    lib = ctypes.CDLL(engine_lib)
    lib.nbco_init_gaussian.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ulonglong,
                                       ctypes.c_ulonglong, ctypes.c_int]
    sx = np.array([0.003, 0.001, 0.01], dtype=np.float32)
    su = np.array([1.095, 1.0, 1.0], dtype=np.float32) * sx
    for n, uniform in ((4096, 0), (30001, 0), (4096, 1)):
        got = np.zeros((2, n, 3), dtype=np.float32)
        assert lib.nbco_init_gaussian(got.ctypes.data, n, sx.ctypes.data, su.ctypes.data, 5351550349027530206, 1248, uniform) == 0
        want = oracle32.init_reference(n, test_mode=bool(uniform))
        assert np.array_equal(got, want[:2])
        if not uniform:
            assert np.allclose(np.sqrt((got[0].astype(np.float64) ** 2).mean(0)), sx, rtol=1e-5)
            assert np.abs(got[0].mean(0)).max() < 1e-6 * sx.max() * 10
    assert lib.nbco_init_gaussian(None, 10, sx.ctypes.data, su.ctypes.data, 1, 0, 0) != 0
