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
