"""nbco_init_gaussian_slice: a rank's rows of the reference's initial state without the full state in memory (host only, no GPU)."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture(scope="module")
def lib(engine_lib):
    L = C.CDLL(engine_lib)
    L.nbco_init_gaussian.argtypes = [C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_ulonglong, C.c_ulonglong, C.c_int]
    L.nbco_init_gaussian_slice.argtypes = [C.c_void_p, C.c_longlong, C.c_longlong, C.c_longlong, C.c_void_p, C.c_void_p, C.c_ulonglong, C.c_ulonglong, C.c_int]
    return L


SX = np.array([0.003, 0.001, 0.01], dtype=np.float32)
SU = np.array([1.095 * 0.003, 0.001, 0.01], dtype=np.float32)
SEED, DISCARD = 5351550349027530206, 1248


@pytest.mark.parametrize("n,uniform", [(4096, 0), (10007, 0), (5000, 1), (1, 0)])
def test_slices_equal_the_rows_of_the_full_state_bit_for_bit(lib, n, uniform):
    full = np.zeros((2, n, 3), dtype=np.float32)
    assert lib.nbco_init_gaussian(full.ctypes.data, n, SX.ctypes.data, SU.ctypes.data, SEED, DISCARD, uniform) == 0
    cuts = sorted({0, n // 3, n // 2, n})
    for first, last in zip(cuts[:-1], cuts[1:]):
        cnt = last - first
        if cnt == 0:
            continue
        part = np.zeros((2, cnt, 3), dtype=np.float32)
        assert lib.nbco_init_gaussian_slice(part.ctypes.data, n, first, cnt, SX.ctypes.data, SU.ctypes.data, SEED, DISCARD, uniform) == 0
        np.testing.assert_array_equal(part, full[:, first:last])


def test_slice_arguments_are_checked(lib):
    buf = np.zeros((2, 8, 3), dtype=np.float32)
    for n, first, cnt in ((8, -1, 4), (8, 4, 5), (8, 0, 0), (0, 0, 0)):
        assert lib.nbco_init_gaussian_slice(buf.ctypes.data, n, first, cnt, SX.ctypes.data, SU.ctypes.data, SEED, DISCARD, 0) != 0
