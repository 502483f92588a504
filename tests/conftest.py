import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle32():
    import numpy as np
    from oracle.pyoracle import Oracle
    return Oracle(np.float32)


@pytest.fixture(scope="session")
def oracle64():
    import numpy as np
    from oracle.pyoracle import Oracle
    return Oracle(np.float64)


@pytest.fixture(scope="session")
def engine_lib():
    """Path of the built C-ABI library (building it if the toolchain is here)."""
    from coulomb_oscillators_amd import build_library, lib_path
    if not os.path.exists(lib_path()):
        build_library()
    return lib_path()


@pytest.fixture()
def engine(engine_lib):
    import torch
    from coulomb_oscillators_amd import Engine
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible (there is no CPU fallback)")
    e = Engine()
    yield e
    e.close()
