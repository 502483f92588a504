"""coulomb_oscillators_amd -- MI355X (gfx950) force / integrate engine for N-body Coulomb oscillators.

The product is the C-ABI library ``libnbco_hip.so`` (sources in ``csrc/``, interface in
``include/nbco.h``) plus the C++20 ``nbco3`` host in ``host/``.  This Python package is the thin
plumbing used by the tests and by ``bench.py``: it loads the library with ctypes and passes
PyTorch device pointers / streams through the C ABI.  There is no CPU fallback: importing works
anywhere, but creating an :class:`Engine` without the built library or without a HIP device raises.
"""
from .engine import (Engine, EngineError, lib_path, build_library, default_opts,
                     EVAL_DIRECT, EVAL_DIRECT_KAHAN, EVAL_FMM_KDTREE, EVAL_FMM_TRACELESS, EVAL_FMM_SYMMETRIC,
                     INTEG_EULER, INTEG_PRE_EULER, INTEG_LEAPFROG, INTEG_FORESTRUTH, INTEG_PEFRL,
                     PHASES)
from .dist import DomainRun, TorchComm, SingleComm, LoopbackWorld, SlabRun, LoopbackSlabs

__all__ = ["Engine", "EngineError", "lib_path", "build_library", "default_opts",
           "EVAL_DIRECT", "EVAL_DIRECT_KAHAN", "EVAL_FMM_KDTREE", "EVAL_FMM_TRACELESS", "EVAL_FMM_SYMMETRIC",
           "INTEG_EULER", "INTEG_PRE_EULER", "INTEG_LEAPFROG", "INTEG_FORESTRUTH", "INTEG_PEFRL", "PHASES",
           "DomainRun", "TorchComm", "SingleComm", "LoopbackWorld", "SlabRun", "LoopbackSlabs"]
