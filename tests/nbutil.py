"""Shared helpers for the parity tests."""
import numpy as np


def force_err(a, ref):
    """max_i |a_i - ref_i| / (|ref_i| + mean_j |ref_j|).

    The per-particle relative error, regularised by the mean force magnitude so that particles
    near the trap centre (net force ~ 0 by cancellation) do not dominate.
    """
    a = np.asarray(a, dtype=np.float64).reshape(-1, 3)
    ref = np.asarray(ref, dtype=np.float64).reshape(-1, 3)
    mag = np.linalg.norm(ref, axis=1)
    return float((np.linalg.norm(a - ref, axis=1) / (mag + mag.mean() + 1e-300)).max())


def directed_pairs(mult, p2p, L):
    mult = np.asarray(mult, dtype=np.int64)
    leaves = mult[(1 << L) - 1:]
    return int((2 * mult[p2p[:, 0]] * mult[p2p[:, 1]]).sum() + (leaves ** 2).sum())


def canon_pairs(pairs):
    """Order-independent representation of an unordered pair list."""
    p = np.sort(np.asarray(pairs, dtype=np.int64).reshape(-1, 2), axis=1)
    keys = p[:, 0] * (1 << 32) + p[:, 1]
    return np.sort(keys)
