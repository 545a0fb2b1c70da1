"""Distance metrics -- mirror of src/playlist.rs:24-142 (the ordering functions of playlist.rs are a
"next" row)."""
import ctypes as C
from typing import Callable, Sequence

import numpy as np

from . import _ffi


def _vec(a):
    return np.ascontiguousarray(a, dtype=np.float32).reshape(-1)


def _single(a, b, metric, m=None) -> float:
    a, b = _vec(a), _vec(b)
    if a.shape != b.shape:
        raise ValueError("vectors must have the same length")
    out = C.c_float()
    mp = None
    if m is not None:
        m = np.ascontiguousarray(m, dtype=np.float32)
        mp = m.ctypes.data
    _ffi.check(_ffi.lib().blissgpu_distance(a.ctypes.data, b.ctypes.data, a.shape[0], metric, mp, C.byref(out)))
    return float(out.value)


def euclidean_distance(a, b) -> float:
    """src/playlist.rs:65-71"""
    return _single(a, b, _ffi.METRIC_EUCLIDEAN)


def cosine_distance(a, b) -> float:
    """src/playlist.rs:76-79"""
    return _single(a, b, _ffi.METRIC_COSINE)


def mahalanobis_distance(a, b, m) -> float:
    """src/playlist.rs:140-142"""
    return _single(a, b, _ffi.METRIC_MAHALANOBIS, m)


def mahalanobis_distance_builder(m) -> Callable[[np.ndarray, np.ndarray], float]:
    """src/playlist.rs:129-131"""
    m = np.ascontiguousarray(m, dtype=np.float32).copy()
    return lambda a, b: mahalanobis_distance(a, b, m)


_METRICS = {"euclidean": _ffi.METRIC_EUCLIDEAN, "cosine": _ffi.METRIC_COSINE, "mahalanobis": _ffi.METRIC_MAHALANOBIS}


def pairwise_distances(A, B, metric="euclidean", m=None) -> np.ndarray:
    """All-pairs matrix out[i, j] = metric(A[i], B[j]) (batched form of DistanceMetric::distance,
    src/playlist.rs:24-59, 256-270)."""
    A = np.ascontiguousarray(A, dtype=np.float32)
    B = np.ascontiguousarray(B, dtype=np.float32)
    if A.ndim != 2 or B.ndim != 2 or A.shape[1] != B.shape[1]:
        raise ValueError("A and B must be [n, d] and [m, d]")
    out = np.empty((A.shape[0], B.shape[0]), np.float32)
    mp = None
    if m is not None:
        m = np.ascontiguousarray(m, dtype=np.float32)
        mp = m.ctypes.data
    _ffi.check(_ffi.lib().blissgpu_pairwise(A.ctypes.data, A.shape[0], B.ctypes.data, B.shape[0], A.shape[1],
                                            _METRICS[metric], mp, out.ctypes.data))
    return out


class FunctionDistanceMetric:
    """Distance to a seed set = sum of the distances to each seed (src/playlist.rs:36-59)."""

    def __init__(self, metric: str, vectors: Sequence[np.ndarray], m=None):
        self.metric, self.m = metric, m
        self.state = np.ascontiguousarray(np.stack([_vec(v) for v in vectors]), dtype=np.float32)

    def distance(self, vector) -> float:
        row = pairwise_distances(self.state, _vec(vector)[None, :], self.metric, self.m)[:, 0]
        acc = np.float32(0.0)
        for v in row:  # iter().sum::<f32>() is sequential
            acc = np.float32(acc + v)
        return float(acc)
