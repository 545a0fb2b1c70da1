"""Distance metrics and playlist ordering -- mirror of src/playlist.rs.

    euclidean / cosine / mahalanobis distance, builders          :65-79, 129-142
    DistanceMetricBuilder / FunctionDistanceMetric               :24-59
    variance_based_weight_matrix                                 :173-221
    closest_to_songs, song_to_song                               :256-326
    dedup_playlist, dedup_playlist_custom_distance               :343-402
    closest_album_to_group                                       :424-485

A metric builder is one of the strings "euclidean" / "cosine", a `MahalanobisBuilder` (or the pair
("mahalanobis", M)), i.e. the metrics the device implements; arbitrary Python callables are not accepted
because the distances are evaluated by the HIP kernels (there is no CPU path).  Songs are anything with an
`.analysis` (Analysis) -- `Song` or a wrapper holding one in `.bliss_song`, like the reference's
`AsRef<Song>`."""
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


# ---------------------------------------------------------------------------------------------------
# playlist ordering (src/playlist.rs:173-485)
# ---------------------------------------------------------------------------------------------------
class MahalanobisBuilder:
    """mahalanobis_distance_builder(m) as a metric builder (src/playlist.rs:129-131)."""

    def __init__(self, m):
        self.m = np.ascontiguousarray(m, dtype=np.float32).copy()


def _metric_of(builder):
    """-> (metric name, M or None)"""
    if isinstance(builder, MahalanobisBuilder):
        return "mahalanobis", builder.m
    if isinstance(builder, tuple) and len(builder) == 2 and builder[0] == "mahalanobis":
        return "mahalanobis", np.ascontiguousarray(builder[1], dtype=np.float32)
    if builder in (euclidean_distance, "euclidean"):
        return "euclidean", None
    if builder in (cosine_distance, "cosine"):
        return "cosine", None
    raise TypeError("metric builder must be euclidean_distance / cosine_distance / MahalanobisBuilder(m): distances "
                    "are evaluated on the GPU, arbitrary callables are not supported")


def _song_of(s):
    return getattr(s, "bliss_song", s)  # AsRef<Song>


def _matrix(songs) -> np.ndarray:
    rows = [np.asarray(_song_of(s).analysis.as_vec(), dtype=np.float32) for s in songs]
    if not rows:
        return np.zeros((0, 1), np.float32)
    return np.ascontiguousarray(np.stack(rows), dtype=np.float32)


def _nan_to_panic(e: "_ffi.BlissGpuError"):
    if e.code == _ffi.ERR_NAN:
        raise ValueError("NaN distance (noisy_float::n32 / argmin().unwrap() panic in the reference)") from e
    raise e


def set_distances(seeds, candidates, metric="euclidean", m=None) -> np.ndarray:
    """FunctionDistanceMetric::distance (src/playlist.rs:52-58) for every row of `candidates`."""
    S = np.ascontiguousarray(np.atleast_2d(seeds), dtype=np.float32)
    X = np.ascontiguousarray(np.atleast_2d(candidates), dtype=np.float32)
    out = np.empty(X.shape[0], np.float32)
    mp = None
    if m is not None:
        m = np.ascontiguousarray(m, dtype=np.float32)
        mp = m.ctypes.data
    _ffi.check(_ffi.lib().blissgpu_set_distance(S.ctypes.data, S.shape[0], X.ctypes.data, X.shape[0], X.shape[1],
                                                _METRICS[metric], mp, out.ctypes.data))
    return out


def closest_to_songs_order(seeds, candidates, metric="euclidean", m=None):
    """Index form of closest_to_songs: -> (order u32[n], distances f32[n])."""
    S = np.ascontiguousarray(np.atleast_2d(seeds), dtype=np.float32)
    X = np.ascontiguousarray(np.atleast_2d(candidates), dtype=np.float32)
    order, dist = np.empty(X.shape[0], np.uint32), np.empty(X.shape[0], np.float32)
    mp = None
    if m is not None:
        m = np.ascontiguousarray(m, dtype=np.float32)
        mp = m.ctypes.data
    try:
        _ffi.check(_ffi.lib().blissgpu_closest_to_songs(S.ctypes.data, S.shape[0], X.ctypes.data, X.shape[0], X.shape[1],
                                                        _METRICS[metric], mp, order.ctypes.data, dist.ctypes.data))
    except _ffi.BlissGpuError as e:
        _nan_to_panic(e)
    return order, dist


def song_to_song_order(seeds, candidates, metric="euclidean", m=None) -> np.ndarray:
    """Index form of song_to_song."""
    S = np.ascontiguousarray(np.atleast_2d(seeds), dtype=np.float32)
    X = np.ascontiguousarray(np.atleast_2d(candidates), dtype=np.float32)
    order = np.empty(X.shape[0], np.uint32)
    mp = None
    if m is not None:
        m = np.ascontiguousarray(m, dtype=np.float32)
        mp = m.ctypes.data
    try:
        _ffi.check(_ffi.lib().blissgpu_song_to_song(S.ctypes.data, S.shape[0], X.ctypes.data, X.shape[0], X.shape[1],
                                                    _METRICS[metric], mp, order.ctypes.data))
    except _ffi.BlissGpuError as e:
        _nan_to_panic(e)
    return order


def closest_to_songs(initial_songs, candidate_songs, metric_builder=euclidean_distance):
    """src/playlist.rs:256-270: candidates sorted (stably) by their distance to the set of initial songs."""
    candidate_songs = list(candidate_songs)
    if not candidate_songs:
        return []
    metric, m = _metric_of(metric_builder)
    order, _ = closest_to_songs_order(_matrix(initial_songs), _matrix(candidate_songs), metric, m)
    return [candidate_songs[i] for i in order]


def song_to_song(initial_songs, candidate_songs, metric_builder=euclidean_distance):
    """src/playlist.rs:272-326: each song is followed by the remaining song closest to it."""
    candidate_songs = list(candidate_songs)
    if not candidate_songs:
        return []
    metric, m = _metric_of(metric_builder)
    order = song_to_song_order(_matrix(initial_songs), _matrix(candidate_songs), metric, m)
    return [candidate_songs[i] for i in order]


def _same_title_artist(a, b) -> bool:
    a, b = _song_of(a), _song_of(b)
    return (a.title is not None and b.title is not None and a.artist is not None and b.artist is not None
            and a.title == b.title and a.artist == b.artist)


def dedup_playlist_custom_distance(playlist, distance_threshold=None, metric_builder=euclidean_distance, window=64):
    """src/playlist.rs:367-402: a song absorbs the songs that follow it while they are closer than the threshold
    (default 0.05) or carry the same non-empty title and artist.  Distances from the current song to the next
    `window` songs are evaluated in one device call."""
    playlist = list(playlist)
    thr = np.float32(0.05 if distance_threshold is None else distance_threshold)
    metric, m = _metric_of(metric_builder)
    X = _matrix(playlist)
    out, i, n = [], 0, len(playlist)
    while i < n:
        j = i + 1
        while j < n:
            hi = min(n, j + window)
            try:
                dist = set_distances(X[i:i + 1], X[j:hi], metric, m)
            except _ffi.BlissGpuError as e:  # pragma: no cover
                _nan_to_panic(e)
            stop = None
            for k in range(j, hi):
                dk = dist[k - j]
                if np.isnan(dk):
                    raise ValueError("NaN distance (noisy_float::n32 panic in the reference)")
                if not (dk < thr or _same_title_artist(playlist[i], playlist[k])):
                    stop = k
                    break
            if stop is not None:
                j = stop
                break
            j = hi
        out.append(playlist[i])
        i = j
    return out


def dedup_playlist(playlist, distance_threshold=None):
    """src/playlist.rs:343-348"""
    return dedup_playlist_custom_distance(playlist, distance_threshold, euclidean_distance)


def variance_based_weight_matrix(seeds) -> np.ndarray:
    """src/playlist.rs:173-221: diagonal Mahalanobis weights ~ 1 / (variance + 1e-6), normalised to sum to d.
    O(seeds x d) host arithmetic in the reference's f32 evaluation order."""
    from .song import ProviderError

    seeds = [np.asarray(s, dtype=np.float32).reshape(-1) for s in seeds]
    if len(seeds) < 2:
        raise ProviderError("seeds must contain more than one element")
    n = seeds[0].shape[0]
    if n == 0:
        raise ProviderError("seed feature vectors must not be empty")
    if any(s.shape[0] != n for s in seeds):
        raise ProviderError("all seed feature vectors must have the same length")
    ns = np.float32(len(seeds))
    mean = np.zeros(n, np.float32)
    for s in seeds:
        mean = mean + s
    mean = mean / ns
    var = np.zeros(n, np.float32)
    for s in seeds:
        diff = s - mean
        var = var + diff * diff
    var = var / ns
    w = np.float32(1.0) / (var + np.float32(1e-6))
    # ndarray's sum(): unrolled_fold, 8 partial sums combined (p0+p4)+(p1+p5)+(p2+p6)+(p3+p7), then the tail
    p = np.zeros(8, np.float32)
    k = 0
    while k + 8 <= n:
        p = p + w[k:k + 8]
        k += 8
    total = np.float32(0.0)
    for u in range(4):
        total = np.float32(total + np.float32(p[u] + p[u + 4]))
    for x in w[k:]:
        total = np.float32(total + x)
    w = w * np.float32(np.float32(n) / total)
    return np.diag(w).astype(np.float32)


def closest_album_to_group(group, pool):
    """src/playlist.rs:424-485: albums of `pool` ordered by the euclidean distance of their mean analysis to the
    group's mean analysis (distances on the device), each album ordered by (disc, track) number."""
    from .song import ProviderError

    group, pool = list(group), list(pool)
    pool = [s for s in pool if not any(_song_of(g) == _song_of(s) for g in group)]
    albums = {}
    for s in pool:
        album = _song_of(s).album
        if album is not None:
            albums.setdefault(album, []).append(np.asarray(_song_of(s).analysis.as_vec(), dtype=np.float32))

    def mean_axis0(rows):  # ndarray mean_axis: sequential f32 row sum / n
        if not rows:
            raise ProviderError("Mean of empty slice")
        acc = np.zeros_like(rows[0])
        for r in rows:
            acc = acc + r
        return acc / np.float32(len(rows))

    first = mean_axis0([np.asarray(_song_of(s).analysis.as_vec(), dtype=np.float32) for s in group])
    names = list(albums.keys())
    playlist = list(group)
    if names:
        means = np.stack([mean_axis0(albums[a]) for a in names])
        order, _ = closest_to_songs_order(first[None, :], means, "euclidean")  # sort_by_key is stable as well
        for a in (names[i] for i in order):
            al = [s for s in pool if _song_of(s).album == a]

            def key(s):
                s = _song_of(s)
                d, t = s.disc_number, s.track_number
                return ((0, 0) if d is None else (1, d), (0, 0) if t is None else (1, t))  # Option: None < Some

            al.sort(key=key)
            playlist.extend(al)
    return playlist
