"""ctypes binding of the CPU parity oracle (oracle/libbliss_oracle.so).

TEST INFRASTRUCTURE ONLY: import this from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never from the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BLISS_ORACLE_SO: another build of the same source (the --coverage build of `make -C oracle coverage`)
_SO = os.environ.get("BLISS_ORACLE_SO") or os.path.join(_HERE, "libbliss_oracle.so")


def build(force=False):
    if os.environ.get("BLISS_ORACLE_SO"):
        return _SO
    src = os.path.join(_HERE, "bliss_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"])
    return _SO


class SwrPlan(C.Structure):
    """bo_swr_plan_t"""
    _fields_ = [("in_rate", C.c_uint32), ("taps", C.c_int), ("phase_count", C.c_int), ("center", C.c_int),
                ("dst_incr", C.c_uint64), ("src_incr", C.c_uint64), ("factor", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        f32p, f64p, u64p, i32p = (C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_uint64),
                                  C.POINTER(C.c_int32))
        sz = C.c_size_t
        sig = {
            "bo_reflect_pad": (None, [f32p, sz, sz, f32p]),
            "bo_stft_frames": (sz, [sz, sz]),
            "bo_stft": (None, [f32p, sz, sz, sz, f64p]),
            "bo_mean": (C.c_float, [f32p, sz]),
            "bo_std": (C.c_float, [f32p, sz]),
            "bo_number_crossings": (C.c_uint32, [f32p, sz]),
            "bo_geometric_mean": (C.c_float, [f32p, sz]),
            "bo_chroma_filter": (None, [C.c_uint32, sz, C.c_uint32, C.c_double, f64p]),
            "bo_pip_track": (sz, [C.c_uint32, f64p, sz, sz, f64p, f64p]),
            "bo_pitch_tuning": (C.c_double, [f64p, sz, C.c_double, C.c_uint32]),
            "bo_estimate_tuning": (C.c_double, [C.c_uint32, f64p, sz, sz, C.c_double, C.c_uint32]),
            "bo_chroma_stft": (None, [C.c_uint32, f64p, sz, sz, C.c_uint32, C.c_double, f64p]),
            "bo_normalize_feature_sequence": (None, [f64p, sz, sz]),
            "bo_extract_interval_features": (None, [f64p, sz, f64p]),
            "bo_chroma_interval_features": (C.c_int, [f64p, sz, f64p]),
            "bo_chroma_desc_do": (C.c_void_p, [f32p, sz, C.POINTER(sz), f64p]),
            "bo_chroma_get_values": (None, [f64p, sz, f32p]),
            "bo_chroma_get_values_v1": (None, [f64p, sz, f32p]),
            "bo_spectral_desc_new": (C.c_void_p, [C.c_uint32]),
            "bo_spectral_desc_do": (None, [C.c_void_p, f32p]),
            "bo_spectral_desc_get": (None, [C.c_void_p, f32p, f32p, f32p]),
            "bo_spectral_desc_series": (sz, [C.c_void_p, C.POINTER(f32p), C.POINTER(f32p), C.POINTER(f32p)]),
            "bo_spectral_desc_free": (None, [C.c_void_p]),
            "bo_pvoc512_norms": (None, [f32p, f32p, f32p]),
            "bo_bpm_desc_new": (C.c_void_p, [C.c_uint32]),
            "bo_bpm_desc_do": (None, [C.c_void_p, f32p, sz]),
            "bo_bpm_desc_get_value": (C.c_float, [C.c_void_p]),
            "bo_bpm_desc_bpms": (sz, [C.c_void_p, C.POINTER(f32p)]),
            "bo_bpm_desc_series": (sz, [C.c_void_p, C.POINTER(f32p), C.POINTER(f32p)]),
            "bo_bpm_desc_free": (None, [C.c_void_p]),
            "bo_loudness": (None, [f32p, sz, C.c_int, f32p]),
            "bo_zcr": (C.c_float, [f32p, sz]),
            "bo_song_analyze": (C.c_int, [f32p, sz, C.c_uint32, f32p]),
            "bo_song_analyze_timed": (C.c_int, [f32p, sz, C.c_uint32, f32p, f64p]),
            "bo_song_analyze_batch": (None, [f32p, u64p, u64p, C.c_uint32, C.c_uint32, f32p, i32p, C.c_uint32]),
            "bo_euclidean_distance": (C.c_float, [f32p, f32p, sz]),
            "bo_cosine_distance": (C.c_float, [f32p, f32p, sz]),
            "bo_mahalanobis_distance": (C.c_float, [f32p, f32p, f32p, sz]),
            "bo_feature_weights": (None, [C.c_uint32, f32p]),
            "bo_pairwise": (None, [f32p, sz, f32p, sz, sz, C.c_int, f32p, f32p, C.c_uint32]),
            "bo_set_distance": (C.c_float, [f32p, sz, f32p, sz, C.c_int, f32p]),
            "bo_closest_to_songs": (C.c_int, [f32p, sz, f32p, sz, sz, C.c_int, f32p, C.POINTER(C.c_uint32), f32p]),
            "bo_song_to_song": (C.c_int, [f32p, sz, f32p, sz, sz, C.c_int, f32p, C.POINTER(C.c_uint32)]),
            "bo_dedup_playlist": (C.c_long, [f32p, sz, sz, C.c_int, f32p, C.c_float, C.POINTER(C.c_uint8),
                                             C.POINTER(C.c_uint32)]),
            "bo_variance_weight_matrix": (C.c_int, [f32p, sz, sz, f32p]),
            "bo_swr_plan": (C.c_int, [C.c_uint32, C.POINTER(SwrPlan)]),
            "bo_swr_filter": (None, [C.POINTER(SwrPlan), f32p]),
            "bo_swr_out_len": (C.c_uint64, [C.c_uint64, C.c_uint32]),
            "bo_swr_resample": (None, [f32p, C.c_uint64, C.c_uint32, f32p]),
            "bo_decode_to_mono": (C.c_uint64, [C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_uint32, f32p]),
            "bo_white_noise": (None, [C.c_uint32, sz, f32p]),
            "bo_set_fft_double": (None, [C.c_int]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        libc = C.CDLL(None)
        libc.free.argtypes = [C.c_void_p]
        L._free = libc.free
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


# ---- utils.rs ----
def reflect_pad(x, pad):
    x = _f32(x)
    out = np.empty(len(x) + 2 * pad, np.float32)
    lib().bo_reflect_pad(_p(x, C.c_float), len(x), pad, _p(out, C.c_float))
    return out


def stft(signal, win, hop):
    """Returns [bins, frames] float64 like the reference's utils::stft."""
    x = _f32(signal)
    frames = lib().bo_stft_frames(len(x), hop)
    out = np.empty((frames, win // 2 + 1), np.float64)
    lib().bo_stft(_p(x, C.c_float), len(x), win, hop, _p(out, C.c_double))
    return out.T


def mean(x):
    x = _f32(x)
    return float(lib().bo_mean(_p(x, C.c_float), len(x)))


def std(x):
    x = _f32(x)
    return float(lib().bo_std(_p(x, C.c_float), len(x)))


def number_crossings(x):
    x = _f32(x)
    return int(lib().bo_number_crossings(_p(x, C.c_float), len(x)))


def geometric_mean(x):
    x = _f32(x)
    return float(lib().bo_geometric_mean(_p(x, C.c_float), len(x)))


# ---- chroma.rs ----
def chroma_filter(sr, n_fft, n_chroma, tuning):
    out = np.empty((n_chroma, n_fft // 2 + 1), np.float64)
    lib().bo_chroma_filter(sr, n_fft, n_chroma, tuning, _p(out, C.c_double))
    return out


def _spec_fm(spectrum):
    """[bins, frames] -> contiguous [frames, bins]"""
    return _f64(np.asarray(spectrum).T)


def pip_track(sr, spectrum, n_fft):
    s = _spec_fm(spectrum)
    cap = s.shape[0] * (s.shape[1] // 2 + 1) + 1
    pit = np.empty(cap, np.float64)
    mag = np.empty(cap, np.float64)
    n = lib().bo_pip_track(sr, _p(s, C.c_double), s.shape[0], n_fft, _p(pit, C.c_double), _p(mag, C.c_double))
    return pit[:n].copy(), mag[:n].copy()


def pitch_tuning(freqs, resolution, bins_per_octave):
    f = _f64(freqs).copy()
    return float(lib().bo_pitch_tuning(_p(f, C.c_double), len(f), resolution, bins_per_octave))


def estimate_tuning(sr, spectrum, n_fft, resolution, bins_per_octave):
    s = _spec_fm(spectrum)
    return float(lib().bo_estimate_tuning(sr, _p(s, C.c_double), s.shape[0], n_fft, resolution, bins_per_octave))


def chroma_stft(sr, spectrum, n_fft, n_chroma, tuning):
    s = _spec_fm(spectrum).copy()
    out = np.empty((n_chroma, s.shape[0]), np.float64)
    lib().bo_chroma_stft(sr, _p(s, C.c_double), s.shape[0], n_fft, n_chroma, tuning, _p(out, C.c_double))
    return out


def normalize_feature_sequence(feat):
    f = _f64(feat).copy()
    lib().bo_normalize_feature_sequence(_p(f, C.c_double), f.shape[0], f.shape[1])
    return f


def extract_interval_features(chroma):
    c = _f64(chroma)
    out = np.empty((10, c.shape[1]), np.float64)
    lib().bo_extract_interval_features(_p(c, C.c_double), c.shape[1], _p(out, C.c_double))
    return out


def chroma_interval_features(chroma):
    c = _f64(chroma)
    out = np.empty(10, np.float64)
    rc = lib().bo_chroma_interval_features(_p(c, C.c_double), c.shape[1], _p(out, C.c_double))
    if rc != 0:
        raise ValueError("Tried to run the chroma descriptor on an empty array.")
    return out


def chroma_desc(signal):
    """ChromaDesc::do_ on the whole signal -> ([12, frames] chroma, tuning)."""
    x = _f32(signal)
    frames = C.c_size_t()
    tuning = C.c_double()
    ptr = lib().bo_chroma_desc_do(_p(x, C.c_float), len(x), C.byref(frames), C.byref(tuning))
    arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(12, frames.value)).copy()
    lib()._free(ptr)
    return arr, tuning.value


def chroma_get_values(chroma, version=2):
    c = _f64(chroma)
    out = np.empty(13 if version == 2 else 10, np.float32)
    fn = lib().bo_chroma_get_values if version == 2 else lib().bo_chroma_get_values_v1
    fn(_p(c, C.c_double), c.shape[1], _p(out, C.c_float))
    return out


# ---- streaming descriptors ----
class SpectralDesc:
    WINDOW_SIZE = 512
    HOP_SIZE = 128

    def __init__(self, sample_rate=22050):
        self._h = lib().bo_spectral_desc_new(sample_rate)

    def do_(self, chunk):
        c = _f32(chunk)
        assert len(c) >= self.HOP_SIZE
        lib().bo_spectral_desc_do(self._h, _p(c, C.c_float))

    def run(self, x, framing="analyze"):
        """framing 'analyze' = windows(512).step_by(128) (song/mod.rs:458-463); 'chunks_exact' = unit tests"""
        x = _f32(x)
        base = x.ctypes.data
        if framing == "analyze":
            starts = range(0, len(x) - 512 + 1, 128)
        else:
            starts = range(0, (len(x) // 128) * 128, 128)
        fn = lib().bo_spectral_desc_do
        for s in starts:
            fn(self._h, C.cast(base + 4 * s, C.POINTER(C.c_float)))
        return self

    def values(self):
        c = np.empty(2, np.float32)
        r = np.empty(2, np.float32)
        f = np.empty(2, np.float32)
        lib().bo_spectral_desc_get(self._h, _p(c, C.c_float), _p(r, C.c_float), _p(f, C.c_float))
        return c, r, f

    def series(self):
        pc, pr, pf = (C.POINTER(C.c_float)() for _ in range(3))
        n = lib().bo_spectral_desc_series(self._h, C.byref(pc), C.byref(pr), C.byref(pf))
        return tuple(np.ctypeslib.as_array(p, shape=(n,)).copy() for p in (pc, pr, pf))

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:
            lib().bo_spectral_desc_free(self._h)
            self._h = None


def pvoc512_norms(window512):
    w = _f32(window512)
    assert len(w) == 512
    a = np.empty(256, np.float32)
    b = np.empty(257, np.float32)
    lib().bo_pvoc512_norms(_p(w, C.c_float), _p(a, C.c_float), _p(b, C.c_float))
    return a, b


class BPMDesc:
    WINDOW_SIZE = 512
    HOP_SIZE = 256

    def __init__(self, sample_rate=22050):
        self._h = lib().bo_bpm_desc_new(sample_rate)
        if not self._h:
            raise ValueError("error while loading aubio tempo object: creation error")

    def do_(self, chunk):
        c = _f32(chunk)
        lib().bo_bpm_desc_do(self._h, _p(c, C.c_float), len(c))

    def run(self, x, framing="analyze"):
        x = _f32(x)
        base = x.ctypes.data
        fn = lib().bo_bpm_desc_do
        if framing == "analyze":
            for s in range(0, len(x) - 512 + 1, 256):
                fn(self._h, C.cast(base + 4 * s, C.POINTER(C.c_float)), 512)
        else:
            for s in range(0, (len(x) // 256) * 256, 256):
                fn(self._h, C.cast(base + 4 * s, C.POINTER(C.c_float)), 256)
        return self

    def get_value(self):
        return float(lib().bo_bpm_desc_get_value(self._h))

    def bpms(self):
        p = C.POINTER(C.c_float)()
        n = lib().bo_bpm_desc_bpms(self._h, C.byref(p))
        return np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.empty(0, np.float32)

    def series(self):
        po, pt = C.POINTER(C.c_float)(), C.POINTER(C.c_float)()
        n = lib().bo_bpm_desc_series(self._h, C.byref(po), C.byref(pt))
        return (np.ctypeslib.as_array(po, shape=(n,)).copy(), np.ctypeslib.as_array(pt, shape=(n,)).copy())

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:
            lib().bo_bpm_desc_free(self._h)
            self._h = None


def loudness(x, chunks_exact=False):
    x = _f32(x)
    out = np.empty(2, np.float32)
    lib().bo_loudness(_p(x, C.c_float), len(x), int(chunks_exact), _p(out, C.c_float))
    return out


def zcr(x):
    x = _f32(x)
    return float(lib().bo_zcr(_p(x, C.c_float), len(x)))


# ---- Song::analyze ----
class AnalysisError(Exception):
    pass


def song_analyze(x, features_version=2):
    x = _f32(x)
    out = np.empty(23 if features_version == 2 else 20, np.float32)
    rc = lib().bo_song_analyze(_p(x, C.c_float), len(x), features_version, _p(out, C.c_float))
    if rc == 1:
        raise AnalysisError("empty or too short song.")
    if rc != 0:
        raise ValueError(f"oracle error {rc}")
    return out


def song_analyze_timed(x, features_version=2, fast=False):
    """-> {descriptor: seconds} of one analysis on one core (tempo, timbral, zcr, loudness, chroma)"""
    x = _f32(x)
    out = np.empty(23, np.float32)
    secs = np.zeros(5, np.float64)
    rc = (lib_fast() if fast else lib()).bo_song_analyze_timed(_p(x, C.c_float), len(x), features_version, _p(out, C.c_float), _p(secs, C.c_double))
    if rc != 0:
        raise ValueError(f"oracle error {rc}")
    return {k: round(float(v), 4) for k, v in zip(("tempo", "timbral", "zcr", "loudness", "chroma"), secs)}


_fast = None


def lib_fast():
    """BASELINE-ONLY build (oracle/Makefile `native`: -O3 -march=native, vectorisable radix-4 FFT), compiled on the
    machine that times it.  Never the checker: parity always uses lib()."""
    global _fast
    if _fast is None:
        so = os.path.join(_HERE, "_native", "libbliss_oracle_fast.so")
        src = os.path.join(_HERE, "bliss_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "native"])
        L = C.CDLL(so)
        L.bo_song_analyze_batch.restype = None
        L.bo_song_analyze_batch.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_uint32,
                                            C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_uint32]
        L.bo_song_analyze_timed.restype = C.c_int
        L.bo_song_analyze_timed.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.c_uint32, C.POINTER(C.c_float),
                                            C.POINTER(C.c_double)]
        _fast = L
    return _fast


def song_analyze_batch(pcm, offsets, lengths, features_version=2, n_threads=1, fast=False):
    """fast=True: the baseline-only build (timing); its rows differ from the checker's by FFT rounding."""
    pcm = _f32(pcm)
    offsets = np.ascontiguousarray(offsets, np.uint64)
    lengths = np.ascontiguousarray(lengths, np.uint64)
    n = len(offsets)
    d = 23 if features_version == 2 else 20
    out = np.empty((n, d), np.float32)
    status = np.empty(n, np.int32)
    (lib_fast() if fast else lib()).bo_song_analyze_batch(_p(pcm, C.c_float), _p(offsets, C.c_uint64), _p(lengths, C.c_uint64), n,
                                                          features_version, _p(out, C.c_float), _p(status, C.c_int32), n_threads)
    return out, status


# ---- distances ----
def euclidean_distance(a, b):
    a, b = _f32(a), _f32(b)
    return float(lib().bo_euclidean_distance(_p(a, C.c_float), _p(b, C.c_float), len(a)))


def cosine_distance(a, b):
    a, b = _f32(a), _f32(b)
    return float(lib().bo_cosine_distance(_p(a, C.c_float), _p(b, C.c_float), len(a)))


def mahalanobis_distance(a, b, m):
    a, b, m = _f32(a), _f32(b), _f32(m)
    return float(lib().bo_mahalanobis_distance(_p(a, C.c_float), _p(b, C.c_float), _p(m, C.c_float), len(a)))


def feature_weights(features_version=2):
    d = 23 if features_version == 2 else 20
    m = np.empty((d, d), np.float32)
    lib().bo_feature_weights(features_version, _p(m, C.c_float))
    return m


def pairwise(A, B, metric="euclidean", M=None, n_threads=1):
    A, B = _f32(A), _f32(B)
    code = {"euclidean": 0, "cosine": 1, "mahalanobis": 2}[metric]
    out = np.empty((A.shape[0], B.shape[0]), np.float32)
    Mp = _p(_f32(M), C.c_float) if M is not None else None
    lib().bo_pairwise(_p(A, C.c_float), A.shape[0], _p(B, C.c_float), B.shape[0], A.shape[1], code, Mp,
                      _p(out, C.c_float), n_threads)
    return out


# ---- playlist ordering (src/playlist.rs:24-59, 173-221, 256-326, 367-402) ----
_METRIC = {"euclidean": 0, "cosine": 1, "mahalanobis": 2}


def _mp(M):
    return _p(_f32(M), C.c_float) if M is not None else None


def set_distance(seeds, v, metric="euclidean", M=None):
    seeds, v = _f32(np.atleast_2d(seeds)), _f32(v)
    return float(lib().bo_set_distance(_p(seeds, C.c_float), seeds.shape[0], _p(v, C.c_float), v.shape[0],
                                       _METRIC[metric], _mp(M)))


def closest_to_songs(seeds, cand, metric="euclidean", M=None):
    """-> (order, distances); raises on NaN like n32()"""
    seeds, cand = _f32(np.atleast_2d(seeds)), _f32(np.atleast_2d(cand))
    n, d = cand.shape
    order, dist = np.empty(n, np.uint32), np.empty(n, np.float32)
    rc = lib().bo_closest_to_songs(_p(seeds, C.c_float), seeds.shape[0], _p(cand, C.c_float), n, d, _METRIC[metric],
                                   _mp(M), _p(order, C.c_uint32), _p(dist, C.c_float))
    if rc:
        raise ValueError("NaN distance")
    return order, dist


def song_to_song(seeds, cand, metric="euclidean", M=None):
    seeds, cand = _f32(np.atleast_2d(seeds)), _f32(np.atleast_2d(cand))
    n, d = cand.shape
    order = np.empty(n, np.uint32)
    rc = lib().bo_song_to_song(_p(seeds, C.c_float), seeds.shape[0], _p(cand, C.c_float), n, d, _METRIC[metric], _mp(M),
                               _p(order, C.c_uint32))
    if rc:
        raise ValueError("NaN distance")
    return order


def dedup_playlist(songs, threshold=0.05, metric="euclidean", M=None, same_meta=None):
    songs = _f32(np.atleast_2d(songs))
    n, d = songs.shape
    kept = np.empty(max(n, 1), np.uint32)
    sm = None
    if same_meta is not None:
        same_meta = np.ascontiguousarray(same_meta, dtype=np.uint8)
        sm = _p(same_meta, C.c_uint8)
    k = lib().bo_dedup_playlist(_p(songs, C.c_float), n, d, _METRIC[metric], _mp(M), threshold, sm, _p(kept, C.c_uint32))
    if k < 0:
        raise ValueError("NaN distance")
    return kept[:k].copy()


def variance_based_weight_matrix(seeds):
    seeds = _f32(np.atleast_2d(seeds))
    n, d = seeds.shape
    m = np.empty((d, d), np.float32)
    rc = lib().bo_variance_weight_matrix(_p(seeds, C.c_float), n, d, _p(m, C.c_float))
    if rc == 1:
        raise ValueError("seeds must contain more than one element")
    if rc == 2:
        raise ValueError("seed feature vectors must not be empty")
    return m


def white_noise(song_index, n):
    out = np.empty(n, np.float32)
    lib().bo_white_noise(song_index, n, _p(out, C.c_float))
    return out


def set_fft_double(on):
    """tests only: FFTs evaluated in f64 then rounded -- measures sensitivity to FFT rounding"""
    lib().bo_set_fft_double(int(bool(on)))


# ---- the decoder's conversion to mono 22 050 Hz f32 (libswresample as FFmpegDecoder drives it, ffmpeg.rs:36-109) ----
def swr_out_len(n_in, in_rate):
    return int(lib().bo_swr_out_len(int(n_in), int(in_rate)))


def swr_filter(in_rate):
    """(bank[phase_count, taps] float32, plan)"""
    p = SwrPlan()
    if lib().bo_swr_plan(int(in_rate), C.byref(p)):
        raise ValueError("bad rate")
    bank = np.zeros((p.phase_count, p.taps), np.float32)
    lib().bo_swr_filter(C.byref(p), _p(bank, C.c_float))
    return bank, p


def decode_to_mono(samples, in_rate):
    """samples: 1-D mono or [frames, channels]; int16 / int32 (FFmpeg's S16 / S32) / float32 -> mono 22 050 Hz float32"""
    a = np.ascontiguousarray(samples)
    if a.dtype not in (np.int16, np.int32):
        a = np.ascontiguousarray(a, dtype=np.float32)
    fmt = 1 if a.dtype == np.int16 else (2 if a.dtype == np.int32 else 0)
    frames = a.shape[0]
    channels = 1 if a.ndim == 1 else a.shape[1]
    out = np.zeros(max(1, swr_out_len(frames, in_rate)), np.float32)
    n = lib().bo_decode_to_mono(a.ctypes.data if a.size else None, fmt, channels, frames, int(in_rate), _p(out, C.c_float))
    return out[:n]
