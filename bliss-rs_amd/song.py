"""Song / Analysis / FeaturesVersion -- mirror of src/song/mod.rs and src/lib.rs (analysis part)."""
import ctypes as C
import enum
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import _ffi

SAMPLE_RATE = 22050  # src/lib.rs:140
CHANNELS = 1         # src/lib.rs:137


class BlissError(Exception):
    """Umbrella error type (src/lib.rs:236-252)."""
    prefix = "error"

    def __init__(self, message):
        super().__init__(message)
        self.message = message

    def __str__(self):
        return f"{self.prefix} - {self.message}"

    def __eq__(self, other):
        return type(self) is type(other) and self.message == other.message

    __hash__ = Exception.__hash__


class DecodingError(BlissError):
    prefix = "error happened while decoding file"


class AnalysisError(BlissError):
    prefix = "error happened while analyzing file"


class ProviderError(BlissError):
    prefix = "error happened with the music library provider"


class FeaturesVersion(enum.IntEnum):
    """src/lib.rs:151-187"""
    Version2 = 2
    Version1 = 1

    def feature_count(self) -> int:
        return 23 if self is FeaturesVersion.Version2 else 20

    def feature_weights(self) -> np.ndarray:
        d = self.feature_count()
        m = np.zeros((d, d), np.float32)
        _ffi.check(_ffi.lib().blissgpu_feature_weights(int(self), m.ctypes.data))
        return m

    def distance_metric(self):
        from .playlist import mahalanobis_distance_builder
        return mahalanobis_distance_builder(self.feature_weights())

    @classmethod
    def try_from(cls, value: int) -> "FeaturesVersion":
        if value in (1, 2):
            return cls(value)
        raise ProviderError(f"This features' version ({value}) does not exist")


FeaturesVersion.LATEST = FeaturesVersion.Version2
NUMBER_FEATURES = FeaturesVersion.LATEST.feature_count()  # src/song/mod.rs:222


class AnalysisIndex(enum.IntEnum):
    """src/song/mod.rs:102-156 (Version2 layout)"""
    Tempo = 0
    Zcr = 1
    MeanSpectralCentroid = 2
    StdDeviationSpectralCentroid = 3
    MeanSpectralRolloff = 4
    StdDeviationSpectralRolloff = 5
    MeanSpectralFlatness = 6
    StdDeviationSpectralFlatness = 7
    MeanLoudness = 8
    StdDeviationLoudness = 9
    Chroma1 = 10
    Chroma2 = 11
    Chroma3 = 12
    Chroma4 = 13
    Chroma5 = 14
    Chroma6 = 15
    Chroma7 = 16
    Chroma8 = 17
    Chroma9 = 18
    Chroma10 = 19
    Chroma11 = 20
    Chroma12 = 21
    Chroma13 = 22


AnalysisIndex.FEATURES_VERSION = FeaturesVersion.LATEST
AnalysisIndexv1 = enum.IntEnum("AnalysisIndexv1", [(m.name, m.value) for m in list(AnalysisIndex)[:20]])
AnalysisIndexv1.FEATURES_VERSION = FeaturesVersion.Version1


@dataclass(frozen=True)
class AnalysisOptions:
    """src/song/mod.rs:248-269.  number_cores is kept for interface parity; the batch is scheduled on
    the GPU, not on a host thread pool."""
    features_version: FeaturesVersion = FeaturesVersion.LATEST
    number_cores: int = field(default_factory=lambda: os.cpu_count() or 1)


class Analysis:
    """src/song/mod.rs:238-371"""

    def __init__(self, analysis: Sequence[float], features_version: FeaturesVersion = FeaturesVersion.LATEST):
        features_version = FeaturesVersion(features_version)
        arr = np.asarray(analysis, dtype=np.float32).copy()
        if arr.ndim != 1 or arr.shape[0] != features_version.feature_count():
            raise ProviderError(
                f"Feature count {arr.size} does not match the expected version feature count "
                f"{features_version.feature_count()}")
        self.internal_analysis = arr
        self.features_version = features_version

    @classmethod
    def new(cls, analysis, features_version):
        return cls(analysis, features_version)

    def as_arr1(self) -> np.ndarray:
        return self.internal_analysis.copy()

    def as_vec(self) -> List[float]:
        return [float(x) for x in self.internal_analysis]

    def __getitem__(self, index):
        want = AnalysisIndex.FEATURES_VERSION if isinstance(index, AnalysisIndex) else AnalysisIndexv1.FEATURES_VERSION
        if self.features_version != want:
            raise RuntimeError("Tried to index features with incompatible indexes")  # the reference panics
        return float(self.internal_analysis[int(index)])

    def __eq__(self, other):
        return (isinstance(other, Analysis) and self.features_version == other.features_version
                and np.array_equal(self.internal_analysis, other.internal_analysis))

    def __repr__(self):
        return f"Analysis (Version {int(self.features_version)}) /* {self.as_vec()} */"

    def distance(self, other: "Analysis") -> float:
        """Default distance for the FeaturesVersion (src/song/mod.rs:364-370)."""
        if self.features_version != other.features_version:
            raise RuntimeError("Mismatched features version between two songs or analysis")  # panic in the reference
        return self.features_version.distance_metric()(self.internal_analysis, other.internal_analysis)


def _as_pcm(sample_array) -> np.ndarray:
    """Decoder output as the library takes it: int16 stays int16 (widened on the device with sample / 32768, FFmpeg's
    s16 -> flt conversion), everything else becomes float32; 1-D = mono, 2-D = [frames, channels] interleaved
    (downmixed on the device like the reference's decoders, src/song/decoder/symphonia.rs:266-300)."""
    a = np.asarray(sample_array)
    if a.dtype not in (np.int16, np.int32):  # int32 = FFmpeg's S32 (24-bit streams left-justified): (float)s / 2^31 on the device
        a = a.astype(np.float32, copy=False)
    if a.ndim == 2 and a.shape[1] == 1:
        a = a[:, 0]
    if a.ndim not in (1, 2):
        raise ProviderError("sample arrays must be 1-D (mono) or 2-D [frames, channels]")
    return np.ascontiguousarray(a)


def _fmt_of(a: np.ndarray) -> int:
    return _ffi.SAMPLE_S16 if a.dtype == np.int16 else (_ffi.SAMPLE_S32 if a.dtype == np.int32 else _ffi.SAMPLE_F32)


def resampled_len(frames: int, sample_rate: int) -> int:
    """Samples at 22 050 Hz that `frames` frames at `sample_rate` become (libswresample's count; device-free)."""
    return int(_ffi.lib().blissgpu_resampled_len(int(frames), int(sample_rate)))


def _results(out, status, version):
    results = []
    for i in range(len(status)):
        if status[i] == _ffi.SONG_OK:
            results.append(Analysis(out[i], version))
        elif status[i] == _ffi.SONG_TOO_SHORT:
            results.append(AnalysisError("empty or too short song."))
        else:
            results.append(AnalysisError(f"analysis failed with status {int(status[i])}"))
    return results


def analyze_decoded_batch(sample_arrays: Sequence[np.ndarray], sample_rates, options: Optional[AnalysisOptions] = None):
    """Decoder output at ANY sample rate -> analysis: what FFmpegDecoder::decode + Song::analyze do together
    (src/song/decoder/ffmpeg.rs:36-109, src/song/decoder.rs:85-101), with libswresample's conversion to mono 22 050 Hz
    done on the device (bit for bit at 44 100 Hz, the rate the reference pins).  sample_rates: one rate for all songs or one per song."""
    options = options or AnalysisOptions()
    version = FeaturesVersion(options.features_version)
    arrays = [_as_pcm(a) for a in sample_arrays]
    n = len(arrays)
    if n == 0:
        return []
    rates = [int(sample_rates)] * n if np.isscalar(sample_rates) else [int(r) for r in sample_rates]
    if len(rates) != n:
        raise ProviderError("one sample rate per song")
    d = version.feature_count()
    out = np.empty((n, d), np.float32)
    status = np.zeros(n, np.int32)
    L = _ffi.lib()
    keep = [a if a.size else np.zeros(1, a.dtype) for a in arrays]
    if n == 1:
        a = arrays[0]
        st = C.c_int32(0)
        _ffi.check(L.blissgpu_analyze_decoded(keep[0].ctypes.data, _fmt_of(a), 1 if a.ndim == 1 else a.shape[1], a.shape[0],
                                              rates[0], int(version), out.ctypes.data, C.byref(st)))
        status[0] = st.value
    else:
        songs = (_ffi.DecodedSong * n)()
        for i, a in enumerate(arrays):
            songs[i] = _ffi.DecodedSong(keep[i].ctypes.data, a.shape[0], rates[i], 1 if a.ndim == 1 else a.shape[1], _fmt_of(a))
        _ffi.check(L.blissgpu_analyze_batch_decoded(songs, n, int(version), out.ctypes.data,
                                                    status.ctypes.data_as(C.POINTER(C.c_int32))))
    return _results(out, status, version)


def analyze_batch(sample_arrays: Sequence[np.ndarray], options: Optional[AnalysisOptions] = None):
    """Bulk Song::analyze_with_options: one GPU batch, per-song result (Analysis or BlissError), like the
    (path, BlissResult<Song>) pairs of analyze_paths_with_options (src/song/decoder.rs:278-332).  A single song goes
    through blissgpu_analyze[_interleaved], whose concurrent callers (threads each analysing one song, as the
    reference's worker pool does, src/song/decoder.rs:299-329) are coalesced into one device batch."""
    options = options or AnalysisOptions()
    version = FeaturesVersion(options.features_version)
    arrays = [_as_pcm(a) for a in sample_arrays]
    n = len(arrays)
    if n == 0:
        return []
    d = version.feature_count()
    out = np.empty((n, d), np.float32)
    status = np.empty(n, np.int32)
    L = _ffi.lib()
    # one device batch per (sample format, channel count) class -- normally there is exactly one
    classes = {}
    for i, a in enumerate(arrays):
        classes.setdefault((_fmt_of(a), 1 if a.ndim == 1 else a.shape[1]), []).append(i)
    for (fmt, channels), idx in classes.items():
        if len(idx) == 1:
            a = arrays[idx[0]]
            buf = a if a.size else np.zeros(1, a.dtype)
            row = np.empty(d, np.float32)
            st = C.c_int32(0)
            _ffi.check(L.blissgpu_analyze_interleaved(buf.ctypes.data, fmt, channels, a.shape[0], int(version),
                                                      row.ctypes.data, C.byref(st)))
            out[idx[0]] = row
            status[idx[0]] = st.value
            continue
        lengths = np.array([arrays[i].shape[0] for i in idx], np.uint64)  # frames
        offsets = np.zeros(len(idx), np.uint64)
        offsets[1:] = np.cumsum(lengths)[:-1]
        pcm = np.concatenate([arrays[i].reshape(-1) for i in idx])
        if pcm.size == 0:
            pcm = np.zeros(1, arrays[idx[0]].dtype)
        res = np.empty((len(idx), d), np.float32)
        st = np.empty(len(idx), np.int32)
        _ffi.check(L.blissgpu_analyze_batch_interleaved(
            pcm.ctypes.data, fmt, channels, offsets.ctypes.data_as(C.POINTER(C.c_uint64)),
            lengths.ctypes.data_as(C.POINTER(C.c_uint64)), len(idx), int(version), res.ctypes.data,
            st.ctypes.data_as(C.POINTER(C.c_int32))))
        out[idx] = res
        status[idx] = st
    return _results(out, status, version)


@dataclass
class Song:
    """src/song/mod.rs:45-76 (metadata fields kept, filled by decoders)."""
    path: str = ""
    artist: Optional[str] = None
    title: Optional[str] = None
    album: Optional[str] = None
    album_artist: Optional[str] = None
    track_number: Optional[int] = None
    disc_number: Optional[int] = None
    genre: Optional[str] = None
    duration: float = 0.0
    analysis: Optional[Analysis] = None
    features_version: FeaturesVersion = FeaturesVersion.LATEST
    cue_info: Optional[object] = None

    @staticmethod
    def analyze(sample_array) -> Analysis:
        """src/song/mod.rs:403-405"""
        return Song.analyze_with_options(sample_array, AnalysisOptions())

    @staticmethod
    def analyze_with_options(sample_array, analysis_options: AnalysisOptions) -> Analysis:
        """src/song/mod.rs:413-508.  Raises AnalysisError("empty or too short song.") for len < 8192."""
        res = analyze_batch([sample_array], analysis_options)[0]
        if isinstance(res, BlissError):
            raise res
        return res

    @staticmethod
    def analyze_decoded(sample_array, sample_rate: int, analysis_options: Optional[AnalysisOptions] = None) -> Analysis:
        """Decoder output at `sample_rate` Hz (1-D mono or [frames, channels]; float32 / int16 / int32): the conversion
        FFmpegDecoder does on the CPU (src/song/decoder/ffmpeg.rs:36-109) runs on the device, then Song::analyze."""
        res = analyze_decoded_batch([sample_array], sample_rate, analysis_options)[0]
        if isinstance(res, BlissError):
            raise res
        return res

    def distance(self, other: "Song") -> float:
        """src/song/mod.rs:519-521"""
        return self.analysis.distance(other.analysis)
