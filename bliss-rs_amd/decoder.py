"""Decoder trait mirror (src/song/decoder.rs:34-333).

Decoding itself (ffmpeg / symphonia, src/song/decoder/*.rs) stays on the CPU and is out of scope
(SURVEY.md section 8): implement `decode` for your container/codec; everything from
PreAnalyzedSong -> Song runs on the GPU.  `analyze_paths` batches the decoded songs into GPU launches
instead of the reference's per-core thread pool (src/song/decoder.rs:282-331).
"""
import abc
import wave
from dataclasses import dataclass, field
from typing import Iterable, Iterator, Optional, Tuple, Union

import numpy as np

from .song import (AnalysisOptions, BlissError, DecodingError, FeaturesVersion, SAMPLE_RATE, Song,
                   analyze_batch, analyze_decoded_batch, resampled_len)


@dataclass
class PreAnalyzedSong:
    """src/song/decoder.rs:34-65: decoded samples + tags.  The reference's decoders deliver mono 22 050 Hz f32; here a
    decoder may stop earlier and hand over what the codec produced -- [frames, channels] int16 / int32 / float32 at
    `sample_rate` -- and the conversion FFmpegDecoder does on the CPU (libswresample, src/song/decoder/ffmpeg.rs:36-109)
    runs on the device with the analysis."""
    path: str = ""
    artist: Optional[str] = None
    title: Optional[str] = None
    album: Optional[str] = None
    album_artist: Optional[str] = None
    track_number: Optional[int] = None
    disc_number: Optional[int] = None
    genre: Optional[str] = None
    duration: float = 0.0
    sample_array: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float32))
    sample_rate: int = SAMPLE_RATE

    def _song(self, analysis, version) -> Song:
        return Song(path=self.path, artist=self.artist, title=self.title, album=self.album,
                    album_artist=self.album_artist, track_number=self.track_number, disc_number=self.disc_number,
                    genre=self.genre, duration=self.duration, analysis=analysis, features_version=version)

    def to_song_with_options(self, analysis_options: AnalysisOptions) -> Song:
        """src/song/decoder.rs:85-101"""
        if self.sample_rate == SAMPLE_RATE:
            analysis = Song.analyze_with_options(self.sample_array, analysis_options)
        else:
            analysis = Song.analyze_decoded(self.sample_array, self.sample_rate, analysis_options)
        return self._song(analysis, FeaturesVersion(analysis_options.features_version))


class Decoder(abc.ABC):
    """src/song/decoder.rs:115-333"""

    @classmethod
    @abc.abstractmethod
    def decode(cls, path: str) -> PreAnalyzedSong:
        """Required method (src/song/decoder.rs:129): file -> mono 22 050 Hz f32 samples."""

    @classmethod
    def song_from_path(cls, path: str) -> Song:
        return cls.song_from_path_with_options(path, AnalysisOptions())

    @classmethod
    def song_from_path_with_options(cls, path: str, analysis_options: AnalysisOptions) -> Song:
        return cls.decode(path).to_song_with_options(analysis_options)

    @classmethod
    def analyze_paths(cls, paths: Iterable[str]) -> Iterator[Tuple[str, Union[Song, BlissError]]]:
        return cls.analyze_paths_with_options(paths, AnalysisOptions())

    @classmethod
    def analyze_paths_with_options(cls, paths: Iterable[str], analysis_options: AnalysisOptions,
                                   batch_songs: int = 256) -> Iterator[Tuple[str, Union[Song, BlissError]]]:
        """Yields (path, Song | BlissError); a bad file never aborts the run (src/song/decoder.rs:313-325)."""
        version = FeaturesVersion(analysis_options.features_version)
        pending = []

        def flush():
            if all(p.sample_rate == SAMPLE_RATE for p in pending):
                results = analyze_batch([p.sample_array for p in pending], analysis_options)
            else:
                results = analyze_decoded_batch([p.sample_array for p in pending], [p.sample_rate for p in pending],
                                                analysis_options)
            for pre, res in zip(pending, results):
                yield pre.path, (res if isinstance(res, BlissError) else pre._song(res, version))
            pending.clear()

        for path in paths:
            try:
                pending.append(cls.decode(path))
            except BlissError as e:
                yield path, e
                continue
            except Exception as e:  # decoder failures are reported per file
                yield path, DecodingError(str(e))
                continue
            if len(pending) >= batch_songs:
                yield from flush()
        if pending:
            yield from flush()


class RawPcmDecoder(Decoder):
    """Decoder for already-decoded PCM: `.npy` (float32, int16 or int32; 1-D mono or [frames, channels]; 22 050 Hz) and
    16-bit `.wav` (any channel count, ANY sample rate).  Samples are passed on as the file holds them: the widening
    (sample / 32768, FFmpeg's conversion), the mono downmix and the resampling to 22 050 Hz (libswresample's default
    resampler, as FFmpegDecoder uses it: src/song/decoder/ffmpeg.rs:36-109) run on the device."""

    @classmethod
    def decode(cls, path: str) -> PreAnalyzedSong:
        rate = SAMPLE_RATE
        try:
            if path.endswith(".npy"):
                a = np.load(path)
                if a.ndim not in (1, 2) or (a.ndim == 2 and not 1 <= a.shape[1] <= 8):
                    raise DecodingError("expected [frames] or [frames, channels <= 8] samples")
                samples = a if a.dtype in (np.int16, np.int32) else a.astype(np.float32)
            else:
                with wave.open(path, "rb") as w:
                    if w.getsampwidth() != 2 or not 1 <= w.getnchannels() <= 8:
                        raise DecodingError("only s16 wav is supported by RawPcmDecoder")
                    rate = w.getframerate()
                    samples = np.frombuffer(w.readframes(w.getnframes()), "<i2")
                    if w.getnchannels() > 1:
                        samples = samples.reshape(-1, w.getnchannels())
        except BlissError:
            raise
        except Exception as e:
            raise DecodingError(f"while opening format for file '{path}': {e}")
        return PreAnalyzedSong(path=path, sample_array=samples, sample_rate=rate,
                               duration=resampled_len(samples.shape[0], rate) / SAMPLE_RATE)
