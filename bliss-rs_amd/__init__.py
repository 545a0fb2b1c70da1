"""bliss_rs_amd -- MI355X-native implementation of bliss-rs's per-song analysis hot path
(`Song::analyze`) and feature-vector distances, behind the reference's own interface.

Host-side mirror of the reference surface (names, argument meaning and error behaviour follow
bliss-rs / `bliss-audio` 0.13.0):

    Song.analyze / Song.analyze_with_options      src/song/mod.rs:403-508
    Analysis, AnalysisIndex, FeaturesVersion      src/song/mod.rs:102-371, src/lib.rs:151-187
    Decoder.{decode, song_from_path, analyze_paths}  src/song/decoder.rs:115-333
    cue.{cue_track_bounds, analyze_cue_tracks}    BlissCueFile::get_songs, src/cue.rs:205-246 (sheet parsing stays with the host)
    euclidean / cosine / mahalanobis distance, closest_to_songs, song_to_song, dedup, ...   src/playlist.rs
    library.{load_feature_matrix, load_songs, store_song}   feature table of src/library.rs:500-531, 1355-1372, 1560-1630

All compute goes through the C ABI of include/blissgpu.h (hand-written HIP kernels for gfx950);
there is no CPU fallback.
"""
from ._ffi import BlissGpuError, LIB_PATH  # noqa: F401
from .song import (  # noqa: F401
    SAMPLE_RATE, CHANNELS, NUMBER_FEATURES, Analysis, AnalysisError, AnalysisIndex, AnalysisIndexv1,
    AnalysisOptions, BlissError, DecodingError, FeaturesVersion, ProviderError, Song, analyze_batch, analyze_decoded_batch,
    resampled_len)
from .decoder import Decoder, PreAnalyzedSong, RawPcmDecoder  # noqa: F401
from . import playlist  # noqa: F401
from . import cue  # noqa: F401
from . import library  # noqa: F401
from .device import Context, Node  # noqa: F401

__version__ = "0.3.0"
