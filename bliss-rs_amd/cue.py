"""CUE tracks as slices of one decoded buffer -- `BlissCueFile::get_songs` (src/cue.rs:205-246).

Parsing the sheet (the `rcue` crate in the reference) and decoding the audio file stay with the host; what the reference
does next is what this module does on the device: the file is converted ONCE to mono 22 050 Hz (FFmpegDecoder's conversion,
`Context.pcm_decode`), every track is the sample range between its INDEX and the next track's, and each range goes through
`Song::analyze_with_options`.  The ranges are (offset, length) pairs of the device buffer: nothing is copied per track.
"""
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np

from .song import (Analysis, AnalysisError, AnalysisOptions, BlissError, FeaturesVersion, SAMPLE_RATE)


def duration_as_secs_f32(secs: int, nanos: int) -> np.float32:
    """`std::time::Duration::as_secs_f32`: `(secs as f32) + (nanos as f32) / 1e9_f32` -- two roundings and a rounded quotient,
    NOT the exact sum rounded once (the two differ in the last bit for 72 of the 360 000 mm:ss:ff values of an 80-minute disc,
    and the sample index src/cue.rs:214-215,232 derives from it then moves by one)."""
    return np.float32(np.float32(int(secs)) + np.float32(int(nanos)) / np.float32(1e9))


def cue_index_duration(mm: int, ss: int, ff: int) -> Tuple[int, int]:
    """An INDEX time `mm:ss:ff` (75 frames per second) as the (secs, nanos) of the `Duration` the sheet parser hands to
    `BlissCueFile` (rcue 0.1.3, a dependency that is not under /root/reference: integer nanoseconds, truncated -- rounding them
    instead gives the same f32 for 73 of the 75 frame values)."""
    return int(mm) * 60 + int(ss), int(ff) * 1_000_000_000 // 75


def _index_secs_f32(index) -> np.float32:
    if isinstance(index, (tuple, list)) and len(index) == 2:
        return duration_as_secs_f32(*index)
    if isinstance(index, (tuple, list)) and len(index) == 3:
        return duration_as_secs_f32(*cue_index_duration(*index))
    return np.float32(index)  # the caller already holds `Duration::as_secs_f32()` (what the Rust binding passes)


def cue_track_bounds(index: Sequence, n_samples: int) -> List[Tuple[int, int]]:
    """(start, end) sample ranges of the tracks of one FILE entry: `(index.as_secs_f32() * SAMPLE_RATE as f32) as usize`
    for each track's first INDEX, the last track running to the end of the decoded file (src/cue.rs:212-236).
    Every index is a `Duration` given as (secs, nanos), or (mm, ss, ff) of the sheet, or -- only when the caller has evaluated
    `as_secs_f32()` itself -- that f32 value; a float computed as mm * 60 + ss + ff / 75 is NOT the same number."""
    starts = [int(_index_secs_f32(s) * np.float32(SAMPLE_RATE)) for s in index]
    return list(zip(starts, starts[1:] + [int(n_samples)]))


def analyze_cue_tracks(ctx, samples, sample_rate: int, index_seconds: Sequence,
                       analysis_options: Optional[AnalysisOptions] = None) -> List[Union[Analysis, BlissError]]:
    """samples: what the decoder delivered for the CUE sheet's audio file (numpy, 1-D mono or [frames, channels]; int16 / int32 /
    float32) at `sample_rate`; index_seconds: the tracks' INDEX 01 times as (secs, nanos) or (mm, ss, ff) -- see cue_track_bounds.  One Analysis (or
    the BlissError the reference would put in that slot) per track, in order."""
    import torch

    options = analysis_options or AnalysisOptions()
    version = FeaturesVersion(options.features_version)
    a = np.ascontiguousarray(samples)
    if a.dtype not in (np.int16, np.int32):
        a = a.astype(np.float32, copy=False)
    pcm = ctx.pcm_decode(torch.from_numpy(np.array(a)).to(f"cuda:{ctx.device}"), int(sample_rate))
    bounds = cue_track_bounds(index_seconds, pcm.numel())
    for s, e in bounds:
        if not 0 <= s <= e <= pcm.numel():
            raise BlissError("CUE index beyond the end of the audio file")
    out, status = ctx.analyze(pcm, [s for s, _ in bounds], [e - s for s, e in bounds], int(version))
    ctx.synchronize()
    rows, st = out.cpu().numpy(), status.cpu().numpy()
    return [Analysis(rows[i], version) if st[i] == 0 else AnalysisError("empty or too short song.") for i in range(len(bounds))]
