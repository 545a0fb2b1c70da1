import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as O  # oracle/oracle.py -- the CPU checker (test infrastructure)

    O.build()
    return O


@pytest.fixture(scope="session")
def literals():
    with open(os.path.join(GOLDEN, "reference_literals.json")) as f:
        return json.load(f)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def golden_pcm():
    """data/s16_mono_22_5kHz.flac decoded exactly as ffmpeg does (s16 / 32768), Adler-32 0x5e01930b."""
    s16 = load_golden("s16_mono_22_5kHz.pcm_s16.npy")
    return (s16.astype(np.float32) / np.float32(32768.0)).astype(np.float32)


@pytest.fixture(scope="session")
def piano_pcm():
    return load_golden("librosa-decoded.npy")


# ---- the reference's audio data files that are NOT at 22 050 Hz (inputs of the resampler row; tests/golden/make_fixtures.py)
_DECODED = {}


def decoded_audio(name):
    """(samples, sample_rate): what FFmpeg's decoder hands to the resampler -- int16 for 16-bit files, int32 with 24-bit
    samples left-justified (AV_SAMPLE_FMT_S32) for the 24-bit ones; 1-D mono or [frames, channels]."""
    if name not in _DECODED:
        path = os.path.join(GOLDEN, name)
        if name.endswith(".wav"):
            import wave

            with wave.open(path, "rb") as w:
                assert w.getsampwidth() == 2
                a = np.frombuffer(w.readframes(w.getnframes()), "<i2").reshape(-1, w.getnchannels())
                rate = w.getframerate()
        else:
            sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
            from flac_decode import decode_flac

            a, rate, bps = decode_flac(path)
            a = a.astype(np.int16) if bps == 16 else (a.astype(np.int64) << (32 - bps)).astype(np.int32)
        _DECODED[name] = (np.ascontiguousarray(a[:, 0] if a.shape[1] == 1 else a), rate)
    return _DECODED[name]


def cue_bounds(index_mm_ss_ff, n_samples):
    """BlissCueFile::get_songs (src/cue.rs:209-246): a track runs from its INDEX to the next track's, both as
    (as_secs_f32() * SAMPLE_RATE as f32) as usize; the last one to the end of the decoded file."""
    starts = [int((np.float32(m * 60 + s) + np.float32(f * 1_000_000_000 // 75) / np.float32(1e9)) * np.float32(22050))
              for m, s, f in index_mm_ss_ff]   # Duration::as_secs_f32 = secs as f32 + nanos as f32 / 1e9
    return list(zip(starts, starts[1:] + [n_samples]))


# ---- the parity policy of DESIGN.md section 4, in one place for the tests that check a few rows beside their main (bit-identity) assert
FEATURE_TOL = 1e-5   # the reference's own tolerance (src/song/mod.rs:582-590), every non-tempo feature, every song
TEMPO_TOL = 1e-5     # ... and tempo, for every song that is not white noise
TEMPO_HARD = 1e-4    # tempo of a white-noise song (the value ends in an interpolated autocorrelation peak that amplifies FFT rounding:
                     # 1.6 % of white-noise songs sit between 1e-5 and 4.3e-5, as often as the oracle against itself on an f64 FFT)


def assert_row_matches_oracle(got, ref, white_noise, what=""):
    """One feature row against the oracle's by the battery's rule: the 22 (19) non-tempo features within 1e-5; tempo within 1e-5,
    or -- for a white-noise song, where one row cannot carry a fraction -- within the hard 1e-4 bound, printed when above 1e-5."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    err = np.abs(got - ref)
    assert err[1:].max() <= FEATURE_TOL, (what, "non-tempo feature", int(err[1:].argmax()) + 1, float(err[1:].max()))
    if white_noise:
        if err[0] > TEMPO_TOL:
            print(f"tempo of {what or 'a white-noise song'}: |gpu - oracle| = {err[0]:.3g} (> 1e-5, <= 1e-4: the recorded white-noise floor)")
        assert err[0] <= TEMPO_HARD, (what, "tempo", float(err[0]))
    else:
        assert err[0] <= TEMPO_TOL, (what, "tempo", float(err[0]))
