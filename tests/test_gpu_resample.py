"""GPU tests of row f1's resampler (-m gpu): decoder output at any sample rate -> mono 22 050 Hz f32 ON THE DEVICE, the way
the reference's FFmpegDecoder does it on the CPU (libswresample with default options, src/song/decoder/ffmpeg.rs:36-109).

  * the reference's own pins: the Adler-32 of the converted stream for data/s32_mono_44_1_kHz.flac, s32_stereo_44_1_kHz.flac
    (ffmpeg.rs:433-445) and no_channel.wav (:471-476) -- the device output must hash to them
  * device == oracle bit for bit over rates (one phase, 147 phases, 441 phases, the 1024-phase non-rational case, up-sampling),
    channel counts, the three sample formats and awkward lengths
  * with it, through the C ABI end to end: the three CUE tracks of data/testcue.flac against the 3 x 23 features
    src/cue.rs:270-415 asserts, and the mixed-rate bulk entry point against per-song conversion + analysis
"""
import zlib

import numpy as np
import pytest

from conftest import assert_row_matches_oracle, cue_bounds, decoded_audio

pytestmark = pytest.mark.gpu

FEATURE_TOL = 1e-5  # src/song/mod.rs:582-590


@pytest.fixture(scope="module")
def bliss():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import bliss_rs_amd

    return bliss_rs_amd


@pytest.fixture(scope="module")
def ctx(bliss):
    c = bliss.Context(0)
    yield c
    c.close()


def _adler(x):
    return zlib.adler32(np.ascontiguousarray(x, dtype="<f4").tobytes()) & 0xFFFFFFFF


def _decode(ctx, samples, rate):
    import torch

    out = ctx.pcm_decode(torch.from_numpy(np.array(samples)).cuda(), rate)
    ctx.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("name", ["s32_mono_44_1_kHz.flac", "s32_stereo_44_1_kHz.flac", "no_channel.wav"])
def test_device_resampler_hits_the_reference_adler32(ctx, oracle, literals, name):
    samples, rate = decoded_audio(name)
    got = _decode(ctx, samples, rate)
    assert _adler(got) == int(literals["resample"]["adler32"][name], 16)
    assert np.array_equal(got.view(np.uint32), oracle.decode_to_mono(samples, rate).view(np.uint32))


@pytest.mark.parametrize("rate", [1000, 8000, 11025, 16000, 24000, 32000, 33075, 37800, 44056, 44100, 48000, 64000, 88200, 96000,
                                  192000, 352800, 768000])
def test_device_resampler_equals_oracle_bitwise(ctx, oracle, rate):
    rng = np.random.default_rng(rate)
    taps = oracle.swr_filter(rate)[1].taps
    # lengths around the start-up minimum (taps + 1 samples), a few tiles, and odd sizes; enough input for >= 2 workgroups
    n_big = max(3000, 2600 * rate // 22050)
    for n in (taps, taps + 1, taps + 2, 2 * taps + 5, n_big, n_big + 1, n_big + 7):
        for channels, dtype in ((1, np.float32), (2, np.int16), (1, np.int32), (3, np.int16), (2, np.float32)):
            if dtype == np.float32:
                x = (rng.random((n, channels), np.float32) - 0.5).astype(np.float32)
            elif dtype == np.int16:
                x = rng.integers(-32768, 32768, (n, channels)).astype(np.int16)
            else:
                x = rng.integers(-2**31, 2**31, (n, channels)).astype(np.int32)
            if channels == 1:
                x = x[:, 0]
            ref = oracle.decode_to_mono(x, rate)
            got = _decode(ctx, x, rate)
            assert got.shape == ref.shape == (oracle.swr_out_len(n, rate),), (rate, n)
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (rate, n, channels, dtype)


def test_device_resampler_three_minutes_48k_stereo(ctx, oracle):
    # a full-size song: 3 minutes of 48 kHz stereo s16 (8.64 M frames -> 3 969 000 samples, 3 876 workgroups)
    rng = np.random.default_rng(48)
    x = rng.integers(-20000, 20000, (8_640_000, 2)).astype(np.int16)
    got = _decode(ctx, x, 48000)
    ref = oracle.decode_to_mono(x, 48000)
    assert len(got) == 3_969_000
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_pass_through_at_22050_is_the_downmix_path(ctx, oracle, literals):
    from conftest import load_golden

    s = load_golden("s16_stereo_22_5kHz.pcm_s16.npy")
    got = _decode(ctx, s, 22050)
    assert _adler(got) == int(literals["resample"]["adler32"]["s16_stereo_22_5kHz.flac"], 16)


def test_cue_tracks_through_the_device(bliss, ctx, oracle, literals):
    """BlissCue::songs_from_path on data/testcue.cue (src/cue.rs:209-246, 270-415): the file is decoded ONCE -- 44.1 kHz stereo
    s16 -> mono 22 050 Hz on the device -- and the tracks are (offset, length) slices of the device buffer."""
    import torch

    cue = literals["resample"]["cue"]
    samples, rate = decoded_audio(cue["file"])
    pcm = ctx.pcm_decode(torch.from_numpy(np.array(samples)).cuda(), rate)
    assert pcm.numel() == 496272
    bounds = cue_bounds(cue["index_mm_ss_ff"], pcm.numel())
    out, status = ctx.analyze(pcm, [a for a, _ in bounds], [b - a for a, b in bounds], 2)
    ctx.synchronize()
    rows = out.cpu().numpy()
    assert status.cpu().tolist() == [0, 0, 0]
    host = pcm.cpu().numpy()
    # the device's conversion IS FFmpeg's: same bits as the oracle's restatement, which is pinned on the Adler-32 values
    assert np.array_equal(host.view(np.uint32), oracle.decode_to_mono(samples, rate).view(np.uint32))
    for t, (row, (a, b), exp) in enumerate(zip(rows, bounds, cue["tracks"])):
        exp = np.array(exp, np.float32)
        ref = oracle.song_analyze(host[a:b], 2)
        oracle.set_fft_double(True)
        try:
            floor = np.abs(oracle.song_analyze(host[a:b], 2).astype(np.float64) - ref)
        finally:
            oracle.set_fft_double(False)
        d_ref, d_orc = np.abs(row - exp), np.abs(row.astype(np.float64) - ref)
        print(f"track {t + 1}: max |gpu - reference literal| {d_ref.max():.2e} (feature {int(d_ref.argmax())}), "
              f"max |gpu - oracle| {d_orc.max():.2e}, oracle f32-vs-f64 floor {floor.max():.2e}")
        # the reference's tolerance on every feature; the flatness pair of the noise-free tracks (piano, a pure tone) may
        # instead follow the oracle's own f32-vs-f64 FFT distance (the policy of tests/test_gpu_round3.py)
        tol = np.full(23, FEATURE_TOL)
        tol[6:8] = np.maximum(FEATURE_TOL, floor[6:8])
        assert (d_orc <= tol).all(), (t, d_orc)
        assert (d_ref <= tol + 4e-7).all(), (t, d_ref)  # + the oracle's own distance from the literals (3.6e-7)


    # FeaturesVersion::Version1 on the same slices (src/cue.rs:417-523): 3 x 20 literals
    out1, status1 = ctx.analyze(pcm, [a for a, _ in bounds], [b - a for a, b in bounds], 1)
    ctx.synchronize()
    rows1 = out1.cpu().numpy()
    assert rows1.shape == (3, 20) and status1.cpu().tolist() == [0, 0, 0]
    for row, exp in zip(rows1, literals["resample"]["cue_v1"]["tracks"]):
        d = np.abs(row - np.array(exp, np.float32))
        assert d[10:].max() < FEATURE_TOL, d  # the ten Version1 chroma features
    # (the first ten features are the same computation in both versions -- and the same literals: held above)
    assert np.array_equal(rows1[:, :10], rows[:, :10])


def test_cue_helper_equals_the_hand_built_slices(bliss, ctx, literals):
    # bliss_rs_amd.cue.analyze_cue_tracks = the steps of the test above behind the interface a host would call
    cue = literals["resample"]["cue"]
    samples, rate = decoded_audio(cue["file"])
    secs = [tuple(msf) for msf in cue["index_mm_ss_ff"]]   # (mm, ss, ff) of the sheet
    res = bliss.cue.analyze_cue_tracks(ctx, samples, rate, secs)
    assert bliss.cue.cue_track_bounds(secs, 496272) == cue_bounds(cue["index_mm_ss_ff"], 496272)
    for r, exp in zip(res, cue["tracks"]):
        assert np.abs(r.as_arr1() - np.array(exp, np.float32)).max() < FEATURE_TOL
    v1 = bliss.cue.analyze_cue_tracks(ctx, samples, rate, secs, bliss.AnalysisOptions(features_version=1))
    for r, exp in zip(v1, literals["resample"]["cue_v1"]["tracks"]):
        assert len(r.as_arr1()) == 20 and np.abs(r.as_arr1() - np.array(exp, np.float32))[10:].max() < FEATURE_TOL
    # a track shorter than the largest window is the reference's AnalysisError in that slot, not a failed call
    short = bliss.cue.analyze_cue_tracks(ctx, samples, rate, [(0, 0), (22, 400_000_000)])   # (secs, nanos)
    assert isinstance(short[1], bliss.AnalysisError) and not isinstance(short[0], bliss.BlissError)


def test_analyze_decoded_single_song(bliss, oracle, literals):
    # Song::analyze on FFmpegDecoder's output for the 44.1 kHz stereo twin of the golden song, through the single-song front
    samples, rate = decoded_audio("s32_stereo_44_1_kHz.flac")
    got = bliss.Song.analyze_decoded(samples, rate).as_arr1()
    ref = oracle.song_analyze(oracle.decode_to_mono(samples, rate), 2)
    assert np.abs(got - ref).max() < FEATURE_TOL
    t = literals["resample"]["analysis_symphonia_s32_stereo_44_1_kHz"]  # the reference's other resampler: its own 0.1
    assert np.abs(got - np.array(t["values"], np.float32)).max() < t["tol"]


def test_analyze_batch_decoded_mixed_library(bliss, ctx, oracle):
    """A library as it comes off the decoders: 44.1 kHz stereo s16, 48 kHz mono f32, 22 050 Hz mono f32 (copied verbatim),
    24-bit mono, a 96 kHz file, one too short -- one call; every row equals converting and analysing that song alone."""
    import torch

    rng = np.random.default_rng(7)
    cue, rate_cue = decoded_audio("testcue.flac")
    s24, rate_24 = decoded_audio("s32_mono_44_1_kHz.flac")
    songs = [
        (cue[: 44100 * 6], rate_cue),
        ((rng.random(48000 * 4, np.float32) - 0.5).astype(np.float32), 48000),
        (oracle.white_noise(3, 22050 * 3), 22050),
        (s24[: 44100 * 5], rate_24),
        (rng.integers(-9000, 9000, (96000 * 2, 2)).astype(np.int16), 96000),
        (rng.integers(-9000, 9000, 15000).astype(np.int16), 44100),  # 7 500 samples at 22 050 Hz: too short
    ]
    res = bliss.analyze_decoded_batch([s for s, _ in songs], [r for _, r in songs])
    assert isinstance(res[5], bliss.AnalysisError) and res[5].message == "empty or too short song."
    noise = [False, True, True, False, True]   # recordings / random samples
    for k, ((s, r), got) in enumerate(zip(songs[:5], res[:5])):
        pcm = ctx.pcm_decode(torch.from_numpy(np.ascontiguousarray(s)).cuda(), r)
        out, status = ctx.analyze(pcm, [0], [pcm.numel()], 2)
        ctx.synchronize()
        assert np.array_equal(out.cpu().numpy()[0].view(np.uint32), got.as_arr1().view(np.uint32))
        ref = oracle.song_analyze(oracle.decode_to_mono(s, r), 2)
        assert_row_matches_oracle(got.as_arr1(), ref, white_noise=noise[k], what=f"mixed library song {k}")
    # v1 rows and the uniform-rate form
    res1 = bliss.analyze_decoded_batch([songs[0][0], songs[3][0]], 44100, bliss.AnalysisOptions(features_version=1))
    assert len(res1[0].as_arr1()) == 20 and np.array_equal(res1[0].as_arr1()[:10], res[0].as_arr1()[:10])


def test_decoded_entry_points_reject_bad_arguments(bliss):
    import ctypes as C

    from bliss_rs_amd import _ffi

    L = _ffi.lib()
    x = np.zeros(20000, np.float32)
    row = np.zeros(23, np.float32)
    for args in ((_ffi.SAMPLE_F32, 1, 0), (_ffi.SAMPLE_F32, 1, 768001), (_ffi.SAMPLE_F32, 0, 44100), (_ffi.SAMPLE_F32, 9, 44100),
                 (7, 1, 44100)):
        fmt, ch, rate = args
        assert L.blissgpu_analyze_decoded(x.ctypes.data, fmt, ch, 1000, rate, 2, row.ctypes.data, None) == _ffi.ERR_INVALID
    songs = (_ffi.DecodedSong * 1)(_ffi.DecodedSong(x.ctypes.data, 1000, 0, 1, 0))
    assert L.blissgpu_analyze_batch_decoded(songs, 1, 2, row.ctypes.data, None) == _ffi.ERR_INVALID
    st = C.c_int32(-1)
    # a stream shorter than the resampler's start-up converts to nothing: too short, not an error
    _ffi.check(L.blissgpu_analyze_decoded(x.ctypes.data, _ffi.SAMPLE_F32, 1, 40, 44100, 2, row.ctypes.data, C.byref(st)))
    assert st.value == _ffi.SONG_TOO_SHORT and np.isnan(row).all()
