"""The CPU oracle (oracle/bliss_oracle.c) against every golden vector / fixture / known-answer
the reference's own tests hold for the hot path.  CPU only."""
import zlib

import numpy as np
import pytest

from conftest import load_golden


def test_pcm_fixture_adler32(golden_pcm, piano_pcm, literals):
    # src/song/decoder/ffmpeg.rs:455-462 and :524-527 pin the decoded f32le stream
    assert zlib.adler32(golden_pcm.astype("<f4").tobytes()) == int(literals["adler32"]["s16_mono_22_5kHz"], 16)
    assert zlib.adler32(piano_pcm.astype("<f4").tobytes()) == int(literals["adler32"]["piano"], 16)
    assert len(golden_pcm) == 244069


# ---- src/song/mod.rs:553-633 end-to-end ----
@pytest.mark.parametrize("version,key", [(2, "analysis_v2_s16_mono_22_5kHz"), (1, "analysis_v1_s16_mono_22_5kHz")])
def test_analyze_golden(oracle, golden_pcm, literals, version, key):
    got = oracle.song_analyze(golden_pcm, version)
    exp = np.array(literals[key]["values"], np.float32)
    assert got.shape == exp.shape
    assert np.abs(got - exp).max() < literals[key]["tol"]


def test_analysis_too_small(oracle):
    # src/song/mod.rs:539-551
    for x in (np.zeros(1, np.float32), np.zeros(0, np.float32), np.zeros(8191, np.float32)):
        with pytest.raises(oracle.AnalysisError, match="empty or too short song."):
            oracle.song_analyze(x)
    oracle.song_analyze(np.zeros(8192, np.float32))  # exactly the largest window is accepted


# ---- src/utils.rs ----
def test_stft_librosa(oracle, piano_pcm):
    # src/utils.rs:527-541
    got = oracle.stft(piano_pcm, 2048, 512)
    exp = load_golden("librosa-stft.npy")
    assert got.shape == exp.shape == (1025, 253)
    assert np.abs(got - exp).max() < 1e-4


def test_reflect_pad(oracle):
    # src/utils.rs:543-551
    x = np.arange(100, dtype=np.float32)
    out = oracle.reflect_pad(x, 3)
    exp = np.concatenate([[3, 2, 1], x, [98, 97, 96]]).astype(np.float32)
    assert np.array_equal(out, exp)
    assert np.array_equal(out, np.pad(x, 3, mode="reflect"))


def test_geometric_mean(oracle):
    # src/utils.rs:238-260
    assert oracle.geometric_mean([0.0, 1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0]) == 0.0
    assert abs(2.0 - oracle.geometric_mean([4.0, 2.0, 1.0, 4.0, 2.0, 1.0, 2.0, 2.0])) < 1e-4
    assert abs(3.668016172818685 - oracle.geometric_mean([256.0, 4.0, 2.0, 1.0, 4.0, 2.0, 1.0, 2.0])) < 1e-4
    assert abs(1.8340080864093417e-05 - oracle.geometric_mean([4.0, 2.0, 1.0, 4.0, 2.0, 1.0, 2.0, 1.0e-40])) < 1e-4
    mx = np.full(256, 2.0 ** 65, np.float32)
    assert abs(2.0 ** 65 - oracle.geometric_mean(mx)) / 2.0 ** 65 < 1e-4
    rng = np.random.default_rng(0)
    x = rng.random(256, dtype=np.float32) * 4
    assert abs(oracle.geometric_mean(x) - np.exp(np.log(x.astype(np.float64)).mean())) < 1e-4


def test_number_crossings(oracle):
    assert oracle.number_crossings(np.zeros(1024, np.float32)) == 0
    sq = np.tile(np.array([-1.0, 1.0], np.float32), 512)
    assert oracle.number_crossings(sq) == 1023
    # src/timbral.rs:270-286
    assert oracle.zcr(np.zeros(1024, np.float32)) == -1.0
    assert abs(0.9980469 - oracle.zcr(sq)) < 0.001


def test_mean_std_match_numpy(oracle):
    rng = np.random.default_rng(1)
    x = rng.random(31004, dtype=np.float32) * 3000
    assert abs(oracle.mean(x) - x.astype(np.float64).mean()) < 1e-1
    assert abs(oracle.std(x) - x.astype(np.float64).std()) < 1e-1


# ---- src/chroma.rs fixtures ----
def test_chroma_filter(oracle):
    got = oracle.chroma_filter(22050, 2048, 12, -0.1)
    assert np.abs(got - load_golden("chroma-filter.npy")).max() < 1e-9


def test_pip_track(oracle):
    p, m = oracle.pip_track(22050, load_golden("spectrum-chroma.npy"), 2048)
    ep, em = load_golden("spectrum-chroma-pitches.npy"), load_golden("spectrum-chroma-mags.npy")
    assert len(p) == len(ep) == 772
    assert np.abs(np.sort(p) - ep).max() < 1e-8
    assert np.abs(np.sort(m) - em).max() < 1e-8


def test_estimate_tuning(oracle, literals):
    t = oracle.estimate_tuning(22050, load_golden("spectrum-chroma.npy"), 2048, 0.01, 12)
    assert abs(literals["estimate_tuning_spectrum_chroma"]["value"] - t) < 1e-6
    # src/chroma.rs:650-653 (empty fix)
    assert oracle.estimate_tuning(22050, np.zeros((4097, 1)), 8192, 0.01, 12) == 0.0


def test_pitch_tuning(oracle):
    assert oracle.pitch_tuning(load_golden("pitch-tuning.npy"), 0.05, 12) == -0.1
    assert oracle.pitch_tuning(np.zeros(0), 0.05, 12) == 0.0


def test_estimate_tuning_and_chroma_stft_decode(oracle, golden_pcm, literals):
    # src/chroma.rs:621-639, 655-665
    spec = oracle.stft(golden_pcm, 8192, 2205)
    assert spec.shape == (4097, 111)
    t = oracle.estimate_tuning(22050, spec, 8192, 0.01, 12)
    assert abs(literals["estimate_tuning_golden_song"]["value"] - t) < 1e-6
    chroma = oracle.chroma_stft(22050, spec, 8192, 12, -0.04999999999999999)
    assert np.abs(chroma - load_golden("chroma.npy")).max() < 1e-7


def test_extract_interval_features(oracle):
    got = oracle.extract_interval_features(load_golden("chroma-interval.npy"))
    assert np.abs(got - load_golden("interval-feature-matrix.npy")).max() < 1e-7


def test_chroma_interval_features(oracle, literals):
    got = oracle.chroma_interval_features(load_golden("chroma.npy"))
    exp = np.array(literals["chroma_interval_features_of_chroma_npy"]["values"])
    assert np.abs(got - exp).max() < 1e-8


def test_normalize_feature_sequence(oracle):
    # src/chroma.rs:542-555
    got = oracle.normalize_feature_sequence(np.array([[0.1, 0.3, 0.4, 0.0], [1.1, 0.53, 1.01, 0.0]]))
    exp = np.array([[0.08333333, 0.36144578, 0.28368794, 0.0], [0.91666667, 0.63855422, 0.71631206, 0.0]])
    assert np.abs(got - exp).max() < 1e-7


def test_chroma_desc(oracle, golden_pcm, literals):
    # src/chroma.rs:569-619
    chroma, tuning = oracle.chroma_desc(golden_pcm)
    assert abs(tuning + 0.05) < 1e-9
    v2 = oracle.chroma_get_values(chroma, 2)
    assert np.abs(v2[:10] - np.array(literals["chroma_desc_v2_first10"]["values"], np.float32)).max() < 1e-6
    v1 = oracle.chroma_get_values(chroma, 1)
    assert np.abs(v1 - np.array(literals["chroma_desc_v1"]["values"], np.float32)).max() < 1e-6


# ---- per-descriptor known answers (these reference tests frame with chunks_exact(HOP)) ----
def test_timbral_known_answers(oracle, golden_pcm, literals):
    lit = literals["timbral_chunks_exact"]
    c, r, f = oracle.SpectralDesc().run(golden_pcm, framing="chunks_exact").values()
    assert np.abs(c - np.array(lit["centroid"]["values"])).max() < lit["centroid"]["tol"]
    assert np.abs(r - np.array(lit["rolloff"]["values"])).max() < lit["rolloff"]["tol"]
    assert np.abs(f - np.array(lit["flatness"]["values"])).max() < lit["flatness"]["tol"]
    # src/timbral.rs:290-299 streams 128-sample chunks: crossings at chunk boundaries are not counted
    n = (len(golden_pcm) // 128) * 128
    crossings = sum(oracle.number_crossings(golden_pcm[s:s + 128]) for s in range(0, n, 128))
    assert abs((2.0 * crossings / n - 1.0) - lit["zcr"]["value"]) < lit["zcr"]["tol"]


def test_timbral_boundaries(oracle):
    # src/timbral.rs:301-312, 351-363, 418-429: one all-zero hop -> every summary is -1
    d = oracle.SpectralDesc(10)
    d.do_(np.zeros(128, np.float32))
    c, r, f = d.values()
    for v in (c, r, f):
        assert np.abs(v - np.array([-1.0, -1.0])).max() < 1e-7


def test_loudness(oracle, golden_pcm, literals):
    lit = literals["loudness_chunks_exact"]
    got = oracle.loudness(golden_pcm, chunks_exact=True)
    assert np.abs(got - np.array(lit["values"])).max() < lit["tol"]
    # src/misc.rs:98-122 boundaries
    assert np.abs(oracle.loudness(np.zeros(1024, np.float32)) - np.array([-1.0, -1.0])).max() < 1e-7
    assert np.abs(oracle.loudness(np.ones(1024, np.float32)) - np.array([1.0, -1.0])).max() < 1e-7
    assert np.abs(oracle.loudness(-np.ones(1024, np.float32)) - np.array([1.0, -1.0])).max() < 1e-7


def test_tempo_real(oracle, golden_pcm, literals):
    lit = literals["tempo_real_chunks_exact"]
    assert abs(oracle.BPMDesc().run(golden_pcm, framing="chunks_exact").get_value() - lit["value"]) < lit["tol"]


def test_tempo_artificial(oracle, literals):
    # src/temporal.rs:120-138: one click per second -> 60 BPM
    one = np.concatenate([np.zeros(22000, np.float32), np.ones(100, np.float32)])
    x = np.tile(one, 100)
    lit = literals["tempo_artificial_60bpm"]
    assert abs(oracle.BPMDesc().run(x, framing="chunks_exact").get_value() - lit["value"]) < lit["tol"]


def test_tempo_boundaries(oracle, literals):
    # src/temporal.rs:140-161
    d = oracle.BPMDesc(10)
    d.do_(np.zeros(1024, np.float32))
    assert d.get_value() == -1.0
    one = np.concatenate([np.zeros(6989, np.float32), np.ones(20, np.float32)])
    x = np.tile(one, 500)
    lit = literals["tempo_artificial_192bpm"]
    assert abs(oracle.BPMDesc().run(x, framing="chunks_exact").get_value() - lit["value"]) < lit["tol"]
    with pytest.raises(ValueError, match="creation error"):
        oracle.BPMDesc(0)


# ---- src/playlist.rs / src/lib.rs distances (assert_eq on f32 in the reference) ----
def test_distance_literals(oracle, literals):
    lit = literals["distances"]
    a = np.ones(20, np.float32)
    a[19] = 0
    b = np.zeros(20, np.float32)
    b[16] = 1
    assert np.float32(oracle.euclidean_distance(a, b)) == np.float32(lit["euclidean"]["value"])
    assert np.float32(oracle.cosine_distance(a, b)) == np.float32(lit["cosine"]["value"])
    h = np.full(20, 0.5, np.float32)
    assert oracle.euclidean_distance(h, h) == 0.0
    assert oracle.cosine_distance(h, h) == 0.0
    # src/playlist.rs:1008-1024
    b2 = np.zeros(20, np.float32)
    b2[0] = 1
    b2[16] = 1
    m = np.zeros((20, 20), np.float32)
    m[0, 0] = m[1, 1] = 1
    assert oracle.mahalanobis_distance(a, b2, m) == 1.0
    # src/lib.rs:272-291: the v2 literal pins ndarray's unrolled_dot summation order
    assert np.float32(oracle.mahalanobis_distance(np.zeros(20), np.ones(20), oracle.feature_weights(1))) == np.float32(
        lit["v1_metric_zeros_ones"]["value"])
    assert np.float32(oracle.mahalanobis_distance(np.zeros(23), np.ones(23), oracle.feature_weights(2))) == np.float32(
        lit["v2_metric_zeros_ones"]["value"])


def test_white_noise_generator(oracle):
    x = oracle.white_noise(3, 10007)
    assert x.dtype == np.float32 and x.min() >= -0.5 and x.max() < 0.5
    assert abs(float(x.mean())) < 0.02 and abs(float(x.std()) - 12 ** -0.5) < 0.01
    assert np.array_equal(x[:4096], oracle.white_noise(3, 4096))  # prefix-stable (counter = sample index / 4)
    assert not np.array_equal(x[:64], oracle.white_noise(4, 64))
