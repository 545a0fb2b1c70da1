"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs, against the reference's golden vectors, and -- at BASELINE.json's full sizes -- through
size-independent properties.

Tolerances (features are normalised to [-1, 1]):
  * integer work (zero crossings, statuses, pitch histogram bins, beat counts): exact
  * distances: bit-exact (the kernel reproduces ndarray's summation order)
  * the 22 non-tempo features: |gpu - oracle| <= FEATURE_TOL = 1e-5, the reference's own tolerance (observed <= 5e-6); the only
    difference between the two paths is f32 FFT rounding (three register radix-16 passes + real split on the GPU,
    radix-2 c2c in the oracle, rustfft in the reference -- no two of them round alike).  Rolloff is a
    per-frame integer bin and the FFT-512 kernel counts it in the reference's summation order (round 3), so the two
    rolloff entries get NO extra allowance here (ROLLOFF_FLIPS = 0; until round 4 every tolerance carried two flipped
    frames).  Only spectra with plateaus between partials keep an allowance, where it is demonstrated
    (test_random_musical_songs_vs_oracle: a frame sitting within an ulp of the threshold ON a plateau).
  * tempo: held to the reference's own 1e-5 (src/song/mod.rs:582-590) for a RECORDED fraction of the songs, every song
    above it listed, and to TEMPO_HARD = 1e-4 for every song.  The fraction is not a convenience: the tempo value ends in
    a parabolic interpolation of an autocorrelation peak that amplifies FFT rounding, and the oracle ITSELF moves by
    more than 1e-5 on 15 of the 1024 bench songs when its FFTs run in f64 instead of f32 (max 5.8e-5); the GPU differs
    from the f32 oracle on 16 of 1024 (max 4.3e-5) and from the f64 oracle on 14 (profiles/r03_full_check_1024songs.json,
    tests/tools/full_check.py --noise-floor).  (Conjecture, not measured here -- there is no Rust toolchain: rustfft's f32
    transforms would sit at the same floor.)  The allowance is for WHITE NOISE, where the beat tracker's autocorrelation
    peak is a near-tie by construction: a test of fewer than 34 songs that are not white noise gets none (the musical
    sets are within 3.6e-7).
  * tuning (discrete, 0.01 semitone bins): must match for every song of the battery; a mismatch would
    be reported, not hidden (white noise makes the histogram argmax a near-tie by construction).
  * tempo differences above TEMPO_HARD are held to a RECORDED expectation (EXPECTED_TEMPO_MISMATCHES = 0 everywhere): a
    flipped beat decision cannot hide inside the allowance for rounding.
"""
import os

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

FEATURE_TOL = 1e-5   # the reference's own tolerance (src/song/mod.rs:582-590); observed <= 5e-6
TEMPO_TOL = 1e-5      # the reference's tolerance; held for >= 1 - TEMPO_NOISE_FRACTION of the songs of a test
TEMPO_HARD = 1e-4     # every song
TEMPO_NOISE_FRACTION = 0.03   # recorded: 1.6 % of the 1024 bench songs (GPU vs oracle), 1.5 % oracle-f32 vs oracle-f64
ROLLOFF_FLIPS = 0     # frames whose rolloff bin may differ from the oracle's (the kernel counts in the reference's order)
EXPECTED_TEMPO_MISMATCHES = 0   # recorded expectation: a change here is a regression to look at, not noise to absorb
N3MIN = 3969000


@pytest.fixture(scope="module")
def bliss():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import bliss_rs_amd

    return bliss_rs_amd


@pytest.fixture(scope="module")
def ctx(bliss):
    return bliss.Context(0)


def _run(ctx, songs, version=2):
    import torch

    lens = [len(s) for s in songs]
    offs = np.concatenate([[0], np.cumsum([(l + 63) // 64 * 64 for l in lens])[:-1]]).astype(np.uint64)
    buf = np.zeros(int(offs[-1]) + lens[-1] + 64, np.float32)
    for s, o in zip(songs, offs):
        buf[int(o):int(o) + len(s)] = s
    pcm = torch.from_numpy(buf).cuda()
    out, status = ctx.analyze(pcm, offs, lens, version)
    ctx.synchronize()
    return out.cpu().numpy(), status.cpu().numpy()


def _tol(n_samples, d):
    tol = np.full(d, FEATURE_TOL)
    tol[0] = TEMPO_HARD
    n_t = (n_samples - 512) // 128 + 1
    flip = 2.0 * (22050.0 / 512.0) / 11025.0 / n_t
    tol[4] += ROLLOFF_FLIPS * flip                                   # mean rolloff
    tol[5] += ROLLOFF_FLIPS * flip * np.sqrt(max(n_t, 1)) * 0.5      # its std moves ~ bin / sqrt(n)
    return tol


def _tempo_gate(errs, names, white_noise=True):
    """tempo |gpu - oracle| of the songs of one test: all <= TEMPO_HARD, and <= TEMPO_TOL (the reference's 1e-5) for all
    but the recorded noise fraction -- ceil(3 %), which is at least one song only when the songs are white noise (or the
    test has 34 or more of them); every song above 1e-5 is listed in the test output"""
    errs = np.asarray(errs, np.float64)
    over = [(names[i], float(errs[i])) for i in np.flatnonzero(errs > TEMPO_TOL)]
    print(f"tempo: {len(errs) - len(over)} of {len(errs)} songs within 1e-5; above it: {over}")
    assert int((errs > TEMPO_HARD).sum()) == EXPECTED_TEMPO_MISMATCHES, over
    allowed = int(np.ceil(TEMPO_NOISE_FRACTION * len(errs))) if (white_noise or len(errs) >= 34) else 0
    assert len(over) <= allowed, over


def battery(oracle):
    rng = np.random.default_rng(7)
    t = np.arange(20 * 22050) / 22050.0
    chord = sum(np.sin(2 * np.pi * f * t) for f in (261.63, 329.63, 392.0)) * 0.2
    songs = {
        "noise_3min": oracle.white_noise(11, N3MIN),
        "noise_47s_odd": oracle.white_noise(12, 47 * 22050 + 1234),
        "noise_exact_hop_multiple": oracle.white_noise(13, 2205 * 300),       # last STFT window dropped
        "min_len_8192": oracle.white_noise(14, 8192),
        "len_8193": oracle.white_noise(15, 8193),
        "silence": np.zeros(3 * 22050, np.float32),
        "dc_offset": np.full(2 * 22050, 0.25, np.float32),
        "click_60bpm": np.tile(np.concatenate([np.zeros(22000, np.float32), np.ones(100, np.float32)]), 40),
        "click_192bpm": np.tile(np.concatenate([np.zeros(6989, np.float32), np.ones(20, np.float32)]), 120),
        "c_major_chord": chord.astype(np.float32),
        "sweep": np.sin(2 * np.pi * (100 + 2000 * t) * t).astype(np.float32) * 0.5,
        "square_wave": np.tile(np.array([-1.0] * 25 + [1.0] * 25, np.float32), 4000),
        "quiet_noise": (rng.standard_normal(5 * 22050) * 1e-4).astype(np.float32),
        "loud_clipped": np.clip(rng.standard_normal(6 * 22050), -1, 1).astype(np.float32),
        "silence_then_noise": np.concatenate([np.zeros(4 * 22050, np.float32), oracle.white_noise(16, 8 * 22050)]),
    }
    return songs


# ---------------------------------------------------------------------------------------------
# golden vectors of the reference (src/song/mod.rs:553-633), at the reference's own tolerance
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("version,key", [(2, "analysis_v2_s16_mono_22_5kHz"), (1, "analysis_v1_s16_mono_22_5kHz")])
def test_golden_song_matches_reference_literals(bliss, golden_pcm, literals, version, key):
    analysis = bliss.Song.analyze_with_options(golden_pcm, bliss.AnalysisOptions(bliss.FeaturesVersion(version)))
    exp = np.array(literals[key]["values"], np.float32)
    got = analysis.as_arr1()
    assert got.shape == exp.shape
    assert np.abs(got - exp).max() < literals[key]["tol"], (got - exp)


def test_golden_song_matches_oracle(ctx, oracle, golden_pcm):
    for version in (2, 1):
        got, status = _run(ctx, [golden_pcm], version)
        ref = oracle.song_analyze(golden_pcm, version)
        assert status.tolist() == [0]
        assert (np.abs(got[0] - ref) <= _tol(len(golden_pcm), len(ref))).all(), np.abs(got[0] - ref)
    tuning, n_bpms = ctx.last_tuning(1)
    assert abs(tuning[0] + 0.05) < 1e-12          # estimate_tuning of the golden song (src/chroma.rs:655-665)
    assert n_bpms[0] == 23                        # beats BPMDesc records on it


# ---------------------------------------------------------------------------------------------
# battery vs oracle
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("version", [2, 1])
def test_battery_vs_oracle(ctx, oracle, version):
    songs = battery(oracle)
    names = list(songs)
    got, status = _run(ctx, [songs[k] for k in names], version)
    tuning, n_bpms = ctx.last_tuning(len(names))
    assert (status == 0).all()
    report, tempo_err = [], []
    for i, k in enumerate(names):
        ref = oracle.song_analyze(songs[k], version)
        _, otuning = oracle.chroma_desc(songs[k])
        err = np.abs(got[i] - ref)
        tol = _tol(len(songs[k]), len(ref))
        report.append(f"{k:26s} max|err| non-tempo {err[1:].max():.2e} tempo {err[0]:.2e} tuning {tuning[i]:+.2f}/{otuning:+.2f}")
        assert abs(tuning[i] - otuning) < 1e-12, f"{k}: tuning gpu {tuning[i]} oracle {otuning}"
        assert (err[1:] <= tol[1:]).all(), f"{k}: {err}"
        tempo_err.append(err[0])
        assert n_bpms[i] == len(oracle.BPMDesc().run(songs[k]).bpms()) or err[0] > tol[0]
    print("\n".join(report))
    _tempo_gate(tempo_err, names)


def musical_battery(oracle):
    """Signals with structure in the places the white-noise bench has none: detuned harmony (the tuning estimate must
    land in another bin and select another filter bank), steady and swung pulses (the beat tracker's lock), modulated
    and harmonic tones, a 10-minute song (the longest the mixed corpus holds), amplitudes far from 1."""
    sr = 22050
    rng = np.random.default_rng(99)

    def chord(freqs, seconds, cents=0.0, amp=0.2):
        t = np.arange(int(seconds * sr)) / sr
        k = 2.0 ** (cents / 1200.0)
        return (sum(np.sin(2 * np.pi * f * k * t) for f in freqs) * amp).astype(np.float32)

    def bursts(period_s, seconds, width_s=0.05, swing=0.0):
        x = np.zeros(int(seconds * sr), np.float32)
        noise = rng.standard_normal(len(x)).astype(np.float32) * 0.5
        k, t0 = 0, 0.0
        while t0 < seconds - width_s:
            a = int(t0 * sr)
            b = a + int(width_s * sr)
            x[a:b] = noise[a:b] * np.hanning(b - a).astype(np.float32)
            k += 1
            t0 = k * period_s + (swing * period_s if k % 2 else 0.0)
        return x

    t15 = np.arange(15 * sr) / sr
    saw = (2.0 * ((110.0 * t15) % 1.0) - 1.0).astype(np.float32) * 0.4
    am = (np.sin(2 * np.pi * 440.0 * t15) * (0.5 + 0.5 * np.sin(2 * np.pi * 2.0 * t15))).astype(np.float32) * 0.6
    return {
        "chord_plus_23_cents": chord((220.0, 277.18, 329.63, 440.0), 25, cents=23.0),
        "chord_minus_41_cents": chord((196.0, 246.94, 293.66), 25, cents=-41.0),
        "chord_over_noise": chord((261.63, 329.63, 392.0), 30, cents=8.0) + oracle.white_noise(31, 30 * sr) * 0.05,
        "bursts_120bpm": bursts(0.5, 40),
        "bursts_93bpm_swing": bursts(60.0 / 93.0, 45, swing=0.17),
        "bursts_172bpm": bursts(60.0 / 172.0, 35, width_s=0.02),
        "am_tone_440": am,
        "sawtooth_110": saw,
        "noise_10min": oracle.white_noise(32, 10 * 60 * sr),
        "tiny_amplitude_1e-12": oracle.white_noise(33, 12 * sr) * np.float32(1e-12),
        "huge_amplitude_1e4": oracle.white_noise(34, 9 * sr) * np.float32(1e4),
    }


def test_musical_battery_vs_oracle(ctx, oracle):
    songs = musical_battery(oracle)
    names = list(songs)
    got, status = _run(ctx, [songs[k] for k in names], 2)
    tuning, n_bpms = ctx.last_tuning(len(names))
    assert (status == 0).all()
    tempo_err, report, tunings = [], [], set()
    for i, k in enumerate(names):
        ref = oracle.song_analyze(songs[k], 2)
        _, otuning = oracle.chroma_desc(songs[k])
        err = np.abs(got[i] - ref)
        tol = _tol(len(songs[k]), len(ref))
        report.append(f"{k:26s} max|err| non-tempo {err[1:].max():.2e} tempo {err[0]:.2e} tuning {tuning[i]:+.2f}/{otuning:+.2f} bpms {n_bpms[i]}")
        assert abs(tuning[i] - otuning) < 1e-12, f"{k}: tuning gpu {tuning[i]} oracle {otuning}"
        assert (err[1:] <= tol[1:]).all(), f"{k}: {err}"
        tempo_err.append(err[0])
        tunings.add(round(float(tuning[i]), 2))
    print("\n".join(report))
    assert len(tunings) >= 4, tunings            # the battery does exercise several filter banks
    noise = [i for i, k in enumerate(names) if "noise" in k or "amplitude" in k]      # white noise at some gain
    music = [i for i in range(len(names)) if i not in noise]
    _tempo_gate([tempo_err[i] for i in music], [names[i] for i in music], white_noise=False)
    _tempo_gate([tempo_err[i] for i in noise], [names[i] for i in noise])


def test_stage_taps_vs_oracle(ctx, oracle, golden_pcm):
    """Every intermediate series against the oracle's streaming descriptors."""
    songs = [golden_pcm, oracle.white_noise(21, 30 * 22050 + 100)]
    _run(ctx, songs, 2)
    for i, x in enumerate(songs):
        sd = oracle.SpectralDesc().run(x)
        c, r, f = sd.series()
        gc, gr, gf = (ctx.debug_fetch(k, i) for k in ("centroid", "rolloff", "flatness"))
        assert len(gc) == len(c) == (len(x) - 512) // 128 + 1
        assert np.abs(gc - c).max() < 2e-2                  # Hz, on values up to 11025
        assert (np.abs(gr - r) > 1e-3).sum() <= 1           # integer bins x 43.07 Hz, counted in the reference's order: the
                                                            # golden song's 1 903 frames hold ONE on which the oracle's own f32
                                                            # and f64 FFTs disagree
        assert np.abs(gf - f).max() < 3e-5                  # log2f/exp2f of the geometric mean: a few ulp on ~0.85
        bd = oracle.BPMDesc().run(x)
        onset, thr = bd.series()
        gflux, gthr = ctx.debug_fetch("flux", i), ctx.debug_fetch("thresholded", i)
        assert len(gflux) == len(onset) == (len(x) - 512) // 256 + 1
        assert np.abs(gflux - onset).max() <= 2e-6 * max(1.0, np.abs(onset).max())
        assert np.abs(gthr - thr).max() <= 2e-6 * max(1.0, np.abs(onset).max())
        # beat-tracker runs: same bpm sequence and beat counts
        bpms = bd.bpms()
        rb, rc = ctx.debug_fetch("run_bpm", i), ctx.debug_fetch("run_count", i)
        assert int(rc.sum()) == len(bpms)
        assert np.allclose(np.repeat(rb, rc), bpms, rtol=2e-5)
        # STFT magnitudes and exact integer work
        spec = oracle.stft(x, 8192, 2205)                   # [4097, frames]
        gspec = ctx.debug_fetch("spectrogram", i)           # [frames, 4097]
        assert gspec.shape == spec.T.shape
        assert np.abs(gspec - spec.T).max() <= 3e-6 * spec.max()
        zc = ctx.debug_fetch("crossings256", i)
        assert int(zc.sum()) == oracle.number_crossings(x)  # exact
        e = ctx.debug_fetch("energy256", i)
        ref_e = np.add.reduceat(x.astype(np.float64) ** 2, np.arange(0, len(x), 256))
        assert np.allclose(e, ref_e, rtol=1e-5, atol=1e-12)


# ---------------------------------------------------------------------------------------------
# errors, ragged batches, independence, determinism
# ---------------------------------------------------------------------------------------------
def test_too_short_and_mixed_batch(bliss, ctx, oracle):
    # src/song/mod.rs:539-551
    for x in (np.zeros(1, np.float32), np.zeros(0, np.float32), np.zeros(8191, np.float32)):
        with pytest.raises(bliss.AnalysisError, match="empty or too short song."):
            bliss.Song.analyze(x)
    songs = [oracle.white_noise(1, 9000), np.zeros(100, np.float32), oracle.white_noise(2, 30000),
             np.zeros(0, np.float32), oracle.white_noise(3, 8192)]
    res = bliss.analyze_batch(songs)
    assert [isinstance(r, bliss.AnalysisError) for r in res] == [False, True, False, True, False]
    assert res[1] == bliss.AnalysisError("empty or too short song.")
    got, status = _run(ctx, [s if len(s) else np.zeros(1, np.float32) for s in songs])
    assert status.tolist() == [0, 1, 0, 1, 0]
    assert np.isnan(got[1]).all() and np.isnan(got[3]).all()
    for i in (0, 2, 4):
        assert np.array_equal(got[i], res[i].as_arr1())     # host-buffer and device-resident entry points agree
        ref = oracle.song_analyze(songs[i])
        assert (np.abs(got[i][1:] - ref[1:]) <= _tol(len(songs[i]), 23)[1:]).all()


def test_batch_composition_and_order_do_not_matter(ctx, oracle):
    songs = [oracle.white_noise(40 + i, n) for i, n in enumerate((50000, 200001, 8192, 123457, 90000))]
    a, _ = _run(ctx, songs)
    perm = [3, 0, 4, 2, 1]
    b, _ = _run(ctx, [songs[p] for p in perm])
    for j, p in enumerate(perm):
        assert np.array_equal(a[p], b[j])                    # bit-identical
    c, _ = _run(ctx, [songs[1]])
    assert np.array_equal(c[0], a[1])
    a2, _ = _run(ctx, songs)
    assert np.array_equal(a, a2)                             # deterministic run to run


def test_workspace_chunking(bliss, oracle):
    """A tiny workspace limit forces several chunks; results must not change."""
    songs = [oracle.white_noise(60 + i, 40000 + 1000 * i) for i in range(6)]
    c1 = bliss.Context(0)
    ref, _ = _run(c1, songs)
    c2 = bliss.Context(0)
    c2.set_workspace_limit(64 << 20)
    got, status = _run(c2, songs)
    assert np.array_equal(ref, got) and (status == 0).all()


def test_frame_and_tile_boundary_lengths(ctx, oracle):
    """Song lengths that sit on (and one sample either side of) every framing / tiling boundary of the kernels: the
    8192-sample minimum, the chroma hop 2205 (ceil(N/2205) frames, last window dropped at exact multiples), 16-frame
    STFT tiles, 64-frame chroma tiles, the FFT-512 hop 128 / tempo hop 256 / 512-frame tiles, 256-sample energy blocks,
    1024-sample loudness chunks and the 128-frame beat-tracker step."""
    lengths = set()
    for base in (8192, 2205 * 4, 2205 * 16, 2205 * 17, 2205 * 32, 2205 * 64, 2205 * 65, 128 * 512, 128 * 512 + 384, 256 * 255,
                 256 * 256, 1024 * 40, 256 * 128 * 2, 256 * 128 * 3 + 128, 16 * 4096, 100000):
        for delta in (-1, 0, 1):
            if base + delta >= 8192:
                lengths.add(base + delta)
    # the FFT-512 kernel names a frame by its position in a 32-frame lane group and tests it against the group's own limits (round
    # 6): timbral frame counts n_t on and either side of group and tile edges, with the remainder r that decides whether the
    # tempo path adds one more FFT frame than the timbral path has (n - 512 = 256 q + r: n_t = 2 q + 1 or 2 q + 2, 2 n_b = 2 q + 2)
    for n_t in (63, 64, 65, 95, 96, 97, 127, 128, 129, 511, 512, 513, 543, 544, 545, 1023, 1024, 1025):
        for r in (0, 127):
            lengths.add(128 * (n_t - 1) + 512 + r)
    lengths = sorted(lengths)
    songs = [oracle.white_noise(700 + i, n) for i, n in enumerate(lengths)]
    got, status = _run(ctx, songs)
    tuning, n_bpms = ctx.last_tuning(len(songs))
    assert (status == 0).all()
    tempo_err = []
    for i, (n, x) in enumerate(zip(lengths, songs)):
        ref = oracle.song_analyze(x)
        _, otuning = oracle.chroma_desc(x)
        err = np.abs(got[i] - ref)
        assert abs(tuning[i] - otuning) < 1e-12, (n, tuning[i], otuning)
        assert (err[1:] <= _tol(n, 23)[1:]).all(), (n, err)
        tempo_err.append(err[0])
    _tempo_gate(tempo_err, lengths)


def test_mixed_duration_corpus_and_cue_slices(bliss, oracle):
    """BASELINE configs[4] in miniature: a ragged corpus of 30 s .. 10 min songs generated in HBM, analysed under a
    workspace limit that forces several chunks; a sample is checked against the oracle and against the same songs
    analysed alone.  Second half: CUE-style tracks are (offset, length) slices of ONE decoded buffer,
    adjacent and overlapping -- the batch descriptor takes them as they are, results equal separate buffers."""
    import torch

    rng = np.random.default_rng(21)
    n = 36
    lens = rng.integers(30 * 22050, 10 * 60 * 22050 + 1, n).astype(np.uint64)
    lens[5] = 10 * 60 * 22050          # the longest allowed
    lens[9] = 30 * 22050               # the shortest
    offs = np.zeros(n, np.uint64)
    offs[1:] = np.cumsum((lens + 63) // 64 * 64)[:-1]
    total = int(offs[-1] + lens[-1])
    ctx = bliss.Context(0)
    ctx.set_workspace_limit(512 << 20)  # a few songs per chunk => many chunks
    pcm = torch.empty(total, dtype=torch.float32, device="cuda")
    ctx.synth_white_noise(pcm, offs, lens, first_song_index=500)
    out, status = ctx.analyze(pcm, offs, lens, 2)
    ctx.synchronize()
    out = out.cpu().numpy()
    assert (status.cpu().numpy() == 0).all() and np.isfinite(out).all()
    alone = bliss.Context(0)
    for i in (5, 9, 20):
        ref = oracle.song_analyze(oracle.white_noise(500 + i, int(lens[i])))
        tol = _tol(int(lens[i]), 23)
        assert (np.abs(out[i][1:] - ref[1:]) <= tol[1:]).all(), i
        assert abs(out[i][0] - ref[0]) <= TEMPO_HARD, i
        one, _ = alone.analyze(pcm, offs[i:i + 1], lens[i:i + 1], 2)
        alone.synchronize()
        assert np.array_equal(one.cpu().numpy()[0], out[i])   # chunk / batch composition does not matter

    # ---- CUE-style slices of one buffer ----
    disc = oracle.white_noise(900, 900_000)
    tracks = [(0, 300_000), (300_000, 250_001), (550_001, 349_999), (100_000, 500_000), (891_808, 8192), (899_000, 1000)]
    d_disc = torch.from_numpy(disc).cuda()
    o = np.array([t[0] for t in tracks], np.uint64)
    l = np.array([t[1] for t in tracks], np.uint64)
    got, st = alone.analyze(d_disc, o, l, 2)
    alone.synchronize()
    got, st = got.cpu().numpy(), st.cpu().numpy()
    assert st.tolist() == [0, 0, 0, 0, 0, 1]                  # the last "track" is shorter than 8192 samples
    sep, _ = _run(alone, [disc[a:a + b] for a, b in tracks[:5]])
    assert np.array_equal(got[:5], sep)


# ---------------------------------------------------------------------------------------------
# full-size (BASELINE configs[1] song size) properties
# ---------------------------------------------------------------------------------------------
def test_full_size_properties(ctx, oracle):
    """3-minute songs generated in HBM: (a) the device generator is bit-identical to the oracle's, (b) a
    sample of songs matches the oracle, (c) ZCR is exact, (d) scaling the PCM by 1/2 (exact in binary
    floating point) leaves every feature bit-identical except the two loudness entries, which move by
    exactly 10*log10(1/4) dB."""
    import torch

    n, N = 24, N3MIN
    offs = np.arange(n, dtype=np.uint64) * np.uint64(N)
    lens = np.full(n, N, np.uint64)
    pcm = torch.empty(n * N, dtype=torch.float32, device="cuda")
    ctx.synth_white_noise(pcm, offs, lens, first_song_index=100)
    out, status = ctx.analyze(pcm, offs, lens, 2)
    ctx.synchronize()
    got = out.cpu().numpy()
    assert (status.cpu().numpy() == 0).all() and np.isfinite(got).all()
    tempo_err = []
    for i in (0, 7, 23):
        x = oracle.white_noise(100 + i, N)
        assert np.array_equal(pcm[i * N:(i + 1) * N].cpu().numpy(), x)
        ref = oracle.song_analyze(x)
        err = np.abs(got[i] - ref)
        assert (err[1:] <= _tol(N, 23)[1:]).all(), err
        tempo_err.append(err[0])
        assert got[i][1] == np.float32(2.0 * np.float32(oracle.number_crossings(x)) / np.float32(N) - 1.0)
    _tempo_gate(tempo_err, [0, 7, 23])
    half = pcm * 0.5
    out2, _ = ctx.analyze(half, offs, lens, 2)
    ctx.synchronize()
    got2 = out2.cpu().numpy()
    keep = [i for i in range(23) if i not in (8, 9)]
    assert np.array_equal(got[:, keep], got2[:, keep])
    shift = 2.0 * (10.0 * np.log10(0.25)) / 90.0
    assert np.abs((got2[:, 8:10] - got[:, 8:10]) - shift).max() < 2e-6


# ---------------------------------------------------------------------------------------------
# distances
# ---------------------------------------------------------------------------------------------
def test_distance_literals(bliss, literals):
    # src/playlist.rs:1008-1024, 1081-1110 ; src/lib.rs:272-291 ; src/song/mod.rs:772-807 -- assert_eq on f32
    lit = literals["distances"]
    P = bliss.playlist
    a = np.ones(20, np.float32)
    a[19] = 0
    b = np.zeros(20, np.float32)
    b[16] = 1
    assert np.float32(P.euclidean_distance(a, b)) == np.float32(lit["euclidean"]["value"])
    assert np.float32(P.cosine_distance(a, b)) == np.float32(lit["cosine"]["value"])
    h = np.full(20, 0.5, np.float32)
    assert P.euclidean_distance(h, h) == 0.0 and P.cosine_distance(h, h) == 0.0
    b2 = b.copy()
    b2[0] = 1
    m = np.zeros((20, 20), np.float32)
    m[0, 0] = m[1, 1] = 1
    assert P.mahalanobis_distance_builder(m)(a, b2) == 1.0
    V1, V2 = bliss.FeaturesVersion.Version1, bliss.FeaturesVersion.Version2
    assert np.float32(V1.distance_metric()(np.zeros(20), np.ones(20))) == np.float32(4.47213595)
    assert np.float32(V2.distance_metric()(np.zeros(23), np.ones(23))) == np.float32(3.4999998)
    first = bliss.Analysis(np.zeros(20), V1)
    second = bliss.Analysis(np.ones(20), V1)
    assert np.float32(first.distance(second)) == np.float32(4.472136)
    s1, s2 = bliss.Song(analysis=first), bliss.Song(analysis=second)
    assert np.float32(s1.distance(s2)) == np.float32(4.472136)
    with pytest.raises(RuntimeError, match="Mismatched features version"):
        first.distance(bliss.Analysis(np.zeros(23), V2))


@pytest.mark.parametrize("d", [23, 20, 7, 33])
def test_pairwise_bit_exact_vs_oracle(bliss, oracle, d):
    rng = np.random.default_rng(d)
    A = rng.uniform(-1, 1, (301, d)).astype(np.float32)
    B = rng.uniform(-1, 1, (517, d)).astype(np.float32)
    R = rng.uniform(-1, 1, (d, d)).astype(np.float32)
    psd = (R @ R.T).astype(np.float32)
    diag = np.diag(rng.uniform(0, 1, d).astype(np.float32)).astype(np.float32)
    for metric, m in (("euclidean", None), ("cosine", None), ("mahalanobis", diag), ("mahalanobis", psd)):
        got = bliss.playlist.pairwise_distances(A, B, metric, m)
        ref = oracle.pairwise(A, B, metric, m)
        assert np.array_equal(got, ref, equal_nan=True), (metric, np.nanmax(np.abs(got - ref)))
    # cosine with a zero vector is NaN in the reference too (no guard, src/playlist.rs:76-79)
    Z = np.zeros((1, d), np.float32)
    assert np.isnan(bliss.playlist.pairwise_distances(Z, B[:4], "cosine")).all()


@pytest.mark.parametrize("n", [1, 31, 256, 257, 777, 1000])
def test_pairwise_self_distance_uses_symmetry_bit_exactly(bliss, ctx, oracle, n):
    """A == B takes the symmetric kernel (upper block triangle + LDS-staged transposed stores); it must equal the general
    kernel (A vs a copy of A) and the oracle bit for bit, for every metric, including ragged edge tiles."""
    import torch

    rng = np.random.default_rng(50 + n)
    for d in (23, 20):
        X = rng.uniform(-1, 1, (n, d)).astype(np.float32)
        A = torch.from_numpy(X).cuda()
        M = torch.from_numpy(oracle.feature_weights(2 if d == 23 else 1)).cuda()
        Q = rng.standard_normal((d, d)).astype(np.float32)
        Mfull = torch.from_numpy((Q @ Q.T / d + np.eye(d, dtype=np.float32)).astype(np.float32)).cuda()
        for metric, m in (("euclidean", None), ("cosine", None), ("mahalanobis", M), ("mahalanobis", Mfull)):
            sym = ctx.pairwise(A, A, metric, m).cpu().numpy()
            gen = ctx.pairwise(A, A.clone(), metric, m).cpu().numpy()
            assert np.array_equal(sym.view(np.uint32), gen.view(np.uint32)), (n, d, metric)
            ref = oracle.pairwise(X, X, metric, None if m is None else m.cpu().numpy())
            assert np.array_equal(sym.view(np.uint32), ref.view(np.uint32)), (n, d, metric)
        host = bliss.playlist.pairwise_distances(X, X, "euclidean")          # same host pointer => one device copy
        assert np.array_equal(host.view(np.uint32), oracle.pairwise(X, X, "euclidean").view(np.uint32))


def test_pairwise_full_size_properties(ctx, oracle):
    """BASELINE configs[3]: 100 000 x 100 000 euclidean matrix on one GPU (40 GB): zero diagonal, exact
    symmetry and 10^5 random entries against the oracle, all bit-exact."""
    import torch

    n, d = 100000, 23
    g = torch.Generator(device="cuda").manual_seed(99)
    A = torch.rand((n, d), generator=g, device="cuda", dtype=torch.float32) * 2 - 1
    D = ctx.pairwise(A, A, "euclidean")
    torch.cuda.synchronize()
    assert float(torch.diagonal(D).abs().max()) == 0.0
    rng = np.random.default_rng(5)
    ii = torch.from_numpy(rng.integers(0, n, 100000)).cuda()
    jj = torch.from_numpy(rng.integers(0, n, 100000)).cuda()
    assert torch.equal(D[ii, jj], D[jj, ii])
    Ah = A.cpu().numpy()
    got = D[ii, jj].cpu().numpy()
    ih, jh = ii.cpu().numpy(), jj.cpu().numpy()
    ref = np.array([oracle.euclidean_distance(Ah[i], Ah[j]) for i, j in zip(ih[:20000], jh[:20000])], np.float32)
    assert np.array_equal(got[:20000], ref)
    assert torch.isfinite(D[::997, ::991]).all()
    del D


# ---------------------------------------------------------------------------------------------
# the reference-shaped host API on the GPU
# ---------------------------------------------------------------------------------------------
def test_decoder_trait_bulk_path(bliss, oracle, tmp_path, golden_pcm):
    good = tmp_path / "golden.npy"
    np.save(good, load_golden("s16_mono_22_5kHz.pcm_s16.npy"))
    short = tmp_path / "short.npy"
    np.save(short, np.zeros(100, np.float32))
    missing = str(tmp_path / "nope.npy")
    D = bliss.RawPcmDecoder
    song = D.song_from_path(str(good))
    assert song.path == str(good) and song.features_version == bliss.FeaturesVersion.LATEST
    assert abs(song.duration - len(golden_pcm) / 22050) < 1e-6
    ref = oracle.song_analyze(golden_pcm)
    assert np.abs(song.analysis.as_arr1() - ref).max() <= FEATURE_TOL   # the reference's recording: all 23 within its own tolerance
    assert song.analysis[bliss.AnalysisIndex.Zcr] == pytest.approx(-0.849141, abs=1e-6)
    res = dict(D.analyze_paths([str(good), str(short), missing]))
    assert isinstance(res[str(good)], bliss.Song) and res[str(good)].analysis == song.analysis
    assert res[str(short)] == bliss.AnalysisError("empty or too short song.")
    assert isinstance(res[missing], bliss.DecodingError)
    v1 = D.song_from_path_with_options(str(good), bliss.AnalysisOptions(bliss.FeaturesVersion.Version1))
    assert len(v1.analysis.as_vec()) == 20
    with pytest.raises(RuntimeError, match="incompatible indexes"):
        v1.analysis[bliss.AnalysisIndex.Tempo]


def test_pcm_feed_s16_and_pipelined_host_batches(bliss, ctx, oracle, golden_pcm):
    """SURVEY.md 8 f1 (the pinnable part): s16 mono PCM is widened on the device exactly like FFmpeg's s16 -> flt
    conversion -- the converted golden song has the reference's Adler-32 (src/song/decoder/ffmpeg.rs:455-462) -- and
    the host entry points (f32 and s16, several pipelined groups, pinned or pageable source) agree bit for bit."""
    import ctypes as C
    import zlib

    import torch

    from bliss_rs_amd import _ffi

    s16 = load_golden("s16_mono_22_5kHz.pcm_s16.npy").astype(np.int16)
    dev = ctx.pcm_s16_to_f32(torch.from_numpy(s16).cuda()).cpu().numpy()
    assert zlib.adler32(dev.astype("<f4").tobytes()) == 0x5E01930B
    assert np.array_equal(dev, golden_pcm)
    # odd offsets / tails exercise the scalar path of the conversion kernel
    odd = ctx.pcm_s16_to_f32(torch.from_numpy(s16).cuda()[3:100003]).cpu().numpy()
    assert np.array_equal(odd, golden_pcm[3:100003])

    ref = bliss.analyze_batch([golden_pcm])[0].as_vec()
    got = bliss.analyze_batch([s16])[0].as_vec()
    assert np.array_equal(np.asarray(got), np.asarray(ref))

    # > 512 Mi samples in one call => several groups, transfers overlapped with analysis; ragged, one too-short song
    rng = np.random.default_rng(9)
    n_long = 6
    songs = [oracle.white_noise(100 + i, 100_000_000 if i < n_long else 50_000 + 1000 * i) for i in range(n_long + 4)]
    songs.append(np.zeros(100, np.float32))
    lens = np.array([len(x) for x in songs], np.uint64)
    offs = np.zeros(len(songs), np.uint64)
    offs[1:] = np.cumsum(lens)[:-1]
    total = int(lens.sum())
    L = _ffi.lib()
    hp = C.c_void_p()
    _ffi.check(L.blissgpu_host_alloc(C.byref(hp), total * 4))
    try:
        pinned = np.ctypeslib.as_array(C.cast(hp, C.POINTER(C.c_float)), shape=(total,))
        for x, o in zip(songs, offs):
            pinned[int(o):int(o) + len(x)] = x
        out = np.empty((len(songs), 23), np.float32)
        st = np.empty(len(songs), np.int32)
        _ffi.check(L.blissgpu_analyze_batch(hp, offs.ctypes.data_as(C.POINTER(C.c_uint64)), lens.ctypes.data_as(C.POINTER(C.c_uint64)),
                                            len(songs), 2, out.ctypes.data, st.ctypes.data_as(C.POINTER(C.c_int32))))
    finally:
        _ffi.check(L.blissgpu_host_free(hp))
    assert st.tolist() == [0] * (len(songs) - 1) + [1]
    # the same songs one group at a time (device-resident path) give the same rows
    small, _ = _run(ctx, songs[n_long:n_long + 4])
    assert np.array_equal(out[n_long:n_long + 4], small)
    big, _ = _run(ctx, songs[:1])
    assert np.array_equal(out[0], big[0])
    assert np.isnan(out[-1]).all()


def test_cpp_host_mirror(tmp_path, literals):
    """bliss-rs_amd/csrc/bliss_audio.hpp (the compiled host layer above the C ABI) on the GPU."""
    import subprocess

    from conftest import ROOT

    exe = tmp_path / "test_bliss_audio"
    libdir = os.path.join(ROOT, "bliss-rs_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "cpp", "test_bliss_audio.cpp"), "-o",
                           str(exe), f"-L{libdir}", "-lblissgpu", f"-Wl,-rpath,{libdir}"])
    raw = tmp_path / "golden.s16"
    load_golden("s16_mono_22_5kHz.pcm_s16.npy").astype("<i2").tofile(raw)
    exp = tmp_path / "expected.txt"
    exp.write_text(" ".join(repr(v) for v in literals["analysis_v2_s16_mono_22_5kHz"]["values"]))
    stereo = tmp_path / "stereo.s16"
    load_golden("s16_stereo_22_5kHz.pcm_s16.npy").astype("<i2").tofile(stereo)
    # decoder output at 44.1 kHz (row f1's resampler through the C++ mirror): the row the Python mirror gets, bit for bit
    from conftest import decoded_audio
    import bliss_rs_amd as bliss

    s44, rate = decoded_audio("s32_stereo_44_1_kHz.flac")
    assert rate == 44100 and s44.dtype == np.int32
    raw44 = tmp_path / "stereo44.s32"
    s44.astype("<i4").tofile(raw44)
    exp44 = tmp_path / "expected44.txt"
    exp44.write_text(" ".join("%.9g" % v for v in bliss.Song.analyze_decoded(s44, rate).as_arr1()))
    cue, rate_cue = decoded_audio("testcue.flac")
    assert rate_cue == 44100 and cue.dtype == np.int16 and cue.shape[1] == 2
    rawcue = tmp_path / "testcue.s16"
    cue.astype("<i2").tofile(rawcue)
    expcue = tmp_path / "expected_cue.txt"
    expcue.write_text(" ".join(repr(v) for t in literals["resample"]["cue"]["tracks"] for v in t))
    out = subprocess.run([str(exe), str(raw), str(exp), str(stereo), str(raw44), str(exp44), str(rawcue), str(expcue)],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all checks passed" in out.stdout
