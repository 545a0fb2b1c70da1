"""GPU tests added in round 4 (-m gpu):

  * stage-level parity of the chroma chain -- the filter banks (chroma_filter, src/chroma.rs:197-267), the f64 MFMA contraction
    + column normalisation (chroma_stft, :393-412) and the interval features (:137-188) -- through three debug taps.  The
    reference's own ground truth for this stage, data/chroma.npy (src/chroma.rs:621-639, tolerance 1e-7), now meets the HIP
    path directly; and with the DEVICE spectrogram fed to the oracle's chroma_stft the two sides differ only in the order
    of an f64 sum, so the contraction is held to 1e-12.
  * the exact grid bench.py times: 1024 three-minute songs in ONE chunk (115 200 FFT-8192 workgroups).
  * tests that switch on when the box has two or more GPUs: the node API over real RCCL against the loopback result.
"""
import os
import sys

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N3MIN = 3969000
FEATURE_TOL = 1e-5


@pytest.fixture(scope="module")
def bliss():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import bliss_rs_amd

    return bliss_rs_amd


@pytest.fixture(scope="module")
def tap_ctx(bliss):
    c = bliss.Context(0)
    c.set_option("debug_chroma", 1)
    yield c
    c.close()


def _pack(songs):
    lens = [len(s) for s in songs]
    offs = np.concatenate([[0], np.cumsum([(l + 63) // 64 * 64 for l in lens])[:-1]]).astype(np.uint64)
    buf = np.zeros(int(offs[-1]) + lens[-1] + 64, np.float32)
    for s, o in zip(songs, offs):
        buf[int(o):int(o) + len(s)] = s
    return buf, offs, lens


def _run(ctx, songs, version=2):
    import torch

    buf, offs, lens = _pack(songs)
    out, status = ctx.analyze(torch.from_numpy(buf).cuda(), offs, lens, version)
    ctx.synchronize()
    return out.cpu().numpy(), status.cpu().numpy()


def _slot_tuning(slot):
    """pitch_tuning's return expression (src/chroma.rs:358) for histogram bin `slot`; slot 100 = tuning 0.0 (no peaks)"""
    return 0.0 if slot >= 100 else (-50.0 + (100.0 * 0.01 * float(slot))) / 100.0


# ---------------------------------------------------------------------------------------------
# a6: the 101 filter banks
# ---------------------------------------------------------------------------------------------
def test_filter_bank_slots_vs_oracle(tap_ctx, oracle):
    """every slot of the context's filter bank against chroma_filter(22050, 8192, 12, tuning) of the oracle at 1e-9, the
    tolerance the reference holds chroma_filter to against data/chroma-filter.npy (src/chroma.rs:563-576); the padding
    columns 4097..4127 of the device rows must be zero (the contraction's K loop runs over them)"""
    worst = 0.0
    for slot in range(101):
        raw = tap_ctx.debug_fetch_raw("filter_bank", slot).reshape(12, 4128)
        assert not raw[:, 4097:].any(), slot
        ref = oracle.chroma_filter(22050, 8192, 12, _slot_tuning(slot))
        worst = max(worst, float(np.abs(raw[:, :4097] - ref).max()))
    print("filter bank: max |device - oracle| over 101 slots =", worst)
    assert worst < 1e-9


# ---------------------------------------------------------------------------------------------
# a7 / a8 on the reference's own recording
# ---------------------------------------------------------------------------------------------
def test_device_chroma_matrix_vs_chroma_npy(tap_ctx, oracle, golden_pcm, literals):
    """src/chroma.rs:621-639 on the device: chroma_stft of data/s16_mono_22_5kHz.flac (tuning -0.05) against
    data/chroma.npy at the reference's 1e-7; against the oracle's chroma_stft of the DEVICE's own spectrogram at 1e-12
    (same terms, another summation order); the interval means against the oracle's chroma_interval_features of the device
    matrix at 1e-12, and against the reference's literals for chroma.npy at 1e-7"""
    got, status = _run(tap_ctx, [golden_pcm], 1)
    assert status[0] == 0
    tuning, _ = tap_ctx.last_tuning(1)
    assert tuning[0] == pytest.approx(-0.05, abs=1e-12)       # the value the reference's test passes to chroma_stft
    chroma = tap_ctx.debug_fetch("chroma", 0)                  # [frames][12]
    expected = load_golden("chroma.npy")                       # [12][frames]
    assert chroma.shape == expected.T.shape
    err_npy = float(np.abs(chroma - expected.T).max())
    spec = tap_ctx.debug_fetch("spectrogram", 0).astype(np.float64)     # [frames][4097]: exactly what the contraction read
    ref = oracle.chroma_stft(22050, spec.T, 8192, 12, tuning[0])         # [12][frames]
    err_oracle = float(np.abs(chroma - ref.T).max())
    print(f"chroma matrix: max |device - chroma.npy| = {err_npy:.3e}, max |device - oracle(device spectrogram)| = {err_oracle:.3e}")
    assert err_npy < 1e-7
    assert err_oracle < 1e-12
    interval = tap_ctx.debug_fetch("interval", 0)
    ref_iv = oracle.chroma_interval_features(chroma.T)
    assert np.abs(interval - ref_iv).max() <= 1e-12 * max(1.0, float(np.abs(ref_iv).max())), (interval, ref_iv)
    lit = np.array(literals["chroma_interval_features_of_chroma_npy"]["values"])
    assert np.abs(interval - lit).max() < 1e-7, (interval, lit)
    # and the row the song ends in is ChromaDesc::get_values_version_1 of those means (src/chroma.rs:97-103)
    assert np.abs(got[0][10:20] - (2.0 * interval.astype(np.float32) / np.float32(0.12) - 1.0)).max() < 1e-6


def test_chroma_stage_on_detuned_songs(tap_ctx, oracle):
    """the same three stages on random musical songs whose tuning estimates select other filter banks (one chunk, so every
    song's taps are available): contraction and interval features at 1e-12 against the oracle fed with the device's own
    spectrogram / chroma matrix; padding rows of a 64-frame tile never leak into a song's last frames"""
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import musical_check

    rng = np.random.default_rng(404)
    songs = [musical_check.make_song(rng)[0] for _ in range(12)]
    songs += [oracle.white_noise(31, 64 * 2205 + 17), oracle.white_noise(32, 8192), np.zeros(3 * 22050, np.float32)]
    got, status = _run(tap_ctx, songs, 2)
    assert (status == 0).all()
    tuning, _ = tap_ctx.last_tuning(len(songs))
    slots = set()
    worst_c = worst_i = 0.0
    for i, x in enumerate(songs):
        chroma = tap_ctx.debug_fetch("chroma", i)
        spec = tap_ctx.debug_fetch("spectrogram", i).astype(np.float64)
        assert chroma.shape[0] == spec.shape[0]
        ref = oracle.chroma_stft(22050, spec.T, 8192, 12, float(tuning[i]))
        worst_c = max(worst_c, float(np.abs(chroma - ref.T).max()))
        interval = tap_ctx.debug_fetch("interval", i)
        ref_iv = oracle.chroma_interval_features(chroma.T)
        worst_i = max(worst_i, float(np.abs(interval - ref_iv).max() / max(1e-300, float(np.abs(ref_iv).max()))))
        slots.add(round((float(tuning[i]) + 0.5) * 100))
    print(f"{len(slots)} distinct filter banks; contraction max abs {worst_c:.3e}; interval features max rel {worst_i:.3e}")
    assert len(slots) >= 8
    assert worst_c < 1e-12
    assert worst_i < 1e-12


def test_chroma_taps_need_the_option(bliss, oracle):
    c = bliss.Context(0)
    _run(c, [oracle.white_noise(5, 30000)])
    with pytest.raises(bliss.BlissGpuError):
        c.debug_fetch("chroma", 0)
    with pytest.raises(bliss.BlissGpuError):
        c.debug_fetch("filter_bank", 101)
    c.close()


# ---------------------------------------------------------------------------------------------
# the grid bench.py times
# ---------------------------------------------------------------------------------------------
def test_1024_songs_in_one_chunk(bliss, oracle):
    """BASELINE configs[1] exactly as bench.py runs it: 1024 three-minute songs generated in HBM, FeaturesVersion 2, ONE
    chunk (115 200 FFT-8192 workgroups, 62 016 FFT-512 workgroups).  Run-to-run determinism of all 1024 rows, 16 songs
    spread over the launch grid analysed alone (bit-identical), the oracle on 32 songs spread over the grid (non-tempo
    features at the reference's 1e-5; the tempo histogram is printed and gated as everywhere else)"""
    import torch
    from concurrent.futures import ThreadPoolExecutor

    torch.cuda.empty_cache()
    n, N = 1024, N3MIN
    offs = np.arange(n, dtype=np.uint64) * np.uint64(N)
    lens = np.full(n, N, np.uint64)
    c = bliss.Context(0)
    pcm = torch.empty(n * N, dtype=torch.float32, device="cuda")
    c.synth_white_noise(pcm, offs, lens, first_song_index=0)
    out, status = c.analyze(pcm, offs, lens, 2)
    c.synchronize()
    got = out.cpu().numpy()
    assert c.last_chunks() == 1
    assert (status.cpu().numpy() == 0).all() and np.isfinite(got).all()
    out2, _ = c.analyze(pcm, offs, lens, 2)
    c.synchronize()
    assert np.array_equal(out2.cpu().numpy(), got)
    alone_idx = np.linspace(0, n - 1, 16).astype(int)
    alone, _ = c.analyze(pcm, offs[alone_idx], lens[alone_idx], 2)
    c.synchronize()
    assert np.array_equal(alone.cpu().numpy(), got[alone_idx])
    picks = np.linspace(0, n - 1, 32).astype(int)
    with ThreadPoolExecutor(max_workers=16) as ex:
        ref = np.stack(list(ex.map(lambda i: oracle.song_analyze(oracle.white_noise(int(i), N), 2), picks)))
    err = np.abs(got[picks] - ref)
    tempo = err[:, 0]
    print("non-tempo max |gpu - oracle| =", float(err[:, 1:].max()))
    print("tempo |gpu - oracle| histogram: <=1e-6:", int((tempo <= 1e-6).sum()), " <=1e-5:", int((tempo <= 1e-5).sum()),
          " <=3e-5:", int((tempo <= 3e-5).sum()), " <=1e-4:", int((tempo <= 1e-4).sum()), " of", len(tempo),
          "; over 1e-5:", [(int(picks[k]), float(tempo[k])) for k in np.flatnonzero(tempo > 1e-5)])
    assert (err[:, 1:] <= FEATURE_TOL).all(), err[:, 1:].max(axis=0)
    assert (tempo <= 1e-4).all()
    assert int((tempo > 1e-5).sum()) <= 1          # 3 % of 32 (the measured f32-FFT noise floor)
    c.close()
    del pcm
    torch.cuda.empty_cache()


# ---------------------------------------------------------------------------------------------
# bench.py --gpus N launched as a PLAIN process (no torch.distributed.run, no WORLD_SIZE) must launch its N ranks itself and
# must never print a line whose n_gpus is not --gpus
# ---------------------------------------------------------------------------------------------
def _bench(args, timeout=600):
    import subprocess

    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                          timeout=timeout, cwd=ROOT, env={k: v for k, v in os.environ.items()
                                                          if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})


def test_bench_plain_launch_starts_its_own_ranks():
    import json

    small = ["--steps", "2", "--warmup", "1", "--songs", "24", "--samples", "400000", "--no-cpu-baseline", "--no-pairwise",
             "--no-host-feed", "--no-playlist", "--no-small-calls"]
    out = _bench(["--gpus", "2", "--share-device"] + small)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["songs_total"] == 48 and "share_device" in r
    assert "torch.distributed.run" in out.stderr


def test_bench_refuses_more_gpus_than_the_box_has():
    import torch

    n = torch.cuda.device_count()
    out = _bench(["--gpus", str(n + 1), "--steps", "1", "--warmup", "0"], timeout=300)
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")], out.stdout[-500:]
    assert "refusing" in out.stderr
    # a rank count that does not match --gpus is refused as well (the driver's launch with a wrong WORLD_SIZE)
    import subprocess

    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT,
                         env=dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert out.returncode != 0 and not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]


# ---------------------------------------------------------------------------------------------
# two or more GPUs: the first RCCL N > 1 evidence (skipped with the reason on a one-GPU box)
# ---------------------------------------------------------------------------------------------
def _device_count():
    import torch

    return torch.cuda.device_count()


def _ragged(oracle, n):
    rng = np.random.default_rng(99)
    lens = [int(v) for v in rng.integers(8192, 40 * 22050, n)]
    lens[3] = 8191                          # a too-short song: NaN row, status 1
    return [oracle.white_noise(900 + i, l) for i, l in enumerate(lens)]


def _node_rows(bliss, devices, buf, offs, lens, rank_of):
    """per-rank gathered matrices + the row-block pairwise matrix of one node over `devices` (an ordinal named twice =
    loopback ranks sharing that GPU)"""
    import torch

    world = len(devices)
    pcm = {d: torch.from_numpy(buf).to(f"cuda:{d}") for d in set(devices)}     # every device holds the whole buffer
    node = bliss.Node(world, devices=devices)
    node.analyze_device([pcm[d].data_ptr() for d in devices], offs, lens, rank_of, 2)
    per_rank = [node.features(r) for r in range(world)]
    dist = node.pairwise("euclidean")
    node.close()
    return per_rank, dist


def test_node_over_real_rccl_equals_loopback(bliss, oracle):
    """blissgpu_node_* with one rank per REAL device (ncclCommInitAll + the grouped ncclAllGather over xGMI,
    node.hip:161-167) against the same plan run on loopback ranks of device 0: gathered matrices and row-block pairwise
    bit for bit, ragged shards, one empty rank (src/song/decoder.rs:282-331 is the reference's bulk path this scales).
    On a one-GPU box the loopback half still runs (at the rank count a node would have: the test's own plumbing is not
    first exercised on the node) and the RCCL half is skipped with the reason."""
    n_dev = _device_count()
    world = n_dev if n_dev >= 2 else 8
    songs = _ragged(oracle, 5 * world + 3)
    buf, offs, lens = _pack(songs)
    rank_of = np.arange(len(songs), dtype=np.uint32) % np.uint32(world)
    rank_of[rank_of == world - 1] = 0       # the last rank stays EMPTY: padding only
    per_b, d_b = _node_rows(bliss, [0] * world, buf, offs, lens, rank_of)
    ctx = bliss.Context(0)
    one, status = _run(ctx, songs, 2)
    ctx.close()
    assert status[3] == 1 and np.isnan(per_b[0][3]).all()
    ok = status == 0
    for r in range(world):                  # every rank holds the same full matrix after the gather
        assert np.array_equal(per_b[r], per_b[0], equal_nan=True), r
    assert np.array_equal(per_b[0][ok], one[ok])
    good = np.flatnonzero(ok)
    assert np.array_equal(d_b[np.ix_(good, good)], bliss.playlist.pairwise_distances(one[good], one[good], "euclidean"))
    if n_dev < 2:
        pytest.skip(f"{n_dev} visible GPU: the loopback half passed with {world} ranks; the RCCL all-gather needs two or more "
                    f"devices (runs on a multi-GPU node)")
    per_a, d_a = _node_rows(bliss, list(range(world)), buf, offs, lens, rank_of)
    for r in range(world):
        assert np.array_equal(per_a[r], per_b[0], equal_nan=True), r
    assert np.array_equal(d_a, d_b, equal_nan=True)


def test_threads_over_real_default_contexts(tmp_path):
    """tests/cpp/test_threads with one default context per REAL device: every device reports batches, every row
    bit-identical to the serial run"""
    world = _device_count()
    if world < 2:
        pytest.skip(f"{world} visible GPU: needs one default context per device of a multi-GPU node")
    import subprocess

    from test_gpu_round3 import _kv, _threads_exe

    exe = _threads_exe(tmp_path)
    env = dict(os.environ)
    env.pop("BLISSGPU_DEFAULT_DEVICES", None)
    r = subprocess.run([str(exe), "32", "16"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout          # every threaded row bit-identical to the serial run
    kv = _kv(r.stdout)
    assert int(kv["default_devices"]) == world
    served = [int(kv[f"default_device_{k}_batches"]) for k in range(world)]
    assert all(v > 0 for v in served), served


# ---------------------------------------------------------------------------------------------
# a default context whose device cannot give a context is retired; the others take its traffic
# ---------------------------------------------------------------------------------------------
def test_front_retires_a_seat_whose_device_is_missing(tmp_path):
    """BLISSGPU_DEFAULT_DEVICES=0,99: the second default context cannot be created (no HIP device 99).  16 threads x 32
    single-song calls must all succeed, bit-identical to the serial run, through seat 0 alone -- before round 4 every batch
    that drew the broken seat failed, and that seat, free again at once, kept drawing traffic."""
    import subprocess

    from test_gpu_round3 import _kv, _threads_exe

    exe = _threads_exe(tmp_path)
    out = subprocess.run([str(exe), "16", "32"], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, BLISSGPU_DEFAULT_DEVICES="0,99"))
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all checks passed" in out.stdout
    kv = _kv(out.stdout)
    assert kv["default_devices"] == "2" and kv["default_device_1_hip_ordinal"] == "99"
    assert int(kv["default_device_0_batches"]) > 0 and int(kv["default_device_1_batches"]) == 0, out.stdout


def test_front_fails_loudly_when_no_seat_is_usable():
    """every default device is missing: the call fails with BLISSGPU_ERR_NO_DEVICE, says which seats were retired and why,
    and a second call fails the same way at once (nothing blocks, nothing is retried for ever)"""
    import subprocess

    code = (
        "import ctypes as C, numpy as np, sys, time\n"
        "sys.path.insert(0, %r)\n"
        "from bliss_rs_amd import _ffi\n"
        "L = _ffi.lib()\n"
        "x = np.zeros(30000, np.float32); out = np.zeros(23, np.float32); st = C.c_int32(0)\n"
        "for k in range(2):\n"
        "    t = time.time()\n"
        "    rc = L.blissgpu_analyze(x.ctypes.data_as(C.POINTER(C.c_float)), len(x), 2, out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(st))\n"
        "    print(rc, round(time.time() - t, 1) < 30, L.blissgpu_last_error().decode())\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, BLISSGPU_DEFAULT_DEVICES="98,99"))
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().splitlines()
    assert len(lines) == 2
    for ln in lines:
        assert ln.startswith("1 True "), ln                       # BLISSGPU_ERR_NO_DEVICE, promptly
        assert "retired, retired" in ln and "HIP device 9" in ln, ln


# ---------------------------------------------------------------------------------------------
# the hand-written stable radix sort behind closest_to_songs (src/playlist.rs:256-270: sort_by_cached_key is stable)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 63, 2048, 2049, 100000, 1 << 20])
def test_closest_to_songs_sort_is_stable_on_heavy_ties(bliss, n):
    """one-dimensional songs on an integer grid: thousands of candidates share every distance, so the order is decided by
    stability alone; sizes around the sort's tile (2048 pairs) and up to 2^20; negative zero and large distances in the
    keys' high bytes"""
    import torch

    rng = np.random.default_rng(n)
    x = rng.integers(-40, 41, n).astype(np.float32)
    x[rng.random(n) < 0.01] *= 1e6          # a few far candidates: all four key bytes are in play
    X = np.zeros((n, 3), np.float32)
    X[:, 0] = x
    seed = np.zeros((1, 3), np.float32)
    ctx = bliss.Context(0)
    order, dist = ctx.closest_to_songs(torch.from_numpy(seed).cuda(), torch.from_numpy(X).cuda(), "euclidean", None, return_distances=True)
    order, dist = order.cpu().numpy(), dist.cpu().numpy()
    assert np.array_equal(dist, np.abs(x))                       # sqrt(x^2) is exact on these values
    assert np.array_equal(order, np.argsort(np.abs(x), kind="stable"))
    ctx.close()


# ---------------------------------------------------------------------------------------------
# BLISSGPU_OPT_TAIL_SPLIT: the tuning estimate + contraction of a one-chunk batch in pieces (a schedule, not an algorithm)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("pieces", [2, 3, 8])
def test_split_tail_gives_the_same_rows(bliss, oracle, pieces):
    """ragged songs (a too-short one, a two-frame one, a silent one among them) so that the pieces cut at uneven song and
    tile boundaries; rows and per-song statuses bit-identical to the unsplit schedule, taps of a song in the LAST piece
    still served"""
    rng = np.random.default_rng(5)
    songs = [oracle.white_noise(300 + i, int(n)) for i, n in enumerate(rng.integers(8192, 30 * 22050, 37))]
    songs[5] = np.zeros(4000, np.float32)
    songs[9] = oracle.white_noise(399, 8192)
    songs[20] = np.zeros(5 * 22050, np.float32)
    ref_ctx = bliss.Context(0)
    ref, ref_status = _run(ref_ctx, songs, 2)
    ref_ctx.close()
    c = bliss.Context(0)
    c.set_option("tail_split", pieces)
    c.set_option("debug_chroma", 1)
    got, status = _run(c, songs, 2)
    assert c.last_chunks() == 1
    assert status.tolist() == ref_status.tolist()
    assert np.array_equal(got, ref, equal_nan=True)
    shortest = int(np.argmin([len(s) if len(s) >= 8192 else 1 << 30 for s in songs]))   # sorted last: in the last piece
    assert c.debug_fetch("chroma", shortest).shape[1] == 12 and np.isfinite(c.debug_fetch("interval", shortest)).all()
    c.close()
