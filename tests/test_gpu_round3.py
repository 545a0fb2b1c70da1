"""GPU tests (-m gpu) of round 3: the N > 1 code of the node API on LOOPBACK ranks (several contexts on the one GPU of
the test box: the plan, the padded rank-major gather buffer with perm = -1 slots, the scatter kernel and the row-block
pairwise are the code an 8-GPU node runs; only the transport -- device-to-device copies instead of ncclAllGather --
differs), one GPU's FULL share of BASELINE configs[2] and configs[4], the default contexts of the single-song front
on more than one context, and the chunk planner's song-count cap."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

FEATURE_TOL = 1e-5   # the reference's own tolerance (src/song/mod.rs:582-590)


@pytest.fixture(scope="module")
def bliss():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import bliss_rs_amd

    return bliss_rs_amd


@pytest.fixture(scope="module")
def ctx(bliss):
    return bliss.Context(0)


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


# ---------------------------------------------------------------------------------------------
# node API with 8 ranks (loopback): every line of gather_rows / scatter_rows_kernel / the row-block pairwise
# ---------------------------------------------------------------------------------------------
def _ragged_songs(oracle, n=21):
    songs = [oracle.white_noise(900 + i, (2 + (5 * i) % 11) * 22050 + 13 * i) for i in range(n)]
    songs[4] = oracle.white_noise(904, 4000)    # too short: its row is NaN on every rank
    return songs


def test_node_eight_loopback_ranks_hand_built_ragged_shards(bliss, ctx, oracle):
    """rank_of_song is given by hand: ragged shards, one EMPTY rank, a rank that holds only the too-short song -- so the
    gather buffer is mostly padding (perm = -1) -- and every rank must end up with the same full matrix, equal bit for
    bit to the single-context analysis."""
    import torch

    songs = _ragged_songs(oracle)
    n = len(songs)
    ref, ref_st = _run(ctx, songs)
    ok = ref_st == 0
    buf, offs, lens = _pack(songs)
    ranks = np.array([0, 0, 0, 0, 6, 2, 2, 2, 3, 4, 4, 4, 4, 5, 5, 0, 7, 7, 7, 0, 2], np.uint32)   # rank 1: empty, rank 6: NaN row only
    assert len(ranks) == n and 1 not in ranks
    d_pcm = torch.from_numpy(buf).cuda()
    node = bliss.Node(8, devices=[0] * 8)
    for version in (2, 1):
        want = ref if version == 2 else _run(ctx, songs, 1)[0]
        node.analyze_device([d_pcm.data_ptr()] * 8, offs, lens, ranks, version)
        for r in range(8):
            got = node.features(r)
            assert np.array_equal(got[ok], want[ok]), (version, r)
            assert np.isnan(got[~ok]).all()
    # row-block sharded pairwise over the 8 ranks (21 rows: blocks of 3, 3, 3, 3, 3, 2, 2, 2) = the one-context matrix
    node.analyze_device([d_pcm.data_ptr()] * 8, offs, lens, ranks, 2)
    good = np.flatnonzero(ok)
    for metric in ("euclidean", "cosine"):
        D = node.pairwise(metric)
        want = bliss.playlist.pairwise_distances(ref[good], ref[good], metric)
        assert np.array_equal(D[np.ix_(good, good)], want), metric
    M = np.diag(np.linspace(0.25, 2.0, 23)).astype(np.float32)
    D = node.pairwise("mahalanobis", M)
    assert np.array_equal(D[np.ix_(good, good)], bliss.playlist.pairwise_distances(ref[good], ref[good], "mahalanobis", M))
    assert [node.row_block(n, r) for r in range(8)] == [(0, 3), (3, 6), (6, 9), (9, 12), (12, 15), (15, 17), (17, 19), (19, 21)]
    node.close()


def test_node_loopback_plan_path_equals_the_context_path(bliss, ctx, oracle):
    """the host form: the library's own plan (blissgpu_shard_plan) over 2, 4 and 8 ranks, one host thread per rank"""
    from bliss_rs_amd.shard import shard_plan, shard_songs

    songs = _ragged_songs(oracle, 19)
    ref, ref_st = _run(ctx, songs)
    ok = ref_st == 0
    buf, offs, lens = _pack(songs)
    for world in (2, 4, 8):
        node = bliss.Node(world, devices=[0] * world)
        out, status = node.analyze(buf, offs, lens, 2)
        assert status.tolist() == ref_st.tolist()
        assert np.array_equal(out[ok], ref[ok]), world
        for r in range(world):
            got = node.features(r)
            assert np.array_equal(got[ok], ref[ok]) and np.isnan(got[~ok]).all(), (world, r)
        planned = node.shard(lens)
        assert np.array_equal(planned, shard_plan(lens, world))
        for r, s in enumerate(shard_songs(lens, world)):
            assert np.array_equal(np.flatnonzero(planned == r), s)
        node.close()


def test_node_argument_validation(bliss):
    from bliss_rs_amd import _ffi

    L = _ffi.lib()
    node = bliss.Node(2, devices=[0, 0])
    lens = np.array([9000, 9000], np.uint64)
    offs = np.array([0, 9024], np.uint64)
    import torch

    d_pcm = torch.zeros(20000, dtype=torch.float32, device="cuda")
    for bad in (2, 0x80000000, 0xFFFFFFFF):      # out of range, including values that are negative as int
        with pytest.raises(_ffi.BlissGpuError):
            node.analyze_device([d_pcm.data_ptr()] * 2, offs, lens, np.array([0, bad], np.uint32), 2)
    assert node.row_block(10, -1) == (10, 10) and node.row_block(10, 2) == (10, 10)
    node.close()
    h = C.c_void_p()
    assert L.blissgpu_node_create(1, (C.c_int * 1)(99), C.byref(h)) == _ffi.ERR_NO_DEVICE


# ---------------------------------------------------------------------------------------------
# the single-song front on MORE THAN ONE default context (src/song/decoder.rs:299-329: N worker threads each calling
# Song::analyze): on an 8-GPU node there is one default context per device; on the one-GPU test box two contexts on
# device 0 stand in for two devices
# ---------------------------------------------------------------------------------------------
def _threads_exe(tmp_path):
    exe = tmp_path / "test_threads"
    libdir = os.path.join(ROOT, "bliss-rs_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", os.path.join(ROOT, "tests", "cpp", "test_threads.cpp"),
                           "-o", str(exe), f"-L{libdir}", "-lblissgpu", f"-Wl,-rpath,{libdir}"])
    return exe


def _kv(stdout):
    return {l.split()[0]: l.split()[1] for l in stdout.splitlines() if len(l.split()) == 2}


def test_single_song_front_spreads_over_two_default_contexts(tmp_path):
    exe = _threads_exe(tmp_path)
    env = dict(os.environ, BLISSGPU_DEFAULT_DEVICES="0,0")
    out = subprocess.run([str(exe), "16", "32"], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all checks passed" in out.stdout          # every threaded row bit-identical to the serial run
    kv = _kv(out.stdout)
    assert kv["default_devices"] == "2"
    assert kv["default_device_0_hip_ordinal"] == "0" and kv["default_device_1_hip_ordinal"] == "0"
    assert int(kv["default_device_0_batches"]) > 0 and int(kv["default_device_1_batches"]) > 0, out.stdout
    print(out.stdout)


def test_default_contexts_follow_the_visible_devices(tmp_path, bliss):
    import torch
    from bliss_rs_amd import _ffi

    exe = _threads_exe(tmp_path)
    out = subprocess.run([str(exe), "4", "8"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    kv = _kv(out.stdout)
    assert int(kv["default_devices"]) == torch.cuda.device_count()
    L = _ffi.lib()
    assert L.blissgpu_default_device_count() == torch.cuda.device_count() and L.blissgpu_default_device(0) == 0
    assert L.blissgpu_default_device(99) == -1


# ---------------------------------------------------------------------------------------------
# BASELINE configs[4] and configs[2] at the size ONE GPU owns on the 8-GPU node
# ---------------------------------------------------------------------------------------------
def _config4_lengths(n_total=50000, seed=20260927):
    """bench.py's mixed_lengths: durations uniform 30 s - 10 min at 22 050 Hz, one seeded draw for the whole corpus"""
    rng = np.random.default_rng(seed)
    return rng.integers(30 * 22050, 600 * 22050 + 1, n_total).astype(np.uint64)


def _tol(n_samples, d, tempo_tol, flips=0):
    """the reference's 1e-5 on every feature; `flips` frames on another rolloff bin on top only where a test demonstrates
    the plateau case (the kernel counts the bins in the reference's summation order: white noise and recordings get none)"""
    tol = np.full(d, FEATURE_TOL)
    tol[0] = tempo_tol
    n_t = (n_samples - 512) // 128 + 1
    flip = 2.0 * (22050.0 / 512.0) / 11025.0 / n_t      # one rolloff bin of one frame (a per-frame integer decision)
    tol[4] += flips * flip
    tol[5] += flips * flip * np.sqrt(max(n_t, 1)) * 0.5
    return tol


def _oracle_rows(oracle, picks, gen_index, lens, version):
    """oracle rows of the picked songs, a few host threads (ctypes releases the GIL)"""
    from concurrent.futures import ThreadPoolExecutor

    def one(i):
        return oracle.song_analyze(oracle.white_noise(int(gen_index[i]), int(lens[i])), version)

    with ThreadPoolExecutor(max_workers=min(16, len(picks))) as ex:
        return np.stack(list(ex.map(one, picks)))


def test_config4_one_gpu_share_full_count(bliss, oracle):
    """configs[4]: rank 0's share of the 50 000-song 30 s - 10 min corpus under the 8-rank plan -- ~6 250 songs, ~174 GB
    of PCM resident, length-bucketed chunks streamed through the two workspace slots.  Checks that do not need 6 250
    oracle runs: run-to-run determinism, bit-identity under another chunking and with songs analysed alone, exact ZCR;
    plus the oracle on 16 songs spread from the shortest to the longest bucket."""
    import torch
    from bliss_rs_amd.shard import shard_plan

    torch.cuda.empty_cache()
    all_lens = _config4_lengths()
    ranks = shard_plan(all_lens, 8)
    global_idx = np.flatnonzero(ranks == 0)
    lens = all_lens[global_idx]
    n = len(lens)
    assert 6000 < n < 6500
    padded = (lens + np.uint64(63)) // np.uint64(64) * np.uint64(64)
    offs = np.zeros(n, np.uint64)
    offs[1:] = np.cumsum(padded)[:-1]
    pcm_bytes = int(padded.sum()) * 4
    slot = 24 << 30
    free_b, _ = torch.cuda.mem_get_info()
    assert free_b > pcm_bytes + 2 * slot * 1.15 + (6 << 30), f"{free_b / 2**30:.0f} GiB free: the full share does not fit"
    c = bliss.Context(0)
    c.set_workspace_limit(slot)
    pcm = torch.empty(int(padded.sum()) + 64, dtype=torch.float32, device="cuda")
    c.synth_white_noise(pcm, offs, lens, song_index=global_idx)
    out, status = c.analyze(pcm, offs, lens, 2)
    c.synchronize()
    got = out.cpu().numpy()
    chunks = c.last_chunks()
    assert (status.cpu().numpy() == 0).all() and np.isfinite(got).all()
    assert chunks >= 8, chunks                                   # ~37 MB of workspace per three-minute equivalent
    # run-to-run determinism of the whole share
    out2, _ = c.analyze(pcm, offs, lens, 2)
    c.synchronize()
    assert np.array_equal(out2.cpu().numpy(), got)
    # another chunking (half the slot: about twice the chunks) -- bit-identical
    c.set_workspace_limit(slot // 2)
    out3, _ = c.analyze(pcm, offs, lens, 2)
    c.synchronize()
    assert c.last_chunks() > chunks
    assert np.array_equal(out3.cpu().numpy(), got)
    # 16 songs spread over the length buckets (shortest ... longest of the share): alone, and against the oracle
    order = np.argsort(lens, kind="stable")
    picks = order[np.linspace(0, n - 1, 16).astype(int)]
    alone, st = c.analyze(pcm, offs[picks], lens[picks], 2)
    c.synchronize()
    assert np.array_equal(alone.cpu().numpy(), got[picks])
    ref = _oracle_rows(oracle, picks, global_idx, lens, 2)
    listed = []
    for k, i in enumerate(picks):
        x_n = int(lens[i])
        err = np.abs(got[i] - ref[k])
        assert (err[1:] <= _tol(x_n, 23, 1.0)[1:]).all(), (int(i), x_n, err)
        if err[0] > 1e-5:
            listed.append((int(global_idx[i]), x_n, float(err[0])))
        assert err[0] <= 1e-4, (int(i), x_n, float(err[0]))
        # exact ZCR from the resident PCM
        x = pcm[int(offs[i]):int(offs[i]) + x_n].cpu().numpy()
        assert got[i][1] == np.float32(2.0 * np.float32(oracle.number_crossings(x)) / np.float32(x_n) - 1.0)
    # tempo at the reference's own 1e-5 (src/song/mod.rs:582-590): every song above it is listed, and there may be at
    # most one among the 16 (the measured fraction and the f32-FFT noise floor)
    print("tempo over 1e-5:", listed)
    assert len(listed) <= 1, listed
    c.close()
    del pcm
    torch.cuda.empty_cache()


def test_config2_one_gpu_share_full_count_through_the_node_api(bliss, oracle):
    """configs[2] at full count: a 10 000-song library on an 8-rank node, 20-dim rows (FeaturesVersion 1), the padded
    gather of 8 x 1 250 rows, the scatter to 10 000 global rows and the row-block pairwise (1 250 x 10 000 per rank).
    The test box has one GPU: the 8 ranks are loopback contexts on it and the library is 8 references to each of 1 250
    resident three-minute songs (19.8 GB of PCM, one GPU's share), so every rank analyses a full 1 250-song share."""
    import torch
    from bliss_rs_amd.shard import shard_plan

    torch.cuda.empty_cache()
    N, share, world = 3969000, 1250, 8
    n_total = share * world
    offs1 = np.arange(share, dtype=np.uint64) * np.uint64(N)
    lens1 = np.full(share, N, np.uint64)
    c = bliss.Context(0)
    c.set_workspace_limit(8 << 30)
    pcm = torch.empty(share * N + 64, dtype=torch.float32, device="cuda")
    c.synth_white_noise(pcm, offs1, lens1, first_song_index=0)
    ref_rows, st = c.analyze(pcm, offs1, lens1, 1)
    c.synchronize()
    ref_rows = ref_rows.cpu().numpy()
    assert (st.cpu().numpy() == 0).all() and ref_rows.shape == (share, 20)
    # the library: song i is resident song i % share
    offs = np.tile(offs1, world)
    lens = np.tile(lens1, world)
    ranks = shard_plan(lens, world)
    assert np.array_equal(ranks, np.arange(n_total) % world)        # equal lengths: round-robin, 1 250 each
    node = bliss.Node(world, devices=[0] * world)
    for r in range(world):
        node.ctx_set_workspace_limit(r, 6 << 30)
    node.analyze_device([pcm.data_ptr()] * world, offs, lens, ranks, 1)
    want = np.tile(ref_rows, (world, 1))
    for r in range(world):
        assert np.array_equal(node.features(r), want), r          # eight contexts, every row bit-identical
    # oracle on 16 songs of the share
    picks = np.linspace(0, share - 1, 16).astype(int)
    ref = _oracle_rows(oracle, picks, np.arange(share), lens1, 1)
    over = []
    for k, i in enumerate(picks):
        err = np.abs(ref_rows[i] - ref[k])
        assert (err[1:] <= _tol(N, 20, 1.0)[1:]).all(), (int(i), err)
        assert err[0] <= 1e-4
        if err[0] > 1e-5:
            over.append((int(i), float(err[0])))
    assert len(over) <= 1, over
    # row-block pairwise over the 8 ranks = the one-context matrix of the 1 250 songs, tiled 8 x 8
    D = node.pairwise("euclidean")
    D0 = c.pairwise(torch.from_numpy(ref_rows).cuda(), torch.from_numpy(ref_rows).cuda(), "euclidean").cpu().numpy()
    assert D.shape == (n_total, n_total)
    for bi in range(world):
        for bj in range(world):
            assert np.array_equal(D[bi * share:(bi + 1) * share, bj * share:(bj + 1) * share], D0), (bi, bj)
    assert (np.diag(D) == 0).all() and np.array_equal(D, D.T)
    node.close()
    c.close()
    del pcm
    torch.cuda.empty_cache()


def test_more_than_65535_songs_in_one_batch(bliss, oracle):
    """70 000 one-second clips fit one chunk's workspace; the beat tracker's (run, song) grid does not address that many
    songs, so the planner closes a chunk at 65 535 songs."""
    import torch

    n, N = 70000, 22050
    c = bliss.Context(0)
    offs = np.arange(n, dtype=np.uint64) * np.uint64(22080)       # 64-sample aligned songs
    lens = np.full(n, N, np.uint64)
    pcm = torch.empty(n * 22080 + 64, dtype=torch.float32, device="cuda")
    c.synth_white_noise(pcm, offs, lens, first_song_index=5000)
    out, status = c.analyze(pcm, offs, lens, 2)
    c.synchronize()
    got = out.cpu().numpy()
    assert (status.cpu().numpy() == 0).all() and np.isfinite(got).all()
    assert c.last_chunks() >= 2
    picks = np.array([0, 1, 65534, 65535, 65536, 69999])
    alone, _ = c.analyze(pcm, offs[picks], lens[picks], 2)
    c.synchronize()
    assert np.array_equal(alone.cpu().numpy(), got[picks])
    ref = oracle.song_analyze(oracle.white_noise(5000 + 65535, N))
    err = np.abs(got[65535] - ref)
    assert (err[1:] <= _tol(N, 23, 1.0)[1:]).all() and err[0] <= 1e-4, err
    c.close()


# ---------------------------------------------------------------------------------------------
# distances at the edges of f32: denormal squares, overflow to inf, zeros / duplicates / negated rows
# ---------------------------------------------------------------------------------------------
def test_pairwise_is_bit_exact_at_the_edges_of_f32(bliss, oracle):
    rng = np.random.default_rng(5)
    d = 23
    base = rng.uniform(-1, 1, (300, d)).astype(np.float32)
    cases = {
        "squares are denormal": base * np.float32(1e-22),
        "tiny": base * np.float32(1e-19),
        "squares near overflow": base * np.float32(1e18),
        "squares overflow": base * np.float32(1e20),
        "zeros, duplicates, negated rows": np.vstack([np.zeros((3, d), np.float32), base[:5], base[:5], -base[:5]]),
        "mixed magnitudes": base * (np.float32(10.0) ** rng.integers(-12, 12, (300, 1)).astype(np.float32)),
    }
    M = np.diag(rng.uniform(0.1, 2, d).astype(np.float32)).astype(np.float32)
    R = rng.uniform(-1, 1, (d, d)).astype(np.float32)
    P = (R @ R.T).astype(np.float32)
    for name, X in cases.items():
        for metric, m in (("euclidean", None), ("cosine", None), ("mahalanobis", M), ("mahalanobis", P)):
            with np.errstate(all="ignore"):
                got = bliss.playlist.pairwise_distances(X, X[:97], metric, m)
                ref = oracle.pairwise(X, X[:97], metric, m)
            assert np.array_equal(got, ref, equal_nan=True), (name, metric)
        with np.errstate(all="ignore"):   # the self-distance kernel (A is B): mirrored blocks
            got = bliss.playlist.pairwise_distances(X, X, "euclidean")
            assert np.array_equal(got, oracle.pairwise(X, X, "euclidean"), equal_nan=True), name


# ---------------------------------------------------------------------------------------------
# the reference's REAL audio (its golden song, its piano recording) cut, mixed and stretched into a battery: tonal
# material with real tunings, real beats and real silences, not white noise
# ---------------------------------------------------------------------------------------------
def test_real_audio_battery_vs_oracle(bliss, ctx, oracle, golden_pcm, piano_pcm, literals):
    g = golden_pcm.astype(np.float32)
    p = piano_pcm.astype(np.float32)
    n = min(len(g), len(p))
    songs = {
        "golden": g,
        "piano": p,
        "golden x4": np.tile(g, 4),
        "golden reversed": g[::-1].copy(),
        "piano then golden": np.concatenate([p, g]),
        "mix 0.5 golden + 0.5 piano": (np.float32(0.5) * g[:n] + np.float32(0.5) * p[:n]).astype(np.float32),
        "2 s of silence, then golden": np.concatenate([np.zeros(2 * 22050, np.float32), g]),
        "golden, then 3 s of silence": np.concatenate([g, np.zeros(3 * 22050, np.float32)]),
        "golden second half": g[len(g) // 2:].copy(),
        "piano at 1e-3": (p * np.float32(1e-3)).astype(np.float32),
        "golden, first 8192 samples": g[:8192].copy(),
        "golden, 60 000 samples from 50 000": g[50000:110000].copy(),
    }
    names = list(songs)
    for version in (2, 1):
        got, status = _run(ctx, [songs[k] for k in names], version)
        assert (status == 0).all()
        tuning, n_bpms = ctx.last_tuning(len(names))
        over = []
        for i, k in enumerate(names):
            ref = oracle.song_analyze(songs[k], version)
            _, otuning = oracle.chroma_desc(songs[k])
            err = np.abs(got[i] - ref)
            assert abs(tuning[i] - otuning) < 1e-12, (k, tuning[i], otuning)          # the tuning estimate is exact
            assert (err[1:] <= _tol(len(songs[k]), len(ref), 1.0)[1:]).all(), (k, version, err)
            assert err[0] <= 1e-4, (k, version, float(err[0]))
            if err[0] > 1e-5:
                over.append((k, float(err[0])))
        print("tempo over 1e-5:", over)
        assert len(over) <= 1, over
        # the golden song keeps the reference's own literals when it sits inside a batch of other material
        key = "analysis_v2_s16_mono_22_5kHz" if version == 2 else "analysis_v1_s16_mono_22_5kHz"
        exp = np.array(literals[key]["values"], np.float32)
        assert np.abs(got[0] - exp).max() < literals[key]["tol"], (version, got[0] - exp)


# ------------------------------------------------------------------------------------------------
# bench.py launched the way the driver launches N > 1 (torch.distributed.run, one rank per GPU) -- on this 1-GPU box the
# ranks share device 0 and the collectives go over gloo (--share-device); everything else is the N > 1 code as it runs
# on a node: per-rank shares, the gathered matrix, the row-block distance kernel, max-over-ranks timing, one JSON line
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("config", ["batch", "library", "mixed"])
def test_bench_two_ranks_sharing_the_device(config):
    import json
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--config", config, "--songs", "24", "--samples", "400000", "--share-device", "--no-cpu-baseline", "--no-pairwise",
           "--no-host-feed", "--no-playlist", "--no-small-calls"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                      # rank 0 prints ONE line
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["warmup"] == 1
    assert r["config"]["songs_total"] == 48                          # both ranks' shares were summed
    assert r["scaling"] == ("strong" if config == "library" else "weak")
    assert r["value"] > 0 and abs(r["value"] - 48 * 2 / (r["ms_per_step"] * 2e-3)) < 0.01 * r["value"]
    assert r["roofline"]["frac"] > 0 and "share_device" in r


# ---------------------------------------------------------------------------------------------
# a two-hour song (a DJ set, an audio book): 158 760 000 samples, 1.24 million timbral frames, 72 000 chroma frames in ONE
# song, twelve times the longest song of the mixed corpus -- beside a 30-second song in the same batch
# ---------------------------------------------------------------------------------------------
def test_two_hour_song_vs_oracle(bliss, oracle):
    import torch

    c = bliss.Context(0)
    lens = np.array([2 * 3600 * 22050, 30 * 22050 + 7], np.uint64)
    offs = np.array([0, (int(lens[0]) + 63) // 64 * 64], np.uint64)
    pcm = torch.empty(int(offs[1]) + int(lens[1]) + 64, dtype=torch.float32, device="cuda")
    c.synth_white_noise(pcm, offs, lens, first_song_index=7777)
    out, status = c.analyze(pcm, offs, lens, 2)
    c.synchronize()
    got = out.cpu().numpy()
    assert (status.cpu().numpy() == 0).all() and np.isfinite(got).all()
    # the device generator and the oracle's are the same function of (song index, sample index)
    x = oracle.white_noise(7777, int(lens[0]))
    assert np.array_equal(pcm[:1 << 20].cpu().numpy(), x[:1 << 20])
    assert np.array_equal(pcm[int(lens[0]) - 4096:int(lens[0])].cpu().numpy(), x[-4096:])
    for i, sig in enumerate((x, oracle.white_noise(7778, int(lens[1])))):
        err = np.abs(got[i] - oracle.song_analyze(sig))
        assert (err[1:] <= _tol(len(sig), 23, 1.0)[1:]).all() and err[0] <= 1e-4, (i, err)
    # and alone = in company
    alone, _ = c.analyze(pcm, offs[:1], lens[:1], 2)
    c.synchronize()
    assert np.array_equal(alone.cpu().numpy()[0], got[0])
    c.close()


# ---------------------------------------------------------------------------------------------
# spectral rolloff is a bin COUNT: discontinuous in the sums it is derived from.  The FFT-512 kernel sums in its own order
# and hands every frame it cannot prove (a running energy within worst-case rounding of the 95 % threshold) to
# rolloff_fix_kernel, which repeats the reference's loop (src/aubio.rs:36-58) literally.
# ---------------------------------------------------------------------------------------------
def _musical_songs(oracle, n, seed):
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import musical_check

    rng = np.random.default_rng(seed)
    return [musical_check.make_song(rng)[0] for _ in range(n)] + [oracle.white_noise(700 + i, 45 * 22050) for i in range(6)]


def test_rolloff_bins_follow_the_reference_order(bliss, oracle):
    """Per-frame rolloff against the oracle on random musical signals (detuned notes, bursts, noise floors: spectra with
    plateaus, where a flipped decision moves the bin by the distance between two partials) and white noise.  What is left
    is the oracle's own sensitivity to FFT rounding: 10 frames in 1.84 million between its f32 and f64 transforms; the
    kernel's own summation order alone gave 168 (tests/tools/rolloff_flips.py)."""
    from concurrent.futures import ThreadPoolExecutor

    songs = _musical_songs(oracle, 60, 2)
    c = bliss.Context(0)
    _run(c, songs, 2)
    gpu = [c.debug_fetch("rolloff", i) for i in range(len(songs))]
    with ThreadPoolExecutor(32) as ex:
        ref = list(ex.map(lambda x: oracle.SpectralDesc().run(x).series()[1], songs))
    frames = sum(len(r) for r in ref)
    flips = sum(int((np.abs(g - r) > 1e-3).sum()) for g, r in zip(gpu, ref))
    print(f"rolloff: {flips} of {frames} frames differ from the oracle")
    assert frames > 350000 and flips <= 8, (flips, frames)     # the kernel's order alone: ~ 35 on these songs
    c.close()


def test_rolloff_guard_agrees_with_the_exact_pass(bliss, oracle):
    """Every frame through the reference-order pass (a test option) must give the rows and the per-frame series the
    guarded kernel gives: the guard's proof, checked on ~ 1.2 million frames of musical signals and white noise."""
    songs = _musical_songs(oracle, 40, 5) + [oracle.white_noise(800 + i, 180 * 22050) for i in range(24)]
    c = bliss.Context(0)
    rows, _ = _run(c, songs, 2)
    series = [c.debug_fetch("rolloff", i) for i in range(len(songs))]
    c.set_option("rolloff_exact_all", 1)
    rows_all, _ = _run(c, songs, 2)
    series_all = [c.debug_fetch("rolloff", i) for i in range(len(songs))]
    assert sum(len(s) for s in series) > 1000000
    assert all(np.array_equal(a, b) for a, b in zip(series, series_all))
    assert np.array_equal(rows, rows_all)
    c.close()


def test_random_musical_songs_vs_oracle(bliss, oracle):
    """90 seeded random musical songs (tests/tools/musical_check.py: detuned note sequences, bursts at 60 - 200 bpm, noise
    floors, gains from 1e-3 to 1; half of them with silent gaps, DC offsets, fades, or barely over the minimum length):
    the tuning estimate is the oracle's on every song, tempo is within the reference's 1e-5 on every song, and so are 20
    of the 22 other features.  The two flatness features of a noise-free tonal song are a geometric mean over bins that
    hold nothing but FFT rounding noise: they are held to the oracle's distance from ITSELF when only its FFT precision
    changes (f32 -> f64), which is 1e-4 and more on such songs."""
    from concurrent.futures import ThreadPoolExecutor

    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import musical_check

    rng = np.random.default_rng(21)
    songs = [musical_check.make_song(rng, mods=(i % 2 == 1))[0] for i in range(90)]
    c = bliss.Context(0)
    got, status = _run(c, songs, 2)
    tuning, _ = c.last_tuning(len(songs))
    assert (status == 0).all()
    with ThreadPoolExecutor(32) as ex:
        ref = np.stack(list(ex.map(lambda x: oracle.song_analyze(x, 2), songs)))
        otuning = np.array(list(ex.map(lambda x: oracle.chroma_desc(x)[1], songs)))
    oracle.set_fft_double(True)
    try:
        with ThreadPoolExecutor(32) as ex:
            ref64 = np.stack(list(ex.map(lambda x: oracle.song_analyze(x, 2), songs)))
    finally:
        oracle.set_fft_double(False)
    err = np.abs(got.astype(np.float64) - ref)
    floor = np.abs(ref.astype(np.float64) - ref64)
    assert np.abs(tuning - otuning).max() < 1e-12 and len(set(np.round(otuning, 2))) >= 40
    assert err[:, 0].max() <= 1e-5, err[:, 0].max()
    others = [j for j in range(1, 23) if j not in (6, 7)]
    tol = np.stack([_tol(len(s), 23, 1.0, flips=2) for s in songs])
    bad = [(int(i), int(others[k]), float(err[i, others[k]]), float(floor[i, others[k]]))
           for i, k in zip(*np.nonzero(err[:, others] > tol[:, others]))]
    # Rolloff (features 4, 5) is a bin count: where a spectrum has plateaus between partials, the frame whose running
    # energy sits within an ulp of the threshold ON a plateau changes its bin by the width of the plateau when a magnitude
    # moves by one ulp -- FFT rounding decides, in the oracle as well (its own two code paths disagree on such frames).  Such a
    # song may exceed the tolerance if no more than two of its frames differ from the oracle's bins (the policy's allowance).
    for i, j, e, fl in bad:
        assert j in (4, 5), bad
        frames = np.abs(c.debug_fetch("rolloff", i) - oracle.SpectralDesc().run(songs[i]).series()[1]) > 1e-3
        print(f"song {i}: rolloff feature {j} off by {e:.2e} with {int(frames.sum())} of {len(frames)} frames on another bin")
        assert frames.sum() <= 2, (i, int(frames.sum()))
    assert len({i for i, *_ in bad}) <= 3, bad
    flat = err[:, 6:8]
    assert (flat <= np.maximum(1e-5, floor[:, 6:8])).all(), (flat.max(), floor[:, 6:8].max())
    print(f"flatness: max |gpu - oracle| {flat.max():.2e}; the oracle's f32-vs-f64 distance on the same songs: {floor[:, 6:8].max():.2e}")
    c.close()
