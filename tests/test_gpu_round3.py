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
    out = subprocess.run([str(exe), "16", "32"], capture_output=True, text=True, timeout=600, env=env)
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
