"""GPU tests of round 6 (-m gpu).

Row f1, first clause ("decoder output -> pinned host staging -> H2D double-buffering"), for memory the caller did NOT pin: a
Rust Vec<f32> out of PreAnalyzedSong (src/song/decoder.rs:34-65, 85-101), FFmpeg's frame buffers
(src/song/decoder/ffmpeg.rs:36-109), a numpy array -- ordinary pageable heap memory -- goes through the library's pinned
staging ring (bliss-rs_amd/csrc/staging_ring.hpp).  Whatever the source memory and the ring's shape, the rows must be the same
bits: the ring only moves bytes.
"""
import ctypes as C
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GROUP_BYTES = 512 << 20  # FEED_GROUP_MIB of scheduler.hip


@pytest.fixture(scope="module")
def bliss():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import bliss_rs_amd

    return bliss_rs_amd


@pytest.fixture(scope="module")
def dctx(bliss):
    """the default context the host-pointer batch entry points run on (borrowed)"""
    c = bliss.Context.default(0)
    yield c
    _default_shape(c)


def _default_shape(dctx):
    for k, v in (("stage_lanes", 4), ("stage_slab_kib", 4096), ("stage_slabs", 3), ("stage_numa", 0)):
        dctx.set_option(k, v)


def _mixed_library(rng, n_songs):
    """What decoders deliver: 44.1 kHz stereo s16 (most files), 48 kHz mono s32, 22 050 Hz mono f32 / s16; one to three and a
    half minutes; one song too short."""
    songs = []
    for i in range(n_songs):
        kind = i % 4
        secs = float(rng.uniform(60, 210))
        if i == 5:
            secs = 0.2
        if kind in (0, 1):
            n = int(secs * 44100)
            a = rng.integers(-20000, 20000, (n, 2), dtype=np.int16)
            songs.append((a, 44100))
        elif kind == 2:
            n = int(secs * 48000)
            a = (rng.integers(-2**30, 2**30, n, dtype=np.int64)).astype(np.int32)
            songs.append((a, 48000))
        else:
            n = int(secs * 22050)
            a = (rng.random(n, np.float32) - np.float32(0.5)) if i % 8 == 3 else rng.integers(-20000, 20000, n, dtype=np.int16)
            songs.append((a, 22050))
    return songs


def _fmt(a):
    from bliss_rs_amd import _ffi

    return {np.dtype(np.float32): _ffi.SAMPLE_F32, np.dtype(np.int16): _ffi.SAMPLE_S16, np.dtype(np.int32): _ffi.SAMPLE_S32}[a.dtype]


def _run_decoded(ptrs, songs, version=2):
    from bliss_rs_amd import _ffi

    L = _ffi.lib()
    n = len(songs)
    arr = (_ffi.DecodedSong * n)()
    for i, ((a, rate), p) in enumerate(zip(songs, ptrs)):
        arr[i] = _ffi.DecodedSong(p, a.shape[0], rate, 1 if a.ndim == 1 else a.shape[1], _fmt(a))
    d = 23 if version == 2 else 20
    out = np.full((n, d), np.nan, np.float32)
    st = np.full(n, -1, np.int32)
    _ffi.check(L.blissgpu_analyze_batch_decoded(arr, n, version, out.ctypes.data, st.ctypes.data_as(C.POINTER(C.c_int32))))
    return out, st


def test_pageable_pinned_and_device_sources_give_identical_rows(bliss, dctx):
    """A mixed-format library that spans >= 3 staging groups, from (a) pageable heap memory through the ring, (b) page-locked
    copies straight to the DMA engines, (c) a mix of both in one call, (d) already on the device (conversion + analysis through
    the device entry points): the same rows bit for bit, and the ring did move (a)'s bytes."""
    import torch

    _default_shape(dctx)
    rng = np.random.default_rng(606)
    songs = _mixed_library(rng, 64)
    raw_bytes = sum(a.nbytes for a, _ in songs)
    assert raw_bytes > 2.2 * GROUP_BYTES, raw_bytes  # >= 3 groups of <= 512 MiB
    before = dctx.staged_bytes()
    rows_a, st_a = _run_decoded([a.ctypes.data for a, _ in songs], songs)
    staged = dctx.staged_bytes() - before
    assert staged == raw_bytes, (staged, raw_bytes)  # every byte of the call went through the slabs, once
    assert st_a[5] == 1 and (np.delete(st_a, 5) == 0).all()
    pinned = [torch.from_numpy(a).pin_memory() for a, _ in songs]
    before = dctx.staged_bytes()
    rows_b, st_b = _run_decoded([t.data_ptr() for t in pinned], songs)
    assert dctx.staged_bytes() == before  # page-locked sources never touch the ring
    assert np.array_equal(st_a, st_b)
    ok = st_a == 0
    assert np.array_equal(rows_a[ok].view(np.uint32), rows_b[ok].view(np.uint32))
    # (c) every other song page-locked
    ptrs = [pinned[i].data_ptr() if i % 2 else songs[i][0].ctypes.data for i in range(len(songs))]
    before = dctx.staged_bytes()
    rows_c, st_c = _run_decoded(ptrs, songs)
    assert dctx.staged_bytes() - before == sum(a.nbytes for i, (a, _) in enumerate(songs) if i % 2 == 0)
    assert np.array_equal(st_a, st_c) and np.array_equal(rows_a[ok].view(np.uint32), rows_c[ok].view(np.uint32))
    # (d) device-resident: decode each song on the device, analyse the batch there
    ctx = bliss.Context(0)
    try:
        mono = [ctx.pcm_decode(t.cuda(), rate) for t, (_, rate) in zip(pinned, songs)]
        lens = [int(m.numel()) for m in mono]
        pad = [(n + 63) // 64 * 64 for n in lens]
        offs = np.concatenate([[0], np.cumsum(pad)[:-1]]).astype(np.uint64)
        pcm = torch.zeros(int(sum(pad)) + 64, dtype=torch.float32, device="cuda")
        for o, m in zip(offs, mono):
            pcm[int(o): int(o) + m.numel()] = m
        out, status = ctx.analyze(pcm, offs, lens)
        ctx.synchronize()
        assert np.array_equal(status.cpu().numpy(), st_a)
        assert np.array_equal(out.cpu().numpy()[ok].view(np.uint32), rows_a[ok].view(np.uint32))
    finally:
        ctx.close()


@pytest.mark.parametrize("shape", [(1, 1, 64, 0), (3, 2, 1000, 1), (16, 8, 256, 0), (2, 3, 65536, 1), (0, 3, 4096, 0)])
def test_ring_shape_does_not_change_a_bit(bliss, dctx, shape):
    """1 lane x 1 slab of 64 KiB (every slab reused at once), odd slab sizes, 16 lanes, slabs larger than a song, workers held
    on the device's NUMA node or not, and the ring switched off (the HIP runtime's own staging): the same rows."""
    from bliss_rs_amd import _ffi

    rng = np.random.default_rng(7)
    N = 1_500_000
    n = 12
    pcm = (rng.random(n * N, np.float32) - np.float32(0.5))
    s16 = rng.integers(-30000, 30000, n * N, dtype=np.int16)
    offs = (np.arange(n, dtype=np.uint64) * np.uint64(N))
    lens = np.full(n, N, np.uint64)
    lens[3] = 5000  # too short
    L = _ffi.lib()

    def run(fn, buf):
        out = np.full((n, 23), np.nan, np.float32)
        st = np.full(n, -1, np.int32)
        _ffi.check(fn(buf.ctypes.data, offs.ctypes.data_as(C.POINTER(C.c_uint64)), lens.ctypes.data_as(C.POINTER(C.c_uint64)), n, 2,
                      out.ctypes.data, st.ctypes.data_as(C.POINTER(C.c_int32))))
        return out, st

    _default_shape(dctx)
    ref_f, st_f = run(L.blissgpu_analyze_batch, pcm)
    ref_s, st_s = run(L.blissgpu_analyze_batch_s16, s16)
    lanes, slabs, kib, numa = shape
    dctx.set_option("stage_numa", numa)
    dctx.set_option("stage_lanes", lanes)
    dctx.set_option("stage_slabs", slabs)
    dctx.set_option("stage_slab_kib", kib)
    before = dctx.staged_bytes()
    got_f, gst_f = run(L.blissgpu_analyze_batch, pcm)
    got_s, gst_s = run(L.blissgpu_analyze_batch_s16, s16)
    moved = dctx.staged_bytes() - before
    assert moved == (0 if lanes == 0 else int(lens.sum()) * 6), (moved, shape)
    ok = st_f == 0
    assert st_f[3] == 1 and np.array_equal(st_f, gst_f) and np.array_equal(st_s, gst_s)
    assert np.array_equal(ref_f[ok].view(np.uint32), got_f[ok].view(np.uint32))
    assert np.array_equal(ref_s[ok].view(np.uint32), got_s[ok].view(np.uint32))


def test_small_pageable_calls_stay_off_the_ring(bliss, dctx):
    """Less than 8 MiB of pageable PCM in a call (a lone short song through the single-song front) is left to the runtime's
    bounce buffer: no worker is woken for it."""
    _default_shape(dctx)
    rng = np.random.default_rng(1)
    x = (rng.random(10 * 22050, np.float32) - np.float32(0.5))
    before = dctx.staged_bytes()
    a = bliss.Song.analyze(x)
    assert dctx.staged_bytes() == before
    assert np.isfinite(a.as_arr1()).all()


def test_threads_calling_the_single_song_entry_point_with_heap_buffers(bliss, dctx):
    """The reference's worker pool (src/song/decoder.rs:299-329): threads each analysing their own freshly decoded Vec<f32>.
    The front coalesces them into batches whose pageable songs go through the ring; every thread gets its own song's row."""
    from bliss_rs_amd import _ffi

    L = _ffi.lib()
    _default_shape(dctx)
    rng = np.random.default_rng(3)
    T, per = 8, 3
    N = 3 * 60 * 22050
    songs = [(rng.random(N, np.float32) - np.float32(0.5)) for _ in range(T * per)]
    rows = np.zeros((T * per, 23), np.float32)
    errs = []

    def work(t):
        for k in range(per):
            i = t * per + k
            st = C.c_int32(-1)
            rc = L.blissgpu_analyze(songs[i].ctypes.data, N, 2, rows[i].ctypes.data, C.byref(st))
            if rc or st.value:
                errs.append((i, rc, st.value))

    before = dctx.staged_bytes()
    th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    assert dctx.staged_bytes() - before == T * per * N * 4
    # the same songs as one batch from page-locked memory
    import torch

    flat = torch.from_numpy(np.concatenate(songs)).pin_memory()
    offs = np.arange(T * per, dtype=np.uint64) * np.uint64(N)
    lens = np.full(T * per, N, np.uint64)
    ref = np.zeros_like(rows)
    st = np.zeros(T * per, np.int32)
    _ffi.check(L.blissgpu_analyze_batch(flat.data_ptr(), offs.ctypes.data_as(C.POINTER(C.c_uint64)), lens.ctypes.data_as(C.POINTER(C.c_uint64)),
                                        T * per, 2, ref.ctypes.data, st.ctypes.data_as(C.POINTER(C.c_int32))))
    assert np.array_equal(rows.view(np.uint32), ref.view(np.uint32))


def test_resample_bank_cache_is_bounded_and_evicted_banks_come_back(bliss, oracle):
    """A context keeps the device filter bank of every input rate it has seen -- up to 64 MiB (round 5 advice: a long-running
    host fed odd rates grew without bound).  Thirty inexact rates (1024 phases x ~100 - 1100 taps, up to 4.5 MB each) push the
    first bank out; asking for the first rate again rebuilds it and gives the same samples bit for bit, and every rate still
    equals the oracle."""
    import torch

    ctx = bliss.Context(0)
    try:
        rng = np.random.default_rng(5)
        x = rng.integers(-20000, 20000, 40000, dtype=np.int16)
        d_x = torch.from_numpy(x).cuda()
        first_rate = 700001
        first = ctx.pcm_decode(d_x, first_rate).cpu().numpy()
        total = 0
        for k in range(30):
            rate = 500009 + 8999 * k   # <= 760 980 Hz (the API's limit is 768 kHz); mostly 1024 phases, 750 - 1140 taps: 100 MiB in all
            got = ctx.pcm_decode(d_x, rate)
            ctx.synchronize()
            taps, phases = oracle.swr_filter(rate)[1].taps, oracle.swr_filter(rate)[1].phase_count
            total += taps * (phases + 1) * 4
            if k % 10 == 0:
                assert np.array_equal(got.cpu().numpy().view(np.uint32), oracle.decode_to_mono(x, rate).view(np.uint32)), rate
        assert total > 80 << 20, total   # more than the cache holds: the first bank has been evicted
        again = ctx.pcm_decode(d_x, first_rate).cpu().numpy()
        assert np.array_equal(first.view(np.uint32), again.view(np.uint32))
        assert np.array_equal(first.view(np.uint32), oracle.decode_to_mono(x, first_rate).view(np.uint32))
    finally:
        ctx.close()


def test_song_to_song_chains_of_many_contexts_do_not_wait_for_each_other(bliss, oracle):
    """The greedy chain is one persistent launch that spins on grid barriers; with more than two contexts alive on a device the
    chains run one at a time (round 5 advice: only the sort was gated).  Five contexts, five threads, each its own pool: every
    chain ends and equals the oracle's."""
    import torch

    n_ctx = 5
    ctxs = [bliss.Context(0) for _ in range(n_ctx)]
    try:
        rng = np.random.default_rng(9)
        pools = [rng.random((3000 + 500 * k, 23), np.float32) for k in range(n_ctx)]
        got, errs = [None] * n_ctx, []

        def work(k):
            try:
                with torch.cuda.stream(torch.cuda.Stream()):
                    c = ctxs[k]
                    c.bind_current_stream()
                    p = torch.from_numpy(pools[k]).cuda()
                    for _ in range(3):
                        order = c.song_to_song(p[:1], p, "euclidean")
                    c.synchronize()
                    got[k] = order.cpu().numpy()
            except Exception as e:  # noqa: BLE001
                errs.append((k, repr(e)))

        th = [threading.Thread(target=work, args=(k,)) for k in range(n_ctx)]
        [t.start() for t in th]
        [t.join(timeout=120) for t in th]
        assert not any(t.is_alive() for t in th), "a chain is still spinning"
        assert not errs, errs
        for k in range(n_ctx):
            assert np.array_equal(got[k], oracle.song_to_song(pools[k][:1], pools[k], "euclidean")), k
    finally:
        for c in ctxs:
            c.close()


def test_the_three_big_kernels_are_not_slower_than_recorded():
    """A regression guard on the per-stage table (what benches/analysis_pipeline.rs:8-126 prints for the reference's stages):
    stft8192 / fft512 / chroma make 34 of the step's 37 ms; each, timed alone (serial mode, best of 5 steps of 1024 three-minute
    songs), must stay within 6 % of the minimum recorded in profiles/kernel_times_serial.json (tests/tools/kernel_times.py; boxes
    of the pool differ by about 3 %)."""
    import json
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref = json.load(open(os.path.join(root, "profiles", "kernel_times_serial.json")))
    sys.path.insert(0, os.path.join(root, "tests", "tools"))
    from kernel_times import BIG, measure

    assert ref["songs"] == 1024
    got = measure(1024, 5)
    report = {k: (got[k]["min_ms"], ref["kernels"][k]["min_ms"]) for k in BIG}
    print("kernel: measured min ms, recorded min ms:", report)
    for k, (now, then) in report.items():
        assert now <= 1.06 * then, (k, now, then)


def test_bench_musical_config_runs_and_checks_itself_against_the_oracle():
    """`bench.py --config musical` (round 5 review: one realistic-content line): the batch shape of configs[1] on the seeded
    musical generator, same JSON contract, its in-run oracle check on every song of a small batch -- tempo within the
    reference's 1e-5 (src/song/mod.rs:582-590), the twenty features that are not flatness within 1e-5 (flatness of a
    noise-free tonal song sits on FFT rounding noise: test_random_musical_songs_vs_oracle holds it to the oracle's own
    f32-vs-f64 distance), and the line says that it is not the metric's workload."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "musical", "--songs", "12", "--samples", "661500",
                          "--steps", "2", "--warmup", "1", "--cpu-songs", "12", "--no-pairwise", "--no-host-feed", "--no-playlist",
                          "--no-small-calls"], capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["config"]["name"] == "musical" and "not_the_metric_workload" in r["config"] and r["data"].startswith("synthetic musical")
    assert r["unit"] == "songs/sec" and r["value"] > 0 and r["roofline"]["kernels_ms_per_step"]["fft512_kernel"] > 0
    cb = r["cpu_baseline"]
    assert cb["checked_songs"] == 12 and "musical" in cb["sample"]
    assert cb["tempo_abs_err"]["over_1e-5"] == 0, cb["tempo_abs_err"]
    assert cb["max_abs_err_vs_gpu_non_tempo"] < 2e-4, cb   # flatness floor; every other feature is checked song by song elsewhere


def test_unproven_rolloff_frames_find_their_way_whatever_the_chunking(bliss):
    """Round 6: an unproven frame's 256 magnitudes wait at the entry of the frame's OWN index in the borrowed stretch and a
    sentinel in the rolloff series marks it (no slot counter, no list).  The stretch is per chunk: the same songs cut into
    one, a few and many chunks (other frame indices, other entries) and with every frame forced through the exact pass must give the
    same rows and the same per-frame rolloff series bit for bit -- and no sentinel may survive in any series."""
    import os
    import sys

    import torch

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests", "tools"))
    import musical_check

    rng = np.random.default_rng(77)
    songs = [musical_check.make_song(rng, mods=(i % 3 == 0))[0] for i in range(20)]
    lens = np.array([len(s) for s in songs], np.uint64)
    padded = (lens + np.uint64(63)) // np.uint64(64) * np.uint64(64)
    offs = np.zeros(len(songs), np.uint64)
    offs[1:] = np.cumsum(padded)[:-1]
    buf = np.zeros(int(padded.sum()) + 64, np.float32)
    for s, o in zip(songs, offs):
        buf[int(o):int(o) + len(s)] = s
    pcm = torch.from_numpy(buf).cuda()

    def run(ws_limit, exact_all):
        c = bliss.Context(0)
        if ws_limit:
            c.set_workspace_limit(ws_limit)   # 0.2 MB of scratch per second of audio: the batch (~ 720 s) needs ~ 150 MB
        c.set_option("rolloff_exact_all", exact_all)
        out, status = c.analyze(pcm, offs, lens, 2)
        c.synchronize()
        rows = out.cpu().numpy()
        n_chunks = c.last_chunks()
        c.close()
        return rows, n_chunks

    base, n1 = run(0, 0)
    assert n1 == 1
    c = bliss.Context(0)
    out, _ = c.analyze(pcm, offs, lens, 2)
    c.synchronize()
    series = [c.debug_fetch("rolloff", i) for i in range(len(songs))]
    c.close()
    assert all((s >= 0.0).all() for s in series), "a ROLLOFF_UNPROVEN sentinel survived"
    assert sum(len(s) for s in series) > 100000
    seen = set()
    for ws_limit, exact_all in ((96 << 20, 0), (32 << 20, 0), (0, 1), (32 << 20, 1)):
        rows, n = run(ws_limit, exact_all)
        seen.add(n)
        assert np.array_equal(rows.view(np.uint32), base.view(np.uint32)), (ws_limit, exact_all, n)
    assert len(seen) >= 3 and max(seen) >= 4, seen   # one chunk, a few, many
