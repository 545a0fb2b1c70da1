"""GPU tests added in round 5 (-m gpu) beside tests/test_gpu_resample.py:

  * the two shapes of the FFT-8192 kernel (BLISSGPU_OPT_STFT_SHAPE: 4 workgroups per CU with the window in registers -- production --
    against the narrow form, 5 per CU, window loaded per frame and the transposes in two halves; plus the two measurement forms that
    carry one change each) give the same spectrogram and the same rows, bit for bit, edge frames included
  * a BLISSGPU_OPT_DEBUG_CHROMA batch that needs more than one chunk is refused (the taps are one buffer per context)
  * blissgpu_default_reset / the ten-year clamp of the single-song deadline are callable and harmless
"""
import numpy as np
import pytest

from conftest import assert_row_matches_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bliss():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import bliss_rs_amd

    return bliss_rs_amd


def _pack(songs):
    lens = [len(s) for s in songs]
    offs = np.concatenate([[0], np.cumsum([(l + 63) // 64 * 64 for l in lens])[:-1]]).astype(np.uint64)
    buf = np.zeros(int(offs[-1]) + lens[-1] + 64, np.float32)
    for s, o in zip(songs, offs):
        buf[int(o):int(o) + len(s)] = s
    return buf, offs, lens


def test_stft_shapes_give_identical_rows(bliss, oracle):
    import torch

    # lengths around the framing's edges: the minimum (every frame reflects at both ends), one frame more, a song whose
    # last super-tile is ragged, three minutes, and 65 songs so that several workgroups of a super-tile are idle
    lens = [8192, 8193, 10397, 12602, 70000, 141121, 22050 * 30 + 7, 3969000] + [22050 * 5 + 11 * i for i in range(60)]
    songs = [oracle.white_noise(4000 + i, n) for i, n in enumerate(lens)]
    buf, offs, ls = _pack(songs)
    d_buf = torch.from_numpy(buf).cuda()
    ctx = bliss.Context(0)
    rows, specs = {}, {}
    for shape in (0, 1, 2, 3):
        ctx.set_option("stft_shape", shape)
        out, status = ctx.analyze(d_buf, offs, ls, 2)
        ctx.synchronize()
        assert (status.cpu().numpy() == 0).all()
        rows[shape] = out.cpu().numpy()
        specs[shape] = [ctx.debug_fetch("spectrogram", i) for i in (0, 1, 2, 5, 7)]
    ctx.set_option("stft_shape", 0)
    for shape in (1, 2, 3):
        assert np.array_equal(rows[shape].view(np.uint32), rows[0].view(np.uint32)), shape
        for a, b in zip(specs[shape], specs[0]):
            assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), shape
    ref = oracle.song_analyze(songs[4], 2)
    assert_row_matches_oracle(rows[1][4], ref, white_noise=True, what="stft shape 1, song 4")  # (and the rows are the right ones)
    ctx.close()


def test_debug_chroma_needs_one_chunk(bliss, oracle):
    import torch

    from bliss_rs_amd import _ffi

    songs = [oracle.white_noise(4100 + i, 22050 * 20) for i in range(12)]
    buf, offs, ls = _pack(songs)
    ctx = bliss.Context(0)
    ctx.set_option("debug_chroma", 1)
    ctx.set_workspace_limit(32 << 20)  # a 20-second song needs ~4 MB: a dozen of them do not fit 32 MB -> several chunks
    with pytest.raises(bliss.BlissGpuError) as e:
        ctx.analyze(torch.from_numpy(buf).cuda(), offs, ls, 2)
    assert e.value.code == _ffi.ERR_INVALID and "one chunk" in str(e.value)
    ctx.set_option("debug_chroma", 0)
    out, status = ctx.analyze(torch.from_numpy(buf).cuda(), offs, ls, 2)  # the same batch without the taps: several chunks, fine
    ctx.synchronize()
    assert ctx.last_chunks() > 1 and (status.cpu().numpy() == 0).all()
    ctx.close()


def test_default_reset_and_deadline_clamp(bliss, oracle):
    from bliss_rs_amd import _ffi

    L = _ffi.lib()
    assert L.blissgpu_set_single_song_timeout_ms(2**63 - 1) == 0  # "never": clamped, must not overflow the deadline clock
    a = bliss.Song.analyze(oracle.white_noise(5, 22050 * 4)).as_arr1()
    assert L.blissgpu_default_reset() == 0
    b = bliss.Song.analyze(oracle.white_noise(5, 22050 * 4)).as_arr1()
    assert np.array_equal(a, b)
    assert L.blissgpu_set_single_song_timeout_ms(0) == 0  # back to the default


def test_decoded_feed_across_staging_groups(bliss, oracle):
    """24 three-minute songs as three decoders would deliver them -- 44.1 kHz stereo s16, 22 050 Hz mono f32 (copied verbatim), 48 kHz
    mono s16 -- in one blissgpu_analyze_batch_decoded call: 0.8 GB of raw samples, i.e. several 512 MiB staging groups with different
    raw / PCM layouts alternating between the two buffers.  Every row must equal the same song analysed alone, bit for bit."""
    rng = np.random.default_rng(12)
    songs, rates = [], []
    for i in range(8):
        songs.append(rng.integers(-12000, 12000, (44100 * 180, 2)).astype(np.int16)); rates.append(44100)
        songs.append(oracle.white_noise(4200 + i, 22050 * 180)); rates.append(22050)
        songs.append(rng.integers(-12000, 12000, 48000 * 180 + 17 * i).astype(np.int16)); rates.append(48000)
    res = bliss.analyze_decoded_batch(songs, rates)
    assert all(not isinstance(r, bliss.BlissError) for r in res)
    for k in (0, 1, 2, 9, 10, 11, 21, 22, 23):
        alone = bliss.Song.analyze_decoded(songs[k], rates[k]).as_arr1()
        assert np.array_equal(alone.view(np.uint32), res[k].as_arr1().view(np.uint32)), k
    ref = oracle.song_analyze(oracle.decode_to_mono(songs[2], 48000), 2)
    assert_row_matches_oracle(res[2].as_arr1(), ref, white_noise=True, what="48 kHz song of the mixed batch")


def test_threads_calling_analyze_decoded_are_coalesced(bliss, oracle):
    """The reference's worker pool, one thread per file (src/song/decoder.rs:299-329), with decoders that deliver their own rates:
    16 threads x 6 calls of Song.analyze_decoded on 44.1 / 48 / 22.05 / 96 kHz songs.  Concurrent calls are coalesced into device
    batches of MIXED formats; every row must equal the same call made alone."""
    from concurrent.futures import ThreadPoolExecutor

    rng = np.random.default_rng(3)
    cases = [
        (rng.integers(-9000, 9000, (44100 * 12, 2)).astype(np.int16), 44100),
        (rng.integers(-9000, 9000, 48000 * 9).astype(np.int16), 48000),
        (oracle.white_noise(4300, 22050 * 10), 22050),
        ((rng.random((96000 * 7, 2), np.float32) - 0.5).astype(np.float32), 96000),
        (rng.integers(-2**30, 2**30, 44100 * 8).astype(np.int32), 44100),
        (rng.integers(-9000, 9000, 15000).astype(np.int16), 44100),  # too short once converted
    ]
    serial = []
    for x, r in cases:
        try:
            serial.append(bliss.Song.analyze_decoded(x, r).as_arr1())
        except bliss.AnalysisError as e:
            serial.append(str(e))

    def work(i):
        x, r = cases[i % len(cases)]
        try:
            return i % len(cases), bliss.Song.analyze_decoded(x, r).as_arr1()
        except bliss.AnalysisError as e:
            return i % len(cases), str(e)

    with ThreadPoolExecutor(16) as ex:
        got = list(ex.map(work, range(96)))
    for k, row in got:
        if isinstance(serial[k], str):
            assert row == serial[k]
        else:
            assert np.array_equal(row.view(np.uint32), serial[k].view(np.uint32)), k
    assert isinstance(serial[5], str) and "too short" in serial[5]


def test_flux_order_option_follows_the_reference_sum(bliss, oracle):
    """BLISSGPU_OPT_FLUX_ORDER = 1 adds SpecFlux's 257 terms in the reference's bin order (src/aubio.rs:455-467).  Only the
    tempo chain may change (every other feature bit-identical), and the onset series must move TOWARDS the oracle's: the default
    order (16 per lane + a tree) deviates from the sequential sum by 1.7e-7 rms on its own (tests/tools/fft_error_model_order.py)."""
    import torch

    songs = [oracle.white_noise(4400 + i, 22050 * 40) for i in range(6)]
    buf, offs, ls = _pack(songs)
    d_buf = torch.from_numpy(buf).cuda()
    ctx = bliss.Context(0)
    rows, flux = {}, {}
    for opt in (0, 1):
        ctx.set_option("flux_order", opt)
        out, status = ctx.analyze(d_buf, offs, ls, 2)
        ctx.synchronize()
        rows[opt] = out.cpu().numpy()
        flux[opt] = [ctx.debug_fetch("flux", i) for i in range(len(songs))]
    ctx.set_option("flux_order", 0)
    assert np.array_equal(rows[0][:, 1:].view(np.uint32), rows[1][:, 1:].view(np.uint32))  # tempo is feature 0
    dev = {0: [], 1: []}
    for i, x in enumerate(songs):
        ref = oracle.BPMDesc().run(x).series()[0].astype(np.float64)
        for opt in (0, 1):
            f = flux[opt][i][: len(ref)].astype(np.float64)
            dev[opt].append(np.sqrt(((f - ref) ** 2).mean()) / np.sqrt((ref ** 2).mean()))
        assert_row_matches_oracle(rows[1][i], oracle.song_analyze(x, 2), white_noise=True, what=f"reference-order flux, song {i}")
    print("rms relative deviation of the onset series from the oracle's: default order", np.mean(dev[0]), " reference order", np.mean(dev[1]))
    assert np.mean(dev[1]) < 0.8 * np.mean(dev[0]), (dev[0], dev[1])
    ctx.close()
