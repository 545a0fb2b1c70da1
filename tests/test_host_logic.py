"""CPU-only tests: host logic of the reference-shaped API, C-ABI export check, sharding, and the
world_size-2 gloo path of the feature all-gather."""
import ctypes
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def bliss():
    import bliss_rs_amd

    return bliss_rs_amd


def test_c_abi_exports_every_declared_symbol(bliss):
    so = bliss.LIB_PATH
    if not os.path.exists(so):
        import __graft_entry__ as g

        g.build()
    lib = ctypes.CDLL(so)
    header = open(os.path.join(ROOT, "include", "blissgpu.h")).read()
    declared = sorted(set(re.findall(r"\b(blissgpu_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/blissgpu.h but not exported"
    from bliss_rs_amd import _ffi

    assert sorted(_ffi.SIGNATURES) == declared
    # no compute without a GPU: version / strerror / feature_count / weights are host-only
    lib.blissgpu_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.blissgpu_version()
    lib.blissgpu_strerror.restype = ctypes.c_char_p
    assert b"no CPU path" in lib.blissgpu_strerror(1)
    assert lib.blissgpu_feature_count(2) == 23 and lib.blissgpu_feature_count(1) == 20 and lib.blissgpu_feature_count(3) == 0


def test_no_cpu_fallback_without_device(bliss):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(bliss.BlissGpuError) as e:
        bliss.Song.analyze(np.zeros(10000, np.float32))
    assert e.value.code == 1  # BLISSGPU_ERR_NO_DEVICE
    with pytest.raises(RuntimeError):
        bliss.Context(0)


def test_features_version_and_weights(bliss):
    FV = bliss.FeaturesVersion
    assert FV.LATEST == FV.Version2 and bliss.NUMBER_FEATURES == 23
    assert FV.Version1.feature_count() == 20 and FV.Version2.feature_count() == 23
    assert FV.try_from(1) == FV.Version1 and FV.try_from(2) == FV.Version2
    with pytest.raises(bliss.ProviderError, match=r"This features' version \(3\) does not exist"):
        FV.try_from(3)
    # src/lib.rs:261-270 test_dimensions_weights ; :209-234 VERSION2_WEIGHTS
    assert FV.Version1.feature_weights().shape == (20, 20) and FV.Version2.feature_weights().shape == (23, 23)
    w = np.diag(FV.Version2.feature_weights())
    assert w[0] == np.float32(0.25) and (w[1:10] == 1).all() and (w[10:] == np.float32(3.0 / 13.0)).all()
    assert np.array_equal(FV.Version1.feature_weights(), np.eye(20, dtype=np.float32))


def test_analysis_container(bliss):
    A, FV = bliss.Analysis, bliss.FeaturesVersion
    with pytest.raises(bliss.ProviderError, match="Feature count 3 does not match the expected version feature count 23"):
        A([0.0, 1.0, 2.0], FV.LATEST)
    a = A(np.arange(23) / 10.0, FV.Version2)
    assert a[bliss.AnalysisIndex.Tempo] == 0.0 and a[bliss.AnalysisIndex.Chroma13] == pytest.approx(2.2)
    assert len(list(bliss.AnalysisIndex)) == 23 and len(list(bliss.AnalysisIndexv1)) == 20
    with pytest.raises(RuntimeError, match="incompatible indexes"):
        a[bliss.AnalysisIndexv1.Tempo]
    assert a == A(a.as_vec(), FV.Version2) and a != A(np.zeros(23), FV.Version2)
    assert a.as_arr1().dtype == np.float32 and "Version 2" in repr(a)
    assert str(bliss.AnalysisError("empty or too short song.")) == \
        "error happened while analyzing file - empty or too short song."
    opts = bliss.AnalysisOptions()
    assert opts.features_version == FV.LATEST and opts.number_cores >= 1


def test_decoder_host_side(bliss, tmp_path):
    class Broken(bliss.Decoder):
        @classmethod
        def decode(cls, path):
            raise bliss.DecodingError(f"while opening format for file '{path}'")

    out = list(Broken.analyze_paths(["a.flac", "b.flac"]))
    assert [p for p, _ in out] == ["a.flac", "b.flac"]
    assert all(isinstance(r, bliss.DecodingError) for _, r in out)
    with pytest.raises(bliss.DecodingError):
        bliss.RawPcmDecoder.decode(str(tmp_path / "missing.wav"))
    # stereo decoder output is passed on interleaved: the downmix runs on the device (blissgpu_analyze_interleaved)
    p = tmp_path / "stereo.npy"
    np.save(p, np.zeros((10, 2), np.float32))
    pre = bliss.RawPcmDecoder.decode(str(p))
    assert pre.sample_array.shape == (10, 2) and pre.duration == 10 / 22050
    bad = tmp_path / "cube.npy"
    np.save(bad, np.zeros((4, 2, 2), np.float32))
    with pytest.raises(bliss.DecodingError):
        bliss.RawPcmDecoder.decode(str(bad))
    # s16 stays s16 (2 bytes per sample over PCIe; widened on the device with sample / 32768)
    q = tmp_path / "s16.npy"
    np.save(q, np.array([16384, -32768], np.int16))
    got = bliss.RawPcmDecoder.decode(str(q)).sample_array
    assert got.dtype == np.int16 and got.tolist() == [16384, -32768]
    import wave
    wv = tmp_path / "stereo.wav"
    with wave.open(str(wv), "wb") as f:
        f.setnchannels(2); f.setsampwidth(2); f.setframerate(22050)
        f.writeframes(np.arange(8, dtype="<i2").tobytes())
    assert bliss.RawPcmDecoder.decode(str(wv)).sample_array.tolist() == [[0, 1], [2, 3], [4, 5], [6, 7]]


def test_flac_test_tool(golden_pcm):
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    from flac_decode import adler32_f32le

    assert adler32_f32le(golden_pcm) == 0x5E01930B


def test_shard_songs_balances_samples(bliss):
    from bliss_rs_amd.shard import row_block, shard_songs

    rng = np.random.default_rng(0)
    lengths = rng.integers(661500, 13230000, 50000)  # BASELINE configs[4]: 30 s - 10 min
    shards = shard_songs(lengths, 8)
    allidx = np.sort(np.concatenate(shards))
    assert np.array_equal(allidx, np.arange(len(lengths)))
    loads = np.array([lengths[s].sum() for s in shards])
    assert loads.max() / loads.min() < 1.0005
    eq = shard_songs(np.full(10000, 3969000), 8)   # configs[2]: 10 000 equal songs -> 1250 each
    assert [len(s) for s in eq] == [1250] * 8
    blocks = [row_block(100003, r, 8) for r in range(8)]
    assert blocks[0][0] == 0 and blocks[-1][1] == 100003 and all(blocks[i][1] == blocks[i + 1][0] for i in range(7))


_WORKER = r"""
import os, sys
import numpy as np
sys.path[:0] = [{root!r}, os.path.join({root!r}, "oracle")]
import torch, torch.distributed as dist
import bliss_rs_amd
from bliss_rs_amd.shard import shard_songs, all_gather_features, row_block
import oracle as O
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
lengths = [9000, 30000, 8192, 12000, 20000][:int(os.environ["N_SONGS"])]
mine = shard_songs(lengths, world)[rank]
# the per-rank analysis is stood in for by the CPU oracle (the sharding + collective is what is under test)
rows = (np.stack([O.song_analyze(O.white_noise(int(i), lengths[int(i)])) for i in mine]) if len(mine)
        else np.zeros((0, 23), np.float32))   # a rank may own no song at all
full = all_gather_features(torch.from_numpy(rows), mine, len(lengths))
ref = np.stack([O.song_analyze(O.white_noise(i, n)) for i, n in enumerate(lengths)])
assert np.array_equal(full.numpy(), ref), "gathered matrix differs"
# the fast path bench.py uses: shard sizes known on every rank (shard_songs is deterministic), indices already a tensor
n_max = max(len(s) for s in shard_songs(lengths, world))
full2 = all_gather_features(torch.from_numpy(rows), torch.as_tensor(mine), len(lengths), n_local_max=n_max)
assert np.array_equal(full2.numpy(), ref), "gathered matrix differs (known shard sizes)"
lo, hi = row_block(len(lengths), rank, world)
D = O.pairwise(ref[lo:hi], ref, "euclidean")
parts = [None] * world
dist.all_gather_object(parts, (lo, hi, D))
whole = np.concatenate([p[2] for p in sorted(parts)], axis=0)
assert np.array_equal(whole, O.pairwise(ref, ref, "euclidean"))
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


@pytest.mark.parametrize("world,n_songs", [(2, 5), (4, 3)])   # (4, 3): ragged shards, one rank owns nothing
def test_gloo_shard_and_all_gather(tmp_path, world, n_songs):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   N_SONGS=str(n_songs))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {rank} ok" in o


def test_cpp_host_mirror_compiles(tmp_path, bliss):
    """The C++ mirror of the reference interface builds against include/blissgpu.h + libblissgpu.so and,
    without a GPU, fails loudly instead of falling back to a CPU path."""
    import torch

    exe = tmp_path / "test_bliss_audio"
    libdir = os.path.join(ROOT, "bliss-rs_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "cpp", "test_bliss_audio.cpp"), "-o",
                           str(exe), f"-L{libdir}", "-lblissgpu", f"-Wl,-rpath,{libdir}"])
    if not torch.cuda.is_available():
        out = subprocess.run([str(exe), "a", "b"], capture_output=True, text=True)
        assert out.returncode != 0 and "no usable HIP device" in out.stderr


def test_cue_track_bounds_follow_duration_as_secs_f32(tmp_path, bliss):
    """BlissCueFile::get_songs turns an INDEX into a sample with `(index.as_secs_f32() * SAMPLE_RATE as f32) as usize`
    (src/cue.rs:214-215, 232), and `Duration::as_secs_f32` is `secs as f32 + nanos as f32 / 1e9` -- two roundings and a rounded
    quotient.  Rounding the exact mm * 60 + ss + ff / 75 once instead moves 72 of the 360 000 track starts of an 80-minute disc
    by one sample.  The Python helper, the test mirror in conftest and the C++ mirror all follow the Duration formula; the twelve
    first diverging indices are pinned by value, and the three INDEX lines of the reference's own sheet (data/testcue.cue:
    244 020, 373 086) are not among them."""
    from conftest import cue_bounds

    f32 = np.float32
    diverging = []
    for mm in range(80):
        for ss in range(60):
            for ff in range(75):
                want = int((f32(mm * 60 + ss) + f32(ff * 1_000_000_000 // 75) / f32(1e9)) * f32(22050))
                once = int(f32(mm * 60 + ss + ff / 75.0) * f32(22050))
                got = bliss.cue.cue_track_bounds([(mm, ss, ff)], 10**9)[0][0] if (want != once or ff == 0 and ss % 20 == 0) else want
                assert got == want, (mm, ss, ff)
                if want != once:
                    diverging.append((mm, ss, ff, want))
    assert len(diverging) == 72
    assert diverging[:12] == [(0, 0, 5, 1469), (0, 0, 11, 3234), (0, 0, 22, 6468), (0, 0, 23, 6761), (0, 0, 44, 12936), (0, 0, 46, 13523),
                              (0, 0, 55, 16169), (0, 1, 5, 23519), (0, 1, 14, 26166), (0, 1, 24, 29105), (0, 1, 46, 35573), (0, 1, 64, 40865)]
    msf = [d[:3] for d in diverging]
    assert [b[0] for b in cue_bounds(msf, 10**9)] == [d[3] for d in diverging]
    assert [b[0] for b in bliss.cue.cue_track_bounds([bliss.cue.cue_index_duration(*m) for m in msf], 10**9)] == [d[3] for d in diverging]
    assert bliss.cue.cue_track_bounds([(0, 0, 0), (0, 11, 5), (0, 16, 69)], 496272) == [(0, 244020), (244020, 373086), (373086, 496272)]
    # the C++ mirror (bliss::CueIndex), device-free
    src = tmp_path / "cue_idx.cpp"
    src.write_text("""
#include <cstdio>
#include "%s/bliss-rs_amd/csrc/bliss_audio.hpp"
int main() {
    for (unsigned mm = 0; mm < 80; mm++) for (unsigned ss = 0; ss < 60; ss++) for (unsigned ff = 0; ff < 75; ff++) {
        const auto b = bliss::cue_track_bounds(std::vector<bliss::CueIndex>{bliss::CueIndex::from_msf(mm, ss, ff)}, 1000000000ull);
        std::printf("%%llu\\n", (unsigned long long)b[0].first);
    }
    return 0;
}
""" % ROOT)
    exe = tmp_path / "cue_idx"
    libdir = os.path.join(ROOT, "bliss-rs_amd")
    for flags in (["-O0"], ["-O2", "-march=native", "-ffp-contract=fast"]):   # (the quotient is rounded before the add whatever the flags)
        subprocess.check_call(["g++", "-std=c++17"] + flags + [str(src), "-o", str(exe), f"-L{libdir}", "-lblissgpu", f"-Wl,-rpath,{libdir}"])
        got = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
        want = [int((f32(mm * 60 + ss) + f32(ff * 1_000_000_000 // 75) / f32(1e9)) * f32(22050))
                for mm in range(80) for ss in range(60) for ff in range(75)]
        assert got == want, flags


# ---------------------------------------------------------------------------------------------
# the sharding plan of the C ABI (device-free) against the torch.distributed form, at the sizes
# configs[2] / configs[4] shard at
# ---------------------------------------------------------------------------------------------
def _config4_lengths(n_total=50000, seed=20260927):
    """bench.py's mixed_lengths: configs[4], durations uniform 30 s - 10 min at 22 050 Hz, one seeded draw"""
    rng = np.random.default_rng(seed)
    return rng.integers(30 * 22050, 600 * 22050 + 1, n_total).astype(np.uint64)


def test_c_shard_plan_equals_shard_songs_on_the_config4_draw(bliss):
    from bliss_rs_amd.shard import shard_plan, shard_songs

    lengths = _config4_lengths()
    for world in (1, 2, 4, 8):
        ranks = shard_plan(lengths, world)
        shards = shard_songs(lengths, world)
        assert sum(len(s) for s in shards) == len(lengths)
        for r, s in enumerate(shards):
            assert np.array_equal(np.flatnonzero(ranks == r), s), (world, r)
        load = np.array([lengths[ranks == r].sum() for r in range(world)], dtype=np.float64)
        assert load.max() - load.min() <= lengths.max()  # greedy longest-first: within one song of each other


def test_c_shard_plan_edge_cases(bliss):
    from bliss_rs_amd.shard import shard_plan, shard_songs
    import ctypes as C
    from bliss_rs_amd import _ffi

    cases = [
        [],                                   # nothing to shard
        [5000],                               # fewer songs than ranks
        [7, 7, 7, 7, 7, 7, 7, 7, 7],          # all ties: fewest songs, then lowest rank
        [0, 0, 10, 0, 3, 3, 3],               # zero-length entries still get a rank
        list(range(1, 40)),                   # ascending
        [3969000] * 1250,                     # configs[2]: one GPU's share of equal songs
    ]
    for lengths in cases:
        for world in (1, 2, 3, 8):
            ranks = shard_plan(lengths, world)
            assert len(ranks) == len(lengths) and (len(lengths) == 0 or ranks.max() < world)
            for r, s in enumerate(shard_songs(lengths, world)):
                assert np.array_equal(np.flatnonzero(ranks == r), s), (lengths[:5], world, r)
    L = _ffi.lib()
    assert L.blissgpu_shard_plan(None, 0, 0, None) == _ffi.ERR_INVALID  # world = 0
    # row blocks tile the matrix for every world size, ragged remainders included
    from bliss_rs_amd.shard import row_block
    for n in (0, 1, 7, 100003):
        for world in (1, 2, 8):
            lo, hi = C.c_uint64(), C.c_uint64()
            prev = 0
            for r in range(world):
                L.blissgpu_row_block(n, world, r, C.byref(lo), C.byref(hi))
                assert (lo.value, hi.value) == row_block(n, r, world) and lo.value == prev
                prev = hi.value
            assert prev == n
            L.blissgpu_row_block(n, world, world, C.byref(lo), C.byref(hi))  # no such rank: empty
            assert lo.value == hi.value == n


def test_default_devices_env_parsing_without_a_gpu(bliss):
    """BLISSGPU_DEFAULT_DEVICES restricts / orders / repeats the default contexts; parsing is device-free (the contexts
    themselves are only created on first use, and fail with NO_DEVICE on a machine without a GPU)."""
    code = r"""
import ctypes, sys
L = ctypes.CDLL(%r)
n = L.blissgpu_default_device_count()
print(n, [L.blissgpu_default_device(k) for k in range(n)], L.blissgpu_default_device(n), L.blissgpu_default_device(-1))
""" % bliss.LIB_PATH
    for env, want in (("0,0", "2 [0, 0] -1 -1"), ("2,0,5", "3 [2, 0, 5] -1 -1"), (" 1", "1 [1] -1 -1")):
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                             env=dict(os.environ, BLISSGPU_DEFAULT_DEVICES=env))
        assert out.returncode == 0 and out.stdout.strip() == want, (env, out.stdout, out.stderr)
    import torch

    if not torch.cuda.is_available():  # unset: one default device, whose context creation reports the missing GPU
        env = {k: v for k, v in os.environ.items() if k != "BLISSGPU_DEFAULT_DEVICES"}
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
        assert out.stdout.strip() == "1 [0] -1 -1", out.stdout + out.stderr


@pytest.mark.parametrize("seats", [1, 2, 8])
def test_coalescing_front_on_the_cpu(tmp_path, seats):
    """The group-commit front of the single-song entry points (bliss-rs_amd/csrc/coalescing_front.hpp) is device-free:
    32 threads x 1500 calls against 1 / 2 / 8 seats with a batch runner that sleeps for a RANDOM time and now and then
    throws in the middle of its batch.  Every request runs exactly once, no seat runs two batches at a time, several seats
    are used -- and the run ends: with more than one seat the first form of this loop could spin with the mutex held when
    another leader had taken a caller's request along (a hung -m gpu run)."""
    exe = tmp_path / "test_front"
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", os.path.join(ROOT, "tests", "cpp", "test_front.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe), "32", "1500", str(seats), "300"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "bad 0" in out.stdout and f"of {seats}" in out.stdout


TSAN_CXX = "/opt/rocm/lib/llvm/bin/clang++"   # (g++ 11's libtsan does not know pthread_cond_clockwait: false reports on wait_for)


@pytest.mark.skipif(not os.path.exists(TSAN_CXX), reason="needs the ROCm clang for -fsanitize=thread")
def test_coalescing_front_under_thread_sanitizer(tmp_path):
    """The same test built with -fsanitize=thread, through every scenario of tests/cpp/test_front.cpp: plain (random batch
    durations, leaders that fail mid-batch), one seat whose device cannot give a context (retired after ONE attempt, all
    its traffic served elsewhere), every seat unusable (every call refused, none blocks), leaders that hang while the
    queued callers come back by themselves at their deadline, and a seat that is unusable at first and serves batches
    again once its rest is over.  No data race, no lock-order report, exit code 0."""
    exe = tmp_path / "test_front_tsan"
    subprocess.check_call([TSAN_CXX, "-std=c++17", "-O1", "-g", "-pthread", "-fsanitize=thread",
                           os.path.join(ROOT, "tests", "cpp", "test_front.cpp"), "-o", str(exe)])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66")
    for args in (["32", "300", "1", "300", "0"], ["32", "300", "4", "300", "0"], ["32", "300", "8", "300", "0"],
                 ["32", "200", "4", "300", "1"], ["16", "50", "4", "300", "2"], ["16", "60", "3", "300", "3"],
                 ["32", "400", "4", "300", "4"]):
        out = subprocess.run([str(exe)] + args, capture_output=True, text=True, timeout=300, env=env)
        assert out.returncode == 0, (args, out.stdout[-1500:], out.stderr[-3000:])
        assert "ThreadSanitizer" not in out.stderr, (args, out.stderr[-3000:])
        assert "bad 0" in out.stdout


def test_staging_ring_on_the_cpu(tmp_path):
    """The pinned staging ring of the host PCM feed (bliss-rs_amd/csrc/staging_ring.hpp) is device-free: against a stand-in
    device whose copy queues execute LATER (a slab refilled before its copy completed corrupts the destination) every byte
    of every transfer arrives, begin / end run once per lane and transfer, an injected copy error comes back from the
    transfer it hit and the next one is clean, drain() covers the error paths, a restart with another shape leaks nothing.
    Several shapes incl. one lane with one slab (every piece waits for the previous copy)."""
    exe = tmp_path / "test_staging"
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", os.path.join(ROOT, "tests", "cpp", "test_staging.cpp"), "-o", str(exe)])
    for args in (["80", "4", "3", "16", "1"], ["80", "1", "1", "4", "2"], ["120", "8", "2", "8", "3"], ["40", "16", "8", "64", "4"]):
        out = subprocess.run([str(exe)] + args, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0 and "staging ring ok" in out.stdout, (args, out.stdout + out.stderr)


@pytest.mark.skipif(not os.path.exists(TSAN_CXX), reason="needs the ROCm clang for -fsanitize=thread")
def test_staging_ring_under_thread_sanitizer(tmp_path):
    """The same test with -fsanitize=thread: the workers, the poster and the stand-in copy engines share the transfer queue,
    the slabs and their events without a data race."""
    exe = tmp_path / "test_staging_tsan"
    subprocess.check_call([TSAN_CXX, "-std=c++17", "-O1", "-g", "-pthread", "-fsanitize=thread",
                           os.path.join(ROOT, "tests", "cpp", "test_staging.cpp"), "-o", str(exe)])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66")
    for args in (["60", "4", "3", "16", "5"], ["60", "1", "1", "4", "6"], ["60", "7", "2", "64", "7"]):
        out = subprocess.run([str(exe)] + args, capture_output=True, text=True, timeout=300, env=env)
        assert out.returncode == 0, (args, out.stdout[-1500:], out.stderr[-3000:])
        assert "ThreadSanitizer" not in out.stderr, (args, out.stderr[-3000:])


def test_rolloff_guard_on_emulated_summation_orders():
    """The FFT-512 kernel counts the rolloff bins in its own summation order and proves, per frame, that the reference's
    sequential order (src/aubio.rs:36-58) gives the same count -- or hands the frame to the exact pass (kernels_fft512.hip,
    the frame epilogue; ROLL_GUARD).  Here both orders are emulated in numpy f32, operation for operation (fused squares,
    lane scan by row_shr 1/2/4/8, butterfly total, the walk as d = running - threshold), on white, pink and sparse spectra
    and on frames ENGINEERED to sit on a tie (a partial holding 5 % of the energy within 3e-7): wherever the two orders
    disagree the guard must have flagged the frame.  (White noise: 2 % flagged, as measured on the GPU.)"""
    import re

    src = open(os.path.join(ROOT, "bliss-rs_amd", "csrc", "kernels_fft512.hip")).read()
    guard = np.float32(float(re.search(r"ROLL_GUARD = ([0-9.]+)f / 16777216\.0f", src).group(1)) / 16777216.0)
    f32 = np.float32

    def fma32(a, b, c):  # round_f32(a * b + c): the product of two f32 is exact in f64
        return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)

    def both_orders(M):
        n = len(M)
        sq = (M * M).astype(f32)  # the reference rounds the square, then adds
        S = np.zeros(n, f32)
        P = np.empty((n, 256), f32)
        for j in range(256):
            S = (S + sq[:, j]).astype(f32)
            P[:, j] = S
        cnt_seq = (P < (S * f32(0.95)).astype(f32)[:, None]).sum(1)
        L = M.reshape(n, 16, 16)  # lane l: bins 16 l .. 16 l + 15
        qx, qy = np.zeros((n, 16), f32), np.zeros((n, 16), f32)
        for i in range(8):
            qx, qy = fma32(L[:, :, 2 * i], L[:, :, 2 * i], qx), fma32(L[:, :, 2 * i + 1], L[:, :, 2 * i + 1], qy)
        sqsum = (qx + qy).astype(f32)
        inc = sqsum.copy()
        for d in (1, 2, 4, 8):
            sh = np.zeros_like(inc)
            sh[:, d:] = inc[:, :-d]
            inc = (inc + sh).astype(f32)
        tot, idx = sqsum.copy(), np.arange(16)
        for perm in (idx ^ 1, idx ^ 2, (idx & 8) | (7 - (idx & 7)), 15 - idx):
            tot = (tot + tot[:, perm]).astype(f32)
        total = tot[:, 0]
        d = ((inc - sqsum).astype(f32) - (total * f32(0.95)).astype(f32)[:, None]).astype(f32)
        near, cnt = np.full((n, 16), np.inf, f32), np.zeros((n, 16), np.int64)
        for e in range(16):
            d = fma32(L[:, :, e], L[:, :, e], d)
            cnt += d < 0
            near = np.minimum(near, np.abs(d))
        risky = ((near <= (total * guard)[:, None]).any(1) | (total < 1e-30)) & (total != 0)
        return cnt_seq, cnt.sum(1), risky

    rng = np.random.default_rng(0)
    n = 50000
    white = np.abs(rng.standard_normal((n, 256))).astype(f32)
    pink = (np.abs(rng.standard_normal((n, 256))) / np.sqrt(1 + np.arange(256))).astype(f32)
    sparse = (1e-6 * np.abs(rng.standard_normal((n, 256)))).astype(f32)
    for _ in range(6):
        sparse[np.arange(n), rng.integers(0, 256, n)] = rng.uniform(0.05, 1.0, n).astype(f32)
    tie = np.zeros((n, 256), f32)
    tie[np.arange(n), rng.integers(5, 100, n)] = 1.0
    tie[np.arange(n), rng.integers(150, 250, n)] = (np.sqrt(0.05 / 0.95) * (1 + rng.uniform(-3e-7, 3e-7, n))).astype(f32)
    tie += (1e-5 * np.abs(rng.standard_normal((n, 256)))).astype(f32) * (rng.random((n, 1)) < 0.5)
    disagreements = 0
    for name, M in (("white", white), ("pink", pink), ("sparse", sparse), ("tie", tie)):
        cs, cg, risky = both_orders(M)
        disagreements += int((cs != cg).sum())
        assert not ((cs != cg) & ~risky).any(), name
        if name == "white":
            assert 0.01 < risky.mean() < 0.035, risky.mean()
    assert disagreements > 20       # the engineered ties do make the orders disagree: the check is not vacuous


def test_rust_binding_declarations_match_the_header():
    """bindings/rust/gpu.rs cannot be compiled here (no rustc in the image); its `extern "C"` block is held to
    include/blissgpu.h declaration by declaration instead -- name, argument count, every argument type, the return type."""
    import re

    header = open(os.path.join(ROOT, "include", "blissgpu.h")).read()
    header = re.sub(r"/\*.*?\*/", " ", header, flags=re.S)
    cdecl = {}
    for m in re.finditer(r"\n((?:const\s+)?[a-z_0-9]+\s*\**)\s*(blissgpu_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", header):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        cdecl[name] = (" ".join(ret.split()), [a.strip() for a in args.split(",")] if args.strip() not in ("", "void") else [])

    def rust_of(ctype):
        t = " ".join(ctype.replace("*", " * ").split())
        t = re.sub(r"\s+[A-Za-z_][A-Za-z_0-9]*$", "", t) if not t.endswith("*") and len(t.split()) > 1 and t.split()[-1] not in (
            "int", "float", "double", "char", "void", "uint64_t", "uint32_t", "int64_t", "int32_t", "int16_t") else t
        base = {"int": "c_int", "float": "f32", "double": "f64", "uint64_t": "u64", "uint32_t": "u32", "int64_t": "i64", "int32_t": "i32",
                "int16_t": "i16", "void": "c_void", "char": "c_char", "blissgpu_ctx": "blissgpu_ctx", "blissgpu_node": "blissgpu_node",
                "blissgpu_decoded_song": "blissgpu_decoded_song"}
        toks = t.split()
        const = toks[0] == "const"
        if const:
            toks = toks[1:]
        stars = toks.count("*")
        ty = base[toks[0]]
        if stars == 0:
            return ty
        out = ("*const " if const else "*mut ") + ty
        for _ in range(stars - 1):
            out = "*mut " + out
        return out

    rs = open(os.path.join(ROOT, "bindings", "rust", "gpu.rs")).read()
    block = rs[rs.index('extern "C" {'):]
    block = block[:block.index("\n    }\n")]
    block = re.sub(r"//[^\n]*", "", block)
    n = 0
    for m in re.finditer(r"pub fn (blissgpu_[a-z0-9_]+)\(([^)]*)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        name, args, ret = m.group(1), m.group(2), (m.group(3) or "()").strip()
        assert name in cdecl, f"{name} is not declared in include/blissgpu.h"
        cret, cargs = cdecl[name]
        rargs = [" ".join(a.split(":", 1)[1].split()) for a in args.split(",") if a.strip()]
        assert len(rargs) == len(cargs), (name, rargs, cargs)
        for ra, ca in zip(rargs, cargs):
            assert ra == rust_of(ca), (name, ra, ca, rust_of(ca))
        assert ret == rust_of(cret), (name, ret, cret)
        n += 1
    assert n >= 30, n


def test_decoded_song_struct_is_the_same_in_c_rust_and_ctypes():
    """blissgpu_decoded_song crosses the ABI by pointer: its fields, their order and their widths must agree between
    include/blissgpu.h, bindings/rust/gpu.rs (#[repr(C)]) and bliss_rs_amd._ffi.DecodedSong."""
    import ctypes as C
    import re

    header = open(os.path.join(ROOT, "include", "blissgpu.h")).read()
    body = re.search(r"typedef struct blissgpu_decoded_song \{(.*?)\} blissgpu_decoded_song;", header, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", " ", body, flags=re.S)
    c_fields = [(m.group(2), " ".join(m.group(1).split())) for m in re.finditer(r"([a-z_0-9 ]+?\**)\s*\b([a-z_]+);", body)]
    assert c_fields == [("pcm", "const void *"), ("frames", "uint64_t"), ("sample_rate", "uint32_t"), ("channels", "uint16_t"),
                        ("sample_format", "uint16_t")], c_fields
    rs = open(os.path.join(ROOT, "bindings", "rust", "gpu.rs")).read()
    rbody = re.search(r"#\[repr\(C\)\]\s*#\[derive\(Clone, Copy\)\]\s*pub struct blissgpu_decoded_song \{(.*?)\}", rs, flags=re.S).group(1)
    r_fields = [(m.group(1), m.group(2).strip()) for m in re.finditer(r"pub ([a-z_]+): ([^,]+),", rbody)]
    assert r_fields == [("pcm", "*const c_void"), ("frames", "u64"), ("sample_rate", "u32"), ("channels", "u16"), ("sample_format", "u16")]
    from bliss_rs_amd import _ffi

    assert [(n, t) for n, t in _ffi.DecodedSong._fields_] == [("pcm", C.c_void_p), ("frames", C.c_uint64), ("sample_rate", C.c_uint32),
                                                             ("channels", C.c_uint16), ("sample_format", C.c_uint16)]
    assert C.sizeof(_ffi.DecodedSong) == 24


def test_musical_bench_batch_is_seeded_per_song():
    """`bench.py --config musical` / full_check.py --musical: song i of the batch comes from the generator seeded with (seed, i)
    -- the same samples whatever the batch size or the number of generating processes, so that a 1024-song oracle check and a
    128-song spot check talk about the same songs."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    from musical_check import make_song, musical_batch

    a, ma = musical_batch(3, 30000, seed=5, procs=1)
    b, mb = musical_batch(5, 30000, seed=5, procs=2)
    assert all(np.array_equal(x, y) for x, y in zip(a, b[:3])) and ma == mb[:3]
    x, _ = make_song(np.random.default_rng([5, 2]), False, 30000)
    assert x.dtype == np.float32 and len(x) == 30000 and np.array_equal(x, a[2])
    c, _ = musical_batch(1, 30000, seed=6, procs=1)
    assert not np.array_equal(c[0], a[0])
