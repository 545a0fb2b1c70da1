#!/usr/bin/env python3
"""Benchmark of the hot path named by BASELINE.json: songs/sec for batched 22 050 Hz f32 analysis (all features) on
N MI355X, synthetic white-noise PCM resident in HBM, with the HBM and FP32 rooflines, the CPU oracle timed beside it,
and pairwise distances/sec over 100 k feature vectors.

    python bench.py --gpus 1 --steps 3 --warmup 1                      # configs[1]: 1024 three-minute songs per GPU
    python bench.py --config mixed                                     # configs[4]: this GPU's share of the 30 s - 10 min corpus
    python bench.py --config library                                   # configs[2]: 10 000 songs, all-gather, sharded pairwise
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W [--config ...]

A "step" = one pass of Song::analyze over this GPU's batch (+ the RCCL all-gather of the feature rows when N > 1;
--config library adds the row-block-sharded pairwise kernel).  Prints ONE JSON line on rank 0.

  config   workload                                                                       scaling
  batch    configs[1]: --songs (1024) x 3 969 000-sample songs per GPU                     weak
  mixed    configs[4]: 6 250 songs per GPU, durations uniform 30 s - 10 min (seeded),      weak (8 GPUs = the 50 000-song corpus)
           sharded over the ranks by sample count, streamed through the chunk scheduler
  library  configs[2]: 10 000 three-minute songs in total, 10 000 / N per GPU              strong
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SONG_SAMPLES = 3969000   # 3 min at 22 050 Hz
HBM_PEAK_GBS = 8000.0    # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
FP32_PEAK = 157.3e12     # MI355X FP32 vector peak, flop/s (same guide)
# FP32-equivalent operation count of one analysis: 1.4e9 flop per 3-minute song with real-input
# FFT counts (1800 FFT-8192 + 31 005 FFT-512 + the f64 chroma contraction + the beat tracker); it scales with the samples
FLOP_PER_SAMPLE = 1.4e9 / SONG_SAMPLES


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=("batch", "mixed", "library", "musical"), default="batch",
                    help="batch = configs[1] (white noise, the metric's workload); musical = the same batch shape on seeded MUSICAL "
                         "signals (detuned notes, percussion, noise floors: tests/tools/musical_check.py) -- white noise is the "
                         "worst case of the peak picker, music the common one")
    ap.add_argument("--songs", type=int, default=0, help="songs per GPU per step (default: 1024 batch, 6250 mixed, 10000/N library)")
    ap.add_argument("--samples", type=int, default=SONG_SAMPLES, help="samples per song (batch / library)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pairwise", action="store_true")
    ap.add_argument("--pairwise-n", type=int, default=100000)
    ap.add_argument("--cpu-songs", type=int, default=128, help="songs timed on the CPU oracle (also the parity spot check)")
    ap.add_argument("--no-host-feed", action="store_true", help="skip the PCIe-inclusive host-buffer measurement")
    ap.add_argument("--no-playlist", action="store_true", help="skip the playlist-ordering measurement")
    ap.add_argument("--no-small-calls", action="store_true", help="skip the single-song latency / threaded small-call measurement")
    ap.add_argument("--ws-limit-gb", type=float, default=0.0, help="workspace limit per chunk slot; 0 = library default")
    ap.add_argument("--host-feed-songs", type=int, default=256)
    ap.add_argument("--seed", type=int, default=20260927)
    ap.add_argument("--serial", action="store_true", help="every kernel on one stream (clean per-kernel timings; not the production schedule)")
    ap.add_argument("--tail-mode", type=int, default=None, help="beat tracker placement (blissgpu_ctx_set_option)")
    ap.add_argument("--pipeline-chunks", type=int, default=None, help="cut batches into at least this many chunks")
    ap.add_argument("--traffic", action="store_true",
                    help="measure roofline.traffic now (two rocprofv3 --pmc passes of a 256-song batch, ~1 min) instead of "
                         "reading profiles/hbm_traffic.json")
    ap.add_argument("--share-device", action="store_true",
                    help="developer check on a 1-GPU box: every rank of an N > 1 launch uses device 0 and the collectives "
                         "go over gloo (RCCL refuses two ranks on one device); exercises the N > 1 control flow, the line "
                         "is marked and is not a measurement")
    ap.add_argument("--node", type=int, default=0,
                    help="drive N GPUs from THIS process through the C-ABI node API (blissgpu_node_*: what a Rust host "
                         "would call) instead of one torch.distributed rank per GPU; configs batch / library")
    ap.add_argument("--node-devices", default="", help="HIP ordinals of the node's ranks, e.g. 0,0 = two loopback ranks on one GPU")
    return ap.parse_args()


def mixed_lengths(n_total, seed):
    """configs[4]: durations uniform in [30 s, 10 min] at 22 050 Hz, one seeded draw for the whole corpus"""
    rng = np.random.default_rng(seed)
    return rng.integers(30 * 22050, 600 * 22050 + 1, n_total).astype(np.uint64)


def kernel_sources_sha():
    """sha256 over the HIP sources the analysis kernels are built from (what a traffic measurement is valid for)"""
    import hashlib

    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "bliss-rs_amd", "csrc")
    for name in ("kernels_chroma.hip", "kernels_fft512.hip", "kernels_tempo.hip", "kernels_finalize.hip", "fft_r16.hpp",
                 "device_utils.hpp", "internal.hpp"):
        h.update(open(os.path.join(csrc, name), "rb").read())
    return h.hexdigest()


def oracle_mod():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O

    return O


def host_cpus():
    """(physical cores, logical CPUs) of this host from /proc/cpuinfo; physical = distinct (physical id, core id) pairs"""
    logical = os.cpu_count() or 1
    try:
        cores, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
        if phys is not None and core is not None:
            cores.add((phys, core))
        return (len(cores) or logical), logical
    except OSError:
        return logical, logical


def cpu_baseline(global_indices, lengths, features_version=2, songs=None):
    """The oracle (C restatement of the reference algorithm, oracle/) on the host cores: the same white-noise songs
    (bit-identical generator), one song per thread at a time (the reference's bulk path runs one Song::analyze per worker
    thread, src/song/decoder.rs:282-329).  TIMED with the baseline-only build (oracle/Makefile `native`: -O3
    -march=native, vectorisable radix-4 FFT, compiled on this machine) over a SWEEP of thread counts up to the logical
    CPU count -- the best one is the value, `cores` is the thread count it used, and the host's physical cores / logical
    CPUs are stated separately; the parity CHECK of the GPU rows uses the default build (the pinned checker), timed too.
    This is a PORT (kind: "port"), not bliss-rs itself: see DESIGN.md section 5 for what that does and does not say."""
    O = oracle_mod()
    physical, ncpu = host_cpus()
    n = len(global_indices)
    pcm = np.concatenate(songs if songs is not None else [O.white_noise(int(g), int(l)) for g, l in zip(global_indices, lengths)])
    lens = np.asarray(lengths, np.uint64)
    offs = np.zeros(n, np.uint64)
    offs[1:] = np.cumsum(lens)[:-1]
    t0 = time.perf_counter()
    out, status = O.song_analyze_batch(pcm, offs, lens, features_version, min(ncpu, n, 64))
    checker_rate = n / (time.perf_counter() - t0)
    tried, fast_note = [], "baseline-only build (-O3 -march=native, radix-4 FFT)"
    try:
        for cores in sorted({min(ncpu, n, c) for c in (32, 64, 128, 256, physical)}):
            t0 = time.perf_counter()
            fast, _ = O.song_analyze_batch(pcm, offs, lens, features_version, cores, fast=True)
            tried.append((n / (time.perf_counter() - t0), cores))
        fast_dev = float(np.abs(fast - out).max())
    except Exception as e:  # noqa: BLE001  (no compiler on the box: fall back to the checker's own timing, say so)
        tried, fast_note, fast_dev = [(checker_rate, min(ncpu, n, 64))], f"default build only ({type(e).__name__})", None
    rate, cores = max(tried)
    res = {"value": round(rate, 3), "unit": "songs/sec", "cores": cores, "kind": "port",
           "host_physical_cores": physical, "host_logical_cpus": ncpu,
           "threads_sweep": {str(c): round(r, 2) for r, c in sorted(tried, key=lambda rc: rc[1])},
           "sample": f"{n} of the same {'white-noise' if songs is None else 'musical'} songs ({int(lens.sum())} samples; spread over the whole batch), "
                     f"oracle/bliss_oracle.c, {fast_note} -- an oracle port, not bliss-rs (rustfft is SIMD mixed-radix); one "
                     f"song per thread, best of the thread-count sweep (threads_sweep) on a host with {physical} physical "
                     f"cores / {ncpu} logical CPUs; default (checker) build: {checker_rate:.1f} songs/s",
           "samples_per_sec": round(rate * float(lens.mean()), 1),
           "max_abs_dev_baseline_build_vs_checker": fast_dev}
    try:  # per-descriptor seconds of ONE song on one core: where the CPU time goes
        k = int(np.argmin(np.abs(lens.astype(np.int64) - int(np.median(lens)))))
        one = pcm[int(offs[k]):int(offs[k] + lens[k])]
        try:
            res["per_descriptor_seconds_one_song"] = O.song_analyze_timed(one, features_version, fast=True)
        except Exception:  # noqa: BLE001
            res["per_descriptor_seconds_one_song"] = O.song_analyze_timed(one, features_version)
    except Exception as e:  # noqa: BLE001
        res["per_descriptor_seconds_one_song"] = {"error": str(e)[:200]}
    return res, out


def small_calls():
    """Single-song latency, 16-thread small-call throughput and ns per Song::distance through the host-pointer C ABI
    (tests/cpp/test_threads.cpp: also the re-entrancy check)."""
    exe = os.path.join(ROOT, "tests", "cpp", "bin", "test_threads")
    src = os.path.join(ROOT, "tests", "cpp", "test_threads.cpp")
    libdir = os.path.join(ROOT, "bliss-rs_amd")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", src, "-o", exe, f"-L{libdir}", "-lblissgpu",
                               f"-Wl,-rpath,{libdir}"])
    out = subprocess.run([exe, "16", "32"], capture_output=True, text=True, timeout=600)
    if out.returncode != 0:
        raise RuntimeError((out.stdout + out.stderr)[-300:])
    res = {}
    for line in out.stdout.splitlines():
        parts = line.split()
        if len(parts) == 2:
            try:
                res[parts[0]] = float(parts[1])
            except ValueError:
                pass
    res["note"] = ("blissgpu_analyze from 16 threads x 32 calls (3-22 s songs, pageable host memory): concurrent callers are "
                   "coalesced into device batches; every row bit-identical to the serial run")
    return res


def main_node(args):
    """--node N: the C-ABI node form (blissgpu_node_*: one host process, N devices, ONE ncclAllGather of the feature rows
    inside the library) -- what a Rust / C host calls.  torch only allocates the per-device PCM buffers here."""
    import torch

    import bliss_rs_amd as bliss

    world = args.node
    devices = [int(x) for x in args.node_devices.split(",")] if args.node_devices else list(range(world))
    if len(devices) != world:
        raise SystemExit("--node-devices must name --node ordinals")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the library has no CPU path")
    if max(devices) >= torch.cuda.device_count():
        raise SystemExit(f"--node-devices {devices}: only {torch.cuda.device_count()} HIP device(s) are visible")
    if args.gpus not in (1, len(set(devices))):
        raise SystemExit(f"--gpus {args.gpus} but --node drives {len(set(devices))} distinct device(s)")
    node = bliss.Node(world, devices=devices)
    loopback = len(set(devices)) < world
    N = args.samples
    if args.config == "library":
        n_total, version, scaling = (args.songs * world) if args.songs else 10000, 1, "strong"
        all_lens = np.full(n_total, N, np.uint64)
        workload = (f"configs[2]: {n_total} pre-decoded {N}-sample (3-min) songs sharded across {world} GPU(s), all-gather of "
                    f"the 20-dim feature rows (FeaturesVersion 1)")
    elif args.config == "batch":
        per = args.songs or 1024
        n_total, version, scaling = per * world, 2, "weak"
        all_lens = np.full(n_total, N, np.uint64)
        workload = (f"configs[1]: batch of {per} synthetic {N}-sample (3-min) white-noise f32 PCM buffers per GPU, full "
                    f"23-feature descriptor set (FeaturesVersion 2)")
    else:
        raise SystemExit("--node supports --config batch / library")
    d = 23 if version == 2 else 20
    ranks = node.shard(all_lens)
    offs = np.zeros(n_total, np.uint64)
    bufs = []
    for r in range(world):
        mine = np.flatnonzero(ranks == r)
        padded = (all_lens[mine] + np.uint64(63)) // np.uint64(64) * np.uint64(64)
        o = np.zeros(len(mine), np.uint64)
        o[1:] = np.cumsum(padded)[:-1]
        offs[mine] = o
        buf = torch.empty(int(padded.sum()) + 64, dtype=torch.float32, device=f"cuda:{devices[r]}")
        node.synth_white_noise(r, buf.data_ptr(), o, all_lens[mine], mine)
        bufs.append(buf)
        if loopback or args.ws_limit_gb > 0:
            node.ctx_set_workspace_limit(r, int((args.ws_limit_gb if args.ws_limit_gb > 0 else 8.0) * (1 << 30)))
    ptrs = [b.data_ptr() for b in bufs]

    def step():
        node.analyze_device(ptrs, offs, all_lens, ranks, version)   # enqueues every rank, gathers, synchronises

    for _ in range(args.warmup):
        step()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    elapsed = time.perf_counter() - t0
    rows = node.features(0)
    total_samples = float(all_lens.sum())
    algo = 4.0 * total_samples + 4.0 * d * n_total
    result = {
        "metric": "songs/sec (3-min 22 050 Hz f32) at 1/2/4/8 GPU; HBM GB/s vs roofline",
        "value": round(n_total * args.steps / elapsed, 2), "unit": "songs/sec", "n_gpus": len(set(devices)), "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": scaling, "vs_baseline": None, "dtype": "f32",
        "data": "synthetic white noise (Philox4x32-10, uniform [-0.5,0.5)), generated in HBM",
        "config": {"workload": workload, "name": args.config, "songs_total": n_total, "samples_per_song": N, "features": d,
                   "ranks": world, "devices": devices,
                   "parallelism": f"ONE process, C-ABI node API (blissgpu_node_analyze_device): songs sharded x{world}, "
                                  + ("loopback ranks on shared GPUs: gather by device-to-device copies" if loopback
                                     else "one grouped ncclAllGather of the feature rows over xGMI")},
        "whole_job_GBps": round(algo * args.steps / elapsed / 1e9, 2),
        "rows_finite": bool(np.isfinite(rows).all()),
        "note": "per-kernel roofline and cpu_baseline are reported by the one-rank-per-GPU form (bench.py --gpus N)",
    }
    if args.config == "library" and not args.no_pairwise:
        t0 = time.perf_counter()
        D = node.pairwise("euclidean")
        result["node_pairwise"] = {"n": n_total, "ms_including_D2H_of_the_matrix": round((time.perf_counter() - t0) * 1e3, 1),
                                   "symmetric": bool(np.array_equal(D, D.T))}
    print(json.dumps(result))
    node.close()


def main():
    args = parse()
    if args.node > 0:
        return main_node(args)
    import torch
    import torch.distributed as dist

    import bliss_rs_amd as bliss
    from bliss_rs_amd.shard import all_gather_features, row_block, shard_songs

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the library has no CPU path")
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > torch.cuda.device_count() and not args.share_device:
        raise SystemExit(f"--gpus {args.gpus} but only {torch.cuda.device_count()} HIP device(s) are visible: refusing to "
                         f"print a line for fewer GPUs than asked for")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # launched as a plain process for N > 1 GPUs: become the launcher of N ranks (one per GPU over RCCL), exactly the
        # command the module docstring gives; the ranks' single JSON line is this process's output
        import socket

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stderr.write(f"bench.py: --gpus {args.gpus} without WORLD_SIZE: launching {' '.join(cmd[1:9])} ...\n")
        sys.stderr.flush()
        raise SystemExit(subprocess.run(cmd).returncode)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: a line whose n_gpus is not --gpus is never printed")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.share_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    ctx = bliss.Context(local_rank)
    if args.serial:
        ctx.set_option("serial", 1)
    if args.tail_mode is not None:
        ctx.set_option("tail_mode", args.tail_mode)
    if args.pipeline_chunks is not None:
        ctx.set_option("pipeline_chunks", args.pipeline_chunks)
    version = 2
    scaling = "weak"
    notes = {}
    # ---- the workload: global song list -> this rank's share ----
    musical_songs = None
    if args.config in ("batch", "musical"):
        n = args.songs or 1024
        N = args.samples
        lens = np.full(n, N, np.uint64)
        global_idx = np.arange(rank * n, rank * n + n)
        n_total = world * n
        workload = (f"configs[1]: batch of {n} synthetic {N}-sample (3-min) white-noise f32 PCM buffers per GPU, full "
                    f"23-feature descriptor set (FeaturesVersion 2)")
        if args.config == "musical":
            sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
            from musical_check import musical_batch

            t_gen = time.perf_counter()
            musical_songs, _meta = musical_batch(n, N, seed=args.seed + rank)
            workload = (f"the shape of configs[1] on MUSICAL content: {n} synthetic {N}-sample (3-min) songs per GPU from the seeded "
                        f"generator of tests/tools/musical_check.py (detuned harmonic notes at a random tempo, percussive bursts, "
                        f"noise floors, gains 1e-3 .. 1; seed {args.seed}), full 23-feature set -- NOT BASELINE's workload (that is "
                        f"--config batch); generated on the host in {time.perf_counter() - t_gen:.0f} s")
    elif args.config == "library":
        n_total = (args.songs * world) if args.songs else 10000
        N = args.samples
        version = 1  # BASELINE configs[2] gathers 20-dim feature vectors = FeaturesVersion 1
        scaling = "strong"
        shards = shard_songs(np.full(n_total, N, np.int64), world)
        global_idx = shards[rank]
        n = len(global_idx)
        lens = np.full(n, N, np.uint64)
        workload = (f"configs[2]: {n_total} pre-decoded {N}-sample (3-min) songs sharded across {world} GPU(s) "
                    f"({n} on this rank), RCCL all-gather of the 20-dim feature rows (FeaturesVersion 1), then the "
                    f"row-block-sharded {n_total} x {n_total} euclidean pairwise kernel on every rank")
    else:  # mixed
        per_gpu = args.songs or 6250
        n_total = per_gpu * world
        all_lens = mixed_lengths(n_total, args.seed)
        shards = shard_songs(all_lens.astype(np.int64), world)
        global_idx = shards[rank]
        lens = all_lens[global_idx]
        # everything must be resident before the timed region: trim the share to what fits beside two chunk slots
        free_b, total_b = torch.cuda.mem_get_info()
        slot_gb = args.ws_limit_gb if args.ws_limit_gb > 0 else 24.0
        budget = free_b - int(2 * slot_gb * (1 << 30) * 1.15) - (6 << 30)
        keep = int(np.searchsorted(np.cumsum((lens + 63) // 64 * 64 * 4), budget))
        if keep < len(lens):
            notes["trimmed"] = f"{len(lens)} -> {keep} songs: the PCM of the full share does not fit {free_b / 2**30:.0f} GiB free"
            global_idx, lens = global_idx[:keep], lens[:keep]
        n = len(global_idx)
        if args.ws_limit_gb <= 0:
            args.ws_limit_gb = slot_gb
        workload = (f"configs[4]: mixed-duration corpus, {per_gpu} songs per GPU x {world} GPU(s) = {n_total} songs "
                    f"(8 GPUs = the 50 000-song corpus), durations uniform 30 s - 10 min (seed {args.seed}), sharded by "
                    f"sample count ({n} songs / {int(lens.sum())} samples on this rank), length-bucketed chunks streamed "
                    f"through two {args.ws_limit_gb:g} GiB workspace slots, full 23-feature set")
        N = int(lens.mean())
    d = 23 if version == 2 else 20
    if args.ws_limit_gb > 0:
        ctx.set_workspace_limit(int(args.ws_limit_gb * (1 << 30)))

    padded = (lens + np.uint64(63)) // np.uint64(64) * np.uint64(64)
    offs = np.zeros(n, np.uint64)
    offs[1:] = np.cumsum(padded)[:-1]
    total_samples = int(lens.sum())
    pcm = torch.empty(int(padded.sum()) + 64, dtype=torch.float32, device="cuda")
    if musical_songs is not None:
        for o, x in zip(offs, musical_songs):
            pcm[int(o): int(o) + len(x)] = torch.from_numpy(x).cuda()
    else:
        # white noise written straight into HBM; song g of the corpus uses generator index g whatever the rank count
        ctx.synth_white_noise(pcm, offs, lens, song_index=global_idx)
    out = torch.empty((n, d), dtype=torch.float32, device="cuda")
    status = torch.empty((n,), dtype=torch.int32, device="cuda")
    global_idx_dev = torch.as_tensor(global_idx, device="cuda")  # uploaded once, not per step
    n_local_max = max(len(s) for s in shards) if args.config not in ("batch", "musical") else n
    D_block = None
    if args.config == "library":
        lo, hi = row_block(n_total, rank, world)
        D_block = torch.empty((hi - lo, n_total), dtype=torch.float32, device="cuda")

    def step():
        ctx.analyze(pcm, offs, lens, version, out=out, status=status)
        full = out
        if world > 1 and args.share_device:   # gloo gathers host tensors only
            full = all_gather_features(out.cpu(), global_idx_dev.cpu(), n_total, n_local_max=n_local_max).cuda()
        elif world > 1:
            full = all_gather_features(out, global_idx_dev, n_total, n_local_max=n_local_max)
        if args.config == "library":
            lo, hi = row_block(n_total, rank, world)
            if world == 1:
                ctx.pairwise(full, full, "euclidean", out=D_block)        # the whole matrix: symmetric kernel
            else:
                ctx.pairwise(full[lo:hi], full, "euclidean", out=D_block)  # this rank's row block, no exchange
        return full

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # the timed region: exactly --steps steps, no per-kernel events inside it
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full = step()
    fence()
    elapsed = time.perf_counter() - t0
    # the same steps again with HIP events around every launch (on the stream each kernel runs on): the kernel durations of
    # the roofline.  Kept out of the timed region: the events cost a few microseconds per launch.
    ctx.profile_enable(True)
    ctx.profile_reset()
    t0p = time.perf_counter()
    for _ in range(args.steps):
        full = step()
    fence()
    elapsed_profiled = time.perf_counter() - t0p
    prof = ctx.profile()
    ctx.profile_enable(False)
    chunks = ctx.last_chunks()
    red_dev = "cpu" if args.share_device else "cuda"
    tot = torch.tensor([float(n), float(total_samples)], dtype=torch.float64, device=red_dev)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    job_songs, job_samples = float(tot[0].item()), float(tot[1].item())
    value = job_songs * args.steps / elapsed

    if rank == 0:
        # ---- HBM roofline of the dominant kernel: algorithmic bytes (4*N + 4*d per song, PCM read once
        # + the feature row) of one launch / its average HIP-event duration (chunked runs: per chunk launch) ----
        skip = ("pairwise_kernel", "synth_kernel", "set_distance_kernel", "song_to_song_kernel")
        analysis_kernels = {k: v for k, v in prof.items() if k not in skip}
        dom = max(analysis_kernels, key=lambda k: analysis_kernels[k][0])
        dom_ms = analysis_kernels[dom][0] / analysis_kernels[dom][1]
        launches_per_step = analysis_kernels[dom][1] / args.steps
        algo_bytes_step = 4.0 * total_samples + 4.0 * d * n
        algo_bytes = algo_bytes_step / launches_per_step
        achieved = algo_bytes / (dom_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                    "avg_launch_ms": round(dom_ms, 4), "launches_per_step": launches_per_step,
                    "algorithmic_bytes_per_launch": algo_bytes,
                    "whole_step_GBps": round(algo_bytes_step / (elapsed / args.steps) / 1e9, 2),
                    "kernels_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in sorted(prof.items())},
                    "kernel_durations_from": f"{args.steps} steps run right after the timed region with HIP events around every "
                                             f"launch ({elapsed_profiled / args.steps * 1e3:.3f} ms per step with the events)",
                    "note": "by operation count the path is FP32-vector/LDS bound (roofline_fp32); see DESIGN.md section 3"}
        # HBM bytes of the dominant kernel from the PMC passes (FETCH_SIZE / WRITE_SIZE, tests/tools/hbm_traffic.sh).  The
        # counters cannot be read from inside the process they profile, so the figure comes from profiles/hbm_traffic.json
        # -- which records the hash of the kernel sources it was measured on and is REFUSED when they have changed since --
        # or, with --traffic, is measured now by running the PMC passes on this box.
        if args.traffic:
            try:
                subprocess.run(["bash", os.path.join(ROOT, "tests", "tools", "hbm_traffic.sh")], cwd=ROOT, timeout=900,
                               capture_output=True, env=dict(os.environ, SONGS="256", TRAFFIC_OUT=os.path.join(ROOT, "gpurun_out", "hbm")))
            except Exception as e:  # noqa: BLE001
                roofline["traffic_source"] = f"--traffic failed: {type(e).__name__}"
        for traffic_file, label in ((os.path.join(ROOT, "gpurun_out", "hbm", "hbm_traffic.json") if args.traffic else "", "measured in this run"),
                                    (os.path.join(ROOT, "profiles", "hbm_traffic.json"), "profiles/hbm_traffic.json")):
            if not traffic_file or not os.path.exists(traffic_file) or roofline["traffic"] is not None:
                continue
            try:
                tf = json.load(open(traffic_file))
                now = kernel_sources_sha()
                if tf.get("kernel_sources_sha256") != now:
                    roofline["traffic_source"] = (f"{label} REFUSED: measured on kernel sources {str(tf.get('kernel_sources_sha256'))[:12]}, "
                                                  f"this build is {now[:12]} (re-run tests/tools/hbm_traffic.sh)")
                    continue
                if tf.get("kernel") == dom and tf.get("songs_per_launch") and args.config != "mixed":
                    roofline["traffic"] = tf["bytes_per_launch"] * (n / launches_per_step / tf["songs_per_launch"])
                    roofline["traffic_over_algorithmic"] = round(roofline["traffic"] / algo_bytes, 3)
                    roofline["traffic_source"] = (f"{label}@{now[:12]}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), "
                                                  f"{tf['songs_per_launch']} songs per launch, scaled by songs; FETCH_SIZE x2 (gfx950), KiB -> bytes")
            except Exception as e:  # noqa: BLE001
                roofline["traffic_source"] = f"{label}: {type(e).__name__}"
        flops_step = FLOP_PER_SAMPLE * total_samples
        fp32 = {"bound": "fp32-vector", "flops": flops_step, "achieved": round(flops_step / (elapsed / args.steps) / 1e12, 3),
                "peak": FP32_PEAK / 1e12, "unit": "TFLOP/s",
                "frac": round(flops_step / (elapsed / args.steps) / FP32_PEAK, 5),
                "note": "whole step on this GPU; 1.4e9 FP32-equivalent flop per 3-min song (real-input FFT counts), scaled by samples"}

        # what BINDS the dominant kernel: its VALU-busy fraction from the committed counter pass (tests/tools/valu_busy.sh ->
        # profiles/valu_busy.json; like the traffic figure it cannot be read from inside the process it profiles and is refused
        # when the kernel sources have changed since it was measured)
        valu = {"bound": "valu-issue", "kernel": dom, "frac": None, "unit": "fraction of SIMD cycles with a VALU instruction in flight"}
        try:
            vb = json.load(open(os.path.join(ROOT, "profiles", "valu_busy.json")))
            if vb.get("kernel_sources_sha256") != kernel_sources_sha():
                valu["source"] = "profiles/valu_busy.json REFUSED: measured on other kernel sources (re-run tests/tools/valu_busy.sh)"
            else:
                valu["frac"] = vb["kernels"][dom]["valu_busy"]
                valu["all"] = {k: v["valu_busy"] for k, v in vb["kernels"].items()}
                valu["source"] = "profiles/valu_busy.json: " + vb.get("note", "")
        except Exception as e:  # noqa: BLE001
            valu["source"] = f"profiles/valu_busy.json: {type(e).__name__}"

        result = {
            "metric": "songs/sec (3-min 22 050 Hz f32) at 1/2/4/8 GPU; HBM GB/s vs roofline",
            "value": round(value, 2), "unit": "songs/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32",
            "data": ("synthetic white noise (Philox4x32-10, uniform [-0.5,0.5)), generated in HBM" if musical_songs is None else
                     "synthetic musical signals (seeded generator, tests/tools/musical_check.py), generated on the host, resident in HBM before the timed region"),
            "config": {"workload": workload, "name": args.config, "songs_per_gpu": n, "songs_total": int(job_songs),
                       "samples_per_song": N, "features": d, "chunks_per_step": int(chunks),
                       "parallelism": f"songs sharded x{world}, all-gather of feature rows" if world > 1 else "single GPU"},
            "samples_per_sec": round(job_samples * args.steps / elapsed, 1),
            "three_minute_song_equivalents_per_sec": round(job_samples * args.steps / elapsed / SONG_SAMPLES, 2),
            "roofline": roofline, "roofline_fp32": fp32, "roofline_valu": valu,
        }
        result.update(notes)
        if args.share_device:
            result["share_device"] = (f"{world} ranks on ONE device, collectives over gloo through the host: a check of the "
                                      f"N > 1 control flow on a 1-GPU box, not a measurement")

        # The headline numbers above are complete at this point; the sections below are extras.  Each one is guarded so
        # that a failure there (a full host, no room for the 40 GB distance matrix, ...) is reported inside the JSON
        # line instead of losing it.
        def guarded(name, fn):
            try:
                result[name] = fn()
            except Exception as e:  # noqa: BLE001
                result[name] = {"error": f"{type(e).__name__}: {e}"[:300]}

        # ---- parity spot check inside the bench run + CPU baseline on the same songs ----
        def section_cpu_baseline():
            if args.config == "mixed":
                # 16 songs spread evenly over the length buckets, shortest to longest (~80 min of audio: the oracle
                # takes ~7 s per pass for the 10-minute song)
                order = np.argsort(lens)
                picks = sorted({int(order[int(q * (n - 1))]) for q in np.linspace(0.0, 1.0, 16)})
            else:
                # spread over the whole launch grid (not the first songs of it): every n / cpu_songs-th song
                k = min(args.cpu_songs, n)
                picks = [int(i) for i in np.linspace(0, n - 1, k).round().astype(int)]
            cb, ref = cpu_baseline(global_idx[picks], lens[picks], version,
                                   songs=[musical_songs[i] for i in picks] if musical_songs is not None else None)
            got = out[picks].cpu().numpy()
            err = np.abs(got - ref)
            cb["checked_songs"] = len(picks)
            cb["checked_song_indices"] = f"{picks[0]}..{picks[-1]} ({len(picks)} songs spread evenly over the {n} of the batch)"
            cb["max_abs_err_vs_gpu_non_tempo"] = float(err[:, 1:].max())
            # tempo against the reference's own tolerance (1e-5, src/song/mod.rs:582-590), as a histogram
            cb["tempo_abs_err"] = {"over_1e-5": int((err[:, 0] > 1e-5).sum()), "over_3e-5": int((err[:, 0] > 3e-5).sum()),
                                   "over_1e-4": int((err[:, 0] > 1e-4).sum()), "max": float(err[:, 0].max()),
                                   "songs_over_1e-5": [int(picks[i]) for i in np.flatnonzero(err[:, 0] > 1e-5)][:32]}
            cb["tempo_mismatches"] = int((err[:, 0] > 1e-4).sum())
            return cb

        if not args.no_cpu_baseline and world == 1 and N >= 8192:
            guarded("cpu_baseline", section_cpu_baseline)
        elif not args.no_cpu_baseline:
            result["cpu_baseline"] = None

        if args.config == "library":
            pk = prof.get("pairwise_kernel")
            if pk:
                rows = D_block.shape[0]
                kms = pk[0] / pk[1]
                pbytes = 4.0 * rows * n_total + 4.0 * d * (rows + n_total)
                result["pairwise_row_block"] = {"rows": rows, "cols": n_total, "kernel_ms": round(kms, 3),
                                                "GBps": round(pbytes / (kms * 1e-3) / 1e9, 1)}

        extras = args.config == "batch" and world == 1
        if musical_songs is not None:
            result["config"]["not_the_metric_workload"] = "BASELINE.json's metric is quoted on white noise: --config batch"

        # ---- PCIe-inclusive rate of the host-buffer entry points (never `value`) ----
        def section_host_feed():
            import ctypes as C

            from bliss_rs_amd import _ffi

            hf = min(args.host_feed_songs, n)
            L = _ffi.lib()
            h_f32 = torch.empty(hf * N, dtype=torch.float32, pin_memory=True)
            h_f32.copy_(pcm[: hf * N])
            h_s16 = torch.empty(hf * N, dtype=torch.int16, pin_memory=True)
            h_s16.copy_((pcm[: hf * N] * 32768.0).round().clamp(-32768, 32767).to(torch.int16))
            pageable = h_f32.numpy().copy()
            o64, l64 = offs[:hf].copy(), lens[:hf].copy()
            res = np.empty((hf, d), np.float32)
            st = np.empty(hf, np.int32)

            def run(fn, ptr):
                t0 = time.perf_counter()
                _ffi.check(fn(ptr, o64.ctypes.data_as(C.POINTER(C.c_uint64)), l64.ctypes.data_as(C.POINTER(C.c_uint64)), hf, 2,
                              res.ctypes.data, st.ctypes.data_as(C.POINTER(C.c_int32))))
                return hf / (time.perf_counter() - t0)

            def med3(f):   # every number of this section: one untimed call, then the median of three
                f()
                return round(statistics.median(f() for _ in range(3)), 1)

            pageable_s16 = h_s16.numpy().copy()
            feed = {"songs": hf, "each": "median of 3 calls after one untimed call",
                    "f32_pinned_songs_per_sec": med3(lambda: run(L.blissgpu_analyze_batch, h_f32.data_ptr())),
                    "f32_pageable_songs_per_sec": med3(lambda: run(L.blissgpu_analyze_batch, pageable.ctypes.data)),
                    "s16_pinned_songs_per_sec": med3(lambda: run(L.blissgpu_analyze_batch_s16, h_s16.data_ptr())),
                    "s16_pageable_songs_per_sec": med3(lambda: run(L.blissgpu_analyze_batch_s16, pageable_s16.ctypes.data))}
            feed["f32_pinned_GBps"] = round(feed["f32_pinned_songs_per_sec"] * N * 4 / 1e9, 2)
            feed["f32_pageable_over_pinned"] = round(feed["f32_pageable_songs_per_sec"] / feed["f32_pinned_songs_per_sec"], 3)
            feed["s16_pageable_over_pinned"] = round(feed["s16_pageable_songs_per_sec"] / feed["s16_pinned_songs_per_sec"], 3)
            feed["note"] = ("blissgpu_analyze_batch[_s16] from host memory: H2D of one group pipelined with the analysis of the previous; "
                            "pageable = ordinary heap memory (what a Rust Vec<f32> is), staged by the library's pinned ring "
                            "(staging_ring.hpp: worker threads fill page-locked slabs ahead of the link)")
            # what a decoder really delivers: 44.1 kHz stereo s16 (31.8 MB per 3-minute song); the device does FFmpegDecoder's
            # conversion (libswresample to mono 22 050 Hz, bit for bit at this rate) in front of the analysis -- blissgpu_analyze_batch_decoded
            hd = min(hf, 64)
            frames44 = 2 * N  # 3 minutes at 44.1 kHz -> N samples at 22 050 Hz
            h_44 = torch.empty(hd * frames44 * 2, dtype=torch.int16, pin_memory=True)
            h_44.random_(-20000, 20000)
            pageable_44 = h_44.numpy().copy()
            res44 = np.empty((hd, d), np.float32)
            st44 = np.empty(hd, np.int32)

            def run44(base):
                songs44 = (_ffi.DecodedSong * hd)()
                for i in range(hd):
                    songs44[i] = _ffi.DecodedSong(base + i * frames44 * 4, frames44, 44100, 2, _ffi.SAMPLE_S16)
                t0 = time.perf_counter()
                _ffi.check(L.blissgpu_analyze_batch_decoded(songs44, hd, 2, res44.ctypes.data, st44.ctypes.data_as(C.POINTER(C.c_int32))))
                return hd / (time.perf_counter() - t0)

            feed["decoded_44k1_stereo_s16_songs"] = hd
            feed["decoded_44k1_stereo_s16_songs_per_sec"] = med3(lambda: run44(h_44.data_ptr()))
            rows_pinned = res44.copy()
            feed["decoded_44k1_stereo_s16_pageable_songs_per_sec"] = med3(lambda: run44(pageable_44.ctypes.data))
            feed["decoded_44k1_stereo_s16_pageable_over_pinned"] = round(
                feed["decoded_44k1_stereo_s16_pageable_songs_per_sec"] / feed["decoded_44k1_stereo_s16_songs_per_sec"], 3)
            feed["decoded_44k1_stereo_s16_GBps"] = round(feed["decoded_44k1_stereo_s16_songs_per_sec"] * frames44 * 4 / 1e9, 2)
            feed["pageable_rows_bit_identical_to_pinned"] = bool(np.array_equal(rows_pinned.view(np.uint32), res44.view(np.uint32)))
            assert (st44 == 0).all() and np.isfinite(res44).all()
            return feed

        if extras and not args.no_host_feed and N == SONG_SAMPLES:
            guarded("host_feed", section_host_feed)

        # ---- the per-call forms: latency, threaded small calls, Song::distance ----
        if extras and not args.no_small_calls:
            guarded("small_calls", small_calls)

        # ---- pairwise distances/sec over 100 k feature vectors (BASELINE configs[3]) ----
        def library_vectors():
            g = torch.Generator(device="cuda").manual_seed(1234)
            return torch.rand((args.pairwise_n, 23), generator=g, device="cuda", dtype=torch.float32) * 2 - 1

        def section_pairwise():
            m = args.pairwise_n
            A = library_vectors()
            B = A.clone()
            D = torch.empty((m, m), dtype=torch.float32, device="cuda")
            res = {"n": m, "d": 23, "metric": "euclidean"}
            for name, rhs in (("self", A), ("general", B)):
                ctx.pairwise(A, rhs, "euclidean", out=D)
                torch.cuda.synchronize()
                ctx.profile_enable(True)
                ctx.profile_reset()
                reps = 3
                t0 = time.perf_counter()
                for _ in range(reps):
                    ctx.pairwise(A, rhs, "euclidean", out=D)
                torch.cuda.synchronize()
                dtp = (time.perf_counter() - t0) / reps
                pms = ctx.profile()["pairwise_kernel"]
                ctx.profile_enable(False)
                kms = pms[0] / pms[1]
                pbytes = 4.0 * m * m + 4.0 * 23 * 2 * m
                sec = {"pairs_per_sec": round(m * m / dtp, 1), "ms": round(dtp * 1e3, 3), "kernel_ms": round(kms, 3),
                       "roofline": {"bound": "hbm", "achieved": round(pbytes / (kms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                                    "unit": "GB/s", "frac": round(pbytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
                if name == "self":
                    res.update(sec)
                    res["note"] = "self-distance matrix of one library (A == B): upper block triangle computed, mirrored on store"
                else:
                    res["general_A_ne_B"] = sec
            # the reference's per-pair loop (src/playlist.rs:65-71, without its per-call allocations) on the host cores: a
            # 1000 x m slab of the same matrix through the oracle, extrapolated to m x m -- and checked
            try:
                O = oracle_mod()
                A_h = A.cpu().numpy()
                slab = min(1000, m)
                threads = min(os.cpu_count() or 1, 64)
                t0 = time.perf_counter()
                ref = O.pairwise(A_h[:slab], A_h, "euclidean", None, threads)
                dtc = time.perf_counter() - t0
                ctx.pairwise(A, B, "euclidean", out=D)
                torch.cuda.synchronize()
                res["cpu_baseline"] = {"pairs_per_sec": round(slab * m / dtc, 1), "threads": threads, "kind": "port",
                                       "sample": f"oracle pairwise over a {slab} x {m} slab ({dtc:.2f} s), the m x m matrix would take {dtc * m / slab:.0f} s",
                                       "slab_bit_identical_to_gpu": bool(np.array_equal(D[:slab].cpu().numpy(), ref))}
            except Exception as e:  # noqa: BLE001
                res["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:200]}
            return res

        if extras and not args.no_pairwise:
            guarded("pairwise", section_pairwise)

        # ---- playlist ordering over the same 100 k-vector library: closest_to_songs + song_to_song ----
        def section_playlist():
            O = oracle_mod()
            m = args.pairwise_n
            A = library_vectors()
            seeds = A[:3].clone()
            ctx.closest_to_songs(seeds, A, "euclidean")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):   # (a single call of ~0.2 ms is timer and interpreter noise: mean of 20)
                order = ctx.closest_to_songs(seeds, A, "euclidean")
            torch.cuda.synchronize()
            t_sort = (time.perf_counter() - t0) / 20
            t0 = time.perf_counter()
            ctx.song_to_song(seeds[:1], A, "euclidean")
            torch.cuda.synchronize()
            t_chain = time.perf_counter() - t0
            A_h = A.cpu().numpy()
            t0 = time.perf_counter()
            ref_order, _ = O.closest_to_songs(seeds.cpu().numpy(), A_h, "euclidean")
            t_sort_cpu = time.perf_counter() - t0
            sub = min(4000, m)  # the CPU chain is O(n^2): time a 4000-song pool and scale by (m / sub)^2
            t0 = time.perf_counter()
            ref_chain = O.song_to_song(A_h[:1], A_h[:sub], "euclidean")
            t_chain_cpu = time.perf_counter() - t0
            return {
                "n": m, "d": 23,
                "closest_to_songs_ms": round(t_sort * 1e3, 3), "closest_to_songs_matches_oracle": bool(np.array_equal(order.cpu().numpy(), ref_order)),
                "closest_to_songs_cpu_ms": round(t_sort_cpu * 1e3, 1),
                "song_to_song_s": round(t_chain, 3), "song_to_song_steps_per_sec": round(m / t_chain, 1),
                "song_to_song_cpu_s_extrapolated": round(t_chain_cpu * (m / sub) ** 2, 1),
                "song_to_song_cpu_sample": f"oracle on a {sub}-song pool ({t_chain_cpu:.2f} s, 1 core), scaled by (n / {sub})^2",
                "song_to_song_sample_matches_oracle": bool(np.array_equal(
                    ctx.song_to_song(A[:1], A[:sub], "euclidean").cpu().numpy(), ref_chain)),
            }

        if extras and not args.no_playlist and not args.no_pairwise:
            guarded("playlist", section_playlist)
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
