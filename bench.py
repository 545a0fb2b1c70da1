#!/usr/bin/env python3
"""Benchmark of the hot path named by BASELINE.json: songs/sec for batched 3-minute 22 050 Hz f32 analysis
(all 23 features) on N MI355X, synthetic white-noise PCM resident in HBM, plus the HBM roofline of the
dominant kernel, the CPU oracle timed beside it, and pairwise distances/sec over 100 k feature vectors.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of Song::analyze over one batch of `--songs` (default 1024 = BASELINE configs[1])
3-minute songs per GPU, followed (N > 1) by the RCCL all-gather of the feature rows.  Scaling is weak:
per-GPU work is fixed.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SONG_SAMPLES = 3969000  # 3 min at 22 050 Hz
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--songs", type=int, default=1024, help="3-minute songs per GPU per step")
    ap.add_argument("--samples", type=int, default=SONG_SAMPLES, help="samples per song")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pairwise", action="store_true")
    ap.add_argument("--pairwise-n", type=int, default=100000)
    ap.add_argument("--cpu-songs", type=int, default=128, help="songs timed on the CPU oracle (also the parity spot check)")
    ap.add_argument("--no-host-feed", action="store_true", help="skip the PCIe-inclusive host-buffer measurement")
    ap.add_argument("--no-playlist", action="store_true", help="skip the playlist-ordering measurement")
    ap.add_argument("--ws-limit-gb", type=float, default=0.0, help="workspace limit (chunks the batch); 0 = library default")
    ap.add_argument("--host-feed-songs", type=int, default=256)
    return ap.parse_args()


def cpu_baseline(n_songs, samples, features_version=2):
    """The oracle (C restatement of the reference algorithm, oracle/) on the host cores: the same
    white-noise songs (bit-identical generator), one song per thread at a time.  The oracle is memory-bound well
    before it runs out of cores (on the 2 x 64-core GPU host it peaks around 32 threads), so two thread counts are
    timed and the better one is reported, with the thread count it used."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O

    ncpu = os.cpu_count() or 1
    pcm = np.concatenate([O.white_noise(i, samples) for i in range(n_songs)])
    offs = np.arange(n_songs, dtype=np.uint64) * np.uint64(samples)
    lens = np.full(n_songs, samples, np.uint64)
    tried = []
    out = None
    for cores in sorted({min(ncpu, n_songs, 32), min(ncpu, n_songs, 64)}):
        t0 = time.perf_counter()
        out, status = O.song_analyze_batch(pcm, offs, lens, features_version, cores)
        tried.append((n_songs / (time.perf_counter() - t0), cores))
    rate, cores = max(tried)
    return {"value": round(rate, 3), "unit": "songs/sec", "cores": cores, "kind": "port",
            "sample": f"{n_songs} of the same {samples}-sample white-noise songs, oracle/bliss_oracle.c; "
                      + ", ".join(f"{c} threads: {r:.1f} songs/s" for r, c in tried) + f" ({ncpu} logical CPUs)"}, out


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    import bliss_rs_amd as bliss
    from bliss_rs_amd.shard import all_gather_features

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the library has no CPU path")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    n, N, d = args.songs, args.samples, 23
    ctx = bliss.Context(local_rank)
    if args.ws_limit_gb > 0:
        ctx.set_workspace_limit(int(args.ws_limit_gb * (1 << 30)))
    offs = np.arange(n, dtype=np.uint64) * np.uint64(N)
    lens = np.full(n, N, np.uint64)
    pcm = torch.empty(n * N, dtype=torch.float32, device="cuda")
    ctx.synth_white_noise(pcm, offs, lens, first_song_index=rank * n)  # global song index = rank*n + i
    out = torch.empty((n, d), dtype=torch.float32, device="cuda")
    status = torch.empty((n,), dtype=torch.int32, device="cuda")
    global_idx = np.arange(rank * n, rank * n + n)
    global_idx_dev = torch.as_tensor(global_idx, device="cuda")  # uploaded once, not per step

    def step():
        ctx.analyze(pcm, offs, lens, 2, out=out, status=status)
        if world > 1:
            return all_gather_features(out, global_idx_dev, world * n, n_local_max=n)
        return out

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ctx.profile_enable(True)
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full = step()
    fence()
    elapsed = time.perf_counter() - t0
    prof = ctx.profile()
    ctx.profile_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    total_songs = world * n * args.steps
    value = total_songs / elapsed

    if rank == 0:
        # ---- roofline of the dominant kernel: algorithmic bytes (4*N + 4*d per song,
        # PCM read once + the feature row) x songs per launch / its average HIP-event duration ----
        analysis_kernels = {k: v for k, v in prof.items() if k not in ("pairwise_kernel", "synth_kernel")}
        dom = max(analysis_kernels, key=lambda k: analysis_kernels[k][0])
        dom_ms = analysis_kernels[dom][0] / analysis_kernels[dom][1]
        algo_bytes = float(n) * (4.0 * N + 4.0 * d)
        achieved = algo_bytes / (dom_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                    "avg_launch_ms": round(dom_ms, 4), "algorithmic_bytes_per_launch": algo_bytes,
                    "kernels_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in sorted(prof.items())},
                    "note": "the path is FP32-vector/LDS bound (~1.4 GFLOP per 15.9 MB song); see DESIGN.md"}
        traffic_file = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(traffic_file):
            try:
                tf = json.load(open(traffic_file))
                if tf.get("kernel") == dom and tf.get("songs_per_launch"):
                    roofline["traffic"] = tf["bytes_per_launch"] * (n / tf["songs_per_launch"])
            except Exception:
                pass

        result = {
            "metric": "songs/sec (3-min 22 050 Hz f32) at 1/2/4/8 GPU; HBM GB/s vs roofline",
            "value": round(value, 2), "unit": "songs/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic white noise (Philox4x32-10, uniform [-0.5,0.5)), generated in HBM",
            "config": {"workload": f"configs[1]: batch of {n} synthetic {N}-sample (3-min) white-noise f32 PCM "
                                   f"buffers per GPU, full 23-feature descriptor set (FeaturesVersion 2)",
                       "songs_per_gpu": n, "samples_per_song": N, "features": d,
                       "parallelism": f"songs sharded x{world}, all-gather of feature rows" if world > 1 else "single GPU"},
            "roofline": roofline,
        }

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
            cb, ref = cpu_baseline(min(args.cpu_songs, n), N)
            got = out[: ref.shape[0]].cpu().numpy()
            err = np.abs(got - ref)
            cb["max_abs_err_vs_gpu_non_tempo"] = float(err[:, 1:].max())
            cb["tempo_mismatches"] = int((err[:, 0] > 1e-4).sum())
            return cb

        if not args.no_cpu_baseline and world == 1 and N >= 8192:
            guarded("cpu_baseline", section_cpu_baseline)
        elif not args.no_cpu_baseline:
            result["cpu_baseline"] = None

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

            run(L.blissgpu_analyze_batch, h_f32.data_ptr())  # warm-up (allocations)
            feed = {"songs": hf,
                    "f32_pinned_songs_per_sec": round(run(L.blissgpu_analyze_batch, h_f32.data_ptr()), 1),
                    "f32_pageable_songs_per_sec": round(run(L.blissgpu_analyze_batch, pageable.ctypes.data), 1),
                    "s16_pinned_songs_per_sec": round(run(L.blissgpu_analyze_batch_s16, h_s16.data_ptr()), 1)}
            feed["f32_pinned_GBps"] = round(feed["f32_pinned_songs_per_sec"] * N * 4 / 1e9, 2)
            feed["note"] = "blissgpu_analyze_batch[_s16] from host memory: H2D of one group pipelined with the analysis of the previous"
            return feed

        if not args.no_host_feed and world == 1 and N >= 8192:
            guarded("host_feed", section_host_feed)

        # ---- pairwise distances/sec over 100 k feature vectors (BASELINE configs[3]) ----
        def library_vectors():
            g = torch.Generator(device="cuda").manual_seed(1234)
            return torch.rand((args.pairwise_n, d), generator=g, device="cuda", dtype=torch.float32) * 2 - 1

        def section_pairwise():
            m = args.pairwise_n
            A = library_vectors()
            D = torch.empty((m, m), dtype=torch.float32, device="cuda")
            ctx.pairwise(A, A, "euclidean", out=D)
            torch.cuda.synchronize()
            ctx.profile_enable(True)
            ctx.profile_reset()
            reps = 3
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.pairwise(A, A, "euclidean", out=D)
            torch.cuda.synchronize()
            dtp = (time.perf_counter() - t0) / reps
            pms = ctx.profile()["pairwise_kernel"]
            ctx.profile_enable(False)
            kms = pms[0] / pms[1]
            pbytes = 4.0 * m * m + 4.0 * d * 2 * m
            return {"n": m, "d": d, "metric": "euclidean", "pairs_per_sec": round(m * m / dtp, 1),
                    "ms": round(dtp * 1e3, 3), "kernel_ms": round(kms, 3),
                    "note": "self-distance matrix of one library (A == B): upper block triangle computed, mirrored on store",
                    "roofline": {"bound": "hbm", "achieved": round(pbytes / (kms * 1e-3) / 1e9, 1),
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(pbytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}

        if not args.no_pairwise and world == 1:
            guarded("pairwise", section_pairwise)

        # ---- playlist ordering over the same 100 k-vector library: closest_to_songs + song_to_song ----
        def section_playlist():
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle as O

            m = args.pairwise_n
            A = library_vectors()
            seeds = A[:3].clone()
            ctx.closest_to_songs(seeds, A, "euclidean")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            order = ctx.closest_to_songs(seeds, A, "euclidean")
            torch.cuda.synchronize()
            t_sort = time.perf_counter() - t0
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
                "n": m, "d": d,
                "closest_to_songs_ms": round(t_sort * 1e3, 3), "closest_to_songs_matches_oracle": bool(np.array_equal(order.cpu().numpy(), ref_order)),
                "closest_to_songs_cpu_ms": round(t_sort_cpu * 1e3, 1),
                "song_to_song_s": round(t_chain, 3), "song_to_song_steps_per_sec": round(m / t_chain, 1),
                "song_to_song_cpu_s_extrapolated": round(t_chain_cpu * (m / sub) ** 2, 1),
                "song_to_song_cpu_sample": f"oracle on a {sub}-song pool ({t_chain_cpu:.2f} s, 1 core), scaled by (n / {sub})^2",
                "song_to_song_sample_matches_oracle": bool(np.array_equal(
                    ctx.song_to_song(A[:1], A[:sub], "euclidean").cpu().numpy(), ref_chain)),
            }

        if not args.no_playlist and not args.no_pairwise and world == 1:
            guarded("playlist", section_playlist)
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
