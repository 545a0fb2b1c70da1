"""Device-resident API: a blissgpu context driven with torch tensors (torch supplies device memory,
streams and torch.distributed; the compute is the HIP library)."""
import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _ffi


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


class Context:
    """blissgpu_ctx wrapper.  Tensors passed in must live on the context's device."""

    def __init__(self, device: int = 0, use_torch_stream: bool = True):
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: bliss_rs_amd has no CPU fallback")
        self.torch = torch
        self.device = device
        torch.cuda.set_device(device)
        self._L = _ffi.lib()
        h = C.c_void_p()
        _ffi.check(self._L.blissgpu_ctx_create(device, C.byref(h)))
        self._h = h
        if use_torch_stream:
            self.bind_current_stream()

    @classmethod
    def default(cls, k: int = 0) -> "Context":
        """The k-th default context of the library -- the one the entry points without a context argument run on
        (blissgpu_default_ctx).  Borrowed: closing this object leaves the context alone."""
        import torch

        self = cls.__new__(cls)
        self.torch = torch
        self._L = _ffi.lib()
        h = C.c_void_p()
        _ffi.check(self._L.blissgpu_default_ctx(int(k), C.byref(h)))
        self._h = h
        self._borrowed = True
        self.device = self._L.blissgpu_default_device(int(k))
        return self

    def staged_bytes(self) -> int:
        """Bytes of pageable host PCM staged through this context's pinned ring so far (blissgpu_ctx_staged_bytes)."""
        return int(self._L.blissgpu_ctx_staged_bytes(self._h))

    def bind_current_stream(self):
        """Launch on torch's current stream when it is a real stream; the legacy default stream (handle 0) cannot be
        adopted (NULL means "the context's own stream"), so in that case every call below is ordered against it with
        events instead (`_pre` / `_post`)."""
        s = self.torch.cuda.current_stream(self.device).cuda_stream
        _ffi.check(self._L.blissgpu_ctx_set_stream(self._h, C.c_void_p(s)))

    def _torch_stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _pre(self):
        """the context's stream waits for the inputs torch has queued on its current stream"""
        _ffi.check(self._L.blissgpu_ctx_wait_stream(self._h, self._torch_stream()))

    def _post(self):
        """torch's current stream waits for the results queued on the context's stream"""
        _ffi.check(self._L.blissgpu_ctx_signal_stream(self._h, self._torch_stream()))

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed", False):
                self._L.blissgpu_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    def synchronize(self):
        _ffi.check(self._L.blissgpu_ctx_synchronize(self._h))

    OPTIONS = {"serial": _ffi.OPT_SERIAL, "tail_mode": _ffi.OPT_TAIL_MODE, "pipeline_chunks": _ffi.OPT_PIPELINE_CHUNKS,
               "cand_budget": _ffi.OPT_CAND_BUDGET, "rolloff_exact_all": _ffi.OPT_ROLLOFF_EXACT_ALL,
               "debug_chroma": _ffi.OPT_DEBUG_CHROMA, "tail_split": _ffi.OPT_TAIL_SPLIT,
               "stft_shape": _ffi.OPT_STFT_SHAPE, "flux_order": _ffi.OPT_FLUX_ORDER,
               "stage_lanes": _ffi.OPT_STAGE_LANES, "stage_slab_kib": _ffi.OPT_STAGE_SLAB_KIB, "stage_slabs": _ffi.OPT_STAGE_SLABS,
               "stage_numa": _ffi.OPT_STAGE_NUMA}

    def set_option(self, name: str, value: int):
        """Scheduling knobs for the measurement tools and the tests (blissgpu_ctx_set_option)."""
        _ffi.check(self._L.blissgpu_ctx_set_option(self._h, self.OPTIONS[name], int(value)))

    def set_workspace_limit(self, nbytes: int):
        """Scratch bytes of ONE chunk slot; larger batches stream through the two slots in length-bucketed chunks."""
        _ffi.check(self._L.blissgpu_ctx_set_workspace_limit(self._h, nbytes))

    def workspace_limit(self) -> int:
        return int(self._L.blissgpu_ctx_get_workspace_limit(self._h))

    def last_chunks(self) -> int:
        """Chunks the last analyze() call was cut into."""
        return int(self._L.blissgpu_debug_last_chunks(self._h))

    # ---- analysis ----
    def analyze(self, pcm, offsets: Sequence[int], lengths: Sequence[int], features_version: int = 2, out=None,
                status=None):
        """pcm: 1-D float32 CUDA tensor holding the songs; returns ([n, d] float32 CUDA tensor, int32 status)."""
        torch = self.torch
        assert pcm.is_cuda and pcm.dtype == torch.float32 and pcm.is_contiguous()
        offsets, lengths = _u64(offsets), _u64(lengths)
        n = len(offsets)
        d = 23 if features_version == 2 else 20
        if out is None:
            out = torch.empty((n, d), dtype=torch.float32, device=pcm.device)
        if status is None:
            status = torch.empty((n,), dtype=torch.int32, device=pcm.device)
        self._pre()
        _ffi.check(self._L.blissgpu_analyze_batch_device(
            self._h, C.c_void_p(pcm.data_ptr()), offsets.ctypes.data_as(C.POINTER(C.c_uint64)),
            lengths.ctypes.data_as(C.POINTER(C.c_uint64)), n, features_version, C.c_void_p(out.data_ptr()),
            C.c_void_p(status.data_ptr())))
        self._post()
        return out, status

    def synth_white_noise(self, pcm, offsets, lengths, first_song_index: int = 0, song_index=None):
        """Benchmark input: song i gets generator index first_song_index + i, or song_index[i] when given."""
        offsets, lengths = _u64(offsets), _u64(lengths)
        self._pre()
        if song_index is None:
            _ffi.check(self._L.blissgpu_synth_white_noise_device(
                self._h, C.c_void_p(pcm.data_ptr()), offsets.ctypes.data_as(C.POINTER(C.c_uint64)),
                lengths.ctypes.data_as(C.POINTER(C.c_uint64)), len(offsets), first_song_index))
        else:
            idx = np.ascontiguousarray(song_index, np.uint32)
            _ffi.check(self._L.blissgpu_synth_white_noise_indexed_device(
                self._h, C.c_void_p(pcm.data_ptr()), offsets.ctypes.data_as(C.POINTER(C.c_uint64)),
                lengths.ctypes.data_as(C.POINTER(C.c_uint64)), idx.ctypes.data_as(C.POINTER(C.c_uint32)), len(offsets)))
        self._post()

    def last_tuning(self, n):
        t = np.empty(n, np.float64)
        b = np.empty(n, np.uint32)
        _ffi.check(self._L.blissgpu_debug_last_tuning(self._h, t.ctypes.data_as(C.POINTER(C.c_double)),
                                                      b.ctypes.data_as(C.POINTER(C.c_uint32)), n))
        return t, b

    TAPS = {"centroid": (0, np.float32), "rolloff": (1, np.float32), "flatness": (2, np.float32),
            "flux": (3, np.float32), "thresholded": (4, np.float32), "run_bpm": (5, np.float32),
            "run_count": (6, np.uint32), "spectrogram": (7, np.float32), "energy256": (8, np.float32),
            "crossings256": (9, np.uint32), "pitch_hist": (10, np.uint32),
            # f64 taps; "chroma" / "interval" need set_option("debug_chroma", 1) before the analysis; for "filter_bank" the
            # `song` argument is the tuning slot (0..99: tuning -0.5 + 0.01 slot, 100: tuning 0.0)
            "chroma": (11, np.float64), "interval": (12, np.float64), "filter_bank": (13, np.float64)}

    def debug_fetch(self, what: str, song: int) -> np.ndarray:
        """Intermediate series of one song of the last chunk (per-stage parity tests)."""
        out = self.debug_fetch_raw(what, song)
        if what == "spectrogram":
            out = out.reshape(-1, 4128)[:, :4097]
        elif what == "filter_bank":
            out = out.reshape(12, 4128)[:, :4097]
        elif what == "chroma":
            out = out.reshape(-1, 12)
        return out

    def debug_fetch_raw(self, what: str, song: int) -> np.ndarray:
        """The tap as the device holds it (flat, padding included)."""
        code, dt = self.TAPS[what]
        n = C.c_uint64()
        probe = np.empty(1, dt)
        _ffi.check(self._L.blissgpu_debug_fetch(self._h, code, song, C.c_void_p(probe.ctypes.data), 0, C.byref(n)))
        out = np.empty(n.value, dt)
        if n.value:
            _ffi.check(self._L.blissgpu_debug_fetch(self._h, code, song, C.c_void_p(out.ctypes.data), n.value, C.byref(n)))
        return out

    # ---- distances ----
    def pairwise(self, A, B, metric: str = "euclidean", M=None, out=None):
        torch = self.torch
        from .playlist import _METRICS

        assert A.is_cuda and B.is_cuda and A.dtype == torch.float32 and B.dtype == torch.float32
        A, B = A.contiguous(), B.contiguous()
        n, d = A.shape
        m = B.shape[0]
        if out is None:
            out = torch.empty((n, m), dtype=torch.float32, device=A.device)
        Mp = None
        if M is not None:
            M = M.contiguous()
            Mp = C.c_void_p(M.data_ptr())
        self._pre()
        _ffi.check(self._L.blissgpu_pairwise_device(self._h, C.c_void_p(A.data_ptr()), n, C.c_void_p(B.data_ptr()), m,
                                                    d, _METRICS[metric], Mp, C.c_void_p(out.data_ptr()), out.stride(0)))
        self._post()
        return out

    def pcm_s16_to_f32(self, pcm_s16, out=None):
        """On-device s16 -> f32 (sample / 32768): FFmpeg's s16 -> flt conversion (src/song/decoder/ffmpeg.rs:36-109)."""
        torch = self.torch
        assert pcm_s16.is_cuda and pcm_s16.dtype == torch.int16
        pcm_s16 = pcm_s16.contiguous()
        if out is None:
            out = torch.empty(pcm_s16.shape, dtype=torch.float32, device=pcm_s16.device)
        self._pre()
        _ffi.check(self._L.blissgpu_pcm_s16_to_f32_device(self._h, C.c_void_p(pcm_s16.data_ptr()), pcm_s16.numel(),
                                                          C.c_void_p(out.data_ptr())))
        self._post()
        return out

    def pcm_downmix(self, pcm, out=None):
        """On-device mono downmix of interleaved [frames, channels] decoder output (int16 or float32): stereo ->
        (L + R) * SQRT_2 / 2, more channels -> their mean (src/song/decoder/symphonia.rs:266-300)."""
        torch = self.torch
        assert pcm.is_cuda and pcm.dim() == 2 and pcm.dtype in (torch.int16, torch.float32)
        pcm = pcm.contiguous()
        frames, channels = pcm.shape
        if out is None:
            out = torch.empty((frames,), dtype=torch.float32, device=pcm.device)
        fmt = _ffi.SAMPLE_S16 if pcm.dtype == torch.int16 else _ffi.SAMPLE_F32
        self._pre()
        _ffi.check(self._L.blissgpu_pcm_downmix_device(self._h, C.c_void_p(pcm.data_ptr()), fmt, channels, frames,
                                                       C.c_void_p(out.data_ptr())))
        self._post()
        return out

    def pcm_decode(self, pcm, sample_rate: int, out=None):
        """On-device conversion of decoder output (1-D mono or [frames, channels]; int16 / int32 / float32) at `sample_rate`
        Hz to the mono 22 050 Hz f32 stream Song::analyze takes: libswresample's default resampler as FFmpegDecoder drives
        it (src/song/decoder/ffmpeg.rs:36-109) -- bit for bit at 44 100 Hz (Adler-32 pins of ffmpeg.rs:433-452, 471-476); other
        rates run the same restatement, held to the oracle only."""
        torch = self.torch
        assert pcm.is_cuda and pcm.dim() in (1, 2) and pcm.dtype in (torch.int16, torch.int32, torch.float32)
        pcm = pcm.contiguous()
        frames = pcm.shape[0]
        channels = 1 if pcm.dim() == 1 else pcm.shape[1]
        n_out = int(self._L.blissgpu_resampled_len(frames, int(sample_rate)))
        if out is None:
            out = torch.empty((n_out,), dtype=torch.float32, device=pcm.device)
        assert out.numel() >= n_out
        fmt = {torch.int16: _ffi.SAMPLE_S16, torch.int32: _ffi.SAMPLE_S32, torch.float32: _ffi.SAMPLE_F32}[pcm.dtype]
        self._pre()
        _ffi.check(self._L.blissgpu_pcm_decode_device(self._h, C.c_void_p(pcm.data_ptr()), fmt, channels, frames,
                                                      int(sample_rate), C.c_void_p(out.data_ptr())))
        self._post()
        return out[:n_out]

    # ---- playlist ordering on device-resident feature matrices (src/playlist.rs:24-59, 256-326) ----
    def _pl_args(self, seeds, cand, M):
        torch = self.torch
        assert seeds.is_cuda and cand.is_cuda and seeds.dtype == torch.float32 and cand.dtype == torch.float32
        seeds, cand = seeds.contiguous(), cand.contiguous()
        if seeds.dim() == 1:
            seeds = seeds[None, :]
        Mp = None
        if M is not None:
            M = M.contiguous()
            Mp = C.c_void_p(M.data_ptr())
        return seeds, cand, M, Mp

    def set_distance(self, seeds, cand, metric: str = "euclidean", M=None):
        """FunctionDistanceMetric::distance of every candidate row to the seed set."""
        from .playlist import _METRICS

        seeds, cand, M, Mp = self._pl_args(seeds, cand, M)
        out = self.torch.empty((cand.shape[0],), dtype=self.torch.float32, device=cand.device)
        self._pre()
        _ffi.check(self._L.blissgpu_set_distance_device(self._h, C.c_void_p(seeds.data_ptr()), seeds.shape[0],
                                                        C.c_void_p(cand.data_ptr()), cand.shape[0], cand.shape[1],
                                                        _METRICS[metric], Mp, C.c_void_p(out.data_ptr())))
        self._post()
        return out

    def closest_to_songs(self, seeds, cand, metric: str = "euclidean", M=None, return_distances=False):
        """Indices of the candidates sorted (stably) by distance to the seed set (int64 tensor)."""
        from .playlist import _METRICS

        torch = self.torch
        seeds, cand, M, Mp = self._pl_args(seeds, cand, M)
        n = cand.shape[0]
        order = torch.empty((n,), dtype=torch.int32, device=cand.device)
        dist = torch.empty((n,), dtype=torch.float32, device=cand.device)
        self._pre()
        _ffi.check(self._L.blissgpu_closest_to_songs_device(self._h, C.c_void_p(seeds.data_ptr()), seeds.shape[0],
                                                            C.c_void_p(cand.data_ptr()), n, cand.shape[1], _METRICS[metric],
                                                            Mp, C.c_void_p(order.data_ptr()), C.c_void_p(dist.data_ptr())))
        self._post()
        order = order.to(torch.int64)
        return (order, dist) if return_distances else order

    def song_to_song(self, seeds, cand, metric: str = "euclidean", M=None):
        """Greedy nearest-neighbour chain over the candidates (int64 index tensor)."""
        from .playlist import _METRICS

        torch = self.torch
        seeds, cand, M, Mp = self._pl_args(seeds, cand, M)
        n = cand.shape[0]
        order = torch.empty((n,), dtype=torch.int32, device=cand.device)
        self._pre()
        _ffi.check(self._L.blissgpu_song_to_song_device(self._h, C.c_void_p(seeds.data_ptr()), seeds.shape[0],
                                                        C.c_void_p(cand.data_ptr()), n, cand.shape[1], _METRICS[metric], Mp,
                                                        C.c_void_p(order.data_ptr())))
        self._post()
        return order.to(torch.int64)

    # ---- profiling ----
    def profile_enable(self, on: bool = True):
        _ffi.check(self._L.blissgpu_profile_enable(self._h, int(on)))

    def profile_reset(self):
        _ffi.check(self._L.blissgpu_profile_reset(self._h))

    def profile(self):
        """{kernel name: (total_ms, launches)} since the last reset."""
        res = {}
        for k in range(self._L.blissgpu_profile_kernel_count()):
            ms, n = C.c_double(), C.c_uint64()
            _ffi.check(self._L.blissgpu_profile_get(self._h, k, C.byref(ms), C.byref(n)))
            if n.value:
                res[self._L.blissgpu_profile_kernel_name(k).decode()] = (ms.value, n.value)
        return res


class Node:
    """blissgpu_node wrapper: ONE process driving several GPUs (RCCL all-gather of the feature rows inside the
    library).  The one-process-per-GPU form on torch.distributed is bliss_rs_amd.shard."""

    def __init__(self, n_devices: int = 1, devices: Optional[Sequence[int]] = None):
        self._L = _ffi.lib()
        h = C.c_void_p()
        dev = (C.c_int * n_devices)(*devices) if devices is not None else None
        _ffi.check(self._L.blissgpu_node_create(n_devices, dev, C.byref(h)))
        self._h = h
        self.n_devices = n_devices

    def close(self):
        if getattr(self, "_h", None):
            self._L.blissgpu_node_destroy(self._h)
            self._h = None

    __del__ = close

    def synth_white_noise(self, rank: int, d_pcm_ptr: int, offsets, lengths, song_index):
        """Benchmark input on `rank`'s device: song i (at d_pcm_ptr + offsets[i]) gets generator index song_index[i]."""
        offsets, lengths = _u64(offsets), _u64(lengths)
        idx = np.ascontiguousarray(song_index, np.uint32)
        ctx = self._L.blissgpu_node_ctx(self._h, rank)
        _ffi.check(self._L.blissgpu_synth_white_noise_indexed_device(
            ctx, C.c_void_p(int(d_pcm_ptr)), offsets.ctypes.data_as(C.POINTER(C.c_uint64)),
            lengths.ctypes.data_as(C.POINTER(C.c_uint64)), idx.ctypes.data_as(C.POINTER(C.c_uint32)), len(offsets)))
        _ffi.check(self._L.blissgpu_ctx_synchronize(ctx))

    def ctx_set_workspace_limit(self, rank: int, nbytes: int):
        """Scratch bytes of one chunk slot of `rank`'s context (several loopback ranks share one GPU's memory)."""
        _ffi.check(self._L.blissgpu_ctx_set_workspace_limit(self._L.blissgpu_node_ctx(self._h, rank), nbytes))

    def shard(self, lengths) -> np.ndarray:
        lengths = _u64(lengths)
        ranks = np.empty(len(lengths), np.uint32)
        _ffi.check(self._L.blissgpu_node_shard(self._h, lengths.ctypes.data_as(C.POINTER(C.c_uint64)), len(lengths),
                                               ranks.ctypes.data_as(C.POINTER(C.c_uint32))))
        return ranks

    def row_block(self, n_rows: int, rank: int):
        lo, hi = C.c_uint64(), C.c_uint64()
        self._L.blissgpu_node_row_block(self._h, n_rows, rank, C.byref(lo), C.byref(hi))
        return lo.value, hi.value

    def analyze(self, pcm: np.ndarray, offsets, lengths, features_version: int = 2):
        """Host PCM -> ([n, d] float32, int32 status); the gathered matrix stays on every device for pairwise()."""
        pcm = np.ascontiguousarray(pcm, np.float32)
        offsets, lengths = _u64(offsets), _u64(lengths)
        n = len(offsets)
        d = 23 if features_version == 2 else 20
        out = np.empty((n, d), np.float32)
        status = np.empty(n, np.int32)
        _ffi.check(self._L.blissgpu_node_analyze(self._h, pcm.ctypes.data, offsets.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                 lengths.ctypes.data_as(C.POINTER(C.c_uint64)), n, features_version,
                                                 out.ctypes.data, status.ctypes.data_as(C.POINTER(C.c_int32))))
        _ffi.check(self._L.blissgpu_node_synchronize(self._h))
        self._n, self._d = n, d
        return out, status

    def analyze_device(self, d_pcm_ptrs: Sequence[int], offsets, lengths, rank_of_song, features_version: int = 2):
        """Device-resident PCM: song i lives on device rank_of_song[i] at d_pcm_ptrs[rank] + offsets[i] (raw pointers)."""
        offsets, lengths = _u64(offsets), _u64(lengths)
        ranks = np.ascontiguousarray(rank_of_song, np.uint32)
        ptrs = (C.c_void_p * self.n_devices)(*[C.c_void_p(int(p)) for p in d_pcm_ptrs])
        _ffi.check(self._L.blissgpu_node_analyze_device(self._h, ptrs, offsets.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                        lengths.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                        ranks.ctypes.data_as(C.POINTER(C.c_uint32)), len(offsets),
                                                        features_version))
        _ffi.check(self._L.blissgpu_node_synchronize(self._h))
        self._n, self._d = len(offsets), 23 if features_version == 2 else 20

    def features(self, rank: int = 0) -> np.ndarray:
        """The gathered matrix as held by `rank`'s device."""
        out = np.empty((self._n, self._d), np.float32)
        ctx = self._L.blissgpu_node_ctx(self._h, rank)
        src = self._L.blissgpu_node_features(self._h, rank)
        _ffi.check(self._L.blissgpu_memcpy_d2h(ctx, out.ctypes.data, src, out.nbytes))
        return out

    def pairwise(self, metric: str = "euclidean", M: Optional[np.ndarray] = None) -> np.ndarray:
        from .playlist import _METRICS

        out = np.empty((self._n, self._n), np.float32)
        Mp = None
        if M is not None:
            M = np.ascontiguousarray(M, np.float32)
            Mp = M.ctypes.data
        _ffi.check(self._L.blissgpu_node_pairwise(self._h, _METRICS[metric], Mp, out.ctypes.data))
        return out
