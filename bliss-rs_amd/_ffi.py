"""ctypes binding of libblissgpu.so (the C ABI declared in include/blissgpu.h).

There is no CPU fallback: if the shared library is missing, or no HIP device is usable, every compute
call raises.  The library is loaded lazily so that `import bliss_rs_amd` (and the build check) works
on a machine without a GPU.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BLISSGPU_LIB: developer aid for A/B timing of two builds of the library on the same box (never a CPU path)
LIB_PATH = os.environ.get("BLISSGPU_LIB") or os.path.join(_HERE, "libblissgpu.so")

OK, ERR_NO_DEVICE, ERR_INVALID, ERR_HIP, ERR_OOM, ERR_NAN, ERR_RCCL, ERR_TIMEOUT = 0, 1, 2, 3, 4, 5, 6, 7
SAMPLE_F32, SAMPLE_S16, SAMPLE_S32 = 0, 1, 2
SAMPLE_RATE = 22050
SONG_OK, SONG_TOO_SHORT = 0, 1
METRIC_EUCLIDEAN, METRIC_COSINE, METRIC_MAHALANOBIS = 0, 1, 2
OPT_SERIAL, OPT_TAIL_MODE, OPT_PIPELINE_CHUNKS, OPT_CAND_BUDGET, OPT_ROLLOFF_EXACT_ALL, OPT_DEBUG_CHROMA, OPT_TAIL_SPLIT = 0, 1, 2, 3, 4, 5, 6
OPT_STFT_SHAPE = 7
OPT_FLUX_ORDER = 8
OPT_STAGE_LANES, OPT_STAGE_SLAB_KIB, OPT_STAGE_SLABS, OPT_STAGE_NUMA = 9, 10, 11, 12

_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)
_vp = C.c_void_p



class DecodedSong(C.Structure):
    """blissgpu_decoded_song: one song as the decoder delivers it (host memory)."""
    _fields_ = [("pcm", C.c_void_p), ("frames", C.c_uint64), ("sample_rate", C.c_uint32), ("channels", C.c_uint16),
                ("sample_format", C.c_uint16)]


# name -> (restype, argtypes); must list every symbol include/blissgpu.h declares
SIGNATURES = {
    "blissgpu_analyze_decoded": (C.c_int, [_vp, C.c_int, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, _vp, _i32p]),
    "blissgpu_analyze_batch_decoded": (C.c_int, [C.POINTER(DecodedSong), C.c_uint32, C.c_uint32, _vp, _i32p]),
    "blissgpu_resampled_len": (C.c_uint64, [C.c_uint64, C.c_uint32]),
    "blissgpu_resample_filter": (C.c_int, [C.c_uint32, _vp, C.c_uint64, _u32p, _u32p]),
    "blissgpu_pcm_decode_device": (C.c_int, [_vp, _vp, C.c_int, C.c_uint32, C.c_uint64, C.c_uint32, _vp]),
    "blissgpu_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "blissgpu_ctx_destroy": (C.c_int, [_vp]),
    "blissgpu_ctx_set_stream": (C.c_int, [_vp, _vp]),
    "blissgpu_ctx_get_stream": (_vp, [_vp]),
    "blissgpu_ctx_wait_stream": (C.c_int, [_vp, _vp]),
    "blissgpu_ctx_signal_stream": (C.c_int, [_vp, _vp]),
    "blissgpu_ctx_set_workspace_limit": (C.c_int, [_vp, C.c_uint64]),
    "blissgpu_ctx_get_workspace_limit": (C.c_uint64, [_vp]),
    "blissgpu_ctx_synchronize": (C.c_int, [_vp]),
    "blissgpu_ctx_set_option": (C.c_int, [_vp, C.c_int, C.c_int64]),
    "blissgpu_ctx_staged_bytes": (C.c_uint64, [_vp]),
    "blissgpu_default_ctx": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "blissgpu_default_device_count": (C.c_int, []),
    "blissgpu_default_device": (C.c_int, [C.c_int]),
    "blissgpu_default_device_batches": (C.c_uint64, [C.c_int]),
    "blissgpu_set_single_song_timeout_ms": (C.c_int, [C.c_int64]),
    "blissgpu_default_reset": (C.c_int, []),
    "blissgpu_feature_count": (C.c_uint32, [C.c_uint32]),
    "blissgpu_analyze": (C.c_int, [_vp, C.c_uint64, C.c_uint32, _vp, _i32p]),
    "blissgpu_analyze_interleaved": (C.c_int, [_vp, C.c_int, C.c_uint32, C.c_uint64, C.c_uint32, _vp, _i32p]),
    "blissgpu_analyze_batch_interleaved": (C.c_int, [_vp, C.c_int, C.c_uint32, _u64p, _u64p, C.c_uint32, C.c_uint32, _vp, _i32p]),
    "blissgpu_pcm_downmix_device": (C.c_int, [_vp, _vp, C.c_int, C.c_uint32, C.c_uint64, _vp]),
    "blissgpu_debug_last_chunks": (C.c_uint64, [_vp]),
    "blissgpu_analyze_batch": (C.c_int, [_vp, _u64p, _u64p, C.c_uint32, C.c_uint32, _vp, _i32p]),
    "blissgpu_analyze_batch_s16": (C.c_int, [_vp, _u64p, _u64p, C.c_uint32, C.c_uint32, _vp, _i32p]),
    "blissgpu_pcm_s16_to_f32_device": (C.c_int, [_vp, _vp, C.c_uint64, _vp]),
    "blissgpu_host_alloc": (C.c_int, [C.POINTER(_vp), C.c_uint64]),
    "blissgpu_host_free": (C.c_int, [_vp]),
    "blissgpu_analyze_batch_device": (C.c_int, [_vp, _vp, _u64p, _u64p, C.c_uint32, C.c_uint32, _vp, _vp]),
    "blissgpu_distance": (C.c_int, [_vp, _vp, C.c_uint32, C.c_int, _vp, _f32p]),
    "blissgpu_pairwise": (C.c_int, [_vp, C.c_uint64, _vp, C.c_uint64, C.c_uint32, C.c_int, _vp, _vp]),
    "blissgpu_pairwise_device": (C.c_int, [_vp, _vp, C.c_uint64, _vp, C.c_uint64, C.c_uint32, C.c_int, _vp, _vp,
                                           C.c_uint64]),
    "blissgpu_set_distance": (C.c_int, [_vp, C.c_uint32, _vp, C.c_uint64, C.c_uint32, C.c_int, _vp, _vp]),
    "blissgpu_closest_to_songs": (C.c_int, [_vp, C.c_uint32, _vp, C.c_uint64, C.c_uint32, C.c_int, _vp, _vp, _vp]),
    "blissgpu_song_to_song": (C.c_int, [_vp, C.c_uint32, _vp, C.c_uint64, C.c_uint32, C.c_int, _vp, _vp]),
    "blissgpu_set_distance_device": (C.c_int, [_vp, _vp, C.c_uint32, _vp, C.c_uint64, C.c_uint32, C.c_int, _vp, _vp]),
    "blissgpu_closest_to_songs_device": (C.c_int, [_vp, _vp, C.c_uint32, _vp, C.c_uint64, C.c_uint32, C.c_int, _vp, _vp,
                                                   _vp]),
    "blissgpu_song_to_song_device": (C.c_int, [_vp, _vp, C.c_uint32, _vp, C.c_uint64, C.c_uint32, C.c_int, _vp, _vp]),
    "blissgpu_shard_plan": (C.c_int, [_u64p, C.c_uint32, C.c_uint32, _u32p]),
    "blissgpu_row_block": (None, [C.c_uint64, C.c_uint32, C.c_uint32, _u64p, _u64p]),
    "blissgpu_node_create": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(_vp)]),
    "blissgpu_node_destroy": (C.c_int, [_vp]),
    "blissgpu_node_device_count": (C.c_int, [_vp]),
    "blissgpu_node_ctx": (_vp, [_vp, C.c_int]),
    "blissgpu_node_shard": (C.c_int, [_vp, _u64p, C.c_uint32, _u32p]),
    "blissgpu_node_row_block": (None, [_vp, C.c_uint64, C.c_int, _u64p, _u64p]),
    "blissgpu_node_analyze": (C.c_int, [_vp, _vp, _u64p, _u64p, C.c_uint32, C.c_uint32, _vp, _i32p]),
    "blissgpu_node_analyze_device": (C.c_int, [_vp, C.POINTER(_vp), _u64p, _u64p, _u32p, C.c_uint32, C.c_uint32]),
    "blissgpu_node_features": (_vp, [_vp, C.c_int]),
    "blissgpu_node_pairwise": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "blissgpu_node_synchronize": (C.c_int, [_vp]),
    "blissgpu_feature_weights": (C.c_int, [C.c_uint32, _vp]),
    "blissgpu_malloc": (C.c_int, [C.POINTER(_vp), C.c_uint64]),
    "blissgpu_free": (C.c_int, [_vp]),
    "blissgpu_memcpy_h2d": (C.c_int, [_vp, _vp, _vp, C.c_uint64]),
    "blissgpu_memcpy_d2h": (C.c_int, [_vp, _vp, _vp, C.c_uint64]),
    "blissgpu_synth_white_noise_device": (C.c_int, [_vp, _vp, _u64p, _u64p, C.c_uint32, C.c_uint32]),
    "blissgpu_synth_white_noise_indexed_device": (C.c_int, [_vp, _vp, _u64p, _u64p, _u32p, C.c_uint32]),
    "blissgpu_profile_enable": (C.c_int, [_vp, C.c_int]),
    "blissgpu_profile_reset": (C.c_int, [_vp]),
    "blissgpu_profile_kernel_count": (C.c_int, []),
    "blissgpu_profile_kernel_name": (C.c_char_p, [C.c_int]),
    "blissgpu_profile_get": (C.c_int, [_vp, C.c_int, _f64p, _u64p]),
    "blissgpu_debug_last_tuning": (C.c_int, [_vp, _f64p, _u32p, C.c_uint32]),
    "blissgpu_debug_fetch": (C.c_int, [_vp, C.c_int, C.c_uint32, _vp, C.c_uint64, _u64p]),
    "blissgpu_strerror": (C.c_char_p, [C.c_int]),
    "blissgpu_last_error": (C.c_char_p, []),
    "blissgpu_version": (C.c_char_p, []),
}

_lib = None


class BlissGpuError(RuntimeError):
    def __init__(self, code, detail):
        super().__init__(f"blissgpu error {code}: {detail}")
        self.code = code


def lib():
    """Load libblissgpu.so (fails loudly when the HIP extension has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        try:  # share torch's HIP runtime instance when torch is around (same SONAME libamdhip64.so.7)
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is optional for the C ABI
            pass
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != OK:
        L = lib()
        detail = L.blissgpu_last_error().decode() or L.blissgpu_strerror(rc).decode()
        raise BlissGpuError(rc, detail)
