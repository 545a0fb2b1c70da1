// blissgpu.hip -- host side of the C ABI declared in include/blissgpu.h: context life cycle, constant tables,
// feature-vector distances, playlist ordering, device-memory helpers, profiling and debug taps.  The analysis
// batches (planning, chunk schedule, PCM feed, coalescing front) live in scheduler.hip, the multi-GPU node in node.hip.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <deque>
#include <mutex>
#include <string>
#include <vector>

#include "ctx.hpp"

using namespace bg;

namespace {
thread_local std::string g_last_error;

const char* const kKernelNames[K_COUNT] = {
    "fft512_kernel",     "onset_kernel",      "beat_kernel",   "stft8192_kernel", "tune_select_kernel",
    "tune_pass2_kernel", "tune_final_kernel", "chroma_kernel",     "summary_kernel", "assemble_kernel", "pairwise_kernel", "set_distance_kernel", "song_to_song_kernel", "synth_kernel", "rolloff_fix_kernel"};
}  // namespace

namespace bg {
int fail(int code, const char* what, const char* detail) {
    g_last_error = std::string(what) + ": " + (detail ? detail : "");
    return code;
}
}  // namespace bg

namespace {

template <typename T>
int upload(T** dst, const std::vector<T>& h) {
    HIP_TRY(hipMalloc((void**)dst, h.size() * sizeof(T)));
    HIP_TRY(hipMemcpy(*dst, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return BLISSGPU_OK;
}

int build_tables(blissgpu_ctx* c) {
    const float PI_F = 3.14159265358979323846f;
    std::vector<float2> tw8(8192), tw5(512);
    for (int k = 0; k < 8192; k++) {
        const double a = -2.0 * M_PI * (double)k / 8192.0;
        tw8[k] = make_float2((float)cos(a), (float)sin(a));
    }
    for (int k = 0; k < 512; k++) {
        const double a = -2.0 * M_PI * (double)k / 512.0;
        tw5[k] = make_float2((float)cos(a), (float)sin(a));
    }
    // periodic Hann, evaluated in f32 exactly as src/utils.rs:37-39
    std::vector<float> hann(8192), hannz(512), rwv(BT_LAGLEN), dfwv(BT_WINLEN);
    // Both device tables hold HALF the window: the real-input split needs X = (A + P) / 2, and a power-of-two scale
    // commutes with every rounding of the (linear) transform, so halving the window once removes a multiply per bin
    // and leaves every magnitude bit-identical.
    for (int n = 0; n < 8192; n++) hann[n] = 0.5f * (0.5f - 0.5f * cosf(2.0f * (float)n * PI_F / 8192.0f));
    // hanningz, src/aubio.rs:151-154
    for (int i = 0; i < 512; i++) hannz[i] = 0.5f * (0.5f * (1.0f - cosf(2.0f * PI_F * (float)i / 512.0f)));
    // BeatTracking::new, src/aubio.rs:911-936
    const float rayparam = 60.0f * (float)SAMPLE_RATE / 120.0f / (float)HOP_B;
    const float dfwvnorm = expf((logf(2.0f) / rayparam) * (float)(BT_WINLEN + 2));
    for (int i = 0; i < BT_LAGLEN; i++) {
        const float i_f = (float)(i + 1);
        rwv[i] = (i_f / (rayparam * rayparam)) * expf(-(i_f * i_f) / (2.0f * (rayparam * rayparam)));
    }
    for (int i = 0; i < BT_WINLEN; i++) dfwv[i] = expf((logf(2.0f) / rayparam) * (float)(i + 1)) / dfwvnorm;
    int rc;
    if ((rc = upload(&c->tw8192, tw8))) return rc;
    if ((rc = upload(&c->tw512, tw5))) return rc;
    if ((rc = upload(&c->hann8192, hann))) return rc;
    if ((rc = upload(&c->hannz512, hannz))) return rc;
    if ((rc = upload(&c->bt_rwv, rwv))) return rc;
    if ((rc = upload(&c->bt_dfwv, dfwv))) return rc;
    const size_t bank_elems = (size_t)(N_TUNING + 1) * BANK_ROWS * BANK_PITCH;
    HIP_TRY(hipMalloc((void**)&c->chroma_bank, bank_elems * sizeof(double)));
    launch_chroma_bank(c->chroma_bank, c->own_stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->own_stream));
    c->tables = DeviceTables{c->tw8192, c->tw512, c->hann8192, c->hannz512, c->chroma_bank, c->bt_rwv, c->bt_dfwv};
    return BLISSGPU_OK;
}

// Process-wide default contexts of the entry points that take no context: one per visible HIP device, created on first
// use.  BLISSGPU_DEFAULT_DEVICES="0,2,3" restricts / orders them (like HIP_VISIBLE_DEVICES, but for this library only);
// an ordinal may be named more than once (several contexts sharing a GPU -- how the multi-context front is tested on a
// one-GPU box).
std::mutex g_default_mu;
std::vector<int> g_default_devices;
// A default context is created under ITS OWN mutex (building the 40 MB filter bank takes a while: the seats of an 8-GPU node
// must not queue behind one another, and the counters below must stay readable meanwhile).  A creation that failed is
// remembered and not repeated on every call -- but only a failure that cannot change is remembered for good (no such
// device, another architecture: BLISSGPU_ERR_NO_DEVICE / _INVALID).  Anything else (no memory for the tables while another
// process holds the device, a HIP error) is tried again after a back-off that doubles from 100 ms to 5 s, so a long-running
// host recovers by itself; blissgpu_default_reset() forgets every remembered failure at once.
struct DefaultSeat {
    std::mutex mu;
    blissgpu_ctx* ctx = nullptr;
    int fails = 0;           // consecutive failed creations
    bool permanent = false;  // the last failure cannot change in this process
    std::chrono::steady_clock::time_point retry_at{};
    int rc = BLISSGPU_OK;
    std::string err;
};
std::deque<DefaultSeat> g_default_seats;  // (stable addresses)
std::vector<uint64_t> g_default_batches;
bool g_default_init = false;
std::atomic<int64_t> g_single_song_timeout_ms{600000};

void default_init_locked() {
    if (g_default_init) return;
    g_default_init = true;
    if (const char* e = getenv("BLISSGPU_DEFAULT_DEVICES")) {
        for (const char* p = e; *p;) {
            char* end = nullptr;
            const long v = strtol(p, &end, 10);
            if (end == p) break;
            if (v >= 0 && v < 4096 && g_default_devices.size() < 64) g_default_devices.push_back((int)v);
            p = *end == ',' ? end + 1 : end;
            if (*end && *end != ',') break;
        }
    }
    if (g_default_devices.empty()) {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count < 1) count = 1;  // no device: ctx_create reports it
        for (int k = 0; k < count && k < 64; k++) g_default_devices.push_back(k);
    }
    g_default_seats.resize(g_default_devices.size());
    g_default_batches.assign(g_default_devices.size(), 0);
}

}  // namespace

namespace bg {
int default_ctx_count() {
    std::lock_guard<std::mutex> lk(g_default_mu);
    default_init_locked();
    return (int)g_default_devices.size();
}
int default_ctx_at(int k, blissgpu_ctx** out) {
    DefaultSeat* seat = nullptr;
    int device = 0;
    {
        std::lock_guard<std::mutex> lk(g_default_mu);
        default_init_locked();
        if (k < 0 || k >= (int)g_default_devices.size()) return fail(BLISSGPU_ERR_INVALID, "default context", "no such default device");
        seat = &g_default_seats[(size_t)k];
        device = g_default_devices[(size_t)k];
    }
    std::lock_guard<std::mutex> lk(seat->mu);
    if (!seat->ctx) {
        const auto now = std::chrono::steady_clock::now();
        if (seat->fails == 0 || (!seat->permanent && now >= seat->retry_at)) {
            seat->rc = blissgpu_ctx_create(device, &seat->ctx);
            if (seat->rc) {
                seat->ctx = nullptr;
                seat->fails++;
                seat->permanent = seat->rc == BLISSGPU_ERR_NO_DEVICE || seat->rc == BLISSGPU_ERR_INVALID;
                seat->retry_at = now + std::chrono::milliseconds(std::min(5000, 100 << std::min(seat->fails - 1, 6)));
                seat->err = std::string("default context ") + std::to_string(k) + " (HIP device " + std::to_string(device) + "): " + blissgpu_last_error();
            } else {
                seat->fails = 0;
                seat->permanent = false;
            }
        }
    }
    if (!seat->ctx) return fail(seat->rc, "default context", seat->err.c_str());
    *out = seat->ctx;
    return BLISSGPU_OK;
}
void default_ctx_forget_failures() {
    std::lock_guard<std::mutex> lk(g_default_mu);
    default_init_locked();
    for (DefaultSeat& seat : g_default_seats) {
        std::lock_guard<std::mutex> sl(seat.mu);
        if (!seat.ctx) { seat.fails = 0; seat.permanent = false; seat.rc = BLISSGPU_OK; }
    }
}
// Live contexts per HIP device in this process.  The single-launch sort and the song_to_song chain spin on grid barriers: their
// workgroups must all be resident at once.  One such kernel asks for at most one 256-thread workgroup per CU, so two contexts'
// worth of them always fit beside each other; with more contexts alive on the device the sort falls back to one launch per
// step, and the chain -- which has no multi-launch form -- runs one at a time per device (persistent_kernel_mutex: launch, wait
// for it, release), so that two spinning grids can never hold each other's missing workgroups out of the CUs.
namespace {
std::mutex g_live_mu;
std::vector<int> g_live_contexts;
std::mutex g_persistent_mu[64];
}
std::mutex& persistent_kernel_mutex(int device) { return g_persistent_mu[(unsigned)device % 64u]; }
void live_context_add(int device, int delta) {
    std::lock_guard<std::mutex> lk(g_live_mu);
    if (device < 0) return;
    if ((size_t)device >= g_live_contexts.size()) g_live_contexts.resize((size_t)device + 1, 0);
    g_live_contexts[(size_t)device] += delta;
}
int live_contexts(int device) {
    std::lock_guard<std::mutex> lk(g_live_mu);
    return device >= 0 && (size_t)device < g_live_contexts.size() ? g_live_contexts[(size_t)device] : 0;
}
int64_t single_song_timeout_ms() { return g_single_song_timeout_ms.load(); }
int default_ctx(blissgpu_ctx** out) { return default_ctx_at(0, out); }
void default_ctx_count_batch(int k) {
    std::lock_guard<std::mutex> lk(g_default_mu);
    if (k >= 0 && k < (int)g_default_batches.size()) g_default_batches[k]++;
}
}  // namespace bg

extern "C" {
int blissgpu_default_device_count(void) { return default_ctx_count(); }
int blissgpu_default_ctx(int k, blissgpu_ctx** ctx) {
    if (!ctx) return fail(BLISSGPU_ERR_INVALID, "blissgpu_default_ctx", "NULL argument");
    *ctx = nullptr;
    return default_ctx_at(k, ctx);
}
int blissgpu_set_single_song_timeout_ms(int64_t ms) {
    // "never" (INT64_MAX) must not overflow the nanosecond clock the deadline is computed on: ten years is never
    constexpr int64_t TEN_YEARS_MS = 10LL * 365 * 24 * 3600 * 1000;
    g_single_song_timeout_ms.store(ms > 0 ? std::min(ms, TEN_YEARS_MS) : 600000);
    return BLISSGPU_OK;
}
int blissgpu_default_reset(void) {
    default_ctx_forget_failures();
    front_revive_all();
    return BLISSGPU_OK;
}
int blissgpu_default_device(int k) {
    std::lock_guard<std::mutex> lk(g_default_mu);
    default_init_locked();
    return (k >= 0 && k < (int)g_default_devices.size()) ? g_default_devices[k] : -1;
}
uint64_t blissgpu_default_device_batches(int k) {
    std::lock_guard<std::mutex> lk(g_default_mu);
    return (k >= 0 && k < (int)g_default_batches.size()) ? g_default_batches[k] : 0;
}
}  // extern "C"

namespace {

int is_diag(const float* M, uint32_t d) {
    for (uint32_t i = 0; i < d; i++)
        for (uint32_t j = 0; j < d; j++)
            if (i != j && M[i * d + j] != 0.0f) return 0;
    return 1;
}

}  // namespace

// Locks the context for the duration of one entry point and selects its device.
#define CTX_ENTER(c, who)                                                     \
    if (!(c)) return fail(BLISSGPU_ERR_INVALID, who, "ctx is NULL");          \
    std::lock_guard<std::recursive_mutex> ctx_lock_((c)->mu);                 \
    HIP_TRY(hipSetDevice((c)->device))

extern "C" {

const char* blissgpu_version(void) { return "blissgpu 0.3.0 (gfx950)"; }
const char* blissgpu_last_error(void) { return g_last_error.c_str(); }

const char* blissgpu_strerror(int code) {
    switch (code) {
        case BLISSGPU_OK: return "ok";
        case BLISSGPU_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU path)";
        case BLISSGPU_ERR_INVALID: return "invalid argument";
        case BLISSGPU_ERR_HIP: return "HIP runtime error";
        case BLISSGPU_ERR_NAN: return "a distance is NaN";
        case BLISSGPU_ERR_OOM: return "out of device memory";
        case BLISSGPU_ERR_RCCL: return "RCCL error";
        case BLISSGPU_ERR_TIMEOUT: return "no default context picked the call up within its deadline";
        default: return "unknown error";
    }
}

uint32_t blissgpu_feature_count(uint32_t v) { return v == BLISSGPU_FEATURES_V1 ? 20u : (v == BLISSGPU_FEATURES_V2 ? 23u : 0u); }

int blissgpu_ctx_create(int device, blissgpu_ctx** out) {
    if (!out) return fail(BLISSGPU_ERR_INVALID, "blissgpu_ctx_create", "ctx is NULL");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0 || device < 0 || device >= count)
        return fail(BLISSGPU_ERR_NO_DEVICE, "hipGetDeviceCount", e != hipSuccess ? hipGetErrorString(e) : "no such device");
    HIP_TRY(hipSetDevice(device));
    blissgpu_ctx* c = new blissgpu_ctx();
    c->device = device;
    (void)hipDeviceGetAttribute(&c->n_cus, hipDeviceAttributeMultiprocessorCount, device);
    hipError_t se = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (se != hipSuccess) { delete c; return fail(BLISSGPU_ERR_HIP, "hipStreamCreate", hipGetErrorString(se)); }
    c->stream = c->own_stream;
    // the side streams carry small latency-bound kernels: at high priority they get a CU slot as soon as one frees up
    // instead of queueing behind the thousands of workgroups of an FFT kernel
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    const int prio_aux = prio_greatest;
    se = hipStreamCreateWithPriority(&c->aux_stream, hipStreamNonBlocking, prio_aux);
    if (se == hipSuccess) se = hipStreamCreateWithPriority(&c->chr_stream, hipStreamNonBlocking, prio_greatest);
    if (se == hipSuccess) se = hipEventCreateWithFlags(&c->ev_interop, hipEventDisableTiming);
    if (se == hipSuccess) se = hipHostMalloc((void**)&c->h_scalar, 64, hipHostMallocDefault);
    if (se != hipSuccess) { blissgpu_ctx_destroy(c); return fail(BLISSGPU_ERR_HIP, "aux stream/events", hipGetErrorString(se)); }
    // Scratch limit per chunk slot: a third of what is free now, at most 64 GiB (1024 three-minute songs need ~37 GB).  A
    // batch that needs more runs as several chunks; a chunk that still does not fit (the caller allocated in the meantime)
    // is halved until it does.
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > 0)
        c->ws_limit = std::max<uint64_t>(256ull << 20, std::min<uint64_t>(64ull << 30, free_b / 3));
    else
        c->ws_limit = 16ull << 30;
    int rc = build_tables(c);
    if (rc) { blissgpu_ctx_destroy(c); return rc; }
    live_context_add(device, 1);
    c->counted_live = true;
    *out = c;
    return BLISSGPU_OK;
}

int blissgpu_ctx_destroy(blissgpu_ctx* c) {
    if (!c) return BLISSGPU_OK;
    if (c->counted_live) live_context_add(c->device, -1);
    {
        std::lock_guard<std::recursive_mutex> lk(c->mu);
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        if (c->aux_stream) (void)hipStreamSynchronize(c->aux_stream);
        if (c->chr_stream) (void)hipStreamSynchronize(c->chr_stream);
        if (c->mask_stream) { (void)hipStreamSynchronize(c->mask_stream); (void)hipStreamDestroy(c->mask_stream); }
        for (auto& v : c->events)
            for (auto& ev : v) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
        (void)hipFree(c->tw8192); (void)hipFree(c->tw512); (void)hipFree(c->hann8192); (void)hipFree(c->hannz512);
        (void)hipFree(c->bt_rwv); (void)hipFree(c->bt_dfwv); (void)hipFree(c->chroma_bank);
        scheduler_release(c);
        c->dbg_tuning.release(); c->dbg_nbpms.release(); c->dbg_chroma.release(); c->dbg_interval.release();
        c->pl_sync.release(); c->pl_keys.release(); c->pl_tmp.release(); c->pl_slots.release();
        c->st_a.release(); c->st_b.release(); c->st_m.release(); c->st_dist.release(); c->st_out.release();
        if (c->h_scalar) (void)hipHostFree(c->h_scalar);
        if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
        if (c->chr_stream) (void)hipStreamDestroy(c->chr_stream);
        if (c->ev_interop) (void)hipEventDestroy(c->ev_interop);
        if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    }
    delete c;
    return BLISSGPU_OK;
}

int blissgpu_ctx_set_stream(blissgpu_ctx* c, void* s) {
    CTX_ENTER(c, "blissgpu_ctx_set_stream");
    (void)hipStreamSynchronize(c->stream);
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return BLISSGPU_OK;
}
void* blissgpu_ctx_get_stream(blissgpu_ctx* c) { return c ? (void*)c->stream : nullptr; }

// Stream interop for hosts that keep their own streams (e.g. torch's current stream, NULL = the legacy default
// stream): order the context's stream after / before work queued on another stream without a host synchronisation.
int blissgpu_ctx_wait_stream(blissgpu_ctx* c, void* producer_stream) {
    CTX_ENTER(c, "blissgpu_ctx_wait_stream");
    if ((hipStream_t)producer_stream == c->stream) return BLISSGPU_OK;
    HIP_TRY(hipEventRecord(c->ev_interop, (hipStream_t)producer_stream));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_interop, 0));
    return BLISSGPU_OK;
}
int blissgpu_ctx_signal_stream(blissgpu_ctx* c, void* consumer_stream) {
    CTX_ENTER(c, "blissgpu_ctx_signal_stream");
    if ((hipStream_t)consumer_stream == c->stream) return BLISSGPU_OK;
    HIP_TRY(hipEventRecord(c->ev_interop, c->stream));
    HIP_TRY(hipStreamWaitEvent((hipStream_t)consumer_stream, c->ev_interop, 0));
    return BLISSGPU_OK;
}

int blissgpu_ctx_set_option(blissgpu_ctx* c, int option, int64_t value) {
    if (!c) return fail(BLISSGPU_ERR_INVALID, "blissgpu_ctx_set_option", "ctx is NULL");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));  // nothing of the previous schedule is in flight when it changes
    switch (option) {
        case BLISSGPU_OPT_SERIAL: c->serial = value != 0; break;
        case BLISSGPU_OPT_TAIL_MODE: c->tail_mode = (int)value; break;
        case BLISSGPU_OPT_PIPELINE_CHUNKS: c->pipeline_chunks = (uint32_t)std::min<int64_t>(64, std::max<int64_t>(1, value)); break;
        case BLISSGPU_OPT_ROLLOFF_EXACT_ALL: c->rolloff_exact_all = value != 0; break;
        case BLISSGPU_OPT_DEBUG_CHROMA: c->debug_chroma = value != 0; break;
        case BLISSGPU_OPT_TAIL_SPLIT: c->tail_split = (int)value; break;
        case BLISSGPU_OPT_FLUX_ORDER: c->flux_order = value != 0; break;
        case BLISSGPU_OPT_STFT_SHAPE: c->stft_shape = (value >= 0 && value <= 3) ? (int)value : 0; break;
        case BLISSGPU_OPT_STAGE_LANES: c->feed.stage_cfg.lanes = (int)std::max<int64_t>(0, std::min<int64_t>(value, bg::MAX_STAGE_LANES)); break;
        case BLISSGPU_OPT_STAGE_SLAB_KIB: c->feed.stage_cfg.slab_bytes = (size_t)std::max<int64_t>(64, std::min<int64_t>(value, 65536)) << 10; break;
        case BLISSGPU_OPT_STAGE_NUMA: c->feed.stage_numa = value != 0; break;
        case BLISSGPU_OPT_STAGE_SLABS: c->feed.stage_cfg.slabs_per_lane = (int)std::max<int64_t>(1, std::min<int64_t>(value, 8)); break;
        case BLISSGPU_OPT_CAND_BUDGET: c->cand_budget = (uint32_t)std::max<int64_t>(0, std::min<int64_t>(value, 714)); break;
        default: return fail(BLISSGPU_ERR_INVALID, "blissgpu_ctx_set_option", "unknown option");
    }
    return BLISSGPU_OK;
}

uint64_t blissgpu_ctx_staged_bytes(blissgpu_ctx* c) {
    if (!c) return 0;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    return c->feed.staged_bytes + (c->feed.ring ? c->feed.ring->bytes_staged() : 0);
}

int blissgpu_ctx_set_workspace_limit(blissgpu_ctx* c, uint64_t bytes) {
    if (!c || bytes < (1ull << 20)) return fail(BLISSGPU_ERR_INVALID, "blissgpu_ctx_set_workspace_limit", "limit < 1 MiB");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    c->ws_limit = bytes;
    return BLISSGPU_OK;
}
uint64_t blissgpu_ctx_get_workspace_limit(blissgpu_ctx* c) { return c ? c->ws_limit : 0; }

int blissgpu_ctx_synchronize(blissgpu_ctx* c) {
    if (!c) return fail(BLISSGPU_ERR_INVALID, "blissgpu_ctx_synchronize", "ctx is NULL");
    hipStream_t st;
    { std::lock_guard<std::recursive_mutex> lk(c->mu); st = c->stream; }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(st));  // not under the lock: other threads may keep enqueueing
    return BLISSGPU_OK;
}

int blissgpu_host_alloc(void** p, uint64_t bytes) {
    if (!p) return fail(BLISSGPU_ERR_INVALID, "blissgpu_host_alloc", "NULL");
    hipError_t e = hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? BLISSGPU_ERR_OOM : BLISSGPU_ERR_NO_DEVICE, "hipHostMalloc", hipGetErrorString(e));
    return BLISSGPU_OK;
}
int blissgpu_host_free(void* p) { HIP_TRY(hipHostFree(p)); return BLISSGPU_OK; }

int blissgpu_feature_weights(uint32_t features_version, float* M) {
    const uint32_t d = blissgpu_feature_count(features_version);
    if (!d || !M) return fail(BLISSGPU_ERR_INVALID, "blissgpu_feature_weights", "bad version or NULL");
    memset(M, 0, sizeof(float) * d * d);
    for (uint32_t i = 0; i < d; i++) {
        float w = 1.0f;
        if (features_version == BLISSGPU_FEATURES_V2) {  // VERSION2_WEIGHTS, src/lib.rs:209-234
            if (i == 0) w = 0.25f;
            else if (i >= 10) w = 3.0f / 13.0f;
        }
        M[i * d + i] = w;
    }
    return BLISSGPU_OK;
}

int blissgpu_pairwise_device(blissgpu_ctx* c, const float* d_A, uint64_t n, const float* d_B, uint64_t m, uint32_t d,
                             int metric, const float* d_M, float* d_out, uint64_t ld_out) {
    if (!c || !d_A || !d_B || !d_out) return fail(BLISSGPU_ERR_INVALID, "blissgpu_pairwise_device", "NULL argument");
    if (d == 0 || d > 64 || metric < 0 || metric > 2 || ld_out < m)
        return fail(BLISSGPU_ERR_INVALID, "blissgpu_pairwise_device", "bad d / metric / ld_out");
    if (metric == BLISSGPU_METRIC_MAHALANOBIS && !d_M)
        return fail(BLISSGPU_ERR_INVALID, "blissgpu_pairwise_device", "mahalanobis needs M");
    CTX_ENTER(c, "blissgpu_pairwise_device");
    int diag = 0;
    if (metric == BLISSGPU_METRIC_MAHALANOBIS) {
        if (d_M == c->st_m.p && c->m_cache.size() == (size_t)d * d) {  // staged by a host form: the host copy is at hand
            diag = is_diag(c->m_cache.data(), d);
        } else {
            std::vector<float> hM((size_t)d * d);
            HIP_TRY(hipMemcpyAsync(hM.data(), d_M, hM.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            diag = is_diag(hM.data(), d);
        }
    }
    {
        Prof p(c, K_PAIRWISE);
        launch_pairwise(d_A, n, d_B, m, d, metric, d_M, diag, d_out, ld_out, c->stream);
    }
    HIP_TRY(hipGetLastError());
    return BLISSGPU_OK;
}

}  // extern "C"

namespace {

// Stage the d x d matrix of a host-pointer call in the context (uploaded only when it differs from the one staged last:
// a library is ordered with the same weights over and over).  Returns the device pointer through *d_M (NULL if none).
int stage_matrix(blissgpu_ctx* c, const float* M, uint32_t d, int metric, const float** d_M) {
    *d_M = nullptr;
    if (metric != BLISSGPU_METRIC_MAHALANOBIS) return BLISSGPU_OK;
    const size_t n = (size_t)d * d;
    int rc = c->st_m.ensure(4096);  // 64 x 64: never regrown, so the pointer identifies "staged by us"
    if (rc) return rc;
    if (c->m_cache.size() != n || memcmp(c->m_cache.data(), M, n * sizeof(float)) != 0) {
        c->m_cache.assign(M, M + n);
        // the source is the context's own copy: the caller's buffer may go away as soon as the call returns
        HIP_TRY(hipMemcpyAsync(c->st_m.p, c->m_cache.data(), n * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    *d_M = c->st_m.p;
    return BLISSGPU_OK;
}

}  // namespace

extern "C" {

int blissgpu_pairwise(const float* A, uint64_t n, const float* B, uint64_t m, uint32_t d, int metric, const float* M,
                      float* out) {
    if (!A || !B || !out) return fail(BLISSGPU_ERR_INVALID, "blissgpu_pairwise", "NULL argument");
    if (d == 0 || d > 64 || metric < 0 || metric > 2) return fail(BLISSGPU_ERR_INVALID, "blissgpu_pairwise", "bad d / metric");
    if (metric == BLISSGPU_METRIC_MAHALANOBIS && !M) return fail(BLISSGPU_ERR_INVALID, "blissgpu_pairwise", "mahalanobis needs M");
    if (n == 0 || m == 0) return BLISSGPU_OK;
    blissgpu_ctx* c;
    int rc = default_ctx(&c);
    if (rc) return rc;
    CTX_ENTER(c, "blissgpu_pairwise");
    // rows of the output are produced in slabs of <= 4 GiB so host-sized problems never need n*m device memory
    const uint64_t slab_rows = std::max<uint64_t>(1, std::min<uint64_t>(n, (1ull << 30) / std::max<uint64_t>(m, 1)));
    const bool self = (A == B && n == m);  // self-distance matrix: one device copy, symmetric kernel
    const float* dM = nullptr;
    if ((rc = c->st_a.ensure(n * d))) return rc;
    if (!self && (rc = c->st_b.ensure(m * d))) return rc;
    if ((rc = c->st_out.ensure(slab_rows * m * sizeof(float)))) return rc;
    if ((rc = stage_matrix(c, M, d, metric, &dM))) return rc;
    float *dA = c->st_a.p, *dB = self ? c->st_a.p : c->st_b.p, *dO = reinterpret_cast<float*>(c->st_out.p);
    hipError_t e = hipMemcpyAsync(dA, A, n * d * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && !self) e = hipMemcpyAsync(dB, B, m * d * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) rc = fail(BLISSGPU_ERR_HIP, "hipMemcpyAsync", hipGetErrorString(e));
    for (uint64_t r0 = 0; !rc && r0 < n; r0 += slab_rows) {
        const uint64_t rows = std::min(slab_rows, n - r0);
        // a row slab of a self-distance matrix is not square: only the full matrix takes the symmetric kernel
        rc = blissgpu_pairwise_device(c, dA + r0 * d, rows, dB, m, d, metric, dM, dO, m);
        if (!rc) {
            e = hipMemcpyAsync(out + r0 * m, dO, rows * m * sizeof(float), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) rc = fail(BLISSGPU_ERR_HIP, "copy back", hipGetErrorString(e));
        }
    }
    (void)hipStreamSynchronize(c->stream);
    return rc;
}

// One pair: both vectors travel in the kernel arguments and the result lands in a page-locked word, so the call is
// one launch + one stream synchronisation -- no allocation, no staging copy (Song::distance, src/song/mod.rs:519-521).
int blissgpu_distance(const float* a, const float* b, uint32_t d, int metric, const float* M, float* out) {
    if (!a || !b || !out) return fail(BLISSGPU_ERR_INVALID, "blissgpu_distance", "NULL argument");
    if (d == 0 || d > 64 || metric < 0 || metric > 2) return fail(BLISSGPU_ERR_INVALID, "blissgpu_distance", "bad d / metric");
    if (metric == BLISSGPU_METRIC_MAHALANOBIS && !M) return fail(BLISSGPU_ERR_INVALID, "blissgpu_distance", "mahalanobis needs M");
    blissgpu_ctx* c;
    int rc = default_ctx(&c);
    if (rc) return rc;
    CTX_ENTER(c, "blissgpu_distance");
    const float* dM = nullptr;
    if ((rc = stage_matrix(c, M, d, metric, &dM))) return rc;
    launch_pair_distance(a, b, d, metric, dM, c->h_scalar, c->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    *out = *c->h_scalar;
    return BLISSGPU_OK;
}

// ---- playlist ordering (src/playlist.rs:24-59, 256-326) ----
static int playlist_args_ok(const char* who, const void* a, const void* b, const void* o, uint32_t n_seeds, uint64_t n,
                            uint32_t d, int metric, const float* M) {
    // An EMPTY seed set is legal: FunctionDistanceMetric::distance sums over no vectors, i.e. 0.0 for every candidate
    // (src/playlist.rs:52-58) -- closest_to_songs then keeps the candidates' order (stable sort) and song_to_song
    // starts from the first candidate.
    if ((n_seeds && !a) || (n && !b) || !o) return fail(BLISSGPU_ERR_INVALID, who, "NULL argument");
    if (d == 0 || d > 64 || metric < 0 || metric > 2) return fail(BLISSGPU_ERR_INVALID, who, "bad d / metric");
    if (metric == BLISSGPU_METRIC_MAHALANOBIS && !M) return fail(BLISSGPU_ERR_INVALID, who, "mahalanobis needs M");
    if (n > 0xFFFFFFFFull) return fail(BLISSGPU_ERR_INVALID, who, "more than 2^32 - 1 candidates");
    return BLISSGPU_OK;
}

// reads the device NaN flag (synchronises the stream)
static int nan_check(blissgpu_ctx* c, const uint32_t* d_flag, const char* who) {
    uint32_t flag = 0;
    HIP_TRY(hipMemcpyAsync(&flag, d_flag, sizeof(flag), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (flag) return fail(BLISSGPU_ERR_NAN, who, "NaN distance (the reference panics here)");
    return BLISSGPU_OK;
}

int blissgpu_set_distance_device(blissgpu_ctx* c, const float* d_seeds, uint32_t n_seeds, const float* d_cand, uint64_t n,
                                 uint32_t d, int metric, const float* d_M, float* d_out) {
    if (!c) return fail(BLISSGPU_ERR_INVALID, "blissgpu_set_distance_device", "ctx is NULL");
    int rc = playlist_args_ok("blissgpu_set_distance_device", d_seeds, d_cand, d_out, n_seeds, n, d, metric, d_M);
    if (rc) return rc;
    if (n == 0) return BLISSGPU_OK;
    CTX_ENTER(c, "blissgpu_set_distance_device");
    rc = c->pl_sync.ensure(4);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(c->pl_sync.p, 0, 4 * sizeof(uint32_t), c->stream));
    {
        Prof p(c, K_SET_DISTANCE);
        launch_set_distance(d_seeds, n_seeds, d_cand, n, d, metric, d_M, d_out, nullptr, nullptr, c->pl_sync.p + 1, c->stream);
    }
    HIP_TRY(hipGetLastError());
    return BLISSGPU_OK;  // NaN distances are returned as data here, like DistanceMetric::distance
}

int blissgpu_closest_to_songs_device(blissgpu_ctx* c, const float* d_seeds, uint32_t n_seeds, const float* d_cand,
                                     uint64_t n, uint32_t d, int metric, const float* d_M, uint32_t* d_order, float* d_dist) {
    if (!c) return fail(BLISSGPU_ERR_INVALID, "blissgpu_closest_to_songs_device", "ctx is NULL");
    int rc = playlist_args_ok("blissgpu_closest_to_songs_device", d_seeds, d_cand, d_order, n_seeds, n, d, metric, d_M);
    if (rc) return rc;
    if (n == 0) return BLISSGPU_OK;
    CTX_ENTER(c, "blissgpu_closest_to_songs_device");
    const uint32_t n32 = (uint32_t)n;
    size_t tmp_bytes = 0;
    HIP_TRY(sort_pairs_u32(nullptr, &tmp_bytes, nullptr, nullptr, nullptr, nullptr, n32, c->stream));
    rc = c->pl_sync.ensure(4);
    if (!rc) rc = c->pl_keys.ensure((size_t)3 * n32);  // keys in | keys out | indices in
    if (!rc) rc = c->pl_tmp.ensure(tmp_bytes);
    if (rc) return rc;
    uint32_t *keys_in = c->pl_keys.p, *keys_out = keys_in + n32, *idx_in = keys_out + n32;
    HIP_TRY(hipMemsetAsync(c->pl_sync.p, 0, 4 * sizeof(uint32_t), c->stream));
    {
        Prof p(c, K_SET_DISTANCE);
        launch_set_distance(d_seeds, n_seeds, d_cand, n, d, metric, d_M, d_dist, keys_in, idx_in, c->pl_sync.p + 1, c->stream);
    }
    HIP_TRY(hipGetLastError());
    // the single-launch sort spins on grid barriers: only while its workgroups are certainly co-resident (one per CU, and at
    // most two contexts' worth of persistent kernels on the device); otherwise one launch per step
    HIP_TRY(sort_pairs_u32(c->pl_tmp.p, &tmp_bytes, keys_in, keys_out, idx_in, d_order, n32, c->stream, c->pl_sync.p + 2,
                           live_contexts(c->device) <= 2 ? (uint32_t)std::max(1, c->n_cus) : 0u));
    return nan_check(c, c->pl_sync.p + 1, "blissgpu_closest_to_songs_device");
}

int blissgpu_song_to_song_device(blissgpu_ctx* c, const float* d_seeds, uint32_t n_seeds, const float* d_cand, uint64_t n,
                                 uint32_t d, int metric, const float* d_M, uint32_t* d_order) {
    if (!c) return fail(BLISSGPU_ERR_INVALID, "blissgpu_song_to_song_device", "ctx is NULL");
    int rc = playlist_args_ok("blissgpu_song_to_song_device", d_seeds, d_cand, d_order, n_seeds, n, d, metric, d_M);
    if (rc) return rc;
    if (n == 0) return BLISSGPU_OK;
    CTX_ENTER(c, "blissgpu_song_to_song_device");
    // one workgroup per 256 candidates up to one per CU (all workgroups must be co-resident: the kernel spins on
    // a grid barrier); each thread then owns ceil(n / (256 G)) <= 64 candidates
    // four candidates per thread (register-resident) is the sweet spot: fewer workgroups make the grid barrier and the
    // slot reduction cheaper (100 k songs: 98 workgroups, 7.3 us per step; 256 workgroups with two each: 11.7 us)
    uint32_t grid = (uint32_t)std::min<uint64_t>((n + 1023) / 1024, (uint64_t)std::min(256, std::max(1, c->n_cus)));
    if ((n + (uint64_t)grid * 256 - 1) / ((uint64_t)grid * 256) > 64)
        return fail(BLISSGPU_ERR_INVALID, "blissgpu_song_to_song_device", "pool too large for one launch (> 64 candidates per thread)");
    rc = c->pl_sync.ensure(4);
    if (!rc) rc = c->pl_slots.ensure((size_t)2 * grid);
    if (rc) return rc;
    // more than two contexts alive on this device: spinning grids of different contexts must not overlap (see live_contexts)
    std::unique_lock<std::mutex> one_at_a_time;
    if (live_contexts(c->device) > 2) one_at_a_time = std::unique_lock<std::mutex>(persistent_kernel_mutex(c->device));
    HIP_TRY(hipMemsetAsync(c->pl_sync.p, 0, 4 * sizeof(uint32_t), c->stream));
    {
        Prof p(c, K_SONG_TO_SONG);
        launch_song_to_song(d_seeds, n_seeds, d_cand, (uint32_t)n, d, metric, d_M, d_order, c->pl_slots.p, c->pl_sync.p, grid,
                            c->stream);
    }
    HIP_TRY(hipGetLastError());
    if (one_at_a_time.owns_lock()) HIP_TRY(hipStreamSynchronize(c->stream));  // the chain has left the CUs before the next one starts
    return nan_check(c, c->pl_sync.p + 1, "blissgpu_song_to_song_device");
}

}  // extern "C"

// host-pointer wrappers: stage seeds / candidates / M in the context's buffers, run the device form, copy the result back
namespace {
struct PlStage {
    blissgpu_ctx* c;
    float *seeds = nullptr, *cand = nullptr, *dist = nullptr;
    const float* M = nullptr;
    void* out = nullptr;
    int up(const float* h_seeds, uint32_t n_seeds, const float* h_cand, uint64_t n, uint32_t d, int metric, const float* h_M,
           size_t out_bytes, bool want_dist) {
        int rc = c->st_a.ensure(std::max<size_t>(1, (size_t)n_seeds * d));
        if (!rc) rc = c->st_b.ensure(std::max<size_t>(1, n * d));
        if (!rc) rc = c->st_out.ensure(std::max<size_t>(4, out_bytes));
        if (!rc && want_dist) rc = c->st_dist.ensure(std::max<size_t>(1, n));
        if (!rc) rc = stage_matrix(c, h_M, d, metric, &M);
        if (rc) return rc;
        seeds = c->st_a.p; cand = c->st_b.p; out = c->st_out.p; dist = want_dist ? c->st_dist.p : nullptr;
        hipError_t e = hipSuccess;
        if (n_seeds) e = hipMemcpyAsync(seeds, h_seeds, (size_t)n_seeds * d * sizeof(float), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess && n) e = hipMemcpyAsync(cand, h_cand, n * d * sizeof(float), hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) return fail(BLISSGPU_ERR_HIP, "hipMemcpyAsync(playlist)", hipGetErrorString(e));
        return BLISSGPU_OK;
    }
    int down(void* h_out, size_t out_bytes, float* h_dist, uint64_t n) {
        hipError_t e = hipSuccess;
        if (out_bytes) e = hipMemcpyAsync(h_out, out, out_bytes, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess && h_dist && n) e = hipMemcpyAsync(h_dist, dist, n * sizeof(float), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return fail(BLISSGPU_ERR_HIP, "copy back(playlist)", hipGetErrorString(e));
        return BLISSGPU_OK;
    }
};
}  // namespace

extern "C" {

int blissgpu_set_distance(const float* seeds, uint32_t n_seeds, const float* cand, uint64_t n, uint32_t d, int metric,
                          const float* M, float* out) {
    int rc = playlist_args_ok("blissgpu_set_distance", seeds, cand, out, n_seeds, n, d, metric, M);
    if (rc || n == 0) return rc;
    blissgpu_ctx* c;
    rc = default_ctx(&c);
    if (rc) return rc;
    CTX_ENTER(c, "blissgpu_set_distance");
    PlStage s{c};
    rc = s.up(seeds, n_seeds, cand, n, d, metric, M, n * sizeof(float), false);
    if (!rc) rc = blissgpu_set_distance_device(c, s.seeds, n_seeds, s.cand, n, d, metric, s.M, (float*)s.out);
    if (!rc) rc = s.down(out, n * sizeof(float), nullptr, 0);
    (void)hipStreamSynchronize(c->stream);
    return rc;
}

int blissgpu_closest_to_songs(const float* seeds, uint32_t n_seeds, const float* cand, uint64_t n, uint32_t d, int metric,
                              const float* M, uint32_t* order, float* dist) {
    int rc = playlist_args_ok("blissgpu_closest_to_songs", seeds, cand, order, n_seeds, n, d, metric, M);
    if (rc || n == 0) return rc;
    blissgpu_ctx* c;
    rc = default_ctx(&c);
    if (rc) return rc;
    CTX_ENTER(c, "blissgpu_closest_to_songs");
    PlStage s{c};
    rc = s.up(seeds, n_seeds, cand, n, d, metric, M, n * sizeof(uint32_t), dist != nullptr);
    if (!rc) rc = blissgpu_closest_to_songs_device(c, s.seeds, n_seeds, s.cand, n, d, metric, s.M, (uint32_t*)s.out, s.dist);
    if (!rc) rc = s.down(order, n * sizeof(uint32_t), dist, n);
    (void)hipStreamSynchronize(c->stream);
    return rc;
}

int blissgpu_song_to_song(const float* seeds, uint32_t n_seeds, const float* cand, uint64_t n, uint32_t d, int metric,
                          const float* M, uint32_t* order) {
    int rc = playlist_args_ok("blissgpu_song_to_song", seeds, cand, order, n_seeds, n, d, metric, M);
    if (rc || n == 0) return rc;
    blissgpu_ctx* c;
    rc = default_ctx(&c);
    if (rc) return rc;
    CTX_ENTER(c, "blissgpu_song_to_song");
    PlStage s{c};
    rc = s.up(seeds, n_seeds, cand, n, d, metric, M, n * sizeof(uint32_t), false);
    if (!rc) rc = blissgpu_song_to_song_device(c, s.seeds, n_seeds, s.cand, n, d, metric, s.M, (uint32_t*)s.out);
    if (!rc) rc = s.down(order, n * sizeof(uint32_t), nullptr, 0);
    (void)hipStreamSynchronize(c->stream);
    return rc;
}

int blissgpu_malloc(void** p, uint64_t bytes) {
    if (!p) return fail(BLISSGPU_ERR_INVALID, "blissgpu_malloc", "NULL");
    hipError_t e = hipMalloc(p, bytes ? bytes : 1);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? BLISSGPU_ERR_OOM : BLISSGPU_ERR_NO_DEVICE, "hipMalloc", hipGetErrorString(e));
    return BLISSGPU_OK;
}
int blissgpu_free(void* p) { HIP_TRY(hipFree(p)); return BLISSGPU_OK; }
int blissgpu_memcpy_h2d(blissgpu_ctx* c, void* dst, const void* src, uint64_t bytes) {
    CTX_ENTER(c, "blissgpu_memcpy_h2d");
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BLISSGPU_OK;
}
int blissgpu_memcpy_d2h(blissgpu_ctx* c, void* dst, const void* src, uint64_t bytes) {
    CTX_ENTER(c, "blissgpu_memcpy_d2h");
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BLISSGPU_OK;
}

static int synth_impl(blissgpu_ctx* c, float* d_pcm, const uint64_t* offsets, const uint64_t* lengths, uint32_t n_songs,
                      uint32_t first_song_index, const uint32_t* song_index) {
    if (!c || !d_pcm || !offsets || !lengths) return fail(BLISSGPU_ERR_INVALID, "blissgpu_synth_white_noise", "NULL argument");
    if (n_songs == 0) return BLISSGPU_OK;
    CTX_ENTER(c, "blissgpu_synth_white_noise");
    std::vector<SongDesc> songs(n_songs);
    std::vector<uint32_t> pfx(n_songs + 1, 0);
    for (uint32_t i = 0; i < n_songs; i++) {
        songs[i] = SongDesc{};
        songs[i].pcm_off = offsets[i];
        songs[i].n = lengths[i];
        songs[i].n_e = (uint32_t)((lengths[i] + 255) / 256);
        pfx[i + 1] = pfx[i] + (uint32_t)((lengths[i] + 4095) / 4096);
    }
    SongDesc* d_songs = nullptr;
    uint32_t *d_pfx = nullptr, *d_idx = nullptr;
    HIP_TRY(hipMalloc((void**)&d_songs, n_songs * sizeof(SongDesc)));
    hipError_t e = hipMalloc((void**)&d_pfx, (n_songs + 1) * sizeof(uint32_t));
    if (e == hipSuccess && song_index) e = hipMalloc((void**)&d_idx, n_songs * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpy(d_songs, songs.data(), n_songs * sizeof(SongDesc), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_pfx, pfx.data(), (n_songs + 1) * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess && song_index) e = hipMemcpy(d_idx, song_index, n_songs * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        Prof p(c, K_SYNTH);
        launch_synth(d_pcm, d_songs, n_songs, d_pfx, pfx[n_songs], first_song_index, d_idx, c->stream);
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_songs);
    (void)hipFree(d_pfx);
    (void)hipFree(d_idx);
    if (e != hipSuccess) return fail(BLISSGPU_ERR_HIP, "synth", hipGetErrorString(e));
    return BLISSGPU_OK;
}

int blissgpu_synth_white_noise_device(blissgpu_ctx* c, float* d_pcm, const uint64_t* offsets, const uint64_t* lengths,
                                      uint32_t n_songs, uint32_t first_song_index) {
    return synth_impl(c, d_pcm, offsets, lengths, n_songs, first_song_index, nullptr);
}

int blissgpu_synth_white_noise_indexed_device(blissgpu_ctx* c, float* d_pcm, const uint64_t* offsets, const uint64_t* lengths,
                                              const uint32_t* song_index, uint32_t n_songs) {
    if (!song_index) return fail(BLISSGPU_ERR_INVALID, "blissgpu_synth_white_noise_indexed_device", "NULL argument");
    return synth_impl(c, d_pcm, offsets, lengths, n_songs, 0, song_index);
}

int blissgpu_profile_enable(blissgpu_ctx* c, int enable) {
    CTX_ENTER(c, "blissgpu_profile_enable");
    c->profiling = enable != 0;
    return BLISSGPU_OK;
}

int blissgpu_profile_reset(blissgpu_ctx* c) {
    CTX_ENTER(c, "blissgpu_profile_reset");
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(c->aux_stream);
    (void)hipStreamSynchronize(c->chr_stream);
    if (c->mask_stream) (void)hipStreamSynchronize(c->mask_stream);
    for (auto& v : c->events) {
        for (auto& ev : v) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
        v.clear();
    }
    return BLISSGPU_OK;
}

int blissgpu_profile_kernel_count(void) { return K_COUNT; }
const char* blissgpu_profile_kernel_name(int k) { return (k >= 0 && k < K_COUNT) ? kKernelNames[k] : ""; }

int blissgpu_profile_get(blissgpu_ctx* c, int k, double* total_ms, uint64_t* launches) {
    if (!c || k < 0 || k >= K_COUNT) return fail(BLISSGPU_ERR_INVALID, "blissgpu_profile_get", "bad kernel id");
    CTX_ENTER(c, "blissgpu_profile_get");
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipStreamSynchronize(c->aux_stream));
    HIP_TRY(hipStreamSynchronize(c->chr_stream));
    if (c->mask_stream) HIP_TRY(hipStreamSynchronize(c->mask_stream));
    double tot = 0.0;
    for (auto& ev : c->events[k]) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, ev.a, ev.b));
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = c->events[k].size();
    return BLISSGPU_OK;
}

uint64_t blissgpu_debug_last_chunks(blissgpu_ctx* c) { return c ? c->last_chunks : 0; }

int blissgpu_debug_last_tuning(blissgpu_ctx* c, double* tuning, uint32_t* n_bpms, uint32_t n_songs) {
    if (!c || n_songs > c->dbg_n) return fail(BLISSGPU_ERR_INVALID, "blissgpu_debug_last_tuning", "no such batch");
    CTX_ENTER(c, "blissgpu_debug_last_tuning");
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<int32_t> idx(n_songs);
    HIP_TRY(hipMemcpy(idx.data(), c->dbg_tuning.p, n_songs * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (tuning)
        for (uint32_t i = 0; i < n_songs; i++)
            tuning[i] = idx[i] < 0 ? 0.0 : (-50.0 + (100.0 * 0.01 * (double)idx[i])) / 100.0;
    if (n_bpms) HIP_TRY(hipMemcpy(n_bpms, c->dbg_nbpms.p, n_songs * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return BLISSGPU_OK;
}

int blissgpu_debug_fetch(blissgpu_ctx* c, int what, uint32_t song, void* dst, uint64_t max_elems, uint64_t* n_elems) {
    if (!c || !dst) return fail(BLISSGPU_ERR_INVALID, "blissgpu_debug_fetch", "NULL argument");
    CTX_ENTER(c, "blissgpu_debug_fetch");
    if (what == BLISSGPU_DEBUG_FILTER_BANK) {  // a table of the context, not of a batch: `song` is the tuning slot
        if (song > (uint32_t)N_TUNING) return fail(BLISSGPU_ERR_INVALID, "blissgpu_debug_fetch", "tuning slot must be 0..100");
        const uint64_t n = (uint64_t)BANK_ROWS * BANK_PITCH;
        if (n_elems) *n_elems = n;
        const uint64_t k = std::min(n, max_elems);
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (k) HIP_TRY(hipMemcpy(dst, c->chroma_bank + (size_t)song * n, k * sizeof(double), hipMemcpyDeviceToHost));
        return BLISSGPU_OK;
    }
    // `song` is the caller's index into the last batch; the chunk keeps its songs in length order
    size_t pos = c->last_songs.size();
    for (size_t i = 0; i < c->last_songs.size(); i++)
        if (c->last_songs[i].row == song) { pos = i; break; }
    if (pos == c->last_songs.size()) return fail(BLISSGPU_ERR_INVALID, "blissgpu_debug_fetch", "no such song in the last chunk");
    HIP_TRY(hipStreamSynchronize(c->stream));
    const SongDesc& d = c->last_songs[pos];
    const Workspace& w = c->last_ws;
    const void* src = nullptr;
    uint64_t n = 0, esz = 4;
    const uint64_t runs = d.ok ? (d.n_b >= (uint32_t)BT_STEP ? (d.n_b - BT_STEP) / BT_STEP + 1 : 0) : 0;
    switch (what) {
        case BLISSGPU_DEBUG_CENTROID: src = w.centroid + d.t_off; n = d.n_t; break;
        case BLISSGPU_DEBUG_ROLLOFF: src = w.rolloff + d.t_off; n = d.n_t; break;
        case BLISSGPU_DEBUG_FLATNESS: src = w.flatness + d.t_off; n = d.n_t; break;
        case BLISSGPU_DEBUG_FLUX: src = w.flux + d.b_off; n = d.n_b; break;
        case BLISSGPU_DEBUG_THRESHOLDED: src = w.thresholded + d.b_off; n = d.n_b; break;
        case BLISSGPU_DEBUG_RUN_BPM: src = w.run_bpm + pos * w.runs_pitch; n = runs; break;
        case BLISSGPU_DEBUG_RUN_COUNT: src = w.run_cnt + pos * w.runs_pitch; n = runs; break;
        case BLISSGPU_DEBUG_SPECTROGRAM: src = w.spec + d.c_off * (size_t)CBINS_PAD; n = (uint64_t)d.n_c * CBINS_PAD; break;
        case BLISSGPU_DEBUG_ENERGY256: src = w.e256 + d.e_off; n = d.n_e; break;
        case BLISSGPU_DEBUG_CROSSINGS256: src = w.zc256 + d.e_off; n = d.n_e; break;
        case BLISSGPU_DEBUG_PITCH_HIST: src = w.hist100 + pos * N_TUNING; n = N_TUNING; break;
        case BLISSGPU_DEBUG_CHROMA:
        case BLISSGPU_DEBUG_INTERVAL:
            if (!w.dbg_chroma || !w.dbg_interval)
                return fail(BLISSGPU_ERR_INVALID, "blissgpu_debug_fetch", "set BLISSGPU_OPT_DEBUG_CHROMA before the analysis");
            esz = 8;
            if (what == BLISSGPU_DEBUG_CHROMA) { src = w.dbg_chroma + d.c_off * 12; n = (uint64_t)d.n_c * 12; }
            else { src = w.dbg_interval + pos * 10; n = 10; }
            break;
        default: return fail(BLISSGPU_ERR_INVALID, "blissgpu_debug_fetch", "unknown tap");
    }
    if (!d.ok) n = 0;
    if (n_elems) *n_elems = n;
    const uint64_t k = std::min(n, max_elems);
    if (k) HIP_TRY(hipMemcpy(dst, src, k * esz, hipMemcpyDeviceToHost));
    return BLISSGPU_OK;
}

}  // extern "C"
