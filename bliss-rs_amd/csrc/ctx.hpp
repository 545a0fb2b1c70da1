// ctx.hpp -- host-side state shared by the translation units behind the C ABI (not part of the ABI):
//   blissgpu.hip   context life cycle, tables, distances, playlist ordering, helpers
//   scheduler.hip  Song::analyze batches: planning, length-bucketed chunks, the two-slot streaming schedule,
//                  the host PCM feed and the coalescing front of the single-song entry points
//   node.hip       one process driving every GPU of the node (RCCL all-gather of the feature rows)
#pragma once
#include <hip/hip_runtime.h>
#include <sched.h>

#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/blissgpu.h"
#include "internal.hpp"
#include "resample.hpp"
#include "staging_ring.hpp"

namespace bg {

int fail(int code, const char* what, const char* detail);  // sets the thread-local last error, returns code

#define HIP_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return ::bg::fail(BLISSGPU_ERR_HIP, #expr, hipGetErrorString(e_)); \
    } while (0)

struct EventPair { hipEvent_t a, b; };

template <typename T>
struct DevBuf {  // grow-only device buffer
    T* p = nullptr;
    size_t cap = 0;  // elements
    int ensure(size_t n) {
        if (n <= cap) return BLISSGPU_OK;
        if (p) (void)hipFree(p);  // hipFree waits for the device: nothing queued can still be using the old block
        p = nullptr;
        cap = 0;
        size_t want = n + n / 8 + 64;
        hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
        if (e != hipSuccess && want > n) {  // no room for the growth margin: take exactly what is needed
            (void)hipGetLastError();
            want = n;
            e = hipMalloc((void**)&p, want * sizeof(T));
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            p = nullptr;
            return fail(BLISSGPU_ERR_OOM, "hipMalloc", hipGetErrorString(e));
        }
        cap = want;
        return BLISSGPU_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

template <typename T>
struct PinnedBuf {  // grow-only page-locked host buffer
    T* p = nullptr;
    size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return BLISSGPU_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = n + n / 4 + 64;
        hipError_t e = hipHostMalloc((void**)&p, want * sizeof(T), hipHostMallocDefault);
        if (e != hipSuccess) { (void)hipGetLastError(); p = nullptr; return fail(BLISSGPU_ERR_OOM, "hipHostMalloc", hipGetErrorString(e)); }
        cap = want;
        return BLISSGPU_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// One of the two chunk slots of the streaming schedule: its own workspace slab, descriptor buffers and events, so
// that chunk k + 1's FFT kernels can start while chunk k's per-song tails (beat tracker, summaries, row assembly)
// are still running on the aux stream.
struct ChunkSlot {
    DevBuf<uint8_t> slab;        // workspace carved per chunk
    DevBuf<uint8_t> desc;        // SongDesc[] + tile prefix arrays
    PinnedBuf<uint8_t> h_desc;   // pinned staging of desc
    hipEvent_t ev_start = nullptr, ev_fork = nullptr, ev_stft = nullptr, ev_sel = nullptr, ev_tune = nullptr, ev_sum = nullptr, ev_chroma = nullptr;
    hipEvent_t ev_acf = nullptr, ev_beat = nullptr;  // masked tail (tail_mode >= 2): autocorrelations done / state machines done
    bool beat_masked = false;              // the chunk in flight ran its beat state machines on the CU-masked stream
    static constexpr int MAX_PIECES = 8;
    hipEvent_t ev_piece[MAX_PIECES] = {};  // the tuning estimate of piece k of the songs is in (split tail)
    int pieces = 0;                        // > 0: the tail of the chunk in flight runs in these pieces
    bg::SongRange piece[MAX_PIECES]{};
    hipEvent_t ev_desc = nullptr;  // recorded on the main stream after the descriptor copy: the staging area is free
    hipEvent_t ev_free = nullptr;  // recorded on the aux stream after the row assembly: the slot is free
    bool used = false;
    bool back_pending = false;   // front half enqueued, chroma contraction + row assembly still to come
    Batch batch{};               // launch parameters of the chunk in flight (back half)
    Workspace ws{};
};

// Persistent staging of the host-pointer entry points (the PCM feed): N_FEED_BUFFERS device PCM buffers (+ raw s16 / multi-channel
// staging) and result buffers, two copy streams (songs alternate between them: one stream moves 54.5 GB/s from pinned
// memory, two 57.3 -- tests/tools/probes/h2d_probe.hip) and the events that order them.  Grow-only; guarded by the
// context mutex.
#ifndef FEED_COPY_STREAMS
#define FEED_COPY_STREAMS 2
#endif
// the staging ring's default shape (measured: tests/tools/stage_sweep.py, profiles/r06_stage_sweep_first.txt)
#ifndef STAGE_LANES_DEFAULT
#define STAGE_LANES_DEFAULT 4
#endif
#ifndef STAGE_SLABS_DEFAULT
#define STAGE_SLABS_DEFAULT 3
#endif
#ifndef STAGE_SLAB_KIB_DEFAULT
#define STAGE_SLAB_KIB_DEFAULT 4096
#endif
#ifndef STAGE_EVENT_FLAGS
#define STAGE_EVENT_FLAGS (hipEventDisableTiming | hipEventBlockingSync)
#endif
constexpr int N_COPY_STREAMS = FEED_COPY_STREAMS;
// Device staging buffers the groups of a call rotate through.  Three: group g + 2 is on the link while group g + 1 waits for
// its turn and group g is analysed: slack for a group whose analysis takes as long as the next group's transfer (mono s16:
// 33 songs cross the link in 5 ms and take 4 - 5 ms to analyse).  Against two buffers the difference is inside the spread of
// the s16 line (profiles/r06_feed_buffers_ab.txt); the third costs device memory only.
#ifndef FEED_BUFFERS
#define FEED_BUFFERS 3
#endif
constexpr int N_FEED_BUFFERS = FEED_BUFFERS;
// one song of the host PCM feed as the decoder delivers it
constexpr uint32_t MAX_SAMPLE_RATE = 768000;  // the resample kernel's 64-bit stream positions hold for any song below this
struct FeedSong {
    const void* p;      // host memory
    uint64_t frames;    // of `channels` interleaved samples
    uint32_t rate;      // Hz; anything but 22 050 goes through the device resampler
    uint8_t fmt;        // BLISSGPU_SAMPLE_*
    uint8_t channels;   // 1..8
    size_t frame_bytes() const { return (size_t)(fmt == BLISSGPU_SAMPLE_S16 ? 2 : 4) * channels; }
    bool direct() const { return fmt == BLISSGPU_SAMPLE_F32 && channels == 1 && rate == SWR_OUT_RATE; }  // copied verbatim
};
// the device resampler's filter bank for one input rate
struct ResampleBank {
    SwrPlan plan;
    float* d_bank = nullptr;
    size_t bytes = 0;
    uint64_t last_use = 0;  // the context's swr_clock when it was last handed out (the cache evicts the least recently used)
};
// The banks a context keeps on the device: a few common rates cost kilobytes (44.1 kHz: 66 taps, 48 kHz: 147 x 72), an inexact
// rate 1024 phases of up to ~1100 taps = 4.5 MB -- a long-running host fed odd rates must not grow without bound.
#ifndef SWR_CACHE_BYTES
#define SWR_CACHE_BYTES (64ull << 20)
#endif

// The device side of the pinned staging ring (staging_ring.hpp): page-locked slabs, one HIP stream per lane.
constexpr int MAX_STAGE_LANES = 16;
struct HipStageDev {
    int device;
    hipStream_t* streams;  // [MAX_STAGE_LANES], owned by the HostFeed
    std::vector<int> cpus; // the CPUs next to the device (its PCIe root's NUMA node), or empty = wherever the scheduler likes
    // A worker settles on the device's side of the machine before it allocates its slabs: on a two-socket host the slab the
    // DMA engine reads, and the thread that fills it, then sit on the socket the GPU hangs off (page-locked memory is placed
    // where the allocating thread runs) -- only the read of the caller's buffer may cross the socket link.
    void thread_begin(int) {
        (void)hipSetDevice(device);
        if (!cpus.empty()) {
            cpu_set_t set;
            CPU_ZERO(&set);
            for (int cpu : cpus)
                if (cpu >= 0 && cpu < CPU_SETSIZE) CPU_SET(cpu, &set);
            (void)sched_setaffinity(0, sizeof(set), &set);  // (refused in a restricted cpuset: the worker stays where it is)
        }
    }
    void* slab_alloc(size_t bytes) {
        void* p = nullptr;
        if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return p;
    }
    void slab_free(void* p) { (void)hipHostFree(p); }
    void* event_create() {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, STAGE_EVENT_FLAGS) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return (void*)e;
    }
    void event_destroy(void* e) { (void)hipEventDestroy((hipEvent_t)e); }
    int copy_async(void* dst, const void* slab, size_t bytes, int lane) {
        return (int)hipMemcpyAsync(dst, slab, bytes, hipMemcpyHostToDevice, streams[lane]);
    }
    int event_record(void* ev, int lane) { return (int)hipEventRecord((hipEvent_t)ev, streams[lane]); }
    int event_wait(void* ev) { return (int)hipEventSynchronize((hipEvent_t)ev); }
    std::string error_string(int code) { return hipGetErrorString((hipError_t)code); }
};

struct HostFeed {
    DevBuf<float> pcm[N_FEED_BUFFERS];
    DevBuf<uint8_t> raw[N_FEED_BUFFERS];
    DevBuf<float> out[N_FEED_BUFFERS];
    PinnedBuf<float> h_rows;  // (FEED_ROWS_STAGED builds only: the call's rows through a page-locked buffer, see scheduler.hip)
    hipEvent_t ev_copied[N_FEED_BUFFERS][N_COPY_STREAMS] = {}, ev_done[N_FEED_BUFFERS] = {};
    hipStream_t copy_stream[N_COPY_STREAMS] = {};
    // pageable sources (what a Rust Vec<f32> or a decoder's frame buffer is): staged by the library through page-locked slabs
    // by worker threads that run ahead of the link -- see staging_ring.hpp.  Started on the first call that brings
    // STAGE_MIN_BYTES of pageable PCM; BLISSGPU_OPT_STAGE_* change the shape (0 lanes = leave the staging to the HIP runtime).
    StageConfig stage_cfg{STAGE_LANES_DEFAULT, STAGE_SLABS_DEFAULT, (size_t)STAGE_SLAB_KIB_DEFAULT << 10};
    std::unique_ptr<StagingRing<HipStageDev>> ring;
    hipStream_t lane_stream[MAX_STAGE_LANES] = {};
    hipEvent_t ev_lane[N_FEED_BUFFERS][MAX_STAGE_LANES] = {};
    bool stage_numa = false;    // BLISSGPU_OPT_STAGE_NUMA: the workers (and their slabs) live on the device's NUMA node
    bool stage_numa_known = false;  // the placement the running ring was started with
    uint64_t staged_calls = 0;  // host-feed calls that went through the ring (statistics)
    uint64_t staged_bytes = 0;  // bytes staged by rings this context has had before the current one (a restart resets the ring's count)
};

}  // namespace bg

struct blissgpu_ctx {
    int device = 0;
    // Every entry point that takes this context holds `mu` while it touches host-side state (staging areas, buffer
    // growth, events); the host-pointer forms hold it for the whole call (they synchronise before returning anyway).
    std::recursive_mutex mu;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;      // FFT + chroma chain (the caller-visible stream)
    hipStream_t aux_stream = nullptr;  // per-song tails: PCM statistics, summaries, beat tracker, row assembly
    hipStream_t chr_stream = nullptr;  // tuning estimate of a chunk, beside the next chunk's FFT kernels
    hipStream_t mask_stream = nullptr; // tail_mode >= 2: a stream confined to `mask_cus` CUs for the beat state machines
    int mask_cus = 0;
    int tail_mode = -1;                // beat tracker: -1 = beside the FFT-8192 kernel unless the batch is one chunk, 0 / 1 force
                                       // beside / behind it; N >= 2: autocorrelations beside it, the state machines beside it on a
                                       // stream confined to N CUs (BLISSGPU_OPT_TAIL_MODE)
    uint32_t pipeline_chunks = 1;      // cut big batches into at least this many chunks (BLISSGPU_OPT_PIPELINE_CHUNKS; measured: the
                                       // per-song tails have a fixed latency per launch, so more chunks than memory needs lose)
    hipEvent_t ev_interop = nullptr;
    bool rolloff_exact_all = false;    // BLISSGPU_OPT_ROLLOFF_EXACT_ALL (tests)
    bool flux_order = false;           // BLISSGPU_OPT_FLUX_ORDER: SpecFlux summed in the reference's bin order (strict; +19 % on the FFT-512 kernel)
    int stft_shape = 0;                // BLISSGPU_OPT_STFT_SHAPE: 0 = four workgroups / CU, window in registers (production); 1 = the narrow form
    int tail_split = 0;                // BLISSGPU_OPT_TAIL_SPLIT: one-chunk batches run the tuning estimate + contraction in two halves
    bool debug_chroma = false;         // BLISSGPU_OPT_DEBUG_CHROMA (tests): keep the chroma matrix / interval means of the last chunk
    bool serial = false;               // BLISSGPU_OPT_SERIAL: single stream (clean per-kernel timings)
    bool counted_live = false;         // this context is in the per-device count of live contexts
    uint64_t ws_limit = 0;             // bytes per chunk slot (set from the free device memory at creation)
    uint32_t cand_budget = bg::CAND_BUDGET_PER_FRAME;  // tuning-candidate pool: slots per chroma frame of a chunk
    // tables
    float2 *tw8192 = nullptr, *tw512 = nullptr;
    float *hann8192 = nullptr, *hannz512 = nullptr, *bt_rwv = nullptr, *bt_dfwv = nullptr;
    double* chroma_bank = nullptr;
    bg::DeviceTables tables{};
    // analysis
    bg::ChunkSlot slot[2];
    uint64_t chunk_seq = 0;            // chunks run so far
    bg::DevBuf<int32_t> dbg_tuning;
    bg::DevBuf<uint32_t> dbg_nbpms;
    bg::DevBuf<double> dbg_chroma, dbg_interval;  // [chroma frames of the last chunk][12], [its songs][10] (BLISSGPU_OPT_DEBUG_CHROMA)
    uint32_t dbg_n = 0;
    bg::Workspace last_ws{};                 // workspace carving of the last chunk (debug taps)
    std::vector<bg::SongDesc> last_songs;    // its descriptors (chunk order; SongDesc::row = the caller's song index)
    uint64_t last_chunks = 0;                // chunks of the last analyze call
    bg::HostFeed feed;
    std::map<uint32_t, bg::ResampleBank> swr_banks;  // filter banks of recently seen input rates (guarded by mu; LRU, SWR_CACHE_BYTES)
    size_t swr_bytes = 0;
    uint64_t swr_clock = 0;
    // distances / playlist ordering scratch
    int n_cus = 0;
    bg::DevBuf<uint32_t> pl_sync, pl_keys;
    bg::DevBuf<uint8_t> pl_tmp;
    bg::DevBuf<unsigned long long> pl_slots;
    bg::DevBuf<float> st_a, st_b, st_m, st_dist;   // staging of the host-pointer distance / playlist forms
    bg::DevBuf<uint8_t> st_out;
    std::vector<float> m_cache;                    // host copy of the matrix in st_m (skip the upload when unchanged)
    float* h_scalar = nullptr;                     // page-locked word the single-pair kernel writes its result to
    // profiling
    bool profiling = false;
    std::vector<bg::EventPair> events[bg::K_COUNT];
};

namespace bg {

struct Prof {  // HIP events around one launch, on the stream the kernel is launched on
    blissgpu_ctx* c;
    int k;
    hipStream_t st;
    EventPair ev{};
    bool on;
    Prof(blissgpu_ctx* ctx, int kernel, hipStream_t stream = nullptr)
        : c(ctx), k(kernel), st(stream ? stream : ctx->stream), on(ctx->profiling) {
        if (on) {
            (void)hipEventCreate(&ev.a);
            (void)hipEventCreate(&ev.b);
            (void)hipEventRecord(ev.a, st);
        }
    }
    ~Prof() {
        if (on) {
            (void)hipEventRecord(ev.b, st);
            c->events[k].push_back(ev);
        }
    }
};

constexpr uint32_t MAX_SONGS_PER_CHUNK = 65535;  // grid.y of the (run, song) launches
constexpr size_t PIPELINE_MIN_CHUNK_BYTES = 512u << 20;  // a pipeline chunk below ~16 three-minute songs no longer fills the GPU

// process-wide default contexts, one per visible device (BLISSGPU_DEFAULT_DEVICES restricts / repeats), created on first use
int default_ctx_count();
int default_ctx_at(int k, blissgpu_ctx** out);
int default_ctx(blissgpu_ctx** out);  // = default_ctx_at(0): the batch / distance / playlist forms without a context argument
void default_ctx_count_batch(int k);  // statistics: one more coalesced batch served by default context k
int64_t single_song_timeout_ms();     // blissgpu_set_single_song_timeout_ms
void default_ctx_forget_failures();   // blissgpu_default_reset: remembered creation failures are forgotten
void front_revive_all();              // ... and the front's retired seats draw traffic again (scheduler.hip)
int live_contexts(int device);        // contexts alive on a HIP device in this process

// scheduler.hip
void scheduler_release(blissgpu_ctx* c);
// Host PCM feed: decoder output in host memory, song by song (any mix of formats, channel counts and sample rates); widening,
// downmix and resampling happen on the device.
// d_rows (device, n_songs x feature_count, may be NULL) receives a copy of the rows that stays on the device.
int analyze_host_songs(blissgpu_ctx* c, const FeedSong* songs, uint32_t n_songs, uint32_t features_version, float* out,
                       int32_t* status, const char* who, float* d_rows = nullptr);
// the uniform form: song i = ptrs[i], lengths[i] FRAMES of `channels` interleaved samples of `sample_format`
int analyze_host_songs(blissgpu_ctx* c, const void* const* ptrs, const uint64_t* lengths, uint32_t n_songs, int sample_format,
                       uint32_t channels, uint32_t features_version, float* out, int32_t* status, const char* who,
                       float* d_rows = nullptr, uint32_t sample_rate = SWR_OUT_RATE);
int resample_bank(blissgpu_ctx* c, uint32_t rate, const ResampleBank** out, const char* who);
// the CPUs local to a HIP device (sysfs local_cpulist of its PCI function); empty when the host has one node or does not say
std::vector<int> device_local_cpus(int device);
// "0-3,8,10-11" -> {0,1,2,3,8,10,11} (device-free: tests/cpp/test_staging.cpp)
std::vector<int> parse_cpulist(const char* text);
int enqueue_decode(blissgpu_ctx* c, const void* d_in, int fmt, uint32_t channels, uint64_t frames, uint32_t rate, float* d_out,
                   uint64_t n_out, hipStream_t st, const char* who);

}  // namespace bg
