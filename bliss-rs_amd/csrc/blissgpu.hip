// blissgpu.hip -- host side of the C ABI declared in include/blissgpu.h: context, constant tables,
// workspace carving, batch scheduling (the GPU replacement of the reference's per-song thread pool,
// src/song/decoder.rs:282-331, and of the five per-descriptor threads, src/song/mod.rs:432-491).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/blissgpu.h"
#include "internal.hpp"

using namespace bg;

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* what, const char* detail) {
    g_last_error = std::string(what) + ": " + (detail ? detail : "");
    return code;
}

#define HIP_TRY(expr)                                                              \
    do {                                                                           \
        hipError_t e_ = (expr);                                                    \
        if (e_ != hipSuccess) return fail(BLISSGPU_ERR_HIP, #expr, hipGetErrorString(e_)); \
    } while (0)

const char* const kKernelNames[K_COUNT] = {
    "pcm_stats_kernel", "fft512_kernel",     "onset_kernel",      "beat_kernel",   "stft8192_kernel", "tune_select_kernel",
    "tune_pass2_kernel", "tune_final_kernel", "chroma_kernel",     "summary_kernel", "assemble_kernel", "pairwise_kernel", "set_distance_kernel", "song_to_song_kernel", "synth_kernel"};

struct EventPair { hipEvent_t a, b; };

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;  // elements
    int ensure(size_t n) {
        if (n <= cap) return BLISSGPU_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 8 + 64;
        hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
        if (e != hipSuccess) { p = nullptr; return fail(BLISSGPU_ERR_OOM, "hipMalloc", hipGetErrorString(e)); }
        cap = want;
        return BLISSGPU_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace

struct blissgpu_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;      // chroma chain + assembly (the caller-visible stream)
    hipStream_t aux_stream = nullptr;  // tempo / timbral / loudness chain, joined before the assembly
    hipEvent_t ev_start = nullptr, ev_fork = nullptr, ev_stft = nullptr, ev_join = nullptr, ev_interop = nullptr;
    bool serial = false;               // BLISSGPU_SERIAL=1: single stream (clean per-kernel timings)
    int overlap_mode = 0;              // BLISSGPU_OVERLAP=1: start the per-song tails before the FFT-8192 kernel (experiment)
    uint64_t ws_limit = 96ull << 30;
    // tables
    float2 *tw8192 = nullptr, *tw512 = nullptr;
    float *hann8192 = nullptr, *hannz512 = nullptr, *bt_rwv = nullptr, *bt_dfwv = nullptr;
    double* chroma_bank = nullptr;
    DeviceTables tables{};
    // workspace: one grow-only slab carved per chunk + small descriptor buffers
    DevBuf<uint8_t> slab;
    DevBuf<uint8_t> desc;       // SongDesc[] + 4 prefix arrays
    uint8_t* h_desc = nullptr;  // pinned staging for desc
    size_t h_desc_cap = 0;
    DevBuf<int32_t> dbg_tuning;
    DevBuf<uint32_t> dbg_nbpms;
    uint32_t dbg_n = 0;
    // playlist ordering scratch
    int n_cus = 0;
    DevBuf<uint32_t> pl_sync, pl_keys;
    DevBuf<uint8_t> pl_tmp;
    DevBuf<unsigned long long> pl_slots;
    Workspace last_ws{};                 // workspace carving of the last chunk (debug taps)
    std::vector<SongDesc> last_songs;    // its descriptors
    // profiling
    bool profiling = false;
    std::vector<EventPair> events[K_COUNT];
};

namespace {

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
    const size_t bank_elems = (size_t)(N_TUNING + 1) * BANK_ROWS * CBINS_PAD;
    HIP_TRY(hipMalloc((void**)&c->chroma_bank, bank_elems * sizeof(double)));
    launch_chroma_bank(c->chroma_bank, c->own_stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->own_stream));
    c->tables = DeviceTables{c->tw8192, c->tw512, c->hann8192, c->hannz512, c->chroma_bank, c->bt_rwv, c->bt_dfwv};
    return BLISSGPU_OK;
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// host-side frame counts; must agree with the reference's framing
void fill_counts(SongDesc& d) {
    const uint64_t n = d.n;
    d.n_t = (uint32_t)((n - W512) / HOP_T + 1);
    d.n_b = (uint32_t)((n - W512) / HOP_B + 1);
    d.n_f = std::max(d.n_t, 2u * d.n_b);
    // src/utils.rs:29-32: rows = (len as f32 / hop as f32).ceil(); the zip with windows() caps it at n/hop + 1
    const uint32_t rows = (uint32_t)ceilf((float)n / (float)HOP_C);
    d.n_c = std::min<uint64_t>(rows, n / HOP_C + 1);
    d.n_e = (uint32_t)((n + 255) / 256);
    d.n_l = (uint32_t)((n + LOUD_W - 1) / LOUD_W);
}

size_t song_ws_bytes(const SongDesc& d) {
    if (!d.ok) return 256;
    size_t b = 0;
    b += (size_t)d.n_t * 12 + (size_t)d.n_b * 8 + (size_t)d.n_e * 8;
    b += (size_t)d.n_c * (CBINS_PAD * 4 + 4);
    b += (size_t)H1_BINS * 4 + N_TUNING * 4 + sizeof(TuningState) + sizeof(TempoState);
    b += (size_t)d.n_c * PIP_MAX_PER_FRAME * 9;
    b += (size_t)d.n_c * (PIP_MAX_PER_FRAME * 4 + 4);
    b += ((size_t)d.n_c / CH_TILE + 1) * 80;
    b += ((size_t)d.n_b / BT_STEP + 2) * 8;
    return b + 4096;
}

struct Carver {
    uint8_t* base;
    size_t off = 0;
    template <typename T>
    T* take(size_t n) {
        off = align_up(off, 256);
        T* p = reinterpret_cast<T*>(base + off);
        off += n * sizeof(T);
        return p;
    }
};

int run_chunk(blissgpu_ctx* c, const float* d_pcm, std::vector<SongDesc>& songs, uint32_t features_version,
              float* d_out) {
    const uint32_t ns = (uint32_t)songs.size();
    if (ns == 0) return BLISSGPU_OK;
    // ---- offsets + tile prefixes ----
    std::vector<uint32_t> pfx_e(ns + 1, 0), pfx_f(ns + 1, 0), pfx_c(ns + 1, 0), pfx_ct(ns + 1, 0), pfx_cw(ns + 1, 0);
    uint64_t tot_t = 0, tot_b = 0, tot_c = 0, tot_e = 0, tot_cand = 0;
    uint32_t max_nb = 0, max_nt = 0, max_runs = 1;
    for (uint32_t i = 0; i < ns; i++) {
        SongDesc& d = songs[i];
        d.t_off = tot_t; d.b_off = tot_b; d.c_off = tot_c; d.e_off = tot_e; d.cand_off = tot_cand;
        if (d.ok) {
            tot_t += d.n_t; tot_b += d.n_b; tot_c += d.n_c; tot_e += d.n_e;
            tot_cand += (uint64_t)d.n_c * PIP_MAX_PER_FRAME;
            max_nb = std::max(max_nb, d.n_b);
            max_nt = std::max(max_nt, d.n_t);
            max_runs = std::max(max_runs, d.n_b / BT_STEP + 1);
        }
        pfx_e[i + 1] = pfx_e[i] + (d.ok ? (d.n_e + 15) / 16 : 0);
        pfx_f[i + 1] = pfx_f[i] + (d.ok ? (d.n_f + F512_TILE - 1) / F512_TILE : 0);
        pfx_c[i + 1] = pfx_c[i] + (d.ok ? (d.n_c + STFT_TILE - 1) / STFT_TILE : 0);
        pfx_ct[i + 1] = pfx_ct[i] + (d.ok ? (d.n_c + CH_TILE - 1) / CH_TILE : 0);
        pfx_cw[i + 1] = pfx_cw[i] + (d.ok ? (d.n_c + 4 * CH_TILE - 1) / (4 * CH_TILE) : 0);
    }
    // ---- descriptors to the device (pinned staging, one async copy) ----
    const size_t desc_bytes = align_up(ns * sizeof(SongDesc), 256) + 5 * align_up((ns + 1) * 4, 256);
    int rc = c->desc.ensure(desc_bytes);
    if (rc) return rc;
    if (desc_bytes > c->h_desc_cap) {
        if (c->h_desc) (void)hipHostFree(c->h_desc);
        c->h_desc = nullptr;
        HIP_TRY(hipHostMalloc((void**)&c->h_desc, desc_bytes + desc_bytes / 4, hipHostMallocDefault));
        c->h_desc_cap = desc_bytes + desc_bytes / 4;
    } else {
        HIP_TRY(hipStreamSynchronize(c->stream));  // the previous chunk may still be reading the staging area
    }
    size_t o = 0;
    const size_t o_songs = o; memcpy(c->h_desc + o, songs.data(), ns * sizeof(SongDesc)); o = align_up(o + ns * sizeof(SongDesc), 256);
    const size_t o_e = o; memcpy(c->h_desc + o, pfx_e.data(), (ns + 1) * 4); o = align_up(o + (ns + 1) * 4, 256);
    const size_t o_f = o; memcpy(c->h_desc + o, pfx_f.data(), (ns + 1) * 4); o = align_up(o + (ns + 1) * 4, 256);
    const size_t o_c = o; memcpy(c->h_desc + o, pfx_c.data(), (ns + 1) * 4); o = align_up(o + (ns + 1) * 4, 256);
    const size_t o_ct = o; memcpy(c->h_desc + o, pfx_ct.data(), (ns + 1) * 4); o = align_up(o + (ns + 1) * 4, 256);
    const size_t o_cw = o; memcpy(c->h_desc + o, pfx_cw.data(), (ns + 1) * 4); o = align_up(o + (ns + 1) * 4, 256);
    HIP_TRY(hipMemcpyAsync(c->desc.p, c->h_desc, o, hipMemcpyHostToDevice, c->stream));

    Batch b{};
    b.pcm = d_pcm;
    b.songs = reinterpret_cast<const SongDesc*>(c->desc.p + o_songs);
    b.n_songs = ns;
    b.pfx_e = reinterpret_cast<const uint32_t*>(c->desc.p + o_e);
    b.pfx_f = reinterpret_cast<const uint32_t*>(c->desc.p + o_f);
    b.pfx_c = reinterpret_cast<const uint32_t*>(c->desc.p + o_c);
    b.pfx_ct = reinterpret_cast<const uint32_t*>(c->desc.p + o_ct);
    b.pfx_cw = reinterpret_cast<const uint32_t*>(c->desc.p + o_cw);
    b.tiles_e = pfx_e[ns]; b.tiles_f = pfx_f[ns]; b.tiles_c = pfx_c[ns]; b.tiles_ct = pfx_ct[ns]; b.tiles_cw = pfx_cw[ns];
    b.total_b = tot_b; b.max_nb = max_nb; b.max_nt = max_nt;

    // ---- carve the workspace ----
    size_t need = 0;
    {
        Carver m{nullptr};
        m.take<float>(tot_t); m.take<float>(tot_t); m.take<float>(tot_t);
        m.take<float>(tot_b); m.take<float>(tot_b);
        m.take<float>(tot_e); m.take<uint32_t>(tot_e);
        m.take<float>(tot_c * CBINS_PAD + 64); m.take<float>(tot_c);
        m.take<uint32_t>((size_t)ns * H1_BINS); m.take<uint32_t>((size_t)ns * N_TUNING);
        m.take<TuningState>(ns);
        m.take<uint32_t>(tot_cand); m.take<uint32_t>(tot_c);
        m.take<double>(tot_cand); m.take<uint8_t>(tot_cand);
        m.take<double>((size_t)b.tiles_ct * 10 + 16);
        m.take<TempoState>(ns);
        m.take<float>((size_t)ns * max_runs); m.take<uint32_t>((size_t)ns * max_runs);
        m.take<float>((size_t)ns * 16);
        need = m.off + 4096;
    }
    if (need > c->slab.cap) HIP_TRY(hipStreamSynchronize(c->stream));
    rc = c->slab.ensure(need);
    if (rc) return rc;
    Carver m{c->slab.p};
    Workspace w{};
    w.centroid = m.take<float>(tot_t); w.rolloff = m.take<float>(tot_t); w.flatness = m.take<float>(tot_t);
    w.flux = m.take<float>(tot_b); w.thresholded = m.take<float>(tot_b);
    w.e256 = m.take<float>(tot_e); w.zc256 = m.take<uint32_t>(tot_e);
    w.spec = m.take<float>(tot_c * CBINS_PAD + 64); w.frame_max = m.take<float>(tot_c);
    w.h1 = m.take<uint32_t>((size_t)ns * H1_BINS); w.hist100 = m.take<uint32_t>((size_t)ns * N_TUNING);
    w.tuning = m.take<TuningState>(ns);
    w.peak_rec = m.take<uint32_t>(tot_cand); w.peak_cnt = m.take<uint32_t>(tot_c);
    w.cand_mag = m.take<double>(tot_cand); w.cand_pb = m.take<uint8_t>(tot_cand);
    w.chroma_part = m.take<double>((size_t)b.tiles_ct * 10 + 16);
    w.tempo = m.take<TempoState>(ns);
    w.run_bpm = m.take<float>((size_t)ns * max_runs); w.run_cnt = m.take<uint32_t>((size_t)ns * max_runs);
    w.runs_pitch = max_runs;
    w.summary = m.take<float>((size_t)ns * 16);

    c->last_ws = w;
    c->last_songs = songs;

    // The reference runs the five descriptors as scoped threads (src/song/mod.rs:432-491).  Here the two
    // FFT-heavy kernels run back to back on the caller-visible stream; the latency-bound tails of the
    // tempo / timbral chains (one workgroup per song: sequential beat tracker, sequential summaries)
    // run beside the chroma chain on the aux stream and are joined before the feature rows are written.
    hipStream_t st = c->stream, sb = c->serial ? c->stream : c->aux_stream;
    const bool two = !c->serial;
    if (two) {
        HIP_TRY(hipEventRecord(c->ev_start, st));
        HIP_TRY(hipStreamWaitEvent(sb, c->ev_start, 0));
    }
    // aux: the HBM-bound PCM statistics pass (only the aux chain consumes it) runs beside the VALU-bound FFT-512
    HIP_TRY(hipMemsetAsync(w.h1, 0, (size_t)ns * H1_BINS * 4, st));
    HIP_TRY(hipMemsetAsync(w.hist100, 0, (size_t)ns * N_TUNING * 4, st));
    if (two && c->overlap_mode == 2) {
        // experiment: the whole tempo / timbral chain on the aux stream, concurrent with the chroma chain
        { Prof p(c, K_PCM_STATS, sb); launch_pcm_stats(b, w, sb); }
        { Prof p(c, K_FFT512, sb); launch_fft512(b, w, c->tables, sb); }
        { Prof p(c, K_ONSET, sb); launch_onset(b, w, sb); }
        { Prof p(c, K_SUMMARY, sb); launch_summary(b, w, sb); }
        { Prof p(c, K_BEAT, sb); launch_beat(b, w, c->tables, sb); }
        HIP_TRY(hipEventRecord(c->ev_join, sb));
        { Prof p(c, K_STFT8192); launch_stft8192(b, w, c->tables, st); }
        { Prof p(c, K_TUNE_SELECT); launch_tune_select(b, w, st); }
        { Prof p(c, K_TUNE_PASS2); launch_tune_pass2(b, w, st); }
        { Prof p(c, K_TUNE_FINAL); launch_tune_final(b, w, st); }
        { Prof p(c, K_CHROMA); launch_chroma(b, w, c->tables, st); }
        HIP_TRY(hipStreamWaitEvent(st, c->ev_join, 0));
        { Prof p(c, K_FINALIZE); launch_finalize(b, w, features_version, d_out, c->dbg_tuning.p, c->dbg_nbpms.p, st); }
        HIP_TRY(hipGetLastError());
        return BLISSGPU_OK;
    }
    { Prof p(c, K_PCM_STATS, sb); launch_pcm_stats(b, w, sb); }
    { Prof p(c, K_FFT512); launch_fft512(b, w, c->tables, st); }
    { Prof p(c, K_ONSET); launch_onset(b, w, st); }
    // aux: the sequential summaries (one lane per song, a few hundred wavefronts in all, memory-latency bound)
    // start as soon as the FFT-512 series exist and run beside the FFT-8192 kernel
    if (two) {
        HIP_TRY(hipEventRecord(c->ev_fork, st));
        HIP_TRY(hipStreamWaitEvent(sb, c->ev_fork, 0));
    }
    { Prof p(c, K_SUMMARY, sb); launch_summary(b, w, sb); }
    { Prof p(c, K_STFT8192); launch_stft8192(b, w, c->tables, st); }
    // The beat tracker (one 256-thread workgroup per song) would displace one of the three FFT-8192 workgroups
    // per CU (168 VGPRs each), so it starts only after that kernel and runs beside the HBM-bound tuning / chroma
    // kernels, which leave registers free.  BLISSGPU_OVERLAP=1 (experiment): start it beside the FFT-8192 kernel.
    if (two && c->overlap_mode == 0) {
        HIP_TRY(hipEventRecord(c->ev_stft, st));
        HIP_TRY(hipStreamWaitEvent(sb, c->ev_stft, 0));
    }
    if (c->overlap_mode != 3) {
        { Prof p(c, K_BEAT, sb); launch_beat(b, w, c->tables, sb); }
        if (two) HIP_TRY(hipEventRecord(c->ev_join, sb));
    }
    { Prof p(c, K_TUNE_SELECT); launch_tune_select(b, w, st); }
    { Prof p(c, K_TUNE_PASS2); launch_tune_pass2(b, w, st); }
    { Prof p(c, K_TUNE_FINAL); launch_tune_final(b, w, st); }
    if (c->overlap_mode == 3) {  // experiment: beat tracker beside the chroma contraction only
        if (two) {
            HIP_TRY(hipEventRecord(c->ev_stft, st));
            HIP_TRY(hipStreamWaitEvent(sb, c->ev_stft, 0));
        }
        { Prof p(c, K_BEAT, sb); launch_beat(b, w, c->tables, sb); }
        if (two) HIP_TRY(hipEventRecord(c->ev_join, sb));
    }
    { Prof p(c, K_CHROMA); launch_chroma(b, w, c->tables, st); }
    if (two) HIP_TRY(hipStreamWaitEvent(st, c->ev_join, 0));
    { Prof p(c, K_FINALIZE); launch_finalize(b, w, features_version, d_out, c->dbg_tuning.p, c->dbg_nbpms.p, st); }
    HIP_TRY(hipGetLastError());
    return BLISSGPU_OK;
}

std::mutex g_default_mu;
blissgpu_ctx* g_default_ctx = nullptr;

int default_ctx(blissgpu_ctx** out) {
    std::lock_guard<std::mutex> lk(g_default_mu);
    if (!g_default_ctx) {
        int rc = blissgpu_ctx_create(0, &g_default_ctx);
        if (rc) return rc;
    }
    *out = g_default_ctx;
    return BLISSGPU_OK;
}

int is_diag(const float* M, uint32_t d) {
    for (uint32_t i = 0; i < d; i++)
        for (uint32_t j = 0; j < d; j++)
            if (i != j && M[i * d + j] != 0.0f) return 0;
    return 1;
}

}  // namespace

// Host-buffer batches (the PCM feed): songs are packed group by group into one of TWO device PCM
// buffers; the H2D copies of group g + 1 run on the copy stream while group g is analysed, so the transfer -- the
// real bottleneck of this entry point (a 3-minute song is 15.9 MB, 7.9 MB as s16) -- is never idle.  BYTES = 4:
// f32 samples copied verbatim; BYTES = 2: s16 samples, widened on the device by pcm_s16_to_f32 (sample / 32768,
// exactly FFmpeg's s16 -> flt conversion, src/song/decoder/ffmpeg.rs:36-109).
template <typename SampleT>
static int analyze_batch_host(const SampleT* pcm, const uint64_t* offsets, const uint64_t* lengths, uint32_t n_songs,
                              uint32_t features_version, float* out, int32_t* status, const char* who) {
    if (n_songs && (!pcm || !offsets || !lengths || !out)) return fail(BLISSGPU_ERR_INVALID, who, "NULL argument");
    const uint32_t d = blissgpu_feature_count(features_version);
    if (!d) return fail(BLISSGPU_ERR_INVALID, who, "features_version must be 1 or 2");
    if (n_songs == 0) return BLISSGPU_OK;
    blissgpu_ctx* c;
    int rc = default_ctx(&c);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    constexpr bool S16 = sizeof(SampleT) == 2;
    // groups of <= 2 GiB of f32 PCM (~128 three-minute songs): large enough to fill the GPU, small enough to pipeline
    const uint64_t group_cap = 512ull << 20;  // samples
    struct Group { uint32_t i0, n; std::vector<uint64_t> doff, dlen; uint64_t total; };
    std::vector<Group> groups;
    for (uint32_t i0 = 0; i0 < n_songs;) {
        Group g{i0, 0, {}, {}, 0};
        uint32_t i1 = i0;
        while (i1 < n_songs && (i1 == i0 || g.total + lengths[i1] <= group_cap)) {
            g.doff.push_back(g.total);
            g.dlen.push_back(lengths[i1]);
            g.total += (lengths[i1] + 63) / 64 * 64;
            i1++;
        }
        g.n = i1 - i0;
        groups.push_back(std::move(g));
        i0 = i1;
    }
    uint64_t max_total = 64, max_n = 1;
    for (const auto& g : groups) { max_total = std::max(max_total, g.total); max_n = std::max<uint64_t>(max_n, g.n); }
    const int nbuf = groups.size() > 1 ? 2 : 1;
    float* d_pcm[2] = {nullptr, nullptr};
    SampleT* d_raw[2] = {nullptr, nullptr};  // s16 staging (S16 only)
    float* d_out[2] = {nullptr, nullptr};
    hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
    hipStream_t copy_stream = nullptr;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(c->stream);
        if (copy_stream) { (void)hipStreamSynchronize(copy_stream); (void)hipStreamDestroy(copy_stream); }
        for (int b = 0; b < 2; b++) {
            (void)hipFree(d_pcm[b]); (void)hipFree(d_raw[b]); (void)hipFree(d_out[b]);
            if (ev_copied[b]) (void)hipEventDestroy(ev_copied[b]);
            if (ev_done[b]) (void)hipEventDestroy(ev_done[b]);
        }
    };
    hipError_t e = hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking);
    for (int b = 0; b < nbuf && e == hipSuccess; b++) {
        e = hipMalloc((void**)&d_pcm[b], max_total * sizeof(float));
        if (e == hipSuccess && S16) e = hipMalloc((void**)&d_raw[b], max_total * sizeof(SampleT));
        if (e == hipSuccess) e = hipMalloc((void**)&d_out[b], max_n * d * sizeof(float));
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev_copied[b], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev_done[b], hipEventDisableTiming);
    }
    if (e != hipSuccess) { cleanup(); return fail(BLISSGPU_ERR_OOM, "hipMalloc(host batch)", hipGetErrorString(e)); }

    auto upload = [&](size_t gi) -> hipError_t {  // H2D of group gi into buffer gi % nbuf, on the copy stream
        const Group& g = groups[gi];
        const int b = (int)(gi % nbuf);
        hipError_t ee = hipSuccess;
        if (gi >= (size_t)nbuf) ee = hipStreamWaitEvent(copy_stream, ev_done[b], 0);  // buffer b is free again
        for (uint32_t k = 0; k < g.n && ee == hipSuccess; k++)
            if (g.dlen[k]) {
                void* dst = S16 ? (void*)(d_raw[b] + g.doff[k]) : (void*)(d_pcm[b] + g.doff[k]);
                ee = hipMemcpyAsync(dst, pcm + offsets[g.i0 + k], g.dlen[k] * sizeof(SampleT), hipMemcpyHostToDevice, copy_stream);
            }
        if (ee == hipSuccess) ee = hipEventRecord(ev_copied[b], copy_stream);
        return ee;
    };

    rc = BLISSGPU_OK;
    e = upload(0);
    for (size_t gi = 0; gi < groups.size() && e == hipSuccess && !rc; gi++) {
        const Group& g = groups[gi];
        const int b = (int)(gi % nbuf);
        e = hipStreamWaitEvent(c->stream, ev_copied[b], 0);
        if (e != hipSuccess) break;
        if (S16) launch_pcm_s16_to_f32(reinterpret_cast<const int16_t*>(d_raw[b]), d_pcm[b], g.total, c->stream);
        rc = blissgpu_analyze_batch_device(c, d_pcm[b], g.doff.data(), g.dlen.data(), g.n, features_version, d_out[b], nullptr);
        if (rc) break;
        e = hipMemcpyAsync(out + (size_t)g.i0 * d, d_out[b], (size_t)g.n * d * sizeof(float), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipEventRecord(ev_done[b], c->stream);
        // the next group's transfer overlaps this group's kernels (pageable sources block the host here, not the GPU)
        if (e == hipSuccess && gi + 1 < groups.size()) e = upload(gi + 1);
        if (status)
            for (uint32_t k = 0; k < g.n; k++)
                status[g.i0 + k] = g.dlen[k] >= (uint64_t)MIN_SAMPLES ? BLISSGPU_SONG_OK : BLISSGPU_SONG_TOO_SHORT;
    }
    if (e == hipSuccess && !rc) e = hipStreamSynchronize(c->stream);
    cleanup();
    if (rc) return rc;
    if (e != hipSuccess) return fail(BLISSGPU_ERR_HIP, who, hipGetErrorString(e));
    return BLISSGPU_OK;
}

extern "C" {

const char* blissgpu_version(void) { return "blissgpu 0.1.0 (gfx950)"; }
const char* blissgpu_last_error(void) { return g_last_error.c_str(); }

const char* blissgpu_strerror(int code) {
    switch (code) {
        case BLISSGPU_OK: return "ok";
        case BLISSGPU_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU path)";
        case BLISSGPU_ERR_INVALID: return "invalid argument";
        case BLISSGPU_ERR_HIP: return "HIP runtime error";
        case BLISSGPU_ERR_NAN: return "a distance is NaN";
        case BLISSGPU_ERR_OOM: return "out of device memory";
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
    se = hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking);
    if (se == hipSuccess) se = hipEventCreateWithFlags(&c->ev_start, hipEventDisableTiming);
    if (se == hipSuccess) se = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    if (se == hipSuccess) se = hipEventCreateWithFlags(&c->ev_stft, hipEventDisableTiming);
    if (se == hipSuccess) se = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
    if (se == hipSuccess) se = hipEventCreateWithFlags(&c->ev_interop, hipEventDisableTiming);
    if (se != hipSuccess) { blissgpu_ctx_destroy(c); return fail(BLISSGPU_ERR_HIP, "aux stream/events", hipGetErrorString(se)); }
    if (const char* e = getenv("BLISSGPU_SERIAL")) c->serial = (e[0] == '1');
    if (const char* e = getenv("BLISSGPU_OVERLAP")) c->overlap_mode = atoi(e);
    int rc = build_tables(c);
    if (rc) { blissgpu_ctx_destroy(c); return rc; }
    *out = c;
    return BLISSGPU_OK;
}

int blissgpu_ctx_destroy(blissgpu_ctx* c) {
    if (!c) return BLISSGPU_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& v : c->events)
        for (auto& ev : v) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    (void)hipFree(c->tw8192); (void)hipFree(c->tw512); (void)hipFree(c->hann8192); (void)hipFree(c->hannz512);
    (void)hipFree(c->bt_rwv); (void)hipFree(c->bt_dfwv); (void)hipFree(c->chroma_bank);
    c->slab.release(); c->desc.release(); c->dbg_tuning.release(); c->dbg_nbpms.release();
    if (c->h_desc) (void)hipHostFree(c->h_desc);
    if (c->aux_stream) { (void)hipStreamSynchronize(c->aux_stream); (void)hipStreamDestroy(c->aux_stream); }
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_stft) (void)hipEventDestroy(c->ev_stft);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_interop) (void)hipEventDestroy(c->ev_interop);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return BLISSGPU_OK;
}

int blissgpu_ctx_set_stream(blissgpu_ctx* c, void* s) {
    if (!c) return fail(BLISSGPU_ERR_INVALID, "blissgpu_ctx_set_stream", "ctx is NULL");
    (void)hipStreamSynchronize(c->stream);
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return BLISSGPU_OK;
}
void* blissgpu_ctx_get_stream(blissgpu_ctx* c) { return c ? (void*)c->stream : nullptr; }

// Stream interop for hosts that keep their own streams (e.g. torch's current stream, NULL = the legacy default
// stream): order the context's stream after / before work queued on another stream without a host synchronisation.
int blissgpu_ctx_wait_stream(blissgpu_ctx* c, void* producer_stream) {
    if (!c) return fail(BLISSGPU_ERR_INVALID, "blissgpu_ctx_wait_stream", "ctx is NULL");
    if ((hipStream_t)producer_stream == c->stream) return BLISSGPU_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventRecord(c->ev_interop, (hipStream_t)producer_stream));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_interop, 0));
    return BLISSGPU_OK;
}
int blissgpu_ctx_signal_stream(blissgpu_ctx* c, void* consumer_stream) {
    if (!c) return fail(BLISSGPU_ERR_INVALID, "blissgpu_ctx_signal_stream", "ctx is NULL");
    if ((hipStream_t)consumer_stream == c->stream) return BLISSGPU_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventRecord(c->ev_interop, c->stream));
    HIP_TRY(hipStreamWaitEvent((hipStream_t)consumer_stream, c->ev_interop, 0));
    return BLISSGPU_OK;
}

int blissgpu_ctx_set_workspace_limit(blissgpu_ctx* c, uint64_t bytes) {
    if (!c || bytes < (64ull << 20)) return fail(BLISSGPU_ERR_INVALID, "blissgpu_ctx_set_workspace_limit", "limit < 64 MiB");
    c->ws_limit = bytes;
    return BLISSGPU_OK;
}

int blissgpu_ctx_synchronize(blissgpu_ctx* c) {
    if (!c) return fail(BLISSGPU_ERR_INVALID, "blissgpu_ctx_synchronize", "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BLISSGPU_OK;
}

int blissgpu_analyze_batch_device(blissgpu_ctx* c, const float* d_pcm, const uint64_t* offsets, const uint64_t* lengths,
                                  uint32_t n_songs, uint32_t features_version, float* d_out, int32_t* d_status) {
    if (!c || (n_songs && (!d_pcm || !offsets || !lengths || !d_out)))
        return fail(BLISSGPU_ERR_INVALID, "blissgpu_analyze_batch_device", "NULL argument");
    if (features_version != BLISSGPU_FEATURES_V1 && features_version != BLISSGPU_FEATURES_V2)
        return fail(BLISSGPU_ERR_INVALID, "blissgpu_analyze_batch_device", "features_version must be 1 or 2");
    if (n_songs == 0) return BLISSGPU_OK;
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    if ((rc = c->dbg_tuning.ensure(n_songs))) return rc;
    if ((rc = c->dbg_nbpms.ensure(n_songs))) return rc;
    c->dbg_n = n_songs;

    std::vector<int32_t> status(n_songs);
    std::vector<SongDesc> chunk;
    size_t chunk_bytes = 0;
    for (uint32_t i = 0; i < n_songs; i++) {
        SongDesc d{};
        d.pcm_off = offsets[i];
        d.n = lengths[i];
        d.row = i;
        d.ok = lengths[i] >= (uint64_t)MIN_SAMPLES;  // src/song/mod.rs:417-430
        status[i] = d.ok ? BLISSGPU_SONG_OK : BLISSGPU_SONG_TOO_SHORT;
        if (d.ok) fill_counts(d);
        const size_t sb = song_ws_bytes(d);
        if (!chunk.empty() && chunk_bytes + sb > c->ws_limit) {
            if ((rc = run_chunk(c, d_pcm, chunk, features_version, d_out))) return rc;
            chunk.clear();
            chunk_bytes = 0;
        }
        chunk.push_back(d);
        chunk_bytes += sb;
    }
    if ((rc = run_chunk(c, d_pcm, chunk, features_version, d_out))) return rc;
    if (d_status) {
        // pageable source: HIP stages it before returning
        HIP_TRY(hipMemcpyAsync(d_status, status.data(), n_songs * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return BLISSGPU_OK;
}

int blissgpu_analyze_batch(const float* pcm, const uint64_t* offsets, const uint64_t* lengths, uint32_t n_songs,
                           uint32_t features_version, float* out, int32_t* status) {
    return analyze_batch_host<float>(pcm, offsets, lengths, n_songs, features_version, out, status, "blissgpu_analyze_batch");
}

int blissgpu_analyze_batch_s16(const int16_t* pcm, const uint64_t* offsets, const uint64_t* lengths, uint32_t n_songs,
                               uint32_t features_version, float* out, int32_t* status) {
    return analyze_batch_host<int16_t>(pcm, offsets, lengths, n_songs, features_version, out, status, "blissgpu_analyze_batch_s16");
}

int blissgpu_pcm_s16_to_f32_device(blissgpu_ctx* c, const int16_t* d_in, uint64_t n, float* d_out) {
    if (!c || (n && (!d_in || !d_out))) return fail(BLISSGPU_ERR_INVALID, "blissgpu_pcm_s16_to_f32_device", "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    launch_pcm_s16_to_f32(d_in, d_out, n, c->stream);
    HIP_TRY(hipGetLastError());
    return BLISSGPU_OK;
}

int blissgpu_host_alloc(void** p, uint64_t bytes) {
    if (!p) return fail(BLISSGPU_ERR_INVALID, "blissgpu_host_alloc", "NULL");
    hipError_t e = hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? BLISSGPU_ERR_OOM : BLISSGPU_ERR_NO_DEVICE, "hipHostMalloc", hipGetErrorString(e));
    return BLISSGPU_OK;
}
int blissgpu_host_free(void* p) { HIP_TRY(hipHostFree(p)); return BLISSGPU_OK; }

int blissgpu_analyze(const float* pcm, uint64_t len, uint32_t features_version, float* out, int32_t* status) {
    const uint64_t off = 0;
    int32_t st = 0;
    const float dummy = 0.0f;
    int rc = blissgpu_analyze_batch(pcm ? pcm : &dummy, &off, &len, 1, features_version, out, &st);
    if (status) *status = st;
    return rc;
}

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
    HIP_TRY(hipSetDevice(c->device));
    int diag = 0;
    if (metric == BLISSGPU_METRIC_MAHALANOBIS) {
        std::vector<float> hM((size_t)d * d);
        HIP_TRY(hipMemcpyAsync(hM.data(), d_M, hM.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        diag = is_diag(hM.data(), d);
    }
    {
        Prof p(c, K_PAIRWISE);
        launch_pairwise(d_A, n, d_B, m, d, metric, d_M, diag, d_out, ld_out, c->stream);
    }
    HIP_TRY(hipGetLastError());
    return BLISSGPU_OK;
}

int blissgpu_pairwise(const float* A, uint64_t n, const float* B, uint64_t m, uint32_t d, int metric, const float* M,
                      float* out) {
    if (!A || !B || !out) return fail(BLISSGPU_ERR_INVALID, "blissgpu_pairwise", "NULL argument");
    if (d == 0 || d > 64 || metric < 0 || metric > 2) return fail(BLISSGPU_ERR_INVALID, "blissgpu_pairwise", "bad d / metric");
    if (metric == BLISSGPU_METRIC_MAHALANOBIS && !M) return fail(BLISSGPU_ERR_INVALID, "blissgpu_pairwise", "mahalanobis needs M");
    if (n == 0 || m == 0) return BLISSGPU_OK;
    blissgpu_ctx* c;
    int rc = default_ctx(&c);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    float *dA = nullptr, *dB = nullptr, *dM = nullptr, *dO = nullptr;
    // rows of the output are produced in slabs of <= 4 GiB so host-sized problems never need n*m device memory
    const uint64_t slab_rows = std::max<uint64_t>(1, std::min<uint64_t>(n, (1ull << 30) / std::max<uint64_t>(m, 1)));
    HIP_TRY(hipMalloc((void**)&dA, n * d * sizeof(float)));
    const bool self = (A == B && n == m);  // self-distance matrix: one device copy, symmetric kernel
    hipError_t e = self ? hipSuccess : hipMalloc((void**)&dB, m * d * sizeof(float));
    if (self) dB = dA;
    if (e == hipSuccess) e = hipMalloc((void**)&dO, slab_rows * m * sizeof(float));
    if (e == hipSuccess && metric == BLISSGPU_METRIC_MAHALANOBIS) e = hipMalloc((void**)&dM, (size_t)d * d * sizeof(float));
    if (e != hipSuccess) {
        (void)hipFree(dA); if (!self) (void)hipFree(dB); (void)hipFree(dO); (void)hipFree(dM);
        return fail(BLISSGPU_ERR_OOM, "hipMalloc(pairwise)", hipGetErrorString(e));
    }
    rc = BLISSGPU_OK;
    e = hipMemcpyAsync(dA, A, n * d * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && !self) e = hipMemcpyAsync(dB, B, m * d * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && dM) e = hipMemcpyAsync(dM, M, (size_t)d * d * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) rc = fail(BLISSGPU_ERR_HIP, "hipMemcpyAsync", hipGetErrorString(e));
    for (uint64_t r0 = 0; !rc && r0 < n; r0 += slab_rows) {
        const uint64_t rows = std::min(slab_rows, n - r0);
        rc = blissgpu_pairwise_device(c, dA + r0 * d, rows, dB, m, d, metric, dM, dO, m);
        if (!rc) {
            e = hipMemcpyAsync(out + r0 * m, dO, rows * m * sizeof(float), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) rc = fail(BLISSGPU_ERR_HIP, "copy back", hipGetErrorString(e));
        }
    }
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(dA); if (!self) (void)hipFree(dB); (void)hipFree(dO); (void)hipFree(dM);
    return rc;
}

int blissgpu_distance(const float* a, const float* b, uint32_t d, int metric, const float* M, float* out) {
    return blissgpu_pairwise(a, 1, b, 1, d, metric, M, out);
}

// ---- playlist ordering (src/playlist.rs:24-59, 256-326) ----
static int playlist_args_ok(const char* who, const void* a, const void* b, const void* o, uint32_t n_seeds, uint64_t n,
                            uint32_t d, int metric, const float* M) {
    if (!a || !b || !o) return fail(BLISSGPU_ERR_INVALID, who, "NULL argument");
    if (d == 0 || d > 64 || metric < 0 || metric > 2) return fail(BLISSGPU_ERR_INVALID, who, "bad d / metric");
    if (metric == BLISSGPU_METRIC_MAHALANOBIS && !M) return fail(BLISSGPU_ERR_INVALID, who, "mahalanobis needs M");
    if (n_seeds == 0) return fail(BLISSGPU_ERR_INVALID, who, "empty seed set");
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
    HIP_TRY(hipSetDevice(c->device));
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
    HIP_TRY(hipSetDevice(c->device));
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
    HIP_TRY(sort_pairs_u32(c->pl_tmp.p, &tmp_bytes, keys_in, keys_out, idx_in, d_order, n32, c->stream));
    return nan_check(c, c->pl_sync.p + 1, "blissgpu_closest_to_songs_device");
}

int blissgpu_song_to_song_device(blissgpu_ctx* c, const float* d_seeds, uint32_t n_seeds, const float* d_cand, uint64_t n,
                                 uint32_t d, int metric, const float* d_M, uint32_t* d_order) {
    if (!c) return fail(BLISSGPU_ERR_INVALID, "blissgpu_song_to_song_device", "ctx is NULL");
    int rc = playlist_args_ok("blissgpu_song_to_song_device", d_seeds, d_cand, d_order, n_seeds, n, d, metric, d_M);
    if (rc) return rc;
    if (n == 0) return BLISSGPU_OK;
    HIP_TRY(hipSetDevice(c->device));
    // one workgroup per 256 candidates up to one per CU (all workgroups must be co-resident: the kernel spins on
    // a grid barrier); each thread then owns ceil(n / (256 G)) <= 64 candidates
    // four candidates per thread (register-resident) is the sweet spot: fewer workgroups make the grid barrier and the
    // slot reduction cheaper (100 k songs: 98 workgroups, 7.3 us per step; 256 workgroups with two each: 11.7 us)
    uint32_t grid = (uint32_t)std::min<uint64_t>((n + 1023) / 1024, (uint64_t)std::min(256, std::max(1, c->n_cus)));
    if (const char* e = getenv("BLISSGPU_S2S_GRID")) grid = std::min<uint32_t>(256u, (uint32_t)std::max(1, atoi(e)));  // experiment
    if ((n + (uint64_t)grid * 256 - 1) / ((uint64_t)grid * 256) > 64)
        return fail(BLISSGPU_ERR_INVALID, "blissgpu_song_to_song_device", "pool too large for one launch (> 64 candidates per thread)");
    rc = c->pl_sync.ensure(4);
    if (!rc) rc = c->pl_slots.ensure((size_t)2 * grid);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(c->pl_sync.p, 0, 4 * sizeof(uint32_t), c->stream));
    {
        Prof p(c, K_SONG_TO_SONG);
        launch_song_to_song(d_seeds, n_seeds, d_cand, (uint32_t)n, d, metric, d_M, d_order, c->pl_slots.p, c->pl_sync.p, grid,
                            c->stream);
    }
    HIP_TRY(hipGetLastError());
    return nan_check(c, c->pl_sync.p + 1, "blissgpu_song_to_song_device");
}

// host-pointer wrappers: stage seeds / candidates / M, run the device form, copy the result back
namespace {
struct PlStage {
    float *seeds = nullptr, *cand = nullptr, *M = nullptr;
    void* out = nullptr;
    float* dist = nullptr;
    ~PlStage() { (void)hipFree(seeds); (void)hipFree(cand); (void)hipFree(M); (void)hipFree(out); (void)hipFree(dist); }
    int up(blissgpu_ctx* c, const float* h_seeds, uint32_t n_seeds, const float* h_cand, uint64_t n, uint32_t d, int metric,
           const float* h_M, size_t out_bytes, bool want_dist) {
        hipError_t e = hipMalloc((void**)&seeds, (size_t)n_seeds * d * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void**)&cand, std::max<size_t>(1, n * d) * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(&out, std::max<size_t>(4, out_bytes));
        if (e == hipSuccess && want_dist) e = hipMalloc((void**)&dist, std::max<size_t>(1, n) * sizeof(float));
        if (e == hipSuccess && metric == BLISSGPU_METRIC_MAHALANOBIS) e = hipMalloc((void**)&M, (size_t)d * d * sizeof(float));
        if (e != hipSuccess) return fail(BLISSGPU_ERR_OOM, "hipMalloc(playlist)", hipGetErrorString(e));
        e = hipMemcpyAsync(seeds, h_seeds, (size_t)n_seeds * d * sizeof(float), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess && n) e = hipMemcpyAsync(cand, h_cand, n * d * sizeof(float), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess && M) e = hipMemcpyAsync(M, h_M, (size_t)d * d * sizeof(float), hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) return fail(BLISSGPU_ERR_HIP, "hipMemcpyAsync(playlist)", hipGetErrorString(e));
        return BLISSGPU_OK;
    }
    int down(blissgpu_ctx* c, void* h_out, size_t out_bytes, float* h_dist, uint64_t n) {
        hipError_t e = hipSuccess;
        if (out_bytes) e = hipMemcpyAsync(h_out, out, out_bytes, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess && h_dist && n) e = hipMemcpyAsync(h_dist, dist, n * sizeof(float), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return fail(BLISSGPU_ERR_HIP, "copy back(playlist)", hipGetErrorString(e));
        return BLISSGPU_OK;
    }
};
}  // namespace

int blissgpu_set_distance(const float* seeds, uint32_t n_seeds, const float* cand, uint64_t n, uint32_t d, int metric,
                          const float* M, float* out) {
    int rc = playlist_args_ok("blissgpu_set_distance", seeds, cand, out, n_seeds, n, d, metric, M);
    if (rc || n == 0) return rc;
    blissgpu_ctx* c;
    rc = default_ctx(&c);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    PlStage s;
    rc = s.up(c, seeds, n_seeds, cand, n, d, metric, M, n * sizeof(float), false);
    if (!rc) rc = blissgpu_set_distance_device(c, s.seeds, n_seeds, s.cand, n, d, metric, s.M, (float*)s.out);
    if (!rc) rc = s.down(c, out, n * sizeof(float), nullptr, 0);
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
    HIP_TRY(hipSetDevice(c->device));
    PlStage s;
    rc = s.up(c, seeds, n_seeds, cand, n, d, metric, M, n * sizeof(uint32_t), dist != nullptr);
    if (!rc) rc = blissgpu_closest_to_songs_device(c, s.seeds, n_seeds, s.cand, n, d, metric, s.M, (uint32_t*)s.out, s.dist);
    if (!rc) rc = s.down(c, order, n * sizeof(uint32_t), dist, n);
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
    HIP_TRY(hipSetDevice(c->device));
    PlStage s;
    rc = s.up(c, seeds, n_seeds, cand, n, d, metric, M, n * sizeof(uint32_t), false);
    if (!rc) rc = blissgpu_song_to_song_device(c, s.seeds, n_seeds, s.cand, n, d, metric, s.M, (uint32_t*)s.out);
    if (!rc) rc = s.down(c, order, n * sizeof(uint32_t), nullptr, 0);
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
    if (!c) return fail(BLISSGPU_ERR_INVALID, "blissgpu_memcpy_h2d", "ctx is NULL");
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BLISSGPU_OK;
}
int blissgpu_memcpy_d2h(blissgpu_ctx* c, void* dst, const void* src, uint64_t bytes) {
    if (!c) return fail(BLISSGPU_ERR_INVALID, "blissgpu_memcpy_d2h", "ctx is NULL");
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BLISSGPU_OK;
}

int blissgpu_synth_white_noise_device(blissgpu_ctx* c, float* d_pcm, const uint64_t* offsets, const uint64_t* lengths,
                                      uint32_t n_songs, uint32_t first_song_index) {
    if (!c || !d_pcm || !offsets || !lengths) return fail(BLISSGPU_ERR_INVALID, "blissgpu_synth_white_noise_device", "NULL argument");
    if (n_songs == 0) return BLISSGPU_OK;
    HIP_TRY(hipSetDevice(c->device));
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
    uint32_t* d_pfx = nullptr;
    HIP_TRY(hipMalloc((void**)&d_songs, n_songs * sizeof(SongDesc)));
    hipError_t e = hipMalloc((void**)&d_pfx, (n_songs + 1) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpy(d_songs, songs.data(), n_songs * sizeof(SongDesc), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_pfx, pfx.data(), (n_songs + 1) * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        Prof p(c, K_SYNTH);
        launch_synth(d_pcm, d_songs, n_songs, d_pfx, pfx[n_songs], first_song_index, c->stream);
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_songs);
    (void)hipFree(d_pfx);
    if (e != hipSuccess) return fail(BLISSGPU_ERR_HIP, "synth", hipGetErrorString(e));
    return BLISSGPU_OK;
}

int blissgpu_profile_enable(blissgpu_ctx* c, int enable) {
    if (!c) return fail(BLISSGPU_ERR_INVALID, "blissgpu_profile_enable", "ctx is NULL");
    c->profiling = enable != 0;
    return BLISSGPU_OK;
}

int blissgpu_profile_reset(blissgpu_ctx* c) {
    if (!c) return fail(BLISSGPU_ERR_INVALID, "blissgpu_profile_reset", "ctx is NULL");
    (void)hipStreamSynchronize(c->stream);
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
    HIP_TRY(hipStreamSynchronize(c->stream));
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

int blissgpu_debug_last_tuning(blissgpu_ctx* c, double* tuning, uint32_t* n_bpms, uint32_t n_songs) {
    if (!c || n_songs > c->dbg_n) return fail(BLISSGPU_ERR_INVALID, "blissgpu_debug_last_tuning", "no such batch");
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
    if (!c || !dst || song >= c->last_songs.size()) return fail(BLISSGPU_ERR_INVALID, "blissgpu_debug_fetch", "no such song in the last chunk");
    HIP_TRY(hipStreamSynchronize(c->stream));
    const SongDesc& d = c->last_songs[song];
    const Workspace& w = c->last_ws;
    const void* src = nullptr;
    uint64_t n = 0, esz = 4;
    switch (what) {
        case BLISSGPU_DEBUG_CENTROID: src = w.centroid + d.t_off; n = d.n_t; break;
        case BLISSGPU_DEBUG_ROLLOFF: src = w.rolloff + d.t_off; n = d.n_t; break;
        case BLISSGPU_DEBUG_FLATNESS: src = w.flatness + d.t_off; n = d.n_t; break;
        case BLISSGPU_DEBUG_FLUX: src = w.flux + d.b_off; n = d.n_b; break;
        case BLISSGPU_DEBUG_THRESHOLDED: src = w.thresholded + d.b_off; n = d.n_b; break;
        case BLISSGPU_DEBUG_RUN_BPM: src = w.run_bpm + (size_t)song * w.runs_pitch; n = d.ok ? (d.n_b >= (uint32_t)BT_STEP ? (d.n_b - BT_STEP) / BT_STEP + 1 : 0) : 0; break;
        case BLISSGPU_DEBUG_RUN_COUNT: src = w.run_cnt + (size_t)song * w.runs_pitch; n = d.ok ? (d.n_b >= (uint32_t)BT_STEP ? (d.n_b - BT_STEP) / BT_STEP + 1 : 0) : 0; break;
        case BLISSGPU_DEBUG_SPECTROGRAM: src = w.spec + d.c_off * (size_t)CBINS_PAD; n = (uint64_t)d.n_c * CBINS_PAD; break;
        case BLISSGPU_DEBUG_ENERGY256: src = w.e256 + d.e_off; n = d.n_e; break;
        case BLISSGPU_DEBUG_CROSSINGS256: src = w.zc256 + d.e_off; n = d.n_e; break;
        case BLISSGPU_DEBUG_PITCH_HIST: src = w.hist100 + (size_t)song * N_TUNING; n = N_TUNING; break;
        default: return fail(BLISSGPU_ERR_INVALID, "blissgpu_debug_fetch", "unknown tap");
    }
    if (!d.ok) n = 0;
    if (n_elems) *n_elems = n;
    const uint64_t k = std::min(n, max_elems);
    if (k) HIP_TRY(hipMemcpy(dst, src, k * esz, hipMemcpyDeviceToHost));
    return BLISSGPU_OK;
}

}  // extern "C"
