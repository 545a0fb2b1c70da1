// scheduler.hip -- batches of Song::analyze on the device: the GPU replacement of the reference's per-song thread pool
// (Decoder::analyze_paths_with_options, src/song/decoder.rs:278-332) and of the five per-descriptor threads of one
// analysis (src/song/mod.rs:432-491).
//
//   plan      songs are ordered by length (longest first: "length bucketing") and cut into chunks whose scratch
//             workspace fits one of the context's TWO chunk slots;
//   schedule  a chunk is enqueued in two halves -- front: the FFT kernels on the main stream, its per-song tails and its
//             tuning estimate on two high-priority side streams; back: the chroma contraction and the row assembly -- and
//             the back half of chunk k follows the front half of chunk k + 1, so the latency-bound kernels of one chunk
//             hide beside the FFT kernels of the next; a slot is reused two chunks later, after its own assembly;
//   feed      host PCM (f32 or s16, mono or interleaved multi-channel) is shipped group by group into two device
//             buffers, the copy of group g + 1 overlapping the analysis of group g;
//   front     concurrent single-song calls (N worker threads each calling Song::analyze, src/song/decoder.rs:299-329)
//             are coalesced: whoever finds no batch running analyses every request that has queued up as ONE batch.
#include <string.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <numeric>

#include "coalescing_front.hpp"
#include "ctx.hpp"

using namespace bg;

namespace {

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// host-side frame counts; must agree with the reference's framing
void fill_counts(SongDesc& d) {
    const uint64_t n = d.n;
    d.n_t = (uint32_t)((n - W512) / HOP_T + 1);
    d.n_b = (uint32_t)((n - W512) / HOP_B + 1);
    d.n_f = std::max(d.n_t, 2u * d.n_b);
    // src/utils.rs:29-32: rows = (len as f32 / hop as f32).ceil(); the zip with windows() caps it at n/hop + 1
    const uint32_t rows = (uint32_t)ceilf((float)n / (float)HOP_C);
    d.n_c = (uint32_t)std::min<uint64_t>(rows, n / HOP_C + 1);
    d.n_e = (uint32_t)((n + 255) / 256);
    d.n_l = (uint32_t)((n + LOUD_W - 1) / LOUD_W);
}

// scratch bytes one song adds to a chunk (upper bound of the carving below, alignment slack included)
size_t song_ws_bytes(const SongDesc& d) {
    if (!d.ok) return 256;
    size_t b = 0;
    b += (size_t)d.n_t * 12 + (size_t)d.n_b * 8 + (size_t)d.n_e * 8;
    b += (size_t)d.n_c * (CBINS_PAD * 4 + 4);
    b += (size_t)H1_BINS * 4 + N_TUNING * 4 + sizeof(TuningState) + sizeof(TempoState);
    b += (size_t)d.n_c * CAND_BUDGET_PER_FRAME * 9;
    b += (size_t)d.n_c * (PIP_MAX_PER_FRAME * 4 + 4);
    b += ((size_t)d.n_c / CH_TILE + 1) * 80;
    b += ((size_t)d.n_b / BT_STEP + 2) * (8 + BT_PRE_STRIDE * 4);
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

struct ChunkTotals {
    uint64_t tot_t = 0, tot_b = 0, tot_c = 0, tot_e = 0;
    uint32_t max_nb = 0, max_nt = 0, max_runs = 1;
    uint32_t tiles_ct = 0;
    uint64_t cand_cap = 0;
};

Workspace carve(uint8_t* base, uint32_t ns, const ChunkTotals& t, size_t* bytes) {
    Carver m{base};
    Workspace w{};
    w.centroid = m.take<float>(t.tot_t); w.rolloff = m.take<float>(t.tot_t); w.flatness = m.take<float>(t.tot_t);
    w.flux = m.take<float>(t.tot_b); w.thresholded = m.take<float>(t.tot_b);
    w.e256 = m.take<float>(t.tot_e); w.zc256 = m.take<uint32_t>(t.tot_e);
    // The spectrogram and the peak records -- both first written by the FFT-8192 kernel -- lie back to back: until that
    // kernel starts, the FFT-512 kernel's unproven-rolloff frames borrow the stretch (256 words per entry; consumed by
    // rolloff_fix_kernel).  18 412 bytes per chroma frame = 17.9 entries per 2205 samples, and a song has one timbral frame
    // per 128 samples (17.2 per 2205): EVERY frame of the chunk would fit, so no entry is ever turned away.
    static_assert((size_t)(CBINS_PAD + PIP_MAX_PER_FRAME) * 4 * HOP_T >= (size_t)256 * 4 * HOP_C,
                  "the borrowed stretch must hold one 256-word entry per timbral frame: bytes per chroma frame x samples per "
                  "timbral frame >= bytes per entry x samples per chroma frame");
    w.spec = m.take<float>(t.tot_c * CBINS_PAD + 64);
    w.peak_rec = m.take<uint32_t>(t.tot_c * PIP_MAX_PER_FRAME);
    w.roll_fix_bytes = (size_t)(reinterpret_cast<uint8_t*>(w.peak_rec + t.tot_c * PIP_MAX_PER_FRAME) - reinterpret_cast<uint8_t*>(w.spec));
    w.frame_max = m.take<float>(t.tot_c);
    w.roll_fix = m.take<RollFix>(1);
    w.h1 = m.take<uint32_t>((size_t)ns * H1_BINS); w.hist100 = m.take<uint32_t>((size_t)ns * N_TUNING);
    w.tuning = m.take<TuningState>(ns);
    w.peak_cnt = m.take<uint32_t>(t.tot_c);
    w.cand_mag = m.take<double>(t.cand_cap); w.cand_pb = m.take<uint8_t>(t.cand_cap);
    w.cand_cursor = m.take<uint32_t>(4);
    w.cand_cap = (uint32_t)t.cand_cap;
    w.chroma_part = m.take<double>((size_t)t.tiles_ct * 10 + 16);
    w.tempo = m.take<TempoState>(ns);
    w.run_bpm = m.take<float>((size_t)ns * t.max_runs); w.run_cnt = m.take<uint32_t>((size_t)ns * t.max_runs);
    w.runs_pitch = t.max_runs;
    w.bt_pre = m.take<float>((size_t)(t.tot_b / BT_STEP + ns + 1) * BT_PRE_STRIDE);
    w.summary = m.take<float>((size_t)ns * 16);
    *bytes = m.off + 4096;
    return w;
}

// A stream whose kernels may only run on `cus` of the device's CUs (hipExtStreamCreateWithCUMask).  The driver deals the
// mask's bits over the XCDs in turn, so the first `cus` bits are cus / 8 CUs on each of the eight.
int ensure_mask_stream(blissgpu_ctx* c, int cus) {
    cus = std::max(1, std::min(cus, c->n_cus));
    if (c->mask_stream && c->mask_cus == cus) return BLISSGPU_OK;
    if (c->mask_stream) { (void)hipStreamSynchronize(c->mask_stream); (void)hipStreamDestroy(c->mask_stream); c->mask_stream = nullptr; }
    std::vector<uint32_t> mask((size_t)(c->n_cus + 31) / 32, 0u);
    for (int i = 0; i < cus; i++) mask[(size_t)i / 32] |= 1u << (i % 32);
    HIP_TRY(hipExtStreamCreateWithCUMask(&c->mask_stream, (uint32_t)mask.size(), mask.data()));
    c->mask_cus = cus;
    return BLISSGPU_OK;
}

int ensure_slot_events(ChunkSlot& s) {
    if (s.ev_free) return BLISSGPU_OK;
    hipEvent_t* evs[] = {&s.ev_start, &s.ev_fork, &s.ev_stft, &s.ev_sel, &s.ev_tune, &s.ev_sum, &s.ev_chroma, &s.ev_desc, &s.ev_free, &s.ev_acf, &s.ev_beat};
    for (hipEvent_t* e : evs) HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
    for (hipEvent_t& e : s.ev_piece) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return BLISSGPU_OK;
}

// One chunk = the songs [songs, songs + ns) (already ordered), enqueued in two halves so that a batch is software-pipelined
// at enqueue time (see blissgpu_analyze_batch_device):
//   front  descriptors, FFT-512 + onset, FFT-8192 on the main stream; behind them, on the two high-priority side streams,
//          the per-song tails (sequential summaries, beat tracker: they need only the FFT-512 series) beside the FFT-8192
//          kernel, and the tuning estimate (histogram walk, radix select) beside whatever the main stream runs next;
//   back   the chroma contraction (needs the tuning) on the main stream and the row assembly on the aux stream.
// The back half of chunk k is enqueued AFTER the front half of chunk k + 1: the latency-bound tuning kernels of chunk k
// then hide beside chunk k + 1's FFT kernels instead of sitting between the FFT-8192 kernel and the contraction.
// Nothing here waits for the device except the reuse of a slot's pinned descriptor staging and buffer growth.
int chunk_front(blissgpu_ctx* c, ChunkSlot& slot, const float* d_pcm, SongDesc* songs, uint32_t ns, bool only_chunk) {
    int rc = ensure_slot_events(slot);
    if (rc) return rc;
    // ---- offsets into the batch-wide series + tile prefixes ----
    std::vector<uint32_t> pfx_f(ns + 1, 0), pfx_c(ns + 1, 0), pfx_ct(ns + 1, 0), pfx_cw(ns + 1, 0);
    ChunkTotals t;
    for (uint32_t i = 0; i < ns; i++) {
        SongDesc& d = songs[i];
        d.t_off = t.tot_t; d.b_off = t.tot_b; d.c_off = t.tot_c; d.e_off = t.tot_e;
        if (d.ok) {
            t.tot_t += d.n_t; t.tot_b += d.n_b; t.tot_c += d.n_c; t.tot_e += d.n_e;
            t.max_nb = std::max(t.max_nb, d.n_b);
            t.max_nt = std::max(t.max_nt, d.n_t);
            t.max_runs = std::max(t.max_runs, d.n_b / BT_STEP + 1);
        }
        pfx_f[i + 1] = pfx_f[i] + (d.ok ? (d.n_f + F512_TILE - 1) / F512_TILE : 0);
        // FFT-8192 workgroups: four per 64-frame super-tile (they interleave its frames, see stft8192_kernel)
        pfx_c[i + 1] = pfx_c[i] + (d.ok ? 4 * ((d.n_c + 4 * STFT_TILE - 1) / (4 * STFT_TILE)) : 0);
        pfx_ct[i + 1] = pfx_ct[i] + (d.ok ? (d.n_c + CH_TILE - 1) / CH_TILE : 0);
        pfx_cw[i + 1] = pfx_cw[i] + (d.ok ? (d.n_c + 4 * CH_TILE - 1) / (4 * CH_TILE) : 0);
    }
    t.tiles_ct = pfx_ct[ns];
    // Tuning candidates (the peaks inside the median's coarse magnitude bins) come from one pool per chunk, handed out
    // on the device once the histogram says how many each song has; a song the pool cannot serve takes the exact
    // re-scan path of tune_final_kernel instead.
    t.cand_cap = std::min<uint64_t>(t.tot_c * (uint64_t)c->cand_budget, 0xFFFFFF00ull) + 64;

    // ---- buffers of the slot (growth frees the old block, which waits for the device) ----
    const size_t desc_bytes = align_up(ns * sizeof(SongDesc), 256) + 4 * align_up((ns + 1) * 4, 256) + 256;
    size_t need = 0;
    (void)carve(nullptr, ns, t, &need);
    if ((rc = slot.desc.ensure(desc_bytes))) return rc;
    if ((rc = slot.slab.ensure(need))) return rc;
    if (slot.used) HIP_TRY(hipEventSynchronize(slot.ev_desc));  // the slot's previous descriptor copy has left the staging area
    if ((rc = slot.h_desc.ensure(desc_bytes))) return rc;

    hipStream_t st = c->stream, sb = c->serial ? c->stream : c->aux_stream, sc = c->serial ? c->stream : c->chr_stream;
    const bool multi = !c->serial;
    // the slot's previous chunk (two chunks back) must have assembled its rows before its workspace is overwritten
    if (slot.used && multi) HIP_TRY(hipStreamWaitEvent(st, slot.ev_free, 0));

    uint8_t* h = slot.h_desc.p;
    size_t o = 0;
    const size_t o_songs = o; memcpy(h + o, songs, ns * sizeof(SongDesc)); o = align_up(o + ns * sizeof(SongDesc), 256);
    const size_t o_f = o; memcpy(h + o, pfx_f.data(), (ns + 1) * 4); o = align_up(o + (ns + 1) * 4, 256);
    const size_t o_c = o; memcpy(h + o, pfx_c.data(), (ns + 1) * 4); o = align_up(o + (ns + 1) * 4, 256);
    const size_t o_ct = o; memcpy(h + o, pfx_ct.data(), (ns + 1) * 4); o = align_up(o + (ns + 1) * 4, 256);
    const size_t o_cw = o; memcpy(h + o, pfx_cw.data(), (ns + 1) * 4); o = align_up(o + (ns + 1) * 4, 256);
    HIP_TRY(hipMemcpyAsync(slot.desc.p, h, o, hipMemcpyHostToDevice, st));
    slot.used = true;

    Batch& b = slot.batch;
    b = Batch{};
    b.pcm = d_pcm;
    b.songs = reinterpret_cast<const SongDesc*>(slot.desc.p + o_songs);
    b.n_songs = ns;
    b.pfx_f = reinterpret_cast<const uint32_t*>(slot.desc.p + o_f);
    b.pfx_c = reinterpret_cast<const uint32_t*>(slot.desc.p + o_c);
    b.pfx_ct = reinterpret_cast<const uint32_t*>(slot.desc.p + o_ct);
    b.pfx_cw = reinterpret_cast<const uint32_t*>(slot.desc.p + o_cw);
    b.tiles_f = pfx_f[ns]; b.tiles_c = pfx_c[ns]; b.tiles_ct = pfx_ct[ns]; b.tiles_cw = pfx_cw[ns];
    b.total_b = t.tot_b; b.max_nb = t.max_nb; b.max_nt = t.max_nt;

    size_t used_bytes = 0;
    slot.ws = carve(slot.slab.p, ns, t, &used_bytes);
    slot.ws.dbg_chroma = nullptr;
    slot.ws.dbg_interval = nullptr;
    if (c->debug_chroma) {  // taps of the parity tests: the buffers belong to the context, the last chunk's contents stay
        if ((rc = c->dbg_chroma.ensure((size_t)t.tot_c * 12 + 12))) return rc;
        if ((rc = c->dbg_interval.ensure((size_t)ns * 10))) return rc;
        slot.ws.dbg_chroma = c->dbg_chroma.p;
        slot.ws.dbg_interval = c->dbg_interval.p;
    }
    const Workspace& w = slot.ws;
    c->last_ws = w;
    c->last_songs.assign(songs, songs + ns);
    {
        // the exact rolloff pass's record (its entries borrow the spectrogram + peak records, see carve)
        RollFix rf{};
        // Every timbral frame of the chunk must have an entry of its own: a frame turned away would keep the unproven
        // fast-path bin without a sign.  The static_assert in carve() proves the bytes-per-sample ratio; this is the
        // per-chunk proof that the rounding of every song's frame counts (many near-minimum-length songs) does not eat it.
        if ((uint64_t)w.roll_fix_bytes / (256 * 4) < t.tot_t)
            return fail(BLISSGPU_ERR_INVALID, "chunk_front", "internal: the borrowed stretch cannot hold one exact-rolloff entry per timbral frame");
        rf.cap = (uint32_t)t.tot_t;
        rf.mags = w.spec;
        memcpy(h + o, &rf, sizeof(rf));
        HIP_TRY(hipMemcpyAsync(w.roll_fix, h + o, sizeof(rf), hipMemcpyHostToDevice, st));
    }
    HIP_TRY(hipEventRecord(slot.ev_desc, st));

    HIP_TRY(hipMemsetAsync(w.h1, 0, (size_t)ns * H1_BINS * 4, st));
    HIP_TRY(hipMemsetAsync(w.hist100, 0, (size_t)ns * N_TUNING * 4, st));
    HIP_TRY(hipMemsetAsync(w.cand_cursor, 0, 16, st));
    { Prof p(c, K_FFT512); launch_fft512(b, w, c->tables, st, c->rolloff_exact_all, c->flux_order); }
    { Prof p(c, K_ROLLFIX); launch_rolloff_fix(b, w, t.tot_t, st); }
    // tails: the reference runs them as the tempo / timbral / loudness threads of src/song/mod.rs:432-491
    if (multi) {
        HIP_TRY(hipEventRecord(slot.ev_fork, st));
        HIP_TRY(hipStreamWaitEvent(sb, slot.ev_fork, 0));
    }
    // (the peak picker heads the aux stream: only the beat tracker reads its series, and on the main stream its 70 us stood
    // between the exact-rolloff pass -- which must leave the borrowed spectrogram stretch first -- and the FFT-8192 kernel)
    { Prof p(c, K_ONSET, sb); launch_onset(b, w, sb); }
    { Prof p(c, K_SUMMARY, sb); launch_summary(b, w, sb); }
    // The beat tracker (the parallel autocorrelation kernel + the one-wavefront-per-song state machine): beside the
    // FFT-8192 kernel its workgroups displace FFT-8192 workgroups (that kernel fills the LDS and the register file:
    // 18.9 instead of 17.3 ms).  In a multi-chunk batch that is still as good a place as any (machine time is conserved;
    // 427 vs 426 ms on the 6 250-song mixed corpus) and keeps the chunk's tail short; a batch of ONE chunk runs it behind
    // the FFT-8192 kernel, where the latency-bound tuning kernels leave the machine half empty (39.5 vs 40.1 ms per 1024
    // songs).
    // BLISSGPU_OPT_TAIL_MODE = N >= 2 (round 6 experiment): the autocorrelations (parallel work) here, beside the FFT-8192
    // kernel, and the state machines -- one wavefront per song, a chain of dependent steps -- beside it as well but on a
    // stream confined to N CUs: they need residency for a long time, not throughput, so they take N CUs' worth of slots
    // from the FFT-8192 kernel instead of a share of every CU.
    slot.beat_masked = multi && c->tail_mode >= 2;
    if (slot.beat_masked) {
        if ((rc = ensure_mask_stream(c, c->tail_mode))) return rc;
        { Prof p(c, K_BEAT, sb); launch_beat_acf(b, w, c->tables, sb); }
        HIP_TRY(hipEventRecord(slot.ev_acf, sb));
        HIP_TRY(hipStreamWaitEvent(c->mask_stream, slot.ev_acf, 0));
        { Prof p(c, K_BEAT, c->mask_stream); launch_beat_track(b, w, c->tables, c->mask_stream); }
        HIP_TRY(hipEventRecord(slot.ev_beat, c->mask_stream));
    }
    // BLISSGPU_OPT_TAIL_MODE = -2 (round 6 experiment): the autocorrelations beside the FFT-8192 kernel, the state machines behind it
    const bool acf_early = multi && c->tail_mode == -2;
    if (acf_early) { Prof p(c, K_BEAT, sb); launch_beat_acf(b, w, c->tables, sb); }
    // BLISSGPU_OPT_TAIL_MODE = -3: the whole beat tracker BESIDE THE CONTRACTION (behind the tuning estimate): the contraction waits for
    // memory with its vector ALUs a fifth busy, the autocorrelations are arithmetic and the state machines latency -- and the
    // tuning chain, the only thing between the FFT-8192 kernel and the contraction, gets the machine to itself.
    const bool beat_last = multi && c->tail_mode == -3;
    const bool beat_late = !slot.beat_masked && !beat_last && (c->tail_mode == 1 || acf_early || (c->tail_mode < 0 && only_chunk));
    if (!beat_late && !slot.beat_masked && !beat_last) { Prof p(c, K_BEAT, sb); launch_beat(b, w, c->tables, sb); }
    { Prof p(c, K_STFT8192); launch_stft8192(b, w, c->tables, st, c->stft_shape); }
    if (multi) {
        HIP_TRY(hipEventRecord(slot.ev_stft, st));
        HIP_TRY(hipStreamWaitEvent(sc, slot.ev_stft, 0));
    }
    // Split tail (BLISSGPU_OPT_TAIL_SPLIT = P, one-chunk batches only): the songs are cut into P pieces of about equal
    // numbers of chroma tiles; tuning(0) -> [contraction(0) on the main stream] beside {tuning(1), beat tracker} ->
    // contraction(1) beside tuning(2) ...: only the first piece's tuning estimate stands between the FFT-8192 kernel and
    // the contraction.
    slot.pieces = 0;
    if (multi && only_chunk && c->tail_split > 1 && ns >= 2) {
        const int want = std::min<int>(std::min<int>(c->tail_split, ChunkSlot::MAX_PIECES), (int)ns);
        uint32_t s0 = 0;
        for (int k = 0; k < want && s0 < ns; k++) {
            uint32_t s1 = s0 + 1;
            const uint64_t goal = (uint64_t)pfx_ct[ns] * (uint64_t)(k + 1) / (uint64_t)want;
            while (s1 < ns && (k == want - 1 || pfx_ct[s1] < goal)) s1++;
            slot.piece[slot.pieces++] = SongRange{s0, s1, pfx_ct[s0], pfx_ct[s1], pfx_cw[s0], pfx_cw[s1]};
            s0 = s1;
        }
        if (slot.pieces < 2) slot.pieces = 0;
    }
    if (slot.pieces) {
        for (int k = 0; k < slot.pieces; k++) {
            { Prof p(c, K_TUNE_SELECT, sc); launch_tune_select(b, w, sc, &slot.piece[k]); }
            if (k == 1 && beat_late) {  // behind a tune_select, as in the unsplit schedule
                HIP_TRY(hipEventRecord(slot.ev_sel, sc));
                HIP_TRY(hipStreamWaitEvent(sb, slot.ev_sel, 0));
                Prof p(c, K_BEAT, sb);
                if (acf_early) launch_beat_track(b, w, c->tables, sb); else launch_beat(b, w, c->tables, sb);
            }
            { Prof p(c, K_TUNE_PASS2, sc); launch_tune_pass2(b, w, sc, &slot.piece[k]); }
            { Prof p(c, K_TUNE_FINAL, sc); launch_tune_final(b, w, sc, &slot.piece[k]); }
            HIP_TRY(hipEventRecord(slot.ev_piece[k], sc));
        }
        if (beat_last) {
            HIP_TRY(hipStreamWaitEvent(sb, slot.ev_piece[slot.pieces - 1], 0));
            Prof p(c, K_BEAT, sb);
            launch_beat(b, w, c->tables, sb);
        }
        HIP_TRY(hipGetLastError());
        slot.back_pending = true;
        return BLISSGPU_OK;
    }
    { Prof p(c, K_TUNE_SELECT, sc); launch_tune_select(b, w, sc); }
    if (beat_late) {
        // behind tune_select, not beside it: a kernel with resident workgroups on every CU holds that 20 us kernel (the
        // head of the critical tuning -> chroma chain) for milliseconds (kernel timeline, tests/tools/timeline.sh)
        if (multi) {
            HIP_TRY(hipEventRecord(slot.ev_sel, sc));
            HIP_TRY(hipStreamWaitEvent(sb, slot.ev_sel, 0));
        }
        Prof p(c, K_BEAT, sb);
        if (acf_early) launch_beat_track(b, w, c->tables, sb); else launch_beat(b, w, c->tables, sb);
    }
    { Prof p(c, K_TUNE_PASS2, sc); launch_tune_pass2(b, w, sc); }
    { Prof p(c, K_TUNE_FINAL, sc); launch_tune_final(b, w, sc); }
    if (multi) HIP_TRY(hipEventRecord(slot.ev_tune, sc));
    if (beat_last) {
        HIP_TRY(hipStreamWaitEvent(sb, slot.ev_tune, 0));
        Prof p(c, K_BEAT, sb);
        launch_beat(b, w, c->tables, sb);
    }
    HIP_TRY(hipGetLastError());
    slot.back_pending = true;
    return BLISSGPU_OK;
}

int chunk_back(blissgpu_ctx* c, ChunkSlot& slot, uint32_t features_version, float* d_out, int32_t* d_status) {
    if (!slot.back_pending) return BLISSGPU_OK;
    slot.back_pending = false;
    hipStream_t st = c->stream, sb = c->serial ? c->stream : c->aux_stream;
    const bool multi = !c->serial;
    if (slot.pieces) {
        for (int k = 0; k < slot.pieces; k++) {
            HIP_TRY(hipStreamWaitEvent(st, slot.ev_piece[k], 0));
            Prof p(c, K_CHROMA);
            launch_chroma(slot.batch, slot.ws, c->tables, st, &slot.piece[k]);
        }
    } else {
        if (multi) HIP_TRY(hipStreamWaitEvent(st, slot.ev_tune, 0));
        Prof p(c, K_CHROMA);
        launch_chroma(slot.batch, slot.ws, c->tables, st);
    }
    if (multi) {
        HIP_TRY(hipEventRecord(slot.ev_chroma, st));
        HIP_TRY(hipStreamWaitEvent(sb, slot.ev_chroma, 0));  // aux already holds the chunk's summaries and beat tracker
        if (slot.beat_masked) HIP_TRY(hipStreamWaitEvent(sb, slot.ev_beat, 0));  // (or the masked stream does)
    }
    { Prof p(c, K_FINALIZE, sb); launch_finalize(slot.batch, slot.ws, features_version, d_out, d_status, c->dbg_tuning.p, c->dbg_nbpms.p, sb); }
    if (multi) HIP_TRY(hipEventRecord(slot.ev_free, sb));
    HIP_TRY(hipGetLastError());
    return BLISSGPU_OK;
}

}  // namespace

namespace bg {

void scheduler_release(blissgpu_ctx* c) {
    for (ChunkSlot& s : c->slot) {
        s.slab.release(); s.desc.release(); s.h_desc.release();
        hipEvent_t evs[] = {s.ev_start, s.ev_fork, s.ev_stft, s.ev_sel, s.ev_tune, s.ev_sum, s.ev_chroma, s.ev_desc, s.ev_free, s.ev_acf, s.ev_beat};
        for (hipEvent_t e : evs)
            if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : s.ev_piece)
            if (e) (void)hipEventDestroy(e);
        s = ChunkSlot{};
    }
    HostFeed& f = c->feed;
    f.ring.reset();  // joins the staging workers, frees the slabs
    for (hipStream_t& ls : f.lane_stream)
        if (ls) { (void)hipStreamSynchronize(ls); (void)hipStreamDestroy(ls); ls = nullptr; }
    for (auto& row : f.ev_lane)
        for (hipEvent_t& ev : row)
            if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
    for (auto& kv : c->swr_banks)
        if (kv.second.d_bank) (void)hipFree(kv.second.d_bank);
    c->swr_banks.clear();
    c->swr_bytes = 0;
    for (hipStream_t& cs : f.copy_stream)
        if (cs) { (void)hipStreamSynchronize(cs); (void)hipStreamDestroy(cs); cs = nullptr; }
    f.h_rows.release();
    for (int b = 0; b < N_FEED_BUFFERS; b++) {
        f.pcm[b].release(); f.raw[b].release(); f.out[b].release();
        for (hipEvent_t& ev : f.ev_copied[b])
            if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
        if (f.ev_done[b]) (void)hipEventDestroy(f.ev_done[b]);
        f.ev_done[b] = nullptr;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Host PCM feed.  Songs are packed group by group into one of TWO device PCM buffers; the H2D copies
// of group g + 1 run on the copy stream while group g is analysed, so the transfer -- the real bottleneck of these
// entry points (a 3-minute song is 15.9 MB as f32, 7.9 MB as s16) -- is never idle.  s16 samples are widened on the
// device (sample / 32768 = FFmpeg's s16 -> flt, src/song/decoder/ffmpeg.rs:36-109) and interleaved channels are
// downmixed there ((L + R) * SQRT_2 / 2 for stereo, the channel mean otherwise: src/song/decoder/symphonia.rs:266-300).
// All staging lives in the context and is reused by later calls.
// ------------------------------------------------------------------------------------------------------------------
#ifndef FEED_GROUP_MIB
#define FEED_GROUP_MIB 512
#endif
#ifndef FEED_GROUP_PCM_FACTOR
#define FEED_GROUP_PCM_FACTOR 1
#endif
#ifndef STAGE_MIN_BYTES
#define STAGE_MIN_BYTES (8ull << 20)
#endif
// the filter bank of one input rate, built once per context (resample.hpp) and kept on the device
int resample_bank(blissgpu_ctx* c, uint32_t rate, const ResampleBank** out, const char* who) {
    auto it = c->swr_banks.find(rate);
    if (it == c->swr_banks.end()) {
        ResampleBank rb;
        if (rate > MAX_SAMPLE_RATE || !swr_make_plan(rate, &rb.plan))
            return fail(BLISSGPU_ERR_INVALID, who, "sample_rate must be 1 .. 768000 Hz");
        std::vector<float> bank;
        swr_make_filter(rb.plan, bank);
        rb.bytes = bank.size() * sizeof(float);
        // room in the cache: the least recently used banks go (hipFree waits for the device, so a kernel still reading one has
        // finished); the bank being added always stays
        while (!c->swr_banks.empty() && c->swr_bytes + rb.bytes > (size_t)SWR_CACHE_BYTES) {
            auto lru = c->swr_banks.begin();
            for (auto q = c->swr_banks.begin(); q != c->swr_banks.end(); ++q)
                if (q->second.last_use < lru->second.last_use) lru = q;
            if (lru->second.d_bank) (void)hipFree(lru->second.d_bank);
            c->swr_bytes -= lru->second.bytes;
            c->swr_banks.erase(lru);
        }
        float* d = nullptr;
        hipError_t e = hipMalloc((void**)&d, rb.bytes);
        if (e != hipSuccess) { (void)hipGetLastError(); return fail(BLISSGPU_ERR_OOM, "hipMalloc(resample bank)", hipGetErrorString(e)); }
        e = hipMemcpy(d, bank.data(), rb.bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(d); return fail(BLISSGPU_ERR_HIP, "hipMemcpy(resample bank)", hipGetErrorString(e)); }
        rb.d_bank = d;
        c->swr_bytes += rb.bytes;
        it = c->swr_banks.emplace(rate, rb).first;
    }
    it->second.last_use = ++c->swr_clock;
    *out = &it->second;
    return BLISSGPU_OK;
}

std::vector<int> parse_cpulist(const char* text) {
    std::vector<int> cpus;
    for (const char* p = text; p && *p;) {
        char* end = nullptr;
        const long a = strtol(p, &end, 10);
        if (end == p || a < 0) break;
        long b = a;
        if (*end == '-') {
            const char* q = end + 1;
            b = strtol(q, &end, 10);
            if (end == q || b < a) break;
        }
        for (long v = a; v <= b && cpus.size() < 4096; v++) cpus.push_back((int)v);
        p = *end == ',' ? end + 1 : end;
        if (*end != ',') break;
    }
    return cpus;
}

std::vector<int> device_local_cpus(int device) {
    char bus[32] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) { (void)hipGetLastError(); return {}; }
    for (char* p = bus; *p; p++) *p = (char)tolower((unsigned char)*p);
    const std::string base = std::string("/sys/bus/pci/devices/") + bus;
    char line[4096] = {0};
    int node = -1;
    if (FILE* f = fopen((base + "/numa_node").c_str(), "r")) {
        if (fgets(line, sizeof(line), f)) node = atoi(line);
        fclose(f);
    }
    if (node < 0) return {};  // one node, or the platform does not say: leave the workers alone
    std::vector<int> cpus;
    if (FILE* f = fopen((base + "/local_cpulist").c_str(), "r")) {
        if (fgets(line, sizeof(line), f)) cpus = parse_cpulist(line);
        fclose(f);
    }
    return cpus;
}

// decoder output of one song (device) -> mono 22 050 Hz f32 (device), on `st`
int enqueue_decode(blissgpu_ctx* c, const void* d_in, int fmt, uint32_t channels, uint64_t frames, uint32_t rate, float* d_out,
                   uint64_t n_out, hipStream_t st, const char* who) {
    if (rate == SWR_OUT_RATE) {
        launch_pcm_convert(d_in, fmt, channels, d_out, frames, st);
        HIP_TRY(hipGetLastError());
        return BLISSGPU_OK;
    }
    const ResampleBank* rb;
    const int rc = resample_bank(c, rate, &rb, who);
    if (rc) return rc;
    HIP_TRY(launch_resample(d_in, fmt, channels, frames, rb->plan, rb->d_bank, d_out, n_out, st));
    return BLISSGPU_OK;
}

int analyze_host_songs(blissgpu_ctx* c, const FeedSong* in, uint32_t n_songs, uint32_t features_version, float* out,
                       int32_t* status, const char* who, float* d_rows) {
    const uint32_t d = blissgpu_feature_count(features_version);
    if (!d) return fail(BLISSGPU_ERR_INVALID, who, "features_version must be 1 or 2");
    if (n_songs == 0) return BLISSGPU_OK;
    // what every song becomes: its mono 22 050 Hz length, and whether it needs the device conversion at all
    std::vector<uint64_t> out_len(n_songs);
    bool any_raw = false;
    for (uint32_t i = 0; i < n_songs; i++) {
        const FeedSong& fs = in[i];
        if (fs.channels == 0 || fs.channels > 8) return fail(BLISSGPU_ERR_INVALID, who, "channels must be 1..8");
        if (fs.fmt != BLISSGPU_SAMPLE_F32 && fs.fmt != BLISSGPU_SAMPLE_S16 && fs.fmt != BLISSGPU_SAMPLE_S32)
            return fail(BLISSGPU_ERR_INVALID, who, "sample_format must be F32, S16 or S32");
        if (fs.rate == 0 || fs.rate > MAX_SAMPLE_RATE) return fail(BLISSGPU_ERR_INVALID, who, "sample_rate must be 1 .. 768000 Hz");
        if (fs.frames && !fs.p) return fail(BLISSGPU_ERR_INVALID, who, "NULL song pointer");
        if (fs.rate == SWR_OUT_RATE) out_len[i] = fs.frames;
        else {
            SwrPlan pl;
            swr_make_plan(fs.rate, &pl);
            out_len[i] = swr_out_len(pl, fs.frames);
        }
        any_raw = any_raw || !fs.direct();
    }
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    HostFeed& f = c->feed;
    // the filter banks of this call's rates are built (a blocking allocation + upload each, once per context and rate) BEFORE the
    // first transfer is queued, not in the middle of the pipeline
    {
        uint32_t last_rate = SWR_OUT_RATE;
        for (uint32_t i = 0; i < n_songs; i++)
            if (in[i].rate != SWR_OUT_RATE && in[i].rate != last_rate && in[i].frames) {
                const ResampleBank* rb;
                const int brc = resample_bank(c, in[i].rate, &rb, who);
                if (brc) return brc;
                last_rate = in[i].rate;
            }
    }
    if (!f.copy_stream[N_COPY_STREAMS - 1]) {
        for (hipStream_t& cs : f.copy_stream)
            if (!cs) HIP_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        for (int b = 0; b < N_FEED_BUFFERS; b++) {
            for (hipEvent_t& ev : f.ev_copied[b])
                if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            if (!f.ev_done[b]) HIP_TRY(hipEventCreateWithFlags(&f.ev_done[b], hipEventDisableTiming));
        }
    }
    // Which sources are pageable?  Page-locked / registered / device memory goes to the copy streams as it is (the DMA engines
    // read it directly); ordinary heap memory -- a Rust Vec<f32>, a decoder's frame buffer, a numpy array -- is staged through
    // the ring's page-locked slabs by its worker threads (staging_ring.hpp).  A call that brings less than STAGE_MIN_BYTES of it
    // is left to the runtime: a short song is one bounce buffer, and the single-song front should not wake six threads for it.
    std::vector<uint8_t> pageable(n_songs, 0);
    bool use_ring = false;
    if (f.stage_cfg.lanes > 0) {
        uint64_t pageable_bytes = 0;
        for (uint32_t i = 0; i < n_songs; i++) {
            if (!in[i].frames) continue;
            hipPointerAttribute_t at;
            const hipError_t pe = hipPointerGetAttributes(&at, in[i].p);
            if (pe != hipSuccess) (void)hipGetLastError();  // (older runtimes report an unregistered pointer as an error)
            pageable[i] = pe != hipSuccess || at.type == hipMemoryTypeUnregistered;
            if (pageable[i]) pageable_bytes += in[i].frames * in[i].frame_bytes();
        }
        use_ring = pageable_bytes >= STAGE_MIN_BYTES;
    }
    if (use_ring) {
        StageConfig want = f.stage_cfg;
        want.lanes = std::min(want.lanes, MAX_STAGE_LANES);
        for (int l = 0; l < want.lanes; l++) {
            if (!f.lane_stream[l]) HIP_TRY(hipStreamCreateWithFlags(&f.lane_stream[l], hipStreamNonBlocking));
            for (int b = 0; b < N_FEED_BUFFERS; b++)
                if (!f.ev_lane[b][l]) HIP_TRY(hipEventCreateWithFlags(&f.ev_lane[b][l], hipEventDisableTiming));
        }
        if (!f.ring) f.ring.reset(new StagingRing<HipStageDev>(HipStageDev{c->device, f.lane_stream, {}}));
        if (!f.ring->running() || !(f.ring->config() == want) || f.stage_numa_known != f.stage_numa) {
            f.ring->stop();
            f.ring->dev().cpus = f.stage_numa ? device_local_cpus(c->device) : std::vector<int>{};
            f.stage_numa_known = f.stage_numa;
            std::string why;
            f.staged_bytes += f.ring->bytes_staged();
            if (!f.ring->start(want, &why)) use_ring = false;  // no page-locked memory to be had: the runtime's own staging still works
        }
    }
    if (use_ring) f.staged_calls++;
    else std::fill(pageable.begin(), pageable.end(), 0);
    const int n_lanes = use_ring ? f.ring->config().lanes : 0;
    const size_t slab_bytes = use_ring ? f.ring->config().slab_bytes : 0;

    // groups of <= 512 MiB of staging (mono f32: ~32 three-minute songs).  The transfer is the bottleneck (a group's
    // analysis takes a tenth of its transfer time), so what a call pays beyond its bytes is the analysis of the LAST group:
    // small groups keep that tail short, and 32 songs still fill the GPU several times over.  The cap is in BYTES of the
    // wider of the two buffers of a group (raw decoder output / mono f32), so an 8-channel f32 batch stages
    // 2 x 512 MiB like a mono one instead of 2 x 4 GiB.
    const uint64_t group_cap = (uint64_t)FEED_GROUP_MIB << 20;  // bytes
    struct Group { uint32_t i0, n; std::vector<uint64_t> doff, dlen, roff; uint64_t total, raw_total, link_total; };
    std::vector<Group> groups;
    for (uint32_t i0 = 0; i0 < n_songs;) {
        Group g{i0, 0, {}, {}, {}, 0, 0, 0};
        uint32_t i1 = i0;
        while (i1 < n_songs) {
            const uint64_t raw = in[i1].direct() ? 0 : (in[i1].frames + 63) / 64 * 64 * in[i1].frame_bytes();  // padded like the PCM
            const uint64_t pcm = (out_len[i1] + 63) / 64 * 64;
            const uint64_t link = in[i1].direct() ? 4 * pcm : raw;  // what crosses the link for this song
            // (FEED_GROUP_PCM_FACTOR = 2 lets a group of mono s16 songs be as long on the LINK as a group of f32 songs -- 66
            // songs, 10 ms -- instead of 33 songs whose 5 ms transfer barely covers the 4 - 5 ms a batch that small takes to
            // analyse, profiles/r06_feed_trace_s16.txt.  Measured: no gain from pageable memory and a longer tail from
            // page-locked memory, profiles/r06_feed_group_rule_ab.txt; the cap stays on the wider buffer.)
            if (i1 > i0 && (g.link_total + link > group_cap || 4 * (g.total + pcm) > FEED_GROUP_PCM_FACTOR * group_cap || g.raw_total + raw > group_cap)) break;
            g.doff.push_back(g.total);
            g.dlen.push_back(out_len[i1]);
            g.roff.push_back(g.raw_total);
            g.total += pcm;
            g.raw_total += raw;
            g.link_total += link;
            i1++;
        }
        g.n = i1 - i0;
        groups.push_back(std::move(g));
        i0 = i1;
    }
    uint64_t max_total = 64, max_raw = 0, max_n = 1;
    for (const auto& g : groups) {
        max_total = std::max(max_total, g.total);
        max_raw = std::max(max_raw, g.raw_total);
        max_n = std::max<uint64_t>(max_n, g.n);
    }
    const int nbuf = (int)std::min<size_t>(groups.size(), (size_t)N_FEED_BUFFERS);
    int rc = BLISSGPU_OK;
    for (int b = 0; b < nbuf && !rc; b++) {
        rc = f.pcm[b].ensure(max_total);
        if (!rc && any_raw) rc = f.raw[b].ensure(max_raw + 256);
        if (!rc) rc = f.out[b].ensure(max_n * d);
    }
#ifdef FEED_ROWS_STAGED
    if (!rc) rc = f.h_rows.ensure((size_t)n_songs * d);
#endif
    if (rc) return rc;

    std::vector<uint64_t> ticket(groups.size(), 0);
    auto upload = [&](size_t gi) -> hipError_t {  // H2D of group gi into buffer gi % nbuf
        const Group& g = groups[gi];
        const int b = (int)(gi % nbuf);
        // buffer b is free again once the analysis that last read it (this call's group gi - 2, or an earlier call:
        // every call ends synchronised) has finished
        const bool wait_free = gi >= (size_t)nbuf;
        hipError_t ee = hipSuccess;
        for (int q = 0; q < N_COPY_STREAMS && ee == hipSuccess && wait_free; q++)
            ee = hipStreamWaitEvent(f.copy_stream[q], f.ev_done[b], 0);
        std::vector<StagePiece> pieces;
        for (uint32_t k = 0; k < g.n && ee == hipSuccess; k++) {
            const FeedSong& fs = in[g.i0 + k];
            if (!fs.frames) continue;
            uint8_t* dst = fs.direct() ? (uint8_t*)(f.pcm[b].p + g.doff[k]) : f.raw[b].p + g.roff[k];
            const size_t bytes = fs.frames * fs.frame_bytes();
            if (pageable[g.i0 + k]) {  // through the ring, a slab at a time, on the workers' lanes
                for (size_t off = 0; off < bytes; off += slab_bytes)
                    pieces.push_back(StagePiece{(const uint8_t*)fs.p + off, dst + off, std::min(slab_bytes, bytes - off)});
            } else {
                ee = hipMemcpyAsync(dst, fs.p, bytes, hipMemcpyHostToDevice, f.copy_stream[k % N_COPY_STREAMS]);
            }
        }
        for (int q = 0; q < N_COPY_STREAMS && ee == hipSuccess; q++) ee = hipEventRecord(f.ev_copied[b][q], f.copy_stream[q]);
        if (use_ring && ee == hipSuccess) {
            hipEvent_t ev_free = f.ev_done[b];
            hipStream_t* ls = f.lane_stream;
            hipEvent_t* ev_in = f.ev_lane[b];
            ticket[gi] = f.ring->post(
                std::move(pieces),
                [=](int lane) { return wait_free ? (int)hipStreamWaitEvent(ls[lane], ev_free, 0) : 0; },
                [=](int lane) { return (int)hipEventRecord(ev_in[lane], ls[lane]); });
        }
        return ee;
    };
    // every exit: no worker still reads the caller's memory, no lane still writes the staging buffers
    auto settle = [&]() -> hipError_t {
        hipError_t es = hipSuccess;
        if (use_ring) {
            std::string why;
            const int re = f.ring->drain(&why);
            if (re) es = (hipError_t)re;
            for (int l = 0; l < n_lanes; l++) {
                const hipError_t eq = hipStreamSynchronize(f.lane_stream[l]);
                if (es == hipSuccess) es = eq;
            }
        }
        return es;
    };

    // BLISSGPU_FEED_TRACE=1 (developer aid): the host-side timeline of the call, group by group, on stderr
    static const bool trace = getenv("BLISSGPU_FEED_TRACE") != nullptr;
    const auto t_call = std::chrono::steady_clock::now();
    auto ms_now = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count(); };
    hipError_t e = hipSuccess;
    size_t posted = 0;
    for (size_t gi = 0; gi < groups.size() && e == hipSuccess && !rc; gi++) {
        const double t_iter = ms_now();
        const Group& g = groups[gi];
        const int b = (int)(gi % nbuf);
        // The transfers of the next nbuf - 1 groups are queued BEFORE this group's analysis is enqueued (the buffer of group
        // gi + nbuf - 1 was released by the event group gi - 1 recorded): enqueuing an analysis can block the host on an
        // earlier chunk's events, and the link must not run dry meanwhile.
        while (posted < groups.size() && posted < gi + (size_t)nbuf && e == hipSuccess) e = upload(posted++);
        for (int q = 0; q < N_COPY_STREAMS && e == hipSuccess; q++) e = hipStreamWaitEvent(c->stream, f.ev_copied[b][q], 0);
        if (use_ring && e == hipSuccess) {
            // the workers have queued every slab of this group on their lanes (they are already filling the next group's)
            std::string why;
            const int re = f.ring->wait_enqueued(ticket[gi], &why);
            if (re) { (void)settle(); (void)hipStreamSynchronize(c->stream); return fail(BLISSGPU_ERR_HIP, who, why.c_str()); }
            for (int l = 0; l < n_lanes && e == hipSuccess; l++) e = hipStreamWaitEvent(c->stream, f.ev_lane[b][l], 0);
        }
        const double t_staged = ms_now();
        if (e != hipSuccess) break;
        // widening / downmix / resampling on the device.  A group of one format at 22 050 Hz (the bulk case: s16 from the
        // decoder) is ONE launch over the whole staging area -- its songs are packed with the same 64-frame padding in both
        // buffers; anything else goes song by song (a launch per song is nothing beside the song's transfer)
        bool uniform = true;
        for (uint32_t k = 0; k < g.n && uniform; k++) {
            const FeedSong &a0 = in[g.i0], &ak = in[g.i0 + k];
            uniform = !ak.direct() && ak.rate == SWR_OUT_RATE && ak.fmt == a0.fmt && ak.channels == a0.channels &&
                      g.roff[k] == g.doff[k] * ak.frame_bytes();
        }
        if (uniform) {
            launch_pcm_convert(f.raw[b].p, in[g.i0].fmt, in[g.i0].channels, f.pcm[b].p, g.total, c->stream);
            e = hipGetLastError();
        } else {
            for (uint32_t k = 0; k < g.n && !rc; k++) {
                const FeedSong& fs = in[g.i0 + k];
                if (fs.direct() || fs.frames == 0) continue;
                rc = enqueue_decode(c, f.raw[b].p + g.roff[k], fs.fmt, fs.channels, fs.frames, fs.rate, f.pcm[b].p + g.doff[k],
                                    g.dlen[k], c->stream, who);
            }
        }
        if (e != hipSuccess || rc) break;
        rc = blissgpu_analyze_batch_device(c, f.pcm[b].p, g.doff.data(), g.dlen.data(), g.n, features_version, f.out[b].p, nullptr);
        if (rc) break;
        const double t_enq = ms_now();
        // (The rows go straight into the caller's array.  Through a page-locked buffer of the context the calling thread would
        // not be held here until the group's analysis has finished -- and the feed measured 5 - 10 % SLOWER that way:
        // FEED_ROWS_STAGED, profiles/r06_feed_rows_ab.txt.)
#ifdef FEED_ROWS_STAGED
        e = hipMemcpyAsync(f.h_rows.p + (size_t)g.i0 * d, f.out[b].p, (size_t)g.n * d * sizeof(float), hipMemcpyDeviceToHost, c->stream);
#else
        e = hipMemcpyAsync(out + (size_t)g.i0 * d, f.out[b].p, (size_t)g.n * d * sizeof(float), hipMemcpyDeviceToHost, c->stream);
#endif
        if (e == hipSuccess && d_rows)
            e = hipMemcpyAsync(d_rows + (size_t)g.i0 * d, f.out[b].p, (size_t)g.n * d * sizeof(float), hipMemcpyDeviceToDevice, c->stream);
        if (e == hipSuccess) e = hipEventRecord(f.ev_done[b], c->stream);
        if (trace)
            fprintf(stderr, "feed group %zu/%zu (%u songs, buffer %d): iteration at %.2f ms, staged %.2f, analysis enqueued %.2f, rows requested %.2f\n",
                    gi, groups.size(), g.n, b, t_iter, t_staged, t_enq, ms_now());
        if (status)
            for (uint32_t k = 0; k < g.n; k++)
                status[g.i0 + k] = g.dlen[k] >= (uint64_t)MIN_SAMPLES ? BLISSGPU_SONG_OK : BLISSGPU_SONG_TOO_SHORT;
    }
    // every exit leaves the streams drained: the staging buffers belong to the next call
    hipError_t e1 = settle();
    for (hipStream_t cs : f.copy_stream) {
        const hipError_t eq = hipStreamSynchronize(cs);
        if (e1 == hipSuccess) e1 = eq;
    }
    const hipError_t e2 = hipStreamSynchronize(c->stream);
    if (trace) fprintf(stderr, "feed call done at %.2f ms (%s)\n", ms_now(), use_ring ? "ring" : "direct");
    if (rc) return rc;
    if (e == hipSuccess) e = e1 != hipSuccess ? e1 : e2;
    if (e != hipSuccess) return fail(BLISSGPU_ERR_HIP, who, hipGetErrorString(e));
#ifdef FEED_ROWS_STAGED
    memcpy(out, f.h_rows.p, (size_t)n_songs * d * sizeof(float));
#endif
    return BLISSGPU_OK;
}

// the uniform form (every song one format, channel count and rate)
int analyze_host_songs(blissgpu_ctx* c, const void* const* ptrs, const uint64_t* lengths, uint32_t n_songs, int sample_format,
                       uint32_t channels, uint32_t features_version, float* out, int32_t* status, const char* who,
                       float* d_rows, uint32_t sample_rate) {
    std::vector<FeedSong> songs(n_songs);
    for (uint32_t i = 0; i < n_songs; i++)
        songs[i] = FeedSong{ptrs[i], lengths[i], sample_rate, (uint8_t)sample_format, (uint8_t)(channels > 255 ? 255 : channels)};
    return analyze_host_songs(c, songs.data(), n_songs, features_version, out, status, who, d_rows);
}

}  // namespace bg

// ------------------------------------------------------------------------------------------------------------------
// Coalescing front of the single-song entry points.  The reference's bulk path is N worker threads each calling
// Song::analyze on its own song (src/song/decoder.rs:299-329); one song cannot fill the GPU, a batch can.  A caller
// queues its request; whoever finds no batch running becomes the leader and analyses EVERYTHING that has queued up as
// one batch (group commit), the others sleep until their request is done.  A lone caller pays no waiting window; under
// load the batch size adapts to the arrival rate.
// ------------------------------------------------------------------------------------------------------------------
namespace {

struct AnalyzeReq {
    const void* pcm;
    uint64_t frames;
    int sample_format;
    uint32_t channels, sample_rate, version;
    float* out;
    int32_t status = BLISSGPU_SONG_OK;
    int rc = BLISSGPU_OK;
    std::string err;
    bool done = false;
    int front_outcome = 0;  // bg::FrontOutcome
};

// the calling thread's current HIP device is the caller's business: a leader works on its seat's device and puts the
// caller's device back (a caller's later hipMalloc / torch allocation must not land on whatever GPU served the batch)
struct DeviceRestore {
    int prev = -1;
    DeviceRestore() { if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); } }
    ~DeviceRestore() { if (prev >= 0) (void)hipSetDevice(prev); }
};

std::mutex g_seat_err_mu;
std::string g_seat_err;  // why the last seat was retired (reported when no seat is left)

// The queue, the seats and the waiting are in coalescing_front.hpp (device-free: tests/cpp/test_front.cpp drives it on the
// CPU); a leader's batch runs on the default context of its seat.
bg::CoalescingFront<AnalyzeReq> g_front;

// true: the batch was dealt with (every request carries its result); false: the seat's device cannot give a context --
// nothing was done, the front retires the seat and hands the batch to another one
bool run_batch(std::vector<AnalyzeReq*>& take, int seat, const char* who) {
    DeviceRestore restore;
    blissgpu_ctx* c = nullptr;
    const int rc0 = default_ctx_at(seat, &c);
    if (rc0) {
        std::lock_guard<std::mutex> lk(g_seat_err_mu);
        g_seat_err = blissgpu_last_error();
        return false;
    }
    // one device batch per features version (the songs of a batch may differ in format, channel count and rate; in practice
    // there is one version)
    std::vector<char> served(take.size(), 0);
    for (size_t a = 0; a < take.size(); a++) {
        if (served[a]) continue;
        std::vector<size_t> cls;
        for (size_t q = a; q < take.size(); q++)
            if (!served[q] && take[q]->version == take[a]->version) {
                cls.push_back(q);
                served[q] = 1;
            }
        const uint32_t d = blissgpu_feature_count(take[a]->version);
        std::vector<bg::FeedSong> songs(cls.size());
        std::vector<int32_t> st(cls.size(), 0);
        std::vector<float> rows(cls.size() * (size_t)std::max(d, 1u));
        for (size_t q = 0; q < cls.size(); q++) {
            const AnalyzeReq* t = take[cls[q]];
            songs[q] = bg::FeedSong{t->pcm, t->frames, t->sample_rate, (uint8_t)t->sample_format, (uint8_t)t->channels};
        }
        const int rc = analyze_host_songs(c, songs.data(), (uint32_t)cls.size(), take[a]->version, rows.data(), st.data(), who);
        const std::string err = rc ? blissgpu_last_error() : "";
        for (size_t q = 0; q < cls.size(); q++) {
            AnalyzeReq* t = take[cls[q]];
            t->rc = rc;
            t->err = err;
            t->status = st[q];
            if (!rc) memcpy(t->out, rows.data() + q * d, d * sizeof(float));
        }
    }
    default_ctx_count_batch(seat);
    return true;
}

int submit(AnalyzeReq& r, const char* who) {
    const bg::FrontOutcome o = g_front.submit(
        r, default_ctx_count(),
        [&](std::vector<AnalyzeReq*>& take, int seat) -> bool {
            // a failed allocation while gathering the batch becomes an error code on every request of the batch
            try {
                return run_batch(take, seat, who);
            } catch (...) {
                for (AnalyzeReq* t : take) { t->rc = BLISSGPU_ERR_OOM; t->err = "out of host memory while gathering the batch"; }
                return true;
            }
        },
        std::chrono::milliseconds(single_song_timeout_ms()));
    if (o == bg::FRONT_NO_SEAT) {
        std::string why;
        { std::lock_guard<std::mutex> lk(g_seat_err_mu); why = g_seat_err; }
        return fail(BLISSGPU_ERR_NO_DEVICE, who, ("no usable default context (" + g_front.describe() + "); last failure: " + why).c_str());
    }
    if (o == bg::FRONT_TIMED_OUT)
        return fail(BLISSGPU_ERR_TIMEOUT, who,
                    ("not picked up by a default context within " + std::to_string(single_song_timeout_ms()) + " ms: " + g_front.describe()).c_str());
    if (r.rc) return fail(r.rc, who, r.err.c_str());
    return BLISSGPU_OK;
}

bool known_format(int sample_format) {
    return sample_format == BLISSGPU_SAMPLE_F32 || sample_format == BLISSGPU_SAMPLE_S16 || sample_format == BLISSGPU_SAMPLE_S32;
}

int analyze_one(const void* pcm, uint64_t frames, int sample_format, uint32_t channels, uint32_t sample_rate, uint32_t version,
                float* out, int32_t* status, const char* who) {
    if (!out || (frames && !pcm)) return fail(BLISSGPU_ERR_INVALID, who, "NULL argument");
    if (!blissgpu_feature_count(version)) return fail(BLISSGPU_ERR_INVALID, who, "features_version must be 1 or 2");
    if (channels == 0 || channels > 8) return fail(BLISSGPU_ERR_INVALID, who, "channels must be 1..8");
    if (sample_rate == 0 || sample_rate > bg::MAX_SAMPLE_RATE) return fail(BLISSGPU_ERR_INVALID, who, "sample_rate must be 1 .. 768000 Hz");
    AnalyzeReq r{pcm, frames, sample_format, channels, sample_rate, version, out};
    const int rc = submit(r, who);
    if (status) *status = r.status;
    return rc;
}

int batch_from_offsets(const void* pcm, size_t frame_bytes, const uint64_t* offsets, const uint64_t* lengths, uint32_t n_songs,
                       int sample_format, uint32_t channels, uint32_t version, float* out, int32_t* status, const char* who) {
    if (n_songs && (!pcm || !offsets || !lengths || !out)) return fail(BLISSGPU_ERR_INVALID, who, "NULL argument");
    if (!blissgpu_feature_count(version)) return fail(BLISSGPU_ERR_INVALID, who, "features_version must be 1 or 2");
    if (channels == 0 || channels > 8) return fail(BLISSGPU_ERR_INVALID, who, "channels must be 1..8");
    if (n_songs == 0) return BLISSGPU_OK;
    DeviceRestore restore;
    blissgpu_ctx* c;
    int rc = default_ctx(&c);
    if (rc) return rc;
    std::vector<const void*> ptrs(n_songs);
    for (uint32_t i = 0; i < n_songs; i++) ptrs[i] = (const uint8_t*)pcm + offsets[i] * frame_bytes;
    return analyze_host_songs(c, ptrs.data(), lengths, n_songs, sample_format, channels, version, out, status, who);
}

}  // namespace

namespace bg {
void front_revive_all() { g_front.revive_all(); }
}  // namespace bg

extern "C" {

int blissgpu_analyze_batch_device(blissgpu_ctx* c, const float* d_pcm, const uint64_t* offsets, const uint64_t* lengths,
                                  uint32_t n_songs, uint32_t features_version, float* d_out, int32_t* d_status) {
    if (!c || (n_songs && (!d_pcm || !offsets || !lengths || !d_out)))
        return fail(BLISSGPU_ERR_INVALID, "blissgpu_analyze_batch_device", "NULL argument");
    if (features_version != BLISSGPU_FEATURES_V1 && features_version != BLISSGPU_FEATURES_V2)
        return fail(BLISSGPU_ERR_INVALID, "blissgpu_analyze_batch_device", "features_version must be 1 or 2");
    if (n_songs == 0) return BLISSGPU_OK;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    if ((rc = c->dbg_tuning.ensure(n_songs))) return rc;
    if ((rc = c->dbg_nbpms.ensure(n_songs))) return rc;
    c->dbg_n = n_songs;

    // ---- plan: descriptors in length order (longest first).  Songs of one chunk then have similar lengths, so the
    // 64 lanes of a summary wavefront and the beat-tracker workgroups of a chunk finish together, and the longest
    // tails start first.  Equal lengths keep the caller's order (stable). ----
    std::vector<SongDesc> songs(n_songs);
    for (uint32_t i = 0; i < n_songs; i++) {
        SongDesc& d = songs[i];
        d = SongDesc{};
        d.pcm_off = offsets[i];
        d.n = lengths[i];
        d.row = i;
        d.ok = lengths[i] >= (uint64_t)MIN_SAMPLES;  // src/song/mod.rs:417-430
        if (d.ok) fill_counts(d);
    }
    std::stable_sort(songs.begin(), songs.end(), [](const SongDesc& a, const SongDesc& b) { return a.n > b.n; });
    // ---- chunks: as many songs as fit one slot's workspace -- and at least PIPELINE_CHUNKS of them when the batch is big
    // enough to fill the GPU several times over, so that the software pipeline below has something to overlap ----
    struct Range { uint32_t b, e; };
    std::vector<Range> todo;
    {
        size_t total_bytes = 0;
        for (uint32_t i = 0; i < n_songs; i++) total_bytes += song_ws_bytes(songs[i]);
        size_t limit = c->ws_limit;
        if (!c->serial && c->pipeline_chunks > 1 && total_bytes / c->pipeline_chunks >= (size_t)PIPELINE_MIN_CHUNK_BYTES)
            limit = std::min<size_t>(limit, total_bytes / c->pipeline_chunks + 1);
        size_t bytes = 0;
        uint32_t b0 = 0;
        for (uint32_t i = 0; i < n_songs; i++) {
            const size_t sb = song_ws_bytes(songs[i]);
            // a chunk also never holds more songs than one grid dimension addresses (the beat tracker's (run, song) grid)
            if (i > b0 && (bytes + sb > limit || i - b0 >= MAX_SONGS_PER_CHUNK)) { todo.push_back({b0, i}); b0 = i; bytes = 0; }
            bytes += sb;
        }
        todo.push_back({b0, n_songs});
    }
    // the chroma / interval taps are ONE buffer per context, sized per chunk: a second chunk's front half would resize it
    // under the first chunk's back half
    if (c->debug_chroma && todo.size() > 1)
        return fail(BLISSGPU_ERR_INVALID, "blissgpu_analyze_batch_device",
                    "BLISSGPU_OPT_DEBUG_CHROMA needs a batch that fits one chunk (raise the workspace limit or analyse fewer songs)");
    std::reverse(todo.begin(), todo.end());  // used as a stack
    uint64_t chunks = 0;
    ChunkSlot* prev = nullptr;  // the chunk whose back half is still to be enqueued
    while (!todo.empty()) {
        const Range r = todo.back();
        todo.pop_back();
        // Every call starts with slot 0, so a batch that fits one chunk never allocates the second slot.  The slot's own
        // back half (chunk k - 2, or the previous call's) is always enqueued by now; its front may only follow it.
        ChunkSlot& slot = c->slot[chunks & 1];
        rc = chunk_front(c, slot, d_pcm, songs.data() + r.b, r.e - r.b, chunks == 0 && todo.empty());
        if (rc == BLISSGPU_ERR_OOM && r.e - r.b > 1) {
            // the device has less free memory than the limit assumed (another process, the caller's own tensors):
            // halve the chunk and go on
            (void)hipGetLastError();
            const uint32_t mid = r.b + (r.e - r.b) / 2;
            todo.push_back({mid, r.e});
            todo.push_back({r.b, mid});
            continue;
        }
        if (rc) return rc;
        c->chunk_seq++;
        chunks++;
        if (prev && (rc = chunk_back(c, *prev, features_version, d_out, d_status))) return rc;
        prev = &slot;
        if (c->serial) { if ((rc = chunk_back(c, slot, features_version, d_out, d_status))) return rc; prev = nullptr; }
    }
    if (prev && (rc = chunk_back(c, *prev, features_version, d_out, d_status))) return rc;
    c->last_chunks = chunks;
    // the caller-visible stream has "done" everything once the tails of the (at most two) chunks in flight are in
    if (!c->serial)
        for (ChunkSlot& s : c->slot)
            if (s.used) HIP_TRY(hipStreamWaitEvent(c->stream, s.ev_free, 0));
    return BLISSGPU_OK;
}

int blissgpu_analyze_batch(const float* pcm, const uint64_t* offsets, const uint64_t* lengths, uint32_t n_songs,
                           uint32_t features_version, float* out, int32_t* status) {
    return batch_from_offsets(pcm, 4, offsets, lengths, n_songs, BLISSGPU_SAMPLE_F32, 1, features_version, out, status,
                              "blissgpu_analyze_batch");
}

int blissgpu_analyze_batch_s16(const int16_t* pcm, const uint64_t* offsets, const uint64_t* lengths, uint32_t n_songs,
                               uint32_t features_version, float* out, int32_t* status) {
    return batch_from_offsets(pcm, 2, offsets, lengths, n_songs, BLISSGPU_SAMPLE_S16, 1, features_version, out, status,
                              "blissgpu_analyze_batch_s16");
}

int blissgpu_analyze_batch_interleaved(const void* pcm, int sample_format, uint32_t channels, const uint64_t* offsets,
                                       const uint64_t* lengths, uint32_t n_songs, uint32_t features_version, float* out,
                                       int32_t* status) {
    if (!known_format(sample_format))
        return fail(BLISSGPU_ERR_INVALID, "blissgpu_analyze_batch_interleaved", "sample_format must be F32, S16 or S32");
    if (channels == 0 || channels > 8) return fail(BLISSGPU_ERR_INVALID, "blissgpu_analyze_batch_interleaved", "channels must be 1..8");
    const int bytes = sample_format == BLISSGPU_SAMPLE_S16 ? 2 : 4;
    return batch_from_offsets(pcm, (size_t)bytes * channels, offsets, lengths, n_songs, sample_format, channels, features_version,
                              out, status, "blissgpu_analyze_batch_interleaved");
}

int blissgpu_analyze_batch_decoded(const blissgpu_decoded_song* songs, uint32_t n_songs, uint32_t features_version, float* out,
                                   int32_t* status) {
    const char* who = "blissgpu_analyze_batch_decoded";
    if (n_songs && (!songs || !out)) return fail(BLISSGPU_ERR_INVALID, who, "NULL argument");
    if (!blissgpu_feature_count(features_version)) return fail(BLISSGPU_ERR_INVALID, who, "features_version must be 1 or 2");
    if (n_songs == 0) return BLISSGPU_OK;
    std::vector<bg::FeedSong> feed(n_songs);
    for (uint32_t i = 0; i < n_songs; i++) {
        const blissgpu_decoded_song& d = songs[i];
        if (d.channels == 0 || d.channels > 8) return fail(BLISSGPU_ERR_INVALID, who, "channels must be 1..8");
        if (!known_format(d.sample_format)) return fail(BLISSGPU_ERR_INVALID, who, "sample_format must be F32, S16 or S32");
        feed[i] = bg::FeedSong{d.pcm, d.frames, d.sample_rate, (uint8_t)d.sample_format, (uint8_t)d.channels};
    }
    DeviceRestore restore;
    blissgpu_ctx* c;
    const int rc = default_ctx(&c);
    if (rc) return rc;
    return analyze_host_songs(c, feed.data(), n_songs, features_version, out, status, who);
}

int blissgpu_analyze(const float* pcm, uint64_t len, uint32_t features_version, float* out, int32_t* status) {
    return analyze_one(pcm, len, BLISSGPU_SAMPLE_F32, 1, bg::SWR_OUT_RATE, features_version, out, status, "blissgpu_analyze");
}

int blissgpu_analyze_interleaved(const void* pcm, int sample_format, uint32_t channels, uint64_t frames,
                                 uint32_t features_version, float* out, int32_t* status) {
    if (!known_format(sample_format))
        return fail(BLISSGPU_ERR_INVALID, "blissgpu_analyze_interleaved", "sample_format must be F32, S16 or S32");
    return analyze_one(pcm, frames, sample_format, channels, bg::SWR_OUT_RATE, features_version, out, status,
                       "blissgpu_analyze_interleaved");
}

int blissgpu_analyze_decoded(const void* pcm, int sample_format, uint32_t channels, uint64_t frames, uint32_t sample_rate,
                             uint32_t features_version, float* out, int32_t* status) {
    if (!known_format(sample_format))
        return fail(BLISSGPU_ERR_INVALID, "blissgpu_analyze_decoded", "sample_format must be F32, S16 or S32");
    return analyze_one(pcm, frames, sample_format, channels, sample_rate, features_version, out, status, "blissgpu_analyze_decoded");
}

int blissgpu_pcm_s16_to_f32_device(blissgpu_ctx* c, const int16_t* d_in, uint64_t n, float* d_out) {
    if (!c || (n && (!d_in || !d_out))) return fail(BLISSGPU_ERR_INVALID, "blissgpu_pcm_s16_to_f32_device", "NULL argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    launch_pcm_convert(d_in, BLISSGPU_SAMPLE_S16, 1, d_out, n, c->stream);
    HIP_TRY(hipGetLastError());
    return BLISSGPU_OK;
}

int blissgpu_pcm_downmix_device(blissgpu_ctx* c, const void* d_in, int sample_format, uint32_t channels, uint64_t frames,
                                float* d_out) {
    if (!c || (frames && (!d_in || !d_out))) return fail(BLISSGPU_ERR_INVALID, "blissgpu_pcm_downmix_device", "NULL argument");
    if (!known_format(sample_format) || channels == 0 || channels > 8)
        return fail(BLISSGPU_ERR_INVALID, "blissgpu_pcm_downmix_device", "bad sample_format / channels");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    launch_pcm_convert(d_in, sample_format, channels, d_out, frames, c->stream);
    HIP_TRY(hipGetLastError());
    return BLISSGPU_OK;
}

uint64_t blissgpu_resampled_len(uint64_t frames, uint32_t sample_rate) {
    if (sample_rate == bg::SWR_OUT_RATE) return frames;
    bg::SwrPlan p;
    if (sample_rate > bg::MAX_SAMPLE_RATE || !bg::swr_make_plan(sample_rate, &p)) return 0;
    return bg::swr_out_len(p, frames);
}

int blissgpu_resample_filter(uint32_t sample_rate, float* bank, uint64_t max_elems, uint32_t* taps, uint32_t* phase_count) {
    bg::SwrPlan p;
    if (sample_rate == 0 || sample_rate > bg::MAX_SAMPLE_RATE || !bg::swr_make_plan(sample_rate, &p))
        return fail(BLISSGPU_ERR_INVALID, "blissgpu_resample_filter", "sample_rate must be 1 .. 768000 Hz");
    if (taps) *taps = (uint32_t)p.taps;
    if (phase_count) *phase_count = (uint32_t)p.phase_count;
    if (bank) {
        std::vector<float> b;
        bg::swr_make_filter(p, b);
        memcpy(bank, b.data(), sizeof(float) * (size_t)std::min<uint64_t>(max_elems, b.size()));
    }
    return BLISSGPU_OK;
}

int blissgpu_pcm_decode_device(blissgpu_ctx* c, const void* d_in, int sample_format, uint32_t channels, uint64_t frames,
                               uint32_t sample_rate, float* d_out) {
    const char* who = "blissgpu_pcm_decode_device";
    if (!known_format(sample_format) || channels == 0 || channels > 8) return fail(BLISSGPU_ERR_INVALID, who, "bad sample_format / channels");
    if (sample_rate == 0 || sample_rate > bg::MAX_SAMPLE_RATE) return fail(BLISSGPU_ERR_INVALID, who, "sample_rate must be 1 .. 768000 Hz");
    const uint64_t n_out = blissgpu_resampled_len(frames, sample_rate);
    if (!c || (n_out && (!d_in || !d_out))) return fail(BLISSGPU_ERR_INVALID, who, "NULL argument");
    if (n_out == 0) return BLISSGPU_OK;  // a stream shorter than the resampler's start-up converts to nothing
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    return bg::enqueue_decode(c, d_in, sample_format, channels, frames, sample_rate, d_out, n_out, c->stream, who);
}

}  // extern "C"
