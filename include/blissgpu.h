/*
 * blissgpu.h -- C ABI of the MI355X-native implementation of bliss-rs's per-song analysis hot path
 * and feature-vector distances.  This is the drop-in boundary: plain pointers and sizes, no C++/torch
 * types, no exceptions across the boundary.  Every entry point cites the reference interface it
 * replaces (file:line into bliss-rs / `bliss-audio` 0.13.0); INTEGRATION.md shows the Rust
 * `extern "C"` binding a bliss-rs maintainer would add.
 *
 * Input contract (same as Song::analyze, src/song/mod.rs:389-392): mono, 22 050 Hz, f32le PCM.
 * Output: one row of 23 (FeaturesVersion::Version2) or 20 (Version1) f32 per song, in the reference's
 * order [tempo, zcr, centroid mean/std, rolloff mean/std, flatness mean/std, loudness mean/std,
 * chroma x13|x10] (src/song/mod.rs:493-498, 102-156).
 *
 * Threading (src/song/decoder.rs:299-329: analyze is called from up to cores + 1 worker threads): EVERY entry point
 * is re-entrant and thread-safe.  Each context carries a mutex; calls on one context are serialised (the host-pointer
 * forms hold it for the whole call, the device forms while they enqueue), calls on different contexts run
 * concurrently.  The entry points without a context argument use process-wide DEFAULT contexts, one per visible HIP
 * device (created on first use; BLISSGPU_DEFAULT_DEVICES="0,2,3" restricts / orders them).
 * Concurrent single-song calls (blissgpu_analyze / blissgpu_analyze_interleaved) are COALESCED: the calls that
 * arrive while a device is busy are analysed together as its next batch, and a batch goes to whichever default device
 * is free -- so N worker threads each calling Song::analyze reach batch throughput on EVERY GPU of the node without
 * changing the caller.  The batch / distance / playlist forms without a context argument run on the first default
 * device (blissgpu_node_* spreads a batch over the node).
 * The library has NO CPU fallback: every compute entry point fails with BLISSGPU_ERR_NO_DEVICE when no
 * gfx950 device / HIP runtime is usable.
 */
#ifndef BLISSGPU_H
#define BLISSGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- return codes (whole-call) ---- */
#define BLISSGPU_OK 0
#define BLISSGPU_ERR_NO_DEVICE 1      /* no usable HIP device: there is no CPU path */
#define BLISSGPU_ERR_INVALID 2        /* bad argument (NULL pointer, unknown features_version/metric, d > 64) */
#define BLISSGPU_ERR_HIP 3            /* a HIP runtime call failed; see blissgpu_last_error() */
#define BLISSGPU_ERR_OOM 4            /* workspace allocation failed */
#define BLISSGPU_ERR_NAN 5            /* a distance is NaN: the reference panics there (n32(), argmin().unwrap()) */
#define BLISSGPU_ERR_RCCL 6           /* librccl could not be loaded or a collective failed (blissgpu_node_* only) */
#define BLISSGPU_ERR_TIMEOUT 7        /* a single-song call was not picked up by any default context within its deadline
                                         (blissgpu_set_single_song_timeout_ms); blissgpu_last_error() names the seats */

/* ---- per-song status, maps 1:1 onto BlissError (src/lib.rs:236-252) ---- */
#define BLISSGPU_SONG_OK 0
#define BLISSGPU_SONG_TOO_SHORT 1     /* AnalysisError("empty or too short song.") -- len < 8192 (src/song/mod.rs:417-430) */

/* ---- FeaturesVersion (src/lib.rs:151-187) ---- */
#define BLISSGPU_FEATURES_V1 1u       /* 20 features */
#define BLISSGPU_FEATURES_V2 2u       /* 23 features (LATEST) */

/* ---- sample formats of the PCM feed (decoder output before the mono f32 conversion) ---- */
#define BLISSGPU_SAMPLE_F32 0
#define BLISSGPU_SAMPLE_S16 1         /* sample / 32768 */
#define BLISSGPU_SAMPLE_S32 2         /* (float)sample / 2^31: how FFmpeg delivers 24- and 32-bit streams */
#define BLISSGPU_SAMPLE_RATE 22050u   /* SAMPLE_RATE (src/lib.rs:140): the rate Song::analyze works at */

/* ---- distance metrics (src/playlist.rs:65-79, 129-142) ---- */
#define BLISSGPU_METRIC_EUCLIDEAN 0
#define BLISSGPU_METRIC_COSINE 1
#define BLISSGPU_METRIC_MAHALANOBIS 2

typedef struct blissgpu_ctx blissgpu_ctx;

/* Context = device selection + constant tables (windows, twiddles, the 100 chroma filter banks of
 * chroma_filter(), src/chroma.rs:197-267) + a grow-only HBM workspace + one HIP stream. */
int blissgpu_ctx_create(int device, blissgpu_ctx **ctx);
int blissgpu_ctx_destroy(blissgpu_ctx *ctx);
/* Launch on a caller-owned stream (a hipStream_t passed as void*, e.g. torch's current stream);
 * NULL restores the context's own stream. */
int blissgpu_ctx_set_stream(blissgpu_ctx *ctx, void *hip_stream);
void *blissgpu_ctx_get_stream(blissgpu_ctx *ctx);
/* Ordering against streams the caller owns, without host synchronisation (hipStream_t passed as void*, NULL = the
 * legacy default stream): wait_stream makes the context's stream wait for everything queued on producer_stream
 * (inputs written there), signal_stream makes consumer_stream wait for everything queued on the context's stream
 * (results read there). */
int blissgpu_ctx_wait_stream(blissgpu_ctx *ctx, void *producer_stream);
int blissgpu_ctx_signal_stream(blissgpu_ctx *ctx, void *consumer_stream);
/* Upper bound in bytes for the scratch workspace of ONE chunk (there are two chunk slots; default: a third of the
 * device memory free at creation, at most 64 GiB).  Larger batches are cut into length-bucketed chunks (longest songs
 * first) that stream through the two slots: chunk k + 1's FFT kernels overlap chunk k's per-song tails.  A chunk that
 * does not fit the memory actually free is halved and retried. */
int blissgpu_ctx_set_workspace_limit(blissgpu_ctx *ctx, uint64_t bytes);
uint64_t blissgpu_ctx_get_workspace_limit(blissgpu_ctx *ctx);
int blissgpu_ctx_synchronize(blissgpu_ctx *ctx);
/* Scheduling knobs of ONE context for the measurement tools and the tests (the library reads no environment variable for
 * them; defaults are what production uses).  Synchronises the context's stream. */
#define BLISSGPU_OPT_SERIAL 0           /* 1: every kernel on one stream, nothing overlaps (clean per-kernel timings) */
#define BLISSGPU_OPT_TAIL_MODE 1        /* beat tracker: -1 auto (default), 0 beside / 1 behind the FFT-8192 kernel.  Measurement forms of
                                           round 6 (same rows; none wins, profiles/r06_tail_mask_ab.txt): N >= 2 beside it with the
                                           per-song state machines on a stream confined to N compute units; -2 only the
                                           autocorrelations beside it; -3 the whole tracker beside the chroma contraction */
#define BLISSGPU_OPT_PIPELINE_CHUNKS 2  /* cut big batches into at least this many chunks (default 1) */
#define BLISSGPU_OPT_CAND_BUDGET 3      /* tuning-candidate pool: slots per chroma frame (default 48; 0 starves the pool) */
#define BLISSGPU_OPT_ROLLOFF_EXACT_ALL 4 /* 1: every frame's rolloff bin through the reference-order pass, not only the frames
                                           the FFT-512 kernel cannot prove (tests: both must give the same rows) */
#define BLISSGPU_OPT_TAIL_SPLIT 6       /* one-chunk batches: P >= 2 = the tuning estimate and the contraction run in P pieces of the
                                           songs (at most 8), the contraction of piece k beside the tuning estimate of piece
                                           k + 1; 0 / 1 = unsplit (default: see DESIGN.md section 3b) */
#define BLISSGPU_OPT_FLUX_ORDER 8       /* 1: SpecFlux (src/aubio.rs:455-467) adds its 257 terms one by one in bin order, as the
                                           reference does, instead of 16 per lane + a tree.  The default order deviates from the
                                           reference's by 1.7e-7 rms per value -- inside every tolerance, and the reason the tempo of
                                           white noise sits on the noisier side of the f32-FFT floor; with this option the device's
                                           tempo is as close to the f64-FFT oracle as a plain f32 FFT's.
                                           FFT-512 kernel +19 %, step +6 %.  Default 0 */
#define BLISSGPU_OPT_STFT_SHAPE 7       /* FFT-8192 kernel: 0 = four workgroups per CU, window in registers (default); 1 = the narrow
                                           form (five per CU: window loaded per frame, transposes in two halves); 2 / 3 = one of
                                           the two changes alone.  Same rows bit for bit; 1 is 8 % slower */
#define BLISSGPU_OPT_DEBUG_CHROMA 5     /* 1: the contraction also keeps chroma_stft's matrix and the assembly the interval
                                           means of the chunk for the CHROMA / INTERVAL taps below (96 B per frame); such a
                                           batch must fit ONE chunk (BLISSGPU_ERR_INVALID otherwise) */
/* The host PCM feed's pinned staging ring (pageable sources -- a Rust Vec<f32> out of PreAnalyzedSong,
 * src/song/decoder.rs:34-65, a decoder's frame buffers -- are copied into page-locked slabs by worker threads that run ahead
 * of the link; page-locked / registered sources are handed to the DMA engines as they are).  The ring restarts with the new
 * shape on the next host-pointer call. */
#define BLISSGPU_OPT_STAGE_LANES 9      /* worker threads = copy queues, 1..16; 0 = no ring: leave the staging of pageable memory
                                           to the HIP runtime (one bounce buffer on the calling thread).  Default 4 */
#define BLISSGPU_OPT_STAGE_SLAB_KIB 10  /* size of one page-locked slab, 64 .. 65536 KiB.  Default 4096 */
#define BLISSGPU_OPT_STAGE_SLABS 11     /* slabs each worker rotates through, 1..8.  Default 3 (4 x 3 x 4 MiB = 48 MiB per context) */
#define BLISSGPU_OPT_STAGE_NUMA 12      /* 1: the workers run, and allocate their slabs, on the CPUs next to the device (sysfs
                                           local_cpulist of its PCI function; a no-op on a one-node host); 0 (default): wherever
                                           the scheduler puts them -- near the caller's buffers, which measured better: a worker
                                           reading the caller's memory across the socket link is slower than the DMA engine
                                           reading the slab across it (profiles/r06_stage_numa_ab.txt) */
int blissgpu_ctx_set_option(blissgpu_ctx *ctx, int option, int64_t value);
/* Bytes of pageable host PCM this context has staged through its pinned ring since it was created (0: every source so far
 * was page-locked, small, or the ring is switched off). */
uint64_t blissgpu_ctx_staged_bytes(blissgpu_ctx *ctx);

/* The default contexts: how many there are, the HIP ordinal of the k-th, and how many coalesced batches of single-song
 * calls it has served so far (load statistics). */
int blissgpu_default_device_count(void);
int blissgpu_default_device(int k);
uint64_t blissgpu_default_device_batches(int k);
/* The k-th default context itself -- the one the entry points WITHOUT a context argument run on (k = 0 serves the batch,
 * distance and playlist forms; every k a seat of the single-song front) -- created now if it does not exist yet.  Borrowed:
 * the library owns it (never pass it to blissgpu_ctx_destroy); use it for blissgpu_ctx_set_option (e.g. the staging ring's
 * shape), the workspace limit, the profile / debug taps. */
int blissgpu_default_ctx(int k, blissgpu_ctx **ctx);
/* How long a single-song call (blissgpu_analyze / _interleaved) may wait for a default context to pick it up before it
 * fails with BLISSGPU_ERR_TIMEOUT instead of blocking (default 600 000 ms; <= 0 restores the default; clamped to ten
 * years).  A default context
 * whose device cannot give a context is retired -- its traffic goes to the others -- and only when all are retired do the
 * calls fail, with the creation error. */
int blissgpu_set_single_song_timeout_ms(int64_t ms);
/* A default context that could not be created is not retried on every call: a failure that cannot change (no such device,
 * another architecture) is remembered for good, anything else (no memory for the tables while another process holds the
 * device, a HIP error) is tried again after a back-off of 100 ms .. 5 s, and a retired seat of the single-song front is offered
 * traffic again after 1 s .. 64 s.  blissgpu_default_reset() forgets every remembered failure and revives every seat now. */
int blissgpu_default_reset(void);

uint32_t blissgpu_feature_count(uint32_t features_version); /* FeaturesVersion::feature_count, src/lib.rs:181-186 */

/* Replaces Song::analyze / Song::analyze_with_options (src/song/mod.rs:403-508) for ONE song in host
 * memory.  Returns BLISSGPU_OK and writes feature_count floats, or BLISSGPU_OK with *status =
 * BLISSGPU_SONG_TOO_SHORT (out filled with NaN).  status may be NULL.  Uses the process-wide default
 * contexts; safe to call from any number of threads (concurrent calls are coalesced into device batches, one in flight
 * per device). */
int blissgpu_analyze(const float *pcm, uint64_t len, uint32_t features_version, float *out, int32_t *status);
/* Same for raw decoder output: `frames` frames of `channels` interleaved samples (BLISSGPU_SAMPLE_F32 / _S16 / _S32) at
 * 22 050 Hz.  s16 is widened with sample / 32768 and channels are downmixed ON THE DEVICE exactly like the reference's
 * decoders: stereo -> (L + R) * SQRT_2 / 2, more channels -> their mean (src/song/decoder/symphonia.rs:266-300; pinned
 * on data/s16_stereo_22_5kHz.flac by Adler-32 0x1d7b2d6d, src/song/decoder/ffmpeg.rs:448-452).  The caller delivers
 * 22 050 Hz here; blissgpu_analyze_decoded takes any rate. */
int blissgpu_analyze_interleaved(const void *pcm, int sample_format, uint32_t channels, uint64_t frames,
                                 uint32_t features_version, float *out, int32_t *status);

/* Bulk form: the compute half of Decoder::analyze_paths_with_options (src/song/decoder.rs:278-332) once the
 * decoders have produced PCM.  pcm holds the songs back to back (song i = pcm[offsets[i] ..
 * offsets[i] + lengths[i])); out is n_songs x feature_count row-major; status has one entry per
 * song -- one bad song never aborts the batch (src/song/decoder.rs:313-325).  Host pointers. */
int blissgpu_analyze_batch(const float *pcm, const uint64_t *offsets, const uint64_t *lengths, uint32_t n_songs,
                           uint32_t features_version, float *out, int32_t *status);

/* Same, for decoders that deliver signed 16-bit mono 22 050 Hz PCM (what FFmpeg hands to the reference's resampler
 * for the golden files, src/song/decoder/ffmpeg.rs:36-109): the samples cross PCIe as 2 bytes and are widened on the
 * device with sample / 32768, FFmpeg's s16 -> flt conversion (bit-identical to converting on the host first).
 * offsets / lengths are in samples.  Both host forms pipeline the transfer of one group of songs with the analysis
 * of the previous one. */
int blissgpu_analyze_batch_s16(const int16_t *pcm, const uint64_t *offsets, const uint64_t *lengths, uint32_t n_songs,
                               uint32_t features_version, float *out, int32_t *status);
/* Bulk form for interleaved multi-channel decoder output (see blissgpu_analyze_interleaved); offsets / lengths are in
 * FRAMES. */
int blissgpu_analyze_batch_interleaved(const void *pcm, int sample_format, uint32_t channels, const uint64_t *offsets,
                                       const uint64_t *lengths, uint32_t n_songs, uint32_t features_version, float *out,
                                       int32_t *status);
/* ---- decoder output at ANY sample rate (src/song/decoder/ffmpeg.rs:36-109) ----
 * The reference's FFmpegDecoder hands every decoded frame to libswresample with its default options (Kaiser-windowed sinc,
 * filter_size 32, cutoff 0.97, exact rational phases) and asks for mono f32 at 22 050 Hz.  These entry points take what the
 * DECODER delivers -- `frames` frames of `channels` interleaved samples at `sample_rate` Hz -- and do that conversion ON THE
 * DEVICE -- bit for bit at 44 100 Hz, the only rate the reference pins --: widening (s16 / s32), libswresample's resampler with the summation order of its AVX2 + FMA3 kernel,
 * the stream mirrored at both ends, stereo = each channel resampled, then l * sqrt(1/2) + r * sqrt(1/2) (more channels: the
 * sequential mean first, as src/song/decoder/symphonia.rs:291-297, then the resampler).  sample_rate 22 050 is the pass-through
 * of blissgpu_analyze_interleaved.  Pinned by the reference's own Adler-32 decoder tests: 0xa0f8b8af
 * (data/s32_mono_44_1_kHz.flac), 0xbbcba1cf (s32_stereo_44_1_kHz.flac) -- ffmpeg.rs:433-445 -- and 0xd594429c (no_channel.wav,
 * :471-476); with it the three CUE tracks of data/testcue.flac give the 3 x 23 features src/cue.rs:270-415 asserts.
 * Rates other than 44 100 Hz (147 / 441 / 1024 phases, up-sampling) run the same restatement of libswresample's published
 * algorithm; no reference test holds a number for them and no FFmpeg build was available to make one: they are held to the
 * oracle's independent restatement bit for bit and to a sinusoid-reconstruction property, i.e. NOT externally pinned. */
int blissgpu_analyze_decoded(const void *pcm, int sample_format, uint32_t channels, uint64_t frames, uint32_t sample_rate,
                             uint32_t features_version, float *out, int32_t *status);
/* Bulk form: every song with its own buffer, format, channel count and rate (a library is a mix of 44.1 and 48 kHz, mono
 * and stereo files).  The compute half of Decoder::analyze_paths_with_options (src/song/decoder.rs:278-332) for a host that
 * keeps its decoder and drops the resampler. */
typedef struct blissgpu_decoded_song {
    const void *pcm;        /* host memory: frames x channels interleaved samples */
    uint64_t frames;
    uint32_t sample_rate;   /* Hz, 1 .. 768 000 */
    uint16_t channels;      /* 1 .. 8 */
    uint16_t sample_format; /* BLISSGPU_SAMPLE_* */
} blissgpu_decoded_song;
int blissgpu_analyze_batch_decoded(const blissgpu_decoded_song *songs, uint32_t n_songs, uint32_t features_version, float *out,
                                   int32_t *status);
/* Number of 22 050 Hz samples `frames` input frames become (device-free; = frames at 22 050 Hz, 0 when the stream is shorter
 * than the resampler's start-up needs or the rate is out of range). */
uint64_t blissgpu_resampled_len(uint64_t frames, uint32_t sample_rate);
/* The resampler's filter bank for one input rate (device-free; test tap): *taps x *phase_count floats, phase-major; bank
 * may be NULL to query the sizes, at most max_elems floats are written. */
int blissgpu_resample_filter(uint32_t sample_rate, float *bank, uint64_t max_elems, uint32_t *taps, uint32_t *phase_count);

/* The conversions alone, device to device (asynchronous on the context's stream). */
int blissgpu_pcm_s16_to_f32_device(blissgpu_ctx *ctx, const int16_t *d_in, uint64_t n_samples, float *d_out);
int blissgpu_pcm_downmix_device(blissgpu_ctx *ctx, const void *d_in, int sample_format, uint32_t channels, uint64_t frames,
                                float *d_out);
/* decoder output at sample_rate -> mono 22 050 Hz f32; d_out holds blissgpu_resampled_len(frames, sample_rate) samples */
int blissgpu_pcm_decode_device(blissgpu_ctx *ctx, const void *d_in, int sample_format, uint32_t channels, uint64_t frames,
                               uint32_t sample_rate, float *d_out);

/* Device-resident form: d_pcm / d_out / d_status are HIP device pointers (d_status may be NULL),
 * offsets / lengths stay on the host (they size the launch).  Asynchronous on the context's
 * stream (d_status is written by the device too: no host synchronisation inside); call blissgpu_ctx_synchronize (or
 * synchronise the stream) before reading d_out.  This is the streaming scheduler of configs like "50 000 songs of
 * 30 s - 10 min": songs are bucketed by length and run as chunks through two workspace slots. */
int blissgpu_analyze_batch_device(blissgpu_ctx *ctx, const float *d_pcm, const uint64_t *offsets,
                                  const uint64_t *lengths, uint32_t n_songs, uint32_t features_version,
                                  float *d_out, int32_t *d_status);

/* Song::distance / Analysis::distance (src/song/mod.rs:364-370, 519-521) and the free functions
 * euclidean_distance / cosine_distance / mahalanobis_distance (src/playlist.rs:65-79, 140-142) for one
 * pair (host pointers; M is d x d row-major, required for MAHALANOBIS, ignored otherwise). */
int blissgpu_distance(const float *a, const float *b, uint32_t d, int metric, const float *M, float *out);

/* All-pairs form: out[i * m + j] = metric(A[i], B[j]).  A is n x d, B is m x d, row-major f32.  This
 * is the batched equivalent of evaluating a DistanceMetric over every candidate
 * (src/playlist.rs:24-59, 256-270).  Host pointers. */
int blissgpu_pairwise(const float *A, uint64_t n, const float *B, uint64_t m, uint32_t d, int metric, const float *M,
                      float *out);
/* Device-resident form; ld_out is the row pitch of d_out in elements (>= m).  Asynchronous. */
int blissgpu_pairwise_device(blissgpu_ctx *ctx, const float *d_A, uint64_t n, const float *d_B, uint64_t m,
                             uint32_t d, int metric, const float *d_M, float *d_out, uint64_t ld_out);

/* ---- playlist ordering (src/playlist.rs:24-59, 256-326) ----
 * A "song" is a row of a feature matrix; results are index permutations into the candidate matrix.
 * The metric built from a set of vectors is FunctionDistanceMetric (src/playlist.rs:36-59): the sequential
 * f32 sum over the set of func(vector_of_the_set, candidate), func = one of the three metrics above.
 * A NaN distance returns BLISSGPU_ERR_NAN (the reference panics: n32() / argmin().unwrap()). */

/* n_seeds may be 0: the sum over an empty set is 0.0 for every candidate, so closest_to_songs returns the candidates
 * in their own order and song_to_song starts from candidate 0. */

/* FunctionDistanceMetric::distance for every candidate: out[j] = sum_i metric(seeds[i], cand[j]). */
int blissgpu_set_distance(const float *seeds, uint32_t n_seeds, const float *cand, uint64_t n, uint32_t d, int metric,
                          const float *M, float *out);
/* closest_to_songs (src/playlist.rs:256-270): order[k] = index of the k-th closest candidate to the seed set
 * (stable: equal distances keep the candidates' order, like sort_by_cached_key); dist (may be NULL) receives
 * the distances in candidate order. */
int blissgpu_closest_to_songs(const float *seeds, uint32_t n_seeds, const float *cand, uint64_t n, uint32_t d,
                              int metric, const float *M, uint32_t *order, float *dist);
/* song_to_song (src/playlist.rs:272-326): greedy nearest-neighbour chain.  order[0] = the candidate closest to the
 * seed set, order[k] = the remaining candidate closest to candidate order[k-1] (first minimum in pool order). */
int blissgpu_song_to_song(const float *seeds, uint32_t n_seeds, const float *cand, uint64_t n, uint32_t d, int metric,
                          const float *M, uint32_t *order);
/* Device-resident forms (all pointers are HIP device pointers; asynchronous except for the NaN check, which
 * synchronises the context's stream before returning). */
int blissgpu_set_distance_device(blissgpu_ctx *ctx, const float *d_seeds, uint32_t n_seeds, const float *d_cand,
                                 uint64_t n, uint32_t d, int metric, const float *d_M, float *d_out);
int blissgpu_closest_to_songs_device(blissgpu_ctx *ctx, const float *d_seeds, uint32_t n_seeds, const float *d_cand,
                                     uint64_t n, uint32_t d, int metric, const float *d_M, uint32_t *d_order,
                                     float *d_dist);
int blissgpu_song_to_song_device(blissgpu_ctx *ctx, const float *d_seeds, uint32_t n_seeds, const float *d_cand,
                                 uint64_t n, uint32_t d, int metric, const float *d_M, uint32_t *d_order);

/* FeaturesVersion::feature_weights (src/lib.rs:168-173, 209-234): d x d row-major diagonal matrix. */
int blissgpu_feature_weights(uint32_t features_version, float *M);

/* ---- one process, every GPU of the node ----
 * Songs are independent (Decoder::analyze_paths treats them so, src/song/decoder.rs:299-329), so a library shards by song:
 * greedy longest-first balance of the samples per device, no data-path collective.  After the local batches ONE RCCL
 * all-gather over xGMI (padded to the largest shard) leaves the full n x d feature matrix on every device; the pairwise
 * kernel is then row-block sharded with no further exchange.  RCCL is loaded at run time; without it node creation
 * fails with BLISSGPU_ERR_RCCL.  (bliss_rs_amd/shard.py is the one-process-per-GPU form of the same plan on
 * torch.distributed.) */
typedef struct blissgpu_node blissgpu_node;
/* The plan, device-free (no HIP call, no context): rank_of_song[i] = rank in [0, world) that analyses song i.  Greedy
 * longest-first assignment balancing the samples per rank, ties -> fewest songs -> lowest rank; deterministic, so every
 * host and every process computes the same plan (bliss_rs_amd.shard.shard_songs is the same function). */
int blissgpu_shard_plan(const uint64_t *lengths, uint32_t n_songs, uint32_t world, uint32_t *rank_of_song);
/* Rows [lo, hi) of an n_rows-row distance matrix computed by `rank` of `world` (device-free; rank >= world: empty). */
void blissgpu_row_block(uint64_t n_rows, uint32_t world, uint32_t rank, uint64_t *lo, uint64_t *hi);
/* devices: HIP ordinals (NULL = 0 .. n_devices-1).  Creates one context per rank and the RCCL communicators
 * (ncclCommInitAll).  A list that names an ordinal more than once creates LOOPBACK ranks -- several contexts sharing a
 * GPU: RCCL cannot (and need not) connect them, the gather is done with device-to-device copies; everything else is
 * the same code.  That is how the N > 1 plan / padding / scatter / row-block paths are tested on a one-GPU box. */
int blissgpu_node_create(int n_devices, const int *devices, blissgpu_node **node);
int blissgpu_node_destroy(blissgpu_node *node);
int blissgpu_node_device_count(blissgpu_node *node);
blissgpu_ctx *blissgpu_node_ctx(blissgpu_node *node, int rank); /* the rank's context (device forms, synth, malloc) */
/* The sharding plan: rank_of_song[i] = device rank that analyses song i. */
int blissgpu_node_shard(blissgpu_node *node, const uint64_t *lengths, uint32_t n_songs, uint32_t *rank_of_song);
/* Rows [lo, hi) of an n_rows-row distance matrix computed by `rank`. */
void blissgpu_node_row_block(blissgpu_node *node, uint64_t n_rows, int rank, uint64_t *lo, uint64_t *hi);
/* Bulk analysis of host PCM (the node form of blissgpu_analyze_batch): shards, feeds every device from its own host
 * thread, gathers.  out (host, n_songs x feature_count) and status as in blissgpu_analyze_batch. */
int blissgpu_node_analyze(blissgpu_node *node, const float *pcm, const uint64_t *offsets, const uint64_t *lengths,
                          uint32_t n_songs, uint32_t features_version, float *out, int32_t *status);
/* Device-resident form: song i lives on device rank_of_song[i] at d_pcm[rank_of_song[i]] + offsets[i].  Asynchronous. */
int blissgpu_node_analyze_device(blissgpu_node *node, const float *const *d_pcm, const uint64_t *offsets,
                                 const uint64_t *lengths, const uint32_t *rank_of_song, uint32_t n_songs,
                                 uint32_t features_version);
/* The gathered n_songs x feature_count matrix of the last analysis on `rank`'s device (valid after synchronize). */
const float *blissgpu_node_features(blissgpu_node *node, int rank);
/* All-pairs distances over the gathered matrix, rows sharded across the devices; out is host memory, n x n. */
int blissgpu_node_pairwise(blissgpu_node *node, int metric, const float *M, float *out);
int blissgpu_node_synchronize(blissgpu_node *node);

/* ---- device memory helpers for hosts without their own HIP binding (Rust/C callers) ---- */
int blissgpu_malloc(void **d_ptr, uint64_t bytes);
int blissgpu_free(void *d_ptr);
int blissgpu_memcpy_h2d(blissgpu_ctx *ctx, void *d_dst, const void *h_src, uint64_t bytes);
int blissgpu_memcpy_d2h(blissgpu_ctx *ctx, void *h_dst, const void *d_src, uint64_t bytes);
/* Page-locked host memory for decoder output: H2D copies from it run at full PCIe rate without HIP's staging copy. */
int blissgpu_host_alloc(void **h_ptr, uint64_t bytes);
int blissgpu_host_free(void *h_ptr);

/* ---- benchmark input: synthetic white noise written straight into HBM.  Song i of the call gets
 * uniform [-0.5, 0.5) samples from Philox4x32-10 with key (0x5EED0000 + first_song_index + i, 0) and
 * counter = sample_index / 4 (bit-identical to the oracle's generator).  No reference counterpart. */
int blissgpu_synth_white_noise_device(blissgpu_ctx *ctx, float *d_pcm, const uint64_t *offsets,
                                      const uint64_t *lengths, uint32_t n_songs, uint32_t first_song_index);
/* Same with an explicit generator index per song (a rank's scattered share of a sharded corpus). */
int blissgpu_synth_white_noise_indexed_device(blissgpu_ctx *ctx, float *d_pcm, const uint64_t *offsets,
                                              const uint64_t *lengths, const uint32_t *song_index, uint32_t n_songs);

/* ---- per-kernel timing with HIP events on the context's stream (for bench.py's roofline) ---- */
int blissgpu_profile_enable(blissgpu_ctx *ctx, int enable);
int blissgpu_profile_reset(blissgpu_ctx *ctx);
int blissgpu_profile_kernel_count(void);
const char *blissgpu_profile_kernel_name(int kernel);
/* total_ms / launches accumulated since the last reset (synchronises the stream) */
int blissgpu_profile_get(blissgpu_ctx *ctx, int kernel, double *total_ms, uint64_t *launches);

/* ---- debug taps used by the parity tests: per-song tuning estimate of the last batch ----
 * tuning[i] = estimate_tuning's value for song i of the last analyze_batch_device call on ctx
 * (src/chroma.rs:361-391); n_bpms[i] = number of beats BPMDesc recorded (src/temporal.rs:50-58). */
int blissgpu_debug_last_tuning(blissgpu_ctx *ctx, double *tuning, uint32_t *n_bpms, uint32_t n_songs);

/* Number of chunks the last blissgpu_analyze_batch_device call on ctx was cut into. */
uint64_t blissgpu_debug_last_chunks(blissgpu_ctx *ctx);

/* Intermediate series of song `song` (the caller's index into the last batch; it must belong to the LAST chunk run
 * on ctx) for the per-stage parity tests.  Copies at most max_elems elements (4 bytes; 8 for the f64 taps) to dst,
 * reports the available count in *n_elems. */
#define BLISSGPU_DEBUG_CENTROID 0      /* f32[n_t]  per-frame spectral centroid in Hz (src/timbral.rs:159-173) */
#define BLISSGPU_DEBUG_ROLLOFF 1       /* f32[n_t]  per-frame rolloff in Hz (:175-194) */
#define BLISSGPU_DEBUG_FLATNESS 2      /* f32[n_t]  per-frame flatness (:196-208) */
#define BLISSGPU_DEBUG_FLUX 3          /* f32[n_b]  SpecFlux onset values (src/aubio.rs:455-467) */
#define BLISSGPU_DEBUG_THRESHOLDED 4   /* f32[n_b]  PeakPicker thresholded values (:757) */
#define BLISSGPU_DEBUG_RUN_BPM 5       /* f32[runs] BeatTracking::get_bpm after each run (:1231-1239) */
#define BLISSGPU_DEBUG_RUN_COUNT 6     /* u32[runs] beats BPMDesc recorded while that bpm was current */
#define BLISSGPU_DEBUG_SPECTROGRAM 7   /* f32[n_c][4128] STFT magnitudes, 4097 valid per row (src/utils.rs:26-64) */
#define BLISSGPU_DEBUG_ENERGY256 8     /* f32[ceil(n/256)] sum of squares per 256 samples */
#define BLISSGPU_DEBUG_CROSSINGS256 9  /* u32[ceil(n/256)] zero crossings per 256 samples */
#define BLISSGPU_DEBUG_PITCH_HIST 10   /* u32[100] pitch-residue histogram (peaks above the median's coarse bin) */
/* f64 taps (elements are 8 bytes).  CHROMA and INTERVAL need BLISSGPU_OPT_DEBUG_CHROMA = 1 before the analysis. */
#define BLISSGPU_DEBUG_CHROMA 11       /* f64[n_c][12] chroma_stft's matrix, frame-major, after the column normalisation
                                          (src/chroma.rs:393-412; the reference holds it against data/chroma.npy, :621-639) */
#define BLISSGPU_DEBUG_INTERVAL 12     /* f64[10] chroma_interval_features' time means (src/chroma.rs:137-155) */
#define BLISSGPU_DEBUG_FILTER_BANK 13  /* f64[12][4128] chroma_filter(22050, 8192, 12, tuning) (src/chroma.rs:197-267), 4097
                                          valid per row; `song` is the TUNING SLOT: 0..99 = tuning -0.5 + 0.01 slot, 100 = 0.0 */
int blissgpu_debug_fetch(blissgpu_ctx *ctx, int what, uint32_t song, void *dst, uint64_t max_elems, uint64_t *n_elems);

const char *blissgpu_strerror(int code);
const char *blissgpu_last_error(void); /* thread-local detail of the last failure */
const char *blissgpu_version(void);

#ifdef __cplusplus
}
#endif
#endif /* BLISSGPU_H */
