/*
 * bliss_oracle.h -- CPU restatement of bliss-rs's per-song analysis hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the parity oracle for the HIP path: it may be
 * imported / linked / executed only by tests/, __graft_entry__.smoke() and the cpu_baseline
 * leg of bench.py.  The product (bliss-rs_amd/) never calls it and has no CPU fallback.
 *
 * It restates, in plain C99, the algorithms of the reference at /root/reference (bliss-audio
 * 0.13.0); every function cites the reference file:line it follows.  Third-party arithmetic the
 * reference delegates to crates that are NOT vendored under /root/reference is restated from
 * its published behaviour and pinned by the reference's own fixtures:
 *   rustfft 6.4.1       (FFT)                  -> own radix-2/4 f32 FFT; pinned by librosa-stft.npy (1e-4),
 *                                                 chroma.npy (1e-7), end-to-end vectors (1e-5)
 *   ndarray 0.17.2      (dot, sum, std_axis)   -> unrolled_dot / Welford restated below
 *   ndarray-stats 0.7.0 (quantile Midpoint, argmax first-max)
 *   noisy_float 0.2.1   (total order on non-NaN floats)
 * The Rust reference itself cannot be built in this image (no rustc/cargo, ~200 crates not
 * vendored, no network), so there is no oracle/_ref; parity is pinned by the reference's golden
 * vectors and .npy fixtures (tests/golden/, see tests/test_oracle_golden.py).
 */
#ifndef BLISS_ORACLE_H
#define BLISS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BO_SAMPLE_RATE 22050u /* src/lib.rs:140 */

/* ---- src/utils.rs ---- */
void bo_reflect_pad(const float *x, size_t n, size_t pad, float *out); /* :11-24, out has n+2*pad */
size_t bo_stft_frames(size_t n, size_t hop);                           /* :29-32 (ceil computed in f32) */
/* :26-64; out is [frames][win/2+1] (frame-major; the reference returns the transposed view) */
void bo_stft(const float *signal, size_t n, size_t win, size_t hop, double *out);
float bo_mean(const float *x, size_t n);                 /* :66-68 */
float bo_std(const float *x, size_t n);                  /* ndarray std_axis(ddof=0), Welford in f32 */
uint32_t bo_number_crossings(const float *x, size_t n);  /* :81-95 */
float bo_geometric_mean(const float *x, size_t n);       /* :101-117 */

/* ---- src/chroma.rs ---- */
void bo_chroma_filter(uint32_t sr, size_t n_fft, uint32_t n_chroma, double tuning,
                      double *out /* [n_chroma][n_fft/2+1] */);                       /* :197-267 */
/* :269-331; spec is [frames][bins]; pitches/mags need capacity frames*bins/2; returns count */
size_t bo_pip_track(uint32_t sr, const double *spec, size_t frames, size_t n_fft,
                    double *pitches, double *mags);
double bo_pitch_tuning(double *freqs, size_t n, double resolution, uint32_t bins_per_octave); /* :334-359 */
double bo_estimate_tuning(uint32_t sr, const double *spec, size_t frames, size_t n_fft,
                          double resolution, uint32_t bins_per_octave);               /* :361-391 */
/* :393-412; spec [frames][bins] is squared in place; out is [n_chroma][frames] */
void bo_chroma_stft(uint32_t sr, double *spec, size_t frames, size_t n_fft, uint32_t n_chroma,
                    double tuning, double *out);
void bo_normalize_feature_sequence(double *feat, size_t rows, size_t cols);           /* :177-188 */
void bo_extract_interval_features(const double *chroma /* [12][frames] */, size_t frames,
                                  double *out /* [10][frames] */);                    /* :157-175 */
int bo_chroma_interval_features(const double *chroma, size_t frames, double out[10]); /* :137-155 */
/* ChromaDesc::do_ (:73-85) on a whole signal; returns malloc'd [12][frames] chroma */
double *bo_chroma_desc_do(const float *signal, size_t n, size_t *frames_out, double *tuning_out);
void bo_chroma_get_values(const double *chroma, size_t frames, float out[13]);        /* :97-126 */
void bo_chroma_get_values_v1(const double *chroma, size_t frames, float out[10]);     /* :128-132 */

/* ---- src/aubio.rs + src/timbral.rs + src/temporal.rs + src/misc.rs: streaming descriptors ---- */
typedef struct bo_spectral_desc bo_spectral_desc;   /* src/timbral.rs:27-209 */
bo_spectral_desc *bo_spectral_desc_new(uint32_t sr);
void bo_spectral_desc_do(bo_spectral_desc *d, const float *chunk /* >= 128 samples */);
void bo_spectral_desc_get(bo_spectral_desc *d, float centroid[2], float rolloff[2], float flatness[2]);
size_t bo_spectral_desc_series(bo_spectral_desc *d, const float **c, const float **r, const float **f);
void bo_spectral_desc_free(bo_spectral_desc *d);
/* one PVoc frame on a 512 window already assembled (debug aid for per-frame parity): 256 norms */
void bo_pvoc512_norms(const float *window512, float norms256[256], float norms257[257]);

typedef struct bo_bpm_desc bo_bpm_desc;              /* src/temporal.rs:32-85 */
bo_bpm_desc *bo_bpm_desc_new(uint32_t sr);
void bo_bpm_desc_do(bo_bpm_desc *d, const float *chunk, size_t chunk_len /* >= 256 */);
float bo_bpm_desc_get_value(bo_bpm_desc *d);
size_t bo_bpm_desc_bpms(bo_bpm_desc *d, const float **bpms);
/* debug taps for stage-by-stage parity: onset (specflux) and thresholded series so far */
size_t bo_bpm_desc_series(bo_bpm_desc *d, const float **onset, const float **thresholded);
void bo_bpm_desc_free(bo_bpm_desc *d);

void bo_loudness(const float *x, size_t n, int chunks_exact, float out[2]); /* src/misc.rs:39-71 */
float bo_zcr(const float *x, size_t n);                                     /* src/timbral.rs:231-258 */

/* ---- src/song/mod.rs:413-508 ---- */
#define BO_OK 0
#define BO_ERR_TOO_SHORT 1 /* AnalysisError("empty or too short song.") */
#define BO_ERR_VERSION 2
int bo_song_analyze(const float *x, size_t n, uint32_t features_version /* 1|2 */, float *out /* 20|23 */);
/* same, with wall-clock seconds per descriptor: secs = {tempo, timbral, zcr, loudness, chroma} (instrumentation) */
int bo_song_analyze_timed(const float *x, size_t n, uint32_t features_version, float *out, double secs[5]);
/* analyse n_songs (ragged batch) on n_threads OS threads, songs dealt round-robin
 * (mirrors analyze_paths_with_options, src/song/decoder.rs:282-331) */
void bo_song_analyze_batch(const float *pcm, const uint64_t *offsets, const uint64_t *lengths,
                           uint32_t n_songs, uint32_t features_version, float *out, int32_t *status,
                           uint32_t n_threads);

/* ---- src/playlist.rs:65-79,129-142 ; src/lib.rs:168-178,209-234 ---- */
float bo_euclidean_distance(const float *a, const float *b, size_t d);
float bo_cosine_distance(const float *a, const float *b, size_t d);
float bo_mahalanobis_distance(const float *a, const float *b, const float *m /* d*d row-major */, size_t d);
void bo_feature_weights(uint32_t features_version, float *m /* d*d */);
void bo_pairwise(const float *A, size_t n, const float *B, size_t m, size_t d, int metric /*0 euclid,1 cosine,2 mahalanobis*/,
                 const float *M, float *out /* n*m */, uint32_t n_threads);

/* ---- playlist ordering, src/playlist.rs:24-59, 173-221, 256-326, 367-402 ---- */
float bo_set_distance(const float *seeds, size_t n_seeds, const float *v, size_t d, int metric, const float *M);
int bo_closest_to_songs(const float *seeds, size_t n_seeds, const float *cand, size_t n, size_t d, int metric,
                        const float *M, uint32_t *order /* n */, float *dist_out /* n or NULL */);
int bo_song_to_song(const float *seeds, size_t n_seeds, const float *cand, size_t n, size_t d, int metric,
                    const float *M, uint32_t *order /* n */);
long bo_dedup_playlist(const float *songs, size_t n, size_t d, int metric, const float *M, float threshold,
                       const uint8_t *same_meta /* n*n or NULL */, uint32_t *kept /* n */);
int bo_variance_weight_matrix(const float *seeds, size_t n_seeds, size_t d, float *m /* d*d */);


/* ---- the decoder's conversion to mono 22 050 Hz f32: libswresample as FFmpegDecoder drives it
 * (src/song/decoder/ffmpeg.rs:36-109); restated from FFmpeg's published source, see bliss_oracle.c.
 * Pinned by the reference's Adler-32 decoder tests (ffmpeg.rs:433-452, 471-476). ---- */
typedef struct {
    uint32_t in_rate;
    int taps, phase_count, center;
    uint64_t dst_incr, src_incr; /* output k sits at floor(k dst_incr / src_incr) / phase_count input samples */
    double factor;
} bo_swr_plan_t;
int bo_swr_plan(uint32_t in_rate, bo_swr_plan_t *p);
void bo_swr_filter(const bo_swr_plan_t *p, float *bank /* [phase_count][taps] */);
uint64_t bo_swr_out_len(uint64_t n_in, uint32_t in_rate);
void bo_swr_resample(const float *x, uint64_t n, uint32_t in_rate, float *out /* bo_swr_out_len */);
/* sample_format 0 f32 / 1 s16 / 2 s32; interleaved channels; returns the sample count written to out */
uint64_t bo_decode_to_mono(const void *pcm, int sample_format, uint32_t channels, uint64_t frames, uint32_t in_rate,
                           float *out);

/* ---- bench/test input generator (not part of the reference): Philox4x32-10 white noise,
 * uniform [-0.5, 0.5), key = (0x5EED0000 + song_index, 0), counter = sample_index / 4 ---- */
void bo_white_noise(uint32_t song_index, size_t n, float *out);

/* tests only: evaluate every FFT in f64 (rounded to f32 once) to measure FFT-rounding sensitivity */
void bo_set_fft_double(int on);

#ifdef __cplusplus
}
#endif
#endif
