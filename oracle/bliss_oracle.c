/*
 * bliss_oracle.c -- CPU restatement of bliss-rs's Song::analyze + distance hot path.
 * TEST INFRASTRUCTURE ONLY (see bliss_oracle.h).  Citations are into /root/reference.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off: Rust never contracts a*b+c, so
 * neither may we; the one fused op in the path, ndarray's Welford `mul_add`, uses fmaf()).
 */
#define _GNU_SOURCE
#include "bliss_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define PI_F 3.14159265358979323846f /* std::f32::consts::PI */

/* ------------------------------------------------------------------------------------------
 * FFT (stands in for rustfft 6.4.1 `plan_fft_forward(n)` + `process`, c2c f32, n = 2^k).
 * Iterative radix-2 decimation-in-time, twiddles rounded once from double.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    size_t n;
    unsigned log2n;
    float *tw_re, *tw_im; /* twiddles exp(-2*pi*i*k/n), k < n (the parity path reads k < n/2) */
    uint32_t *rev;
    float *sr, *si;       /* scratch of the baseline-only transform below */
} bo_fft;

static bo_fft *fft_new(size_t n) {
    bo_fft *p = (bo_fft *)malloc(sizeof *p);
    p->n = n;
    p->log2n = 0;
    while (((size_t)1 << p->log2n) < n) p->log2n++;
    p->tw_re = (float *)malloc(sizeof(float) * (n + 1));
    p->tw_im = (float *)malloc(sizeof(float) * (n + 1));
    p->rev = (uint32_t *)malloc(sizeof(uint32_t) * n);
    p->sr = (float *)malloc(sizeof(float) * n);
    p->si = (float *)malloc(sizeof(float) * n);
    for (size_t k = 0; k < n; k++) {
        double a = -2.0 * M_PI * (double)k / (double)n;
        p->tw_re[k] = (float)cos(a);
        p->tw_im[k] = (float)sin(a);
    }
    for (size_t i = 0; i < n; i++) {
        uint32_t r = 0;
        for (unsigned b = 0; b < p->log2n; b++)
            if (i & ((size_t)1 << b)) r |= 1u << (p->log2n - 1 - b);
        p->rev[i] = r;
    }
    return p;
}

static void fft_free(bo_fft *p) {
    if (!p) return;
    free(p->tw_re);
    free(p->tw_im);
    free(p->rev);
    free(p->sr);
    free(p->si);
    free(p);
}

/* Sensitivity switch (tests only): run the butterflies in f64 and round the result to f32 once.
 * Not the reference's arithmetic (rustfft works in f32); it exists to measure how much each feature
 * moves under FFT rounding alone, which bounds what GPU-vs-CPU parity can mean. */
static int g_fft_double = 0;
void bo_set_fft_double(int on) { g_fft_double = on; }

static void fft_forward_f64(const bo_fft *p, float *re32, float *im32) {
    const size_t n = p->n;
    double *re = (double *)malloc(sizeof(double) * n), *im = (double *)malloc(sizeof(double) * n);
    for (size_t i = 0; i < n; i++) { re[p->rev[i]] = re32[i]; im[p->rev[i]] = im32[i]; }
    for (size_t len = 2; len <= n; len <<= 1) {
        const size_t half = len >> 1;
        for (size_t i = 0; i < n; i += len)
            for (size_t j = 0; j < half; j++) {
                const double a = -2.0 * M_PI * (double)j / (double)len;
                const double wr = cos(a), wi = sin(a);
                const double xr = re[i + j + half], xi = im[i + j + half];
                const double vr = xr * wr - xi * wi, vi = xr * wi + xi * wr;
                const double ur = re[i + j], ui = im[i + j];
                re[i + j] = ur + vr; im[i + j] = ui + vi;
                re[i + j + half] = ur - vr; im[i + j + half] = ui - vi;
            }
    }
    for (size_t i = 0; i < n; i++) { re32[i] = (float)re[i]; im32[i] = (float)im[i]; }
    free(re); free(im);
}

#ifdef BO_BASELINE_FAST_FFT
/* BASELINE-ONLY build (oracle/Makefile target `native`, -O3 -march=native): a Stockham autosort radix-4 transform whose
 * inner loops are contiguous, so the compiler vectorises them -- a fairer stand-in for rustfft's SIMD mixed-radix
 * kernels when the port is TIMED beside the GPU (bench.py cpu_baseline).  It rounds differently from the radix-2
 * transform below, so it is never used for parity: the checker is always the default build. */
static void fft_forward_fast(const bo_fft *p, float *re, float *im) {
    const size_t n = p->n;
    float *xr = re, *xi = im, *yr = p->sr, *yi = p->si;
    size_t len = n, s = 1;
    while (len >= 4) {
        const size_t m = len / 4, ts = n / len;
        for (size_t pp = 0; pp < m; pp++) {
            const float w1r = p->tw_re[pp * ts], w1i = p->tw_im[pp * ts];
            const float w2r = p->tw_re[2 * pp * ts], w2i = p->tw_im[2 * pp * ts];
            const float w3r = p->tw_re[3 * pp * ts], w3i = p->tw_im[3 * pp * ts];
            const float *ar = xr + s * pp, *ai = xi + s * pp, *br = ar + s * m, *bi = ai + s * m;
            const float *cr = br + s * m, *ci = bi + s * m, *dr = cr + s * m, *di = ci + s * m;
            float *o0r = yr + s * 4 * pp, *o0i = yi + s * 4 * pp;
            for (size_t q = 0; q < s; q++) {
                const float apcr = ar[q] + cr[q], apci = ai[q] + ci[q], amcr = ar[q] - cr[q], amci = ai[q] - ci[q];
                const float bpdr = br[q] + dr[q], bpdi = bi[q] + di[q], bmdr = br[q] - dr[q], bmdi = bi[q] - di[q];
                /* (-i) * (b - d) = (bmd.im, -bmd.re) */
                const float t1r = amcr + bmdi, t1i = amci - bmdr, t3r = amcr - bmdi, t3i = amci + bmdr;
                const float t2r = apcr - bpdr, t2i = apci - bpdi;
                o0r[q] = apcr + bpdr;
                o0i[q] = apci + bpdi;
                o0r[q + s] = t1r * w1r - t1i * w1i;
                o0i[q + s] = t1r * w1i + t1i * w1r;
                o0r[q + 2 * s] = t2r * w2r - t2i * w2i;
                o0i[q + 2 * s] = t2r * w2i + t2i * w2r;
                o0r[q + 3 * s] = t3r * w3r - t3i * w3i;
                o0i[q + 3 * s] = t3r * w3i + t3i * w3r;
            }
        }
        float *t = xr; xr = yr; yr = t;
        t = xi; xi = yi; yi = t;
        len /= 4;
        s *= 4;
    }
    if (len == 2) {
        for (size_t q = 0; q < s; q++) {
            const float ar = xr[q], ai = xi[q], br = xr[q + s], bi = xi[q + s];
            yr[q] = ar + br; yi[q] = ai + bi;
            yr[q + s] = ar - br; yi[q + s] = ai - bi;
        }
        float *t = xr; xr = yr; yr = t;
        t = xi; xi = yi; yi = t;
    }
    if (xr != re) { memcpy(re, xr, sizeof(float) * n); memcpy(im, xi, sizeof(float) * n); }
}
#endif

/* in-place forward FFT on separate re/im arrays */
static void fft_forward(const bo_fft *p, float *re, float *im) {
    const size_t n = p->n;
    if (g_fft_double) { fft_forward_f64(p, re, im); return; }
#ifdef BO_BASELINE_FAST_FFT
    fft_forward_fast(p, re, im);
    return;
#endif
    for (size_t i = 0; i < n; i++) {
        size_t j = p->rev[i];
        if (j > i) {
            float t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const size_t half = len >> 1, stride = n / len;
        for (size_t i = 0; i < n; i += len) {
            for (size_t j = 0; j < half; j++) {
                const float wr = p->tw_re[j * stride], wi = p->tw_im[j * stride];
                const float xr = re[i + j + half], xi = im[i + j + half];
                const float vr = xr * wr - xi * wi;
                const float vi = xr * wi + xi * wr;
                const float ur = re[i + j], ui = im[i + j];
                re[i + j] = ur + vr;
                im[i + j] = ui + vi;
                re[i + j + half] = ur - vr;
                im[i + j + half] = ui - vi;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * src/utils.rs
 * ---------------------------------------------------------------------------------------- */

/* src/utils.rs:11-24 -- numpy mode="reflect" (edge sample not repeated) */
void bo_reflect_pad(const float *x, size_t n, size_t pad, float *out) {
    for (size_t i = 0; i < pad; i++) out[i] = x[pad - i];             /* array[1..=pad].rev() */
    memcpy(out + pad, x, n * sizeof(float));
    for (size_t i = 0; i < pad; i++) out[pad + n + i] = x[n - 2 - i]; /* array[n-1-pad..n-1].rev() */
}

/* src/utils.rs:29-32 -- (len as f32 / hop as f32).ceil() */
size_t bo_stft_frames(size_t n, size_t hop) {
    return (size_t)ceilf((float)n / (float)hop);
}

/* src/utils.rs:26-64 */
void bo_stft(const float *signal, size_t n, size_t win, size_t hop, double *out) {
    const size_t rows = bo_stft_frames(n, hop), bins = win / 2 + 1, pad = win / 2;
    const size_t padded_n = n + 2 * pad;
    float *padded = (float *)malloc(sizeof(float) * padded_n);
    float *hann = (float *)malloc(sizeof(float) * win);
    float *re = (float *)malloc(sizeof(float) * win), *im = (float *)malloc(sizeof(float) * win);
    bo_fft *plan = fft_new(win);
    bo_reflect_pad(signal, n, pad, padded);
    /* :37-39  0.5 - 0.5 * cos(2. * n as f32 * PI / W as f32), all in f32 */
    for (size_t k = 0; k < win; k++) hann[k] = 0.5f - 0.5f * cosf(2.0f * (float)k * PI_F / (float)win);
    memset(out, 0, sizeof(double) * rows * bins);
    /* :44-47  windows(win).step_by(hop) zipped with the rows: the shorter one ends the loop */
    const size_t n_windows = (padded_n >= win) ? (padded_n - win) / hop + 1 : 0;
    const size_t frames = n_windows < rows ? n_windows : rows;
    for (size_t f = 0; f < frames; f++) {
        const float *w = padded + f * hop;
        for (size_t k = 0; k < win; k++) { re[k] = w[k] * hann[k]; im[k] = 0.0f; }
        fft_forward(plan, re, im);
        double *o = out + f * bins;
        for (size_t k = 0; k < bins; k++) o[k] = (double)sqrtf(re[k] * re[k] + im[k] * im[k]); /* :60 */
    }
    fft_free(plan);
    free(padded); free(hann); free(re); free(im);
}

/* src/utils.rs:66-68 -- sequential f32 sum / len */
float bo_mean(const float *x, size_t n) {
    float s = 0.0f;
    for (size_t i = 0; i < n; i++) s += x[i];
    return s / (float)n;
}

/* ndarray 0.17 `std_axis(Axis(0), 0.)` on a 1-D f32 array (called at src/timbral.rs:61-63,
 * 85-87,115-117 and src/misc.rs:52): Welford, one fused mul_add per element, population std. */
float bo_std(const float *x, size_t n) {
    float mean = 0.0f, sum_sq = 0.0f;
    for (size_t i = 0; i < n; i++) {
        const float count = (float)(i + 1);
        const float delta = x[i] - mean;
        mean = mean + delta / count;
        sum_sq = fmaf(x[i] - mean, delta, sum_sq);
    }
    return sqrtf(sum_sq / ((float)n - 0.0f));
}

/* src/utils.rs:81-95 */
uint32_t bo_number_crossings(const float *x, size_t n) {
    uint32_t crossings = 0;
    int was_positive = x[0] > 0.0f;
    for (size_t i = 0; i < n; i++) {
        const int is_positive = x[i] > 0.0f;
        if (was_positive != is_positive) { crossings++; was_positive = is_positive; }
    }
    return crossings;
}

static inline uint64_t f64_bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static inline double bits_f64(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }

/* src/utils.rs:101-117 */
float bo_geometric_mean(const float *x, size_t n) {
    int32_t exponents = 0;
    double mantissas = 1.0;
    for (size_t c = 0; c + 8 <= n; c += 8) {
        const float *ch = x + c;
        double m = ((double)ch[0] * (double)ch[1]) * ((double)ch[2] * (double)ch[3]);
        m *= 3.273390607896142e150; /* 2^500 */
        m *= ((double)ch[4] * (double)ch[5]) * ((double)ch[6] * (double)ch[7]);
        if (m == 0.0) return 0.0f;
        exponents += (int32_t)(f64_bits(m) >> 52);
        mantissas *= bits_f64((f64_bits(m) & 0xFFFFFFFFFFFFFULL) | 0x3FF0000000000000ULL);
    }
    return exp2f((log2f((float)mantissas) + (float)exponents) / (float)(uint32_t)n - (1023.0f + 500.0f) / 8.0f);
}

/* ------------------------------------------------------------------------------------------
 * src/chroma.rs
 * ---------------------------------------------------------------------------------------- */

/* src/chroma.rs:197-267 (librosa.filters.chroma) */
void bo_chroma_filter(uint32_t sr, size_t n_fft, uint32_t n_chroma, double tuning, double *out) {
    const double ctroct = 5.0, octwidth = 2.0;
    const double ncf = (double)n_chroma;
    const double nc2 = (double)(uint32_t)round(ncf / 2.0);
    const size_t len = n_fft + 1, keep = 1 + n_fft / 2;
    double *fb = (double *)malloc(sizeof(double) * len);
    double *bw = (double *)malloc(sizeof(double) * len);
    double *wts = (double *)malloc(sizeof(double) * n_chroma * len);
    /* Array::linspace(0, sr, n_fft+1): a + step*i ; utils.rs:119-129 hz_to_octs_inplace */
    const double step = ((double)sr - 0.0) / (double)(len - 1);
    const double a440 = 440.0 * pow(2.0, tuning / (double)n_chroma);
    for (size_t i = 0; i < len; i++) {
        double f = 0.0 + step * (double)i;
        f /= a440 / 16.0;
        fb[i] = log2(f) * ncf;
    }
    fb[0] = fb[1] - 1.5 * ncf;
    for (size_t i = 0; i + 1 < len; i++) {
        const double d = fb[i + 1] - fb[i];
        bw[i] = (d <= 1.0) ? 1.0 : d;
    }
    bw[len - 1] = 1.0;
    for (uint32_t c = 0; c < n_chroma; c++)
        for (size_t i = 0; i < len; i++) {
            double d = -(double)c + fb[i];
            d = fmod(d + nc2 + 10.0 * ncf, ncf) - nc2;
            d = d / bw[i];
            wts[c * len + i] = exp(-0.5 * (2.0 * d) * (2.0 * d));
        }
    for (size_t i = 0; i < len; i++) {
        double s = 0.0;
        for (uint32_t c = 0; c < n_chroma; c++) s += wts[c * len + i] * wts[c * len + i];
        s = sqrt(s);
        if (s < DBL_MIN) s = 1.0;
        for (uint32_t c = 0; c < n_chroma; c++) wts[c * len + i] /= s;
    }
    for (size_t i = 0; i < len; i++) {
        const double y = (fb[i] / ncf - ctroct) / octwidth;
        const double g = exp(-0.5 * (y * y));
        for (uint32_t c = 0; c < n_chroma; c++) wts[c * len + i] *= g;
    }
    /* np.roll(wts, -3, axis=0), keep the first n_fft/2+1 columns */
    for (uint32_t r = 0; r < n_chroma; r++) {
        const uint32_t src = (r + 3) % n_chroma;
        memcpy(out + (size_t)r * keep, wts + (size_t)src * len, sizeof(double) * keep);
    }
    free(fb); free(bw); free(wts);
}

/* src/chroma.rs:269-331 */
size_t bo_pip_track(uint32_t sr, const double *spec, size_t frames, size_t n_fft, double *pitches, double *mags) {
    const double srf = (double)sr;
    const double fmin = 150.0, fmax = fmin > 0 ? (4000.0 < srf / 2.0 ? 4000.0 : srf / 2.0) : 0;
    const double threshold = 0.1;
    const size_t bins = 1 + n_fft / 2;
    const double fstep = (srf / 2.0 - 0.0) / (double)(bins - 1);
    long beginning = -1, end = -1;
    for (size_t k = 0; k < bins; k++) {
        const double f = 0.0 + fstep * (double)k;
        if (fmin <= f && f < fmax) { if (beginning < 0) beginning = (long)k; end = (long)k; }
    }
    if (beginning < 0) return 0;
    double *ref = (double *)malloc(sizeof(double) * frames);
    for (size_t j = 0; j < frames; j++) {
        const double *col = spec + j * bins;
        double mx = col[0];
        for (size_t k = 0; k < bins; k++) mx = (mx > col[k]) ? mx : col[k];
        ref[j] = threshold * mx;
    }
    size_t cnt = 0;
    /* Zip::indexed over (i, j): i (bin) outer, j (frame) inner; rows beginning..end-3 */
    for (long i = 0; i < (end - 3) - beginning; i++) {
        const size_t c = (size_t)(i + beginning + 1);
        for (size_t j = 0; j < frames; j++) {
            const double before = spec[j * bins + c - 1], elem = spec[j * bins + c], after = spec[j * bins + c + 1];
            if (elem > ref[j] && after <= elem && before < elem) {
                const double avg = 0.5 * (after - before);
                double shift = 2.0 * elem - after - before;
                if (fabs(shift) < DBL_MIN) shift += 1.0;
                shift = avg / shift;
                pitches[cnt] = ((double)c + shift) * srf / (double)n_fft;
                mags[cnt] = elem + 0.5 * avg * shift;
                cnt++;
            }
        }
    }
    free(ref);
    return cnt;
}

/* src/chroma.rs:334-359 */
double bo_pitch_tuning(double *freqs, size_t n, double resolution, uint32_t bins_per_octave) {
    if (n == 0) return 0.0;
    const size_t nbins = (size_t)((0.5 - -0.5) / resolution);
    size_t *counts = (size_t *)calloc(nbins, sizeof(size_t));
    const double a440 = 440.0 * pow(2.0, 0.0 / 12.0);
    for (size_t i = 0; i < n; i++) {
        double x = freqs[i] / (a440 / 16.0);
        x = log2(x);
        x = fmod((double)bins_per_octave * x, 1.0);
        if (x >= 0.5) x -= 1.0;
        freqs[i] = x;
        const double q = (x - -0.5) / resolution;
        size_t idx = (q > 0.0) ? (size_t)q : 0; /* `as usize` saturates */
        if (idx >= nbins) idx = nbins - 1;      /* the reference would panic here; never reached by its tests */
        counts[idx]++;
    }
    size_t best = 0; /* ndarray-stats argmax: first maximum */
    for (size_t k = 1; k < nbins; k++) if (counts[k] > counts[best]) best = k;
    free(counts);
    return (-50.0 + (100.0 * resolution * (double)best)) / 100.0;
}

static int cmp_f64(const void *a, const void *b) {
    const double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

/* ndarray-stats quantile(0.5, Midpoint): lower = floor(q*(n-1)), higher = ceil; lower + (higher-lower)/2 */
static double midpoint_median_f64(const double *v, size_t n) {
    double *tmp = (double *)malloc(sizeof(double) * n);
    memcpy(tmp, v, sizeof(double) * n);
    qsort(tmp, n, sizeof(double), cmp_f64);
    const double fi = 0.5 * (double)(n - 1);
    const size_t lo = (size_t)floor(fi), hi = (size_t)ceil(fi);
    const double r = tmp[lo] + (tmp[hi] - tmp[lo]) / 2.0;
    free(tmp);
    return r;
}

/* src/chroma.rs:361-391 */
double bo_estimate_tuning(uint32_t sr, const double *spec, size_t frames, size_t n_fft, double resolution,
                          uint32_t bins_per_octave) {
    const size_t bins = 1 + n_fft / 2;
    const size_t cap = frames * (bins / 2 + 1) + 1;
    double *pitch = (double *)malloc(sizeof(double) * cap), *mag = (double *)malloc(sizeof(double) * cap);
    const size_t n = bo_pip_track(sr, spec, frames, n_fft, pitch, mag);
    double result = 0.0;
    if (n > 0) {
        size_t m = 0; /* keep p > 0 */
        for (size_t i = 0; i < n; i++) if (pitch[i] > 0.0) { pitch[m] = pitch[i]; mag[m] = mag[i]; m++; }
        if (m > 0) {
            const double thr = midpoint_median_f64(mag, m);
            size_t k = 0;
            for (size_t i = 0; i < m; i++) if (mag[i] >= thr) pitch[k++] = pitch[i];
            result = bo_pitch_tuning(pitch, k, resolution, bins_per_octave);
        }
    }
    free(pitch); free(mag);
    return result;
}

/* src/chroma.rs:393-412 */
void bo_chroma_stft(uint32_t sr, double *spec, size_t frames, size_t n_fft, uint32_t n_chroma, double tuning,
                    double *out) {
    const size_t bins = 1 + n_fft / 2;
    double *filt = (double *)malloc(sizeof(double) * n_chroma * bins);
    for (size_t i = 0; i < frames * bins; i++) spec[i] = spec[i] * spec[i];
    bo_chroma_filter(sr, n_fft, n_chroma, tuning, filt);
    for (size_t j = 0; j < frames; j++) {
        const double *s = spec + j * bins;
        double sum = 0.0;
        for (uint32_t c = 0; c < n_chroma; c++) {
            const double *f = filt + (size_t)c * bins;
            double acc = 0.0;
            for (size_t k = 0; k < bins; k++) acc += f[k] * s[k];
            out[(size_t)c * frames + j] = acc;
            sum += fabs(acc);
        }
        if (sum < DBL_MIN) sum = 1.0;
        for (uint32_t c = 0; c < n_chroma; c++) out[(size_t)c * frames + j] /= sum;
    }
    free(filt);
}

/* src/chroma.rs:177-188 -- L1-normalise every column of a [rows][cols] array */
void bo_normalize_feature_sequence(double *feat, size_t rows, size_t cols) {
    for (size_t j = 0; j < cols; j++) {
        double sum = 0.0;
        for (size_t r = 0; r < rows; r++) sum += fabs(feat[r * cols + j]);
        if (sum < 0.0001) sum = 1.0;
        for (size_t r = 0; r < rows; r++) feat[r * cols + j] /= sum;
    }
}

/* templates of src/chroma.rs:139-152, as the pitch classes selected by each column */
static const int TEMPLATE_LEN[10] = {2, 2, 2, 2, 2, 2, 3, 3, 3, 3};
static const int TEMPLATE_PC[10][3] = {{0, 1, 0}, {0, 2, 0}, {0, 3, 0}, {0, 4, 0}, {0, 5, 0},
                                       {0, 6, 0}, {0, 4, 7}, {0, 3, 7}, {0, 3, 6}, {0, 4, 8}};

/* src/chroma.rs:157-175 */
void bo_extract_interval_features(const double *chroma, size_t frames, double *out) {
    for (int t = 0; t < 10; t++) {
        double *o = out + (size_t)t * frames;
        for (size_t j = 0; j < frames; j++) o[j] = 0.0;
        for (int shift = 0; shift < 12; shift++) {
            /* rotate_right(shift): template bit k moves to (k+shift)%12; the product runs over
             * ascending row index of the rolled vector (Array::product) */
            int rows[3];
            for (int e = 0; e < TEMPLATE_LEN[t]; e++) rows[e] = (TEMPLATE_PC[t][e] + shift) % 12;
            for (int a = 0; a < TEMPLATE_LEN[t]; a++)
                for (int b = a + 1; b < TEMPLATE_LEN[t]; b++)
                    if (rows[b] < rows[a]) { int x = rows[a]; rows[a] = rows[b]; rows[b] = x; }
            for (size_t j = 0; j < frames; j++) {
                double p = 1.0;
                for (int e = 0; e < TEMPLATE_LEN[t]; e++) p *= chroma[(size_t)rows[e] * frames + j];
                o[j] += p;
            }
        }
    }
}

/* src/chroma.rs:137-155 */
int bo_chroma_interval_features(const double *chroma, size_t frames, double out[10]) {
    if (frames == 0) return -1; /* AnalysisError("Tried to run the chroma descriptor on an empty array...") */
    double *c = (double *)malloc(sizeof(double) * 12 * frames);
    double *m = (double *)malloc(sizeof(double) * 10 * frames);
    for (size_t i = 0; i < 12 * frames; i++) c[i] = exp(chroma[i] * 15.0);
    bo_normalize_feature_sequence(c, 12, frames);
    bo_extract_interval_features(c, frames, m);
    for (int t = 0; t < 10; t++) {
        double s = 0.0;
        for (size_t j = 0; j < frames; j++) s += m[(size_t)t * frames + j];
        out[t] = s / (double)frames;
    }
    free(c); free(m);
    return 0;
}

/* src/chroma.rs:73-85 */
double *bo_chroma_desc_do(const float *signal, size_t n, size_t *frames_out, double *tuning_out) {
    const size_t win = 8192, hop = 2205, bins = win / 2 + 1;
    const size_t frames = bo_stft_frames(n, hop);
    double *spec = (double *)malloc(sizeof(double) * frames * bins);
    bo_stft(signal, n, win, hop, spec);
    const double tuning = bo_estimate_tuning(BO_SAMPLE_RATE, spec, frames, win, 0.01, 12);
    double *chroma = (double *)malloc(sizeof(double) * 12 * frames);
    bo_chroma_stft(BO_SAMPLE_RATE, spec, frames, win, 12, tuning, chroma);
    free(spec);
    *frames_out = frames;
    if (tuning_out) *tuning_out = tuning;
    return chroma;
}

/* src/chroma.rs:97-126 */
void bo_chroma_get_values(const double *chroma, size_t frames, float out[13]) {
    double raw[10];
    bo_chroma_interval_features(chroma, frames, raw);
    double n1 = 0.0, n2 = 0.0;
    for (int i = 0; i < 6; i++) n1 += raw[i] * raw[i];
    for (int i = 6; i < 10; i++) n2 += raw[i] * raw[i];
    n1 = sqrt(n1); n2 = sqrt(n2);
    if (n1 > 0.0) for (int i = 0; i < 6; i++) raw[i] /= n1;
    if (n2 > 0.0) for (int i = 6; i < 10; i++) raw[i] /= n2;
    for (int i = 0; i < 10; i++) out[i] = 2.0f * ((float)raw[i] - 0.0f) / (1.0f - 0.0f) - 1.0f; /* Normalize, :33-36 */
    out[10] = fminf(2.0f * ((float)n1 - 0.0f) / (0.25f - 0.0f) - 1.0f, 1.0f);
    out[11] = fminf(2.0f * ((float)n2 - 0.0f) / (0.025f - 0.0f) - 1.0f, 1.0f);
    const double angle = atan2(20.0 * n2, n1 + 1e-12);
    out[12] = 2.0f * ((float)angle - 0.0f) / (1.57079632679489661923f - 0.0f) - 1.0f;
}

/* src/chroma.rs:128-132 */
void bo_chroma_get_values_v1(const double *chroma, size_t frames, float out[10]) {
    double raw[10];
    bo_chroma_interval_features(chroma, frames, raw);
    for (int i = 0; i < 10; i++) out[i] = 2.0f * ((float)raw[i] - 0.0f) / (0.12f - 0.0f) - 1.0f;
}

/* ------------------------------------------------------------------------------------------
 * src/aubio.rs -- phase vocoder (PVoc :119-265, PVocTempo :274-426)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    size_t win_s, hop_s;
    float *data, *dataold, *window, *re, *im;
    bo_fft *fft;
} bo_pvoc;

static void pvoc_init(bo_pvoc *p, size_t win_s, size_t hop_s) {
    p->win_s = win_s; p->hop_s = hop_s;
    p->data = (float *)calloc(win_s, sizeof(float));
    p->dataold = (float *)calloc(win_s > hop_s ? win_s - hop_s : 1, sizeof(float));
    p->window = (float *)malloc(sizeof(float) * win_s);
    p->re = (float *)malloc(sizeof(float) * win_s);
    p->im = (float *)malloc(sizeof(float) * win_s);
    /* :151-154 hanningz: 0.5 * (1.0 - cos(2.0 * PI * i as f32 / win_s as f32)) */
    for (size_t i = 0; i < win_s; i++) p->window[i] = 0.5f * (1.0f - cosf(2.0f * PI_F * (float)i / (float)win_s));
    p->fft = fft_new(win_s);
}

static void pvoc_release(bo_pvoc *p) {
    free(p->data); free(p->dataold); free(p->window); free(p->re); free(p->im);
    fft_free(p->fft);
}

/* shared front half of PVoc::do_ (:198-235) and PVocTempo::do_ (:354-398): slide, window,
 * fftshift, c2c FFT.  Leaves the spectrum in p->re / p->im. */
static void pvoc_spectrum(bo_pvoc *p, const float *input) {
    const size_t end = p->win_s - p->hop_s;
    for (size_t i = 0; i < end; i++) p->data[i] = p->dataold[i];
    memcpy(p->data + end, input, sizeof(float) * p->hop_s);
    for (size_t i = 0; i < end; i++) p->dataold[i] = p->data[i + p->hop_s];
    for (size_t i = 0; i < p->win_s; i++) p->data[i] *= p->window[i];
    const size_t half = p->win_s / 2; /* win_s even on this path */
    for (size_t j = 0; j < half; j++) { float t = p->data[j]; p->data[j] = p->data[j + half]; p->data[j + half] = t; }
    for (size_t i = 0; i < p->win_s; i++) { p->re[i] = p->data[i]; p->im[i] = 0.0f; }
    fft_forward(p->fft, p->re, p->im);
}

/* PVoc::do_ :237-261 -- the "buggy" 256-bin packing: bin 255 := |Re X[256]| */
static void pvoc_norms_buggy(const bo_pvoc *p, float *norm /* win_s/2 */) {
    const size_t nb = p->win_s / 2;
    norm[0] = fabsf(p->re[0]);
    for (size_t i = 1; i + 1 < nb; i++) norm[i] = sqrtf(p->re[i] * p->re[i] + p->im[i] * p->im[i]);
    norm[nb - 1] = fabsf(p->re[p->win_s / 2]);
}

/* PVocTempo::do_ :400-422 -- correct win_s/2+1 bins */
static void pvoc_norms_full(const bo_pvoc *p, float *norm /* win_s/2+1 */) {
    const size_t nb = p->win_s / 2 + 1;
    norm[0] = fabsf(p->re[0]);
    for (size_t i = 1; i + 1 < nb; i++) norm[i] = sqrtf(p->re[i] * p->re[i] + p->im[i] * p->im[i]);
    norm[nb - 1] = fabsf(p->re[p->win_s / 2]);
}

void bo_pvoc512_norms(const float *window512, float norms256[256], float norms257[257]) {
    bo_pvoc p;
    pvoc_init(&p, 512, 128);
    /* feed the 512 samples as four hops so the sliding buffer ends up holding exactly them */
    for (int h = 0; h < 4; h++) pvoc_spectrum(&p, window512 + 128 * h);
    if (norms256) pvoc_norms_buggy(&p, norms256);
    if (norms257) pvoc_norms_full(&p, norms257);
    pvoc_release(&p);
}

/* ---- src/aubio.rs:16-71 ---- */
static float spectral_centroid(const float *norm, size_t n) {
    float sum = 0.0f;
    for (size_t j = 0; j < n; j++) sum += norm[j];
    if (sum == 0.0f) return 0.0f;
    float sc = 0.0f;
    for (size_t j = 0; j < n; j++) sc += (float)j * norm[j];
    return sc / sum;
}

static float spectral_rolloff(const float *norm, size_t n) {
    float cumsum = 0.0f, rollsum = 0.0f;
    for (size_t j = 0; j < n; j++) cumsum += norm[j] * norm[j];
    if (cumsum == 0.0f) return 0.0f;
    cumsum *= 0.95f;
    size_t j = 0;
    while (rollsum < cumsum && j < n) { rollsum += norm[j] * norm[j]; j++; }
    return (float)j;
}

static float bin_to_freq(float bin, float sample_rate, float fft_size) {
    const float freq = sample_rate / fft_size;
    return freq * fmaxf(bin, 0.0f);
}

/* ------------------------------------------------------------------------------------------
 * src/timbral.rs:27-209 SpectralDesc
 * ---------------------------------------------------------------------------------------- */
typedef struct { float *v; size_t n, cap; } fvec;
static void fvec_push(fvec *a, float x) {
    if (a->n == a->cap) { a->cap = a->cap ? a->cap * 2 : 1024; a->v = (float *)realloc(a->v, a->cap * sizeof(float)); }
    a->v[a->n++] = x;
}

struct bo_spectral_desc {
    bo_pvoc pv;
    uint32_t sample_rate;
    float norm[256];
    fvec centroid, rolloff, flatness;
};

bo_spectral_desc *bo_spectral_desc_new(uint32_t sr) {
    bo_spectral_desc *d = (bo_spectral_desc *)calloc(1, sizeof *d);
    pvoc_init(&d->pv, 512, 128);
    d->sample_rate = sr;
    return d;
}

/* src/timbral.rs:154-209 */
void bo_spectral_desc_do(bo_spectral_desc *d, const float *chunk) {
    pvoc_spectrum(&d->pv, chunk);
    pvoc_norms_buggy(&d->pv, d->norm);
    float bin = spectral_centroid(d->norm, 256);
    fvec_push(&d->centroid, bin_to_freq(bin, (float)d->sample_rate, 512.0f));
    bin = spectral_rolloff(d->norm, 256);
    if (bin > 512.0f / 2.0f) bin = 512.0f / 2.0f;
    fvec_push(&d->rolloff, bin_to_freq(bin, (float)d->sample_rate, 512.0f));
    const float geo = bo_geometric_mean(d->norm, 256);
    if (geo == 0.0f) { fvec_push(&d->flatness, 0.0f); return; }
    fvec_push(&d->flatness, geo / bo_mean(d->norm, 256));
}

/* Normalize (src/utils.rs:70-77) with (min, max) */
static float normalize(float v, float mn, float mx) { return 2.0f * (v - mn) / (mx - mn) - 1.0f; }

/* src/timbral.rs:57-122 */
void bo_spectral_desc_get(bo_spectral_desc *d, float centroid[2], float rolloff[2], float flatness[2]) {
    const float mx = (float)BO_SAMPLE_RATE / 2.0f;
    centroid[0] = normalize(bo_mean(d->centroid.v, d->centroid.n), 0.0f, mx);
    centroid[1] = normalize(bo_std(d->centroid.v, d->centroid.n), 0.0f, mx);
    rolloff[0] = normalize(bo_mean(d->rolloff.v, d->rolloff.n), 0.0f, mx);
    rolloff[1] = normalize(bo_std(d->rolloff.v, d->rolloff.n), 0.0f, mx);
    flatness[0] = 2.0f * (bo_mean(d->flatness.v, d->flatness.n) - 0.0f) / (1.0f - 0.0f) - 1.0f;
    flatness[1] = 2.0f * (bo_std(d->flatness.v, d->flatness.n) - 0.0f) / (1.0f - 0.0f) - 1.0f;
}

size_t bo_spectral_desc_series(bo_spectral_desc *d, const float **c, const float **r, const float **f) {
    if (c) *c = d->centroid.v;
    if (r) *r = d->rolloff.v;
    if (f) *f = d->flatness.v;
    return d->centroid.n;
}

void bo_spectral_desc_free(bo_spectral_desc *d) {
    if (!d) return;
    pvoc_release(&d->pv);
    free(d->centroid.v); free(d->rolloff.v); free(d->flatness.v);
    free(d);
}

/* ------------------------------------------------------------------------------------------
 * src/aubio.rs tempo part
 * ---------------------------------------------------------------------------------------- */
static float vec_mean(const float *x, size_t n) { /* :472-478 */
    if (n == 0) return 0.0f;
    float s = 0.0f;
    for (size_t i = 0; i < n; i++) s += x[i];
    return s / (float)n;
}

#define FSWAP(a, b) do { float t_ = (a); (a) = (b); (b) = t_; } while (0)

/* :482-554 quickselect median (element (n-1)/2 of the sorted data) */
static float vec_median(float *data, size_t n) {
    if (n == 0) return 0.0f;
    size_t low = 0, high = n - 1;
    const size_t median = (low + high) / 2;
    for (;;) {
        if (high <= low) return data[median];
        if (high == low + 1) {
            if (data[low] > data[high]) FSWAP(data[low], data[high]);
            return data[median];
        }
        const size_t middle = (low + high) / 2;
        if (data[middle] > data[high]) FSWAP(data[middle], data[high]);
        if (data[low] > data[high]) FSWAP(data[low], data[high]);
        if (data[middle] > data[low]) FSWAP(data[middle], data[low]);
        FSWAP(data[middle], data[low + 1]);
        size_t ll = low + 1, hh = high;
        for (;;) {
            do ll++; while (data[low] > data[ll]);
            do hh--; while (data[hh] > data[low]);
            if (hh < ll) break;
            FSWAP(data[ll], data[hh]);
        }
        FSWAP(data[low], data[hh]);
        if (hh <= median) low = ll;
        if (hh >= median) high = hh - 1;
    }
}

/* :576-604 */
static float vec_quadratic_peak_pos(const float *x, size_t len, size_t pos) {
    if (pos == 0 || pos >= len - 1) return (float)pos;
    const float s0 = x[pos - 1], s1 = x[pos], s2 = x[pos + 1];
    return (float)pos + 0.5f * (s0 - s2) / (s0 - 2.0f * s1 + s2);
}

/* :608-686 biquad + filtfilt with state reset after each pass */
typedef struct { float b0, b1, b2, a1, a2, x1, x2, y1, y2; } bo_biquad;
static float biquad_sample(bo_biquad *q, float x0) {
    const float y0 = q->b0 * x0 + q->b1 * q->x1 + q->b2 * q->x2 - q->a1 * q->y1 - q->a2 * q->y2;
    q->x2 = q->x1; q->x1 = x0; q->y2 = q->y1; q->y1 = y0;
    return y0;
}
static void biquad_reset(bo_biquad *q) { q->x1 = q->x2 = q->y1 = q->y2 = 0.0f; }
static void biquad_filtfilt(bo_biquad *q, float *data, float *tmp, size_t n) {
    for (size_t i = 0; i < n; i++) data[i] = biquad_sample(q, data[i]);
    biquad_reset(q);
    for (size_t i = 0; i < n; i++) tmp[n - i - 1] = data[i];
    for (size_t i = 0; i < n; i++) tmp[i] = biquad_sample(q, tmp[i]);
    biquad_reset(q);
    for (size_t i = 0; i < n; i++) data[i] = tmp[n - i - 1];
}

/* :692-779 PeakPicker (win_post 5, win_pre 1 => 7 taps) */
typedef struct {
    float threshold;
    bo_biquad biquad;
    float onset_keep[7], onset_proc[7], scratch[7], onset_peek[3], thresholded;
} bo_peakpicker;

static void peakpicker_init(bo_peakpicker *p) {
    memset(p, 0, sizeof *p);
    p->threshold = 0.1f;
    p->biquad.b0 = 0.1599879f; p->biquad.b1 = 0.31997577f; p->biquad.b2 = 0.1599879f;
    p->biquad.a1 = 0.23484048f; p->biquad.a2 = 0.0f;
}

static float peakpicker_do(bo_peakpicker *p, float onset) {
    for (int i = 0; i < 6; i++) p->onset_keep[i] = p->onset_keep[i + 1];
    p->onset_keep[6] = onset;
    memcpy(p->onset_proc, p->onset_keep, sizeof p->onset_proc);
    biquad_filtfilt(&p->biquad, p->onset_proc, p->scratch, 7);
    const float mean = vec_mean(p->onset_proc, 7);
    memcpy(p->scratch, p->onset_proc, sizeof p->scratch);
    const float median = vec_median(p->scratch, 7);
    p->onset_peek[0] = p->onset_peek[1];
    p->onset_peek[1] = p->onset_peek[2];
    p->thresholded = p->onset_proc[5] - median - mean * p->threshold;
    p->onset_peek[2] = p->thresholded;
    /* vec_peakpick(peek, 1) :567-572 */
    if (p->onset_peek[1] > p->onset_peek[0] && p->onset_peek[1] > p->onset_peek[2] && p->onset_peek[1] > 0.0f)
        return vec_quadratic_peak_pos(p->onset_peek, 3, 1);
    return 0.0f;
}

/* :787-799 -- LAST index attaining the max, starting from tmp = 0 */
static size_t vec_max_elem(const float *x, size_t n) {
    size_t pos = 0;
    float tmp = 0.0f;
    for (size_t j = 0; j < n; j++) if (tmp <= x[j]) { pos = j; tmp = x[j]; }
    return pos;
}

/* :819-828 */
static void vec_autocorr(const float *in, float *out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        float tmp = 0.0f;
        for (size_t j = i; j < n; j++) tmp += in[j - i] * in[j];
        out[i] = tmp / (float)(n - i);
    }
}

/* :834-1240 BeatTracking */
typedef struct {
    size_t hop_size, winlen, laglen, step;
    uint32_t samplerate, timesig, rayparam, flagstep;
    int32_t counter;
    float *rwv, *gwv, *dfwv, *dfrev, *acf, *acfout, *phwv, *phout;
    float lastbeat, g_var, gp, bp, rp, rp1, rp2;
} bo_beattracking;

static void beattracking_init(bo_beattracking *b, size_t winlen, size_t hop_size, uint32_t samplerate) {
    memset(b, 0, sizeof *b);
    const float rayparam_float = 60.0f * (float)samplerate / 120.0f / (float)hop_size;
    b->rayparam = (uint32_t)rayparam_float;
    const float dfwvnorm = expf((logf(2.0f) / rayparam_float) * (float)(winlen + 2));
    b->hop_size = hop_size; b->samplerate = samplerate; b->winlen = winlen;
    b->laglen = winlen / 4; b->step = winlen / 4;
    b->rwv = (float *)calloc(b->laglen, sizeof(float));
    b->gwv = (float *)calloc(b->laglen, sizeof(float));
    b->dfwv = (float *)calloc(winlen, sizeof(float));
    b->dfrev = (float *)calloc(winlen, sizeof(float));
    b->acf = (float *)calloc(winlen, sizeof(float));
    b->acfout = (float *)calloc(b->laglen, sizeof(float));
    b->phwv = (float *)calloc(2 * b->laglen, sizeof(float));
    b->phout = (float *)calloc(winlen, sizeof(float));
    for (size_t i = 0; i < b->laglen; i++) {
        const float i_f = (float)(i + 1);
        b->rwv[i] = (i_f / (rayparam_float * rayparam_float)) *
                    expf(-(i_f * i_f) / (2.0f * (rayparam_float * rayparam_float)));
    }
    for (size_t i = 0; i < winlen; i++) b->dfwv[i] = expf((logf(2.0f) / rayparam_float) * (float)(i + 1)) / dfwvnorm;
    for (size_t i = 0; i < 2 * b->laglen; i++) b->phwv[i] = 1.0f;
    b->g_var = 3.901f;
    b->rp = 1.0f;
}

static void beattracking_release(bo_beattracking *b) {
    free(b->rwv); free(b->gwv); free(b->dfwv); free(b->dfrev); free(b->acf); free(b->acfout); free(b->phwv); free(b->phout);
}

/* :864-907 */
static uint32_t get_timesig(const float *acf, size_t acflen, size_t gp) {
    if (gp < 2) return 4;
    float three = 0.0f, four = 0.0f;
    if (acflen > 6 * gp + 2) {
        for (int k = -2; k < 2; k++) {
            three += acf[(size_t)(3 * (long)gp + k)];
            four += acf[(size_t)(4 * (long)gp + k)];
        }
    } else {
        for (int k = -2; k < 2; k++) {
            const size_t i3 = (size_t)(3 * (long)gp + k), i6 = (size_t)(6 * (long)gp + k);
            const size_t i4 = (size_t)(4 * (long)gp + k), i2 = (size_t)(2 * (long)gp + k);
            if (i3 < acflen && i6 < acflen) three += acf[i3] + acf[i6];
            else if (i3 < acflen) three += acf[i3];
            if (i4 < acflen && i2 < acflen) four += acf[i4] + acf[i2];
            else if (i4 < acflen) four += acf[i4];
        }
    }
    return three > four ? 3 : 4;
}

/* :1096-1227 */
static void beattracking_checkstate(bo_beattracking *b) {
    const size_t laglen = b->laglen, acflen = b->winlen, step = b->step;
    int32_t counter = b->counter;
    uint32_t flagstep = b->flagstep;
    float gp = b->gp;
    const float rp = b->rp;
    float rp1 = b->rp1, rp2 = b->rp2, bp;
    int flagconst = 0;
    if (gp > 0.0f) {
        for (size_t i = 0; i < laglen; i++) b->acfout[i] = 0.0f;
        for (size_t i = 1; i < laglen - 1; i++)
            for (uint32_t a = 1; a <= b->timesig; a++)
                for (uint32_t bb = 1; bb < 2 * a; bb++) {
                    const size_t idx = i * a + bb - 1;
                    if (idx < acflen) b->acfout[i] += b->acf[idx];
                }
        for (size_t i = 0; i < laglen; i++) b->acfout[i] *= b->gwv[i];
        const size_t maxindex = vec_max_elem(b->acfout, laglen);
        gp = vec_quadratic_peak_pos(b->acfout, laglen, maxindex);
    } else {
        gp = 0.0f;
    }
    if (counter == 0) {
        if (fabsf(gp - rp) > 2.0f * b->g_var) { flagstep = 1; counter = 3; }
        else flagstep = 0;
    }
    if (counter == 1 && flagstep == 1) {
        if (fabsf(2.0f * rp - rp1 - rp2) < b->g_var) { flagconst = 1; counter = 0; }
        else { flagconst = 0; counter = 2; }
    } else if (counter > 0) {
        counter -= 1;
    }
    rp2 = rp1;
    rp1 = rp;
    if (flagconst) {
        gp = rp;
        b->timesig = get_timesig(b->acf, acflen, (size_t)gp);
        for (size_t j = 0; j < laglen; j++) {
            const float diff = (float)(j + 1) - gp;
            b->gwv[j] = expf(-0.5f * diff * diff / (b->g_var * b->g_var));
        }
        bp = gp;
        for (size_t j = 0; j < 2 * laglen; j++) b->phwv[j] = 1.0f;
    } else if (b->timesig > 0) {
        bp = gp;
        if ((float)step > b->lastbeat) {
            for (size_t j = 0; j < 2 * laglen; j++) {
                const float diff = 1.0f + (float)j - (float)step + b->lastbeat;
                b->phwv[j] = expf(-0.5f * diff * diff / (bp / 8.0f));
            }
        } else {
            for (size_t j = 0; j < 2 * laglen; j++) b->phwv[j] = 1.0f;
        }
    } else {
        bp = rp;
        for (size_t j = 0; j < 2 * laglen; j++) b->phwv[j] = 1.0f;
    }
    while (bp > 0.0f && bp < 25.0f) bp *= 2.0f;
    b->counter = counter; b->flagstep = flagstep; b->gp = gp; b->bp = bp; b->rp1 = rp1; b->rp2 = rp2;
}

/* :966-1092 ; output has `step` slots */
static void beattracking_do(bo_beattracking *b, const float *dfframe, float *output) {
    const size_t step = b->step, laglen = b->laglen, winlen = b->winlen, outlen = b->step;
    const size_t numelem = b->timesig == 0 ? 4 : b->timesig;
    for (size_t i = 0; i < winlen; i++) b->dfrev[i] = dfframe[i] * b->dfwv[i];
    for (size_t j = 0; j < winlen / 2; j++) FSWAP(b->dfrev[j], b->dfrev[winlen - 1 - j]);
    vec_autocorr(dfframe, b->acf, winlen);
    for (size_t i = 0; i < laglen; i++) b->acfout[i] = 0.0f;
    for (size_t i = 1; i < laglen - 1; i++)
        for (size_t a = 1; a <= numelem; a++)
            for (size_t bb = 1; bb < 2 * a; bb++)
                if (i * a + bb - 1 < winlen) b->acfout[i] += b->acf[i * a + bb - 1] / (2.0f * (float)a - 1.0f);
    for (size_t i = 0; i < laglen; i++) b->acfout[i] *= b->rwv[i];
    size_t maxindex = vec_max_elem(b->acfout, laglen);
    if (maxindex > 0 && maxindex < laglen - 1) b->rp = vec_quadratic_peak_pos(b->acfout, laglen, maxindex);
    else b->rp = (float)b->rayparam;
    beattracking_checkstate(b);
    const float bp = b->bp;
    if (bp == 0.0f) { for (size_t i = 0; i < outlen; i++) output[i] = 0.0f; return; }
    const size_t kmax = (size_t)floorf((float)winlen / bp);
    for (size_t i = 0; i < winlen; i++) b->phout[i] = 0.0f;
    for (size_t i = 0; (float)i < bp && i < winlen; i++)
        for (size_t k = 0; k < kmax; k++) {
            const size_t idx = i + (size_t)floorf((bp * (float)k) + 0.5f);
            if (idx < winlen) b->phout[i] += b->dfrev[idx];
        }
    /* vec_weight: length = min(phout.len, phwv.len) = 2*laglen */
    for (size_t i = 0; i < 2 * laglen && i < winlen; i++) b->phout[i] *= b->phwv[i];
    maxindex = vec_max_elem(b->phout, winlen);
    float phase;
    if (maxindex >= winlen - 1) phase = (float)step - b->lastbeat;
    else phase = vec_quadratic_peak_pos(b->phout, winlen, maxindex);
    phase += 1.0f;
    for (size_t i = 0; i < outlen; i++) output[i] = 0.0f;
    size_t i = 1;
    float beat = bp - phase;
    if (((float)step - b->lastbeat - phase) < -0.40f * bp) beat += bp;
    while (beat + bp < 0.0f) beat += bp;
    if (beat >= 0.0f && i < outlen) { output[i] = beat; i++; }
    while (beat + bp <= (float)step && i < outlen) { beat += bp; output[i] = beat; i++; }
    b->lastbeat = beat;
    output[0] = (float)i;
}

static float beattracking_get_bpm(const bo_beattracking *b) { /* :1231-1239 */
    if (b->bp != 0.0f) {
        const float period_samples = (float)b->hop_size * b->bp;
        const float period_s = period_samples / (float)b->samplerate;
        return 60.0f / period_s;
    }
    return 0.0f;
}

/* :1258-1276 */
static float level_lin(const float *x, size_t n) {
    float e = 0.0f;
    for (size_t i = 0; i < n; i++) e += x[i] * x[i];
    return e / (float)n;
}
static int is_silence(const float *x, size_t n, float threshold) { return 10.0f * log10f(level_lin(x, n)) < threshold; }

/* :1284-1450 Tempo ; src/temporal.rs:32-85 BPMDesc */
struct bo_bpm_desc {
    bo_pvoc pv;
    float oldmag[257], norm[257];
    bo_peakpicker pp;
    bo_beattracking bt;
    float *dfframe, *out;
    float silence;
    long blockpos;
    size_t winlen, step, hop_size;
    fvec bpms, onset_series, thresholded_series;
};

bo_bpm_desc *bo_bpm_desc_new(uint32_t sr) {
    const size_t buf_size = 512, hop_size = 256;
    if (sr < 1) return NULL; /* "error while loading aubio tempo object: creation error" */
    bo_bpm_desc *d = (bo_bpm_desc *)calloc(1, sizeof *d);
    size_t winlen = 1, want = (size_t)((5.8f * (float)sr) / (float)hop_size);
    while (winlen < want) winlen <<= 1;
    if (winlen < 4) winlen = 4;
    d->winlen = winlen; d->step = winlen / 4; d->hop_size = hop_size;
    pvoc_init(&d->pv, buf_size, hop_size);
    peakpicker_init(&d->pp);
    d->pp.threshold = 0.3f; /* :1347 */
    beattracking_init(&d->bt, winlen, hop_size, sr);
    d->dfframe = (float *)calloc(winlen, sizeof(float));
    d->out = (float *)calloc(d->step, sizeof(float));
    d->silence = -90.0f;
    d->blockpos = 0;
    return d;
}

/* Tempo::do_ :1378-1443 followed by BPMDesc::do_ (src/temporal.rs:50-58) */
void bo_bpm_desc_do(bo_bpm_desc *d, const float *chunk, size_t chunk_len) {
    const size_t winlen = d->winlen, step = d->step;
    pvoc_spectrum(&d->pv, chunk);
    pvoc_norms_full(&d->pv, d->norm);
    float of = 0.0f; /* SpecFlux :455-467 */
    for (size_t j = 0; j < 257; j++) {
        if (d->norm[j] > d->oldmag[j]) of += d->norm[j] - d->oldmag[j];
        d->oldmag[j] = d->norm[j];
    }
    fvec_push(&d->onset_series, of);
    if (d->blockpos == (long)step - 1) {
        beattracking_do(&d->bt, d->dfframe, d->out);
        for (size_t i = 0; i < winlen - step; i++) d->dfframe[i] = d->dfframe[i + step];
        for (size_t i = winlen - step; i < winlen; i++) d->dfframe[i] = 0.0f;
        d->blockpos = -1;
    }
    d->blockpos += 1;
    peakpicker_do(&d->pp, of);
    const float thresholded = d->pp.thresholded;
    fvec_push(&d->thresholded_series, thresholded);
    d->dfframe[winlen - step + (size_t)d->blockpos] = thresholded;
    float tempo_out = 0.0f;
    const size_t num_beats = (size_t)d->out[0];
    for (size_t i = 1; i < num_beats; i++) {
        const float beat_pos = d->out[i];
        if (d->blockpos == (long)floorf(beat_pos)) {
            tempo_out = beat_pos - floorf(beat_pos);
            if (is_silence(chunk, chunk_len, d->silence)) tempo_out = 0.0f;
        }
    }
    if (tempo_out > 0.0f) fvec_push(&d->bpms, beattracking_get_bpm(&d->bt));
}

static int cmp_f32(const void *a, const void *b) {
    const float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}

/* src/temporal.rs:66-77 -- Midpoint median over n32 values, then Normalize(0, 206) */
float bo_bpm_desc_get_value(bo_bpm_desc *d) {
    if (d->bpms.n == 0) return -1.0f;
    const size_t n = d->bpms.n;
    float *tmp = (float *)malloc(sizeof(float) * n);
    memcpy(tmp, d->bpms.v, sizeof(float) * n);
    qsort(tmp, n, sizeof(float), cmp_f32);
    const double fi = 0.5 * (double)(n - 1);
    const size_t lo = (size_t)floor(fi), hi = (size_t)ceil(fi);
    const float median = tmp[lo] + (tmp[hi] - tmp[lo]) / 2.0f;
    free(tmp);
    return normalize(median, 0.0f, 206.0f);
}

size_t bo_bpm_desc_bpms(bo_bpm_desc *d, const float **bpms) { if (bpms) *bpms = d->bpms.v; return d->bpms.n; }
size_t bo_bpm_desc_series(bo_bpm_desc *d, const float **onset, const float **thresholded) {
    if (onset) *onset = d->onset_series.v;
    if (thresholded) *thresholded = d->thresholded_series.v;
    return d->onset_series.n;
}

void bo_bpm_desc_free(bo_bpm_desc *d) {
    if (!d) return;
    pvoc_release(&d->pv);
    beattracking_release(&d->bt);
    free(d->dfframe); free(d->out); free(d->bpms.v); free(d->onset_series.v); free(d->thresholded_series.v);
    free(d);
}

/* ------------------------------------------------------------------------------------------
 * src/misc.rs:39-71 LoudnessDesc ; src/timbral.rs:231-258 ZeroCrossingRateDesc
 * ---------------------------------------------------------------------------------------- */
void bo_loudness(const float *x, size_t n, int chunks_exact, float out[2]) {
    const size_t W = 1024;
    const size_t nchunks = chunks_exact ? n / W : (n + W - 1) / W;
    float *values = (float *)malloc(sizeof(float) * (nchunks ? nchunks : 1));
    for (size_t c = 0; c < nchunks; c++) {
        const size_t len = (c * W + W <= n) ? W : n - c * W;
        values[c] = level_lin(x + c * W, len); /* src/misc.rs:12-18 */
    }
    float std_value = bo_std(values, nchunks);
    float mean_value = bo_mean(values, nchunks);
    if (mean_value < 1e-9f) mean_value = 1e-9f;
    if (std_value < 1e-9f) std_value = 1e-9f;
    out[0] = normalize(10.0f * log10f(mean_value), -90.0f, 0.0f);
    out[1] = normalize(10.0f * log10f(std_value), -90.0f, 0.0f);
    free(values);
}

float bo_zcr(const float *x, size_t n) {
    const uint32_t c = bo_number_crossings(x, n);
    return normalize((float)c / (float)n, 0.0f, 1.0f);
}

/* ------------------------------------------------------------------------------------------
 * src/song/mod.rs:413-508 Song::analyze_with_options
 * ---------------------------------------------------------------------------------------- */
int bo_song_analyze(const float *x, size_t n, uint32_t features_version, float *out) {
    if (features_version != 1 && features_version != 2) return BO_ERR_VERSION;
    if (n < 8192) return BO_ERR_TOO_SHORT; /* max(512, 8192, 512, 1024), :417-430 */
    /* tempo: windows(512).step_by(256), :433-443 */
    bo_bpm_desc *bpm = bo_bpm_desc_new(BO_SAMPLE_RATE);
    for (size_t s = 0; s + 512 <= n; s += 256) bo_bpm_desc_do(bpm, x + s, 512);
    const float tempo = bo_bpm_desc_get_value(bpm);
    bo_bpm_desc_free(bpm);
    /* timbral: windows(512).step_by(128), :456-468 */
    bo_spectral_desc *sd = bo_spectral_desc_new(BO_SAMPLE_RATE);
    for (size_t s = 0; s + 512 <= n; s += 128) bo_spectral_desc_do(sd, x + s);
    float centroid[2], rolloff[2], flatness[2];
    bo_spectral_desc_get(sd, centroid, rolloff, flatness);
    bo_spectral_desc_free(sd);
    const float zcr = bo_zcr(x, n);       /* :470-474, one call on the whole song */
    float loud[2];
    bo_loudness(x, n, 0, loud);           /* :476-484, chunks(1024) incl. the partial tail */
    size_t frames;
    double *chroma = bo_chroma_desc_do(x, n, &frames, NULL); /* :445-453 */
    float ch[13];
    if (features_version == 1) bo_chroma_get_values_v1(chroma, frames, ch);
    else bo_chroma_get_values(chroma, frames, ch);
    free(chroma);
    /* :493-498 */
    out[0] = tempo; out[1] = zcr;
    out[2] = centroid[0]; out[3] = centroid[1];
    out[4] = rolloff[0]; out[5] = rolloff[1];
    out[6] = flatness[0]; out[7] = flatness[1];
    out[8] = loud[0]; out[9] = loud[1];
    const int nch = features_version == 1 ? 10 : 13;
    for (int i = 0; i < nch; i++) out[10 + i] = ch[i];
    return BO_OK;
}

/* The same analysis with a wall-clock timer around each descriptor (SURVEY.md 8d "where time goes today" on the CPU):
 * secs[0..4] = tempo, timbral, zero-crossing rate, loudness, chroma.  Test / bench instrumentation, no reference
 * counterpart; the feature row is the one bo_song_analyze returns. */
static double now_seconds(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int bo_song_analyze_timed(const float *x, size_t n, uint32_t features_version, float *out, double secs[5]) {
    if (features_version != 1 && features_version != 2) return BO_ERR_VERSION;
    if (n < 8192) return BO_ERR_TOO_SHORT;
    double t0 = now_seconds();
    bo_bpm_desc *bpm = bo_bpm_desc_new(BO_SAMPLE_RATE);
    for (size_t s = 0; s + 512 <= n; s += 256) bo_bpm_desc_do(bpm, x + s, 512);
    out[0] = bo_bpm_desc_get_value(bpm);
    bo_bpm_desc_free(bpm);
    double t1 = now_seconds(); secs[0] = t1 - t0; t0 = t1;
    bo_spectral_desc *sd = bo_spectral_desc_new(BO_SAMPLE_RATE);
    for (size_t s = 0; s + 512 <= n; s += 128) bo_spectral_desc_do(sd, x + s);
    bo_spectral_desc_get(sd, out + 2, out + 4, out + 6);
    bo_spectral_desc_free(sd);
    t1 = now_seconds(); secs[1] = t1 - t0; t0 = t1;
    out[1] = bo_zcr(x, n);
    t1 = now_seconds(); secs[2] = t1 - t0; t0 = t1;
    bo_loudness(x, n, 0, out + 8);
    t1 = now_seconds(); secs[3] = t1 - t0; t0 = t1;
    size_t frames;
    double *chroma = bo_chroma_desc_do(x, n, &frames, NULL);
    if (features_version == 1) bo_chroma_get_values_v1(chroma, frames, out + 10);
    else bo_chroma_get_values(chroma, frames, out + 10);
    free(chroma);
    secs[4] = now_seconds() - t0;
    return BO_OK;
}

typedef struct {
    const float *pcm; const uint64_t *offsets, *lengths; uint32_t n_songs, version, tid, nthreads;
    float *out; int32_t *status;
} batch_job;

static void *batch_worker(void *arg) {
    batch_job *j = (batch_job *)arg;
    const uint32_t d = j->version == 1 ? 20 : 23;
    for (uint32_t s = j->tid; s < j->n_songs; s += j->nthreads) {
        float *o = j->out + (size_t)s * d;
        for (uint32_t k = 0; k < d; k++) o[k] = NAN;
        j->status[s] = bo_song_analyze(j->pcm + j->offsets[s], (size_t)j->lengths[s], j->version, o);
    }
    return NULL;
}

void bo_song_analyze_batch(const float *pcm, const uint64_t *offsets, const uint64_t *lengths, uint32_t n_songs,
                           uint32_t features_version, float *out, int32_t *status, uint32_t n_threads) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > n_songs) n_threads = n_songs ? n_songs : 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * n_threads);
    batch_job *jobs = (batch_job *)malloc(sizeof(batch_job) * n_threads);
    for (uint32_t t = 0; t < n_threads; t++) {
        jobs[t] = (batch_job){pcm, offsets, lengths, n_songs, features_version, t, n_threads, out, status};
        pthread_create(&th[t], NULL, batch_worker, &jobs[t]);
    }
    for (uint32_t t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    free(th); free(jobs);
}

/* ------------------------------------------------------------------------------------------
 * src/playlist.rs distances.  ndarray's 1-D f32 `dot` on contiguous data is `unrolled_dot`
 * (8 partial sums, pairwise combine, sequential tail); Array1.dot(Array2) evaluates
 * column-by-column with a plain sequential sum (columns are strided).
 * ---------------------------------------------------------------------------------------- */
static float unrolled_dot(const float *xs, const float *ys, size_t len) {
    float sum = 0.0f, p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t i = 0;
    for (; i + 8 <= len; i += 8)
        for (int k = 0; k < 8; k++) p[k] = p[k] + xs[i + k] * ys[i + k];
    sum = sum + (p[0] + p[4]);
    sum = sum + (p[1] + p[5]);
    sum = sum + (p[2] + p[6]);
    sum = sum + (p[3] + p[7]);
    for (; i < len; i++) sum = sum + xs[i] * ys[i];
    return sum;
}

/* src/playlist.rs:140-142  (a-b).dot(m).dot(&(a-b)).sqrt() */
float bo_mahalanobis_distance(const float *a, const float *b, const float *m, size_t d) {
    float v[64], t[64];
    if (d > 64) return NAN;
    for (size_t i = 0; i < d; i++) v[i] = a[i] - b[i];
    for (size_t j = 0; j < d; j++) {
        float s = 0.0f;
        for (size_t i = 0; i < d; i++) s = s + v[i] * m[i * d + j];
        t[j] = s;
    }
    return sqrtf(unrolled_dot(t, v, d));
}

/* src/playlist.rs:65-71 -- same expression with m = eye(d) */
float bo_euclidean_distance(const float *a, const float *b, size_t d) {
    float m[64 * 64];
    if (d > 64) return NAN;
    memset(m, 0, sizeof(float) * d * d);
    for (size_t i = 0; i < d; i++) m[i * d + i] = 1.0f;
    return bo_mahalanobis_distance(a, b, m, d);
}

/* src/playlist.rs:76-79 */
float bo_cosine_distance(const float *a, const float *b, size_t d) {
    const float similarity = unrolled_dot(a, b, d) / (sqrtf(unrolled_dot(a, a, d)) * sqrtf(unrolled_dot(b, b, d)));
    return 1.0f - similarity;
}

/* src/lib.rs:168-173, 209-234 */
void bo_feature_weights(uint32_t features_version, float *m) {
    const size_t d = features_version == 1 ? 20 : 23;
    memset(m, 0, sizeof(float) * d * d);
    for (size_t i = 0; i < d; i++) {
        float w = 1.0f;
        if (features_version != 1) { if (i == 0) w = 0.25f; else if (i >= 10) w = 3.0f / 13.0f; }
        m[i * d + i] = w;
    }
}

typedef struct { const float *A, *B, *M; size_t n, m, d; int metric; float *out; uint32_t tid, nthreads; } pw_job;
static void *pw_worker(void *arg) {
    pw_job *j = (pw_job *)arg;
    for (size_t r = j->tid; r < j->n; r += j->nthreads)
        for (size_t c = 0; c < j->m; c++) {
            const float *a = j->A + r * j->d, *b = j->B + c * j->d;
            float v;
            if (j->metric == 0) v = bo_euclidean_distance(a, b, j->d);
            else if (j->metric == 1) v = bo_cosine_distance(a, b, j->d);
            else v = bo_mahalanobis_distance(a, b, j->M, j->d);
            j->out[r * j->m + c] = v;
        }
    return NULL;
}

void bo_pairwise(const float *A, size_t n, const float *B, size_t m, size_t d, int metric, const float *M, float *out,
                 uint32_t n_threads) {
    if (n_threads < 1) n_threads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * n_threads);
    pw_job *jobs = (pw_job *)malloc(sizeof(pw_job) * n_threads);
    for (uint32_t t = 0; t < n_threads; t++) {
        jobs[t] = (pw_job){A, B, M, n, m, d, metric, out, t, n_threads};
        pthread_create(&th[t], NULL, pw_worker, &jobs[t]);
    }
    for (uint32_t t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    free(th); free(jobs);
}

/* ------------------------------------------------------------------------------------------
 * Playlist ordering: src/playlist.rs:24-59 (FunctionDistanceMetric), :173-221
 * (variance_based_weight_matrix), :256-270 (closest_to_songs), :272-326 (song_to_song),
 * :367-402 (dedup_playlist_custom_distance).  Songs are rows of feature matrices; the results are
 * index permutations / kept-index lists.
 * ---------------------------------------------------------------------------------------- */
static float pw_single(const float *a, const float *b, const float *M, size_t d, int metric) {
    if (metric == 0) return bo_euclidean_distance(a, b, d);
    if (metric == 1) return bo_cosine_distance(a, b, d);
    return bo_mahalanobis_distance(a, b, M, d);
}

/* FunctionDistanceMetric::distance (:52-58): self.state.iter().map(|v| func(v, vector)).sum::<f32>() */
float bo_set_distance(const float *seeds, size_t n_seeds, const float *v, size_t d, int metric, const float *M) {
    float acc = 0.0f;
    for (size_t i = 0; i < n_seeds; i++) acc += pw_single(seeds + i * d, v, M, d, metric);
    return acc;
}

/* closest_to_songs (:256-270): sort_by_cached_key(n32(distance to the seed set)) -- a STABLE sort;
 * n32 panics on NaN => return -1.  order[k] = index of the k-th song of the playlist. */
int bo_closest_to_songs(const float *seeds, size_t n_seeds, const float *cand, size_t n, size_t d, int metric,
                        const float *M, uint32_t *order, float *dist_out) {
    float *key = (float *)malloc(sizeof(float) * (n ? n : 1));
    uint32_t *tmp = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    int rc = 0;
    for (size_t j = 0; j < n; j++) {
        key[j] = bo_set_distance(seeds, n_seeds, cand + j * d, d, metric, M);
        if (key[j] != key[j]) rc = -1;
        order[j] = (uint32_t)j;
    }
    if (rc == 0) {  /* bottom-up merge sort: stable, like slice::sort_by_cached_key */
        for (size_t w = 1; w < n; w *= 2) {
            for (size_t lo = 0; lo < n; lo += 2 * w) {
                size_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
                size_t i = lo, j = mid, k = lo;
                while (i < mid && j < hi) tmp[k++] = (key[order[j]] < key[order[i]]) ? order[j++] : order[i++];
                while (i < mid) tmp[k++] = order[i++];
                while (j < hi) tmp[k++] = order[j++];
            }
            memcpy(order, tmp, sizeof(uint32_t) * n);
        }
    }
    if (dist_out) memcpy(dist_out, key, sizeof(float) * n);
    free(key); free(tmp);
    return rc;
}

/* song_to_song (:272-326): greedy nearest-neighbour chain.  The first metric is built from all the
 * initial songs, every later one from the single song just emitted; `distances.argmin()`
 * (ndarray-stats: first minimum, Err on NaN => -1) over the pool in its current order, and
 * Vec::remove keeps the relative order of the rest. */
int bo_song_to_song(const float *seeds, size_t n_seeds, const float *cand, size_t n, size_t d, int metric,
                    const float *M, uint32_t *order) {
    uint32_t *pool = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    float *cur = (float *)malloc(sizeof(float) * (n_seeds > 1 ? n_seeds : 1) * d);
    size_t n_cur = n_seeds, n_pool = n;
    memcpy(cur, seeds, sizeof(float) * n_seeds * d);
    for (size_t j = 0; j < n; j++) pool[j] = (uint32_t)j;
    int rc = 0;
    for (size_t step = 0; step < n && rc == 0; step++) {
        size_t best = 0;
        float best_d = 0.0f;
        for (size_t j = 0; j < n_pool; j++) {
            const float dj = bo_set_distance(cur, n_cur, cand + (size_t)pool[j] * d, d, metric, M);
            if (dj != dj) { rc = -1; break; }
            if (j == 0 || dj < best_d) { best = j; best_d = dj; }
        }
        if (rc) break;
        const uint32_t idx = pool[best];
        memmove(pool + best, pool + best + 1, sizeof(uint32_t) * (n_pool - best - 1));
        n_pool--;
        order[step] = idx;
        memcpy(cur, cand + (size_t)idx * d, sizeof(float) * d);
        n_cur = 1;
    }
    free(pool); free(cur);
    return rc;
}

/* dedup_playlist_custom_distance (:367-402): s1 absorbs the following songs while
 * n32(distance(s1, s2)) < threshold or (same_meta(s1, s2)); same_meta[i*n + j] (may be NULL) says
 * whether songs i and j have the same non-empty title and artist.  Returns the number kept. */
long bo_dedup_playlist(const float *songs, size_t n, size_t d, int metric, const float *M, float threshold,
                       const uint8_t *same_meta, uint32_t *kept) {
    size_t n_kept = 0, i = 0;
    while (i < n) {
        size_t j = i + 1;
        while (j < n) {
            const float dist = bo_set_distance(songs + i * d, 1, songs + j * d, d, metric, M);
            if (dist != dist) return -1;
            const int same = (dist < threshold) || (same_meta && same_meta[i * n + j]);
            if (!same) break;
            j++;
        }
        kept[n_kept++] = (uint32_t)i;
        i = j;
    }
    return (long)n_kept;
}

/* variance_based_weight_matrix (:173-221): 0 ok, 1 "seeds must contain more than one element",
 * 2 "seed feature vectors must not be empty" */
int bo_variance_weight_matrix(const float *seeds, size_t n_seeds, size_t d, float *m) {
    if (n_seeds < 2) return 1;
    if (d == 0) return 2;
    const float ns = (float)n_seeds;
    float *mean = (float *)calloc(d, sizeof(float)), *var = (float *)calloc(d, sizeof(float));
    for (size_t i = 0; i < n_seeds; i++)
        for (size_t k = 0; k < d; k++) mean[k] += seeds[i * d + k];
    for (size_t k = 0; k < d; k++) mean[k] /= ns;
    for (size_t i = 0; i < n_seeds; i++)
        for (size_t k = 0; k < d; k++) {
            const float diff = seeds[i * d + k] - mean[k];
            var[k] = var[k] + diff * diff;
        }
    float sum = 0.0f;
    for (size_t k = 0; k < d; k++) {
        var[k] /= ns;
        var[k] = 1.0f / (var[k] + 1e-6f);  /* weights */
    }
    /* ndarray sum(): unrolled 8-lane pairwise for contiguous data -- same reduction as unrolled_dot's */
    {
        float p[8] = {0};
        size_t k = 0;
        for (; k + 8 <= d; k += 8)
            for (int u = 0; u < 8; u++) p[u] += var[k + u];
        sum += (p[0] + p[4]);
        sum += (p[1] + p[5]);
        sum += (p[2] + p[6]);
        sum += (p[3] + p[7]);
        for (; k < d; k++) sum += var[k];
    }
    const float scale = (float)d / sum;
    memset(m, 0, sizeof(float) * d * d);
    for (size_t k = 0; k < d; k++) m[k * d + k] = var[k] * scale;
    free(mean); free(var);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Synthetic white noise (bench/test input, not from the reference): Philox4x32-10.
 * ---------------------------------------------------------------------------------------- */
static inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1, uint32_t r[4]) {
    uint32_t c[4] = {c0, c1, 0u, 0u};
    for (int round = 0; round < 10; round++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    r[0] = c[0]; r[1] = c[1]; r[2] = c[2]; r[3] = c[3];
}

void bo_white_noise(uint32_t song_index, size_t n, float *out) {
    const uint32_t k0 = 0x5EED0000u + song_index;
    for (size_t blk = 0; blk * 4 < n; blk++) {
        uint32_t r[4];
        philox4x32_10((uint32_t)blk, (uint32_t)((uint64_t)blk >> 32), k0, 0u, r);
        for (size_t e = 0; e < 4 && blk * 4 + e < n; e++)
            out[blk * 4 + e] = (float)(r[e] >> 8) * (1.0f / 16777216.0f) - 0.5f;
    }
}

/* ------------------------------------------------------------------------------------------
 * The decoder's sample-rate / channel conversion.
 *
 * FFmpegDecoder hands every decoded frame to libswresample (src/song/decoder/ffmpeg.rs:36-109:
 * `software::resampling::context::Context::get(in_format, in_layout, in_rate, F32 packed, MONO,
 * 22050)` + `run` per frame + `flush`), with every option at its default.  libswresample is part
 * of FFmpeg, a C library the crate links through ffmpeg-next / ffmpeg-sys-next (Cargo.lock); it is
 * NOT under /root/reference, so its published algorithm (libswresample/resample.c,
 * resample_template.c, rematrix.c, swresample.c of FFmpeg 4.x - 7.x) is restated here:
 *
 *   options      filter_size 32, phase_shift 10, cutoff 0.97, Kaiser window beta 9, exact_rational on,
 *                linear_interp off, internal sample format FLTP (float), no dither for float output
 *   filter bank  build_filter(): factor = min(out * cutoff / in, 1); taps = ceil(32 / factor) rounded up to
 *                even; phase_count = out / gcd(in, out) when that is <= 1024 (exact_rational), else 1024;
 *                tap i of phase ph: x = pi ((i - center) - ph / phase_count) factor, y = sin(x) / x (a
 *                sin LUT with alternating sign when factor == 1), times I0(9 sqrt(1 - w^2)),
 *                w = 2 x / (factor taps pi); divided by the sum of phase 0; rounded to f32; for an even
 *                phase_count the phases above the middle are mirror copies
 *   stepping     output k sits at position floor(k dst_incr / src_incr) in units of 1 / phase_count input
 *                samples, dst_incr / src_incr = in phase_count / out; phase = position mod phase_count (the
 *                lower phase: no interpolation), first tap at sample position / phase_count - center
 *   edges        the stream starts with `taps` samples mirrored about sample 0 (invert_initial_buffer) and
 *                the first output is centred on sample 0; flush mirrors (min(left, taps) + 1) / 2 samples
 *                behind the end (edge sample repeated) and outputs while a full window is left
 *   arithmetic   s16 -> float: s * 2^-15, s32 -> float: (float)s * 2^-31; the dot product as
 *                ff_resample_common_float_fma3 sums it (eight lanes of fused multiply-adds over taps
 *                i = j mod 8, then (j, j + 4), (0 + 2, 1 + 3), (0 + 1)): the form FFmpeg runs on every
 *                x86-64 with AVX2 + FMA3; channels > mono: resample_first (1 * 1 / channels - 1 < 0 in
 *                integer arithmetic), i.e. every channel is resampled, THEN mixed: stereo -> mono =
 *                l * sqrt(1/2) + r * sqrt(1/2) in float (no normalisation for float output)
 *
 * PINNED by the reference's own decoder tests: the Adler-32 of the f32le stream must be 0xa0f8b8af for
 * data/s32_mono_44_1_kHz.flac, 0xbbcba1cf for data/s32_stereo_44_1_kHz.flac (ffmpeg.rs:433-445) and
 * 0xd594429c for data/no_channel.wav (:471-476) -- tests/test_oracle_golden.py holds this code to all three
 * (and to 0x1d7b2d6d for the 22 050 Hz stereo file, :448-452).  Those pins cover 44 100 -> 22 050 Hz (one phase);
 * other rates use the same code path but no reference test holds a number for them.
 * More than two channels have no FFmpeg pin (the matrix depends on the layout) and follow the reference's other decoder:
 * the sequential channel mean first (src/song/decoder/symphonia.rs:291-297), then the resampler.
 * ---------------------------------------------------------------------------------------- */
#define BO_SWR_FILTER_SIZE 32
#define BO_SWR_PHASE_SHIFT 10
#define BO_SWR_CUTOFF 0.97
#define BO_SWR_KAISER_BETA 9.0

static uint64_t gcd_u64(uint64_t a, uint64_t b) { while (b) { const uint64_t t = a % b; a = b; b = t; } return a; }

/* modified Bessel function I0 (power series; every term is positive, so the sum is good to a few ulp for x <= 9) */
static double swr_bessel_i0(double x) {
    const double q = 0.25 * x * x;
    double term = 1.0, sum = 1.0;
    for (int k = 1; k < 200; k++) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < sum * 1e-18) break;
    }
    return sum;
}

int bo_swr_plan(uint32_t in_rate, bo_swr_plan_t *p) {
    if (in_rate == 0 || !p) return 1;
    const uint32_t out_rate = BO_SAMPLE_RATE;
    const double factor = fmin((double)out_rate * BO_SWR_CUTOFF / (double)in_rate, 1.0);
    int taps = (int)ceil(BO_SWR_FILTER_SIZE / factor);
    if (taps < 1) taps = 1;
    if (taps > 1) taps = (taps + 1) & ~1;
    int phase_count = 1 << BO_SWR_PHASE_SHIFT;
    const uint64_t g = gcd_u64(in_rate, out_rate);
    if (out_rate / g <= (uint64_t)phase_count) phase_count = (int)(out_rate / g); /* exact_rational */
    /* dst_incr / src_incr = in_rate * phase_count / out_rate, reduced */
    uint64_t num = (uint64_t)in_rate * (uint64_t)phase_count, den = out_rate;
    const uint64_t g2 = gcd_u64(num, den);
    p->in_rate = in_rate;
    p->taps = taps;
    p->phase_count = phase_count;
    p->center = (taps - 1) / 2;
    p->dst_incr = num / g2;
    p->src_incr = den / g2;
    p->factor = factor;
    return 0;
}

/* build_filter() for AV_SAMPLE_FMT_FLTP: bank[phase_count][taps] */
void bo_swr_filter(const bo_swr_plan_t *p, float *bank) {
    const int taps = p->taps, pc = p->phase_count, center = p->center;
    const int ph_nb = (pc % 2) ? pc : pc / 2 + 1;
    const double factor = p->factor;
    double *tab = (double *)malloc(sizeof(double) * (size_t)(taps + 1));
    double *sin_lut = (double *)malloc(sizeof(double) * (size_t)ph_nb);
    double norm = 0.0;
    if (factor == 1.0)
        for (int ph = 0; ph < ph_nb; ph++) sin_lut[ph] = sin(M_PI * ph / pc) * ((center & 1) ? 1 : -1);
    for (int ph = 0; ph < ph_nb; ph++) {
        double s = factor == 1.0 ? sin_lut[ph] : 0.0;
        for (int i = 0; i < taps; i++) {
            const double x = M_PI * ((double)(i - center) - (double)ph / pc) * factor;
            double y;
            if (x == 0) y = 1.0;
            else if (factor == 1.0) y = s / x;
            else y = sin(x) / x;
            const double w = 2.0 * x / (factor * taps * M_PI);
            const double r = 1.0 - w * w;
            y *= swr_bessel_i0(BO_SWR_KAISER_BETA * sqrt(r > 0.0 ? r : 0.0));
            tab[i] = y;
            s = -s;
            if (!ph) norm += y;
        }
        for (int i = 0; i < taps; i++) bank[(size_t)ph * taps + i] = (float)(tab[i] * 1.0 / norm);
        /* even phase_count: phase pc - ph is the tap-reversed copy of phase ph; the middle phase is copied onto itself in
           place, tap by tap (its second half becomes the mirror of its first); phase pc itself -- only read by the linear
           interpolation, which is off -- is not kept */
        if (pc % 2 == 0 && ph > 0)
            for (int i = 0; i < taps; i++) bank[(size_t)(pc - ph) * taps + (taps - 1 - i)] = bank[(size_t)ph * taps + i];
    }
    free(tab);
    free(sin_lut);
}

/* first tap (may be negative) and phase of output k */
static inline void swr_position(const bo_swr_plan_t *p, uint64_t k, int64_t *first, int *phase) {
    const unsigned __int128 pos = (unsigned __int128)k * p->dst_incr / p->src_incr;
    *first = (int64_t)(uint64_t)(pos / (uint64_t)p->phase_count) - p->center;
    *phase = (int)(uint64_t)(pos % (uint64_t)p->phase_count);
}

/* number of outputs k >= 0 whose window [first, first + taps) ends at or before sample index `limit` (exclusive) */
static uint64_t swr_count_until(const bo_swr_plan_t *p, int64_t limit) {
    /* first_k + taps <= limit  <=>  floor(pos_k / pc) <= limit - taps + center =: m  <=>  pos_k < (m + 1) pc
       <=>  k dst / src < (m + 1) pc  <=>  k < (m + 1) pc src / dst */
    const int64_t m = limit - p->taps + p->center;
    if (m < 0) return 0;
    const unsigned __int128 a = (unsigned __int128)(uint64_t)(m + 1) * (uint64_t)p->phase_count * p->src_incr;
    return (uint64_t)((a + p->dst_incr - 1) / p->dst_incr);
}

uint64_t bo_swr_out_len(uint64_t n_in, uint32_t in_rate) {
    if (in_rate == BO_SAMPLE_RATE) return n_in;
    bo_swr_plan_t p;
    if (bo_swr_plan(in_rate, &p)) return 0;
    if (n_in < (uint64_t)p.taps + 1) return 0; /* invert_initial_buffer never gets its taps + 1 samples */
    const uint64_t k_fail = swr_count_until(&p, (int64_t)n_in); /* first output that lacks input */
    int64_t first; int ph;
    swr_position(&p, k_fail, &first, &ph);
    int64_t left = (int64_t)n_in - first;
    if (left < 0) left = 0;
    if (left > p.taps) left = p.taps;
    const int64_t reflection = (left + 1) / 2;
    return swr_count_until(&p, (int64_t)n_in + reflection);
}

static inline float swr_sample(const float *x, uint64_t n, int64_t i) {
    if (i < 0) i = -i;                                   /* invert_initial_buffer: x[-j] = x[j] */
    else if ((uint64_t)i >= n) i = 2 * (int64_t)n - 1 - i; /* resample_flush: x[n + j] = x[n - 1 - j] */
    return x[i];
}

/* one channel; out has bo_swr_out_len(n, in_rate) samples */
void bo_swr_resample(const float *x, uint64_t n, uint32_t in_rate, float *out) {
    if (in_rate == BO_SAMPLE_RATE) { memcpy(out, x, sizeof(float) * n); return; }
    bo_swr_plan_t p;
    bo_swr_plan(in_rate, &p);
    const uint64_t n_out = bo_swr_out_len(n, in_rate);
    float *bank = (float *)malloc(sizeof(float) * (size_t)p.phase_count * (size_t)p.taps);
    bo_swr_filter(&p, bank);
    for (uint64_t k = 0; k < n_out; k++) {
        int64_t first; int ph;
        swr_position(&p, k, &first, &ph);
        const float *f = bank + (size_t)ph * p.taps;
        float acc[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        for (int i = 0; i < p.taps; i++) acc[i & 7] = fmaf(swr_sample(x, n, first + i), f[i], acc[i & 7]);
        const float b0 = acc[0] + acc[4], b1 = acc[1] + acc[5], b2 = acc[2] + acc[6], b3 = acc[3] + acc[7];
        const float c0 = b0 + b2, c1 = b1 + b3;
        out[k] = c0 + c1;
    }
    free(bank);
}

static inline float swr_to_float(const void *pcm, int sample_format, uint64_t i) {
    switch (sample_format) {
    case 1: return (float)((const int16_t *)pcm)[i] * (1.0f / (1 << 15));   /* conv S16 -> FLT */
    case 2: return (float)((const int32_t *)pcm)[i] * (1.0f / (1U << 31));  /* conv S32 -> FLT */
    default: return ((const float *)pcm)[i];
    }
}

/* FFmpegDecoder's conversion of decoded frames (sample_format 0 f32 / 1 s16 / 2 s32, interleaved channels) to the mono
 * 22 050 Hz f32 stream Song::analyze takes.  out has bo_swr_out_len(frames, in_rate) samples; returns that count. */
uint64_t bo_decode_to_mono(const void *pcm, int sample_format, uint32_t channels, uint64_t frames, uint32_t in_rate, float *out) {
    const uint64_t n_out = bo_swr_out_len(frames, in_rate);
    if (channels == 0) return 0;
    float *ch = (float *)malloc(sizeof(float) * (size_t)(frames ? frames : 1));
    if (channels == 2) {
        /* resample_first: each channel through the resampler, then the matrix (both coefficients (float)M_SQRT1_2) */
        float *r0 = (float *)malloc(sizeof(float) * (size_t)(n_out ? n_out : 1));
        float *r1 = (float *)malloc(sizeof(float) * (size_t)(n_out ? n_out : 1));
        for (uint64_t i = 0; i < frames; i++) ch[i] = swr_to_float(pcm, sample_format, 2 * i);
        if (in_rate == BO_SAMPLE_RATE) memcpy(r0, ch, sizeof(float) * frames); else bo_swr_resample(ch, frames, in_rate, r0);
        for (uint64_t i = 0; i < frames; i++) ch[i] = swr_to_float(pcm, sample_format, 2 * i + 1);
        if (in_rate == BO_SAMPLE_RATE) memcpy(r1, ch, sizeof(float) * frames); else bo_swr_resample(ch, frames, in_rate, r1);
        const float c = (float)M_SQRT1_2;
        for (uint64_t k = 0; k < n_out; k++) out[k] = r0[k] * c + r1[k] * c;
        free(r0); free(r1);
    } else {
        for (uint64_t i = 0; i < frames; i++) {
            if (channels == 1) ch[i] = swr_to_float(pcm, sample_format, i);
            else {
                float acc = 0.0f;
                for (uint32_t c = 0; c < channels; c++) acc = acc + swr_to_float(pcm, sample_format, (uint64_t)channels * i + c);
                ch[i] = acc / (float)channels;
            }
        }
        if (in_rate == BO_SAMPLE_RATE) memcpy(out, ch, sizeof(float) * frames); else bo_swr_resample(ch, frames, in_rate, out);
    }
    free(ch);
    return n_out;
}
