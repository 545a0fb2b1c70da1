// resample.hpp -- the plan and the filter bank of the device resampler (device-free host code).
//
// The reference's FFmpegDecoder sends every decoded frame through libswresample with its default options
// (src/song/decoder/ffmpeg.rs:36-109: Context::get(in_format, in_layout, in_rate, F32 packed, MONO, 22050), run per
// frame, flush).  The device kernel in kernels_pcm.hip reproduces that conversion -- bit for bit at 44 100 Hz, the one-phase
// 2 : 1 case every reference file needs and its Adler-32 tests pin; the many-phase paths (48 kHz: 147 phases, 441, the
// 1024-phase inexact case, up-sampling with its flush reflection) are the same restatement of the published algorithm, held to
// the oracle's independent one and to a sinusoid-reconstruction property but to no external number --; this file computes what
// it needs on the host, in double like FFmpeg's build_filter():
//   * taps  = ceil(32 / factor) rounded up to even, factor = min(22050 * 0.97 / in_rate, 1)  (filter_size 32, cutoff 0.97)
//   * phase_count = 22050 / gcd(in_rate, 22050) when <= 1024 (exact_rational), else 1024 (phase_shift 10, nearest lower
//     phase, no interpolation)
//   * tap i of phase p: sinc(x) * I0(9 sqrt(1 - w^2)), x = pi ((i - center) - p / phase_count) factor,
//     w = 2 x / (factor taps pi); divided by the tap sum of phase 0, rounded to f32 (Kaiser window, beta 9)
//   * output k sits at floor(k dst_incr / src_incr) / phase_count input samples, dst_incr / src_incr =
//     in_rate * phase_count / 22050
//   * the stream is extended by `taps` samples mirrored about sample 0 and by (min(left, taps) + 1) / 2 samples
//     mirrored behind the end; the output count follows from that (= ceil(frames * 22050 / in_rate) for every rate tried)
// Pinned through the kernel by the reference's Adler-32 decoder tests (ffmpeg.rs:433-452, 471-476): 44 100 Hz only.
// (swr_i0 below is a power series where FFmpeg's bessel() is a rational approximation: equal to the last f32 bit of every bank
// entry at 44 100 Hz -- the hashes say so -- and unverified elsewhere for the same reason.)
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace bg {

constexpr uint32_t SWR_OUT_RATE = 22050;  // SAMPLE_RATE, src/lib.rs:140

struct SwrPlan {
    uint32_t in_rate = 0;
    int taps = 0, phase_count = 0, center = 0;
    uint64_t dst_incr = 0, src_incr = 0;
    double factor = 0.0;
};

inline uint64_t swr_gcd(uint64_t a, uint64_t b) {
    while (b) { const uint64_t t = a % b; a = b; b = t; }
    return a;
}

inline bool swr_make_plan(uint32_t in_rate, SwrPlan* p) {
    if (in_rate == 0) return false;
    p->in_rate = in_rate;
    p->factor = std::fmin((double)SWR_OUT_RATE * 0.97 / (double)in_rate, 1.0);
    int taps = (int)std::ceil(32.0 / p->factor);
    if (taps > 1) taps = (taps + 1) & ~1;
    p->taps = taps;
    p->center = (taps - 1) / 2;
    const uint64_t reduced = SWR_OUT_RATE / swr_gcd(in_rate, SWR_OUT_RATE);
    p->phase_count = reduced <= 1024 ? (int)reduced : 1024;
    const uint64_t num = (uint64_t)in_rate * (uint64_t)p->phase_count, den = SWR_OUT_RATE, g = swr_gcd(num, den);
    p->dst_incr = num / g;
    p->src_incr = den / g;
    return true;
}

// I0 by its power series (all terms positive: a few ulp for arguments up to the window's beta = 9)
inline double swr_i0(double x) {
    const double q = 0.25 * x * x;
    double term = 1.0, sum = 1.0;
    for (int k = 1; k < 200; k++) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < sum * 1e-18) break;
    }
    return sum;
}

// bank[phase][tap], phase_count x taps floats
inline void swr_make_filter(const SwrPlan& p, std::vector<float>& bank) {
    const int taps = p.taps, pc = p.phase_count, center = p.center;
    const int directly = (pc % 2) ? pc : pc / 2 + 1;  // an even bank mirrors its upper phases from the lower ones
    const double pi = 3.14159265358979323846, factor = p.factor;
    bank.assign((size_t)pc * taps, 0.0f);
    std::vector<double> tab(taps);
    double norm = 0.0;
    for (int ph = 0; ph < directly; ph++) {
        // without a low-pass (up-sampling) sin(x) only changes sign from tap to tap: one sine per phase
        double s = factor == 1.0 ? std::sin(pi * ph / pc) * ((center & 1) ? 1 : -1) : 0.0;
        for (int i = 0; i < taps; i++) {
            const double x = pi * ((double)(i - center) - (double)ph / pc) * factor;
            double y = x == 0 ? 1.0 : (factor == 1.0 ? s / x : std::sin(x) / x);
            const double w = 2.0 * x / (factor * taps * pi), r = 1.0 - w * w;
            y *= swr_i0(9.0 * std::sqrt(r > 0.0 ? r : 0.0));
            tab[i] = y;
            s = -s;
            if (ph == 0) norm += y;
        }
        float* row = bank.data() + (size_t)ph * taps;
        for (int i = 0; i < taps; i++) row[i] = (float)(tab[i] * 1.0 / norm);
        if (pc % 2 == 0 && ph > 0) {  // tap-reversed copy (the middle phase onto itself, tap by tap)
            float* mirror = bank.data() + (size_t)(pc - ph) * taps;
            for (int i = 0; i < taps; i++) mirror[taps - 1 - i] = row[i];
        }
    }
}

// outputs k >= 0 whose window [first_k, first_k + taps) ends at or before sample `limit`
inline uint64_t swr_count_until(const SwrPlan& p, int64_t limit) {
    const int64_t m = limit - p.taps + p.center;  // first_k + taps <= limit  <=>  floor(pos_k / phase_count) <= m
    if (m < 0) return 0;
    const unsigned __int128 a = (unsigned __int128)(uint64_t)(m + 1) * (uint64_t)p.phase_count * p.src_incr;
    return (uint64_t)((a + p.dst_incr - 1) / p.dst_incr);
}

inline int64_t swr_first_tap(const SwrPlan& p, uint64_t k) {
    const unsigned __int128 pos = (unsigned __int128)k * p.dst_incr / p.src_incr;
    return (int64_t)(uint64_t)(pos / (uint64_t)p.phase_count) - p.center;
}

inline uint64_t swr_out_len(const SwrPlan& p, uint64_t n_in) {
    if (n_in < (uint64_t)p.taps + 1) return 0;  // the resampler never sees the taps + 1 samples its start needs
    const uint64_t k_fail = swr_count_until(p, (int64_t)n_in);  // the first output the input alone cannot serve
    int64_t left = (int64_t)n_in - swr_first_tap(p, k_fail);
    left = left < 0 ? 0 : (left > p.taps ? p.taps : left);
    return swr_count_until(p, (int64_t)n_in + (left + 1) / 2);
}

}  // namespace bg
