// kernels_finalize.hip -- per-song summaries and assembly of the feature row
// (compiled with -ffp-contract=off: the timbral summaries below follow the reference's scalar f32
// evaluation order bit for bit, so the only GPU/CPU difference left there is FFT rounding; the
// loudness chunks are assembled from 256-sample lane-tree partial sums, i.e. they match the
// reference's sequential 1024-sample sums to rounding, not bit for bit).
//
//   mean  : utils::mean (src/utils.rs:66-68), sequential f32 sum / len
//   std   : ndarray std_axis(ddof = 0): Welford with one fused mul_add (src/timbral.rs:61-63 ...)
//   zcr   : ZeroCrossingRateDesc::get_value (src/timbral.rs:250-252)
//   loud  : LoudnessDesc::get_value (src/misc.rs:51-65)
//   chroma: ChromaDesc::get_values / get_values_version_1 (src/chroma.rs:97-132)
//   order : [tempo, zcr, centroid x2, rolloff x2, flatness x2, loudness x2, chroma x10|13]
//           (src/song/mod.rs:493-498)
#include "device_utils.hpp"
#include "internal.hpp"

namespace bg {

__device__ __forceinline__ float normalize(float v, float mn, float mx) { return 2.0f * (v - mn) / (mx - mn) - 1.0f; }

// The loops below are sequential by definition (they reproduce the reference's rounding order).  A wavefront
// executing ONE such chain still pays a full 4-cycle VALU issue per instruction, so the mapping is one LANE per
// song: 64 songs advance their (independent) chains in lock step and a role costs 1/64 of the issue slots.
// Loads are issued 16 elements at a time (16-byte loads once the lane's cursor is aligned) so the dependent
// arithmetic chain never waits on memory.
constexpr int SEQ_CHUNK = 16;

template <typename Step>
__device__ __forceinline__ void seq_for_each(const float* __restrict__ x, uint32_t n, Step&& step) {
    uint32_t i = 0;
    while (i < n && ((reinterpret_cast<uintptr_t>(x + i) & 15) != 0)) { step(x[i], i); i++; }
    for (; i + SEQ_CHUNK <= n; i += SEQ_CHUNK) {
        float4 v[SEQ_CHUNK / 4];
#pragma unroll
        for (int u = 0; u < SEQ_CHUNK / 4; u++) v[u] = *reinterpret_cast<const float4*>(x + i + 4 * u);
#pragma unroll
        for (int u = 0; u < SEQ_CHUNK / 4; u++) {
            step(v[u].x, i + 4 * u);
            step(v[u].y, i + 4 * u + 1);
            step(v[u].z, i + 4 * u + 2);
            step(v[u].w, i + 4 * u + 3);
        }
    }
    for (; i < n; i++) step(x[i], i);
}

__device__ float seq_mean(const float* __restrict__ x, uint32_t n) {
    float s = 0.0f;
    seq_for_each(x, n, [&](float v, uint32_t) { s += v; });
    return s / (float)n;
}

__device__ __forceinline__ void welford_step(float v, uint32_t i, float& mean, float& sum_sq) {
    const float count = (float)(i + 1);
    const float delta = v - mean;
    mean = mean + delta / count;
    sum_sq = __fmaf_rn(v - mean, delta, sum_sq);
}

__device__ float seq_std(const float* __restrict__ x, uint32_t n) {
    float mean = 0.0f, sum_sq = 0.0f;
    seq_for_each(x, n, [&](float v, uint32_t i) { welford_step(v, i, mean, sum_sq); });
    return sqrtf(sum_sq / ((float)n - 0.0f));
}

enum Role { R_CENT_MEAN = 0, R_CENT_STD, R_ROLL_MEAN, R_ROLL_STD, R_FLAT_MEAN, R_FLAT_STD, R_LOUD, R_ZCR, R_COUNT };

// summary[s][0..15]: slots 1..9 = features 1..9 (zcr, centroid, rolloff, flatness, loudness)
// grid = (ceil(n_songs / 64), R_COUNT): one wavefront = one role of 64 songs (lane = song)
__global__ __launch_bounds__(64) void summary_kernel(const SongDesc* __restrict__ songs, uint32_t n_songs,
                                                     const float* __restrict__ centroid,
                                                     const float* __restrict__ rolloff,
                                                     const float* __restrict__ flatness,
                                                     const float* __restrict__ e256,
                                                     const uint32_t* __restrict__ zc256,
                                                     float* __restrict__ summary) {
    const uint32_t s = blockIdx.x * 64 + threadIdx.x;
    if (s >= n_songs) return;
    const SongDesc sd = songs[s];
    if (!sd.ok) return;
    float* feat = summary + (size_t)s * 16;
    const float half_sr = (float)SAMPLE_RATE / 2.0f;
    switch ((int)blockIdx.y) {  // wave-uniform
        case R_CENT_MEAN: feat[2] = normalize(seq_mean(centroid + sd.t_off, sd.n_t), 0.0f, half_sr); break;
        case R_CENT_STD: feat[3] = normalize(seq_std(centroid + sd.t_off, sd.n_t), 0.0f, half_sr); break;
        case R_ROLL_MEAN: feat[4] = normalize(seq_mean(rolloff + sd.t_off, sd.n_t), 0.0f, half_sr); break;
        case R_ROLL_STD: feat[5] = normalize(seq_std(rolloff + sd.t_off, sd.n_t), 0.0f, half_sr); break;
        case R_FLAT_MEAN: feat[6] = 2.0f * (seq_mean(flatness + sd.t_off, sd.n_t) - 0.0f) / (1.0f - 0.0f) - 1.0f; break;
        case R_FLAT_STD: feat[7] = 2.0f * (seq_std(flatness + sd.t_off, sd.n_t) - 0.0f) / (1.0f - 0.0f) - 1.0f; break;
        case R_LOUD: {
            // chunk level = sum of squares over <=1024 samples / len (src/misc.rs:12-18); the four
            // 256-sample partial sums are added in order
            const float* e = e256 + sd.e_off;
            float msum = 0.0f, mean = 0.0f, sum_sq = 0.0f;
            for (uint32_t c = 0; c < sd.n_l; c++) {
                float en = 0.0f;
                for (uint32_t q = 4 * c; q < 4 * c + 4 && q < sd.n_e; q++) en += e[q];
                const uint64_t len = ((uint64_t)(c + 1) * LOUD_W <= sd.n) ? LOUD_W : sd.n - (uint64_t)c * LOUD_W;
                const float v = en / (float)len;
                msum += v;
                welford_step(v, c, mean, sum_sq);
            }
            float mean_value = msum / (float)sd.n_l;
            float std_value = sqrtf(sum_sq / ((float)sd.n_l - 0.0f));
            if (mean_value < 1e-9f) mean_value = 1e-9f;
            if (std_value < 1e-9f) std_value = 1e-9f;
            feat[8] = normalize(10.0f * log10f(mean_value), -90.0f, 0.0f);
            feat[9] = normalize(10.0f * log10f(std_value), -90.0f, 0.0f);
            break;
        }
        case R_ZCR: {
            const uint32_t* z = zc256 + sd.e_off;
            uint32_t c = 0;
            for (uint32_t q = 0; q < sd.n_e; q++) c += z[q];
            feat[1] = normalize((float)c / (float)sd.n, 0.0f, 1.0f);
            break;
        }
        default: break;
    }
}

// one thread per song: chroma summary (ChromaDesc::get_values*), tempo, and the row itself
__global__ __launch_bounds__(64) void assemble_kernel(const SongDesc* __restrict__ songs, uint32_t n_songs,
                                                      const uint32_t* __restrict__ pfx_ct,
                                                      const float* __restrict__ summary,
                                                      const double* __restrict__ chroma_part,
                                                      const TempoState* __restrict__ tempo,
                                                      const TuningState* __restrict__ tuning,
                                                      uint32_t features_version, float* __restrict__ out,
                                                      int32_t* __restrict__ status,
                                                      int32_t* __restrict__ dbg_tuning,
                                                      uint32_t* __restrict__ dbg_nbpms) {
    const uint32_t s = blockIdx.x * 64 + threadIdx.x;
    if (s >= n_songs) return;
    const SongDesc sd = songs[s];
    const uint32_t d = features_version == 1 ? 20 : 23;
    float* o = out + (size_t)sd.row * d;
    // per-song status, 1:1 with BlissError (BLISSGPU_SONG_OK / BLISSGPU_SONG_TOO_SHORT, src/song/mod.rs:417-430)
    if (status) status[sd.row] = sd.ok ? 0 : 1;
    if (!sd.ok) {
        for (uint32_t k = 0; k < d; k++) o[k] = __int_as_float(0x7fc00000);  // NaN row; status says why
        dbg_tuning[sd.row] = -1;
        dbg_nbpms[sd.row] = 0;
        return;
    }
    float feat[23];
    feat[0] = tempo[s].tempo;
    for (int k = 1; k < 10; k++) feat[k] = summary[(size_t)s * 16 + k];
    // chroma_interval_features' time mean (src/chroma.rs:154), tiles summed in order
    const uint32_t t0 = pfx_ct[s], t1 = pfx_ct[s + 1];
    double raw[10];
    for (int t = 0; t < 10; t++) {
        double acc = 0.0;
        for (uint32_t k = t0; k < t1; k++) acc += chroma_part[(size_t)k * 10 + t];
        raw[t] = acc / (double)sd.n_c;
    }
    if (features_version == 1) {
        for (int t = 0; t < 10; t++) feat[10 + t] = 2.0f * ((float)raw[t] - 0.0f) / (0.12f - 0.0f) - 1.0f;
    } else {
        double n1 = 0.0, n2 = 0.0;
        for (int t = 0; t < 6; t++) n1 += raw[t] * raw[t];
        for (int t = 6; t < 10; t++) n2 += raw[t] * raw[t];
        n1 = sqrt(n1);
        n2 = sqrt(n2);
        if (n1 > 0.0) for (int t = 0; t < 6; t++) raw[t] /= n1;
        if (n2 > 0.0) for (int t = 6; t < 10; t++) raw[t] /= n2;
        for (int t = 0; t < 10; t++) feat[10 + t] = 2.0f * ((float)raw[t] - 0.0f) / (1.0f - 0.0f) - 1.0f;
        feat[20] = fminf(2.0f * ((float)n1 - 0.0f) / (0.25f - 0.0f) - 1.0f, 1.0f);
        feat[21] = fminf(2.0f * ((float)n2 - 0.0f) / (0.025f - 0.0f) - 1.0f, 1.0f);
        const double angle = atan2(20.0 * n2, n1 + 1e-12);
        feat[22] = 2.0f * ((float)angle - 0.0f) / (1.57079632679489661923f - 0.0f) - 1.0f;
    }
    for (uint32_t k = 0; k < d; k++) o[k] = feat[k];
    dbg_tuning[sd.row] = tuning[s].tuning_idx;
    dbg_nbpms[sd.row] = tempo[s].n_bpms;
}

void launch_summary(const Batch& b, const Workspace& w, hipStream_t st) {
    if (b.n_songs == 0) return;
    hipLaunchKernelGGL(summary_kernel, dim3((b.n_songs + 63) / 64, R_COUNT), dim3(64), 0, st, b.songs, b.n_songs, w.centroid,
                       w.rolloff, w.flatness, w.e256, w.zc256, w.summary);
}

void launch_finalize(const Batch& b, const Workspace& w, uint32_t features_version, float* d_out, int32_t* d_status,
                     int32_t* dbg_tuning, uint32_t* dbg_nbpms, hipStream_t st) {
    if (b.n_songs == 0) return;
    hipLaunchKernelGGL(assemble_kernel, dim3((b.n_songs + 63) / 64), dim3(64), 0, st, b.songs, b.n_songs, b.pfx_ct,
                       w.summary, w.chroma_part, w.tempo, w.tuning, features_version, d_out, d_status, dbg_tuning, dbg_nbpms);
}

}  // namespace bg
