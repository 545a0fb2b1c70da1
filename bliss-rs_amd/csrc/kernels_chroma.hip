// kernels_chroma.hip -- the chroma descriptor (src/chroma.rs) on the device.
//
//   chroma_bank_kernel : chroma_filter(22050, 8192, 12, tuning) (src/chroma.rs:197-267) for the 100
//                        tunings pitch_tuning can return (+ tuning 0.0), built once per context.
//   stft8192_kernel    : utils::stft(signal, 8192, 2205) (src/utils.rs:26-64): reflect pad, periodic
//                        Hann (f32), FFT, |X| -- one workgroup per chroma frame, the 8192 real samples
//                        packed as 4096 complex values, six Stockham radix-4 passes in LDS.  The f32
//                        magnitudes (exactly the values the reference widens to f64) are stored for the
//                        contraction; while the frame is still in LDS the kernel also runs pip_track's
//                        peak test (src/chroma.rs:269-331) and counts peaks by coarse magnitude bin.
//   tune_*_kernel      : estimate_tuning (src/chroma.rs:361-391): exact Midpoint median of the peak
//                        magnitudes, then the 100-bin histogram of pitch residues of the peaks at or
//                        above the median and its first argmax (pitch_tuning, :334-359).
//   chroma_kernel      : chroma_stft (:393-412) as an f64 MFMA contraction filter(12x4097) x S^2,
//                        followed per frame by the L1 normalisation, exp(15x), normalisation and the
//                        10 interval templates x 12 rotations (:137-188), summed over the tile.
#include <float.h>

#include "device_utils.hpp"
#include "internal.hpp"

namespace bg {

typedef double double4_t __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// chroma filter bank
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double tuning_of_slot(int slot) {
    // pitch_tuning's return expression (src/chroma.rs:358) with resolution 0.01
    return slot >= N_TUNING ? 0.0 : (-50.0 + (100.0 * 0.01 * (double)slot)) / 100.0;
}

__global__ __launch_bounds__(256) void chroma_bank_kernel(double* __restrict__ bank) {
#pragma clang fp contract(off)
    const int slot = blockIdx.y;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= CBINS_PAD) return;
    double* out = bank + (size_t)slot * BANK_ROWS * CBINS_PAD;
    if (k >= CBINS) {
        for (int r = 0; r < BANK_ROWS; r++) out[(size_t)r * CBINS_PAD + k] = 0.0;
        return;
    }
    const double tuning = tuning_of_slot(slot);
    const double ncf = 12.0, nc2 = 6.0;
    const double step = 22050.0 / 8192.0;  // Array::linspace(0, sr, n_fft + 1)
    const double a440 = 440.0 * pow(2.0, tuning / 12.0);
    auto fbin = [&](int i) { return log2((0.0 + step * (double)i) / (a440 / 16.0)) * ncf; };
    const double fb1 = fbin(1);
    const double fb = (k == 0) ? fb1 - 1.5 * ncf : fbin(k);
    const double fbn = fbin(k + 1);  // k + 1 <= 4097 < n_fft + 1
    double bw = fbn - fb;
    if (bw <= 1.0) bw = 1.0;
    double w[12], l2 = 0.0;
    for (int c = 0; c < 12; c++) {
        double d = -(double)c + fb;
        d = fmod(d + nc2 + 10.0 * ncf, ncf) - nc2;
        d = d / bw;
        w[c] = exp(-0.5 * (2.0 * d) * (2.0 * d));
        l2 += w[c] * w[c];
    }
    l2 = sqrt(l2);
    if (l2 < DBL_MIN) l2 = 1.0;
    const double y = (fb / ncf - 5.0) / 2.0;
    const double g = exp(-0.5 * (y * y));
    for (int r = 0; r < BANK_ROWS; r++) {
        double v = 0.0;
        if (r < 12) v = (w[(r + 3) % 12] / l2) * g;  // np.roll(-3) along the chroma axis
        out[(size_t)r * CBINS_PAD + k] = v;
    }
}

void launch_chroma_bank(double* bank, hipStream_t st) {
    hipLaunchKernelGGL(chroma_bank_kernel, dim3((CBINS_PAD + 255) / 256, N_TUNING + 1), dim3(256), 0, st, bank);
}

// ------------------------------------------------------------------------------------------------
// pip_track peak test on three neighbouring magnitudes (src/chroma.rs:317-327) + the pitch-residue
// bin pitch_tuning would file it under (:342-351).  f64, no contraction: identical in both passes.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool pip_peak(float sb, float se, float sa, double ref, int c, double* mag_out,
                                         int* pb_out) {
#pragma clang fp contract(off)
    const double before = (double)sb, elem = (double)se, after = (double)sa;
    if (!(elem > ref && after <= elem && before < elem)) return false;
    const double avg = 0.5 * (after - before);
    double shift = 2.0 * elem - after - before;
    if (fabs(shift) < DBL_MIN) shift += 1.0;
    shift = avg / shift;
    const double pitch = ((double)c + shift) * 22050.0 / 8192.0;
    if (!(pitch > 0.0)) return false;  // estimate_tuning keeps p > 0 only (:370-375)
    *mag_out = elem + 0.5 * avg * shift;
    double x = log2(pitch / (440.0 / 16.0));
    x = fmod(12.0 * x, 1.0);
    if (x >= 0.5) x -= 1.0;
    const double q = (x - -0.5) / 0.01;
    int idx = (q > 0.0) ? (int)q : 0;
    if (idx > N_TUNING - 1) idx = N_TUNING - 1;
    *pb_out = idx;
    return true;
}

__device__ __forceinline__ uint32_t coarse_bin(double mag) {
    const uint32_t b = __float_as_uint((float)mag) >> 18;  // monotone in mag for mag > 0
    return b < (uint32_t)H1_BINS ? b : (uint32_t)H1_BINS - 1;
}

// ------------------------------------------------------------------------------------------------
// STFT 8192 / hop 2205
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stft8192_kernel(const float* __restrict__ pcm,
                                                       const SongDesc* __restrict__ songs, uint32_t n_songs,
                                                       const uint32_t* __restrict__ pfx_c,
                                                       const float* __restrict__ hann,
                                                       const float2* __restrict__ tw, float* __restrict__ spec,
                                                       float* __restrict__ frame_max, uint32_t* __restrict__ h1) {
    __shared__ float2 bufA[4096];
    __shared__ float2 bufB[4096];
    __shared__ float red[4];
    const uint32_t s = find_segment(pfx_c, n_songs, blockIdx.x);
    const SongDesc sd = songs[s];
    const uint32_t f = blockIdx.x - pfx_c[s];
    const float* __restrict__ x = pcm + sd.pcm_off;
    const int tid = threadIdx.x;
    const long n = (long)sd.n;

    // reflect_pad (src/utils.rs:11-24) + window (:37-39, :49)
    float* xin = reinterpret_cast<float*>(bufA);
    const long w0 = (long)f * HOP_C - W8192 / 2;
#pragma unroll 4
    for (int j = 0; j < 32; j++) {
        const int i = tid + 256 * j;
        long p = w0 + i;
        if (p < 0) p = -p;
        else if (p >= n) p = 2 * n - 2 - p;
        xin[i] = x[p] * hann[i];
    }
    __syncthreads();
    // 4096-point complex FFT: 6 radix-4 passes, ping-pong A -> B -> ... -> A
    {
        float2* src = bufA;
        float2* dst = bufB;
#pragma unroll
        for (int Ns = 1; Ns < 4096; Ns *= 4) {
#pragma unroll
            for (int q = 0; q < 4; q++) stockham_r4<4096>(src, dst, tid + 256 * q, Ns, tw, 2);
            __syncthreads();
            float2* t = src; src = dst; dst = t;
        }
    }
    // split into the 4097 real-FFT bins, magnitude in f32 (:60)
    float* mags = reinterpret_cast<float*>(bufB);
    float* row = spec + (sd.c_off + f) * (size_t)CBINS_PAD;
    float mx = 0.0f;
    for (int k = tid; k <= 4096; k += 256) {
        float m;
        if (k == 0 || k == 4096) {
            const float2 z0 = bufA[0];
            m = fabsf(k == 0 ? z0.x + z0.y : z0.x - z0.y);
        } else {
            const float2 X = real_split(bufA[k], bufA[4096 - k], tw[k]);
            m = sqrtf(X.x * X.x + X.y * X.y);
        }
        mags[k] = m;
        row[k] = m;
        mx = fmaxf(mx, m);
    }
    if (tid < CBINS_PAD - CBINS) row[CBINS + tid] = 0.0f;
    mx = wave_max(mx);
    if (lane_id() == 0) red[wave_id()] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (tid == 0) frame_max[sd.c_off + f] = mx;
    // pip_track pass 1: count peaks by coarse magnitude bin
    const double ref = 0.1 * (double)mx;
    uint32_t* hist = h1 + (size_t)s * H1_BINS;
    for (int c = PIP_LO + tid; c <= PIP_HI; c += 256) {
        double mag;
        int pb;
        if (pip_peak(mags[c - 1], mags[c], mags[c + 1], ref, c, &mag, &pb)) atomicAdd(&hist[coarse_bin(mag)], 1u);
    }
}

void launch_stft8192(const Batch& b, const Workspace& w, const DeviceTables& t, hipStream_t st) {
    if (b.tiles_c == 0) return;
    hipLaunchKernelGGL(stft8192_kernel, dim3(b.tiles_c), dim3(256), 0, st, b.pcm, b.songs, b.n_songs, b.pfx_c,
                       t.hann8192, t.tw8192, w.spec, w.frame_max, w.h1);
}

// ------------------------------------------------------------------------------------------------
// tuning: locate the coarse bins that hold the two middle order statistics
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tune_select_kernel(const SongDesc* __restrict__ songs,
                                                          const uint32_t* __restrict__ h1,
                                                          TuningState* __restrict__ tuning) {
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t s_total;
    const uint32_t s = blockIdx.x;
    const int tid = threadIdx.x;
    TuningState* ts = tuning + s;
    if (!songs[s].ok) {
        if (tid == 0) { ts->n_peaks = 0; ts->tuning_idx = -1; ts->n_cand = 0; ts->b_lo = 1; ts->b_hi = 0; ts->below = 0; }
        return;
    }
    const uint32_t* hist = h1 + (size_t)s * H1_BINS;
    constexpr int PER = H1_BINS / 256;  // 32 consecutive bins per thread
    uint32_t local[PER], sum = 0;
#pragma unroll
    for (int i = 0; i < PER; i++) { local[i] = hist[tid * PER + i]; sum += local[i]; }
    const uint32_t incl = wave_scan_incl_u32(sum);
    if (lane_id() == 63) wsum[wave_id()] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave_id(); w++) base += wsum[w];
    if (tid == 255) s_total = base + incl;
    __syncthreads();
    const uint32_t total = s_total;
    uint32_t before = base + incl - sum;  // peaks in bins below this thread's first bin
    if (total == 0) {
        if (tid == 0) { ts->n_peaks = 0; ts->tuning_idx = -1; ts->n_cand = 0; ts->b_lo = 1; ts->b_hi = 0; ts->below = 0; }
        return;
    }
    // ndarray-stats Midpoint: lower = floor(0.5*(n-1)), higher = ceil(0.5*(n-1))
    const uint32_t r_lo = (total - 1) / 2, r_hi = total - 1 - r_lo;
    if (tid == 0) { ts->n_peaks = total; ts->n_cand = 0; ts->tuning_idx = -1; }
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const uint32_t c = local[i];
        if (c) {
            if (before <= r_lo && r_lo < before + c) { ts->b_lo = tid * PER + i; ts->below = before; }
            if (before <= r_hi && r_hi < before + c) ts->b_hi = tid * PER + i;
        }
        before += c;
    }
}

void launch_tune_select(const Batch& b, const Workspace& w, hipStream_t st) {
    if (b.n_songs == 0) return;
    hipLaunchKernelGGL(tune_select_kernel, dim3(b.n_songs), dim3(256), 0, st, b.songs, w.h1, w.tuning);
}

// ------------------------------------------------------------------------------------------------
// tuning pass 2: re-run the peak test on the stored magnitudes; peaks above the median's coarse bin
// go straight into the pitch histogram, peaks inside it are kept as candidates for the exact select
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tune_pass2_kernel(const SongDesc* __restrict__ songs, uint32_t n_songs,
                                                         const uint32_t* __restrict__ pfx_ct,
                                                         const float* __restrict__ spec,
                                                         const float* __restrict__ frame_max,
                                                         TuningState* __restrict__ tuning,
                                                         uint32_t* __restrict__ hist100,
                                                         double* __restrict__ cand_mag,
                                                         uint8_t* __restrict__ cand_pb) {
    __shared__ uint32_t hist[N_TUNING];
    const uint32_t s = find_segment(pfx_ct, n_songs, blockIdx.x);
    const SongDesc sd = songs[s];
    const uint32_t tile = blockIdx.x - pfx_ct[s];
    const int tid = threadIdx.x, lane = lane_id(), wave = wave_id();
    TuningState* ts = tuning + s;
    const uint32_t b_lo = ts->b_lo, b_hi = ts->b_hi;
    if (ts->n_peaks == 0) return;
    if (tid < N_TUNING) hist[tid] = 0;
    __syncthreads();
    for (int i = 0; i < CH_TILE / 4; i++) {
        const uint32_t f = tile * CH_TILE + wave + 4 * i;
        if (f >= sd.n_c) break;
        const float* row = spec + (sd.c_off + f) * (size_t)CBINS_PAD;
        const double ref = 0.1 * (double)frame_max[sd.c_off + f];
        for (int c = PIP_LO + lane; c <= PIP_HI; c += WAVE) {
            double mag;
            int pb;
            if (pip_peak(row[c - 1], row[c], row[c + 1], ref, c, &mag, &pb)) {
                const uint32_t b = coarse_bin(mag);
                if (b > b_hi) {
                    atomicAdd(&hist[pb], 1u);
                } else if (b >= b_lo) {
                    const uint32_t slot = atomicAdd(&ts->n_cand, 1u);
                    cand_mag[sd.cand_off + slot] = mag;
                    cand_pb[sd.cand_off + slot] = (uint8_t)pb;
                }
            }
        }
    }
    __syncthreads();
    if (tid < N_TUNING && hist[tid]) atomicAdd(&hist100[(size_t)s * N_TUNING + tid], hist[tid]);
}

void launch_tune_pass2(const Batch& b, const Workspace& w, hipStream_t st) {
    if (b.tiles_ct == 0) return;
    hipLaunchKernelGGL(tune_pass2_kernel, dim3(b.tiles_ct), dim3(256), 0, st, b.songs, b.n_songs, b.pfx_ct, w.spec,
                       w.frame_max, w.tuning, w.hist100, w.cand_mag, w.cand_pb);
}

// ------------------------------------------------------------------------------------------------
// tuning final: exact order statistics among the candidates (8-bit MSD radix select on the
// order-preserving u64 image of the f64 magnitudes), Midpoint threshold, histogram, first argmax
// ------------------------------------------------------------------------------------------------
__device__ uint64_t block_radix_select(const double* __restrict__ v, uint32_t n, uint32_t rank, uint32_t* hist,
                                       uint32_t* s_digit, uint32_t* s_rank) {
    const int tid = threadIdx.x;
    uint64_t prefix = 0, mask = 0;
    for (int shift = 56; shift >= 0; shift -= 8) {
        hist[tid] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < n; i += 256) {
            const uint64_t k = f64_key(v[i]);
            if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 0xFF], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t acc = 0, d = 0;
            for (; d < 256; d++) {
                if (rank < acc + hist[d]) break;
                acc += hist[d];
            }
            *s_digit = d;
            *s_rank = rank - acc;
        }
        __syncthreads();
        prefix |= (uint64_t)(*s_digit) << shift;
        mask |= 0xFFull << shift;
        rank = *s_rank;
        __syncthreads();
    }
    return prefix;
}

__global__ __launch_bounds__(256) void tune_final_kernel(const SongDesc* __restrict__ songs,
                                                         TuningState* __restrict__ tuning,
                                                         const uint32_t* __restrict__ hist100,
                                                         const double* __restrict__ cand_mag,
                                                         const uint8_t* __restrict__ cand_pb) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t s_digit, s_rank, s_cnt_le;
    __shared__ unsigned long long s_min_gt;
    const uint32_t s = blockIdx.x;
    const int tid = threadIdx.x;
    TuningState* ts = tuning + s;
    const SongDesc sd = songs[s];
    if (!sd.ok || ts->n_peaks == 0) return;  // tuning_idx stays -1 => tuning 0.0 (src/chroma.rs:377-379)
    const uint32_t total = ts->n_peaks, nc = ts->n_cand;
    const uint32_t r_lo = (total - 1) / 2, r_hi = total - 1 - r_lo;
    const uint32_t k_lo = r_lo - ts->below, k_hi = r_hi - ts->below;
    const double* v = cand_mag + sd.cand_off;
    const uint8_t* pb = cand_pb + sd.cand_off;

    const uint64_t key_lo = block_radix_select(v, nc, k_lo, hist, &s_digit, &s_rank);
    uint64_t key_hi = key_lo;
    if (k_hi != k_lo) {
        // the next order statistic: key_lo again if it is repeated, else the smallest key above it
        if (tid == 0) { s_cnt_le = 0; s_min_gt = ~0ull; }
        __syncthreads();
        uint32_t cnt = 0;
        unsigned long long mn = ~0ull;
        for (uint32_t i = tid; i < nc; i += 256) {
            const uint64_t k = f64_key(v[i]);
            if (k <= key_lo) cnt++;
            else if (k < mn) mn = k;
        }
        atomicAdd(&s_cnt_le, cnt);
        atomicMin(&s_min_gt, mn);
        __syncthreads();
        if (s_cnt_le <= k_hi) key_hi = s_min_gt;
    }
    double thr;
    {
#pragma clang fp contract(off)
        const double lo = key_f64(key_lo), hi = key_f64(key_hi);
        thr = lo + (hi - lo) / 2.0;  // Midpoint interpolation
    }
    __syncthreads();
    if (tid < N_TUNING) hist[tid] = hist100[(size_t)s * N_TUNING + tid];
    __syncthreads();
    for (uint32_t i = tid; i < nc; i += 256)
        if (v[i] >= thr) atomicAdd(&hist[pb[i]], 1u);
    __syncthreads();
    if (tid == 0) {
        uint32_t best = 0;  // ndarray-stats argmax keeps the first maximum
        for (uint32_t k = 1; k < N_TUNING; k++)
            if (hist[k] > hist[best]) best = k;
        ts->tuning_idx = (int32_t)best;
    }
}

void launch_tune_final(const Batch& b, const Workspace& w, hipStream_t st) {
    if (b.n_songs == 0) return;
    hipLaunchKernelGGL(tune_final_kernel, dim3(b.n_songs), dim3(256), 0, st, b.songs, w.tuning, w.hist100,
                       w.cand_mag, w.cand_pb);
}

// ------------------------------------------------------------------------------------------------
// chroma_stft contraction + chroma_interval_features
// ------------------------------------------------------------------------------------------------
// templates of src/chroma.rs:139-152 as the pitch classes each column selects
static constexpr int TMPL_LEN[10] = {2, 2, 2, 2, 2, 2, 3, 3, 3, 3};
static constexpr int TMPL_PC[10][3] = {{0, 1, 0}, {0, 2, 0}, {0, 3, 0}, {0, 4, 0}, {0, 5, 0},
                                       {0, 6, 0}, {0, 4, 7}, {0, 3, 7}, {0, 3, 6}, {0, 4, 8}};

// extract_interval_features (:157-175) for one frame: sum over the 12 rotations of the product of the
// selected pitch classes (ascending row order, like Array::product over the rolled template)
template <int T>
__device__ __forceinline__ double interval_feature(const double (&c)[12]) {
    double acc = 0.0;
#pragma unroll
    for (int shift = 0; shift < 12; shift++) {
        const int r0 = (TMPL_PC[T][0] + shift) % 12, r1 = (TMPL_PC[T][1] + shift) % 12;
        if (TMPL_LEN[T] == 2) {
            acc += c[r0 < r1 ? r0 : r1] * c[r0 < r1 ? r1 : r0];
        } else {
            const int r2 = (TMPL_PC[T][2] + shift) % 12;
            const int lo = r0 < r1 ? (r0 < r2 ? r0 : r2) : (r1 < r2 ? r1 : r2);
            const int hi = r0 > r1 ? (r0 > r2 ? r0 : r2) : (r1 > r2 ? r1 : r2);
            const int mid = r0 + r1 + r2 - lo - hi;
            acc += (c[lo] * c[mid]) * c[hi];
        }
    }
    return acc;
}

__global__ __launch_bounds__(256) void chroma_kernel(const SongDesc* __restrict__ songs, uint32_t n_songs,
                                                     const uint32_t* __restrict__ pfx_ct,
                                                     const float* __restrict__ spec,
                                                     const double* __restrict__ bank,
                                                     const TuningState* __restrict__ tuning,
                                                     double* __restrict__ chroma_part) {
    __shared__ double tile_c[4][16][13];
    __shared__ double part[4][10];
    const uint32_t s = find_segment(pfx_ct, n_songs, blockIdx.x);
    const SongDesc sd = songs[s];
    const uint32_t tile = blockIdx.x - pfx_ct[s];
    const int lane = lane_id(), wave = wave_id();
    const int i16 = lane & 15, g = lane >> 4;
    const int tidx = tuning[s].tuning_idx;
    const int slot = (tidx < 0) ? N_TUNING : tidx;

    const uint32_t f0 = tile * CH_TILE + wave * 16;
    double feat[10];
#pragma unroll
    for (int t = 0; t < 10; t++) feat[t] = 0.0;

    if (f0 < sd.n_c) {  // wave-uniform
        uint32_t fj = f0 + i16;
        if (fj >= sd.n_c) fj = sd.n_c - 1;
        const double* __restrict__ arow = bank + ((size_t)slot * BANK_ROWS + i16) * CBINS_PAD + 4 * g;
        const float* __restrict__ brow = spec + (sd.c_off + fj) * (size_t)CBINS_PAD + 4 * g;
        double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 2
        for (int st = 0; st < CBINS_PAD / 16; st++) {
            const double4_t a = *reinterpret_cast<const double4_t*>(arow + 16 * st);
            const float4 b = *reinterpret_cast<const float4*>(brow + 16 * st);
            const double b0 = (double)b.x * (double)b.x, b1 = (double)b.y * (double)b.y;
            const double b2 = (double)b.z * (double)b.z, b3 = (double)b.w * (double)b.w;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a.z, b2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a.w, b3, acc, 0, 0, 0);
        }
        // C[row = g + 4r][col = i16]: rows are chroma classes, columns are frames
#pragma unroll
        for (int r = 0; r < 3; r++) tile_c[wave][i16][g + 4 * r] = acc[r];
        __builtin_amdgcn_wave_barrier();
        if (lane < 16 && f0 + lane < sd.n_c) {
            double c[12], sum = 0.0;
#pragma unroll
            for (int k = 0; k < 12; k++) { c[k] = tile_c[wave][lane][k]; sum += fabs(c[k]); }
            if (sum < DBL_MIN) sum = 1.0;          // chroma_stft column normalisation (:404-410)
            double esum = 0.0;
#pragma unroll
            for (int k = 0; k < 12; k++) { c[k] = exp((c[k] / sum) * 15.0); esum += fabs(c[k]); }
            if (esum < 0.0001) esum = 1.0;         // normalize_feature_sequence (:177-188)
#pragma unroll
            for (int k = 0; k < 12; k++) c[k] /= esum;
            feat[0] = interval_feature<0>(c); feat[1] = interval_feature<1>(c);
            feat[2] = interval_feature<2>(c); feat[3] = interval_feature<3>(c);
            feat[4] = interval_feature<4>(c); feat[5] = interval_feature<5>(c);
            feat[6] = interval_feature<6>(c); feat[7] = interval_feature<7>(c);
            feat[8] = interval_feature<8>(c); feat[9] = interval_feature<9>(c);
        }
    }
#pragma unroll
    for (int t = 0; t < 10; t++) {
        // lanes 16..63 hold zeros; sum the 16 frames of this wave
        double v = feat[t];
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
        if (lane == 0) part[wave][t] = v;
    }
    __syncthreads();
    if (threadIdx.x < 10) {
        const int t = threadIdx.x;
        chroma_part[(size_t)blockIdx.x * 10 + t] = ((part[0][t] + part[1][t]) + part[2][t]) + part[3][t];
    }
}

void launch_chroma(const Batch& b, const Workspace& w, const DeviceTables& t, hipStream_t st) {
    if (b.tiles_ct == 0) return;
    hipLaunchKernelGGL(chroma_kernel, dim3(b.tiles_ct), dim3(256), 0, st, b.songs, b.n_songs, b.pfx_ct, w.spec,
                       t.chroma_bank, w.tuning, w.chroma_part);
}

}  // namespace bg
