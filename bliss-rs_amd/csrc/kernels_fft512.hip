// kernels_fft512.hip -- the W=512 phase-vocoder frames shared by the timbral and tempo descriptors.
//
// Reference: PVoc::do_ (src/aubio.rs:182-264) / PVocTempo::do_ (:338-425) feed a sliding 512-sample
// buffer (zero history) with `hop` new samples, apply the hanningz window, fftshift, c2c FFT-512.
// Net effect: FFT frame k covers x[(k+1)*128-512, (k+1)*128), zeros for
// negative indices; timbral frame k == FFT frame k; tempo frame j == FFT frame 2j+1.  fftshift only
// multiplies X[k] by (-1)^k, so magnitudes are unchanged and it is not performed here.
//
// Per frame this kernel produces
//   * spectral centroid / rolloff / flatness on the reference's "buggy" 256-bin vector whose bin 255
//     is |Re X[256]| (src/aubio.rs:240-261, 16-58; src/timbral.rs:154-209; src/utils.rs:101-117)
//   * for odd frames, the SpecFlux onset value over the correct 257 bins (src/aubio.rs:455-467).
//
// Mapping: one wavefront per frame, frames of a wave processed in increasing order so the previous
// tempo frame's magnitudes stay in LDS.  The 512 real samples are packed as 256 complex values and
// transformed by four Stockham radix-4 passes in LDS (one butterfly per lane), then split.
#include "device_utils.hpp"
#include "internal.hpp"

namespace bg {

constexpr int FRAMES_PER_WAVE = F512_TILE / 4;

struct WaveLds {
    float2 a[256];
    float2 b[256];
    float prevmag[260];  // 257 magnitudes of the previous tempo frame (+pad)
};

// Computes the 257 magnitudes of FFT frame k of song sd into registers:
// lane l gets |X[4l..4l+3]| in m[0..3]; the Nyquist magnitude |X[256]| is returned in *nyq (all lanes).
__device__ __forceinline__ void fft512_frame(const float* __restrict__ x, long k, WaveLds& lds,
                                             const float* __restrict__ hannz, const float2* __restrict__ tw512,
                                             float m[4], float* nyq) {
    const int lane = lane_id();
    const long start = (k + 1) * HOP_T - W512;
    float* xin = reinterpret_cast<float*>(lds.a);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int i = lane + 64 * j;
        const long idx = start + i;
        const float v = (idx >= 0) ? x[idx] : 0.0f;
        xin[i] = v * hannz[i];
    }
    __builtin_amdgcn_wave_barrier();
    stockham_r4<256>(lds.a, lds.b, lane, 1, tw512, 2);
    __builtin_amdgcn_wave_barrier();
    stockham_r4<256>(lds.b, lds.a, lane, 4, tw512, 2);
    __builtin_amdgcn_wave_barrier();
    stockham_r4<256>(lds.a, lds.b, lane, 16, tw512, 2);
    __builtin_amdgcn_wave_barrier();
    stockham_r4<256>(lds.b, lds.a, lane, 64, tw512, 2);
    __builtin_amdgcn_wave_barrier();
    const float2* z = lds.a;
    const float2 z0 = z[0];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int kk = 4 * lane + e;
        const float2 zk = z[kk];
        const float2 zm = z[(256 - kk) & 255];
        const float2 X = real_split(zk, zm, tw512[kk]);
        m[e] = (kk == 0) ? fabsf(z0.x + z0.y) : sqrtf(X.x * X.x + X.y * X.y);
    }
    *nyq = fabsf(z0.x - z0.y);
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(256) void fft512_kernel(const float* __restrict__ pcm,
                                                     const SongDesc* __restrict__ songs, uint32_t n_songs,
                                                     const uint32_t* __restrict__ pfx_f,
                                                     const float* __restrict__ hannz,
                                                     const float2* __restrict__ tw512, float* __restrict__ centroid,
                                                     float* __restrict__ rolloff, float* __restrict__ flatness,
                                                     float* __restrict__ flux) {
    __shared__ WaveLds lds_all[4];
    const uint32_t s = find_segment(pfx_f, n_songs, blockIdx.x);
    const SongDesc sd = songs[s];
    const uint32_t tile = blockIdx.x - pfx_f[s];
    const float* __restrict__ x = pcm + sd.pcm_off;
    const int lane = lane_id(), wave = wave_id();
    WaveLds& lds = lds_all[wave];

    const long k_begin = (long)tile * F512_TILE + (long)wave * FRAMES_PER_WAVE;  // even
    if (k_begin >= (long)sd.n_f) return;
    const long k_end = (k_begin + FRAMES_PER_WAVE < (long)sd.n_f) ? k_begin + FRAMES_PER_WAVE : (long)sd.n_f;

    float m[4], nyq;
    // halo: magnitudes of the previous tempo frame (FFT frame k_begin-1), zeros before the song starts
    if (k_begin >= 1) {
        fft512_frame(x, k_begin - 1, lds, hannz, tw512, m, &nyq);
#pragma unroll
        for (int e = 0; e < 4; e++) lds.prevmag[4 * lane + e] = m[e];
        if (lane == 0) lds.prevmag[256] = nyq;
    } else {
#pragma unroll
        for (int e = 0; e < 4; e++) lds.prevmag[4 * lane + e] = 0.0f;
        if (lane == 0) lds.prevmag[256] = 0.0f;
    }
    __builtin_amdgcn_wave_barrier();

    for (long k = k_begin; k < k_end; k++) {
        fft512_frame(x, k, lds, hannz, tw512, m, &nyq);

        if (k & 1) {  // tempo frame j = (k-1)/2 : SpecFlux over bins 0..256
            float f = 0.0f;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float old = lds.prevmag[4 * lane + e];
                if (m[e] > old) f += m[e] - old;
                lds.prevmag[4 * lane + e] = m[e];
            }
            if (lane == 0) {
                const float old = lds.prevmag[256];
                if (nyq > old) f += nyq - old;
                lds.prevmag[256] = nyq;
            }
            f = wave_sum(f);
            const long j = (k - 1) >> 1;
            if (lane == 0 && j < (long)sd.n_b) flux[sd.b_off + j] = f;
        }

        if (k < (long)sd.n_t) {
            // the 256-bin vector the reference's timbral path sees: bin 255 := |Re X[256]|
            if (lane == 63) m[3] = nyq;
            float sum = 0.0f, wsum = 0.0f, sq[4], sqsum = 0.0f;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                sum += m[e];
                wsum += (float)(4 * lane + e) * m[e];
                sq[e] = m[e] * m[e];
                sqsum += sq[e];
            }
            const float total = wave_sum(sum);
            const float wtotal = wave_sum(wsum);
            // spectral_centroid (src/aubio.rs:16-29) then bin_to_freq (:68-71)
            const float cbin = (total == 0.0f) ? 0.0f : wtotal / total;
            const float freq_per_bin = (float)SAMPLE_RATE / (float)W512;
            // spectral_rolloff (src/aubio.rs:36-58): bins consumed until the running energy reaches 95 %
            const float incl = wave_scan_incl(sqsum);
            const float cum_total = __shfl(incl, 63, WAVE);
            float rbin = 0.0f;
            if (cum_total != 0.0f) {
                const float thr = cum_total * 0.95f;
                float run = incl - sqsum;
                uint32_t below = 0;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    run += sq[e];
                    below += (run < thr) ? 1u : 0u;
                }
                const uint32_t c = wave_sum(below);
                rbin = (float)((c < 256u) ? c + 1u : 256u);
            }
            // geometric_mean (src/utils.rs:101-117): groups of 8 in f64, exponents and mantissas apart
            double p = ((double)m[0] * (double)m[1]) * ((double)m[2] * (double)m[3]);
            const double p_hi = __shfl_down(p, 1, WAVE);
            int expo = 0;
            double mant = 1.0;
            int zero = 0;
            if ((lane & 1) == 0) {
                double g = p * 3.273390607896142e150;
                g *= p_hi;
                if (g == 0.0) zero = 1;
                const uint64_t bits = (uint64_t)__double_as_longlong(g);
                expo = (int)(bits >> 52);
                mant = __longlong_as_double((long long)((bits & 0xFFFFFFFFFFFFFull) | 0x3FF0000000000000ull));
            }
            const int any_zero = __any(zero);
            const int exps = wave_sum(expo);
            const double mants = wave_prod(mant);
            float flat = 0.0f;
            if (!any_zero) {
                const float geo = exp2f((log2f((float)mants) + (float)exps) / 256.0f - (1023.0f + 500.0f) / 8.0f);
                if (geo != 0.0f) flat = geo / (total / 256.0f);
            }
            if (lane == 0) {
                centroid[sd.t_off + k] = freq_per_bin * fmaxf(cbin, 0.0f);
                rolloff[sd.t_off + k] = freq_per_bin * fmaxf(rbin, 0.0f);
                flatness[sd.t_off + k] = flat;
            }
        }
    }
}

void launch_fft512(const Batch& b, const Workspace& w, const DeviceTables& t, hipStream_t st) {
    if (b.tiles_f == 0) return;
    hipLaunchKernelGGL(fft512_kernel, dim3(b.tiles_f), dim3(256), 0, st, b.pcm, b.songs, b.n_songs, b.pfx_f,
                       t.hannz512, t.tw512, w.centroid, w.rolloff, w.flatness, w.flux);
}

}  // namespace bg
