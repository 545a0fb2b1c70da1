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
// Mapping: a 16-lane group owns one frame at a time (4 frames per wavefront, 16 per workgroup).  The
// 512 real samples are packed as 256 complex values = 16 x 16: each lane holds 16 of them in registers,
// runs a radix-16 pass, transposes inside its group through a padded LDS tile, runs the second radix-16
// pass, and a second LDS round trip re-orders the spectrum so that lane l ends up with the 16
// CONSECUTIVE bins 16l..16l+15 (needed by the rolloff prefix sum) and with Z[256-k] for the real-input
// split.  Reductions stay inside the 16-lane row (DPP).  A group walks FRAMES_PER_GROUP consecutive
// frames so the previous tempo frame's magnitudes stay in registers (one halo FFT per group).
#include <stdlib.h>

#include "device_utils.hpp"
#include "fft_r16.hpp"
#include "internal.hpp"

namespace bg {

constexpr int GROUPS_PER_WG = 16;
constexpr int FRAMES_PER_GROUP = F512_TILE / GROUPS_PER_WG;
constexpr int GRP_PITCH = 272;  // float2 per group tile: 16 rows of 17, and == 128 B (mod 256 B) between groups

typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 buf_load_f2(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return mk(__uint_as_float(v.x), __uint_as_float(v.y));
}

// ---- reductions across the 16 lanes of a DPP row ----
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
constexpr int DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_XOR2 = 0x4E, DPP_ROW_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140;
template <typename T>
__device__ __forceinline__ T row16_sum(T v) {  // every lane of the row gets the row total
    v += dpp_mov<DPP_QUAD_XOR1>(v);
    v += dpp_mov<DPP_QUAD_XOR2>(v);
    v += dpp_mov<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_mov<DPP_ROW_MIRROR>(v);
    return v;
}
__device__ __forceinline__ float row16_scan_incl(float v) {  // row_shr:n with zero fill
    v += dpp_mov<0x111>(v);
    v += dpp_mov<0x112>(v);
    v += dpp_mov<0x114>(v);
    v += dpp_mov<0x118>(v);
    return v;
}

struct FrameMags {
    float m[16];  // |X[16l + e]|
    float nyq;    // |X[256]| (every lane)
};

// 257 magnitudes of FFT frame k: lane l of the group gets bins 16l..16l+15
// constant tables staged once per workgroup in LDS (broadcast reads, no VMEM traffic per frame)
struct Tables512 {
    f2 win[256];    // (hannz[2n], hannz[2n+1]), n = 16 n1 + l
    f2 tw256[256];  // W_256^(l*k1) at [16 k1 + l]
    f2 tw512[256];  // W_512^(16 l + e) at [16 e + l]: lane-contiguous (a [16 l + e] layout is a 16-way bank conflict)
};

// raw samples of one frame: row n1 of lane l = (x[s + 32 n1 + 2l], x[s + 32 n1 + 2l + 1]); samples before the song
// start are 0 (the reference's zero-initialised sliding buffer); s is a multiple of 128, so pairs never straddle 0
template <int ABL>
__device__ __forceinline__ void fft512_load(__amdgpu_buffer_rsrc_t r_x, long rel_start, int l, f2 (&raw)[16]) {
    if (ABL == 3) {
#pragma unroll
        for (int n1 = 0; n1 < 16; n1++) raw[n1] = mk((float)(l + n1), 1.0f);
    } else if (rel_start >= 0) {
        const uint32_t xoff = (uint32_t)((rel_start + 2 * l) * 4);
#pragma unroll
        for (int n1 = 0; n1 < 16; n1++) raw[n1] = buf_load_f2(r_x, xoff, 128u * n1);
    } else {
#pragma unroll
        for (int n1 = 0; n1 < 16; n1++) {
            const long idx = rel_start + 32 * n1 + 2 * l;
            raw[n1] = mk(0.0f, 0.0f);
            if (idx >= 0) raw[n1] = buf_load_f2(r_x, (uint32_t)(idx * 4), 0);
        }
    }
}

template <int ABL>
__device__ __forceinline__ void fft512_compute(const f2 (&raw)[16], int l, f2* tile, const Tables512* tabs, FrameMags& out) {
    f2 v[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; n1++) v[n1] = raw[n1] * tabs->win[16 * n1 + l];
    radix16(v);  // over n1 -> A[k1] at v[R16(k1)]
#pragma unroll
    for (int k1 = 1; k1 < 16; k1++) v[R16(k1)] = cmul_pk(v[R16(k1)], tabs->tw256[16 * k1 + l]);  // W_256^(l*k1)
#pragma unroll
    for (int k1 = 0; k1 < 16; k1++) tile[k1 * 17 + l] = v[R16(k1)];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int n2 = 0; n2 < 16; n2++) v[n2] = tile[l * 17 + n2];
    __builtin_amdgcn_wave_barrier();
    if (ABL != 4) radix16(v);  // over n2 -> Z[l + 16 k2] at v[R16(k2)]
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++) tile[k2 * 17 + l] = v[R16(k2)];  // Z[k] at k + (k >> 4)
    __builtin_amdgcn_wave_barrier();
    const f2 z0 = tile[0];
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const f2 zk = tile[l * 17 + e];
        // Z[256 - k], k = 16 l + e  (k = 0 pairs with itself)
        const int mi = (e == 0) ? ((l == 0) ? 0 : 17 * (16 - l)) : (17 * (15 - l) + 16 - e);
        const f2 zm = tile[mi];
        if (ABL == 2) out.m[e] = zk.x + zm.y;
        else if (ABL == 5) out.m[e] = 0.5f * split_one_sq(zk, zm, tabs->tw512[16 * e + l]);
        else if (ABL == 6) out.m[e] = mag_from_sq(split_one_sq(zk, zm, mk(0.6f, 0.8f)));
        else if (ABL == 7) out.m[e] = mag_from_sq(split_one_sq(zk, zk, tabs->tw512[16 * e + l]));
        else out.m[e] = mag_from_sq(split_one_sq(zk, zm, tabs->tw512[16 * e + l]));  // W_512^k, k = 16 l + e
    }
    // Z is halved (half window): X[0] = 2 (Re Z[0] + Im Z[0]), X[256] = 2 (Re Z[0] - Im Z[0])
    if (l == 0) out.m[0] = 2.0f * fabsf(z0.x + z0.y);
    out.nyq = 2.0f * fabsf(z0.x - z0.y);
    __builtin_amdgcn_wave_barrier();
}

template <int ABL>  // ABL != 0: timing ablations (developer aid, BLISSGPU_ABL512), results are wrong
__global__ __launch_bounds__(256, 3) void fft512_kernel(const float* __restrict__ pcm,
                                                        const SongDesc* __restrict__ songs, uint32_t n_songs,
                                                        const uint32_t* __restrict__ pfx_f,
                                                        const float* __restrict__ hannz,
                                                        const float2* __restrict__ tw512,
                                                        float* __restrict__ centroid, float* __restrict__ rolloff,
                                                        float* __restrict__ flatness, float* __restrict__ flux) {
    __shared__ f2 lds[GROUPS_PER_WG * GRP_PITCH];
    __shared__ Tables512 tabs_s;
    {
        const int t = threadIdx.x;  // t = 16 a + b
        const float2 a = tw512[2 * (((t >> 4) * (t & 15)) & 255)], b = tw512[16 * (t & 15) + (t >> 4)];
        tabs_s.win[t] = mk(hannz[2 * t], hannz[2 * t + 1]);
        tabs_s.tw256[t] = mk(a.x, a.y);  // W_256^(k1*l) = W_512^(2 k1 l)
        tabs_s.tw512[t] = mk(b.x, b.y);
    }
    __syncthreads();
    const Tables512* tabs = &tabs_s;
    const uint32_t s = find_segment(pfx_f, n_songs, blockIdx.x);
    const SongDesc sd = songs[s];
    const uint32_t tile_idx = blockIdx.x - pfx_f[s];
    const int grp = threadIdx.x >> 4, l = threadIdx.x & 15;
    f2* tile = lds + grp * GRP_PITCH;

    const long k_begin = (long)tile_idx * F512_TILE + (long)grp * FRAMES_PER_GROUP;  // even
    const long k_end = (k_begin + FRAMES_PER_GROUP < (long)sd.n_f) ? k_begin + FRAMES_PER_GROUP : (long)sd.n_f;

    // descriptor over this workgroup's slice of the song (keeps lane offsets 32-bit for any song length);
    // offsets before the song start wrap to huge values and read 0 = the reference's zero history
    const long tile_first = (long)tile_idx * F512_TILE * HOP_T - (W512 - HOP_T) - HOP_T;  // sample of frame (tile*T - 1)
    const long base = tile_first > 0 ? tile_first : 0;
    const uint64_t avail = (sd.n - (uint64_t)base) * 4;
    const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(pcm + sd.pcm_off + base), 0, (uint32_t)(avail < 0xFFFFFFFFull ? avail : 0xFFFFFFFFull), 0x00020000);

    FrameMags cur, prev;
    const bool active = k_begin < (long)sd.n_f;
    // halo: magnitudes of the previous tempo frame (FFT frame k_begin - 1); zeros before the song starts
    f2 raw[16];
    if (active && k_begin >= 1) {
        fft512_load<ABL>(r_x, k_begin * HOP_T - W512 - base, l, raw);
        fft512_compute<ABL>(raw, l, tile, tabs, prev);
    } else {
#pragma unroll
        for (int e = 0; e < 16; e++) prev.m[e] = 0.0f;
        prev.nyq = 0.0f;
    }

    // Consecutive frames share 384 of their 512 samples: the raw rows stay in registers, every frame shifts them by
    // four rows and loads only the four new ones -- issued a whole frame ahead, so their latency is off the critical path.
    if (active) fft512_load<ABL>(r_x, (k_begin + 1) * HOP_T - W512 - base, l, raw);
    for (long k = k_begin; k < k_end; k++) {
        fft512_compute<ABL>(raw, l, tile, tabs, cur);
        if (k + 1 < k_end) {
#pragma unroll
            for (int n1 = 0; n1 < 12; n1++) raw[n1] = raw[n1 + 4];
            // rows 12..15 of frame k + 1 = samples [(k + 2) * 128 - 128, (k + 2) * 128): never before the song start
            // (the lane offset addresses row 12 itself: a negative frame start must not wrap the 32-bit lane offset,
            // the descriptor's range check does not see the scalar offset)
            const uint32_t xoff = (uint32_t)(((k + 2) * HOP_T - HOP_T - base + 2 * l) * 4);
#pragma unroll
            for (int n1 = 12; n1 < 16; n1++) raw[n1] = (ABL == 3) ? mk((float)(l + n1), 1.0f) : buf_load_f2(r_x, xoff, 128u * (n1 - 12));
        }

        if (k & 1) {  // tempo frame j = (k-1)/2 : SpecFlux over bins 0..256
            float f = 0.0f;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                if (cur.m[e] > prev.m[e]) f += cur.m[e] - prev.m[e];
                prev.m[e] = cur.m[e];
            }
            if (l == 0 && cur.nyq > prev.nyq) f += cur.nyq - prev.nyq;
            prev.nyq = cur.nyq;
            f = row16_sum(f);
            const long j = (k - 1) >> 1;
            if (l == 0 && j < (long)sd.n_b) flux[sd.b_off + j] = f;
        }

        if (ABL != 1 && k < (long)sd.n_t) {
            // the 256-bin vector the reference's timbral path sees: bin 255 := |Re X[256]|
            if (l == 15) cur.m[15] = cur.nyq;
            float sum = 0.0f, wsum = 0.0f, sqsum = 0.0f;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                sum += cur.m[e];
                wsum += (float)(16 * l + e) * cur.m[e];
                sqsum += cur.m[e] * cur.m[e];
            }
            const float total = row16_sum(sum);
            const float wtotal = row16_sum(wsum);
            // spectral_centroid (src/aubio.rs:16-29) then bin_to_freq (:68-71)
            const float cbin = (total == 0.0f) ? 0.0f : wtotal / total;
            const float freq_per_bin = (float)SAMPLE_RATE / (float)W512;
            // spectral_rolloff (src/aubio.rs:36-58): bins consumed until the running energy reaches 95 %
            const float incl = row16_scan_incl(sqsum);
            const float cum_total = row16_sum(sqsum);
            float rbin = 0.0f;
            if (cum_total != 0.0f) {
                const float thr = cum_total * 0.95f;
                float run = incl - sqsum;
                int below = 0;
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    run += cur.m[e] * cur.m[e];
                    below += (run < thr) ? 1 : 0;
                }
                const int c = row16_sum(below);
                rbin = (float)((c < 256) ? c + 1 : 256);
            }
            // geometric_mean (src/utils.rs:101-117): groups of 8 in f64, exponents and mantissas apart
            int expo = 0, zero = 0;
            double mant = 1.0;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const float* c8 = cur.m + 8 * h;
                double g = ((double)c8[0] * (double)c8[1]) * ((double)c8[2] * (double)c8[3]);
                g *= 3.273390607896142e150;
                g *= ((double)c8[4] * (double)c8[5]) * ((double)c8[6] * (double)c8[7]);
                if (g == 0.0) zero = 1;
                const uint64_t bits = (uint64_t)__double_as_longlong(g);
                expo += (int)(bits >> 52);
                mant *= __longlong_as_double((long long)((bits & 0xFFFFFFFFFFFFFull) | 0x3FF0000000000000ull));
            }
            const int any_zero = row16_sum(zero);
            const int exps = row16_sum(expo);
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) mant *= __shfl_xor(mant, off, 16);
            float flat = 0.0f;
            if (!any_zero) {
                const float geo = exp2f((log2f((float)mant) + (float)exps) / 256.0f - (1023.0f + 500.0f) / 8.0f);
                if (geo != 0.0f) flat = geo / (total / 256.0f);
            }
            if (l == 0) {
                centroid[sd.t_off + k] = freq_per_bin * fmaxf(cbin, 0.0f);
                rolloff[sd.t_off + k] = freq_per_bin * fmaxf(rbin, 0.0f);
                flatness[sd.t_off + k] = flat;
            }
        }
    }
}

void launch_fft512(const Batch& b, const Workspace& w, const DeviceTables& t, hipStream_t st) {
    if (b.tiles_f == 0) return;
    static const int abl = getenv("BLISSGPU_ABL512") ? atoi(getenv("BLISSGPU_ABL512")) : 0;
#define LAUNCH_F512(A) hipLaunchKernelGGL(fft512_kernel<A>, dim3(b.tiles_f), dim3(256), 0, st, b.pcm, b.songs, b.n_songs, \
                                          b.pfx_f, t.hannz512, t.tw512, w.centroid, w.rolloff, w.flatness, w.flux)
    if (abl == 1) LAUNCH_F512(1);
    else if (abl == 2) LAUNCH_F512(2);
    else if (abl == 3) LAUNCH_F512(3);
    else if (abl == 4) LAUNCH_F512(4);
    else if (abl == 5) LAUNCH_F512(5);
    else if (abl == 6) LAUNCH_F512(6);
    else if (abl == 7) LAUNCH_F512(7);
    else LAUNCH_F512(0);
#undef LAUNCH_F512
}

}  // namespace bg
