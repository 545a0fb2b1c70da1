// kernels_pcm.hip -- PCM in and out of the analysis format:
//   * raw decoder output (s16 / s32 / f32, interleaved channels, any sample rate) -> the mono 22 050 Hz f32 PCM
//     Song::analyze takes: widening, downmix and libswresample's default resampler, bit for bit
//   * Philox4x32-10 white-noise synthesis for the benchmark (no reference counterpart).
// (The per-block PCM statistics -- sums of squares, zero crossings -- are computed by the FFT-512 kernel, which
// already holds every sample in registers: kernels_fft512.hip.)
#include "../../include/blissgpu.h"
#include "device_utils.hpp"
#include "internal.hpp"
#include "resample.hpp"

namespace bg {

// ---- raw decoder output -> the mono f32 PCM Song::analyze takes (the PCM feed) ----
//   s16 -> f32   : sample / 32768 (exact in f32), FFmpeg's AV_SAMPLE_FMT_S16 -> FLT conversion as used by the reference's
//                  decoder (src/song/decoder/ffmpeg.rs:36-109); halves the PCIe bytes of the feed
//   stereo -> mono: (L + R) * SQRT_2 / 2 in f32, in exactly this order (src/song/decoder/symphonia.rs:281-285; "recovers
//                  the exact behavior of ffmpeg": both decoders pin the result on data/s16_stereo_22_5kHz.flac to
//                  Adler-32 0x1d7b2d6d, ffmpeg.rs:448-452, symphonia.rs:594-610)
//   > 2 channels : sequential f32 sum / channel count (symphonia.rs:291-297)
//   s32 -> f32   : (float)sample * 2^-31, FFmpeg's AV_SAMPLE_FMT_S32 -> FLT conversion (24-bit FLAC arrives left-justified in
//                  32 bits; the int -> float conversion rounds to nearest even like cvtsi2ss)
__device__ __forceinline__ float pcm_sample(const int16_t* p, uint64_t i) { return (float)p[i] * (1.0f / 32768.0f); }
__device__ __forceinline__ float pcm_sample(const int32_t* p, uint64_t i) { return (float)p[i] * (1.0f / 2147483648.0f); }
__device__ __forceinline__ float pcm_sample(const float* p, uint64_t i) { return p[i]; }

template <typename SampleT>
__global__ __launch_bounds__(256) void pcm_convert_kernel(const SampleT* __restrict__ in, uint32_t channels,
                                                          float* __restrict__ out, uint64_t frames) {
#pragma clang fp contract(off)
    const uint64_t i0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= frames) return;
    float r[4];
    const int nf = frames - i0 < 4 ? (int)(frames - i0) : 4;
    if (channels == 1) {
        for (int k = 0; k < nf; k++) r[k] = pcm_sample(in, i0 + k);
    } else if (channels == 2) {
        for (int k = 0; k < nf; k++) {
            const float l = pcm_sample(in, 2 * (i0 + k)), rr = pcm_sample(in, 2 * (i0 + k) + 1);
            r[k] = (l + rr) * 1.41421356237309504880f / 2.0f;
        }
    } else {
        for (int k = 0; k < nf; k++) {
            float acc = 0.0f;
            for (uint32_t c = 0; c < channels; c++) acc = acc + pcm_sample(in, (uint64_t)channels * (i0 + k) + c);
            r[k] = acc / (float)channels;
        }
    }
    if (nf == 4 && ((reinterpret_cast<uintptr_t>(out + i0) & 15) == 0)) {
        *reinterpret_cast<float4*>(out + i0) = make_float4(r[0], r[1], r[2], r[3]);
    } else {
        for (int k = 0; k < nf; k++) out[i0 + k] = r[k];
    }
}

void launch_pcm_convert(const void* in, int sample_format, uint32_t channels, float* out, uint64_t frames, hipStream_t st) {
    if (frames == 0) return;
    const dim3 grid((uint32_t)((frames + 1023) / 1024));
    if (sample_format == BLISSGPU_SAMPLE_S16)
        hipLaunchKernelGGL(pcm_convert_kernel<int16_t>, grid, dim3(256), 0, st, (const int16_t*)in, channels, out, frames);
    else if (sample_format == BLISSGPU_SAMPLE_S32)
        hipLaunchKernelGGL(pcm_convert_kernel<int32_t>, grid, dim3(256), 0, st, (const int32_t*)in, channels, out, frames);
    else
        hipLaunchKernelGGL(pcm_convert_kernel<float>, grid, dim3(256), 0, st, (const float*)in, channels, out, frames);
}

// ------------------------------------------------------------------------------------------------------------------
// Sample-rate conversion to 22 050 Hz: libswresample's default resampler as FFmpegDecoder drives it
// (src/song/decoder/ffmpeg.rs:36-109), reproduced bit for bit.  resample.hpp has the plan and the filter bank; this
// kernel is the convolution  out[k] = sum_i bank[phase_k][i] * x[first_k + i]:
//   * the sum in the order of FFmpeg's ff_resample_common_float_fma3 (what runs on every x86-64 with AVX2 + FMA3):
//     eight accumulators, tap i into accumulator i mod 8 by fused multiply-add in increasing i, then (j) + (j + 4),
//     (0) + (2) and (1) + (3), and the sum of those two
//   * x mirrored about sample 0 in front of the stream (invert_initial_buffer) and about the last sample behind it
//     (resample_flush: the edge sample repeats)
//   * two channels: each channel through the resampler FIRST (swr_init's resample_first: 1 * 1 / 2 - 1 < 22050 / in - 1 in its
//     integer arithmetic), then the matrix l * sqrt(1/2) + r * sqrt(1/2) in float, unfused (no normalisation for float
//     output); more than two channels have no FFmpeg pin (the matrix depends on the layout) and follow the reference's other
//     decoder: the sequential channel mean first (src/song/decoder/symphonia.rs:291-297), then the resampler
// Pinned by the reference's Adler-32 decoder tests: 0xa0f8b8af (data/s32_mono_44_1_kHz.flac), 0xbbcba1cf
// (s32_stereo_44_1_kHz.flac), 0xd594429c (no_channel.wav) -- ffmpeg.rs:433-445, 471-476.
//
// A workgroup produces a tile of consecutive outputs: the input span the tile needs is staged in LDS once (coalesced
// loads, converted to f32, the mirrors resolved there), the filter bank too when it fits (44.1 kHz: 66 floats; 48 kHz:
// 147 phases x 72 taps = 42 KB); a thread computes four outputs.  The feed is PCIe-bound: a 3-minute 44.1 kHz stereo s16
// song is 32 MB on the link (0.6 ms) and 0.06 ms of resample2_kernel below (0.12 ms of this general kernel).
// ------------------------------------------------------------------------------------------------------------------
constexpr int RS_THREADS = 256, RS_PER_THREAD = 4, RS_TILE = RS_THREADS * RS_PER_THREAD;
constexpr int RS_SPAN_CAP = 10240;  // floats of input span per tile at most (the host sizes the tile to fit)
constexpr int RS_LDS_CAP = 16384;   // floats of LDS per workgroup: the span, then the filter bank when it fits behind it
                                    // (larger banks are read through the caches)

struct ResampleArgs {
    const void* in;
    float* out;
    const float* bank;       // [phase_count][taps]
    uint64_t frames, n_out;
    uint64_t dst_incr, src_incr;
    uint32_t channels, tile;  // outputs per workgroup (<= RS_TILE)
    int taps, phase_count, center;
    int span_cap;             // floats of LDS reserved for the span
    int bank_in_lds;
};

// one input frame as the resampler sees it: channel `c` of a stereo pair, the only channel, or the mean of many
template <typename SampleT>
__device__ __forceinline__ float rs_input(const SampleT* in, uint32_t channels, uint64_t frame, int c) {
#pragma clang fp contract(off)
    if (channels == 1) return pcm_sample(in, frame);
    if (channels == 2) return pcm_sample(in, 2 * frame + (uint64_t)c);
    float acc = 0.0f;
    for (uint32_t q = 0; q < channels; q++) acc = acc + pcm_sample(in, (uint64_t)channels * frame + q);
    return acc / (float)channels;
}

template <typename SampleT>
__global__ __launch_bounds__(RS_THREADS) void resample_kernel(ResampleArgs a) {
#pragma clang fp contract(off)
    extern __shared__ float rs_lds[];
    float* span = rs_lds;
    float* lbank = rs_lds + a.span_cap;
    const SampleT* __restrict__ in = reinterpret_cast<const SampleT*>(a.in);
    const int t = threadIdx.x;
    const uint64_t k0 = (uint64_t)blockIdx.x * a.tile;
    if (k0 >= a.n_out) return;
    const uint64_t k1 = k0 + a.tile < a.n_out ? k0 + a.tile : a.n_out;  // exclusive
    const uint64_t pc = (uint64_t)a.phase_count;
    auto first_tap = [&](uint64_t k, int* phase) -> int64_t {
        const uint64_t pos = k * a.dst_incr / a.src_incr;  // k < 2^32, dst_incr < 2^31: no overflow
        *phase = (int)(pos % pc);
        return (int64_t)(pos / pc) - a.center;
    };
    int ph_dummy;
    const int64_t lo = first_tap(k0, &ph_dummy);
    const int span_len = (int)(first_tap(k1 - 1, &ph_dummy) + a.taps - lo);  // <= span_cap by the host's choice of tile
    if (a.bank_in_lds)
        for (int i = t; i < a.phase_count * a.taps; i += RS_THREADS) lbank[i] = a.bank[i];
    const int passes = a.channels == 2 ? 2 : 1;
    float res[2][RS_PER_THREAD];
    for (int c = 0; c < passes; c++) {
        if (c) __syncthreads();  // every thread is done with the previous channel's span
        for (int i = t; i < span_len; i += RS_THREADS) {
            int64_t s = lo + i;
            if (s < 0) s = -s;                                                     // x[-j] = x[j]
            else if ((uint64_t)s >= a.frames) s = 2 * (int64_t)a.frames - 1 - s;   // x[n + j] = x[n - 1 - j]
            span[i] = rs_input(in, a.channels, (uint64_t)s, c);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RS_PER_THREAD; j++) {
            const uint64_t k = k0 + (uint64_t)(t + RS_THREADS * j);
            res[c][j] = 0.0f;
            if (k < k1) {
                int ph;
                const int64_t first = first_tap(k, &ph);
                const float* __restrict__ x = span + (first - lo);
                const float* __restrict__ f = (a.bank_in_lds ? lbank : a.bank) + (size_t)ph * a.taps;
                float acc[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                int i = 0;
                for (; i + 8 <= a.taps; i += 8) {
#pragma unroll
                    for (int u = 0; u < 8; u++) acc[u] = __builtin_fmaf(x[i + u], f[i + u], acc[u]);
                }
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (i + u < a.taps) acc[u] = __builtin_fmaf(x[i + u], f[i + u], acc[u]);
                const float b0 = acc[0] + acc[4], b1 = acc[1] + acc[5], b2 = acc[2] + acc[6], b3 = acc[3] + acc[7];
                const float c0 = b0 + b2, c1 = b1 + b3;
                res[c][j] = c0 + c1;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < RS_PER_THREAD; j++) {
        const uint64_t k = k0 + (uint64_t)(t + RS_THREADS * j);
        if (k < k1) {
            float y = res[0][j];
            if (passes == 2) {
                const float m = 0.70710678118654752440f;  // (float)M_SQRT1_2, both matrix coefficients
                const float l = res[0][j] * m, r = res[1][j] * m;
                y = l + r;
            }
            a.out[k] = y;
        }
    }
}

// ---- the 2 : 1 case (44 100 -> 22 050 Hz: one phase, 66 taps) -- what almost every file needs -- with its own kernel: the 66
// coefficients are wave-uniform (scalar loads), a thread computes FOUR CONSECUTIVE outputs from 72 samples it reads once
// (18 x 16 bytes from LDS instead of 4 x 66 words: the windows of neighbouring outputs overlap in all but two samples), and a
// stereo file is read once for both channels.  Same operands, same order of the same fused multiply-adds per output.
constexpr int R2_TAPS = 66, R2_CENTER = 32, R2_SPAN = 2 * RS_TILE + 128;  // floats per channel: 2 (tile - 1) + taps, padded

template <typename SampleT, int CH>  // CH = 1 or 2
__global__ __launch_bounds__(RS_THREADS) void resample2_kernel(ResampleArgs a) {
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(16))) float span[CH][R2_SPAN];
    const SampleT* __restrict__ in = reinterpret_cast<const SampleT*>(a.in);
    const int t = threadIdx.x;
    const uint64_t k0 = (uint64_t)blockIdx.x * RS_TILE;
    if (k0 >= a.n_out) return;
    const uint64_t k1 = k0 + RS_TILE < a.n_out ? k0 + RS_TILE : a.n_out;  // exclusive
    const int64_t lo = 2 * (int64_t)k0 - R2_CENTER;                       // first tap of output k0
    const int span_len = 2 * (int)(k1 - 1 - k0) + R2_TAPS;
    for (int i = t; i < span_len; i += RS_THREADS) {
        int64_t s = lo + i;
        if (s < 0) s = -s;                                                     // x[-j] = x[j]
        else if ((uint64_t)s >= a.frames) s = 2 * (int64_t)a.frames - 1 - s;   // x[n + j] = x[n - 1 - j]
#pragma unroll
        for (int c = 0; c < CH; c++) span[c][i] = pcm_sample(in, (uint64_t)CH * (uint64_t)s + (uint64_t)c);
    }
    __syncthreads();
    float res[CH][4];
#pragma unroll
    for (int c = 0; c < CH; c++) {
        float x[72];  // samples first(k0 + 4 t) .. + 71: output j of this thread uses x[2 j .. 2 j + 65]
        const float4* src = reinterpret_cast<const float4*>(&span[c][8 * t]);
#pragma unroll
        for (int q = 0; q < 18; q++) {
            const float4 v = src[q];
            x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
        }
        float acc[4][8];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int u = 0; u < 8; u++) acc[j][u] = 0.0f;
#pragma unroll
        for (int i = 0; i < R2_TAPS; i++) {
            const float f = a.bank[i];  // wave-uniform: a scalar load
#pragma unroll
            for (int j = 0; j < 4; j++) acc[j][i & 7] = __builtin_fmaf(x[2 * j + i], f, acc[j][i & 7]);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float b0 = acc[j][0] + acc[j][4], b1 = acc[j][1] + acc[j][5], b2 = acc[j][2] + acc[j][6], b3 = acc[j][3] + acc[j][7];
            const float c0 = b0 + b2, c1 = b1 + b3;
            res[c][j] = c0 + c1;
        }
    }
    float y[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        y[j] = res[0][j];
        if (CH == 2) {
            const float m = 0.70710678118654752440f;  // (float)M_SQRT1_2, both matrix coefficients
            const float l = res[0][j] * m, r = res[CH - 1][j] * m;
            y[j] = l + r;
        }
    }
    const uint64_t k = k0 + 4 * (uint64_t)t;
    if (k + 3 < k1 && ((reinterpret_cast<uintptr_t>(a.out + k) & 15) == 0)) {
        *reinterpret_cast<float4*>(a.out + k) = make_float4(y[0], y[1], y[2], y[3]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (k + j < k1) a.out[k + j] = y[j];
    }
}

template <typename SampleT>
static void launch_resample2(const ResampleArgs& a, uint32_t grid, hipStream_t st) {
    if (a.channels == 1) hipLaunchKernelGGL((resample2_kernel<SampleT, 1>), dim3(grid), dim3(RS_THREADS), 0, st, a);
    else hipLaunchKernelGGL((resample2_kernel<SampleT, 2>), dim3(grid), dim3(RS_THREADS), 0, st, a);
}

hipError_t launch_resample(const void* in, int sample_format, uint32_t channels, uint64_t frames, const SwrPlan& p,
                           const float* d_bank, float* out, uint64_t n_out, hipStream_t st) {
    if (n_out == 0) return hipSuccess;
    if (p.phase_count == 1 && p.src_incr == 1 && p.dst_incr == 2 && p.taps == R2_TAPS && p.center == R2_CENTER && channels <= 2) {
        const uint64_t grid2 = (n_out + RS_TILE - 1) / RS_TILE;
        if (grid2 > 0x7fffffffull) return hipErrorInvalidValue;
        ResampleArgs a{in, out, d_bank, frames, n_out, p.dst_incr, p.src_incr, channels, (uint32_t)RS_TILE, p.taps, p.phase_count,
                       p.center, 0, 0};
        if (sample_format == BLISSGPU_SAMPLE_S16) launch_resample2<int16_t>(a, (uint32_t)grid2, st);
        else if (sample_format == BLISSGPU_SAMPLE_S32) launch_resample2<int32_t>(a, (uint32_t)grid2, st);
        else launch_resample2<float>(a, (uint32_t)grid2, st);
        return hipGetLastError();
    }
    // the tile's input span: (tile - 1) outputs further down the stream + one window, rounded up
    const uint64_t room = (uint64_t)(RS_SPAN_CAP - p.taps - 2);
    uint64_t tile = room * SWR_OUT_RATE / p.in_rate;
    if (tile > (uint64_t)RS_TILE) tile = RS_TILE;
    if (p.taps + 2 >= RS_SPAN_CAP || tile < 1) return hipErrorInvalidValue;  // in_rate beyond ~6 MHz
    const uint64_t grid = (n_out + tile - 1) / tile;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    // LDS: what the tile's span can reach ((tile - 1) outputs down the stream, rounded up, + one window), then the bank
    int span_cap = (int)((tile - 1) * p.in_rate / SWR_OUT_RATE) + p.taps + 2;
    span_cap = (span_cap + 63) & ~63;
    if (span_cap > RS_SPAN_CAP) span_cap = RS_SPAN_CAP;
    const size_t bank_floats = (size_t)p.phase_count * p.taps;
    const int bank_in_lds = (size_t)span_cap + bank_floats <= (size_t)RS_LDS_CAP ? 1 : 0;
    const size_t lds_bytes = sizeof(float) * ((size_t)span_cap + (bank_in_lds ? bank_floats : 0));
    ResampleArgs a{in, out, d_bank, frames, n_out, p.dst_incr, p.src_incr, channels, (uint32_t)tile, p.taps, p.phase_count,
                   p.center, span_cap, bank_in_lds};
    if (sample_format == BLISSGPU_SAMPLE_S16)
        hipLaunchKernelGGL(resample_kernel<int16_t>, dim3((uint32_t)grid), dim3(RS_THREADS), lds_bytes, st, a);
    else if (sample_format == BLISSGPU_SAMPLE_S32)
        hipLaunchKernelGGL(resample_kernel<int32_t>, dim3((uint32_t)grid), dim3(RS_THREADS), lds_bytes, st, a);
    else
        hipLaunchKernelGGL(resample_kernel<float>, dim3((uint32_t)grid), dim3(RS_THREADS), lds_bytes, st, a);
    return hipGetLastError();
}

// ---- synthetic white noise: uniform [-0.5, 0.5), Philox4x32-10, key = (0x5EED0000 + song, 0),
// counter = (sample_index / 4, 0, 0, 0); bit-identical to oracle/bliss_oracle.c bo_white_noise ----
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1, uint32_t r[4]) {
    uint32_t c[4] = {c0, c1, 0u, 0u};
#pragma unroll
    for (int round = 0; round < 10; round++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    r[0] = c[0]; r[1] = c[1]; r[2] = c[2]; r[3] = c[3];
}

__global__ __launch_bounds__(256) void synth_kernel(float* __restrict__ pcm, const SongDesc* __restrict__ songs,
                                                    uint32_t n_songs, const uint32_t* __restrict__ pfx_e,
                                                    uint32_t first_song_index, const uint32_t* __restrict__ song_index) {
    const uint32_t s = find_segment(pfx_e, n_songs, blockIdx.x);
    const SongDesc sd = songs[s];
    const uint32_t tile = blockIdx.x - pfx_e[s];
    float* __restrict__ x = pcm + sd.pcm_off;
    const uint32_t k0 = 0x5EED0000u + (song_index ? song_index[s] : first_song_index + s);
    // tile = 4096 samples = 1024 Philox blocks, 4 per thread
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint64_t blk = (uint64_t)tile * 1024 + i * 256 + threadIdx.x;
        const uint64_t idx = blk * 4;
        if (idx >= sd.n) continue;
        uint32_t r[4];
        philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), k0, 0u, r);
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (idx + e < sd.n) x[idx + e] = (float)(r[e] >> 8) * (1.0f / 16777216.0f) - 0.5f;
    }
}

void launch_synth(float* pcm, const SongDesc* songs, uint32_t n_songs, const uint32_t* pfx_e, uint32_t tiles_e,
                  uint32_t first_song_index, const uint32_t* song_index, hipStream_t st) {
    if (tiles_e == 0) return;
    hipLaunchKernelGGL(synth_kernel, dim3(tiles_e), dim3(256), 0, st, pcm, songs, n_songs, pfx_e, first_song_index, song_index);
}

}  // namespace bg
