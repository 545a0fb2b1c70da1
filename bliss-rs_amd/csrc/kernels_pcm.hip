// kernels_pcm.hip -- PCM in and out of the analysis format:
//   * raw decoder output (s16 / f32, interleaved channels) -> the mono f32 PCM Song::analyze takes
//   * Philox4x32-10 white-noise synthesis for the benchmark (no reference counterpart).
// (The per-block PCM statistics -- sums of squares, zero crossings -- are computed by the FFT-512 kernel, which
// already holds every sample in registers: kernels_fft512.hip.)
#include "device_utils.hpp"
#include "internal.hpp"

namespace bg {

// ---- raw decoder output -> the mono f32 PCM Song::analyze takes (the PCM feed) ----
//   s16 -> f32   : sample / 32768 (exact in f32), FFmpeg's AV_SAMPLE_FMT_S16 -> FLT conversion as used by the reference's
//                  decoder (src/song/decoder/ffmpeg.rs:36-109); halves the PCIe bytes of the feed
//   stereo -> mono: (L + R) * SQRT_2 / 2 in f32, in exactly this order (src/song/decoder/symphonia.rs:281-285; "recovers
//                  the exact behavior of ffmpeg": both decoders pin the result on data/s16_stereo_22_5kHz.flac to
//                  Adler-32 0x1d7b2d6d, ffmpeg.rs:448-452, symphonia.rs:594-610)
//   > 2 channels : sequential f32 sum / channel count (symphonia.rs:291-297)
__device__ __forceinline__ float pcm_sample(const int16_t* p, uint64_t i) { return (float)p[i] * (1.0f / 32768.0f); }
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

void launch_pcm_convert(const void* in, int bytes_per_sample, uint32_t channels, float* out, uint64_t frames, hipStream_t st) {
    if (frames == 0) return;
    const dim3 grid((uint32_t)((frames + 1023) / 1024));
    if (bytes_per_sample == 2)
        hipLaunchKernelGGL(pcm_convert_kernel<int16_t>, grid, dim3(256), 0, st, (const int16_t*)in, channels, out, frames);
    else
        hipLaunchKernelGGL(pcm_convert_kernel<float>, grid, dim3(256), 0, st, (const float*)in, channels, out, frames);
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
