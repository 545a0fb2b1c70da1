// kernels_pcm.hip -- one coalesced pass over the batch PCM:
//   * per 256-sample block: sum of squares (feeds LoudnessDesc, src/misc.rs:12-18,46-65, and the
//     tempo silence test, src/aubio.rs:1258-1276) and zero-crossing count (number_crossings,
//     src/utils.rs:81-95: a crossing is a change of `x > 0` between consecutive samples)
//   * Philox4x32-10 white-noise synthesis for the benchmark (no reference counterpart).
// HBM-bound: 4 bytes read per sample, 8 bytes written per 256 samples.
#include "device_utils.hpp"
#include "internal.hpp"

namespace bg {

constexpr int PCM_TILE_BLOCKS = 16;  // 256-sample blocks per workgroup (4 per wave)

__global__ __launch_bounds__(256) void pcm_stats_kernel(const float* __restrict__ pcm,
                                                        const SongDesc* __restrict__ songs, uint32_t n_songs,
                                                        const uint32_t* __restrict__ pfx_e, float* __restrict__ e256,
                                                        uint32_t* __restrict__ zc256) {
    const uint32_t s = find_segment(pfx_e, n_songs, blockIdx.x);
    const SongDesc sd = songs[s];
    const uint32_t tile = blockIdx.x - pfx_e[s];
    const float* __restrict__ x = pcm + sd.pcm_off;
    const int lane = lane_id(), wave = wave_id();
    const bool aligned16 = ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    {
        // fast path: the wave's four blocks are complete and 16-byte aligned -> all four loads in flight at once
        const uint32_t q0 = tile * PCM_TILE_BLOCKS + wave * (PCM_TILE_BLOCKS / 4);
        if (aligned16 && (uint64_t)(q0 + PCM_TILE_BLOCKS / 4) * 256 <= sd.n) {  // wave-uniform
            const uint64_t base0 = (uint64_t)q0 * 256;
            float4 v[PCM_TILE_BLOCKS / 4];
#pragma unroll
            for (int i = 0; i < PCM_TILE_BLOCKS / 4; i++) {  // streamed once: non-temporal, the FFT kernels own the L2
                typedef float f32x4_t __attribute__((ext_vector_type(4)));
                const f32x4_t q = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(x + base0 + 256 * i + 4 * lane));
                v[i] = make_float4(q.x, q.y, q.z, q.w);
            }
            uint32_t carry = (x[base0 > 0 ? base0 - 1 : 0] > 0.0f) ? 1u : 0u;  // positivity of the sample before the block
            float ss[PCM_TILE_BLOCKS / 4];
            uint32_t zc[PCM_TILE_BLOCKS / 4];
#pragma unroll
            for (int i = 0; i < PCM_TILE_BLOCKS / 4; i++) {
                const float4 w = v[i];
                ss[i] = (w.x * w.x + w.y * w.y) + (w.z * w.z + w.w * w.w);
                const uint32_t p0 = w.x > 0.0f, p1 = w.y > 0.0f, p2 = w.z > 0.0f, p3 = w.w > 0.0f;
                uint32_t before = __shfl_up(p3, 1, WAVE);
                if (lane == 0) before = carry;
                zc[i] = (p0 != before) + (p1 != p0) + (p2 != p1) + (p3 != p2);
                carry = __shfl(p3, 63, WAVE);
            }
#pragma unroll
            for (int i = 0; i < PCM_TILE_BLOCKS / 4; i++) {
                const float s_tot = wave_sum(ss[i]);
                const uint32_t z_tot = wave_sum(zc[i]);
                if (lane == 0) {
                    e256[sd.e_off + q0 + i] = s_tot;
                    zc256[sd.e_off + q0 + i] = z_tot;
                }
            }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < PCM_TILE_BLOCKS / 4; i++) {
        const uint32_t q = tile * PCM_TILE_BLOCKS + wave * (PCM_TILE_BLOCKS / 4) + i;
        if (q >= sd.n_e) break;  // wave-uniform
        const uint64_t base = (uint64_t)q * 256;
        // positivity of the sample before the block (the first sample of the song compares with itself)
        uint32_t prev_pos = (x[base > 0 ? base - 1 : 0] > 0.0f) ? 1u : 0u;
        float ss = 0.0f;
        uint32_t zc = 0;
        if (aligned16 && base + 256 <= sd.n) {  // wave-uniform: one 16-byte load per lane covers the block
            const float4 v = *reinterpret_cast<const float4*>(x + base + 4 * lane);
            ss = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            const uint32_t p0 = v.x > 0.0f, p1 = v.y > 0.0f, p2 = v.z > 0.0f, p3 = v.w > 0.0f;
            uint32_t before = __shfl_up(p3, 1, WAVE);
            if (lane == 0) before = prev_pos;
            zc = (p0 != before) + (p1 != p0) + (p2 != p1) + (p3 != p2);
            ss = wave_sum(ss);
            zc = wave_sum(zc);
            if (lane == 0) {
                e256[sd.e_off + q] = ss;
                zc256[sd.e_off + q] = zc;
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint64_t idx = base + j * 64 + lane;
            const bool valid = idx < sd.n;
            const float v = valid ? x[idx] : 0.0f;
            ss += v * v;
            const uint64_t pos = __ballot(valid && v > 0.0f);
            const uint64_t vmask = __ballot(valid);
            const uint64_t shifted = (pos << 1) | (uint64_t)prev_pos;
            zc += (uint32_t)__popcll((pos ^ shifted) & vmask);
            prev_pos = (uint32_t)(pos >> 63);
        }
        ss = wave_sum(ss);
        if (lane == 0) {
            e256[sd.e_off + q] = ss;
            zc256[sd.e_off + q] = zc;
        }
    }
}

void launch_pcm_stats(const Batch& b, const Workspace& w, hipStream_t st) {
    if (b.tiles_e == 0) return;
    hipLaunchKernelGGL(pcm_stats_kernel, dim3(b.tiles_e), dim3(256), 0, st, b.pcm, b.songs, b.n_songs, b.pfx_e,
                       w.e256, w.zc256);
}

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
