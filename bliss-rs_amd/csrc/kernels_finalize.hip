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

// The chains below are sequential by definition (they reproduce the reference's rounding order).  A wavefront
// executing ONE such chain still pays a full 4-cycle VALU issue per instruction, so the mapping is one LANE per
// song: 64 songs advance their (independent) chains in lock step and a role costs 1/64 of the issue slots.
//
// Memory side: every lane streams its own series with 16-byte loads, 32 elements (eight loads) per step, and the NEXT
// 32 elements are requested before the current ones are consumed: the ~100 cycles of dependent arithmetic per element
// then cover the memory latency, and the kernel runs at the speed of its longest chain.  No LDS: the kernel must be
// able to run beside the FFT-8192 kernel, whose four workgroups per CU leave 2 KB of LDS free.
constexpr int SEQ_CHUNK = 32;

template <typename T, typename Step>
__device__ __forceinline__ void seq_for_each(const T* __restrict__ x, uint32_t n, Step&& step) {
    typedef T v4 __attribute__((ext_vector_type(4)));
    uint32_t i = 0;
    while (i < n && ((reinterpret_cast<uintptr_t>(x + i) & 15) != 0)) { step(x[i], i); i++; }
    const uint32_t n_chunks = (n - i) / SEQ_CHUNK;
    v4 cur[SEQ_CHUNK / 4], nxt[SEQ_CHUNK / 4];
    if (n_chunks) {
#pragma unroll
        for (int u = 0; u < SEQ_CHUNK / 4; u++) cur[u] = *reinterpret_cast<const v4*>(x + i + 4 * u);
    }
    for (uint32_t c = 0; c < n_chunks; c++) {
        if (c + 1 < n_chunks) {
#pragma unroll
            for (int u = 0; u < SEQ_CHUNK / 4; u++) nxt[u] = *reinterpret_cast<const v4*>(x + i + SEQ_CHUNK + 4 * u);
        }
#pragma unroll
        for (int u = 0; u < SEQ_CHUNK / 4; u++) {
            step(cur[u].x, i + 4 * u);
            step(cur[u].y, i + 4 * u + 1);
            step(cur[u].z, i + 4 * u + 2);
            step(cur[u].w, i + 4 * u + 3);
        }
#pragma unroll
        for (int u = 0; u < SEQ_CHUNK / 4; u++) cur[u] = nxt[u];
        i += SEQ_CHUNK;
    }
    for (; i < n; i++) step(x[i], i);
}

__device__ __forceinline__ void welford_step(float v, uint32_t i, float& mean, float& sum_sq) {
    const float count = (float)(i + 1);
    const float delta = v - mean;
    mean = mean + delta / count;
    sum_sq = __fmaf_rn(v - mean, delta, sum_sq);
}

enum Role { R_CENT = 0, R_ROLL, R_FLAT, R_LOUD, R_ZCR, R_COUNT };

// summary[s][0..15]: slots 1..9 = features 1..9 (zcr, centroid, rolloff, flatness, loudness)
// grid = (ceil(n_songs / 64), R_COUNT): one wavefront = one role of 64 songs (lane = song).  A timbral role carries the
// mean chain (one add per element) and the Welford chain of the same series: the series is streamed once.
__global__ __launch_bounds__(64) void summary_kernel(const SongDesc* __restrict__ songs, uint32_t n_songs,
                                                     const float* __restrict__ centroid,
                                                     const float* __restrict__ rolloff,
                                                     const float* __restrict__ flatness,
                                                     const float* __restrict__ e256,
                                                     const uint32_t* __restrict__ zc256,
                                                     float* __restrict__ summary) {
    const uint32_t s = blockIdx.x * 64 + threadIdx.x;
    // lanes past the batch (or too-short songs) idle with n = 0
    SongDesc sd{};
    if (s < n_songs) sd = songs[s];
    const bool live = s < n_songs && sd.ok;
    float* feat = summary + (size_t)(live ? s : 0) * 16;
    const float half_sr = (float)SAMPLE_RATE / 2.0f;
    const int role = (int)blockIdx.y;  // wave-uniform
    if (role <= R_FLAT) {
        const float* series = role == R_CENT ? centroid : (role == R_ROLL ? rolloff : flatness);
        const uint32_t n = live ? sd.n_t : 0u;
        float sum = 0.0f, mean = 0.0f, sum_sq = 0.0f;
        seq_for_each(series + (live ? sd.t_off : 0), n, [&](float v, uint32_t i) {
            sum += v;                            // utils::mean: sequential f32 sum
            welford_step(v, i, mean, sum_sq);    // ndarray std_axis
        });
        if (live) {
            const float mean_value = sum / (float)n;
            const float std_value = sqrtf(sum_sq / ((float)n - 0.0f));
            if (role == R_FLAT) {
                feat[6] = 2.0f * (mean_value - 0.0f) / (1.0f - 0.0f) - 1.0f;
                feat[7] = 2.0f * (std_value - 0.0f) / (1.0f - 0.0f) - 1.0f;
            } else {
                feat[role == R_CENT ? 2 : 4] = normalize(mean_value, 0.0f, half_sr);
                feat[role == R_CENT ? 3 : 5] = normalize(std_value, 0.0f, half_sr);
            }
        }
    } else if (role == R_LOUD) {
        // chunk level = sum of squares over <= 1024 samples / len (src/misc.rs:12-18); the four 256-sample partial
        // sums of a chunk are added in order as they stream by
        const uint32_t n = live ? sd.n_e : 0u;
        float msum = 0.0f, mean = 0.0f, sum_sq = 0.0f, en = 0.0f;
        seq_for_each(e256 + (live ? sd.e_off : 0), n, [&](float v, uint32_t q) {
            en += v;
            if ((q & 3u) == 3u || q + 1 == n) {  // last block of chunk c = q / 4
                const uint32_t c = q >> 2;
                const uint64_t len = ((uint64_t)(c + 1) * LOUD_W <= sd.n) ? LOUD_W : sd.n - (uint64_t)c * LOUD_W;
                const float lv = en / (float)len;
                msum += lv;
                welford_step(lv, c, mean, sum_sq);
                en = 0.0f;
            }
        });
        if (live) {
            float mean_value = msum / (float)sd.n_l;
            float std_value = sqrtf(sum_sq / ((float)sd.n_l - 0.0f));
            if (mean_value < 1e-9f) mean_value = 1e-9f;
            if (std_value < 1e-9f) std_value = 1e-9f;
            feat[8] = normalize(10.0f * log10f(mean_value), -90.0f, 0.0f);
            feat[9] = normalize(10.0f * log10f(std_value), -90.0f, 0.0f);
        }
    } else {
        const uint32_t n = live ? sd.n_e : 0u;
        uint32_t c = 0;
        seq_for_each(zc256 + (live ? sd.e_off : 0), n, [&](uint32_t v, uint32_t) { c += v; });
        if (live) feat[1] = normalize((float)c / (float)sd.n, 0.0f, 1.0f);
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
                                                      uint32_t* __restrict__ dbg_nbpms,
                                                      double* __restrict__ dbg_interval) {
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
        if (dbg_interval) dbg_interval[(size_t)s * 10 + t] = raw[t];  // tap of the parity tests
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

// ------------------------------------------------------------------------------------------------
// spectral_rolloff (src/aubio.rs:36-58) as the reference writes it -- the 256 squares added one by one in f32, twice --
// for the frames whose bin count the FFT-512 kernel could not prove from its own summation order (a running energy within
// worst-case rounding of the 95 % threshold).  Thread per frame; this translation unit never fuses a * b + c.
// ------------------------------------------------------------------------------------------------
#ifndef RF_STRETCH_FRAMES
#define RF_STRETCH_FRAMES 1024
#endif
constexpr uint32_t RF_STRETCH = RF_STRETCH_FRAMES;  // frames of the rolloff series a wavefront looks through

__global__ __launch_bounds__(64) void rolloff_fix_kernel(const float* __restrict__ mags, float* __restrict__ rolloff, uint32_t n) {
    // One wavefront per RF_STRETCH frames of the chunk's rolloff series.  It first collects the frames the FFT-512 kernel
    // left as ROLLOFF_UNPROVEN (a few per cent; their 256 magnitudes wait at the entry of their own index), then takes 64
    // of them at a time, a lane per frame.  A frame's magnitudes are fetched with coalesced 16-byte loads (four frames x
    // 256 B per instruction), 64 bins of all 64 frames at a time, and turned through a padded LDS tile (rows of 17 float4:
    // 16-byte writes and reads, no bank conflicts) so that every lane holds the 64 bins of ITS frame in registers and walks
    // them there; the loads of the next piece are in flight during the walk.  (Round 6: the walk used to read LDS one float
    // at a time, each read waited for -- 320 dependent LDS round trips per frame were most of the kernel's time.  Keeping
    // all 256 bins of the 64 frames in LDS -- one fetch instead of two -- leaves two wavefronts per CU and takes twice as
    // long; 16 whole frames per pass, read once and contiguously, with 16 walking lanes is slower as well.)
    __shared__ float4 tile[64 * 17];
    __shared__ uint32_t list[RF_STRETCH];
    const float freq_per_bin = (float)SAMPLE_RATE / (float)W512;
    const uint32_t lane = threadIdx.x;
    const uint32_t first = blockIdx.x * RF_STRETCH;
    uint32_t count = 0;
    {
        float v[RF_STRETCH / 64];  // every load of the stretch in flight together
#pragma unroll
        for (uint32_t i = 0; i < RF_STRETCH / 64; i++) {
            const uint32_t t = first + 64 * i + lane;
            v[i] = t < n ? rolloff[t] : 0.0f;
        }
#pragma unroll
        for (uint32_t i = 0; i < RF_STRETCH / 64; i++) {
            const bool hit = v[i] == ROLLOFF_UNPROVEN;
            const uint64_t hits = __ballot(hit);
            if (hit) list[count + __popcll(hits & ((1ull << lane) - 1))] = first + 64 * i + lane;
            count += (uint32_t)__popcll(hits);
        }
    }
    __syncthreads();  // the list is complete
#ifdef RF_SCAN_ONLY  // (timing probe: what the search alone costs)
    if (count < 4096) return;
#endif
    for (uint32_t base = 0; base < count; base += 64) {
        // piece p of the chunk's 64 frames: 16 loads in flight together (the wavefront pays the memory latency once per piece)
        auto fetch = [&](int p, float4 (&v)[16]) {
#pragma unroll
            for (uint32_t q = 0; q < 16; q++) {
                const uint32_t e = base + 4 * q + (lane >> 4), f4 = lane & 15;
                v[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (e < count) v[q] = *reinterpret_cast<const float4*>(mags + (size_t)list[e] * 256 + 64 * p + 4 * f4);
            }
        };
        // through the tile: lane l ends up with bins 64 p .. 64 p + 63 of frame base + l
        auto turn = [&](const float4 (&v)[16], float4 (&r)[16]) {
            __syncthreads();  // the tile's last readers are done
#pragma unroll
            for (uint32_t q = 0; q < 16; q++) tile[(4 * q + (lane >> 4)) * 17 + (lane & 15)] = v[q];
            __syncthreads();
#pragma unroll
            for (uint32_t q = 0; q < 16; q++) r[q] = tile[lane * 17 + q];
        };
        float4 v[16], r[16];
        float cumsum = 0.0f, upto[3] = {0.0f, 0.0f, 0.0f};  // upto[p]: the running sum after piece p -- what `rollsum` will be there as well
#pragma unroll 1
        for (int p = 0; p < 4; p++) {
            fetch(p, v);
            turn(v, r);
#pragma unroll
            for (int q = 0; q < 16; q++) {
                cumsum += r[q].x * r[q].x;
                cumsum += r[q].y * r[q].y;
                cumsum += r[q].z * r[q].z;
                cumsum += r[q].w * r[q].w;
            }
            upto[0] = p == 0 ? cumsum : upto[0];
            upto[1] = p == 1 ? cumsum : upto[1];
            upto[2] = p == 2 ? cumsum : upto[2];
        }
        // `while rollsum < cumsum && j < len { rollsum += sq[j]; j += 1 }` walks the same partial sums again: it passes
        // every piece whose last partial sum is still below the threshold and stops inside the first other one, so only
        // that piece has to be walked (white noise crosses in the last piece in every lane: its bins are still in `r`)
        const float thr = cumsum * 0.95f;
        int pc = 0;
        float rollsum = 0.0f;
#pragma unroll
        for (int p = 0; p < 3; p++)
            if (upto[p] < thr && pc == p) { pc = p + 1; rollsum = upto[p]; }
        int j = 64 * pc;
        for (int p = 3; p >= 0; p--) {
            if (!__any(pc == p && rollsum < thr)) continue;
            if (p != 3) { fetch(p, v); turn(v, r); }
            if (pc == p) {
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const float sq[4] = {r[q].x * r[q].x, r[q].y * r[q].y, r[q].z * r[q].z, r[q].w * r[q].w};
#pragma unroll
                    for (int z = 0; z < 4; z++) {
                        const bool go = rollsum < thr;  // monotone: once false it stays false
                        rollsum = go ? rollsum + sq[z] : rollsum;
                        j += go ? 1 : 0;
                    }
                }
            }
        }
        const float bins = cumsum != 0.0f ? (float)j : 0.0f;
        if (base + lane < count) rolloff[list[base + lane]] = freq_per_bin * fmaxf(bins, 0.0f);
    }
}

void launch_rolloff_fix(const Batch& b, const Workspace& w, uint64_t total_t, hipStream_t st) {
    if (b.tiles_f == 0 || total_t == 0) return;
    const uint32_t blocks = (uint32_t)((total_t + RF_STRETCH - 1) / RF_STRETCH);
    hipLaunchKernelGGL(rolloff_fix_kernel, dim3(blocks), dim3(64), 0, st, w.spec, w.rolloff, (uint32_t)total_t);  // (RollFix::mags = w.spec)
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
                       w.summary, w.chroma_part, w.tempo, w.tuning, features_version, d_out, d_status, dbg_tuning, dbg_nbpms,
                       w.dbg_interval);
}

}  // namespace bg
