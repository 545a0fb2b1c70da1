// kernels_chroma.hip -- the chroma descriptor (src/chroma.rs) on the device.
//
//   chroma_bank_kernel : chroma_filter(22050, 8192, 12, tuning) (src/chroma.rs:197-267) for the 100 tunings
//                        pitch_tuning can return (+ tuning 0.0), built once per context (f64, 40 MB).
//   stft8192_kernel    : utils::stft(signal, 8192, 2205) (src/utils.rs:26-64): reflect pad, periodic Hann (f32), FFT,
//                        |X|.  A 256-thread workgroup owns 16 frames of a song -- every FOURTH frame of a 64-frame
//                        super-tile whose four workgroups share an XCD's L2, so the 73 % overlap of consecutive frames is
//                        fetched from HBM once.  A frame's 8192 reals are 4096 complex values = 16 x 16 x 16: three
//                        register radix-16 passes (fft_r16.hpp), two padded LDS transposes, a third LDS trip for the
//                        mirrored real-input split.  The f32 magnitudes (exactly what the reference widens to f64) go to
//                        HBM as 16-byte stores; while the row is still in LDS the kernel runs pip_track's peak test
//                        (src/chroma.rs:269-331), classifies every peak (coarse magnitude bin, pitch-residue bin) into
//                        one 32-bit record and counts the peaks by coarse magnitude bin.
//   tune_*_kernel      : estimate_tuning (src/chroma.rs:361-391): the histogram locates the coarse bins of the two middle
//                        order statistics, pass 2 walks the peak records (f64 only for the ~3 % that need it), the final
//                        kernel radix-selects the exact Midpoint median among the candidates and takes the first
//                        argmax of the 100-bin pitch histogram (pitch_tuning, :334-359).
//   chroma_kernel      : chroma_stft (:393-412) as an f64 MFMA contraction filter(12 x 4097) x S^2, followed per frame by
//                        the L1 normalisation, exp(15x), normalisation and the 10 interval templates x 12 rotations
//                        (:137-188), summed per 64-frame tile.
#include <float.h>
#include <stdlib.h>

#include <type_traits>

#include "device_utils.hpp"
#include "fft_r16.hpp"
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
    if (k >= BANK_PITCH) return;
    double* out = bank + (size_t)slot * BANK_ROWS * BANK_PITCH;
    if (k >= CBINS) {
        for (int r = 0; r < BANK_ROWS; r++) out[(size_t)r * BANK_PITCH + k] = 0.0;
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
        out[(size_t)r * BANK_PITCH + k] = v;
    }
}

void launch_chroma_bank(double* bank, hipStream_t st) {
    hipLaunchKernelGGL(chroma_bank_kernel, dim3((BANK_PITCH + 255) / 256, N_TUNING + 1), dim3(256), 0, st, bank);
}

// ------------------------------------------------------------------------------------------------
// pip_track peak test on three neighbouring magnitudes (src/chroma.rs:317-327) + the pitch-residue
// bin pitch_tuning would file it under (:342-351).  f64, no contraction: identical in both passes.
// ------------------------------------------------------------------------------------------------
// the peak condition alone (:318): cheap, evaluated by every lane; f32 -> f64 widening is monotone,
// so the two neighbour comparisons give the same outcome in f32
__device__ __forceinline__ bool pip_is_peak(float sb, float se, float sa, double ref) {
    return (sa <= se) && (sb < se) && ((double)se > ref);
}

// append the centre bins flagged by `is_peak` to an LDS list (order is irrelevant downstream)
__device__ __forceinline__ void wave_append(bool is_peak, int c, uint16_t* list, uint32_t* count) {
    const uint64_t mask = __ballot(is_peak);
    if (mask == 0) return;
    const uint32_t lane = (uint32_t)lane_id();
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(count, (uint32_t)__popcll(mask));
    base = __shfl(base, 0, WAVE);
    if (is_peak) list[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)c;
}

__device__ __forceinline__ bool pip_peak_core(float sb, float se, float sa, double ref, int c, double* mag_out,
                                              double* pitch_out) {
#pragma clang fp contract(off)
    if (!(sa <= se && sb < se)) return false;
    const double before = (double)sb, elem = (double)se, after = (double)sa;
    if (!(elem > ref)) return false;
    const double avg = 0.5 * (after - before);
    double shift = 2.0 * elem - after - before;
    if (fabs(shift) < DBL_MIN) shift += 1.0;
    shift = avg / shift;
    const double pitch = ((double)c + shift) * 22050.0 / 8192.0;
    if (!(pitch > 0.0)) return false;  // estimate_tuning keeps p > 0 only (:370-375)
    *mag_out = elem + 0.5 * avg * shift;
    *pitch_out = pitch;
    return true;
}

// magnitude only (pass 1)
__device__ __forceinline__ bool pip_peak_mag(float sb, float se, float sa, double ref, int c, double* mag_out) {
    double pitch;
    return pip_peak_core(sb, se, sa, ref, c, mag_out, &pitch);
}

// pitch_tuning's residue bin of a peak (:342-351)
__device__ __forceinline__ int pitch_bin(double pitch) {
#pragma clang fp contract(off)
    double x = log2(pitch / (440.0 / 16.0));
    x = 12.0 * x;
    x = x - trunc(x);  // == fmod(x, 1.0) exactly (the subtraction of the integer part is exact in binary fp)
    if (x >= 0.5) x -= 1.0;
    const double q = (x - -0.5) / 0.01;
    int idx = (q > 0.0) ? (int)q : 0;
    if (idx > N_TUNING - 1) idx = N_TUNING - 1;
    return idx;
}

__device__ __forceinline__ uint32_t coarse_bin(double mag) {
    const uint32_t b = __float_as_uint((float)mag) >> COARSE_SHIFT;  // monotone in mag for mag > 0
    return b < (uint32_t)H1_BINS ? b : (uint32_t)H1_BINS - 1;
}

// `(double)se > ref` as an f32 comparison: thr = the largest float <= ref (then se > thr <=> (double)se > ref)
__device__ __forceinline__ float ref_floor_f32(double ref) {
    float f = (float)ref;
    if ((double)f > ref) f = __uint_as_float(__float_as_uint(f) - 1u);  // ref >= 0, so f > 0 here
    return f;
}

// The f64 interpolation above is only needed bit-exactly near a decision boundary.  For an established peak
// (sa <= se, sb < se) the f32 evaluation below is within 2 ulp(se) of the f64 magnitude: |avg| <= den / 2
// bounds the interpolation term by se / 4 and its error by ~0.5 ulp(se), den > 0 is never rounded to zero
// (2 se - sa >= se > sb), plus two final roundings.  A 16-ulp guard band around the coarse-bin edges
// (2^COARSE_SHIFT ulp apart) therefore makes the f32 bin provably equal to coarse_bin() of the f64 magnitude;
// the ~1e-4 of peaks inside the band take the f64 path.
__device__ __forceinline__ uint32_t peak_coarse_bin(float sb, float se, float sa, double ref, int c) {
    const float avg = 0.5f * (sa - sb);
    const float den = (2.0f * se - sa) - sb;
    const float shift = avg * __builtin_amdgcn_rcpf(den);
    const uint32_t bits = __float_as_uint(se + (0.5f * avg) * shift), low = bits & COARSE_LOW_MASK;
    if (se >= 1e-30f && low >= 16u && low <= COARSE_LOW_MASK - 16u) {
        const uint32_t b = bits >> COARSE_SHIFT;
        return b < (uint32_t)H1_BINS ? b : (uint32_t)H1_BINS - 1;
    }
    double mag;
    pip_peak_mag(sb, se, sa, ref, c, &mag);  // always true for c >= PIP_LO (pitch > 0)
    return coarse_bin(mag);
}

// pitch-residue bin in f32, or -1 when it is not provably the f64 pitch_bin(): the peak must not be flat
// (den >= 2^-8 se bounds the shift error by 2.8e-5 FFT bins, i.e. 8.5e-4 tuning bins at c >= 57) and the bin
// coordinate must be at least PITCH_GUARD away from an integer (v_log_f32 at 1 ulp of a result below 16, the constant
// add, the x12 and the rounding of c + shift stay below 1.9e-3 bins: 2.8e-3 in all).  Checked against the f64 path on
// 2e9 random peaks from sharp to flat (tests/tools/probes/guard_probe.hip): no disagreement down to a guard of 0.002,
// the first ones at 0.001.  The records whose bin is not provable take tune_pass2_kernel's f64 path -- 1.6 % of the
// peaks with this guard; with 0.02 / 2^-10 it was 4 % and that path is two thirds of that kernel's time.
constexpr float PITCH_GUARD = 0.008f, PITCH_FLAT_LIMIT = 0.00390625f;
__device__ __forceinline__ int peak_pitch_bin_f32(float sb, float se, float sa, int c) {
    const float avg = 0.5f * (sa - sb);
    const float den = (2.0f * se - sa) - sb;
    if (!(se >= 1e-30f) || den < se * PITCH_FLAT_LIMIT) return -1;
    const float shift = avg * __builtin_amdgcn_rcpf(den);
    // 12 log2(pitch / 27.5) with pitch = (c + shift) * 22050 / 8192
    float x = 12.0f * (__builtin_amdgcn_logf((float)c + shift) + -3.3528687f);  // log2(22050 / 8192 / 27.5)
    x = x - truncf(x);
    if (x >= 0.5f) x -= 1.0f;
    const float q = (x + 0.5f) * 100.0f;
    const float fl = floorf(q), fr = q - fl;
    if (fr < PITCH_GUARD || fr > 1.0f - PITCH_GUARD) return -1;
    const int idx = (int)fl;
    return idx < 0 ? 0 : (idx > N_TUNING - 1 ? N_TUNING - 1 : idx);
}

// rare exact path of the classifiers, kept out of line so that it does not inflate the register pressure of the hot loop
__device__ __attribute__((noinline)) uint32_t coarse_bin_exact(float sb, float se, float sa, double ref, int c) {
    double mag;
    pip_peak_mag(sb, se, sa, ref, c, &mag);  // always true for c >= PIP_LO (pitch > 0)
    return coarse_bin(mag);
}

// coarse magnitude bin AND pitch-residue bin of one established peak from a single evaluation of the parabolic
// shift (same arithmetic and guard bands as peak_coarse_bin / peak_pitch_bin_f32, which it must agree with bit
// for bit: tuning pass 2 and the histogram rely on it)
__device__ __forceinline__ uint32_t peak_classify(float sb, float se, float sa, double ref, int c, int* pitch_bin_out) {
    const float avg = 0.5f * (sa - sb);
    const float den = (2.0f * se - sa) - sb;
    const float shift = avg * __builtin_amdgcn_rcpf(den);
    const bool normal = se >= 1e-30f;
    // pitch bin
    int pb = -1;
    if (normal && !(den < se * PITCH_FLAT_LIMIT)) {
        float x = 12.0f * (__builtin_amdgcn_logf((float)c + shift) + -3.3528687f);
        x = x - truncf(x);
        if (x >= 0.5f) x -= 1.0f;
        const float q = (x + 0.5f) * 100.0f;
        const float fl = floorf(q), fr = q - fl;
        if (!(fr < PITCH_GUARD || fr > 1.0f - PITCH_GUARD)) {
            const int idx = (int)fl;
            pb = idx < 0 ? 0 : (idx > N_TUNING - 1 ? N_TUNING - 1 : idx);
        }
    }
    *pitch_bin_out = pb;
    // coarse bin
    const uint32_t bits = __float_as_uint(se + (0.5f * avg) * shift), low = bits & COARSE_LOW_MASK;
    if (normal && low >= 16u && low <= COARSE_LOW_MASK - 16u) {
        const uint32_t b = bits >> COARSE_SHIFT;
        return b < (uint32_t)H1_BINS ? b : (uint32_t)H1_BINS - 1;
    }
    return coarse_bin_exact(sb, se, sa, ref, c);
}

// One 32-bit record per peak, written by the STFT kernel while the frame is still in LDS and consumed by tuning
// pass 2 (which then never re-scans the spectrogram): exact coarse magnitude bin (14 bits) | pitch-residue bin + 1
// (7 bits, 0 = the f32 evaluation was not provable, take the f64 path) | centre bin (11 bits).
__device__ __forceinline__ uint32_t peak_record(uint32_t coarse, int pitch_bin_or_neg, int c) {
    return (coarse << 18) | ((uint32_t)(pitch_bin_or_neg + 1) << 11) | (uint32_t)c;
}

// ------------------------------------------------------------------------------------------------
// STFT 8192 / hop 2205
//
// One workgroup (256 threads) transforms STFT_FRAMES_PER_WG consecutive frames of one song.  A frame's
// 8192 real samples are packed as 4096 complex values z[n] = x[2n] + i x[2n+1] and transformed as
// 16 x 16 x 16: every thread keeps 16 complex values in registers and runs three radix-16 passes with
// two padded (bank-conflict-free) LDS transposes in between; a third LDS round trip pairs Z[k] with
// Z[4096-k] for the real-input split.  The window (read once per workgroup) stays in registers.
// ------------------------------------------------------------------------------------------------
constexpr int STFT_FRAMES_PER_WG = STFT_TILE;
constexpr int STFT_GROUP = 4;     // workgroups that share a super-tile of STFT_GROUP * STFT_TILE frames, one frame in four each
constexpr int LHIST_BINS = 512;  // 8 octaves of 64 coarse bins
constexpr int EX1_PITCH = 257;   // k1-major rows of 256 (+1): the 16 lanes of a ds_read2_b64 group tile all 32 banks
constexpr int EX2_PITCH = 272;   // j1-major rows of 256 (+16): shifts odd rows by 32 banks
constexpr int STFT_LDS = 16 * EX2_PITCH;  // float2 elements (34 816 B)
constexpr int MAGS_TOP = 2 * 4096;        // word index (of the exchange buffer) where magnitude words 4096..4127 live

// W_32^j = (cos, -sin)(2 pi j / 32), j < 8
__device__ constexpr float CONST_COS32[8] = {1.0f, 0.98078528040323044f, 0.92387953251128674f, 0.83146961230254524f,
                                             0.70710678118654752f, 0.55557023301960222f, 0.38268343236508977f, 0.19509032201612827f};
__device__ constexpr float CONST_SIN32[8] = {0.0f, 0.19509032201612827f, 0.38268343236508977f, 0.55557023301960222f,
                                             0.70710678118654752f, 0.83146961230254524f, 0.92387953251128674f, 0.98078528040323044f};

typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
// (non-temporal LOADS of the spectrogram were tried in pass 2 and the chroma contraction: 8-12 % slower)
// 8-byte load through a buffer descriptor: 32-bit lane offset + scalar offset (no 64-bit address VGPRs)
__device__ __forceinline__ f2 buf_load_f2(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return mk(__uint_as_float(v.x), __uint_as_float(v.y));
}

__device__ __forceinline__ long reflect_index(long p, long n) {
    // numpy mode="reflect" on the original signal (src/utils.rs:11-24); pad = 4096 <= n - 2 always holds
    if (p < 0) p = -p;
    else if (p >= n) p = 2 * n - 2 - p;
    return p;
}

// Hooks of the per-phase cycle trace (tests/tools/probes/stft_trace/stft_trace.hip defines them and then includes this
// file; in the library they are empty)
#ifndef TRACE
#define TRACE_DECL
#define TRACE_INIT() do {} while (0)
#define TRACE_START() do {} while (0)
#define TRACE(k) do {} while (0)
#define TRACE_FLUSH() do {} while (0)
#endif

// NARROW = false (production): 4 workgroups per CU at 128 VGPRs / 40.4 KB LDS, spill-free, the window in registers.
// NARROW = true (BLISSGPU_OPT_STFT_SHAPE = 1; the "other shape" of the round-4 review, built in round 5 for the A/B): 5
// workgroups per CU -- <= 96 VGPRs and 23 KB LDS.  What has to give: the window leaves the registers and is loaded with
// every frame's samples (16 more vector-memory loads per frame-thread); the two transposes go through a 17 KB float buffer
// in two halves (re, then im: twice the LDS instructions and four more barriers per frame); the magnitude row can no longer
// hide in a dead half of the exchange buffer (one more barrier).  Same arithmetic, same operands: rows bit-identical.
constexpr int STFT_LDS_N = 16 * EX2_PITCH;  // floats (17 408 B): 16 rows of 272 floats >= the 4128-float magnitude row
constexpr int EXN1_PITCH = 260;             // floats; == 4 (mod 32): the 64 readers of pass 2 fall two to a bank
template <bool LOADWIN, bool HALVES>  // NARROW = both; one alone is a measurement form (shapes 2 and 3)
__global__ __launch_bounds__(256, (LOADWIN && HALVES) ? 5 : 4) void stft8192_kernel(const float* __restrict__ pcm,
                                                       const SongDesc* __restrict__ songs, uint32_t n_songs,
                                                       const uint32_t* __restrict__ pfx_c,
                                                       const float* __restrict__ hann,
                                                       const float2* __restrict__ tw,
                                                       float* __restrict__ spec,
                                                       float* __restrict__ frame_max, uint32_t* __restrict__ h1,
                                                       uint32_t* __restrict__ peak_rec, uint32_t* __restrict__ peak_cnt,
                                                       uint32_t n_tiles) {
    constexpr bool NARROW = HALVES;  // the LDS layout follows the transposes
    __shared__ __attribute__((aligned(16))) float lds_raw[NARROW ? STFT_LDS_N : 2 * STFT_LDS];
    f2* const lds = reinterpret_cast<f2*>(lds_raw);
    float* const ldsf = lds_raw;
    constexpr int MAGS_TOP_W = NARROW ? 4096 : MAGS_TOP;  // word index of magnitude word 4096 (narrow: the row is contiguous)
    __shared__ float red[4];
    // peaks are first counted in an LDS window of the coarse-magnitude histogram (a frame's peaks lie
    // within [0.1 max, ~max], i.e. ~220 coarse bins) and flushed once per workgroup: global atomics on
    // a song's few hot histogram lines would otherwise serialise at the L2
    __shared__ uint32_t lhist[LHIST_BINS];
    __shared__ uint32_t lhist_base;
    __shared__ uint16_t peak_list[PIP_MAX_PER_FRAME + 2];
    __shared__ uint32_t peak_count;
    __shared__ __attribute__((aligned(16))) f2 tw256[256];  // W_256^(m k) at [16 m + k]: symmetric in (m, k)
    TRACE_DECL
    TRACE_INIT();
    {
        const float2 a = tw[32 * (((threadIdx.x >> 4) * (threadIdx.x & 15)) & 255)];
        tw256[threadIdx.x] = mk(a.x, a.y);
    }
    // Which frames a workgroup owns is chosen for the L2.  Consecutive frames share 5 987 of their 8 192 samples; a
    // workgroup walking 16 CONSECUTIVE frames re-reads them one frame period later, by when the 128 workgroups of its XCD
    // have streamed their own 32 KB windows through the 4 MiB L2 -- the PCM was fetched 2.2 times.  Instead the FOUR
    // workgroups of a 64-frame super-tile take every fourth frame each (member m: frames 64 s + m, + 4, + 8, ...) and sit
    // on the SAME XCD (blocks b, b + 8, b + 16, b + 24: dispatch puts block b on XCD b % 8), so the four frames being
    // loaded at any time overlap and a sample's 3.7 readers arrive within a fraction of a frame period.  A workgroup's
    // own consecutive frames (4 hops = 8 820 samples apart) no longer overlap at all.  Placement is a matter of speed
    // only: any block -> XCD map gives the same results.
    const uint32_t bx = blockIdx.x;
    const uint32_t wg = (bx & ~31u) | ((bx & 7u) << 2) | ((bx >> 3) & 3u);  // logical tile of this block
    if (wg >= n_tiles) return;                                             // grid padded to a multiple of 32
    const uint32_t s = find_segment(pfx_c, n_songs, wg);
    const SongDesc sd = songs[s];
    const uint32_t tile = wg - pfx_c[s];                                   // 4 x super-tile + member
    const uint32_t f_first = (tile >> 2) * (uint32_t)(STFT_GROUP * STFT_FRAMES_PER_WG) + (tile & 3u);
    const float* __restrict__ x = pcm + sd.pcm_off;
    const int t = threadIdx.x;
    const long n = (long)sd.n;
    const int lo4 = t & 15, hi4 = t >> 4;

    // descriptors: window table (8192 f32) and twiddle table (8192 float2); wave-uniform
    const __amdgpu_buffer_rsrc_t r_hann = __builtin_amdgcn_make_buffer_rsrc((void*)hann, 0, W8192 * 4, 0x00020000);


    const uint32_t t8 = 8u * (uint32_t)t;  // byte offset of (x[2t], x[2t+1]) / (hann[2t], hann[2t+1])
    // the window values of this thread's 16 complex inputs stay in registers for all frames of the tile
    f2 win[16];  // (LOADWIN: fetched per frame, see load_window)
    if constexpr (!LOADWIN) {
#pragma unroll
        for (int n1 = 0; n1 < 16; n1++) win[n1] = buf_load_f2(r_hann, t8, 2048u * n1);
    }
    // Twiddles come from a 256-entry LDS table and two per-thread constants instead of per-frame loads of
    // 4096- and 2048-entry tables (every vector-memory load in the frame loop costs an in-order vmcnt wait):
    //   pass 1:  W_4096^((16 m1 + m2) k1) = W_256^(m1 k1) * W_4096^(m2 k1); the second factor does not depend
    //            on m1, so it commutes with the pass-2 DFT over m1 and is applied as the thread constant c_p2
    //   split:   W_8192^(t + 256 j) = W_8192^t * W_32^j
    f2 c_p2, c_sp;
    {
        const float2 a = tw[2 * hi4 * lo4], b = tw[t];
        c_p2 = mk(a.x, a.y);
        c_sp = mk(b.x, b.y);
        asm volatile("" : "+v"(c_p2), "+v"(c_sp));  // consume here: no vmcnt wait on them inside the frame loop
    }
    uint32_t* hist = h1 + (size_t)s * H1_BINS;
    for (int i = t; i < LHIST_BINS; i += 256) lhist[i] = 0;
    bool have_base = false;
    // tw256 is read across wavefronts by the first frame's pass 1.  Without this barrier a wavefront that got ahead read
    // whatever the previous workgroup left in the LDS -- the same table, unless that was another kernel's data: about one
    // frame in 10^8 was transformed with stale twiddles (a chroma row off by 1e-6 .. 4e-3 once in some 10^4 songs, found
    // by tests/tools/determinism_check.py; every parity test passed).
    __syncthreads();

    // W_256^(m k): the table is symmetric, so the fifteen twiddles a thread needs in a pass (fixed m = t >> 4) are 128
    // consecutive bytes, read two at a time (ds_read_b128: 4 LDS cycles per pair; two 8-byte reads a row apart are merged
    // by hipcc into ds_read2_b64 at 8)
    auto tw_at = [&](int m, int k) -> f2 {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 pair = reinterpret_cast<const f4*>(tw256)[8 * m + (k >> 1)];
        return (k & 1) ? mk(pair.z, pair.w) : mk(pair.x, pair.y);
    };
    // raw samples of one frame: z[256*n1 + t] = (x[w0 + 2n], x[w0 + 2n + 1]), reflect only at the song edges
    // LOADWIN: the window is fetched (L1 / L2 hits) right in front of the multiply that consumes it -- requested with the
    // samples, or ahead of the peak classification, it is spilled across the peak phase (32 registers, measured)
    auto load_window = [&]() {
        if constexpr (LOADWIN) {
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++) win[n1] = buf_load_f2(r_hann, t8, 2048u * n1);
        }
    };
    auto load_frame = [&](uint32_t f, f2 (&xr)[16]) {
        const long w0 = (long)f * HOP_C - W8192 / 2;
        if (w0 >= 0 && w0 + W8192 <= n) {
            const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc((void*)(x + w0), 0, W8192 * 4, 0x00020000);
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++) xr[n1] = buf_load_f2(r_x, t8, 2048u * n1);
        } else {
            // numpy mode="reflect" (src/utils.rs:11-24) in 32-bit offsets relative to x + w0: positions before
            // the song mirror about sample 0, positions past the end about sample n - 1
            const int before = w0 < 0 ? (int)(-w0) : 0;
            const long rel_l = n - w0;
            const int rel = rel_l > 0x3fffffffL ? 0x3fffffff : (int)rel_l;
            const float* __restrict__ xw = x + w0;
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++) {
                int q0 = 2 * (256 * n1 + t), q1 = q0 + 1;
                q0 = q0 < before ? 2 * before - q0 : (q0 >= rel ? 2 * rel - 2 - q0 : q0);
                q1 = q1 < before ? 2 * before - q1 : (q1 >= rel ? 2 * rel - 2 - q1 : q1);
                xr[n1] = mk(xw[q0], xw[q1]);
            }
        }
    };

    // The next frame's samples are requested as soon as the current frame's registers are free (after the
    // split), so their HBM/L2 latency overlaps the peak-picking phase instead of stalling the next frame.
    f2 v[16];
    // The window multiply (the first use of the prefetched samples, i.e. the vmcnt wait) sits at the END of the
    // loop body: there every path has issued the loads followed by the stores, so the wait is "all but the
    // stores"; at the loop head the prologue path (no stores) would force a full vmcnt(0) drain per frame.
    if (f_first < sd.n_c) {
        load_frame(f_first, v);
        load_window();
#pragma unroll
        for (int n1 = 0; n1 < 16; n1++) v[n1] = v[n1] * win[n1];  // window (src/utils.rs:37-39, :49)
    }
    TRACE_START();
#pragma unroll 1
    for (int fi = 0; fi < STFT_FRAMES_PER_WG; fi++) {
        const uint32_t f = f_first + (uint32_t)(STFT_GROUP * fi);
        if (f >= sd.n_c) break;  // uniform
        // ---- pass 1: DFT over n1 at n2 = t = 16 m1 + m2; twiddle W_256^(m1 k1) ----
        radix16(v);
#pragma unroll
        for (int k1 = 1; k1 < 16; k1++)
            v[R16(k1)] = cmul_pk(v[R16(k1)], tw_at(hi4, k1));
        if constexpr (NARROW) {
            // the transpose in two halves through the float buffer: real parts, then imaginary parts
            f2 u[16];
#pragma unroll
            for (int k1 = 0; k1 < 16; k1++) ldsf[k1 * EXN1_PITCH + t] = v[R16(k1)].x;
            __syncthreads();
#pragma unroll
            for (int m1 = 0; m1 < 16; m1++) u[m1].x = ldsf[lo4 * EXN1_PITCH + 16 * m1 + hi4];
            __syncthreads();
#pragma unroll
            for (int k1 = 0; k1 < 16; k1++) ldsf[k1 * EXN1_PITCH + t] = v[R16(k1)].y;
            __syncthreads();
#pragma unroll
            for (int m1 = 0; m1 < 16; m1++) u[m1].y = ldsf[lo4 * EXN1_PITCH + 16 * m1 + hi4];
#pragma unroll
            for (int m1 = 0; m1 < 16; m1++) v[m1] = u[m1];
        } else {
#pragma unroll
        for (int k1 = 0; k1 < 16; k1++) lds[k1 * EX1_PITCH + t] = v[R16(k1)];
        TRACE(0);
        __syncthreads();
        TRACE(1);
        // ---- pass 2: thread (k1 = lo4, m2 = hi4): DFT over m1; twiddle W_4096^(m2 k1) * W_256^(m2 j1) ----
#pragma unroll
        for (int m1 = 0; m1 < 16; m1++) v[m1] = lds[lo4 * EX1_PITCH + 16 * m1 + hi4];
        }
        radix16(v);
        v[R16(0)] = cmul_pk(v[R16(0)], c_p2);
#pragma unroll
        for (int j1 = 1; j1 < 16; j1++) v[R16(j1)] = cmul_pk(v[R16(j1)], cmul_pk(tw_at(hi4, j1), c_p2));
        TRACE(2);
        __syncthreads();
        TRACE(3);
        if constexpr (NARROW) {
            f2 u[16];
#pragma unroll
            for (int j1 = 0; j1 < 16; j1++) ldsf[j1 * EX2_PITCH + t] = v[R16(j1)].x;
            __syncthreads();
#pragma unroll
            for (int m2 = 0; m2 < 16; m2++) u[m2].x = ldsf[hi4 * EX2_PITCH + 16 * m2 + lo4];
            __syncthreads();
#pragma unroll
            for (int j1 = 0; j1 < 16; j1++) ldsf[j1 * EX2_PITCH + t] = v[R16(j1)].y;
            __syncthreads();
#pragma unroll
            for (int m2 = 0; m2 < 16; m2++) u[m2].y = ldsf[hi4 * EX2_PITCH + 16 * m2 + lo4];
#pragma unroll
            for (int m2 = 0; m2 < 16; m2++) v[m2] = u[m2];
        } else {
#pragma unroll
        for (int j1 = 0; j1 < 16; j1++) lds[j1 * EX2_PITCH + t] = v[R16(j1)];  // = j1*272 + m2*16 + k1
        TRACE(4);
        __syncthreads();
        TRACE(5);
        // ---- pass 3: thread (k1 = lo4, j1 = hi4): DFT over m2 -> Z[t + 256*j2] ----
#pragma unroll
        for (int m2 = 0; m2 < 16; m2++) v[m2] = lds[hi4 * EX2_PITCH + 16 * m2 + lo4];
        }
        radix16(v);
        TRACE(6);
        __syncthreads();
        TRACE(7);
        // only the upper half (bins 2049..4095, the mirrors of this workgroup's bins 1..2047) is ever read back
        constexpr int ZOFF = NARROW ? 2048 : 0;  // narrow: Z[k], k >= 2048, lives at complex slot k - 2048 (16 KB in all)
#pragma unroll
        for (int j2 = 8; j2 < 16; j2++) lds[t + 256 * j2 - ZOFF] = v[R16(j2)];
        TRACE(8);
        __syncthreads();
        TRACE(9);
        // ---- real-input split + magnitude (src/utils.rs:60).  Z[k] and Z[4096-k] yield X[k] AND X[4096-k]:
        // thread t pairs its bins k = t + 256 j, j < 8, with their mirrors (k = 0 pairs DC with Nyquist);
        // bin 2048 (its own mirror) is done by thread 0 ----
        float m_lo[8], m_hi[8], m_mid = 0.0f;
        float mx = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float CJ = CONST_COS32[j], SJ = CONST_SIN32[j];
            const int k = t + 256 * j;
            float sq_k, sq_m;
            const f2 w = j == 0 ? c_sp : cmul_pk_s(c_sp, mk(CJ, -SJ));
            // k = 0 pairs DC with itself (thread 0's own register); every other mirror is in the upper half
            const f2 zm = lds[(k == 0 ? 2048 : 4096 - k) - ZOFF];
            split_pair_sq(v[R16(j)], (j == 0 && t == 0) ? v[R16(0)] : zm, w, sq_k, sq_m);
            m_lo[j] = mag_from_sq(sq_k);
            m_hi[j] = mag_from_sq(sq_m);
            mx = fmaxf(mx, fmaxf(m_lo[j], m_hi[j]));
        }
        if (t == 0) {
            m_mid = mag_from_sq(split_one_sq(v[R16(8)], v[R16(8)], mk(0.0f, -1.0f)));  // k = 2048: W_8192^2048 = -i
            mx = fmaxf(mx, m_mid);
        }
        const bool has_next = fi + 1 < STFT_FRAMES_PER_WG && f + STFT_GROUP < sd.n_c;  // uniform
        if (has_next) load_frame(f + STFT_GROUP, v);
        mx = wave_max_dpp(mx);
        // The split only reads the UPPER half of the exchange buffer (complex slots 2049..4095 = bytes 16 392..32 767);
        // the row of 4128 magnitudes goes into the dead lower half (words 0..4095) and, for bin 4096 and the zero padding,
        // into the 2 KB behind the upper half -- no barrier between the split reads and these writes.
        if constexpr (NARROW) __syncthreads();  // the row overwrites the mirrored half of Z: every split read must be done
        float* mags = reinterpret_cast<float*>(lds);
        float* mags_top = mags + MAGS_TOP_W - 4096;  // mags_top[4096 + i] = word MAGS_TOP_W + i
#pragma unroll
        for (int j = 0; j < 8; j++) {
            mags[t + 256 * j] = m_lo[j];
            if (j == 0 && t == 0) mags_top[4096] = m_hi[0];  // bin 4096 (Nyquist)
            else mags[4096 - (t + 256 * j)] = m_hi[j];
        }
        if (t == 0) mags[2048] = m_mid;
        if (t >= 1 && t < CBINS_PAD - 4096) mags_top[4096 + t] = 0.0f;  // zero padding after bin 4096
        {
            // the slot index is re-derived from the thread id inside the loop: kept live across the whole frame loop
            // it was the first register to be spilled, and its reload forced a vmcnt(0) drain per frame
            int slot = t >> 6;
            asm volatile("" : "+v"(slot));
            if (lane_id() == 0) red[slot] = mx;
        }
        if (t == 0) peak_count = 0;
        TRACE(10);
        __syncthreads();
        TRACE(11);
        // The spectrogram row goes to HBM from the LDS copy as 16-byte stores (5 per thread instead of 18 scalar
        // ones).  They are issued BEHIND the next frame's loads: vmcnt retires in order, so the wait for those
        // loads at the end of the iteration is "all but the stores" and never waits for an HBM write acknowledge.
        auto store_row = [&]() {
            const __amdgpu_buffer_rsrc_t r_row = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(spec + (sd.c_off + f) * (size_t)CBINS_PAD), 0, CBINS_PAD * 4, 0x00020000);
            const u32x4_t* mags4 = reinterpret_cast<const u32x4_t*>(lds);
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const int q = t + 256 * i;
                if (i < 4 || q < CBINS_PAD / 4) {
                    const u32x4_t val = i < 4 ? mags4[q] : mags4[q - 1024 + MAGS_TOP_W / 4];
                    __builtin_amdgcn_raw_buffer_store_b128(val, r_row, 16u * (uint32_t)q, 0, 2);  // nt: streamed once
                }
            }
        };
        store_row();
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (t == 0) frame_max[sd.c_off + f] = mx;
        if (!have_base) {  // uniform: anchor the LDS window LHIST_BINS/2 bins below the first frame's maximum
            if (t == 0) {
                const uint32_t top = coarse_bin((double)mx);
                lhist_base = top > (uint32_t)(LHIST_BINS * 3 / 4) ? top - (uint32_t)(LHIST_BINS * 3 / 4) : 0u;
            }
            have_base = true;
            __syncthreads();
        }
        TRACE(12);
        const uint32_t lbase = lhist_base;
        // The peak phase is a chain of LDS round trips with little arithmetic between them, the transform passes are long
        // runs of arithmetic: a wavefront in the peak phase gets the issue slot first, so its few instructions never queue
        // behind another workgroup's radix pass (FFT-8192 kernel -1.9 %; raising the priority of the exchange / split
        // phases as well, or lowering it inside the radix passes only, is slower than no priorities at all).
        __builtin_amdgcn_s_setprio(2);
        // ---- pip_track pass 1: count peaks by coarse magnitude bin.  Every lane tests its six bins; the peaks (about a
        // third of the bins) are compacted through an LDS list so the interpolation arithmetic runs on dense
        // wavefronts instead of six times under a one-third-full exec mask ----
        const double ref = 0.1 * (double)mx;
        const float thr = ref_floor_f32(ref);
        {
            // nine magnitudes at a time (one wait for the LDS instead of three), then branch-free tests
            uint32_t hits = 0;  // bit j: bin t + 256 j is a peak
#pragma unroll
            for (int h = 0; h < 2; h++) {
                float sb3[3], se3[3], sa3[3];
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    const int c = t + 256 * (3 * h + q);
                    sb3[q] = mags[c > 0 ? c - 1 : 0];
                    se3[q] = mags[c];
                    sa3[q] = mags[c + 1];
                }
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    const int j = 3 * h + q, c = t + 256 * j;
                    const uint32_t in_range = (256 * j >= PIP_LO && 256 * j + 255 <= PIP_HI) ? 1u : (uint32_t)(c >= PIP_LO) & (uint32_t)(c <= PIP_HI);
                    const uint32_t pk = in_range & (uint32_t)(sa3[q] <= se3[q]) & (uint32_t)(sb3[q] < se3[q]) & (uint32_t)(se3[q] > thr);
                    hits |= pk << j;
                }
            }
            const uint32_t mine = (uint32_t)__popc(hits);
            const uint32_t incl = wave_scan_incl_u32_dpp(mine);
            uint32_t wbase = 0;
            if (lane_id() == 63) wbase = atomicAdd(&peak_count, incl);  // the wave's slice of the list
            uint32_t pos = (uint32_t)__builtin_amdgcn_readlane((int)wbase, 63) + incl - mine;
#pragma unroll
            for (int j = 0; j < 6; j++)
                if ((hits >> j) & 1u) peak_list[pos++] = (uint16_t)(t + 256 * j);
            TRACE(13);
            __syncthreads();
            TRACE(14);
            const uint32_t n_peaks = peak_count;
            uint32_t* __restrict__ recs = peak_rec + (sd.c_off + f) * (size_t)PIP_MAX_PER_FRAME;
            if (t == 0) peak_cnt[sd.c_off + f] = n_peaks;
            // the next list entry is requested together with this peak's magnitudes: the dependent LDS latencies
            // (list -> magnitudes) overlap across the two or three peaks a thread classifies
            int c_next = t < n_peaks ? (int)peak_list[t] : PIP_LO;
            for (uint32_t i = t; i < n_peaks; i += 256) {
                const int c = c_next;
                const float sb = mags[c - 1], se = mags[c], sa = mags[c + 1];
                c_next = i + 256 < n_peaks ? (int)peak_list[i + 256] : PIP_LO;
                int pb;
                const uint32_t b = peak_classify(sb, se, sa, ref, c, &pb), rel = b - lbase;
                recs[i] = peak_record(b, pb, c);
                if (rel < (uint32_t)LHIST_BINS) atomicAdd(&lhist[rel], 1u);
                else atomicAdd(&hist[b], 1u);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (has_next) {
            load_window();
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++) v[n1] = v[n1] * win[n1];  // window of the next frame
        }
        TRACE(15);
        // mags (lds) is reused by the next frame.  (Moving this barrier behind the next frame's register-only pass-1
        // arithmetic, so that early waves do not idle here, costs 12 spilled VGPRs at 128 -- measured slower.)
        __syncthreads();
        TRACE(16);
    }
    TRACE_FLUSH();
    if (have_base) {
        const uint32_t lbase = lhist_base;
        for (int i = t; i < LHIST_BINS; i += 256) {
            const uint32_t c = lhist[i];
            if (c) atomicAdd(&hist[lbase + i], c);
        }
    }
}

void launch_stft8192(const Batch& b, const Workspace& w, const DeviceTables& t, hipStream_t st, int shape) {
    if (b.tiles_c == 0) return;
    auto k = stft8192_kernel<false, false>;
    if (shape == 1) k = stft8192_kernel<true, true>;
    else if (shape == 2) k = stft8192_kernel<true, false>;   // measurement forms: the window loaded per frame only,
    else if (shape == 3) k = stft8192_kernel<false, true>;   // the transposes in two halves only (both at 4 workgroups / CU)
    hipLaunchKernelGGL(k, dim3((b.tiles_c + 31u) & ~31u), dim3(256), 0, st,
                       b.pcm, b.songs, b.n_songs, b.pfx_c, t.hann8192, t.tw8192, w.spec, w.frame_max, w.h1, w.peak_rec, w.peak_cnt,
                       b.tiles_c);
}

// ------------------------------------------------------------------------------------------------
// tuning: locate the coarse bins that hold the two middle order statistics
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tune_select_kernel(const SongDesc* __restrict__ songs,
                                                          const uint32_t* __restrict__ h1,
                                                          TuningState* __restrict__ tuning,
                                                          uint32_t* __restrict__ cand_cursor, uint32_t cand_pool,
                                                          uint32_t first_song) {
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t s_total, s_blo, s_bhi;
    const uint32_t s = blockIdx.x + first_song;
    const int tid = threadIdx.x;
    TuningState* ts = tuning + s;
    auto no_peaks = [&]() {
        if (tid == 0) {
            ts->n_peaks = 0; ts->tuning_idx = -1; ts->n_cand = 0; ts->b_lo = 1; ts->b_hi = 0; ts->below = 0;
            ts->cand_off = 0; ts->cand_cap = 0;
        }
    };
    if (!songs[s].ok) { no_peaks(); return; }
    const uint32_t* hist = h1 + (size_t)s * H1_BINS;
    constexpr int PER = H1_BINS / 256;  // 64 consecutive bins per thread (64 VGPRs: no spill at 136)
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
    if (total == 0) { no_peaks(); return; }
    // ndarray-stats Midpoint: lower = floor(0.5*(n-1)), higher = ceil(0.5*(n-1))
    const uint32_t r_lo = (total - 1) / 2, r_hi = total - 1 - r_lo;
    if (tid == 0) { ts->n_peaks = total; ts->n_cand = 0; ts->tuning_idx = -1; }
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const uint32_t c = local[i];
        if (c) {
            if (before <= r_lo && r_lo < before + c) { ts->b_lo = tid * PER + i; ts->below = before; s_blo = tid * PER + i; }
            if (before <= r_hi && r_hi < before + c) { ts->b_hi = tid * PER + i; s_bhi = tid * PER + i; }
        }
        before += c;
    }
    __syncthreads();
    // The peaks inside [b_lo, b_hi] are the candidates of the exact select; the histogram already knows how many there
    // are, so the song takes exactly that many slots from the chunk's pool.  A song the pool cannot serve (cand_cap = 0)
    // is handled by the re-scan path of tune_final_kernel.
    const uint32_t blo = s_blo, bhi = s_bhi;
    uint32_t mine = 0;
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const uint32_t bin = tid * PER + i;
        if (bin >= blo && bin <= bhi) mine += local[i];
    }
    mine = wave_sum(mine);
    if (lane_id() == 0) wsum[wave_id()] = mine;
    __syncthreads();
    if (tid == 0) {
        const uint32_t want = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        const uint32_t off = atomicAdd(cand_cursor, want);
        if (off <= cand_pool && want <= cand_pool - off) {
            ts->cand_off = off;
            ts->cand_cap = want;
        } else {
            atomicSub(cand_cursor, want);  // leave the room to the songs that fit
            ts->cand_off = 0;
            ts->cand_cap = 0;
        }
    }
}

void launch_tune_select(const Batch& b, const Workspace& w, hipStream_t st, const SongRange* r) {
    const uint32_t s0 = r ? r->s0 : 0, s1 = r ? r->s1 : b.n_songs;
    if (s1 <= s0) return;
    hipLaunchKernelGGL(tune_select_kernel, dim3(s1 - s0), dim3(256), 0, st, b.songs, w.h1, w.tuning, w.cand_cursor,
                       w.cand_cap, s0);
}

// ------------------------------------------------------------------------------------------------
// tuning pass 2: re-run the peak test on the stored magnitudes; peaks above the median's coarse bin
// go straight into the pitch histogram, peaks inside it are kept as candidates for the exact select
// ------------------------------------------------------------------------------------------------
constexpr int P2_SLOW_CAP = 1024;  // per-wave list of peaks that need the f64 path
constexpr int P2_FRAMES_PER_WAVE = CH_TILE / 4;

// A wave owns a frame at a time and walks the frame's peak records: peaks above the median's coarse bins add their
// (pre-computed) pitch bin to the histogram, peaks below are dropped, and the few that need f64 -- records without a
// provable pitch bin, candidates inside the median's bins -- are queued in an LDS list that is flushed across
// frames on dense wavefronts, reading their three magnitudes back from the spectrogram.
__global__ __launch_bounds__(256) void tune_pass2_kernel(const SongDesc* __restrict__ songs, uint32_t n_songs,
                                                         const uint32_t* __restrict__ pfx_ct,
                                                         const float* __restrict__ spec,
                                                         const float* __restrict__ frame_max,
                                                         const uint32_t* __restrict__ peak_rec,
                                                         const uint32_t* __restrict__ peak_cnt,
                                                         TuningState* __restrict__ tuning,
                                                         uint32_t* __restrict__ hist100,
                                                         double* __restrict__ cand_mag,
                                                         uint8_t* __restrict__ cand_pb, uint32_t first_tile) {
    __shared__ uint32_t hist[N_TUNING];
    __shared__ uint32_t slow_list[4][P2_SLOW_CAP];  // (frame slot << 16) | centre bin
    const uint32_t bx = blockIdx.x + first_tile;
    const uint32_t s = find_segment(pfx_ct, n_songs, bx);
    const SongDesc sd = songs[s];
    const uint32_t tile = bx - pfx_ct[s];
    const int tid = threadIdx.x, lane = lane_id(), wave = wave_id();
    TuningState* ts = tuning + s;
    const uint32_t b_lo = ts->b_lo, b_hi = ts->b_hi;
    if (ts->n_peaks == 0) return;
    const uint32_t cand_off = ts->cand_off;
    const uint32_t cand_cap = ts->cand_cap;
    const bool pooled = cand_cap != 0;  // false: the pool was exhausted, tune_final re-scans the records instead
    if (tid < N_TUNING) hist[tid] = 0;
    __syncthreads();
    uint32_t n_slow = 0;  // wave-uniform

    auto flush = [&]() {
        for (uint32_t i = lane; i < n_slow; i += WAVE) {
            const uint32_t e = slow_list[wave][i];
            const int c = (int)(e & 0xFFFFu);
            const uint32_t f = tile * CH_TILE + wave + 4 * (e >> 16);
            const float* row = spec + (sd.c_off + f) * (size_t)CBINS_PAD;
            const double ref = 0.1 * (double)frame_max[sd.c_off + f];
            double mag, pitch;
            if (pip_peak_core(row[c - 1], row[c], row[c + 1], ref, c, &mag, &pitch)) {
                const uint32_t b = coarse_bin(mag);
                if (b > b_hi) {
                    atomicAdd(&hist[pitch_bin(pitch)], 1u);
                } else if (b >= b_lo) {
                    // The pool slice was sized from the STFT-time histogram of these very bins; should the count ever
                    // disagree, the surplus stays out of the neighbour's slice and tune_final_kernel sees n_cand != cand_cap.
                    const uint32_t slot = atomicAdd(&ts->n_cand, 1u);
                    if (pooled && slot < cand_cap) {
                        cand_mag[cand_off + slot] = mag;
                        cand_pb[cand_off + slot] = (uint8_t)pitch_bin(pitch);
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        n_slow = 0;
    };
    // A frame has up to 714 records = 12 words per lane; they are requested four at a time (and the next frame's record
    // count one frame ahead), so a frame pays the memory latency three times at most instead of once per 64 records.
    uint32_t n_rec_next = 0;
    {
        const uint32_t f0 = tile * CH_TILE + wave;
        if (f0 < sd.n_c) n_rec_next = peak_cnt[sd.c_off + f0];
    }
    for (int i = 0; i < P2_FRAMES_PER_WAVE; i++) {
        const uint32_t f = tile * CH_TILE + wave + 4 * i;
        if (f >= sd.n_c) break;  // wave-uniform
        // (a frame holds at most PIP_MAX_PER_FRAME records -- two neighbouring bins cannot both be peaks; the clamp costs one
        // instruction per frame and keeps a count that was never written, should a future change leave one, from walking off
        // the record row)
        const uint32_t n_rec = n_rec_next < (uint32_t)PIP_MAX_PER_FRAME ? n_rec_next : (uint32_t)PIP_MAX_PER_FRAME;
        if (i + 1 < P2_FRAMES_PER_WAVE && f + 4 < sd.n_c) n_rec_next = peak_cnt[sd.c_off + f + 4];
        const uint32_t* __restrict__ recs = peak_rec + (sd.c_off + f) * (size_t)PIP_MAX_PER_FRAME;
        if (n_slow + PIP_MAX_PER_FRAME > P2_SLOW_CAP) flush();  // wave-uniform
        for (uint32_t j0 = 0; j0 < n_rec; j0 += 4 * WAVE) {  // uniform trip count
            uint32_t r4[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t j = j0 + WAVE * u + lane;
                r4[u] = j < n_rec ? recs[j] : 0xFFFFFFFFu;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (j0 + WAVE * u >= n_rec) break;  // wave-uniform
                const uint32_t j = j0 + WAVE * u + lane;
                bool slow = false;
                uint32_t c = 0;
                if (j < n_rec) {
                    const uint32_t r = r4[u];
                    const uint32_t b = r >> 18, pbf = (r >> 11) & 0x7Fu;
                    c = r & 0x7FFu;
                    if (b > b_hi) {
                        if (pbf) atomicAdd(&hist[pbf - 1], 1u);
                        else slow = true;
                    } else if (b >= b_lo) {
                        slow = true;
                    }
                }
                const uint64_t mask = __ballot(slow);
                if (mask) {
                    if (slow) slow_list[wave][n_slow + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = ((uint32_t)i << 16) | c;
                    n_slow += (uint32_t)__popcll(mask);
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    flush();
    __syncthreads();
    if (tid < N_TUNING && hist[tid]) atomicAdd(&hist100[(size_t)s * N_TUNING + tid], hist[tid]);
}

void launch_tune_pass2(const Batch& b, const Workspace& w, hipStream_t st, const SongRange* r) {
    const uint32_t t0 = r ? r->ct0 : 0, t1 = r ? r->ct1 : b.tiles_ct;
    if (t1 <= t0) return;
    hipLaunchKernelGGL(tune_pass2_kernel, dim3(t1 - t0), dim3(256), 0, st, b.songs, b.n_songs, b.pfx_ct, w.spec,
                       w.frame_max, w.peak_rec, w.peak_cnt, w.tuning, w.hist100, w.cand_mag, w.cand_pb, t0);
}

// ------------------------------------------------------------------------------------------------
// tuning final: exact order statistics among the candidates (8-bit MSD radix select on the
// order-preserving u64 image of the f64 magnitudes), Midpoint threshold, histogram, first argmax
// ------------------------------------------------------------------------------------------------
// for_each(fn) calls fn(magnitude, pitch bin) for every candidate of the song, spread over the 256 threads
template <typename ForEach>
__device__ uint64_t block_radix_select(ForEach&& for_each, uint32_t rank, uint32_t* hist, uint32_t* s_digit, uint32_t* s_rank,
                                       uint32_t* s_wave) {
    const int tid = threadIdx.x;
    uint64_t prefix = 0, mask = 0;
    for (int shift = 56; shift >= 0; shift -= 8) {
        hist[tid] = 0;
        __syncthreads();
        for_each([&](double mag, int) {
            const uint64_t k = f64_key(mag);
            const bool in = (k & mask) == prefix;
            const uint32_t d = (uint32_t)(k >> shift) & 0xFFu;
            // the candidates share their leading bytes (they come from one or two coarse magnitude bins): a wavefront whose
            // keys all carry the same digit adds its count once instead of serialising 64 atomics on one LDS word
            const uint64_t m = __ballot(in);
            if (m != 0) {
                const int first = __ffsll((unsigned long long)m) - 1;
                const uint32_t d0 = (uint32_t)__shfl((int)d, first, WAVE);
                if (__ballot(in && d == d0) == m) {
                    if (lane_id() == first) atomicAdd(&hist[d0], (uint32_t)__popcll(m));
                } else if (in) {
                    atomicAdd(&hist[d], 1u);
                }
            }
        });
        __syncthreads();
        {
            // digit d with below(d) <= rank < below(d) + hist[d]: one thread per digit, prefix sums by wave scan (a serial
            // walk over the 256 LDS counters by one thread cost ~25 us per pass, eight passes per song)
            const uint32_t h = hist[tid];
            const uint32_t incl = wave_scan_incl_u32(h);
            if (lane_id() == 63) s_wave[wave_id()] = incl;
            __syncthreads();
            uint32_t below = incl - h;
            for (int w = 0; w < wave_id(); w++) below += s_wave[w];
            if (h != 0 && below <= rank && rank < below + h) {  // exactly one digit qualifies (rank < number of keys)
                *s_digit = (uint32_t)tid;
                *s_rank = rank - below;
            }
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
                                                         const uint8_t* __restrict__ cand_pb,
                                                         const float* __restrict__ spec,
                                                         const float* __restrict__ frame_max,
                                                         const uint32_t* __restrict__ peak_rec,
                                                         const uint32_t* __restrict__ peak_cnt, uint32_t first_song) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t s_digit, s_rank, s_cnt_le, s_wave[4];
    __shared__ unsigned long long s_min_gt;
    const uint32_t s = blockIdx.x + first_song;
    const int tid = threadIdx.x;
    TuningState* ts = tuning + s;
    const SongDesc sd = songs[s];
    if (!sd.ok || ts->n_peaks == 0) return;  // tuning_idx stays -1 => tuning 0.0 (src/chroma.rs:377-379)
    const uint32_t total = ts->n_peaks, nc = ts->n_cand;
    const uint32_t r_lo = (total - 1) / 2, r_hi = total - 1 - r_lo;
    const uint32_t k_lo = r_lo - ts->below, k_hi = r_hi - ts->below;
    const uint32_t b_lo = ts->b_lo, b_hi = ts->b_hi;
    // pooled candidates are trusted only when pass 2 filed exactly as many as the histogram promised; anything else
    // (never observed) takes the exact re-scan of the records
    const bool pooled = ts->cand_cap != 0 && nc == ts->cand_cap;
    const double* v = cand_mag + ts->cand_off;
    const uint8_t* pb = cand_pb + ts->cand_off;

    // Candidates: the pool slots tuning pass 2 filled -- or, for a song the pool could not serve, the same peaks
    // re-derived from the peak records and the stored spectrogram (identical arithmetic, pip_peak_core): slower by the
    // nine passes over the records, taken only by songs whose peaks crowd into the median's coarse magnitude bins
    // (e.g. click tracks: a flat spectrum) in a chunk whose pool is already full.
    auto for_each = [&](auto&& fn) {
        if (pooled) {
            // eight candidates requested per trip: one at a time, every pass of the select paid the L2 latency nc / 256 times
            for (uint32_t i0 = tid; i0 < nc; i0 += 8 * 256) {
                double m8[8];
                int b8[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint32_t i = i0 + 256u * u;
                    m8[u] = i < nc ? v[i] : 0.0;
                    b8[u] = i < nc ? (int)pb[i] : 0;
                }
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (i0 + 256u * u < nc) fn(m8[u], b8[u]);
            }
        } else {
            for (uint32_t f = 0; f < sd.n_c; f++) {
                const uint32_t n_rec_raw = peak_cnt[sd.c_off + f];
                const uint32_t n_rec = n_rec_raw < (uint32_t)PIP_MAX_PER_FRAME ? n_rec_raw : (uint32_t)PIP_MAX_PER_FRAME;
                const uint32_t* __restrict__ recs = peak_rec + (sd.c_off + f) * (size_t)PIP_MAX_PER_FRAME;
                const float* __restrict__ row = spec + (sd.c_off + f) * (size_t)CBINS_PAD;
                const double ref = 0.1 * (double)frame_max[sd.c_off + f];
                for (uint32_t j = tid; j < n_rec; j += 256) {
                    const uint32_t r = recs[j], b = r >> 18;
                    if (b < b_lo || b > b_hi) continue;
                    const int c = (int)(r & 0x7FFu);
                    double mag, pitch;
                    if (pip_peak_core(row[c - 1], row[c], row[c + 1], ref, c, &mag, &pitch)) fn(mag, pitch_bin(pitch));
                }
            }
        }
    };

    const uint64_t key_lo = block_radix_select(for_each, k_lo, hist, &s_digit, &s_rank, s_wave);
    uint64_t key_hi = key_lo;
    if (k_hi != k_lo) {
        // the next order statistic: key_lo again if it is repeated, else the smallest key above it
        if (tid == 0) { s_cnt_le = 0; s_min_gt = ~0ull; }
        __syncthreads();
        uint32_t cnt = 0;
        unsigned long long mn = ~0ull;
        for_each([&](double mag, int) {
            const uint64_t k = f64_key(mag);
            if (k <= key_lo) cnt++;
            else if (k < mn) mn = k;
        });
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
    for_each([&](double mag, int bin) {
        if (mag >= thr) atomicAdd(&hist[bin], 1u);
    });
    __syncthreads();
    if (tid == 0) {
        uint32_t best = 0;  // ndarray-stats argmax keeps the first maximum
        for (uint32_t k = 1; k < N_TUNING; k++)
            if (hist[k] > hist[best]) best = k;
        ts->tuning_idx = (int32_t)best;
    }
}

void launch_tune_final(const Batch& b, const Workspace& w, hipStream_t st, const SongRange* r) {
    const uint32_t s0 = r ? r->s0 : 0, s1 = r ? r->s1 : b.n_songs;
    if (s1 <= s0) return;
    hipLaunchKernelGGL(tune_final_kernel, dim3(s1 - s0), dim3(256), 0, st, b.songs, w.tuning, w.hist100,
                       w.cand_mag, w.cand_pb, w.spec, w.frame_max, w.peak_rec, w.peak_cnt, s0);
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

__global__ __launch_bounds__(256) void chroma_kernel(const SongDesc* __restrict__ songs, uint32_t n_songs, const uint32_t* __restrict__ pfx_cw,
                              const uint32_t* __restrict__ pfx_ct, const float* __restrict__ spec,
                              const double* __restrict__ bank, const TuningState* __restrict__ tuning,
                              double* __restrict__ chroma_part, double* __restrict__ dbg_chroma, uint32_t first_wg);

void launch_chroma(const Batch& b, const Workspace& w, const DeviceTables& t, hipStream_t st, const SongRange* r) {
    const uint32_t g0 = r ? r->cw0 : 0, g1 = r ? r->cw1 : b.tiles_cw;
    if (g1 <= g0) return;
    hipLaunchKernelGGL(chroma_kernel, dim3(g1 - g0), dim3(256), 0, st, b.songs, b.n_songs, b.pfx_cw, b.pfx_ct, w.spec,
                       t.chroma_bank, w.tuning, w.chroma_part, w.dbg_chroma, g0);
}

// (a probe translation unit that brings its own contraction -- tests/tools/probes/handpipe -- defines this before it
// includes this file)
#ifndef BG_PROBE_REPLACES_CHROMA_KERNEL
// One wavefront owns one 64-frame tile (= one chroma_part slot): four 16-frame MFMA sub-tiles share every
// filter (A) fragment, so the L2-resident filter bank is read once per 64 frames instead of once per 16.
__global__ __launch_bounds__(256) void chroma_kernel(const SongDesc* __restrict__ songs, uint32_t n_songs,
                                                     const uint32_t* __restrict__ pfx_cw,
                                                     const uint32_t* __restrict__ pfx_ct,
                                                     const float* __restrict__ spec,
                                                     const double* __restrict__ bank,
                                                     const TuningState* __restrict__ tuning,
                                                     double* __restrict__ chroma_part,
                                                     double* __restrict__ dbg_chroma, uint32_t first_wg) {
    __shared__ double tile_c[4][4][16][13];
    const uint32_t bx = blockIdx.x + first_wg;
    const uint32_t s = find_segment(pfx_cw, n_songs, bx);
    const SongDesc sd = songs[s];
    const int lane = lane_id(), wave = wave_id();
    const uint32_t tile64 = (bx - pfx_cw[s]) * 4 + wave;   // 64-frame tile of this wave
    const uint32_t n_tiles = pfx_ct[s + 1] - pfx_ct[s];
    if (tile64 >= n_tiles) return;  // wave-uniform; no workgroup barriers below
    const int i16 = lane & 15, g = lane >> 4;
    const int tidx = tuning[s].tuning_idx;
    const int slot = (tidx < 0) ? N_TUNING : tidx;
    const uint32_t f0 = tile64 * CH_TILE;

    // K is consumed in an order chosen for the memory system, not 0, 1, 2, ...: within a 32-bin (128-byte) block the four
    // lanes of a frame sit 32 bytes apart and take two 16-byte pieces each, so EVERY load instruction touches both
    // 64-byte halves of its 16 lines.  With the natural order (lane g on bins 4g..4g+3: 64 contiguous bytes per frame and
    // instruction) the same bytes arrive at 4.7 TB/s instead of 5.9 (tests/tools/probes/bw_probe.hip).  The filter (A)
    // fragments follow the same permutation, so the sum over K has the same terms.  (Filter rows 12..15 of the 16-row MFMA
    // tile do not exist: those lanes re-read row 11 and their results are never stored.)
    const double* __restrict__ arow = bank + ((size_t)slot * BANK_ROWS + (i16 < BANK_ROWS ? i16 : BANK_ROWS - 1)) * BANK_PITCH + 8 * g;
    const float* brow[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint32_t fj = f0 + 16 * q + i16;
        if (fj >= sd.n_c) fj = sd.n_c - 1;  // clamped rows are computed and discarded
        brow[q] = spec + (sd.c_off + fj) * (size_t)CBINS_PAD + 8 * g;
    }
    // two independent accumulator chains per sub-tile so consecutive MFMAs do not wait on each other
    double4_t acc[4][2];
#pragma unroll
    for (int q = 0; q < 4; q++) { acc[q][0] = double4_t{0.0, 0.0, 0.0, 0.0}; acc[q][1] = double4_t{0.0, 0.0, 0.0, 0.0}; }
    // K loop in 32-bin blocks (two 16-bin MFMA steps), software-pipelined over two register buffers: block j+1's ten
    // loads are issued before block j's 32 MFMAs, so a wavefront always has a block in flight while it computes (with two
    // wavefronts per SIMD, a load -> drain -> MFMA loop leaves the matrix pipe idle whenever the HBM latency exceeds one
    // block's MFMA time).  Every load and every wait here is the compiler's.
    //
    // A faster variant of this loop (v_mfma_f64_4x4x4_4b_f64: three instructions cover the 12 chroma rows exactly and run
    // at 77 instead of 61 TFLOP/s; spectrogram ring of three blocks and filter values staged through the LDS, all loads as
    // asm statements with hand-counted s_waitcnt vmcnt) ran at 6.0 ms, passed every parity test -- and returned a wrong
    // chroma row about once in 5 000 songs (tests/tools/determinism_check.py; the same MFMAs with compiler-managed loads
    // are clean but take 9.1 ms, because the compiler then drains the whole queue per block).  Round 3 bisected it to the
    // counted wait: `s_waitcnt vmcnt(8)` behind a mix of global->LDS transfers and ordinary loads fails, vmcnt(7) / (4) /
    // (0) are clean (profiles/r03_handpipe_bisect.txt; probe build: tests/tools/probes/handpipe).  The variant is not
    // shipped; nothing in the library mixes global->LDS transfers with counted waits.
    struct KBlock {
        double4_t a[2];
        float4 b[2][4];
    };
    auto k_load = [&](int k0, KBlock& kb) {
#pragma unroll
        for (int u = 0; u < 2; u++) kb.a[u] = *reinterpret_cast<const double4_t*>(arow + k0 + 4 * u);
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int u = 0; u < 2; u++) kb.b[u][q] = *reinterpret_cast<const float4*>(brow[q] + k0 + 4 * u);
        __builtin_amdgcn_sched_barrier(0);  // the block's loads stay where they are written: ahead of the previous block's MFMAs
    };
    auto k_mma = [&](const KBlock& kb) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float4 v = kb.b[u][q];
                const double b0 = (double)v.x * (double)v.x, b1 = (double)v.y * (double)v.y;
                const double b2 = (double)v.z * (double)v.z, b3 = (double)v.w * (double)v.w;
                acc[q][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(kb.a[u].x, b0, acc[q][0], 0, 0, 0);
                acc[q][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(kb.a[u].y, b1, acc[q][1], 0, 0, 0);
                acc[q][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(kb.a[u].z, b2, acc[q][0], 0, 0, 0);
                acc[q][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(kb.a[u].w, b3, acc[q][1], 0, 0, 0);
            }
        }
    };
    constexpr int KBLOCKS = CBINS_PAD / 32;
    static_assert(CBINS_PAD % 32 == 0 && BANK_PITCH == CBINS_PAD && KBLOCKS % 2 == 1, "pairs of blocks and one last block");
    KBlock k0buf, k1buf;
    k_load(0, k0buf);
#pragma unroll 1
    for (int kb = 0; kb + 2 < KBLOCKS; kb += 2) {
        k_load(32 * (kb + 1), k1buf);
        k_mma(k0buf);
        k_load(32 * (kb + 2), k0buf);
        k_mma(k1buf);
    }
    k_mma(k0buf);
    // Epilogue on all 64 lanes at once: the four C tiles go through LDS so that lane 16 q + i owns frame 16 q + i
    // (12 chroma values), instead of four passes of the f64 exp / template arithmetic on 16 active lanes each.
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const double4_t cacc = acc[q][0] + acc[q][1];
        // C[row = g + 4r][col = i16]: rows are chroma classes, columns are frames
#pragma unroll
        for (int r = 0; r < 3; r++) tile_c[wave][q][i16][g + 4 * r] = cacc[r];
    }
    __builtin_amdgcn_wave_barrier();
    double feat[10];
#pragma unroll
    for (int t = 0; t < 10; t++) feat[t] = 0.0;
    if (f0 + lane < sd.n_c) {
        double c[12], sum = 0.0;
#pragma unroll
        for (int k = 0; k < 12; k++) { c[k] = tile_c[wave][lane >> 4][lane & 15][k]; sum += fabs(c[k]); }
        if (sum < DBL_MIN) sum = 1.0;          // chroma_stft column normalisation (:404-410)
        if (dbg_chroma) {  // wave-uniform (tap of the parity tests): chroma_stft's column of this frame
#pragma unroll
            for (int k = 0; k < 12; k++) dbg_chroma[(sd.c_off + f0 + lane) * 12 + k] = c[k] / sum;
        }
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
#pragma unroll
    for (int t = 0; t < 10; t++) {
        // sum over the 64 frames of this tile in the order of the previous layout: the four frames {i, 16 + i, 32 + i,
        // 48 + i} first (sequentially, q ascending), then the 16 partial sums by xor-butterfly
        double v = feat[t];
        const double v1 = __shfl(v, (lane & 15) + 16, WAVE), v2 = __shfl(v, (lane & 15) + 32, WAVE), v3 = __shfl(v, (lane & 15) + 48, WAVE);
        v = __shfl(v, lane & 15, WAVE);
        v = ((v + v1) + v2) + v3;
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
        if (lane == 0) chroma_part[(size_t)(pfx_ct[s] + tile64) * 10 + t] = v;
    }
}

#endif  // BG_PROBE_REPLACES_CHROMA_KERNEL

}  // namespace bg
