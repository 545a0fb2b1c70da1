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
//   * for odd frames, the SpecFlux onset value over the correct 257 bins (src/aubio.rs:455-467)
//   * per 256-sample block the sum of squares and the zero-crossing count of the PCM itself (LoudnessDesc,
//     src/misc.rs:12-18; the tempo silence test, src/aubio.rs:1258-1276; number_crossings, src/utils.rs:81-95):
//     the kernel already holds every sample in registers, so there is no separate pass over the PCM.
//
// Mapping: a 16-lane group owns one frame at a time (4 frames per wavefront, 16 per workgroup).  The
// 512 real samples are packed as 256 complex values = 16 x 16: each lane holds 16 of them in registers,
// runs a radix-16 pass, transposes inside its group through a padded LDS tile, runs the second radix-16
// pass, and a second LDS round trip re-orders the spectrum so that lane l ends up with the 16
// CONSECUTIVE bins 16l..16l+15 (needed by the rolloff prefix sum) and with Z[256-k] for the real-input
// split.  Reductions stay inside the 16-lane row (DPP).  A group walks FRAMES_PER_GROUP consecutive
// frames so the previous tempo frame's magnitudes stay in registers (one halo FFT per group).  The lane's
// constants (window, twiddles) are register resident: two waves per SIMD without table reads beat three
// waves with LDS tables.
#include <float.h>
#include <stdlib.h>

#include <type_traits>

#include "device_utils.hpp"
#include "fft_r16.hpp"
#include "internal.hpp"

namespace bg {

#ifndef FFT512_PAIR_SPLIT
#define FFT512_PAIR_SPLIT 1
#endif
constexpr int GROUPS_PER_WG = 16;
constexpr int FRAMES_PER_GROUP = F512_TILE / GROUPS_PER_WG;
// rolloff: distance (relative to the frame's energy) below which the parallel and the sequential summation orders might
// disagree about a bin; 704 u, u = 2^-24 (see the frame epilogue)
constexpr float ROLL_GUARD = 704.0f / 16777216.0f;
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
// product of a double across the row: the two halves of the value travel by DPP (VALU only -- a ds_bpermute butterfly
// would pay the LDS crossbar latency four times per frame)
__device__ __forceinline__ double row16_prod(double v) {
    auto step = [](double x, auto ctrl) {
        const long long b = __double_as_longlong(x);
        const int lo = dpp_mov<decltype(ctrl)::value>((int)(b & 0xFFFFFFFFll)), hi = dpp_mov<decltype(ctrl)::value>((int)(b >> 32));
        return x * __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    };
    v = step(v, std::integral_constant<int, DPP_QUAD_XOR1>{});
    v = step(v, std::integral_constant<int, DPP_QUAD_XOR2>{});
    v = step(v, std::integral_constant<int, DPP_ROW_HALF_MIRROR>{});
    v = step(v, std::integral_constant<int, DPP_ROW_MIRROR>{});
    return v;
}
__device__ __forceinline__ float row16_scan_incl(float v) {  // row_shr:n with zero fill
    v += dpp_mov<0x111>(v);
    v += dpp_mov<0x112>(v);
    v += dpp_mov<0x114>(v);
    v += dpp_mov<0x118>(v);
    return v;
}

// Several row reductions at once, the chains interleaved by hand.  Left to the compiler, two float reductions issued together are
// paired by the SLP vectoriser into v_pk_add_f32 -- which takes no DPP operand, so every step becomes two v_mov_b32_dpp + one
// packed add per PAIR (three instructions where two v_add_f32_dpp do) -- and a single chain pays two wait states between a DPP
// read and the add that wrote its source.  Interleaved, every DPP source was written at least three instructions earlier (the
// leading s_nop covers whatever instruction stands in front of the block: hazards inside an asm statement are not the
// compiler's).  dpp(v) + v is v + dpp(v): the same bits as row16_sum / row16_scan_incl.
#define BG_DPP_STEP(OP, DST, SRC, CTRL) OP " " DST ", " SRC ", " SRC " " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
// a, b, d: row totals of the inputs in every lane; c: inclusive scan along the row (row_shr with zero fill).  The first step
// writes the results' registers from the inputs', so an input that stays live costs no copy.
__device__ __forceinline__ void row16_sum3_scan1(float in_a, float in_b, float in_c, float in_d, float& a, float& b, float& c, float& d) {
    asm("s_nop 1\n\t"
        BG_DPP_STEP("v_add_f32_dpp", "%0", "%4", "quad_perm:[1,0,3,2]") BG_DPP_STEP("v_add_f32_dpp", "%1", "%5", "quad_perm:[1,0,3,2]")
        BG_DPP_STEP("v_add_f32_dpp", "%2", "%6", "row_shr:1") BG_DPP_STEP("v_add_f32_dpp", "%3", "%7", "quad_perm:[1,0,3,2]")
        BG_DPP_STEP("v_add_f32_dpp", "%0", "%0", "quad_perm:[2,3,0,1]") BG_DPP_STEP("v_add_f32_dpp", "%1", "%1", "quad_perm:[2,3,0,1]")
        BG_DPP_STEP("v_add_f32_dpp", "%2", "%2", "row_shr:2") BG_DPP_STEP("v_add_f32_dpp", "%3", "%3", "quad_perm:[2,3,0,1]")
        BG_DPP_STEP("v_add_f32_dpp", "%0", "%0", "row_half_mirror") BG_DPP_STEP("v_add_f32_dpp", "%1", "%1", "row_half_mirror")
        BG_DPP_STEP("v_add_f32_dpp", "%2", "%2", "row_shr:4") BG_DPP_STEP("v_add_f32_dpp", "%3", "%3", "row_half_mirror")
        BG_DPP_STEP("v_add_f32_dpp", "%0", "%0", "row_mirror") BG_DPP_STEP("v_add_f32_dpp", "%1", "%1", "row_mirror")
        BG_DPP_STEP("v_add_f32_dpp", "%2", "%2", "row_shr:8") BG_DPP_STEP("v_add_f32_dpp", "%3", "%3", "row_mirror")
        : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
        : "v"(in_a), "v"(in_b), "v"(in_c), "v"(in_d));
}
// two float row totals and an integer one, three chains (a DPP source written two instructions earlier: exactly the two wait
// states the read needs), in place
#define BG_DPP3(CTRL)                                                                                           \
    BG_DPP_STEP("v_add_f32_dpp", "%0", "%0", CTRL) BG_DPP_STEP("v_add_f32_dpp", "%1", "%1", CTRL) BG_DPP_STEP("v_add_u32_dpp", "%2", "%2", CTRL)
__device__ __forceinline__ void row16_sum2f_1u(float& a, float& b, uint32_t& c) {
    asm("s_nop 1\n\t" BG_DPP3("quad_perm:[1,0,3,2]") BG_DPP3("quad_perm:[2,3,0,1]") BG_DPP3("row_half_mirror") BG_DPP3("row_mirror")
        : "+v"(a), "+v"(b), "+v"(c));
}

struct FrameMags {
    float m[16];  // |X[16l + e]|
    float nyq;    // |X[256]| (every lane)
};

// Raw sample rows of a lane group live in a ROTATING register window: row n1 of the frame with rotation R (= 4 x its
// position in the unrolled loop body, mod 16) is raw[(R + n1) & 15].  Consecutive frames share 384 of their 512 samples,
// so a frame only loads its four new rows over the four oldest ones -- no register moves.
template <int R>
__device__ __forceinline__ f2& row(f2 (&raw)[16], int n1) { return raw[(R + n1) & 15]; }

// all 16 rows of one frame: row n1 of lane l = (x[s + 32 n1 + 2l], x[s + 32 n1 + 2l + 1]); samples before the song start
// are 0 (the reference's zero-initialised sliding buffer); s is a multiple of 128, so pairs never straddle 0
__device__ __forceinline__ void fft512_load(__amdgpu_buffer_rsrc_t r_x, long rel_start, int l, f2 (&raw)[16]) {
    if (rel_start >= 0) {
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

// the lane's constants, register resident for the whole kernel (94 VGPRs): the window of its 16 sample pairs, its 15
// inter-pass twiddles, its 16 split twiddles
struct RegTables {
    f2 win[16];    // (hannz[2n], hannz[2n+1]), n = 16 n1 + l
    f2 tw256[16];  // W_256^(l*k1)
    f2 tw512[16];  // W_512^(16 l + e)
};

// 257 magnitudes of one frame: lane l of the group gets bins 16l..16l+15
template <int R>
__device__ __forceinline__ void fft512_compute(f2 (&raw)[16], int l, f2* tile, const RegTables& tabs, FrameMags& out) {
    f2 v[16];
    // Issue priority: low inside the two radix passes (long runs of arithmetic), high for everything else in the frame
    // -- the LDS exchanges, the split reads, the timbral epilogue and the PCM statistics are short bursts between waits,
    // and a wavefront in one of them should not queue behind the other wavefront's radix pass (FFT-512 kernel -2.5 %)
    __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int n1 = 0; n1 < 16; n1++) v[n1] = row<R>(raw, n1) * tabs.win[n1];
    radix16(v);  // over n1 -> A[k1] at v[R16(k1)]
#pragma unroll
    for (int k1 = 1; k1 < 16; k1++) v[R16(k1)] = cmul_pk(v[R16(k1)], tabs.tw256[k1]);  // W_256^(l*k1)
    __builtin_amdgcn_s_setprio(2);
#pragma unroll
    for (int k1 = 0; k1 < 16; k1++) tile[k1 * 17 + l] = v[R16(k1)];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int n2 = 0; n2 < 16; n2++) v[n2] = tile[l * 17 + n2];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_setprio(0);
    radix16(v);  // over n2 -> Z[l + 16 k2] at v[R16(k2)]
    __builtin_amdgcn_s_setprio(2);
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++) tile[k2 * 17 + l] = v[R16(k2)];  // Z[k] at k + (k >> 4)
    __builtin_amdgcn_wave_barrier();
    const f2 z0 = tile[0];
#if FFT512_PAIR_SPLIT
    // Z[k] and Z[256 - k] yield X[k] AND X[256 - k] (split_pair_sq: 10 instructions for two bins, the single form 7 for one).
    // The mirrors of lane l's bins 16 l + e, e = 1 .. 15, are lane 15 - l's bins 16 (15 - l) + 16 - e: every lane splits its
    // pairs e = 1 .. 8, keeps its own eight bins and hands the mirrored ones to its partner by one row_mirror DPP move each
    // (lane l <-> 15 - l of the 16-lane row) -- 9 splits and 18 LDS reads per lane and frame instead of 16 and 32, and only
    // nine split twiddles stay in registers.  Bin e = 0 (k = 16 l, mirror 16 (16 - l): another lane pair) keeps the single
    // form.  The bins e = 9 .. 15 now come out of the partner's pair as |A - P| instead of their own single split -- the
    // same bits: with Z[k] and Z[256 - k] exchanged and W_512^(256 - k) = -conj(W_512^k) (exact in the table: sin and cos
    // of mirrored angles are rounded from the same f64 values) every intermediate of the single form is the conjugate or
    // the negative of the pair form's, and IEEE operations are sign-symmetric.  Measured: rows bit-identical to round 3's
    // (kbench hashes), FFT-512 kernel 12.55 -> 11.88 ms per 1024 songs.
    {
        const f2 zk = tile[l * 17], zm = tile[(l == 0) ? 0 : 17 * (16 - l)];  // k = 0 pairs with itself
        out.m[0] = mag_from_sq(split_one_sq(zk, zm, tabs.tw512[0]));
    }
    float mir[8];
#pragma unroll
    for (int e = 1; e <= 8; e++) {
        const f2 zk = tile[l * 17 + e], zm = tile[17 * (15 - l) + 16 - e];
        float sq_k, sq_m;
        split_pair_sq(zk, zm, tabs.tw512[e], sq_k, sq_m);  // W_512^k, k = 16 l + e
        out.m[e] = mag_from_sq(sq_k);
        if (e < 8) mir[e] = mag_from_sq(sq_m);                // |X[256 - k]| = the partner's bin 16 - e
    }
#pragma unroll
    for (int e = 9; e < 16; e++) out.m[e] = dpp_mov<DPP_ROW_MIRROR>(mir[16 - e]);
#else
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const f2 zk = tile[l * 17 + e];
        // Z[256 - k], k = 16 l + e  (k = 0 pairs with itself)
        const int mi = (e == 0) ? ((l == 0) ? 0 : 17 * (16 - l)) : (17 * (15 - l) + 16 - e);
        const f2 zm = tile[mi];
        out.m[e] = mag_from_sq(split_one_sq(zk, zm, tabs.tw512[e]));  // W_512^k, k = 16 l + e
    }
#endif
    // Z is halved (half window): X[0] = 2 (Re Z[0] + Im Z[0]), X[256] = 2 (Re Z[0] - Im Z[0])
    if (l == 0) out.m[0] = 2.0f * fabsf(z0.x + z0.y);
    out.nyq = 2.0f * fabsf(z0.x - z0.y);
    __builtin_amdgcn_wave_barrier();
}

// ---- statistics of the PCM itself, folded into this kernel because it is the one that already holds every sample in
// registers: per 256-sample block the sum of squares (LoudnessDesc, src/misc.rs:12-18,46-65, and the tempo silence test,
// src/aubio.rs:1258-1276) and the zero-crossing count (number_crossings, src/utils.rs:81-95: a crossing is a change of
// `x > 0` between consecutive samples; the first sample of the song compares with itself).  FFT frame k brings in the
// 128 samples [128 k, 128 k + 128) = rows 12..15 of its window; `before` is row 11, whose last sample precedes them. ----
// acc += bit `lane` of mask: one VALU instruction (the 64-bit lane mask is the carry-in of an add-with-carry)
__device__ __forceinline__ void add_lane_bit(uint32_t& acc, uint64_t mask) {
    asm("v_addc_co_u32_e64 %0, vcc, 0, %0, %1" : "+v"(acc) : "s"(mask) : "vcc");
}

// The sign tests produce 64-bit lane masks (v_cmp writes an SGPR pair); comparing a sample with its predecessor is then
// SCALAR bit arithmetic on those masks -- the idle scalar unit instead of the busy vector one: within a row the samples
// run x(lane 0), y(lane 0), x(lane 1), ..., so "y against its x" is Mx ^ My and "x against the y before it" is
// Mx ^ (My << 1) with bit 0 of every 16-lane group replaced by lane 15 of the row before (or, at the very first sample
// of the song, by the sample itself: no crossing).  A lane then adds its own bit of each mask.
__device__ __forceinline__ void stats128(const f2 r0, const f2 r1, const f2 r2, const f2 r3, const f2 before, bool song_start,
                                         float& ss, uint32_t& zc) {
    ss += ((r0.x * r0.x + r0.y * r0.y) + (r1.x * r1.x + r1.y * r1.y)) + ((r2.x * r2.x + r2.y * r2.y) + (r3.x * r3.x + r3.y * r3.y));
    constexpr uint64_t LOW = 0x0001000100010001ull;  // lane 0 of the four 16-lane groups
    const uint64_t x0 = __ballot(r0.x > 0.0f), y0 = __ballot(r0.y > 0.0f), x1 = __ballot(r1.x > 0.0f), y1 = __ballot(r1.y > 0.0f);
    const uint64_t x2 = __ballot(r2.x > 0.0f), y2 = __ballot(r2.y > 0.0f), x3 = __ballot(r3.x > 0.0f), y3 = __ballot(r3.y > 0.0f);
    const uint64_t yb = __ballot(before.y > 0.0f), start = __builtin_amdgcn_ballot_w64(song_start);
    auto shifted = [&](uint64_t y, uint64_t prev_bits) { return ((y << 1) & ~LOW) | prev_bits; };
    const uint64_t p0 = (((yb >> 15) & ~start) | (x0 & start)) & LOW;
    add_lane_bit(zc, x0 ^ y0);
    add_lane_bit(zc, x0 ^ shifted(y0, p0));
    add_lane_bit(zc, x1 ^ y1);
    add_lane_bit(zc, x1 ^ shifted(y1, (y0 >> 15) & LOW));
    add_lane_bit(zc, x2 ^ y2);
    add_lane_bit(zc, x2 ^ shifted(y2, (y1 >> 15) & LOW));
    add_lane_bit(zc, x3 ^ y3);
    add_lane_bit(zc, x3 ^ shifted(y3, (y2 >> 15) & LOW));
}

// ROLLOFF_EXACT_ALL: tests only -- every frame takes the reference-order pass.  SEQ_FLUX: BLISSGPU_OPT_FLUX_ORDER.
template <bool ROLLOFF_EXACT_ALL, bool SEQ_FLUX>
__global__ __launch_bounds__(256, 2) void fft512_kernel(const float* __restrict__ pcm,
                                                        const SongDesc* __restrict__ songs, uint32_t n_songs,
                                                        const uint32_t* __restrict__ pfx_f,
                                                        const float* __restrict__ hannz,
                                                        const float2* __restrict__ tw512,
                                                        float* __restrict__ centroid, float* __restrict__ rolloff,
                                                        float* __restrict__ flatness, float* __restrict__ flux,
                                                        float* __restrict__ e256, uint32_t* __restrict__ zc256,
                                                        const RollFix* __restrict__ fix) {
    __shared__ f2 lds[GROUPS_PER_WG * GRP_PITCH];
    // magnitudes of every group's FIRST tempo frame, for the group before it (see `share` below)
    __shared__ float halo[GROUPS_PER_WG][16 * 16 + 1];
    RegTables tabs;
    {
        const int l0 = threadIdx.x & 15;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const float2 a = tw512[2 * ((i * l0) & 255)], b = tw512[16 * l0 + i];
            tabs.win[i] = mk(hannz[2 * (16 * i + l0)], hannz[2 * (16 * i + l0) + 1]);
            tabs.tw256[i] = mk(a.x, a.y);  // W_256^(k1*l) = W_512^(2 k1 l)
            tabs.tw512[i] = mk(b.x, b.y);
        }
    }
    const uint32_t s = find_segment(pfx_f, n_songs, blockIdx.x);
    const SongDesc sd = songs[s];
    const uint32_t tile_idx = blockIdx.x - pfx_f[s];
    const int grp = threadIdx.x >> 4, l = threadIdx.x & 15;
    f2* tile = lds + grp * GRP_PITCH;

    const long k_begin = (long)tile_idx * F512_TILE + (long)grp * FRAMES_PER_GROUP;  // a multiple of 32
    const long k_end = (k_begin + FRAMES_PER_GROUP < (long)sd.n_f) ? k_begin + FRAMES_PER_GROUP : (long)sd.n_f;

    // descriptor over this workgroup's slice of the song (keeps lane offsets 32-bit for any song length);
    // offsets before the song start wrap to huge values and read 0 = the reference's zero history
    const long tile_first = (long)tile_idx * F512_TILE * HOP_T - (W512 - HOP_T) - HOP_T;  // sample of frame (tile*T - 1)
    const long base = tile_first > 0 ? tile_first : 0;
    const uint64_t avail = (sd.n - (uint64_t)base) * 4;
    const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(pcm + sd.pcm_off + base), 0, (uint32_t)(avail < 0xFFFFFFFFull ? avail : 0xFFFFFFFFull), 0x00020000);

    // Two magnitude sets alternate so that "the previous tempo frame" is never copied: over four frames k0 .. k0 + 3
    // (k0 a multiple of 4) the previous odd frame lives in A, frames k0 and k0 + 1 are computed into B (the flux of
    // k0 + 1 is B against A), frames k0 + 2 and k0 + 3 into A (flux: A against B).
    FrameMags A, B;
    const bool active = k_begin < (long)sd.n_f;
    // Inside the frame loop a frame is named by its position r in the group (k = k_begin + r, r = 0 .. 31: the loop counter, the
    // same in every lane) and every bound is a lane constant relative to k_begin: the 64-bit frame index cost an add and a
    // 64-bit compare per test, five of each per frame.
    const int len = active ? (int)(k_end - k_begin) : 0;                                                      // frames of this group
    const long left_t = (long)sd.n_t - k_begin, left_b = (long)sd.n_b - (k_begin >> 1);
    const int lim_t = left_t < 0 ? 0 : (left_t > FRAMES_PER_GROUP ? FRAMES_PER_GROUP : (int)left_t);          // r < lim_t <=> k < n_t
    const int lim_b = left_b < 0 ? 0 : (left_b > FRAMES_PER_GROUP ? FRAMES_PER_GROUP : (int)left_b);          // (r - 1) / 2 < lim_b <=> q < n_b
    const bool at_song_start = k_begin == 0;
    // lane offset of rows 12..15 of frame k_begin + 1 (r = 0 loads them for r + 1); frame r adds 512 r bytes as a scalar offset
    const uint32_t xoff0 = (uint32_t)(((k_begin + 2) * HOP_T - HOP_T - base + 2 * l) * 4);
    f2 raw[16];
    // The flux of a group's first tempo frame (FFT frame k_begin + 1) needs the magnitudes of the tempo frame before it,
    // FFT frame k_begin - 1 -- the last frame of the group before.  Only the first group of a workgroup transforms that
    // halo frame itself (33 transforms for 32 frames); the other fifteen publish the magnitudes of their first tempo
    // frame in LDS and leave that one flux value to their predecessor, which has frame k_begin - 1 in registers when its
    // loop ends.  Same operands, same order of additions: the series is bit-identical.
    const bool share = active && grp > 0;
    // halo: magnitudes of the previous tempo frame (FFT frame k_begin - 1); zeros before the song starts
    if (active && k_begin >= 1 && !share) {
        fft512_load(r_x, k_begin * HOP_T - W512 - base, l, raw);
        fft512_compute<0>(raw, l, tile, tabs, A);
    } else {
#pragma unroll
        for (int e = 0; e < 16; e++) A.m[e] = 0.0f;
        A.nyq = 0.0f;
    }
    if (active) fft512_load(r_x, (k_begin + 1) * HOP_T - W512 - base, l, raw);

    float ss_acc = 0.0f;    // per-lane partial sums of the current 256-sample block
    uint32_t zc_acc = 0;
    // reduced sums of the timbral frame this lane will finish (see finish16)
    float st_total = 0.0f, st_wtotal = 0.0f, st_mant = 1.0f;
    int st_ints = 0;
    const float freq_per_bin = (float)SAMPLE_RATE / (float)W512;

    // one frame of the unrolled body: J = position in the body (k = kb + J), CUR / PREV = the magnitude sets as above
    auto frame = [&](auto jc, int r_lane, FrameMags& cur, FrameMags& prev) {
        constexpr int J = decltype(jc)::value;
        // (the counter of a loop whose exit diverges lives in a vector register although every lane holds the same value:
        // taken as a scalar, the tests against it, `r & 15` and the row loads' scalar offset are scalar operands)
        const int r = __builtin_amdgcn_readfirstlane(r_lane);
        constexpr int R = (4 * J) & 15;
        stats128(row<R>(raw, 12), row<R>(raw, 13), row<R>(raw, 14), row<R>(raw, 15), row<R>(raw, 11), at_song_start && r == 0, ss_acc, zc_acc);
        __builtin_amdgcn_sched_barrier(0);
        fft512_compute<R>(raw, l, tile, tabs, cur);
        if (r + 1 < len) {
            // rows 12..15 of frame k + 1 = samples [(k + 1) * 128, (k + 2) * 128) replace this frame's rows 0..3; issued a
            // whole frame ahead, so their latency is off the critical path.  (Never before the song's first sample, so nothing
            // here relies on the descriptor's range check, which does not see the scalar offset.)
            const uint32_t soff = (uint32_t)r * (uint32_t)(HOP_T * 4);
#pragma unroll
            for (int n1 = 0; n1 < 4; n1++) row<R>(raw, n1) = buf_load_f2(r_x, xoff0, soff + 128u * n1);
        }
        if (J & 1) {
            // 256-sample block q = (k - 1) / 2 is complete: reduce over the 16 lanes, lane 0 writes
            // (block energy, crossings and -- in the default order -- the flux: one interleaved block of three row reductions)
            float ss = ss_acc;
            uint32_t zc = zc_acc;
            const int qr = (r - 1) >> 1;  // q = k_begin / 2 + qr
            const long q = (k_begin >> 1) + qr;
            ss_acc = 0.0f;
            zc_acc = 0;
            // tempo frame j = (k-1)/2 : SpecFlux over bins 0..256 (src/aubio.rs:455-467)
            // `if (cur > prev) f += cur - prev` as f += max(cur - prev, 0): the same sum bit for bit (a positive difference
            // of two floats is never rounded to zero, adding +0 to a non-negative f changes nothing, and max drops a NaN
            // difference exactly as the comparison does) in three instructions per bin instead of four
            float f = 0.0f;
            if constexpr (SEQ_FLUX) {
            // BLISSGPU_OPT_FLUX_ORDER = 1: the 257 terms added one by one in bin order, as src/aubio.rs:455-467 does -- lane l
            // continues lane l - 1's running sum (16 rounds of 16 dependent adds).  The 16-per-lane + tree order of the default
            // deviates from that by 1.7e-7 rms even on exact magnitudes, which is what put the device's tempo on the noisier
            // side of the f32 floor (profiles/r05_flux_order_ab6.txt); this form costs the kernel 19 %
#pragma unroll 1
            for (int step = 0; step < 16; step++) {
                const float carry = dpp_mov<0x111>(f);  // row_shr:1 -- lane l receives lane l - 1's running sum (lane 0: 0)
                if (l == step) {
                    f = carry;
#pragma unroll
                    for (int e = 0; e < 16; e++) f += fmaxf(cur.m[e] - prev.m[e], 0.0f);
                }
            }
            f = __shfl(f, (threadIdx.x & 48) + 15, 64) + fmaxf(cur.nyq - prev.nyq, 0.0f);  // lane 15's total, then bin 256
            ss = row16_sum(ss);
            zc = (uint32_t)row16_sum((int)zc);
            } else {
            // (the differences two bins per packed subtraction; the sum in the same order as bin by bin)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const f2 d = mk(cur.m[2 * i], cur.m[2 * i + 1]) - mk(prev.m[2 * i], prev.m[2 * i + 1]);
                f += fmaxf(d.x, 0.0f);
                f += fmaxf(d.y, 0.0f);
            }
            if (l == 0) f += fmaxf(cur.nyq - prev.nyq, 0.0f);
            row16_sum2f_1u(ss, f, zc);
            }
            if (l == 0) { e256[sd.e_off + q] = ss; zc256[sd.e_off + q] = zc; }
            if (J == 1 && share && r == 1) {
                // no previous tempo frame here: hand this frame's magnitudes to the group before
#pragma unroll
                for (int e = 0; e < 16; e++) halo[grp][16 * e + l] = cur.m[e];
                if (l == 0) halo[grp][256] = cur.nyq;
            } else if (l == 0 && qr < lim_b) {
                flux[sd.b_off + q] = f;
            }
        }
        if (r < lim_t) {
            // the 256-bin vector the reference's timbral path sees: bin 255 := |Re X[256]| (src/aubio.rs:240-261)
            const float m15 = (l == 15) ? cur.nyq : cur.m[15];
            // sum m, sum e m (e = bin index inside the lane) and sum m^2, two bins per packed instruction
            f2 s2 = mk(0.0f, 0.0f), e2 = mk(0.0f, 0.0f), q2 = mk(0.0f, 0.0f);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const f2 mp = mk(cur.m[2 * i], i == 7 ? m15 : cur.m[2 * i + 1]);
                s2 = s2 + mp;
                e2 = e2 + mp * mk((float)(2 * i), (float)(2 * i + 1));
                q2 = q2 + mp * mp;
            }
            // (the three horizontal adds as opaque instructions: left to the SLP vectoriser two of them become ONE packed add behind
            // three register moves that line the halves up)
            auto hadd = [](f2 v) __attribute__((always_inline)) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(v.x), "v"(v.y)); return r; };
            const float sum = hadd(s2), sqsum = hadd(q2);
            const float wsum = (float)(16 * l) * sum + hadd(e2);  // sum (16 l + e) m_e
            // the frame's four float reductions in one interleaved block (row16_sum3_scan1): total, weighted total, and -- for the
            // rolloff below -- the inclusive scan and the total of the lanes' energies
            float total, wtotal, incl, cum_total;
            row16_sum3_scan1(sum, wsum, sqsum, sqsum, total, wtotal, incl, cum_total);
            // spectral_rolloff (src/aubio.rs:36-58): bins consumed until the running energy reaches 95 %.  The reference
            // adds the 256 squares one by one in f32; here a lane scan supplies the energy below the lane's first bin and
            // the lane walks its 16 bins.  A bin COUNT is discontinuous in those sums, so the two orders must not be
            // allowed to disagree: the walk is carried as d = (running energy) - threshold, and a frame in which some |d|
            // comes within ROLL_GUARD x total of zero -- closer than the worst-case rounding of both orders together:
            // 256 u + 257 u for the sequential sums of total and running energy, < 100 u for the scan, the tree, the fused
            // squares and the walk (u = 2^-24) -- hands its 256 magnitudes to rolloff_fix_kernel, which repeats the
            // reference's loop literally (about 2 % of white-noise frames; every other frame's count is provably the
            // sequential one).
            const float thr = cum_total * 0.95f;
            float d = (incl - sqsum) - thr;
            float near;
            uint32_t signs = 0;  // sign bit of d after each bin, shifted in: d < 0 <=> that bin is still below the threshold
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const float m0 = cur.m[2 * i], m1 = i == 7 ? m15 : cur.m[2 * i + 1];
                const float d0 = fmaf(m0, m0, d);
                d = fmaf(m1, m1, d0);
                signs = __builtin_amdgcn_alignbit(signs, __float_as_uint(d0), 31);  // (signs << 1) | (d0 < 0)
                signs = __builtin_amdgcn_alignbit(signs, __float_as_uint(d), 31);
                if (i == 0) asm("v_min_f32 %0, |%1|, |%2|" : "=v"(near) : "v"(d0), "v"(d));  // (no FLT_MAX to load first)
                else asm("v_min3_f32 %0, %0, |%1|, |%2|" : "+v"(near) : "v"(d0), "v"(d));
            }
            // not (near > guard): also catches NaN; totals in the denormal range have no relative bound; a frame of digital
            // silence is 0 in either order
            const bool risky = ROLLOFF_EXACT_ALL || (cum_total != 0.0f && (!(near > cum_total * ROLL_GUARD) || cum_total < 1e-30f));
            // geometric_mean (src/utils.rs:101-117): groups of 8 in f64, exponents and mantissas apart.  (f64 multiplies and
            // conversions issue at the f32 rate on this chip; the same split done on the f32 bit patterns -- shift, and-or,
            // f32 product -- needs MORE instructions: +59 over the kernel, measured by count in round 4 and dropped.)
            int expo = 0, zero = 0;
            double mant = 1.0;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const float c7 = h == 1 ? m15 : cur.m[7];
                const float* c8 = cur.m + 8 * h;
                double g = ((double)c8[0] * (double)c8[1]) * ((double)c8[2] * (double)c8[3]);
                g *= 3.273390607896142e150;
                g *= ((double)c8[4] * (double)c8[5]) * ((double)c8[6] * (double)c7);
                if (g == 0.0) zero = 1;
                const uint64_t bits = (uint64_t)__double_as_longlong(g);
                expo += (int)(bits >> 52);
                mant *= __longlong_as_double((long long)((bits & 0xFFFFFFFFFFFFFull) | 0x3FF0000000000000ull));
            }
            // one reduction for the three integers: rolloff count (<= 256: 9 bits) | lanes with a zero group (<= 16: 5 bits) |
            // biased f64 exponents (two per lane, <= 16 x 4094: 16 bits)
            const int red = row16_sum((int)((uint32_t)__popc(signs) | ((uint32_t)zero << 9) | ((uint32_t)expo << 14)));
            const int c = red & 511, any_zero = (red >> 9) & 31, exps = (int)((uint32_t)red >> 14);
            mant = row16_prod(mant);
            // The scalar tail (two divisions, log2, exp2, the normalisations: ~40 instructions) would run identically on all
            // 16 lanes of the group; instead lane k mod 16 keeps the reduced sums of frame k and every 16 frames each lane
            // finishes ITS frame -- one pass of the scalar tail serves 16 frames, and the three stores become coalesced.
            const bool mine = l == (r & 15);  // (k_begin is a multiple of 32)
            // an unproven frame (any of the group's 16 lanes within the guard) is marked in the packed integers: its rolloff
            // leaves this kernel as ROLLOFF_UNPROVEN and rolloff_fix_kernel finds it there
            const uint64_t risky_lanes = __builtin_amdgcn_ballot_w64(risky);  // (the builtin: __ballot of a combined condition is lowered to v_cndmask + v_cmp)
            const uint32_t grp_mask = (uint32_t)(risky_lanes >> (threadIdx.x & 48)) & 0xFFFFu;
            // packed integers: rolloff count (9 bits) | any zero group (bit 9) | zero energy (bit 10) | exponent sum (16 bits
            // from bit 11) | unproven (bit 27)
            const int packed = c | (any_zero ? 1 << 9 : 0) | (cum_total == 0.0f ? 1 << 10 : 0) | (exps << 11) | (grp_mask != 0 ? 1 << 27 : 0);
            st_total = mine ? total : st_total;
            st_wtotal = mine ? wtotal : st_wtotal;
            st_ints = mine ? packed : st_ints;
            st_mant = mine ? (float)mant : st_mant;
            // (at the very end of the frame: the branch splits the block the scheduler works on)
            if (risky_lanes != 0) {  // wave-uniform; rare
                if (grp_mask != 0) {  // this frame goes to the exact pass
                    // Round 6: the frame's 256 magnitudes go to the entry of its OWN index in the borrowed stretch (room for
                    // every frame of the chunk: chunk_front proves it) -- no slot to reserve, so no round trip to a global
                    // counter in the middle of the frame loop (the returning atomic held the wavefront for ~2 us whenever one
                    // of its four frames was unproven: 8 % of the trips on white noise, 15 % on music, where this kernel
                    // was 11 % slower than on noise).
                    // (the stretch is named by a record in memory, read only here: a kernel argument would live in SGPRs
                    // through the whole frame loop, which has none to spare)
                    float4* dst = reinterpret_cast<float4*>(fix->mags + (size_t)(sd.t_off + (uint64_t)(k_begin + r)) * 256 + 16 * l);
                    dst[0] = make_float4(cur.m[0], cur.m[1], cur.m[2], cur.m[3]);
                    dst[1] = make_float4(cur.m[4], cur.m[5], cur.m[6], cur.m[7]);
                    dst[2] = make_float4(cur.m[8], cur.m[9], cur.m[10], cur.m[11]);
                    dst[3] = make_float4(cur.m[12], cur.m[13], cur.m[14], m15);
                }
            }
        }
    };
    // finish the frames [k16, k16 + 16) whose sums the lanes hold (lane l: frame k16 + l)
    auto finish16 = [&](int r16) {
        const int r = r16 + l;
        const long k = k_begin + r;
        if (r < len && r < lim_t) {
            // spectral_centroid (src/aubio.rs:16-29) then bin_to_freq (:68-71)
            const float cbin = (st_total == 0.0f) ? 0.0f : st_wtotal / st_total;
            const int st_c = st_ints & 511, st_exps = (st_ints >> 11) & 0xFFFF;
            const float rbin = (st_ints & (1 << 10)) ? 0.0f : (float)((st_c < 256) ? st_c + 1 : 256);
            float flat = 0.0f;
            if (!(st_ints & (1 << 9))) {
                const float geo = exp2f((log2f(st_mant) + (float)st_exps) / 256.0f - (1023.0f + 500.0f) / 8.0f);
                if (geo != 0.0f) flat = geo / (st_total / 256.0f);
            }
            centroid[sd.t_off + k] = freq_per_bin * fmaxf(cbin, 0.0f);
            rolloff[sd.t_off + k] = (st_ints & (1 << 27)) ? ROLLOFF_UNPROVEN : freq_per_bin * fmaxf(rbin, 0.0f);
            flatness[sd.t_off + k] = flat;
        }
    };

    for (int rb = 0; rb < len; rb += 4) {
        frame(std::integral_constant<int, 0>{}, rb, B, A);
        __builtin_amdgcn_sched_barrier(0);  // keep the frames apart: interleaving two of them costs more registers than it hides
        if (rb + 1 < len) frame(std::integral_constant<int, 1>{}, rb + 1, B, A);
        __builtin_amdgcn_sched_barrier(0);
        if (rb + 2 < len) frame(std::integral_constant<int, 2>{}, rb + 2, A, B);
        __builtin_amdgcn_sched_barrier(0);
        if (rb + 3 < len) frame(std::integral_constant<int, 3>{}, rb + 3, A, B);
        __builtin_amdgcn_sched_barrier(0);
        if ((rb & 15) == 12 || rb + 4 >= len) finish16(rb & ~15);  // 16 frames stashed, or the group's last frames
    }

    // ---- the flux of the NEXT group's first tempo frame (FFT frame k_begin + 33) against this group's last one (k_begin +
    // 31, in set A when the loop has run its 32 frames) ----
    __syncthreads();
    if (grp + 1 < GROUPS_PER_WG && k_begin + FRAMES_PER_GROUP + 1 < (long)sd.n_f) {
        const long q = (k_begin + FRAMES_PER_GROUP) >> 1;
        float f = 0.0f;
        const float cn = halo[grp + 1][256];
        if constexpr (SEQ_FLUX) {
        float hc[16];
#pragma unroll
        for (int e = 0; e < 16; e++) hc[e] = halo[grp + 1][16 * e + l];
#pragma unroll 1
        for (int step = 0; step < 16; step++) {
            const float carry = dpp_mov<0x111>(f);
            if (l == step) {
                f = carry;
#pragma unroll
                for (int e = 0; e < 16; e++) f += fmaxf(hc[e] - A.m[e], 0.0f);
            }
        }
        f = __shfl(f, (threadIdx.x & 48) + 15, 64) + fmaxf(cn - A.nyq, 0.0f);
        } else {
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const float c = halo[grp + 1][16 * e + l];
            f += fmaxf(c - A.m[e], 0.0f);
        }
        if (l == 0) f += fmaxf(cn - A.nyq, 0.0f);
        f = row16_sum(f);
        }
        if (l == 0 && q < (long)sd.n_b) flux[sd.b_off + q] = f;
    }

    // ---- the samples no FFT frame brings in: [128 n_f, n), 256 .. 511 of them, handled by the group that owns the
    // song's last frame.  If n_f is odd the first of these blocks already holds the partial sums of frame n_f - 1. ----
    if (active && k_end == (long)sd.n_f) {
        const float* __restrict__ x = pcm + sd.pcm_off;
        const uint64_t t0 = (uint64_t)sd.n_f * HOP_T;
        for (uint64_t q = t0 >> 8; q < (uint64_t)sd.n_e; q++) {
            const uint64_t lo = (q << 8) > t0 ? (q << 8) : t0;
            const uint64_t hi = ((q + 1) << 8) < sd.n ? ((q + 1) << 8) : sd.n;
            for (uint64_t i = lo + (uint64_t)l; i < hi; i += 16) {
                const float v = x[i], pv = x[i - 1];  // i >= 128: a predecessor always exists
                ss_acc += v * v;
                zc_acc += ((v > 0.0f) != (pv > 0.0f)) ? 1u : 0u;
            }
            const float ss = row16_sum(ss_acc);
            const uint32_t zc = (uint32_t)row16_sum((int)zc_acc);
            if (l == 0) { e256[sd.e_off + q] = ss; zc256[sd.e_off + q] = zc; }
            ss_acc = 0.0f;
            zc_acc = 0;
        }
    }
}

void launch_fft512(const Batch& b, const Workspace& w, const DeviceTables& t, hipStream_t st, bool rolloff_exact_all, bool seq_flux) {
    if (b.tiles_f == 0) return;
    auto k = fft512_kernel<false, false>;
    if (rolloff_exact_all && seq_flux) k = fft512_kernel<true, true>;
    else if (rolloff_exact_all) k = fft512_kernel<true, false>;
    else if (seq_flux) k = fft512_kernel<false, true>;
    hipLaunchKernelGGL(k, dim3(b.tiles_f), dim3(256), 0, st, b.pcm, b.songs, b.n_songs, b.pfx_f, t.hannz512,
                       t.tw512, w.centroid, w.rolloff, w.flatness, w.flux, w.e256, w.zc256, w.roll_fix);
}

}  // namespace bg
