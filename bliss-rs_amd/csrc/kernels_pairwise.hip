// kernels_pairwise.hip -- all-pairs feature-vector distances (compiled with -ffp-contract=off).
//
// Reference (src/playlist.rs):
//   euclidean_distance   :65-71   (a-b).dot(eye).dot(a-b).sqrt()
//   cosine_distance      :76-79   1 - a.b / (sqrt(a.a) * sqrt(b.b))
//   mahalanobis_distance :140-142 (a-b).dot(M).dot(a-b).sqrt()        (default metric: lib.rs:168-178)
// The f32 results are pinned by `assert_eq` literals (playlist.rs:1023,1088,1104; lib.rs:280,289), the
// last of which (3.4999998) depends on ndarray's summation order, so the kernel reproduces it:
//   * Array1.dot(Array2) walks each (strided) column with a plain sequential sum;
//   * Array1.dot(Array1) is `unrolled_dot`: 8 partial sums, combined (p0+p4)+(p1+p5)+(p2+p6)+(p3+p7),
//     then the <8 tail elements sequentially.
// With M = I (or any diagonal M) the vector-matrix product degenerates exactly to an element-wise
// product, which is the fast path.
//
// 4 bytes per pair out, 4*d*(n+m) bytes in.  A workgroup owns 256 columns -- each lane keeps 4 column vectors in
// registers -- and walks PW_RT tiles of 128 rows down them (round 6: the columns are 92 strided loads per lane and used
// to be reloaded for every tile); rows come from LDS as broadcasts, and every wave-store is 64 lanes x 16 B = 1 KiB of
// one output row.  Bound by the vector ALU, not by its stores (profiles/r06_pairwise_ablation.txt: the self-distance
// kernel takes 9.7 of its 10.7 ms with every store removed): 136 packed operations + four correctly rounded square
// roots (a quarter-rate instruction each) per four outputs is what the reference's summation order costs.
#include <stdlib.h>

#include <type_traits>

#include "device_utils.hpp"
#include "internal.hpp"

// store flavours of the two halves of a self-distance block (1 = non-temporal, the shipped form; 0 = plain: measured in
// profiles/r06_pairwise_staging_ab.txt)
#ifndef PW_NT_DIRECT
#define PW_NT_DIRECT 1
#endif
#ifndef PW_NT_MIRROR
#define PW_NT_MIRROR 1
#endif
#define PW_STORE4(NT, value, ptr) do { if (NT) __builtin_nontemporal_store((value), (ptr)); else *(ptr) = (value); } while (0)

namespace bg {

constexpr int PW_ROWS = 128, PW_COLS = 256, PW_CPT = 4;  // tile rows, tile cols, cols per thread
#ifndef PW_RT
#define PW_RT 8   // 128-row tiles a workgroup walks with its 256 columns in registers (even: the self-distance kernel works in 256 x 256 blocks)
#endif
enum { METRIC_EUCLIDEAN = 0, METRIC_COSINE = 1, METRIC_MAHALANOBIS = 2 };

typedef float f2 __attribute__((ext_vector_type(2)));

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N-1>{})
template <int N, typename F, int I = 0>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, F, I + 1>(static_cast<F&&>(f));
    }
}
__device__ __forceinline__ f2 splat(float x) { f2 r; r.x = x; r.y = x; return r; }
// an index every lane of the wavefront computed alike, moved to scalar registers: with a 32-bit lane offset on top the store
// takes the `saddr + voffset` form and no lane keeps a 64-bit address (two VGPRs each) across the arithmetic
__device__ __forceinline__ uint64_t wave_uniform(uint64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// Correctly rounded sqrt (what Rust's f32::sqrt / the oracle's sqrtf return) for the two halves of a pair.
// v_sqrt_f32 is within 1 ulp; the exact residuals against the two neighbours (one FMA each) pick the correctly
// rounded value -- the compiler's own sqrtf expansion minus its input scaling for x < 2^-96, which v_sqrt_f32
// needs because it does not take denormal inputs.  The rare tiny inputs (and only those) use sqrtf; 0, inf and
// NaN come out right (NaN residuals fail both comparisons).
__device__ __forceinline__ float sqrt_rn_fast(float x) {
    const float r = __builtin_amdgcn_sqrtf(x);
    const float rd = __uint_as_float(__float_as_uint(r) - 1u), ru = __uint_as_float(__float_as_uint(r) + 1u);
    const float vp = __builtin_fmaf(-rd, r, x), vs = __builtin_fmaf(-ru, r, x);
    float o = (vp <= 0.0f) ? rd : r;
    o = (vs > 0.0f) ? ru : o;
    return o;
}
__device__ __forceinline__ f2 sqrt_rn2(f2 x) {
    f2 r;
    const bool tiny = (x.x < 0x1p-96f && x.x != 0.0f) || (x.y < 0x1p-96f && x.y != 0.0f);
    if (__builtin_expect(__ballot(tiny) != 0ull, 0)) {  // wave-uniform
        r.x = sqrtf(x.x);
        r.y = sqrtf(x.y);
    } else {
        r.x = sqrt_rn_fast(x.x);
        r.y = sqrt_rn_fast(x.y);
    }
    return r;
}

// (a, a) - b and (a, a) * b where a is the LO (HI = false) or HI half of a register pair: the broadcast
// is an op_sel modifier of the packed instruction instead of two v_mov per element
template <bool HI>
__device__ __forceinline__ f2 bsub(f2 apair, f2 b) {
    f2 r;
    if (HI) asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(apair), "v"(b));
    else asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(apair), "v"(b));
    return r;
}
template <bool HI>
__device__ __forceinline__ f2 bmul(f2 apair, f2 b) {
    f2 r;
    if (HI) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(apair), "v"(b));
    else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(apair), "v"(b));
    return r;
}

// ndarray::numeric_util::unrolled_dot over compile-time length D, evaluated for TWO pairs at once in packed
// f32 (v_pk_mul_f32 / v_pk_add_f32: each half rounds exactly like the scalar op; this translation unit is
// compiled with -ffp-contract=off so the multiply and the add stay separate, as in the reference).
template <int D, typename FX, typename FY>
__device__ __forceinline__ f2 unrolled_dot2(FX xs, FY ys) {
    f2 p[8];
#pragma unroll
    for (int u = 0; u < 8; u++) p[u] = splat(0.0f);
    constexpr int BODY = (D / 8) * 8;
#pragma unroll
    for (int k = 0; k < BODY; k++) p[k & 7] = p[k & 7] + xs(k) * ys(k);
    f2 sum = splat(0.0f);
    sum = sum + (p[0] + p[4]);
    sum = sum + (p[1] + p[5]);
    sum = sum + (p[2] + p[6]);
    sum = sum + (p[3] + p[7]);
#pragma unroll
    for (int k = BODY; k < D; k++) sum = sum + xs(k) * ys(k);
    return sum;
}

template <int D, typename FX, typename FY>
__device__ __forceinline__ float unrolled_dot(FX xs, FY ys) {
    float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    constexpr int BODY = (D / 8) * 8;
#pragma unroll
    for (int k = 0; k < BODY; k++) p[k & 7] = p[k & 7] + xs(k) * ys(k);
    float sum = 0.0f;
    sum = sum + (p[0] + p[4]);
    sum = sum + (p[1] + p[5]);
    sum = sum + (p[2] + p[6]);
    sum = sum + (p[3] + p[7]);
#pragma unroll
    for (int k = BODY; k < D; k++) sum = sum + xs(k) * ys(k);
    return sum;
}

// SYM: A and B are the same matrix.  Every metric here is symmetric bit for bit (negating a - b leaves its squares, the
// quadratic form and the dot products unchanged, and both norm tables come from the same rows), so only the 256 x 256
// blocks on or above the diagonal are computed; an off-diagonal block is also written transposed, 32 rows at a time
// through an LDS tile so that the transposed stores are 128-byte runs.
template <int D, int METRIC, bool DIAG, bool SYM = false>
__global__ __launch_bounds__(256, (METRIC == METRIC_EUCLIDEAN ? 3 : 2)) void pairwise_kernel(const float* __restrict__ A, uint64_t n,
                                                       const float* __restrict__ B, uint64_t m,
                                                       const float* __restrict__ M, float* __restrict__ out,
                                                       uint64_t ld_out) {
    constexpr int DP = (D + 3) & ~3;  // LDS row pitch: 16-byte aligned rows -> ds_read_b128 broadcasts
    __shared__ __attribute__((aligned(16))) float sa[PW_ROWS][DP];
    __shared__ float sna[PW_ROWS];
    __shared__ float sm[(METRIC == METRIC_MAHALANOBIS) ? D * D : 1];
#ifndef PW_T_ROWS
#define PW_T_ROWS 32
#endif
    constexpr int T_ROWS = PW_T_ROWS;  // rows staged per transposed write
    // (Rounds 3 - 5 staged the tile transposed, st[column][row] with an odd pitch: four lanes to a bank on the storing side, 38 %
    // of the kernel's LDS cycles.  Three conflict-free forms of THAT tile lost 10 % each -- every one of them paid for its
    // addresses with vector-ALU instructions, which is what this kernel is short of.  The row-major tile below needs none.)
#ifndef PW_ST_ROWMAJOR
#define PW_ST_ROWMAJOR 1   // (0: the transposed staging of rounds 3 - 5, profiles/r06_pairwise_staging_ab.txt)
#endif
#ifndef PW_T_UNROLLED
#define PW_T_UNROLLED 1
#endif
#if PW_ST_ROWMAJOR
    // staged as computed, st[row][column]: a lane's four outputs of a row are 16 consecutive bytes (two 8-byte LDS stores, no
    // four-lanes-to-a-bank conflicts); the transposition happens on the reading side.  Pitch 258 words: row r starts at bank
    // 2 r, so the 8 x 8 lanes of a wavefront that assemble eight 128-byte runs (lane = 8 * column + chunk, words at rows
    // 4 chunk + q) fall on banks 8 chunk + 2 q + column: all different.
    constexpr int ST_PITCH = PW_COLS + 2;
#ifndef PW_T_PIPE
#define PW_T_PIPE 0   // (measured: 10.3 against 9.8 ms, profiles/r06_pairwise_staging_ab.txt)
#endif
    // PW_T_PIPE (self-distance kernel): groups of 16 rows, two tiles -- the mirrored stores of group g ride in the row loop
    // of group g + 1 (one 16-byte piece per row and thread) instead of a phase of their own between two barriers.  Pitch 260
    // words: rows start 16-byte aligned (one ds_write_b128 per lane and row) at bank 4 r, and the 4 x 16 lanes of a wavefront
    // that assemble sixteen 64-byte runs (lane = 4 * column + chunk, words at rows 4 chunk + q) fall on banks 16 chunk + 4 q +
    // column: all different.
    constexpr bool TPIPE = SYM && PW_T_PIPE;
    constexpr int TP_ROWS = 16, TP_PITCH = PW_COLS + 4;
    __shared__ __attribute__((aligned(16))) float st[SYM ? (TPIPE ? 2 * TP_ROWS * TP_PITCH : T_ROWS * ST_PITCH) : 1];
    auto st_at = [&](int col, int row) -> float& { return st[row * ST_PITCH + col]; };
#else
    constexpr int T_PITCH = T_ROWS + 1;
    __shared__ float st[SYM ? PW_COLS : 1][SYM ? T_PITCH : 1];  // staged TRANSPOSED: st[column][row]
    auto st_at = [&](int col, int row) -> float& { return st[col][row]; };
#endif
    const int tid = threadIdx.x, lane = lane_id(), wave = wave_id();
    const uint32_t bj = blockIdx.x;  // 256-column block
    bool do_t = false;               // this tile's block lies above the diagonal: it is also written transposed
    // A workgroup walks RT row tiles of 128 rows with ITS columns kept in registers (round 6): the columns are 92 strided 4-byte
    // loads per lane, the same in all four wavefronts -- as much traffic through the CU's load path as 40 rows of arithmetic --
    // and were reloaded for every 128 rows.
    constexpr int RT = PW_RT;
    static_assert(RT % (PW_COLS / PW_ROWS) == 0, "a workgroup of the self-distance kernel walks whole 256 x 256 blocks");
    uint64_t i0 = (uint64_t)blockIdx.y * PW_ROWS * RT;
    if (SYM && i0 / PW_COLS > bj) return;  // every block of this workgroup lies below the diagonal: mirrored from above
    const uint64_t j0 = (uint64_t)blockIdx.x * PW_COLS + (uint64_t)lane * PW_CPT;
    uint64_t rows_here = 0;
    if (METRIC == METRIC_MAHALANOBIS) {
        for (int e = tid; e < D * D; e += 256) sm[e] = M[e];
        __syncthreads();
    }

    // the lane's 4 column vectors as two packed pairs: bp[h][k] = (B[j0+2h][k], B[j0+2h+1][k])
    f2 bp[2][D];
    f2 nb[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const uint64_t ja = j0 + 2 * h, jb = ja + 1;
#pragma unroll
        for (int k = 0; k < D; k++) {
            bp[h][k].x = (ja < m) ? B[ja * D + k] : 0.0f;
            bp[h][k].y = (jb < m) ? B[jb * D + k] : 0.0f;
        }
        if (METRIC == METRIC_COSINE) {
            const f2 q = unrolled_dot2<D>([&](int k) { return bp[h][k]; }, [&](int k) { return bp[h][k]; });
            nb[h].x = sqrtf(q.x);
            nb[h].y = sqrtf(q.y);
        }
    }
    float wdiag[D];
#pragma unroll
    for (int k = 0; k < D; k++) wdiag[k] = DIAG ? sm[(k * D + k) % (METRIC == METRIC_MAHALANOBIS ? D * D : 1)] : 0.0f;

    const bool vec_ok = ((ld_out & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0) && (j0 + 3 < m);
    // ---- one row of the tile against the lane's four columns, in pieces so that the loop below can put the tail of one row (the
    // correctly rounded square root: a quarter-rate instruction and two dependent steps behind it) beside the packed arithmetic
    // of the next.  Measured: 0.7 % (profiles/r06_pairwise_two_rows_ab.txt) -- with three wavefronts per SIMD the latencies at
    // the head and the tail of a row were already covered; kept for the A != B Euclidean form, where it costs no registers.
    typedef float f4 __attribute__((ext_vector_type(4)));
#ifndef PW_PIPELINED
#define PW_PIPELINED 1
#endif
    constexpr bool PIPELINED = PW_PIPELINED && METRIC == METRIC_EUCLIDEAN && !SYM;  // (the other forms spill at three waves per SIMD)
    auto load_row = [&](f2 (&ap)[DP / 2], int r) __attribute__((always_inline)) {
#pragma unroll
        for (int k4 = 0; k4 < DP / 4; k4++) {  // same address in every lane: LDS broadcast
            const float4 q = *reinterpret_cast<const float4*>(&sa[r][4 * k4]);
            ap[2 * k4].x = q.x; ap[2 * k4].y = q.y; ap[2 * k4 + 1].x = q.z; ap[2 * k4 + 1].y = q.w;
        }
    };
    // the reference's sum for the two columns of pair h (before the square root / the cosine's division)
    auto row_sum = [&](const f2 (&ap)[DP / 2], auto hc) __attribute__((always_inline)) -> f2 {
        constexpr int h = decltype(hc)::value;
        // term k of the unrolled_dot for the two columns of pair h
        auto term = [&](auto kc) __attribute__((always_inline)) -> f2 {
            constexpr int k = decltype(kc)::value;
            constexpr bool HI = (k & 1) != 0;
            if (METRIC == METRIC_COSINE) return bmul<HI>(ap[k / 2], bp[h][k]);
            const f2 v = bsub<HI>(ap[k / 2], bp[h][k]);
            if (METRIC == METRIC_EUCLIDEAN) return v * v;
            if (DIAG) return (v * splat(wdiag[k])) * v;
            return v;  // (general M: the difference itself)
        };
        if (METRIC == METRIC_MAHALANOBIS && !DIAG) {
            f2 v[D], t[D];
            static_for<D>([&](auto kc) { v[decltype(kc)::value] = term(kc); });
#pragma unroll 1
            for (int jj = 0; jj < D; jj++) {
                f2 acc = splat(0.0f);
#pragma unroll
                for (int ii = 0; ii < D; ii++) acc = acc + v[ii] * splat(sm[ii * D + jj]);
                t[jj] = acc;
            }
            return unrolled_dot2<D>([&](int k) { return t[k]; }, [&](int k) { return v[k]; });
        }
        f2 p[8];
        constexpr int BODY = (D / 8) * 8;
        f2 sum;
        if (METRIC == METRIC_EUCLIDEAN && BODY >= 8) {
            // every term is a square (>= +0), so the reference's `0.0 + term` and `0.0 + (p0 + p4)` are exact
            // identities: start the eight partial sums at their first term (10 of 78 packed instructions less)
            static_for<8>([&](auto kc) { p[decltype(kc)::value] = term(kc); });
            static_for<BODY - 8>([&](auto kc) { constexpr int k = 8 + decltype(kc)::value; p[k & 7] = p[k & 7] + term(std::integral_constant<int, k>{}); });
            sum = p[0] + p[4];
        } else {
#pragma unroll
            for (int u = 0; u < 8; u++) p[u] = splat(0.0f);
            static_for<BODY>([&](auto kc) { constexpr int k = decltype(kc)::value; p[k & 7] = p[k & 7] + term(kc); });
            sum = splat(0.0f);
            sum = sum + (p[0] + p[4]);
        }
        sum = sum + (p[1] + p[5]);
        sum = sum + (p[2] + p[6]);
        sum = sum + (p[3] + p[7]);
        static_for<D - BODY>([&](auto kc) { sum = sum + term(std::integral_constant<int, BODY + decltype(kc)::value>{}); });
        return sum;
    };
    // sum -> distance, the branch-free part (exact unless one of the four sums is a tiny positive number: fix_row)
    auto finish_fast = [&](f2 sum, int h, int r) __attribute__((always_inline)) -> f2 {
        if (METRIC == METRIC_COSINE) return splat(1.0f) - sum / (splat(sna[r]) * nb[h]);
        f2 q;
        q.x = sqrt_rn_fast(sum.x);
        q.y = sqrt_rn_fast(sum.y);
        return q;
    };
    // v_sqrt_f32 does not take denormal inputs: a row with a sum below 2^-96 goes the long way (sqrtf), wave-uniformly.  The test
    // is one unsigned minimum over the four bit patterns -- x in [0, 2^-96) <=> bits(x) < bits(2^-96); inf / NaN / negative sums are
    // large -- and lets ZERO take the long way too (it comes out right there, and a zero distance is a duplicate song: rare).
    // Only the diagonal blocks of a self-distance matrix hold a zero in every row; there (zero_is_common, workgroup-uniform) the
    // test excludes it as rounds 3 - 6 did everywhere: x in (0, 2^-96) <=> bits(x) - 1 < bits(2^-96) - 1, a subtraction per sum.
    bool zero_is_common = false;
    auto is_tiny = [&](f2 a, f2 b) __attribute__((always_inline)) -> bool {
        if (METRIC == METRIC_COSINE) return false;
        if (zero_is_common) {
            const uint32_t lo = min(min(__float_as_uint(a.x) - 1u, __float_as_uint(a.y) - 1u), min(__float_as_uint(b.x) - 1u, __float_as_uint(b.y) - 1u));
            return lo < 0x0F800000u - 1u;
        }
        const uint32_t lo = min(min(__float_as_uint(a.x), __float_as_uint(a.y)), min(__float_as_uint(b.x), __float_as_uint(b.y)));
        return lo < 0x0F800000u;
    };
    auto slow_roots = [&](f2 s) __attribute__((always_inline)) -> f2 { f2 q; q.x = sqrtf(s.x); q.y = sqrtf(s.y); return q; };
    auto emit_row = [&](int r, int R0, f2 q0, f2 q1) __attribute__((always_inline)) {
#ifndef PW_SADDR
#define PW_SADDR 1
#endif
        // (self-distance form: r is the wavefront's row in a scalar register, the row pointer scalar arithmetic + the lane's 16 bytes)
        float* orow = (SYM && PW_SADDR) ? out + wave_uniform((i0 + r) * ld_out + (uint64_t)bj * PW_COLS) + (uint32_t)(lane * PW_CPT)
                                        : out + (i0 + r) * ld_out + j0;
#ifdef PW_ABL_NO_DSTORE
        if (q0.x != 12345.678f) {
        } else
#endif
        if (vec_ok) {
            f4 q4;
            q4.x = q0.x; q4.y = q0.y; q4.z = q1.x; q4.w = q1.y;
            PW_STORE4(PW_NT_DIRECT, q4, reinterpret_cast<f4*>(__builtin_assume_aligned(orow, 16)));  // written once, never re-read here
        } else {
            const float res[PW_CPT] = {q0.x, q0.y, q1.x, q1.y};
#pragma unroll
            for (int c = 0; c < PW_CPT; c++)
                if (j0 + c < m) orow[c] = res[c];
        }
#ifdef PW_ABL_NO_STAGE   // (ablation: wrong results, the cost of the LDS stores of the staging)
        if (do_t && q0.x == 12345.678f) {
#else
        if (do_t) {
#endif
#if PW_ST_ROWMAJOR
            f2* d2 = reinterpret_cast<f2*>(__builtin_assume_aligned(&st_at(PW_CPT * lane, r - R0), 8));
            d2[0] = q0;
            d2[1] = q1;
#else
            st_at(PW_CPT * lane + 0, r - R0) = q0.x;
            st_at(PW_CPT * lane + 1, r - R0) = q0.y;
            st_at(PW_CPT * lane + 2, r - R0) = q1.x;
            st_at(PW_CPT * lane + 3, r - R0) = q1.y;
#endif
        }
    };
    if constexpr (TPIPE) {
        const bool aligned = ((ld_out & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
        const bool cols_full = ((uint64_t)bj + 1) * PW_COLS <= m;
        const int chunk = tid & 3, jcol = tid >> 2;  // this thread's piece of a pass: rows 4 chunk .. 4 chunk + 3 of column 64 pass + jcol
        bool pend = false;     // a staged group waits for its mirrored stores (workgroup-uniform)
        int pend_buf = 0;
        uint64_t pend_first = 0;  // out index of this thread's piece of pass 0 of the waiting group
        int g = 0;
        auto piece = [&](int buf, uint64_t first, int pass) __attribute__((always_inline)) {
            const float* src = st + buf * (TP_ROWS * TP_PITCH) + (4 * chunk) * TP_PITCH + 64 * pass + jcol;
            f4 v4;
            v4.x = src[0]; v4.y = src[TP_PITCH]; v4.z = src[2 * TP_PITCH]; v4.w = src[3 * TP_PITCH];
            __builtin_nontemporal_store(v4, reinterpret_cast<f4*>(__builtin_assume_aligned(out + first + (uint64_t)pass * 64 * ld_out, 16)));
        };
        for (int t = 0; t < RT && i0 < n; t++, i0 += PW_ROWS) {
            const uint32_t bi = (uint32_t)(i0 / PW_COLS);  // block row of this tile
            if (bi > bj) break;                            // mirrored from the block (bj, bi)
            do_t = bj > bi;
            if (t) __syncthreads();  // every wavefront has finished with the previous tile's rows
            rows_here = (n - i0 < (uint64_t)PW_ROWS) ? n - i0 : (uint64_t)PW_ROWS;
            for (int e = tid; e < (int)rows_here * D; e += 256) sa[e / D][e % D] = A[i0 * D + e];
            __syncthreads();
            if (METRIC == METRIC_COSINE) {
                if (tid < (int)rows_here) {
                    const float* a = sa[tid];
                    sna[tid] = sqrtf(unrolled_dot<D>([&](int k) { return a[k]; }, [&](int k) { return a[k]; }));
                }
                __syncthreads();
            }
            for (int R0 = 0; R0 < (int)rows_here; R0 += TP_ROWS, g++) {
                const int buf = g & 1;
                const int r_end = ((int)rows_here < R0 + TP_ROWS) ? (int)rows_here : R0 + TP_ROWS;
                const bool whole = r_end - R0 == TP_ROWS;
                // the waiting group's tile is complete, and nobody still reads the tile this group is about to fill
                if (pend || do_t) __syncthreads();
                if (pend && !whole) {  // a ragged group has no four rows per wavefront to carry the pieces: all four now
#pragma unroll
                    for (int ps = 0; ps < 4; ps++) piece(pend_buf, pend_first, ps);
                    pend = false;
                }
                float* stb = st + buf * (TP_ROWS * TP_PITCH);
                int pass = 0;
#pragma unroll 1
                for (int r = R0 + wave; r < r_end; r += 4, pass++) {
                    f2 ap[DP / 2];
                    load_row(ap, r);
                    const f2 s0 = row_sum(ap, std::integral_constant<int, 0>{}), s1 = row_sum(ap, std::integral_constant<int, 1>{});
                    f2 q0, q1;
                    if (METRIC == METRIC_COSINE) { q0 = finish_fast(s0, 0, r); q1 = finish_fast(s1, 1, r); }
                    else { q0 = sqrt_rn2(s0); q1 = sqrt_rn2(s1); }
                    float* orow = out + (i0 + r) * ld_out + j0;
                    f4 q4;
                    q4.x = q0.x; q4.y = q0.y; q4.z = q1.x; q4.w = q1.y;
                    if (vec_ok) {
                        __builtin_nontemporal_store(q4, reinterpret_cast<f4*>(__builtin_assume_aligned(orow, 16)));
                    } else {
                        const float res[PW_CPT] = {q0.x, q0.y, q1.x, q1.y};
#pragma unroll
                        for (int c = 0; c < PW_CPT; c++)
                            if (j0 + c < m) orow[c] = res[c];
                    }
                    if (do_t) *reinterpret_cast<f4*>(__builtin_assume_aligned(stb + (r - R0) * TP_PITCH + PW_CPT * lane, 16)) = q4;
                    if (pend) piece(pend_buf, pend_first, pass);
                }
                pend = false;
                if (do_t) {
                    if (whole && aligned && cols_full) {
                        pend = true;
                        pend_buf = buf;
                        pend_first = ((uint64_t)bj * PW_COLS + (uint64_t)jcol) * ld_out + i0 + (uint64_t)R0 + 4 * chunk;
                    } else {  // ragged rows / columns or an unaligned matrix: element by element, between two barriers
                        __syncthreads();
                        const int nvalid = r_end - R0;
                        for (int e = tid; e < PW_COLS * TP_ROWS; e += 256) {
                            const int j = e / TP_ROWS, rr = e % TP_ROWS;
                            const uint64_t jrow = (uint64_t)bj * PW_COLS + (uint64_t)j;
                            if (jrow < m && rr < nvalid) out[jrow * ld_out + i0 + (uint64_t)R0 + rr] = stb[rr * TP_PITCH + j];
                        }
                        __syncthreads();
                    }
                }
            }
        }
        if (pend) {
            __syncthreads();
#pragma unroll
            for (int ps = 0; ps < 4; ps++) piece(pend_buf, pend_first, ps);
        }
        return;
    }
    for (int t = 0; t < RT && i0 < n; t++, i0 += PW_ROWS) {
    if (SYM) {
        const uint32_t bi = (uint32_t)(i0 / PW_COLS);  // block row of this tile
        if (bi > bj) break;                            // mirrored from the block (bj, bi)
#ifndef PW_ABL_NO_T   // (ablation builds, tests/tools/variant_obj.sh: what each phase of the self-distance kernel costs)
        do_t = bj > bi;
#endif
        zero_is_common = bi == bj;
    }
    // stage the row tile and, for cosine, the row norms
    if (t) __syncthreads();  // every wavefront has finished with the previous tile's rows
    rows_here = (n - i0 < (uint64_t)PW_ROWS) ? n - i0 : (uint64_t)PW_ROWS;
    {
        // (the thread index made opaque per tile: otherwise the per-lane 64-bit source pointer of this loop is computed once, kept
        // across the arithmetic of all eight tiles -- and spilled, at three wavefronts per SIMD; see PW_DRAIN_BEFORE_GROUPS)
        int t0 = tid;
        if (SYM) asm volatile("" : "+v"(t0));
        for (int e = t0; e < (int)rows_here * D; e += 256) sa[e / D][e % D] = A[i0 * D + e];
    }
    __syncthreads();
    if (METRIC == METRIC_COSINE) {
        if (tid < (int)rows_here) {
            const float* a = sa[tid];
            sna[tid] = sqrtf(unrolled_dot<D>([&](int k) { return a[k]; }, [&](int k) { return a[k]; }));
        }
        __syncthreads();
    }
    // The wavefront's index in a SCALAR register, re-made per tile: everything derived from it (the LDS address of its rows, the
    // row pointer of its direct stores) is then scalar arithmetic per row instead of per-lane 64-bit values computed once and
    // kept -- at three wavefronts per SIMD those were spilled, and the compiler's wait for a reloaded value sat inside the group
    // loop as `s_waitcnt vmcnt(0)`; on gfx950 that counter also holds the wavefront's STORES, so every 32 rows a wavefront stood
    // still until the 16 KiB it had just written were acknowledged by memory.
    int wave_u = wave;
    if (SYM) {
        int w = wave;
        asm volatile("" : "+v"(w));  // (opaque: re-made in this tile, not hoisted to the top of the kernel and carried)
        wave_u = __builtin_amdgcn_readfirstlane(w);
    }
    for (int R0 = 0; R0 < (int)rows_here; R0 += T_ROWS) {
    {
        const int r_end = ((int)rows_here < R0 + T_ROWS) ? (int)rows_here : R0 + T_ROWS;
        int r = R0 + wave_u;
        if (PIPELINED) {
            // two rows per trip, in ONE basic block: the first row's square roots are independent of the second row's 136
            // packed operations (and its LDS reads of the first row's last differences), so the scheduler overlaps them
#pragma unroll 1
            for (; r + 4 < r_end; r += 8) {
                f2 ap[DP / 2];  // the row, two features per register pair
                load_row(ap, r);
                const f2 sa0 = row_sum(ap, std::integral_constant<int, 0>{}), sa1 = row_sum(ap, std::integral_constant<int, 1>{});
                load_row(ap, r + 4);
                f2 qa0 = finish_fast(sa0, 0, r), qa1 = finish_fast(sa1, 1, r);
                const f2 sb0 = row_sum(ap, std::integral_constant<int, 0>{}), sb1 = row_sum(ap, std::integral_constant<int, 1>{});
                f2 qb0 = finish_fast(sb0, 0, r + 4), qb1 = finish_fast(sb1, 1, r + 4);
                const bool tiny = is_tiny(sa0, sa1) || is_tiny(sb0, sb1);
                if (__builtin_expect(__ballot(tiny) != 0ull, 0)) {  // wave-uniform, next to never
                    qa0 = slow_roots(sa0); qa1 = slow_roots(sa1); qb0 = slow_roots(sb0); qb1 = slow_roots(sb1);
                }
                emit_row(r, R0, qa0, qa1);
                emit_row(r + 4, R0, qb0, qb1);
            }
        }
#pragma unroll 1
        for (; r < r_end; r += 4) {   // (the odd row of a ragged tile; every row of the general-M form)
            f2 ap[DP / 2];
            load_row(ap, r);
            const f2 s0 = row_sum(ap, std::integral_constant<int, 0>{}), s1 = row_sum(ap, std::integral_constant<int, 1>{});
            f2 q0 = finish_fast(s0, 0, r), q1 = finish_fast(s1, 1, r);
            if (__builtin_expect(__ballot(is_tiny(s0, s1)) != 0ull, 0)) { q0 = slow_roots(s0); q1 = slow_roots(s1); }  // wave-uniform; ONE test per row
            emit_row(r, R0, q0, q1);
        }
    }
    if (do_t) {  // workgroup-uniform
#ifndef PW_ABL_NO_TBARRIER   // (ablation: wrong results, the cost of the two barriers around the mirrored stores)
        __syncthreads();
#endif
        // mirrored block: its row bj * 256 + j holds the staged column j, 32 consecutive outputs = one 128-byte run,
        // assembled by 8 consecutive lanes (16 bytes each) so that every store instruction writes whole lines
        const int nvalid = ((int)rows_here - R0 < T_ROWS) ? (int)rows_here - R0 : T_ROWS;
        const bool fast = nvalid == T_ROWS && ((ld_out & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
        constexpr int LPR = T_ROWS / 4, JPP = 256 / LPR;  // lanes per run of T_ROWS floats, runs per pass of the workgroup
        const int chunk = tid % LPR;
#if PW_T_UNROLLED
        // every run of the pass read from the tile back to back, then stored: one LDS round trip per 32-row group instead
        // of one per run (the loop below waits for each run's words before it stores them)
        if (fast && ((uint64_t)bj + 1) * PW_COLS <= m && ld_out < (1ull << 26)) {  // (32 rows x ld_out floats stay below 2^32 bytes)
#if PW_SADDR
            // uniform part in scalar registers, the lane's part (row tid / LPR of the pass, four floats at 4 * chunk) in 32 bits
            int t1 = tid;
            asm volatile("" : "+v"(t1));  // (the lane's offset computed here, not once and held -- spilled -- across the arithmetic)
            const uint32_t lane_off = (uint32_t)(t1 / LPR) * (uint32_t)ld_out + 4u * (uint32_t)(t1 % LPR);
            float* dst = out + wave_uniform((uint64_t)bj * PW_COLS * ld_out + i0 + (uint64_t)R0) + lane_off;
#else
            uint64_t first = ((uint64_t)bj * PW_COLS + (uint64_t)(tid / LPR)) * ld_out + i0 + (uint64_t)R0 + 4 * chunk;
            asm volatile("" : "+v"(first));  // (computed here, not held across the arithmetic)
            float* dst = out + first;
#endif
            const uint64_t step = (uint64_t)JPP * ld_out;
            f4 v4[PW_COLS / JPP];
#pragma unroll
            for (int p = 0; p < PW_COLS / JPP; p++) {
                const int j = JPP * p + tid / LPR;
                v4[p].x = st_at(j, 4 * chunk); v4[p].y = st_at(j, 4 * chunk + 1); v4[p].z = st_at(j, 4 * chunk + 2); v4[p].w = st_at(j, 4 * chunk + 3);
            }
#pragma unroll
            for (int p = 0; p < PW_COLS / JPP; p++)
#ifdef PW_ABL_NO_TSTORE
                if (v4[p].x == 12345.678f)
#endif
                PW_STORE4(PW_NT_MIRROR, v4[p], reinterpret_cast<f4*>(__builtin_assume_aligned(dst + p * step, 16)));
        } else
#endif
#pragma unroll 1   // (unrolled, the eight row pointers are hoisted out of the tile loop and held across the arithmetic: 16 registers)
        for (int p = 0; p < PW_COLS / JPP; p++) {
            const int j = JPP * p + tid / LPR;
            const uint64_t jrow = (uint64_t)bj * PW_COLS + (uint64_t)j;
            if (jrow >= m) continue;
            float* dst = out + jrow * ld_out + i0 + (uint64_t)R0 + 4 * chunk;
#ifdef PW_ABL_NO_TSTORE
            if (st_at(j, 4 * chunk) != 12345.678f) continue;
#endif
            if (fast) {
                typedef float f4 __attribute__((ext_vector_type(4)));
                f4 v4;
                v4.x = st_at(j, 4 * chunk); v4.y = st_at(j, 4 * chunk + 1); v4.z = st_at(j, 4 * chunk + 2); v4.w = st_at(j, 4 * chunk + 3);
                __builtin_nontemporal_store(v4, reinterpret_cast<f4*>(__builtin_assume_aligned(dst, 16)));
            } else {
                for (int q = 0; q < 4; q++)
                    if (4 * chunk + q < nvalid) dst[q] = st_at(j, 4 * chunk + q);
            }
        }
#ifndef PW_ABL_NO_TBARRIER
        __syncthreads();
#endif
    }
    }
    }
}

// generic feature count (d <= 64): one thread per pair, vectors read from global/L2
__global__ __launch_bounds__(256) void pairwise_generic_kernel(const float* __restrict__ A, uint64_t n,
                                                               const float* __restrict__ B, uint64_t m, uint32_t d,
                                                               int metric, const float* __restrict__ M,
                                                               float* __restrict__ out, uint64_t ld_out) {
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t i = blockIdx.y;
    if (j >= m || i >= n) return;
    const float* a = A + i * d;
    const float* b = B + j * d;
    float v[64], t[64];
    auto udot = [&](const float* xs, const float* ys) {
        float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        uint32_t k = 0;
        for (; k + 8 <= d; k += 8)
            for (int u = 0; u < 8; u++) p[u] = p[u] + xs[k + u] * ys[k + u];
        float sum = 0.0f;
        sum = sum + (p[0] + p[4]);
        sum = sum + (p[1] + p[5]);
        sum = sum + (p[2] + p[6]);
        sum = sum + (p[3] + p[7]);
        for (; k < d; k++) sum = sum + xs[k] * ys[k];
        return sum;
    };
    float res;
    if (metric == METRIC_COSINE) {
        res = 1.0f - udot(a, b) / (sqrtf(udot(a, a)) * sqrtf(udot(b, b)));
    } else {
        for (uint32_t k = 0; k < d; k++) v[k] = a[k] - b[k];
        if (metric == METRIC_EUCLIDEAN) {
            res = sqrtf(udot(v, v));
        } else {
            for (uint32_t jj = 0; jj < d; jj++) {
                float s = 0.0f;
                for (uint32_t ii = 0; ii < d; ii++) s = s + v[ii] * M[ii * d + jj];
                t[jj] = s;
            }
            res = sqrtf(udot(t, v));
        }
    }
    out[i * ld_out + j] = res;
}

template <int D>
static void launch_d(const float* A, uint64_t n, const float* B, uint64_t m, int metric, const float* M, int diag,
                     float* out, uint64_t ld, hipStream_t st) {
    dim3 grid((uint32_t)((m + PW_COLS - 1) / PW_COLS), (uint32_t)((n + (uint64_t)PW_ROWS * PW_RT - 1) / ((uint64_t)PW_ROWS * PW_RT)));
    if (A == B && n == m) {  // self-distance matrix: upper block triangle + mirrored stores
        if (metric == METRIC_EUCLIDEAN)
            hipLaunchKernelGGL((pairwise_kernel<D, METRIC_EUCLIDEAN, false, true>), grid, dim3(256), 0, st, A, n, B, m, M, out, ld);
        else if (metric == METRIC_COSINE)
            hipLaunchKernelGGL((pairwise_kernel<D, METRIC_COSINE, false, true>), grid, dim3(256), 0, st, A, n, B, m, M, out, ld);
        else if (diag)
            hipLaunchKernelGGL((pairwise_kernel<D, METRIC_MAHALANOBIS, true, true>), grid, dim3(256), 0, st, A, n, B, m, M, out, ld);
        else
            hipLaunchKernelGGL((pairwise_kernel<D, METRIC_MAHALANOBIS, false, true>), grid, dim3(256), 0, st, A, n, B, m, M, out, ld);
        return;
    }
    if (metric == METRIC_EUCLIDEAN)
        hipLaunchKernelGGL((pairwise_kernel<D, METRIC_EUCLIDEAN, false>), grid, dim3(256), 0, st, A, n, B, m, M, out, ld);
    else if (metric == METRIC_COSINE)
        hipLaunchKernelGGL((pairwise_kernel<D, METRIC_COSINE, false>), grid, dim3(256), 0, st, A, n, B, m, M, out, ld);
    else if (diag)
        hipLaunchKernelGGL((pairwise_kernel<D, METRIC_MAHALANOBIS, true>), grid, dim3(256), 0, st, A, n, B, m, M, out, ld);
    else
        hipLaunchKernelGGL((pairwise_kernel<D, METRIC_MAHALANOBIS, false>), grid, dim3(256), 0, st, A, n, B, m, M, out, ld);
}

void launch_pairwise(const float* A, uint64_t n, const float* B, uint64_t m, uint32_t d, int metric, const float* M,
                     int m_is_diag, float* out, uint64_t ld_out, hipStream_t st) {
    if (n == 0 || m == 0) return;
    // grid.y is limited to 65535 row tiles (8.3 M rows); larger problems are sliced by the caller
    if (d == 23) launch_d<23>(A, n, B, m, metric, M, m_is_diag, out, ld_out, st);
    else if (d == 20) launch_d<20>(A, n, B, m, metric, M, m_is_diag, out, ld_out, st);
    else
        for (uint64_t r0 = 0; r0 < n; r0 += 65535) {  // grid.y <= 65535
            const uint64_t rows = (n - r0 < 65535) ? n - r0 : 65535;
            hipLaunchKernelGGL(pairwise_generic_kernel, dim3((uint32_t)((m + 255) / 256), (uint32_t)rows), dim3(256), 0,
                               st, A + r0 * d, rows, B, m, d, metric, M, out + r0 * ld_out, ld_out);
        }
}

}  // namespace bg
