// kernels_playlist.hip -- playlist ordering on the device (compiled with -ffp-contract=off).
//
// Reference (src/playlist.rs):
//   FunctionDistanceMetric::distance :52-58   sum over the seed set of func(seed, vector), sequential f32
//   closest_to_songs                 :256-270 stable sort of the candidates by that distance
//   song_to_song                     :272-326 greedy nearest-neighbour chain: argmin (first minimum) over the pool,
//                                             remove, the emitted song becomes the only seed
// The per-pair arithmetic is the bit-exact restatement used by the pairwise kernel (ndarray's unrolled_dot
// order, no FMA), so orders -- including ties -- match the reference's.
//
// song_to_song is n dependent steps of an O(n d) scan.  One persistent launch runs all of them: the pool is
// spread over G <= #CU workgroups (each thread owns the candidates gtid, gtid + T, ...), a step is a
// local scan -> workgroup min -> one 8-byte slot per workgroup -> grid barrier -> every workgroup reduces the
// G slots redundantly.  Keys are (order-preserving u32 image of the distance) << 32 | candidate index, so the
// u64 minimum is the first minimum in pool order (Vec::remove keeps the relative order of the pool).
#include "device_utils.hpp"
#include "internal.hpp"

namespace bg {

enum { PL_EUCLIDEAN = 0, PL_COSINE = 1, PL_MAHALANOBIS = 2 };
constexpr int PL_DMAX = 64;

// ndarray::numeric_util::unrolled_dot (8 partial sums, (p0+p4)+(p1+p5)+(p2+p6)+(p3+p7), tail sequentially)
template <typename FX, typename FY>
__device__ __forceinline__ float pl_udot(FX xs, FY ys, uint32_t d) {
    float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    uint32_t k = 0;
    for (; k + 8 <= d; k += 8)
#pragma unroll
        for (int u = 0; u < 8; u++) p[u] = p[u] + xs(k + u) * ys(k + u);
    float sum = 0.0f;
    sum = sum + (p[0] + p[4]);
    sum = sum + (p[1] + p[5]);
    sum = sum + (p[2] + p[6]);
    sum = sum + (p[3] + p[7]);
    for (; k < d; k++) sum = sum + xs(k) * ys(k);
    return sum;
}

// euclidean / cosine / mahalanobis distance of src/playlist.rs:65-79,140-142 between a (any address space)
// and b; same evaluation order as pairwise_generic_kernel
__device__ __forceinline__ float pl_distance(const float* a, const float* b, uint32_t d, int metric,
                                             const float* __restrict__ M) {
    if (metric == PL_COSINE) {
        const float ab = pl_udot([&](uint32_t k) { return a[k]; }, [&](uint32_t k) { return b[k]; }, d);
        const float aa = pl_udot([&](uint32_t k) { return a[k]; }, [&](uint32_t k) { return a[k]; }, d);
        const float bb = pl_udot([&](uint32_t k) { return b[k]; }, [&](uint32_t k) { return b[k]; }, d);
        return 1.0f - ab / (sqrtf(aa) * sqrtf(bb));
    }
    if (metric == PL_EUCLIDEAN) {
        return sqrtf(pl_udot([&](uint32_t k) { return a[k] - b[k]; }, [&](uint32_t k) { return a[k] - b[k]; }, d));
    }
    float t[PL_DMAX];
    for (uint32_t jj = 0; jj < d; jj++) {
        float s = 0.0f;
        for (uint32_t ii = 0; ii < d; ii++) s = s + (a[ii] - b[ii]) * M[ii * d + jj];
        t[jj] = s;
    }
    return sqrtf(pl_udot([&](uint32_t k) { return t[k]; }, [&](uint32_t k) { return a[k] - b[k]; }, d));
}

// FunctionDistanceMetric::distance: sequential f32 sum over the seeds, in seed order
__device__ __forceinline__ float pl_set_distance(const float* __restrict__ seeds, uint32_t n_seeds, const float* v,
                                                 uint32_t d, int metric, const float* __restrict__ M) {
    float acc = 0.0f;
    for (uint32_t i = 0; i < n_seeds; i++) acc = acc + pl_distance(seeds + (size_t)i * d, v, d, metric, M);
    return acc;
}

// order-preserving u32 image of a non-NaN float; -0.0 and +0.0 compare equal in the reference (partial_cmp /
// n32), so both map to the image of +0.0
__device__ __forceinline__ uint32_t f32_key(float v) {
    if (v == 0.0f) v = 0.0f;
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// ---- distance of every candidate to the seed set (+ sort keys, + NaN flag) ----
__global__ __launch_bounds__(256) void set_distance_kernel(const float* __restrict__ seeds, uint32_t n_seeds,
                                                           const float* __restrict__ cand, uint64_t n, uint32_t d,
                                                           int metric, const float* __restrict__ M,
                                                           float* __restrict__ dist, uint32_t* __restrict__ keys,
                                                           uint32_t* __restrict__ idx, uint32_t* __restrict__ nan_flag) {
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    float c[PL_DMAX];
    for (uint32_t k = 0; k < d; k++) c[k] = cand[j * d + k];
    const float v = pl_set_distance(seeds, n_seeds, c, d, metric, M);
    if (dist) dist[j] = v;
    if (keys) {
        keys[j] = f32_key(v);
        idx[j] = (uint32_t)j;
    }
    if (v != v) atomicOr(nan_flag, 1u);
}

void launch_set_distance(const float* seeds, uint32_t n_seeds, const float* cand, uint64_t n, uint32_t d, int metric,
                         const float* M, float* dist, uint32_t* keys, uint32_t* idx, uint32_t* nan_flag, hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(set_distance_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, seeds, n_seeds, cand, n,
                       d, metric, M, dist, keys, idx, nan_flag);
}

// ---- stable LSD radix sort of (key, index) pairs: equal distances keep their original order, like
// slice::sort_by_cached_key (src/playlist.rs:256-270).  Four passes of eight bits; per pass
//   radix_hist_kernel     a workgroup counts the digits of its tile of 2048 pairs (LDS histogram)     -> hist[digit][tile]
//   radix_scan_kernel     one workgroup turns the digit-major table into exclusive offsets (a stable sort places digit d
//                         of tile b behind every smaller digit and behind digit d of the tiles before b)
//   radix_scatter_kernel  a wavefront owns 512 consecutive pairs of the tile, eight at a time in order: the lanes that hold
//                         the same digit find one another with eight ballots (one per digit bit), a lane's rank among them
//                         is the population count of the match mask below it, the wavefront's running count per digit
//                         lives in LDS; the four wavefronts' counts are stacked in tile order and added to the tile's
//                         offset.  No pair ever overtakes an equal one.
// The pairs ping-pong in -> tmp -> in -> tmp -> out.  100 000 pairs: twelve launches of a few microseconds.
constexpr int RS_ITEMS = 8, RS_TILE = 256 * RS_ITEMS;

__global__ __launch_bounds__(256) void radix_hist_kernel(const uint32_t* __restrict__ keys, uint32_t n, int shift,
                                                         uint32_t* __restrict__ hist, uint32_t n_tiles) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * (uint32_t)RS_TILE;
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        const uint32_t j = base + (uint32_t)i * 256u + threadIdx.x;
        if (j < n) atomicAdd(&h[(keys[j] >> shift) & 0xFFu], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * n_tiles + blockIdx.x] = h[threadIdx.x];
}

// exclusive scan of `len` counters in place, one workgroup (len = 256 x tiles: 12 544 for 100 000 pairs)
__global__ __launch_bounds__(256) void radix_scan_kernel(uint32_t* __restrict__ v, uint32_t len) {
    __shared__ uint32_t wsum[4];
    const uint32_t per = (len + 255u) / 256u;            // consecutive counters per thread
    const uint32_t lo = threadIdx.x * per, hi = lo + per < len ? lo + per : len;
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += v[i];
    const uint32_t incl = wave_scan_incl_u32(sum);
    if (lane_id() == 63) wsum[wave_id()] = incl;
    __syncthreads();
    uint32_t run = incl - sum;
    for (int w = 0; w < wave_id(); w++) run += wsum[w];
    for (uint32_t i = lo; i < hi; i++) {
        const uint32_t c = v[i];
        v[i] = run;
        run += c;
    }
}

// one tile of the scatter; `tile_first[d]` (thread d's argument) = the output slot of the tile's first pair with digit d
__device__ __forceinline__ void radix_scatter_tile(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                   uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint32_t n,
                                                   int shift, uint32_t tile, uint32_t tile_first, uint32_t (*wcount)[256]) {
    // wcount[w][d]: pairs with digit d wavefront w has placed so far; later: its first output slot
    const int lane = lane_id(), wave = wave_id();
    for (int i = threadIdx.x; i < 4 * 256; i += 256) (&wcount[0][0])[i] = 0;
    __syncthreads();
    const uint32_t base = tile * (uint32_t)RS_TILE + (uint32_t)wave * (64u * RS_ITEMS);
    uint32_t key[RS_ITEMS], val[RS_ITEMS], rank[RS_ITEMS];
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        const uint32_t j = base + 64u * i + (uint32_t)lane;
        key[i] = j < n ? keys_in[j] : 0xFFFFFFFFu;
        val[i] = j < n ? vals_in[j] : 0u;
    }
    const uint64_t below = (1ull << lane) - 1ull;
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        const bool live = base + 64u * i + (uint32_t)lane < n;
        const uint32_t dgt = (key[i] >> shift) & 0xFFu;
        uint64_t match = __ballot(live);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const uint64_t has = __ballot((dgt >> b) & 1u);
            match &= ((dgt >> b) & 1u) ? has : ~has;
        }
        // every lane of a match group reads the group's count so far, then its first lane adds the group (a wavefront's
        // LDS accesses execute in program order)
        const uint32_t before = live ? wcount[wave][dgt] : 0u;
        rank[i] = before + (uint32_t)__popcll(match & below);
        if (live && (match & below) == 0) wcount[wave][dgt] = before + (uint32_t)__popcll(match);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    {
        // thread d: where each wavefront's share of the tile's digit d starts (tile order)
        const uint32_t d = threadIdx.x;
        uint32_t run = tile_first;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const uint32_t c = wcount[w][d];
            wcount[w][d] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        if (base + 64u * i + (uint32_t)lane < n) {
            const uint32_t dst = wcount[wave][(key[i] >> shift) & 0xFFu] + rank[i];
            keys_out[dst] = key[i];
            vals_out[dst] = val[i];
        }
    }
}

__global__ __launch_bounds__(256) void radix_scatter_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                            uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                            uint32_t n, int shift, const uint32_t* __restrict__ offs,
                                                            uint32_t n_tiles) {
    __shared__ uint32_t wcount[4][256];
    radix_scatter_tile(keys_in, vals_in, keys_out, vals_out, n, shift, blockIdx.x, offs[(size_t)threadIdx.x * n_tiles + blockIdx.x], wcount);
}

// The whole sort in ONE launch when its tiles fit the machine at once (<= one workgroup per CU: 2048 pairs each, i.e. up to
// ~half a million pairs): the eight dependent steps of the four passes are separated by a grid barrier (an arrival counter
// polled by one thread per workgroup, as in song_to_song_kernel) instead of by eleven more launches -- at 100 000 pairs a
// launch costs more than the work it carries.  Every workgroup derives its own offsets from the digit-major table: thread
// d sums row d (the tiles before its own -> where the tile's share of the digit starts; all tiles -> the digit's total),
// the totals are scanned over the 256 digits in the workgroup.
__global__ __launch_bounds__(256) void radix_sort_persistent_kernel(uint32_t* __restrict__ keys_a, uint32_t* __restrict__ vals_a,
                                                                    uint32_t* __restrict__ keys_b, uint32_t* __restrict__ vals_b,
                                                                    uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                                    uint32_t n, uint32_t* hist, uint32_t* sync) {
    __shared__ uint32_t wcount[4][256];
    __shared__ uint32_t h[256];
    __shared__ uint32_t wsum[4];
    const uint32_t G = gridDim.x, tile = blockIdx.x, tid = threadIdx.x;
    uint32_t epoch = 0;
    auto grid_barrier = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // this thread's table entries / pairs are visible to the device
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t target = G * (++epoch);
            while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    };
    for (int pass = 0; pass < 4; pass++) {
        const uint32_t *ki = (pass & 1) ? keys_b : keys_a, *vi = (pass & 1) ? vals_b : vals_a;
        uint32_t *ko = pass == 3 ? keys_out : ((pass & 1) ? keys_a : keys_b), *vo = pass == 3 ? vals_out : ((pass & 1) ? vals_a : vals_b);
        const int shift = 8 * pass;
        h[tid] = 0;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RS_ITEMS; i++) {
            const uint32_t j = tile * (uint32_t)RS_TILE + (uint32_t)i * 256u + tid;
            if (j < n) atomicAdd(&h[(ki[j] >> shift) & 0xFFu], 1u);
        }
        __syncthreads();
        hist[(size_t)tid * G + tile] = h[tid];
        grid_barrier();
        uint32_t before = 0, total = 0;
        for (uint32_t t = 0; t < G; t++) {
            const uint32_t c = hist[(size_t)tid * G + t];
            before += t < tile ? c : 0u;
            total += c;
        }
        const uint32_t incl = wave_scan_incl_u32(total);
        if (lane_id() == 63) wsum[wave_id()] = incl;
        __syncthreads();
        uint32_t first = incl - total + before;  // pairs with smaller digits + this digit in the tiles before
        for (int w = 0; w < wave_id(); w++) first += wsum[w];
        radix_scatter_tile(ki, vi, ko, vo, n, shift, tile, first, wcount);
        grid_barrier();  // the next pass reads what every tile has written -- and overwrites the table
    }
}

// tmp == NULL: report the scratch bytes.  keys_in / vals_in are overwritten (they serve as the second ping-pong buffer).
// `sync` = one zeroed 32-bit word for the single-launch form, taken when the tiles fit `max_coresident` workgroups.
hipError_t sort_pairs_u32(void* tmp, size_t* tmp_bytes, uint32_t* keys_in, uint32_t* keys_out, uint32_t* vals_in,
                          uint32_t* vals_out, uint32_t n, hipStream_t st, uint32_t* sync, uint32_t max_coresident) {
    const uint32_t n_tiles = (n + RS_TILE - 1) / RS_TILE;
    const size_t need = ((size_t)2 * n + (size_t)256 * n_tiles + 64) * sizeof(uint32_t);
    if (!tmp) { *tmp_bytes = need; return hipSuccess; }
    if (*tmp_bytes < need) return hipErrorInvalidValue;
    if (n == 0) return hipSuccess;
    uint32_t* t_keys = reinterpret_cast<uint32_t*>(tmp);
    uint32_t* t_vals = t_keys + n;
    uint32_t* hist = t_vals + n;
    if (sync && n_tiles <= max_coresident && n_tiles <= 256u) {
        hipLaunchKernelGGL(radix_sort_persistent_kernel, dim3(n_tiles), dim3(256), 0, st, keys_in, vals_in, t_keys, t_vals, keys_out,
                           vals_out, n, hist, sync);
        return hipGetLastError();
    }
    for (int pass = 0; pass < 4; pass++) {
        const uint32_t *ki = (pass & 1) ? t_keys : keys_in, *vi = (pass & 1) ? t_vals : vals_in;
        uint32_t *ko = pass == 3 ? keys_out : ((pass & 1) ? keys_in : t_keys), *vo = pass == 3 ? vals_out : ((pass & 1) ? vals_in : t_vals);
        hipLaunchKernelGGL(radix_hist_kernel, dim3(n_tiles), dim3(256), 0, st, ki, n, 8 * pass, hist, n_tiles);
        hipLaunchKernelGGL(radix_scan_kernel, dim3(1), dim3(256), 0, st, hist, 256u * n_tiles);
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(n_tiles), dim3(256), 0, st, ki, vi, ko, vo, n, 8 * pass, hist, n_tiles);
    }
    return hipGetLastError();
}

// ---- song_to_song: one persistent launch, grid barrier per step ----
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(v, off, WAVE);
        v = o < v ? o : v;
    }
    return v;
}

// PER > 0: every thread keeps its PER candidates (D features each) in registers for the whole chain, so a step
// touches memory only for the G slots and the winner's row.  PER == 0: generic path, candidates re-read from
// L2 / MALL every step (any d <= 64, up to 64 candidates per thread).
template <int D, int PER>
__global__ __launch_bounds__(256) void song_to_song_kernel(const float* __restrict__ seeds, uint32_t n_seeds,
                                                           const float* __restrict__ cand, uint32_t n, uint32_t d_rt,
                                                           int metric, const float* __restrict__ M,
                                                           uint32_t* __restrict__ order,
                                                           unsigned long long* slots,  // [2][gridDim.x]
                                                           uint32_t* sync) {           // [0] barrier, [1] NaN flag (for the host)
    __shared__ unsigned long long red[4];
    __shared__ unsigned long long s_win;
    __shared__ float s_cur[PL_DMAX];
    const uint32_t d = PER > 0 ? (uint32_t)D : d_rt;
    const uint32_t G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
    const uint32_t T = G * 256, gtid = wg * 256 + tid;
    const int lane = lane_id(), wave = wave_id();
    const uint32_t per = PER > 0 ? (uint32_t)PER : (n + T - 1) / T;  // candidates per thread (<= 64, checked by the host)
    unsigned long long alive = 0;
    for (uint32_t k = 0; k < per; k++)
        if (gtid + k * T < n) alive |= 1ull << k;
    float mine[PER > 0 ? PER : 1][PER > 0 ? D : 1];
    if (PER > 0) {
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const uint32_t j = gtid + (uint32_t)k * T;
#pragma unroll
            for (int q = 0; q < D; q++) mine[k][q] = (j < n) ? cand[(size_t)j * D + q] : 0.0f;
        }
    }

    for (uint32_t step = 0; step < n; step++) {
        // ---- local scan: the metric is the seed set at step 0, the previously emitted song afterwards ----
        unsigned long long best = ~0ull;
        bool saw_nan = false;
        auto visit = [&](const float* c, uint32_t j) {
            const float v = (step == 0) ? pl_set_distance(seeds, n_seeds, c, d, metric, M)
                                        : (0.0f + pl_distance(s_cur, c, d, metric, M));
            if (v != v) saw_nan = true;
            const unsigned long long key = ((unsigned long long)f32_key(v) << 32) | j;
            best = key < best ? key : best;
        };
        if (PER > 0) {
#pragma unroll
            for (int k = 0; k < PER; k++)
                if ((alive >> k) & 1ull) visit(mine[k], gtid + (uint32_t)k * T);
        } else {
            for (uint32_t k = 0; k < per; k++) {
                if (!((alive >> k) & 1ull)) continue;
                const uint32_t j = gtid + k * T;
                float c[PL_DMAX];
                for (uint32_t q = 0; q < d; q++) c[q] = cand[(size_t)j * d + q];
                visit(c, j);
            }
        }
        // A NaN distance (the reference panics there) travels THROUGH the barrier protocol as the reserved slot value 0
        // (no valid key is 0: the image of -inf is 0x007FFFFF): every workgroup reduces the same G slots after the same
        // barrier, so the exit below is taken by all of them at the same step.  (A side flag read outside the barrier
        // could be seen by a slower workgroup one step early, which then never arrives at the barrier the faster ones
        // are already spinning on.)
        if (saw_nan) best = 0ull;
        best = wave_min_u64(best);
        if (lane == 0) red[wave] = best;
        __syncthreads();
        if (tid == 0) {
            unsigned long long b = red[0];
            for (int w = 1; w < 4; w++) b = red[w] < b ? red[w] : b;
            __hip_atomic_store(&slots[(size_t)(step & 1) * G + wg], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // grid barrier: monotonically increasing arrival counter; release on arrive, relaxed polling and ONE
            // acquire fence on leave (an acquire per poll would invalidate the caches on every iteration)
            __hip_atomic_fetch_add(&sync[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t target = G * (step + 1);
            while (__hip_atomic_load(&sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target)
                __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        // ---- every workgroup reduces the G slots (G <= 256: one per thread) ----
        unsigned long long w = ~0ull;
        if (tid < G) w = __hip_atomic_load(&slots[(size_t)(step & 1) * G + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        w = wave_min_u64(w);
        if (lane == 0) red[wave] = w;
        __syncthreads();
        if (tid == 0) {
            unsigned long long b = red[0];
            for (int q = 1; q < 4; q++) b = red[q] < b ? red[q] : b;
            s_win = b;
        }
        __syncthreads();
        if (s_win == 0ull) {  // NaN: grid-uniform exit
            if (wg == 0 && tid == 0) __hip_atomic_store(&sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        const uint32_t win = (uint32_t)(s_win & 0xFFFFFFFFull);
        if (wg == 0 && tid == 0) order[step] = win;
        if (win % T == gtid) alive &= ~(1ull << (win / T));
        if (tid < d) s_cur[tid] = cand[(size_t)win * d + tid];
        __syncthreads();
    }
}

void launch_song_to_song(const float* seeds, uint32_t n_seeds, const float* cand, uint32_t n, uint32_t d, int metric,
                         const float* M, uint32_t* order, unsigned long long* slots, uint32_t* sync, uint32_t grid,
                         hipStream_t st) {
    if (n == 0) return;
    const uint32_t per = (n + grid * 256 - 1) / (grid * 256);
#define S2S(DD, PP) hipLaunchKernelGGL((song_to_song_kernel<DD, PP>), dim3(grid), dim3(256), 0, st, seeds, n_seeds, cand, n, d, \
                                       metric, M, order, slots, sync)
    if (d == 23 && per <= 1) S2S(23, 1);
    else if (d == 23 && per <= 2) S2S(23, 2);
    else if (d == 23 && per <= 4) S2S(23, 4);
    else if (d == 20 && per <= 1) S2S(20, 1);
    else if (d == 20 && per <= 2) S2S(20, 2);
    else if (d == 20 && per <= 4) S2S(20, 4);
    else S2S(1, 0);
#undef S2S
}

// ---- one pair, both vectors passed BY VALUE in the kernel arguments: Song::distance / Analysis::distance
// (src/song/mod.rs:364-370, 519-521) without staging copies; the result goes to a page-locked host word ----
struct PairArgs { float a[PL_DMAX]; float b[PL_DMAX]; };

__global__ __launch_bounds__(64) void pair_distance_kernel(PairArgs p, uint32_t d, int metric, const float* __restrict__ M,
                                                           float* __restrict__ out) {
    __shared__ float sa[PL_DMAX], sb[PL_DMAX], st[PL_DMAX];
    const uint32_t tid = threadIdx.x;
    sa[tid] = p.a[tid];
    sb[tid] = p.b[tid];
    __syncthreads();
    if (metric != PL_MAHALANOBIS) {
        if (tid == 0) *out = pl_distance(sa, sb, d, metric, M);
        return;
    }
    // (a - b) . M: column jj is an independent sequential sum over ii (Array1.dot(Array2) walks each column in order), so
    // the d columns go to d lanes; the final unrolled_dot stays on one lane.  Same operations, same order as pl_distance.
    if (tid < d) {
        float acc = 0.0f;
        for (uint32_t ii = 0; ii < d; ii++) acc = acc + (sa[ii] - sb[ii]) * M[ii * d + tid];
        st[tid] = acc;
    }
    __syncthreads();
    if (tid == 0) *out = sqrtf(pl_udot([&](uint32_t k) { return st[k]; }, [&](uint32_t k) { return sa[k] - sb[k]; }, d));
}

void launch_pair_distance(const float* a, const float* b, uint32_t d, int metric, const float* d_M, float* out, hipStream_t st) {
    PairArgs p;
    for (uint32_t k = 0; k < (uint32_t)PL_DMAX; k++) { p.a[k] = k < d ? a[k] : 0.0f; p.b[k] = k < d ? b[k] : 0.0f; }
    hipLaunchKernelGGL(pair_distance_kernel, dim3(1), dim3(64), 0, st, p, d, metric, d_M, out);
}

}  // namespace bg
