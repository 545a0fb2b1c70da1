// kernels_tempo.hip -- aubio-style tempo on the device (compiled with -ffp-contract=off so every
// f32 operation rounds exactly like the reference's scalar Rust).
//
//   onset_kernel : PeakPicker threshold (src/aubio.rs:733-768, 661-685, 482-554): a pure function of
//                  the last 7 SpecFlux values, so one thread per tempo frame.
//   beat_acf_kernel   : the state-independent half of BeatTracking::do_ (src/aubio.rs:966-1003): the
//                  autocorrelation of every run's detection-function frame, its comb filterbank sums
//                  for both time signatures, the Rayleigh-weighted period and get_timesig (:864-907)
//                  -- one workgroup per (song, run), ~124 k of them for 1024 three-minute songs.
//   beat_track_kernel : the sequential half: checkstate (:1096-1227), beat phase (:1025-1054),
//                  Tempo::do_'s beat / silence test (:1378-1443), get_bpm and BPMDesc's median
//                  (src/temporal.rs:50-77).  The chain (gp/rp1/rp2/counter/flagstep/timesig/lastbeat/
//                  gwv/phwv) is walked by ONE wavefront per song with no workgroup barrier: its inputs are
//                  prefetched a run ahead, so a run costs a few microseconds and the kernel holds 1/16 of
//                  the registers the former one-workgroup-per-song tracker did.
#include "device_utils.hpp"
#include "internal.hpp"

namespace bg {

// correctly rounded f32 exp via f64 (glibc's expf, which the reference links, is correctly rounded
// in all but vanishingly rare cases)
__device__ __forceinline__ float exp_f32(float x) { return (float)exp((double)x); }

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void onset_kernel(const SongDesc* __restrict__ songs,
                                                    const float* __restrict__ flux,
                                                    float* __restrict__ thresholded) {
    const SongDesc sd = songs[blockIdx.y];
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (!sd.ok || t >= sd.n_b) return;
    const float* f = flux + sd.b_off;
    // onset_keep after pushing frame t: [f[t-6] .. f[t]], zeros before the song starts
    float p[7], tmp[7];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const long idx = (long)t - 6 + i;
        p[i] = idx >= 0 ? f[idx] : 0.0f;
    }
    // Biquad::do_filtfilt (state reset before each pass); a2 = 0
    const float b0 = 0.1599879f, b1 = 0.31997577f, b2 = 0.1599879f, a1 = 0.23484048f, a2 = 0.0f;
    float x1 = 0.f, x2 = 0.f, y1 = 0.f, y2 = 0.f;
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const float x0 = p[i];
        const float y0 = b0 * x0 + b1 * x1 + b2 * x2 - a1 * y1 - a2 * y2;
        x2 = x1; x1 = x0; y2 = y1; y1 = y0;
        p[i] = y0;
    }
#pragma unroll
    for (int i = 0; i < 7; i++) tmp[6 - i] = p[i];
    x1 = x2 = y1 = y2 = 0.f;
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const float x0 = tmp[i];
        const float y0 = b0 * x0 + b1 * x1 + b2 * x2 - a1 * y1 - a2 * y2;
        x2 = x1; x1 = x0; y2 = y1; y1 = y0;
        tmp[i] = y0;
    }
#pragma unroll
    for (int i = 0; i < 7; i++) p[i] = tmp[6 - i];
    // vec_mean (sequential) and vec_median (element (n-1)/2 = 3 of the sorted 7)
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < 7; i++) sum += p[i];
    const float mean = sum / 7.0f;
    const float p5 = p[5];
#pragma unroll
    for (int i = 0; i < 7; i++)
#pragma unroll
        for (int j = 0; j + 1 < 7 - i; j++) {
            const float lo = fminf(p[j], p[j + 1]), hi = fmaxf(p[j], p[j + 1]);
            p[j] = lo; p[j + 1] = hi;
        }
    const float median = p[3];
    thresholded[sd.b_off + t] = p5 - median - mean * 0.3f;  // threshold 0.3 (src/aubio.rs:1347)
}

void launch_onset(const Batch& b, const Workspace& w, hipStream_t st) {
    // grid.x sized for the longest song is computed by the caller through total_b / n_songs upper bound
    if (b.n_songs == 0 || b.total_b == 0) return;
    // the host passes the maximum n_b in tiles_e's sibling: recomputed here from total_b would be wrong
    // for ragged batches, so Batch carries it in max_nb (see blissgpu.hip)
    hipLaunchKernelGGL(onset_kernel, dim3((uint32_t)((b.max_nb + 255) / 256), b.n_songs), dim3(256), 0, st, b.songs,
                       w.flux, w.thresholded);
}

// ------------------------------------------------------------------------------------------------
// vec_max_elem (src/aubio.rs:787-799): last index attaining the maximum, scanning with tmp = 0 (so a
// vector with no element >= 0 yields index 0).  One wavefront; element i = lane + 64 q is x[q].
// Returns (value, index) with value = -1 when no element is >= 0.
template <int Q>
__device__ __forceinline__ void wave_argmax_last(const float (&x)[Q], float* val, int* idx) {
    const int lane = lane_id();
    float best = -1.0f;  // "no candidate" marker: candidates are >= 0
    int bidx = 0;
#pragma unroll
    for (int q = 0; q < Q; q++) {
        if (x[q] >= 0.0f && x[q] >= best) { best = x[q]; bidx = lane + 64 * q; }  // index increases: keeps the last among equals
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off, WAVE);
        const int oi = __shfl_xor(bidx, off, WAVE);
        if (ov > best || (ov == best && oi > bidx)) { best = ov; bidx = oi; }
    }
    *val = best;
    *idx = (best < 0.0f) ? 0 : bidx;
}

// vec_quadratic_peak_pos (src/aubio.rs:576-604)
__device__ __forceinline__ float quad_peak_pos(const float* x, int len, int pos) {
    if (pos == 0 || pos >= len - 1) return (float)pos;
    const float s0 = x[pos - 1], s1 = x[pos], s2 = x[pos + 1];
    return (float)pos + 0.5f * (s0 - s2) / (s0 - 2.0f * s1 + s2);
}

// get_timesig (src/aubio.rs:864-907), acflen = 512
__device__ float bt_timesig(const float* acf, long gp) {
    if (gp < 2) return 4;
    float three = 0.0f, four = 0.0f;
    const long acflen = BT_WINLEN;
    if (acflen > 6 * gp + 2) {
        for (long k = -2; k < 2; k++) { three += acf[3 * gp + k]; four += acf[4 * gp + k]; }
    } else {
        for (long k = -2; k < 2; k++) {
            const long i3 = 3 * gp + k, i6 = 6 * gp + k, i4 = 4 * gp + k, i2 = 2 * gp + k;
            if (i3 < acflen && i6 < acflen) three += acf[i3] + acf[i6];
            else if (i3 < acflen) three += acf[i3];
            if (i4 < acflen && i2 < acflen) four += acf[i4] + acf[i2];
            else if (i4 < acflen) four += acf[i4];
        }
    }
    return three > four ? 3 : 4;
}

// Per-run record written by beat_acf_kernel: everything BeatTracking::do_ derives from the run's frame alone.
//   [0, 128)   g3: unweighted comb sums for time signature 3 (the gwv path of checkstate, :1110-1124)
//   [128, 256) g4: the same for time signature 4
//   [256] rp3  [257] rp4 : Rayleigh-weighted period with numelem 3 / 4 (:987-1013)
//   [258] ts3  [259] ts4 : get_timesig(acf, rp3 / rp4), what checkstate would set when it locks onto that period
constexpr int BT_PRE_G3 = 0, BT_PRE_G4 = BT_LAGLEN, BT_PRE_RP3 = 256, BT_PRE_RP4 = 257, BT_PRE_TS3 = 258, BT_PRE_TS4 = 259;
static_assert(BT_PRE_STRIDE >= 260, "per-run record");

typedef float bt_f2 __attribute__((ext_vector_type(2)));

// acc01 += f[i] * (w[i], w[i+1]), acc23 += f[i] * (w[i+2], w[i+3]) for 16 consecutive i: the four lags (L .. L + 3) of
// one thread, w = frame + L.  The packed multiply / add takes two lags per instruction; w1 is the same series one element
// on (a second copy in LDS), so that the odd steps' operand pairs also arrive in even-aligned register pairs (gfx950 takes
// 64-bit operands from aligned pairs only, and a copy per step costs more than the LDS read).  Four lags per thread
// instead of two halve the LDS bytes per multiply-add: 18 + 18 values feed 64 of them.
template <typename F>
__device__ __forceinline__ void acf_block16(bt_f2& acc01, bt_f2& acc23, const float* w, const float* w1, F&& f_at) {
    float r[18], r1[18];
#pragma unroll
    for (int u = 0; u < 18; u++) { r[u] = w[u]; r1[u] = w1[u]; }
#pragma unroll
    for (int u = 0; u < 16; u += 2) {
        const float f0 = f_at(u), f1 = f_at(u + 1);
        const bt_f2 fv0 = {f0, f0}, fv1 = {f1, f1};
        const bt_f2 e01 = {r[u], r[u + 1]}, e23 = {r[u + 2], r[u + 3]};      // step u:     w[u .. u+3]
        const bt_f2 o01 = {r1[u], r1[u + 1]}, o23 = {r1[u + 2], r1[u + 3]};  // step u + 1: w[u+1 .. u+4]
        acc01 = acc01 + fv0 * e01;
        acc23 = acc23 + fv0 * e23;
        acc01 = acc01 + fv1 * o01;
        acc23 = acc23 + fv1 * o23;
    }
}

__global__ __launch_bounds__(64) void beat_acf_kernel(const SongDesc* __restrict__ songs,
                                                      const float* __restrict__ thresholded,
                                                      const float* __restrict__ rwv_tab,
                                                      float* __restrict__ pre_all) {
    __shared__ float df[2 * BT_WINLEN];  // [512, 1024) stays zero: the ACF loops read past the frame instead of predicating
    __shared__ float df1[2 * BT_WINLEN];  // df1[j] = df[j + 1]
    __shared__ float acf[BT_WINLEN];
    __shared__ float acfout[2][BT_LAGLEN];
    const uint32_t s = blockIdx.y;
    const long m = blockIdx.x;
    const SongDesc sd = songs[s];
    if (!sd.ok) return;
    const long n_b = sd.n_b;
    const long n_runs = (n_b >= BT_STEP) ? (n_b - BT_STEP) / BT_STEP + 1 : 0;
    if (m >= n_runs) return;
    const int tid = threadIdx.x;  // one wavefront per run
    const float* thr = thresholded + sd.b_off;
    float* pre = pre_all + ((size_t)(sd.b_off / BT_STEP) + s + (size_t)m) * BT_PRE_STRIDE;

    // ---- dfframe for run m: s[128(m+1)-512+i], s[x] = 0 for x <= 0, thr[x-1] otherwise (Tempo::do_ :1389-1416) ----
    {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {  // all eight loads in flight together; the address is clamped instead of branching
            const long xi = 128 * (m + 1) - 512 + tid + 64 * q;
            const float x = thr[xi > 0 ? xi - 1 : 0];
            v[q] = (xi <= 0) ? 0.0f : x;
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int i = tid + 64 * q;
            df[i] = v[q];
            if (i > 0) df1[i - 1] = v[q];
            df[BT_WINLEN + i] = 0.0f;
            df1[BT_WINLEN - 1 + i] = 0.0f;
        }
    }
    __syncthreads();
    // vec_autocorr (:819-828): acf[L] = (sum_{j=L}^{511} f[j-L] f[j]) / (512 - L), summed in j order.  A thread owns
    // the lag quads (4t .. 4t+3) and (508-4t .. 511-4t) -- a long and a short one -- and runs i = j - L over the four
    // lags of a quad at once (packed multiply, packed add: two lags per instruction).  The first factor f[i] is
    // wave-uniform, so for runs whose frame lies inside the song it comes from SCALAR loads of the thresholded series
    // (the kernel is bound by LDS bandwidth; a broadcast LDS read of f[i] would add to it).  The loops run to the
    // wave-uniform bounds 512 and 256 (the smallest lag of each kind is 0 and 256) and read the zero padding behind the
    // frame instead of predicating (x + f * 0 == x exactly, so every sum is bit-identical to the reference's).
    {
        const int la = 4 * tid, lb = BT_WINLEN - 4 - 4 * tid;
        constexpr int na = BT_WINLEN, nb = BT_WINLEN / 2;
        bt_f2 a01 = {0.0f, 0.0f}, a23 = {0.0f, 0.0f}, b01 = {0.0f, 0.0f}, b23 = {0.0f, 0.0f};
        if (m >= 4) {
            const float* __restrict__ fs = thr + (128 * (m + 1) - 512) - 1;  // fs[i] == dfframe[i], wave-uniform
            for (int i = 0; i < na; i += 16) {
                float sv[16];
#pragma unroll
                for (int u = 0; u < 16; u++) sv[u] = fs[i + u];
                acf_block16(a01, a23, df + la + i, df1 + la + i, [&](int u) { return sv[u]; });
            }
            for (int i = 0; i < nb; i += 16) {
                float sv[16];
#pragma unroll
                for (int u = 0; u < 16; u++) sv[u] = fs[i + u];
                acf_block16(b01, b23, df + lb + i, df1 + lb + i, [&](int u) { return sv[u]; });
            }
        } else {
            for (int i = 0; i < na; i += 16) acf_block16(a01, a23, df + la + i, df1 + la + i, [&](int u) { return df[i + u]; });
            for (int i = 0; i < nb; i += 16) acf_block16(b01, b23, df + lb + i, df1 + lb + i, [&](int u) { return df[i + u]; });
        }
        acf[la] = a01.x / (float)(BT_WINLEN - la);
        acf[la + 1] = a01.y / (float)(BT_WINLEN - la - 1);
        acf[la + 2] = a23.x / (float)(BT_WINLEN - la - 2);
        acf[la + 3] = a23.y / (float)(BT_WINLEN - la - 3);
        acf[lb] = b01.x / (float)(BT_WINLEN - lb);
        acf[lb + 1] = b01.y / (float)(BT_WINLEN - lb - 1);
        acf[lb + 2] = b23.x / (float)(BT_WINLEN - lb - 2);
        acf[lb + 3] = b23.y / (float)(BT_WINLEN - lb - 3);
    }
    __syncthreads();
    // shift-invariant comb filterbank (:987-1003): the sums run a = 1 .. numelem in order, so the numelem = 4 value is
    // the numelem = 3 value continued.  Rayleigh path: terms divided by 2a - 1, weighted by rwv; the unweighted sums
    // are what checkstate weights by its Gaussian (:1110-1124).
#pragma unroll
    for (int half = 0; half < 2; half++) {
        const int l = tid + 64 * half;
        float v3 = 0.0f, v4 = 0.0f, g3 = 0.0f, g4 = 0.0f;
        if (l >= 1 && l < BT_LAGLEN - 1) {
            float v = 0.0f, g = 0.0f;
            for (int a = 1; a <= 4; a++) {
                if (a == 4) { v3 = v; g3 = g; }
                for (int b = 1; b < 2 * a; b++) {
                    const int idx = l * a + b - 1;
                    if (idx < BT_WINLEN) {
                        v += acf[idx] / (2.0f * (float)a - 1.0f);
                        g += acf[idx];
                    }
                }
            }
            v4 = v; g4 = g;
        }
        const float r = rwv_tab[l];
        acfout[0][l] = v3 * r;
        acfout[1][l] = v4 * r;
        pre[BT_PRE_G3 + l] = g3;
        pre[BT_PRE_G4 + l] = g4;
    }
    __syncthreads();
#pragma unroll
    for (int which = 0; which < 2; which++) {
        const float x[2] = {acfout[which][tid], acfout[which][tid + 64]};
        float best;
        int maxindex;
        wave_argmax_last(x, &best, &maxindex);
        if (tid == 0) {
            const int rayparam = 43;  // (60*22050/120/256) as u32
            const float rp = (maxindex > 0 && maxindex < BT_LAGLEN - 1) ? quad_peak_pos(acfout[which], BT_LAGLEN, maxindex)
                                                                       : (float)rayparam;
            pre[BT_PRE_RP3 + which] = rp;
            pre[BT_PRE_TS3 + which] = bt_timesig(acf, (long)rp);
        }
    }
}

struct BtShared {
    float dfrev[BT_WINLEN];
    float acfout[BT_LAGLEN];
    float phout[BT_LAGLEN + 1];  // [128] stays zero: phout is zero from the beat period on (bp < 128)
    float out[BT_STEP];
};

__global__ __launch_bounds__(64) void beat_track_kernel(const SongDesc* __restrict__ songs,
                                                        const float* __restrict__ thresholded,
                                                        const float* __restrict__ e256,
                                                        const float* __restrict__ dfwv_tab,
                                                        const float* __restrict__ pre_all,
                                                        float* __restrict__ run_bpm, uint32_t* __restrict__ run_cnt,
                                                        uint32_t runs_pitch, TempoState* __restrict__ tempo_out) {
    __shared__ BtShared sh;
    const uint32_t s = blockIdx.x;
    const SongDesc sd = songs[s];
    const int lane = threadIdx.x;
    if (!sd.ok) {
        if (lane == 0) { tempo_out[s].tempo = -1.0f; tempo_out[s].n_bpms = 0; }
        return;
    }
    const float* thr = thresholded + sd.b_off;
    const float* en = e256 + sd.e_off;
    const float* pre_song = pre_all + ((size_t)(sd.b_off / BT_STEP) + s) * BT_PRE_STRIDE;
    float* rbpm = run_bpm + (size_t)s * runs_pitch;
    uint32_t* rcnt = run_cnt + (size_t)s * runs_pitch;

    float dfwv[8];
#pragma unroll
    for (int q = 0; q < 8; q++) dfwv[q] = dfwv_tab[lane + 64 * q];
    float gwv[2] = {0.0f, 0.0f};
    if (lane == 0) sh.phout[BT_LAGLEN] = 0.0f;
    // scalar state: every lane carries the same values
    float gp_state = 0.0f, rp1 = 0.0f, rp2 = 0.0f, lastbeat = 0.0f;
    int counter = 0, flagstep = 0, timesig = 0;

    const float g_var = 3.901f;
    const long n_b = sd.n_b;
    long n_runs = (n_b >= BT_STEP) ? (n_b - BT_STEP) / BT_STEP + 1 : 0;  // runs happen at tempo frames 127 + 128 m
    if (n_runs > (long)runs_pitch) n_runs = (long)runs_pitch;

    // everything a run reads from memory is independent of the tracker's state: fetched one run ahead
    struct RunIn {
        float f[8];         // dfframe[lane + 64 q]
        float g3[2], g4[2];  // comb sums at lags lane, lane + 64
        float rp3, rp4, ts3, ts4;
        float e[2], e1[2];   // 256-sample energies of tempo frames 127 + 128 m + (lane, lane + 64) and the following block
    };
    auto fetch = [&](long m, RunIn& r) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const long xi = 128 * (m + 1) - 512 + lane + 64 * q;
            r.f[q] = (xi <= 0) ? 0.0f : thr[xi - 1];
        }
        const float* pre = pre_song + (size_t)m * BT_PRE_STRIDE;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            r.g3[q] = pre[BT_PRE_G3 + lane + 64 * q];
            r.g4[q] = pre[BT_PRE_G4 + lane + 64 * q];
            const long t = 127 + 128 * m + lane + 64 * q;
            r.e[q] = t < n_b ? en[t] : 1.0f;
            r.e1[q] = t < n_b ? en[t + 1] : 1.0f;
        }
        r.rp3 = pre[BT_PRE_RP3]; r.rp4 = pre[BT_PRE_RP4]; r.ts3 = pre[BT_PRE_TS3]; r.ts4 = pre[BT_PRE_TS4];
    };
    RunIn cur, nxt;
    if (n_runs > 0) fetch(0, cur);

    for (long m = 0; m < n_runs; m++) {
        if (m + 1 < n_runs) fetch(m + 1, nxt);
        // dfrev = reverse(dfframe * dfwv) (:975-980)
#pragma unroll
        for (int q = 0; q < 8; q++) sh.dfrev[BT_WINLEN - 1 - (lane + 64 * q)] = cur.f[q] * dfwv[q];
        // the Rayleigh-weighted period for the current numelem (:983-1013)
        const float rp = (timesig == 3) ? cur.rp3 : cur.rp4;

        // ---- checkstate (:1096-1227) ----
        float gp = gp_state;
        if (gp > 0.0f) {  // uniform; implies timesig in {3, 4}
            const float x[2] = {(timesig == 3 ? cur.g3[0] : cur.g4[0]) * gwv[0], (timesig == 3 ? cur.g3[1] : cur.g4[1]) * gwv[1]};
            sh.acfout[lane] = x[0];
            sh.acfout[lane + 64] = x[1];
            float best;
            int maxindex;
            wave_argmax_last(x, &best, &maxindex);
            __syncthreads();
            gp = quad_peak_pos(sh.acfout, BT_LAGLEN, maxindex);
        } else {
            gp = 0.0f;
        }
        int flagconst = 0;
        if (counter == 0) {
            if (fabsf(gp - rp) > 2.0f * g_var) { flagstep = 1; counter = 3; }
            else flagstep = 0;
        }
        if (counter == 1 && flagstep == 1) {
            if (fabsf(2.0f * rp - rp1 - rp2) < g_var) { flagconst = 1; counter = 0; }
            else { flagconst = 0; counter = 2; }
        } else if (counter > 0) {
            counter -= 1;
        }
        rp2 = rp1;
        rp1 = rp;
        float bp;
        int phw_mode;  // 0 = flat, 1 = gaussian
        if (flagconst) {
            gp = rp;
            timesig = (int)((timesig == 3) ? cur.ts3 : cur.ts4);  // get_timesig(acf, gp) with gp = this run's rp
            bp = gp;
            phw_mode = 0;
        } else if (timesig > 0) {
            bp = gp;
            phw_mode = ((float)BT_STEP > lastbeat) ? 1 : 0;
        } else {
            bp = rp;
            phw_mode = 0;
        }
        const float bpw = bp;
        while (bp > 0.0f && bp < 25.0f) bp *= 2.0f;
        gp_state = gp;
        if (flagconst) {
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const float diff = (float)(lane + 64 * q + 1) - gp;
                gwv[q] = exp_f32(-0.5f * diff * diff / (g_var * g_var));
            }
        }

        uint32_t nbeats = 0;
        if (bp != 0.0f) {  // uniform
            // beat phase (:1025-1054): phout[i] is non-zero only for i < bp (< 128), where phwv applies
            const int kmax = (int)floorf((float)BT_WINLEN / bp);
            __syncthreads();  // dfrev is complete
            float ph[2];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int i = lane + 64 * q;
                float v = 0.0f;
                if ((float)i < bp) {
                    for (int k = 0; k < kmax; k++) {
                        const int idx = i + (int)floorf((bp * (float)k) + 0.5f);
                        if (idx < BT_WINLEN) v += sh.dfrev[idx];
                    }
                }
                float w = 1.0f;
                if (phw_mode == 1) {
                    const float diff = 1.0f + (float)i - (float)BT_STEP + lastbeat;
                    w = exp_f32(-0.5f * diff * diff / (bpw / 8.0f));
                }
                ph[q] = v * w;
                sh.phout[i] = ph[q];
            }
            float best;
            int maxindex;
            wave_argmax_last(ph, &best, &maxindex);
            // the frame's remaining entries (128..511) are zero: the last of them wins unless an earlier one is positive
            if (!(best > 0.0f)) maxindex = BT_WINLEN - 1;
            __syncthreads();
            float phase;
            if (maxindex >= BT_WINLEN - 1) phase = (float)BT_STEP - lastbeat;
            else phase = quad_peak_pos(sh.phout, BT_WINLEN, maxindex);  // maxindex < 128: reads at most phout[128] == 0
            phase += 1.0f;
            int i = 1;
            float beat = bp - phase;
            if (((float)BT_STEP - lastbeat - phase) < -0.40f * bp) beat += bp;
            while (beat + bp < 0.0f) beat += bp;
            if (beat >= 0.0f && i < BT_STEP) { if (lane == 0) sh.out[i] = beat; i++; }
            while (beat + bp <= (float)BT_STEP && i < BT_STEP) { beat += bp; if (lane == 0) sh.out[i] = beat; i++; }
            lastbeat = beat;
            nbeats = (uint32_t)i;
            __syncthreads();
        }

        // ---- frames 127+128m .. 127+128m+127 use out/bp of this run (Tempo::do_ :1418-1438) ----
        uint32_t hits = 0;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int j = lane + 64 * q;  // blockpos
            const long t = 127 + 128 * m + j;
            float tempo = 0.0f;
            if (t < n_b) {
                for (uint32_t i = 1; i < nbeats; i++) {
                    const float beat_pos = sh.out[i];
                    if (j == (int)floorf(beat_pos)) {
                        tempo = beat_pos - floorf(beat_pos);
                        // is_silence over the 512-sample window analyze() passed: x[256t, 256t+512),
                        // i.e. the 256-sample energy blocks t and t+1
                        const float level = (cur.e[q] + cur.e1[q]) / 512.0f;
                        if (10.0f * log10f(level) < -90.0f) tempo = 0.0f;
                    }
                }
            }
            hits += (uint32_t)__popcll(__ballot(tempo > 0.0f));
        }
        if (lane == 0) {
            // BeatTracking::get_bpm (:1231-1239)
            float bpm = 0.0f;
            if (bp != 0.0f) {
                const float period_samples = (float)HOP_B * bp;
                const float period_s = period_samples / (float)SAMPLE_RATE;
                bpm = 60.0f / period_s;
            }
            rbpm[m] = bpm;
            rcnt[m] = hits;
        }
        __syncthreads();  // out / dfrev / phout are rewritten by the next run
        cur = nxt;
    }
    __threadfence_block();
    __syncthreads();

    // ---- BPMDesc::get_value (src/temporal.rs:66-77): Midpoint median of the pushed bpms ----
    const long runs = n_runs;
    uint32_t tot = 0;
    for (long m = lane; m < runs; m += WAVE) tot += rcnt[m];
    const uint32_t total = wave_sum(tot);
    if (total == 0) {
        if (lane == 0) { tempo_out[s].tempo = -1.0f; tempo_out[s].n_bpms = 0; }
        return;
    }
    const uint32_t r_lo = (total - 1) / 2, r_hi = (total - 1) - r_lo == r_lo ? r_lo : r_lo + 1;
    // equal values yield the same number, so "any lane that holds the order statistic" is well defined
    float lo = 0.0f, hi = 0.0f;
    bool has_lo = false, has_hi = false;
    for (long m = lane; m < runs; m += WAVE) {
        const uint32_t c = rcnt[m];
        if (c == 0) continue;
        const float v = rbpm[m];
        uint32_t less = 0, leq = 0;
        for (long q = 0; q < runs; q++) {
            const uint32_t cq = rcnt[q];
            if (cq == 0) continue;
            const float vq = rbpm[q];
            if (vq < v) less += cq;
            if (vq <= v) leq += cq;
        }
        if (less <= r_lo && r_lo < leq) { lo = v; has_lo = true; }
        if (less <= r_hi && r_hi < leq) { hi = v; has_hi = true; }
    }
    const uint64_t m_lo = __ballot(has_lo), m_hi = __ballot(has_hi);
    const float s_lo = m_lo ? __shfl(lo, __ffsll((unsigned long long)m_lo) - 1, WAVE) : 0.0f;
    const float s_hi = m_hi ? __shfl(hi, __ffsll((unsigned long long)m_hi) - 1, WAVE) : 0.0f;
    if (lane == 0) {
        const float median = s_lo + (s_hi - s_lo) / 2.0f;
        tempo_out[s].tempo = 2.0f * (median - 0.0f) / (206.0f - 0.0f) - 1.0f;
        tempo_out[s].n_bpms = total;
    }
}

void launch_beat_acf(const Batch& b, const Workspace& w, const DeviceTables& t, hipStream_t st) {
    if (b.n_songs == 0) return;
    const uint32_t max_runs = b.max_nb >= (uint32_t)BT_STEP ? (b.max_nb - BT_STEP) / BT_STEP + 1 : 0;
    if (max_runs > 0)
        hipLaunchKernelGGL(beat_acf_kernel, dim3(max_runs, b.n_songs), dim3(64), 0, st, b.songs, w.thresholded, t.bt_rwv,
                           w.bt_pre);
}

void launch_beat_track(const Batch& b, const Workspace& w, const DeviceTables& t, hipStream_t st) {
    if (b.n_songs == 0) return;
    hipLaunchKernelGGL(beat_track_kernel, dim3(b.n_songs), dim3(64), 0, st, b.songs, w.thresholded, w.e256, t.bt_dfwv,
                       w.bt_pre, w.run_bpm, w.run_cnt, w.runs_pitch, w.tempo);
}

void launch_beat(const Batch& b, const Workspace& w, const DeviceTables& t, hipStream_t st) {
    launch_beat_acf(b, w, t, st);
    launch_beat_track(b, w, t, st);
}

}  // namespace bg
