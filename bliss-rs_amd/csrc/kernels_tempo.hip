// kernels_tempo.hip -- aubio-style tempo on the device (compiled with -ffp-contract=off so every
// f32 operation rounds exactly like the reference's scalar Rust).
//
//   onset_kernel : PeakPicker threshold (src/aubio.rs:733-768, 661-685, 482-554): a pure function of
//                  the last 7 SpecFlux values, so one thread per tempo frame.
//   beat_kernel  : Tempo::do_ (src/aubio.rs:1378-1443) + BeatTracking::{do_, checkstate, get_timesig,
//                  get_bpm} (:966-1240) + BPMDesc (src/temporal.rs:50-77).  The beat tracker is the one
//                  sequential chain on the path (state gp/rp1/rp2/counter/flagstep/timesig/lastbeat/
//                  gwv/phwv), so one workgroup per song walks its ~121 runs while songs run in parallel.
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
// vec_max_elem (src/aubio.rs:787-799): last index attaining the maximum, scanning with tmp = 0
// (so a vector with no element >= 0 yields index 0).  Block-wide; result broadcast through LDS.
__device__ __forceinline__ int block_argmax_last(const float* v, int n, float* s_val, int* s_idx) {
    const int tid = threadIdx.x;
    float best = -1.0f;  // "no candidate" marker: candidates are >= 0
    int bidx = 0;
    for (int i = tid; i < n; i += 256) {
        const float x = v[i];
        if (x >= 0.0f && x >= best) { best = x; bidx = i; }  // i increases: keeps the last among equals
    }
    // wave reduce (value, then larger index)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off, WAVE);
        const int oi = __shfl_xor(bidx, off, WAVE);
        if (ov > best || (ov == best && oi > bidx)) { best = ov; bidx = oi; }
    }
    __syncthreads();
    if (lane_id() == 0) { s_val[wave_id()] = best; s_idx[wave_id()] = bidx; }
    __syncthreads();
    float b = s_val[0];
    int bi = s_idx[0];
#pragma unroll
    for (int w = 1; w < 4; w++) {
        if (s_val[w] > b || (s_val[w] == b && s_idx[w] > bi)) { b = s_val[w]; bi = s_idx[w]; }
    }
    return (b < 0.0f) ? 0 : bi;
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

struct BtShared {
    float dfframe[2 * BT_WINLEN];  // [512, 1024) stays zero: the ACF loops below read past the frame instead of predicating
    float dfrev[BT_WINLEN], acf[BT_WINLEN], phout[BT_WINLEN], dfwv[BT_WINLEN];
    float acfout[BT_LAGLEN], rwv[BT_LAGLEN], gwv[BT_LAGLEN], out[BT_STEP];
    float phwv[2 * BT_LAGLEN];
    float red_val[4];
    int red_idx[4];
    // scalar state (written by thread 0, read by all after a barrier)
    float rp, gp, bp, rp1, rp2, lastbeat;
    int counter, flagstep, timesig, flagconst, phw_mode;
    uint32_t hit_count;
};


__global__ __launch_bounds__(256) void beat_kernel(const SongDesc* __restrict__ songs,
                                                   const float* __restrict__ thresholded,
                                                   const float* __restrict__ e256,
                                                   const float* __restrict__ rwv_tab,
                                                   const float* __restrict__ dfwv_tab,
                                                   float* __restrict__ run_bpm, uint32_t* __restrict__ run_cnt,
                                                   uint32_t runs_pitch, TempoState* __restrict__ tempo_out) {
    __shared__ BtShared sh;
    const uint32_t s = blockIdx.x;
    const SongDesc sd = songs[s];
    const int tid = threadIdx.x;
    if (!sd.ok) {
        if (tid == 0) { tempo_out[s].tempo = -1.0f; tempo_out[s].n_bpms = 0; }
        return;
    }
    const float* thr = thresholded + sd.b_off;
    const float* en = e256 + sd.e_off;
    float* rbpm = run_bpm + (size_t)s * runs_pitch;
    uint32_t* rcnt = run_cnt + (size_t)s * runs_pitch;

    for (int i = tid; i < BT_WINLEN; i += 256) { sh.dfwv[i] = dfwv_tab[i]; sh.dfframe[BT_WINLEN + i] = 0.0f; }
    for (int i = tid; i < BT_LAGLEN; i += 256) { sh.rwv[i] = rwv_tab[i]; sh.gwv[i] = 0.0f; }
    for (int i = tid; i < 2 * BT_LAGLEN; i += 256) sh.phwv[i] = 1.0f;
    if (tid == 0) {
        sh.rp = 1.0f; sh.gp = 0.0f; sh.bp = 0.0f; sh.rp1 = 0.0f; sh.rp2 = 0.0f; sh.lastbeat = 0.0f;
        sh.counter = 0; sh.flagstep = 0; sh.timesig = 0;
    }
    __syncthreads();

    const float g_var = 3.901f;
    const int rayparam = 43;  // (60*22050/120/256) as u32
    // beat-tracker runs happen at tempo frames 127 + 128*m
    const long n_b = sd.n_b;
    const long n_runs = (n_b >= BT_STEP) ? (n_b - BT_STEP) / BT_STEP + 1 : 0;

    for (long m = 0; m < n_runs && m < (long)runs_pitch; m++) {
        // ---- dfframe for run m: s[128(m+1)-512+i], s[x] = 0 for x <= 0, thr[x-1] otherwise ----
        for (int i = tid; i < BT_WINLEN; i += 256) {
            const long xi = 128 * (m + 1) - 512 + i;
            sh.dfframe[i] = (xi <= 0) ? 0.0f : thr[xi - 1];
        }
        __syncthreads();
        // dfrev = reverse(dfframe * dfwv)
        for (int i = tid; i < BT_WINLEN; i += 256) sh.dfrev[BT_WINLEN - 1 - i] = sh.dfframe[i] * sh.dfwv[i];
        // vec_autocorr (:819-828): acf[L] = (sum_{j=L}^{511} f[j-L] f[j]) / (512 - L), summed in j order.  Thread t computes
        // lags t and 511-t (513 products in total).  Written over i = j - L the first factor f[i] is the same for every
        // lane, so for runs whose frame lies inside the song it comes from SCALAR loads of the thresholded series and
        // only f[i + L] is an LDS read; the loops run to the wave-uniform bound and read the zero padding behind the
        // frame instead of predicating (x + f[i] * 0 == x exactly, so the sums are bit-identical).
        if (m >= 4) {
            const float* __restrict__ fs = thr + (128 * (m + 1) - 512) - 1;  // fs[i] == dfframe[i], wave-uniform
            const int wave = tid >> 6;
            {
                const float* va = sh.dfframe + tid;  // va[i] = f[i + lag], lag = tid
                float acc = 0.0f;
                const int na = BT_WINLEN - 64 * wave;
                for (int i = 0; i < na; i += 16) {
                    float sv[16];
#pragma unroll
                    for (int u = 0; u < 16; u++) sv[u] = fs[i + u];
#pragma unroll
                    for (int u = 0; u < 16; u++) acc += sv[u] * va[i + u];
                }
                sh.acf[tid] = acc / (float)(BT_WINLEN - tid);
            }
            {
                const int lag = BT_WINLEN - 1 - tid;
                const float* vb = sh.dfframe + lag;
                float acc = 0.0f;
                const int nb = 64 * wave + 64;
                for (int i = 0; i < nb; i += 16) {
                    float sv[16];
#pragma unroll
                    for (int u = 0; u < 16; u++) sv[u] = fs[i + u];
#pragma unroll
                    for (int u = 0; u < 16; u++) acc += sv[u] * vb[i + u];
                }
                sh.acf[lag] = acc / (float)(BT_WINLEN - lag);
            }
        } else {
            const int lag_a = tid, lag_b = BT_WINLEN - 1 - tid;
            float acc = 0.0f;
            for (int j = lag_a; j < BT_WINLEN; j++) acc += sh.dfframe[j - lag_a] * sh.dfframe[j];
            sh.acf[lag_a] = acc / (float)(BT_WINLEN - lag_a);
            acc = 0.0f;
            for (int j = lag_b; j < BT_WINLEN; j++) acc += sh.dfframe[j - lag_b] * sh.dfframe[j];
            sh.acf[lag_b] = acc / (float)(BT_WINLEN - lag_b);
        }
        __syncthreads();
        // shift-invariant comb filterbank (:987-1003)
        const int numelem = (sh.timesig == 0) ? 4 : sh.timesig;
        if (tid < BT_LAGLEN) {
            float v = 0.0f;
            if (tid >= 1 && tid < BT_LAGLEN - 1) {
                for (int a = 1; a <= numelem; a++)
                    for (int b = 1; b < 2 * a; b++) {
                        const int idx = tid * a + b - 1;
                        if (idx < BT_WINLEN) v += sh.acf[idx] / (2.0f * (float)a - 1.0f);
                    }
            }
            sh.acfout[tid] = v * sh.rwv[tid];
        }
        __syncthreads();
        int maxindex = block_argmax_last(sh.acfout, BT_LAGLEN, sh.red_val, sh.red_idx);
        if (tid == 0) {
            if (maxindex > 0 && maxindex < BT_LAGLEN - 1) sh.rp = quad_peak_pos(sh.acfout, BT_LAGLEN, maxindex);
            else sh.rp = (float)rayparam;
        }
        __syncthreads();

        // ---- checkstate (:1096-1227) ----
        float gp = sh.gp;
        if (gp > 0.0f) {  // uniform
            if (tid < BT_LAGLEN) {
                float v = 0.0f;
                if (tid >= 1 && tid < BT_LAGLEN - 1) {
                    for (int a = 1; a <= sh.timesig; a++)
                        for (int b = 1; b < 2 * a; b++) {
                            const int idx = tid * a + b - 1;
                            if (idx < BT_WINLEN) v += sh.acf[idx];
                        }
                }
                sh.acfout[tid] = v * sh.gwv[tid];
            }
            __syncthreads();
            maxindex = block_argmax_last(sh.acfout, BT_LAGLEN, sh.red_val, sh.red_idx);
            gp = quad_peak_pos(sh.acfout, BT_LAGLEN, maxindex);
        } else {
            gp = 0.0f;
        }
        __syncthreads();
        if (tid == 0) {
            int counter = sh.counter, flagstep = sh.flagstep, flagconst = 0;
            const float rp = sh.rp;
            float rp1 = sh.rp1, rp2 = sh.rp2;
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
                sh.timesig = (int)bt_timesig(sh.acf, (long)gp);
                bp = gp;
                phw_mode = 0;
            } else if (sh.timesig > 0) {
                bp = gp;
                phw_mode = ((float)BT_STEP > sh.lastbeat) ? 1 : 0;
            } else {
                bp = rp;
                phw_mode = 0;
            }
            const float bp_for_phwv = bp;
            while (bp > 0.0f && bp < 25.0f) bp *= 2.0f;
            sh.counter = counter; sh.flagstep = flagstep; sh.gp = gp; sh.bp = bp; sh.rp1 = rp1; sh.rp2 = rp2;
            sh.flagconst = flagconst; sh.phw_mode = phw_mode;
            sh.red_val[0] = bp_for_phwv;
        }
        __syncthreads();
        {
            const float gpn = sh.gp, bpw = sh.red_val[0], lastbeat = sh.lastbeat;
            if (sh.flagconst && tid < BT_LAGLEN) {
                const float diff = (float)(tid + 1) - gpn;
                sh.gwv[tid] = exp_f32(-0.5f * diff * diff / (g_var * g_var));
            }
            if (tid < 2 * BT_LAGLEN) {
                if (sh.phw_mode == 1) {
                    const float diff = 1.0f + (float)tid - (float)BT_STEP + lastbeat;
                    sh.phwv[tid] = exp_f32(-0.5f * diff * diff / (bpw / 8.0f));
                } else {
                    sh.phwv[tid] = 1.0f;
                }
            }
        }
        __syncthreads();

        const float bp = sh.bp;
        uint32_t nbeats = 0;
        if (bp == 0.0f) {  // uniform
            if (tid < BT_STEP) sh.out[tid] = 0.0f;
            __syncthreads();
        } else {
            // beat phase (:1025-1054)
            const int kmax = (int)floorf((float)BT_WINLEN / bp);
            for (int i = tid; i < BT_WINLEN; i += 256) {
                float v = 0.0f;
                if ((float)i < bp) {
                    for (int k = 0; k < kmax; k++) {
                        const int idx = i + (int)floorf((bp * (float)k) + 0.5f);
                        if (idx < BT_WINLEN) v += sh.dfrev[idx];
                    }
                }
                if (i < 2 * BT_LAGLEN) v *= sh.phwv[i];
                sh.phout[i] = v;
            }
            __syncthreads();
            maxindex = block_argmax_last(sh.phout, BT_WINLEN, sh.red_val, sh.red_idx);
            if (tid < BT_STEP) sh.out[tid] = 0.0f;
            __syncthreads();
            if (tid == 0) {
                float phase;
                if (maxindex >= BT_WINLEN - 1) phase = (float)BT_STEP - sh.lastbeat;
                else phase = quad_peak_pos(sh.phout, BT_WINLEN, maxindex);
                phase += 1.0f;
                int i = 1;
                float beat = bp - phase;
                if (((float)BT_STEP - sh.lastbeat - phase) < -0.40f * bp) beat += bp;
                while (beat + bp < 0.0f) beat += bp;
                if (beat >= 0.0f && i < BT_STEP) { sh.out[i] = beat; i++; }
                while (beat + bp <= (float)BT_STEP && i < BT_STEP) { beat += bp; sh.out[i] = beat; i++; }
                sh.lastbeat = beat;
                sh.out[0] = (float)i;
            }
            __syncthreads();
            nbeats = (uint32_t)sh.out[0];
        }

        // ---- frames 127+128m .. 127+128m+127 use out/bp of this run (Tempo::do_ :1418-1438) ----
        if (tid == 0) sh.hit_count = 0;
        __syncthreads();
        if (tid < BT_STEP) {
            const long t = 127 + 128 * m + tid;  // blockpos == tid
            if (t < n_b) {
                float tempo = 0.0f;
                for (uint32_t i = 1; i < nbeats; i++) {
                    const float beat_pos = sh.out[i];
                    if (tid == (int)floorf(beat_pos)) {
                        tempo = beat_pos - floorf(beat_pos);
                        // is_silence over the 512-sample window analyze() passed: x[256t, 256t+512),
                        // i.e. the 256-sample energy blocks t and t+1
                        const float level = (en[t] + en[t + 1]) / 512.0f;
                        if (10.0f * log10f(level) < -90.0f) tempo = 0.0f;
                    }
                }
                if (tempo > 0.0f) atomicAdd(&sh.hit_count, 1u);
            }
        }
        __syncthreads();
        if (tid == 0) {
            // BeatTracking::get_bpm (:1231-1239)
            float bpm = 0.0f;
            if (bp != 0.0f) {
                const float period_samples = (float)HOP_B * bp;
                const float period_s = period_samples / (float)SAMPLE_RATE;
                bpm = 60.0f / period_s;
            }
            rbpm[m] = bpm;
            rcnt[m] = sh.hit_count;
        }
        __syncthreads();
    }

    // ---- BPMDesc::get_value (src/temporal.rs:66-77): Midpoint median of the pushed bpms ----
    __shared__ uint32_t s_total;
    __shared__ float s_lo, s_hi;
    const long runs = (n_runs < (long)runs_pitch) ? n_runs : (long)runs_pitch;
    if (tid == 0) {
        uint32_t tot = 0;
        for (long m = 0; m < runs; m++) tot += rcnt[m];
        s_total = tot;
        s_lo = 0.0f; s_hi = 0.0f;
    }
    __syncthreads();
    const uint32_t total = s_total;
    if (total == 0) {
        if (tid == 0) { tempo_out[s].tempo = -1.0f; tempo_out[s].n_bpms = 0; }
        return;
    }
    const uint32_t r_lo = (total - 1) / 2, r_hi = (total - 1) - r_lo == r_lo ? r_lo : r_lo + 1;
    for (long m = tid; m < runs; m += 256) {
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
        if (less <= r_lo && r_lo < leq) s_lo = v;  // equal values write the same number: benign
        if (less <= r_hi && r_hi < leq) s_hi = v;
    }
    __syncthreads();
    if (tid == 0) {
        const float median = s_lo + (s_hi - s_lo) / 2.0f;
        tempo_out[s].tempo = 2.0f * (median - 0.0f) / (206.0f - 0.0f) - 1.0f;
        tempo_out[s].n_bpms = total;
    }
}

void launch_beat(const Batch& b, const Workspace& w, const DeviceTables& t, hipStream_t st) {
    if (b.n_songs == 0) return;
    hipLaunchKernelGGL(beat_kernel, dim3(b.n_songs), dim3(256), 0, st, b.songs, w.thresholded, w.e256, t.bt_rwv,
                       t.bt_dfwv, w.run_bpm, w.run_cnt, w.runs_pitch, w.tempo);
}

}  // namespace bg
