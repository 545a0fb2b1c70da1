// Guard-band probe (developer tool): how close to a pitch-bin boundary may the f32 classification of a peak
// (peak_classify in kernels_chroma.hip) be trusted?  Random established peaks (sb < se >= sa, centre bins 57..1483, from
// flat to sharp) are classified in f32 with several guard widths and flatness limits and compared with the f64 path
// (pip_peak_core + pitch_bin, the reference's arithmetic).  Prints, per setting, the share of peaks sent to the f64 path
// and the number of peaks whose f32 bin differs from the f64 bin (must be 0 with a wide margin).
// build (from the repo root): hipcc --offload-arch=gfx950 -O3 -std=c++17 -I bliss-rs_amd/csrc -I include -o tests/tools/probes/guard_probe tests/tools/probes/guard_probe.hip
#include "kernels_chroma.hip"
#include <stdio.h>
#include <stdlib.h>
using namespace bg;
__device__ __forceinline__ uint32_t rng(uint64_t& s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); }
__device__ __forceinline__ int pitch_bin_f32_guard(float sb, float se, float sa, int c, float guard, float flat_limit) {
    const float avg = 0.5f * (sa - sb);
    const float den = (2.0f * se - sa) - sb;
    if (!(se >= 1e-30f) || den < se * flat_limit) return -1;
    const float shift = avg * __builtin_amdgcn_rcpf(den);
    float x = 12.0f * (__builtin_amdgcn_logf((float)c + shift) + -3.3528687f);
    x = x - truncf(x);
    if (x >= 0.5f) x -= 1.0f;
    const float q = (x + 0.5f) * 100.0f;
    const float fl = floorf(q), fr = q - fl;
    if (fr < guard || fr > 1.0f - guard) return -1;
    const int idx = (int)fl;
    return idx < 0 ? 0 : (idx > N_TUNING - 1 ? N_TUNING - 1 : idx);
}
constexpr int NG = 6;  // setting 0 = the production classifier itself (peak_pitch_bin_f32 with PITCH_GUARD / PITCH_FLAT_LIMIT)
__constant__ float GUARD[NG] = {PITCH_GUARD, 0.02f, 0.004f, 0.002f, 0.001f, 0.0005f};
__constant__ float FLAT[NG] = {PITCH_FLAT_LIMIT, 1.0f / 1024, 1.0f / 256, 1.0f / 256, 1.0f / 256, 1.0f / 256};
__global__ void probe(unsigned long long* slow, unsigned long long* bad, int iters, uint64_t seed) {
    uint64_t s = seed + 0x9E3779B97F4A7C15ull * (blockIdx.x * 256 + threadIdx.x + 1);
    unsigned long long lslow[NG] = {}, lbad[NG] = {};
    for (int it = 0; it < iters; it++) {
        const int c = 57 + (int)(rng(s) % (1483 - 57 + 1));
        const float se = ldexpf(1.0f + (float)(rng(s) & 0x7FFFFF) * 1.1920929e-7f, (int)(rng(s) % 24) - 8);
        // neighbours: a fraction of se, from nearly equal (flat peak) to far below
        const int mode = rng(s) % 4;
        float fa = (float)(rng(s) & 0xFFFFFF) * 5.9604645e-8f, fb = (float)(rng(s) & 0xFFFFFF) * 5.9604645e-8f;
        if (mode == 1) { fa = 1.0f - fa * 0.01f; fb = 1.0f - fb * 0.01f; }           // flat
        if (mode == 2) { fa = 1.0f - fa * 1e-4f; fb = 1.0f - fb * 1e-4f; }           // very flat
        const float sa = fminf(se, se * fa);
        float sb = se * fb;
        if (!(sb < se)) sb = se * 0.99999f;
        if (!(sa <= se && sb < se)) continue;
        const double ref = 0.0;
        double mag, pitch;
        if (!pip_peak_core(sb, se, sa, ref, c, &mag, &pitch)) continue;
        const int exact = pitch_bin(pitch);
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const int pb = g == 0 ? peak_pitch_bin_f32(sb, se, sa, c) : pitch_bin_f32_guard(sb, se, sa, c, GUARD[g], FLAT[g]);
            if (pb < 0) lslow[g]++;
            else if (pb != exact) lbad[g]++;
        }
    }
    for (int g = 0; g < NG; g++) { atomicAdd(&slow[g], lslow[g]); atomicAdd(&bad[g], lbad[g]); }
}
int main(int argc, char** argv) {
    unsigned long long *slow, *bad, hs[NG], hb[NG];
    (void)hipMalloc(&slow, NG * 8); (void)hipMalloc(&bad, NG * 8);
    (void)hipMemset(slow, 0, NG * 8); (void)hipMemset(bad, 0, NG * 8);
    const int blocks = 4096, iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, slow, bad, iters, 12345ull);
    (void)hipMemcpy(hs, slow, NG * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(hb, bad, NG * 8, hipMemcpyDeviceToHost);
    const double total = (double)blocks * 256 * iters;
    const float G[NG] = {PITCH_GUARD, 0.02f, 0.004f, 0.002f, 0.001f, 0.0005f};
    const float F[NG] = {PITCH_FLAT_LIMIT, 1.0f / 1024, 1.0f / 256, 1.0f / 256, 1.0f / 256, 1.0f / 256};
    for (int g = 0; g < NG; g++)
        printf("%s guard %.4f, den >= se / %4.0f: %.2f %% of ~%.1e peaks to the f64 path, %llu f32 bins differ from the f64 bin\n",
               g == 0 ? "production" : "variant   ", G[g], 1.0 / F[g], 100.0 * hs[g] / total, total, hb[g]);
    return 0;
}
