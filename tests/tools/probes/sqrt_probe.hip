// sqrt_probe.hip -- how far is v_sqrt_f32 from the correctly rounded square root?  (The FFT kernels take every magnitude
// with one v_sqrt_f32; the oracle -- like the reference -- with a correctly rounded sqrtf.)
//   build: hipcc --offload-arch=gfx950 -O2 -o tests/tools/probes/sqrt_probe tests/tools/probes/sqrt_probe.hip
// Prints, over N random inputs in [2^-20, 2^20) and over a dense sweep of one binade, the share of results that differ
// from sqrtf by 0 / 1 / more ulp, and the RMS error in ulp against the exact root (f64).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void k(const float* x, float* y, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = __builtin_amdgcn_sqrtf(x[i]);
}

// every non-negative finite f32 (bit patterns 0 .. 0x7F800000, denormal inputs included) against its successor: the number of
// places where v_sqrt_f32 DEcreases.  0 = monotone, so the maximum of a set of magnitudes is the root of the largest power --
// what lets stft8192_kernel take one root for the frame maximum of the bins it stores as powers (internal.hpp, SPEC_POWER_FROM).
__global__ void k_monotone(unsigned long long* decreases, unsigned int* first_bad) {
    const uint32_t stride = gridDim.x * 256u;
    unsigned long long bad = 0;
    for (uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x; b < 0x7F800000ull; b += stride) {
        const float lo = __builtin_amdgcn_sqrtf(__uint_as_float((uint32_t)b)), hi = __builtin_amdgcn_sqrtf(__uint_as_float((uint32_t)b + 1u));
        if (!(hi >= lo)) { bad++; atomicMin(first_bad, (uint32_t)b); }
    }
    if (bad) atomicAdd(decreases, bad);
}

static void report(const char* what, const std::vector<float>& x, const std::vector<float>& y) {
    size_t same = 0, one = 0, more = 0;
    double se_hw = 0, se_rn = 0;
    for (size_t i = 0; i < x.size(); i++) {
        const float rn = sqrtf(x[i]);
        uint32_t a, b;
        memcpy(&a, &rn, 4); memcpy(&b, &y[i], 4);
        const int d = (int)a - (int)b;
        if (d == 0) same++; else if (d == 1 || d == -1) one++; else more++;
        const double exact = sqrt((double)x[i]);
        const double ulp = ldexp(1.0, ilogb(exact) - 23);
        se_hw += ((double)y[i] - exact) * ((double)y[i] - exact) / (ulp * ulp);
        se_rn += ((double)rn - exact) * ((double)rn - exact) / (ulp * ulp);
    }
    const double n = (double)x.size();
    printf("%s: n=%zu  same as sqrtf %.4f  off by one ulp %.4f  more %.6f  rms error: v_sqrt_f32 %.4f ulp, correctly rounded %.4f ulp\n", what,
           x.size(), same / n, one / n, more / n, sqrt(se_hw / n), sqrt(se_rn / n));
}

int main() {
    const size_t N = 1u << 24;
    std::vector<float> x(N), y(N);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < N; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const uint32_t mant = (uint32_t)(s >> 41), e = 107 + (uint32_t)((s >> 8) % 40);
        const uint32_t bits = (e << 23) | mant;
        memcpy(&x[i], &bits, 4);
    }
    float *dx, *dy;
    hipMalloc(&dx, N * 4); hipMalloc(&dy, N * 4);
    hipMemcpy(dx, x.data(), N * 4, hipMemcpyHostToDevice);
    k<<<(unsigned)((N + 255) / 256), 256>>>(dx, dy, N);
    hipMemcpy(y.data(), dy, N * 4, hipMemcpyDeviceToHost);
    report("random [2^-20, 2^20)", x, y);
    for (size_t i = 0; i < N; i++) { const uint32_t bits = (127u << 23) | (uint32_t)(i & 0x7FFFFF) | ((uint32_t)(i >> 23) << 23); memcpy(&x[i], &bits, 4); }
    hipMemcpy(dx, x.data(), N * 4, hipMemcpyHostToDevice);
    k<<<(unsigned)((N + 255) / 256), 256>>>(dx, dy, N);
    hipMemcpy(y.data(), dy, N * 4, hipMemcpyDeviceToHost);
    report("every float of [1, 4)", x, y);
    unsigned long long* d_bad; unsigned int* d_first;
    unsigned long long bad = 0; unsigned int first = 0xFFFFFFFFu;
    hipMalloc(&d_bad, 8); hipMalloc(&d_first, 4);
    hipMemcpy(d_bad, &bad, 8, hipMemcpyHostToDevice); hipMemcpy(d_first, &first, 4, hipMemcpyHostToDevice);
    k_monotone<<<4096, 256>>>(d_bad, d_first);
    hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&first, d_first, 4, hipMemcpyDeviceToHost);
    printf("monotone sweep over every non-negative finite f32 (2139095040 neighbours): v_sqrt_f32 decreases at %llu places%s\n", bad,
           bad ? "" : " -- monotone");
    if (bad) printf("  first at bit pattern 0x%08x\n", first);
    return bad ? 1 : 0;
}
