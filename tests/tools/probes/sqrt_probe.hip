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
    return 0;
}
