// h2d_probe.hip -- host -> device rate of pinned 15.9 MB pieces (one three-minute song each) issued on 1, 2 or 4 streams.
// Build: hipcc --offload-arch=gfx950 -O3 -o h2d_probe h2d_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>

#define CHECK(e)                                                                      \
    do {                                                                              \
        hipError_t r_ = (e);                                                          \
        if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); exit(1); } \
    } while (0)

int main() {
    const size_t piece = 3969000ull * 4, n = 256;
    char *h, *d;
    CHECK(hipHostMalloc((void**)&h, piece * n, hipHostMallocDefault));
    CHECK(hipMalloc((void**)&d, piece * n));
    for (size_t i = 0; i < piece * n; i += 4096) h[i] = (char)i;
    hipStream_t st[4];
    for (int i = 0; i < 4; i++) CHECK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    for (int ns = 1; ns <= 4; ns *= 2)
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            for (size_t k = 0; k < n; k++) CHECK(hipMemcpyAsync(d + k * piece, h + k * piece, piece, hipMemcpyHostToDevice, st[k % ns]));
            CHECK(hipDeviceSynchronize());
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (rep == 2) printf("%d stream(s): %.2f GB/s (%zu pieces of %.1f MB)\n", ns, piece * n / s / 1e9, n, piece / 1e6);
        }
    // halves of every piece on two streams
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        for (size_t k = 0; k < n; k++)
            for (int hlf = 0; hlf < 2; hlf++)
                CHECK(hipMemcpyAsync(d + k * piece + hlf * (piece / 2), h + k * piece + hlf * (piece / 2), piece / 2, hipMemcpyHostToDevice, st[hlf]));
        CHECK(hipDeviceSynchronize());
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (rep == 2) printf("2 streams, half a piece each: %.2f GB/s\n", piece * n / s / 1e9);
    }
    // one big copy
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        CHECK(hipMemcpyAsync(d, h, piece * n, hipMemcpyHostToDevice, st[0]));
        CHECK(hipDeviceSynchronize());
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (rep == 2) printf("one %.1f GB copy: %.2f GB/s\n", piece * n / 1e9, piece * n / s / 1e9);
    }
    return 0;
}
