// tw_probe.hip -- HBM write rate of the self-distance kernel's two store patterns, without any arithmetic: a 100 000 x 100 000
// f32 matrix written once in RUNS of 128 B .. 4 KiB at a row pitch of 400 000 B (non-temporal 16-byte stores, 64 lanes x 16 B =
// 1 KiB per store instruction, i.e. 1024 / run different matrix rows per instruction).  The mirrored half of the self-distance
// matrix is written in 128-byte runs (32 staged rows); the direct half in 1-KiB runs.
// Build: hipcc --offload-arch=gfx950 -O3 -o tw_probe tw_probe.hip ;  run: ./tw_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(e)                                                                      \
    do {                                                                              \
        hipError_t r_ = (e);                                                          \
        if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); exit(1); } \
    } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

// A workgroup (256 threads) writes a block of ROWS_PER_WG matrix rows x RUN_F floats: (RUN_F / 4) lanes cover one run, the rest
// of the wave covers other rows.  Blocks tile the matrix: blockIdx.x along the columns, blockIdx.y along the rows.
template <int RUN_F>
__global__ __launch_bounds__(256) void write_runs(float* out, uint64_t n, uint64_t ld) {
    constexpr int LPR = RUN_F / 4;                 // lanes per run
    constexpr int ROWS_PER_PASS = 256 / LPR;       // matrix rows one pass of the workgroup touches
    constexpr int BLOCK_ROWS = 256;                // rows per workgroup
    const uint64_t c0 = (uint64_t)blockIdx.x * RUN_F + 4 * (threadIdx.x % LPR);
    const uint64_t r0 = (uint64_t)blockIdx.y * BLOCK_ROWS + threadIdx.x / LPR;
    if (c0 + 3 >= n) return;
    f4 v;
    v.x = (float)threadIdx.x; v.y = 1.f; v.z = 2.f; v.w = 3.f;
#pragma unroll 4
    for (int p = 0; p < BLOCK_ROWS / ROWS_PER_PASS; p++) {
        const uint64_t r = r0 + (uint64_t)p * ROWS_PER_PASS;
        if (r < n) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(out + r * ld + c0));
    }
}

template <int RUN_F>
static void run(float* d, uint64_t n) {
    const dim3 grid((uint32_t)((n + RUN_F - 1) / RUN_F), (uint32_t)((n + 255) / 256));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(write_runs<RUN_F>, grid, dim3(256), 0, 0, d, n, n);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (rep && ms < best) best = ms;
    }
    printf("runs of %5d B: %.3f ms = %.2f TB/s (%llu x %llu f32, pitch %llu B)\n", RUN_F * 4, best, 4.0 * n * n / best / 1e9,
           (unsigned long long)n, (unsigned long long)n, (unsigned long long)(n * 4));
}

int main(int argc, char** argv) {
    const uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 100000;
    float* d;
    CHECK(hipMalloc((void**)&d, n * n * 4 + 4096));
    run<32>(d, n);
    run<64>(d, n);
    run<128>(d, n);
    run<256>(d, n);
    run<512>(d, n);
    run<1024>(d, n);
    return 0;
}
