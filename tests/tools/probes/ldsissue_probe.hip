// ldsissue_probe.hip -- does an LDS instruction of one wavefront take VALU issue time from ANOTHER wavefront on the same
// SIMD?  One 512-thread workgroup per CU: wavefronts 0-3 (one per SIMD) run only packed FMAs, wavefronts 4-7 (the second
// wavefront of each SIMD) run only LDS reads, few enough not to saturate the LDS pipe.  If the two kinds issue
// independently, the VALU wavefronts take the same time with and without the LDS wavefronts beside them.
// Build: hipcc --offload-arch=gfx950 -O3 -o ldsissue_probe ldsissue_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// mode: 1 = VALU waves only, 2 = LDS waves only, 3 = both; WIDE: ds_read_b128 instead of ds_read_b64 (same bytes: half the instructions)
template <int WIDE>
__global__ __launch_bounds__(512) void probe(unsigned long long* ticks, float* sink, int iters, int mode, int lds_per_iter) {
    __shared__ char lds[64 * 1024];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool valu_wave = wave < 4;
    f2 a[16];
    for (int i = 0; i < 16; i++) a[i] = (f2){(float)threadIdx.x, (float)i};
    const f2 b = {0.999f, 1.001f}, c = {0.001f, -0.001f};
    f4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (valu_wave) {
        if (mode & 1) {
#pragma unroll 1
            for (int it = 0; it < iters; it++)
                for (int r = 0; r < 6; r++)
                    asm volatile("v_pk_fma_f32 %0, %0, %16, %17\n\tv_pk_fma_f32 %1, %1, %16, %17\n\tv_pk_fma_f32 %2, %2, %16, %17\n\tv_pk_fma_f32 %3, %3, %16, %17\n\t"
                                 "v_pk_fma_f32 %4, %4, %16, %17\n\tv_pk_fma_f32 %5, %5, %16, %17\n\tv_pk_fma_f32 %6, %6, %16, %17\n\tv_pk_fma_f32 %7, %7, %16, %17\n\t"
                                 "v_pk_fma_f32 %8, %8, %16, %17\n\tv_pk_fma_f32 %9, %9, %16, %17\n\tv_pk_fma_f32 %10, %10, %16, %17\n\tv_pk_fma_f32 %11, %11, %16, %17\n\t"
                                 "v_pk_fma_f32 %12, %12, %16, %17\n\tv_pk_fma_f32 %13, %13, %16, %17\n\tv_pk_fma_f32 %14, %14, %16, %17\n\tv_pk_fma_f32 %15, %15, %16, %17"
                                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]),
                                   "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(b), "v"(c));
        }
    } else if (mode & 2) {
        const uint32_t base = (uint32_t)(size_t)lds + (uint32_t)((wave - 4) * 16384) + (uint32_t)lane * (WIDE ? 16u : 8u);
#pragma unroll 1
        for (int it = 0; it < iters; it++) {
            for (int k = 0; k < lds_per_iter; k += 4) {
                if (WIDE) {
                    f4 x0, x1;
                    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)" : "=&v"(x0), "=&v"(x1) : "v"(base) : "memory");
                    acc += x0 + x1;
                } else {
                    f2 x0, x1, x2, x3;
                    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:512\n\tds_read_b64 %2, %4 offset:1024\n\tds_read_b64 %3, %4 offset:1536\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(base) : "memory");
                    acc.x += x0.x + x1.x + x2.x + x3.x;
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = acc.x + acc.y;
    for (int i = 0; i < 16; i++) s += a[i].x + a[i].y;
    if (s == 12345.678f) sink[threadIdx.x] = s;
    if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4)) ticks[wave == 0 ? 0 : 1] = t1 - t0;
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    unsigned long long* ticks; float* sink;
    CHECK(hipMalloc(&ticks, 64)); CHECK(hipMalloc(&sink, 4096));
    const int iters = 2000;
    for (int wide = 0; wide < 2; wide++)
        for (int lds_per_iter : {16, 32, 64}) {
            unsigned long long h[3][2];
            for (int mode = 1; mode <= 3; mode++) {
                for (int rep = 0; rep < 2; rep++) {
                    CHECK(hipMemset(ticks, 0, 64));
                    if (wide) hipLaunchKernelGGL((probe<1>), dim3(p.multiProcessorCount), dim3(512), 0, 0, ticks, sink, iters, mode, lds_per_iter);
                    else hipLaunchKernelGGL((probe<0>), dim3(p.multiProcessorCount), dim3(512), 0, 0, ticks, sink, iters, mode, lds_per_iter);
                    CHECK(hipDeviceSynchronize());
                }
                CHECK(hipMemcpy(h[mode - 1], ticks, 16, hipMemcpyDeviceToHost));
            }
            printf("%s, %2d b64-equivalents per iteration beside 96 packed FMAs: VALU wave alone %.0f cycles/iter, LDS wave alone %.0f, together: VALU wave %.0f, LDS wave %.0f\n",
                   wide ? "ds_read_b128" : "ds_read_b64 ", lds_per_iter, (double)h[0][0] / iters, (double)h[1][1] / iters, (double)h[2][0] / iters, (double)h[2][1] / iters);
        }
    return 0;
}
