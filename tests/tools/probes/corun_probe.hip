// corun_probe.hip -- do a VALU-bound kernel and an HBM-bound kernel from two streams overlap on gfx950, i.e. is
// time(V || M) closer to max(V, M) or to V + M?  V: 256-thread workgroups of dependent-free packed FMAs, sized by its
// register demand (-DV_REGS via launch bounds) so that 2 workgroups fill a CU's register file like the FFT-512 kernel;
// M: 128-thread workgroups streaming a 16 GiB buffer with 16-byte loads, LDS-padded so that at most 4 fit a CU.
// Build: hipcc --offload-arch=gfx950 -O3 -o corun_probe corun_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(e)                                                                      \
    do {                                                                              \
        hipError_t r_ = (e);                                                          \
        if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); exit(1); } \
    } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// VALU kernel: NACC packed accumulators per lane keep `regs` high; iters x NACC packed FMAs per wavefront
template <int NACC>
__global__ __launch_bounds__(256) void valu_kernel(float* out, int iters) {
    f2 a[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) a[i] = (f2){(float)threadIdx.x, (float)i};
    const f2 b = {0.999f, 1.001f}, c = {0.001f, -0.001f};
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += a[i].x + a[i].y;
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

// streaming kernel: every workgroup reads `bytes_per_wg` contiguous bytes, 8 x 16-byte loads in flight per lane
__global__ __launch_bounds__(128) void mem_kernel(const f4* __restrict__ src, float* out, size_t vec_per_wg) {
    extern __shared__ char pad[];
    const f4* p = src + (size_t)blockIdx.x * vec_per_wg + threadIdx.x;
    f4 acc = {0, 0, 0, 0};
    for (size_t i = 0; i < vec_per_wg; i += 128 * 8) {
        f4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = __builtin_nontemporal_load(p + i + 128 * u);
#pragma unroll
        for (int u = 0; u < 8; u++) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;
    if (threadIdx.x == 9999) pad[0] = 1;
}

int main(int argc, char** argv) {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const size_t bytes = 16ull << 30;
    f4* src;
    float* out;
    CHECK(hipMalloc(&src, bytes));
    CHECK(hipMalloc(&out, 64 << 20));
    CHECK(hipMemset(src, 0, bytes));
    hipStream_t sv, sm;
    int lo, hi;
    CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipEvent_t v0, v1, m0, m1;
    CHECK(hipEventCreate(&v0)); CHECK(hipEventCreate(&v1)); CHECK(hipEventCreate(&m0)); CHECK(hipEventCreate(&m1));
    const int v_wgs = cus * 2 * 40;          // 40 rounds of a full machine (2 workgroups per CU)
    const int v_iters = 600;
    const int m_wgs = 8192;
    const size_t vec_per_wg = bytes / 16 / m_wgs;  // 2 MiB per workgroup
    for (int prio = 0; prio < 2; prio++) {
        CHECK(hipStreamCreateWithPriority(&sv, hipStreamNonBlocking, lo));
        CHECK(hipStreamCreateWithPriority(&sm, hipStreamNonBlocking, prio ? hi : lo));
        for (int lds_kb = 30; lds_kb <= 30; lds_kb += 30) {
            auto launch_v = [&]() { hipLaunchKernelGGL((valu_kernel<120>), dim3(v_wgs), dim3(256), 0, sv, out, v_iters); };
            auto launch_m = [&]() { hipLaunchKernelGGL(mem_kernel, dim3(m_wgs), dim3(128), lds_kb * 1024, sm, src, out, vec_per_wg); };
            float tv = 0, tm = 0, tvv = 0, tmm = 0;
            for (int rep = 0; rep < 2; rep++) {
                CHECK(hipEventRecord(v0, sv)); launch_v(); CHECK(hipEventRecord(v1, sv)); CHECK(hipDeviceSynchronize());
                CHECK(hipEventElapsedTime(&tv, v0, v1));
                CHECK(hipEventRecord(m0, sm)); launch_m(); CHECK(hipEventRecord(m1, sm)); CHECK(hipDeviceSynchronize());
                CHECK(hipEventElapsedTime(&tm, m0, m1));
            }
            // together: V first, M a moment later on the other stream
            for (int rep = 0; rep < 2; rep++) {
                CHECK(hipEventRecord(v0, sv)); launch_v(); CHECK(hipEventRecord(v1, sv));
                CHECK(hipEventRecord(m0, sm)); launch_m(); CHECK(hipEventRecord(m1, sm));
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventElapsedTime(&tvv, v0, v1));
                CHECK(hipEventElapsedTime(&tmm, m0, m1));
            }
            float span = 0;
            CHECK(hipEventElapsedTime(&span, v0, m1));
            printf("M stream priority %s, M LDS pad %d KB: V alone %.3f ms, M alone %.3f ms (%.2f TB/s); together: V %.3f ms, M %.3f ms, V start -> M end %.3f ms (sum %.3f, max %.3f)\n",
                   prio ? "high" : "normal", lds_kb, tv, tm, bytes / (tm * 1e-3) / 1e12, tvv, tmm, span, tv + tm, tv > tm ? tv : tm);
        }
        CHECK(hipStreamDestroy(sv));
        CHECK(hipStreamDestroy(sm));
    }
    return 0;
}
