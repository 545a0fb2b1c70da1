// issue_probe.hip -- can a gfx950 CU overlap VALU work with LDS traffic, and across which wavefronts?
//
// The FFT-8192 kernel's counters say its time is (VALU busy) + (LDS busy), not max of the two.  This probe measures the
// two resources alone, together in one wavefront, split across workgroups (every SIMD hosts VALU-only and LDS-only
// waves) and in the FFT kernel's phase structure (arithmetic -> 16 writes -> barrier -> 16 reads), at 1/2/4 workgroups
// per CU.  Build: hipcc --offload-arch=gfx950 -O3 -o issue_probe issue_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CHECK(e)                                                                      \
    do {                                                                              \
        hipError_t r_ = (e);                                                          \
        if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); exit(1); } \
    } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

// 16 independent packed FMAs (dependency distance 16)
#define VALU16(a, b, c)                                                                                      \
    asm volatile(                                                                                            \
        "v_pk_fma_f32 %0, %0, %16, %17\n\tv_pk_fma_f32 %1, %1, %16, %17\n\tv_pk_fma_f32 %2, %2, %16, %17\n\t"    \
        "v_pk_fma_f32 %3, %3, %16, %17\n\tv_pk_fma_f32 %4, %4, %16, %17\n\tv_pk_fma_f32 %5, %5, %16, %17\n\t"    \
        "v_pk_fma_f32 %6, %6, %16, %17\n\tv_pk_fma_f32 %7, %7, %16, %17\n\tv_pk_fma_f32 %8, %8, %16, %17\n\t"    \
        "v_pk_fma_f32 %9, %9, %16, %17\n\tv_pk_fma_f32 %10, %10, %16, %17\n\tv_pk_fma_f32 %11, %11, %16, %17\n\t" \
        "v_pk_fma_f32 %12, %12, %16, %17\n\tv_pk_fma_f32 %13, %13, %16, %17\n\tv_pk_fma_f32 %14, %14, %16, %17\n\t" \
        "v_pk_fma_f32 %15, %15, %16, %17"                                                                    \
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),   \
          "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) \
        : "v"(b), "v"(c))

// 16 conflict-free 8-byte LDS writes of a wave-private 8 KB region (addr = byte address of lane's slot 0)
#define LDSW16(addr, d)                                                                                          \
    asm volatile(                                                                                                \
        "ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:512\n\tds_write_b64 %0, %3 offset:1024\n\t"             \
        "ds_write_b64 %0, %4 offset:1536\n\tds_write_b64 %0, %5 offset:2048\n\tds_write_b64 %0, %6 offset:2560\n\t" \
        "ds_write_b64 %0, %7 offset:3072\n\tds_write_b64 %0, %8 offset:3584\n\tds_write_b64 %0, %1 offset:4096\n\t" \
        "ds_write_b64 %0, %2 offset:4608\n\tds_write_b64 %0, %3 offset:5120\n\tds_write_b64 %0, %4 offset:5632\n\t" \
        "ds_write_b64 %0, %5 offset:6144\n\tds_write_b64 %0, %6 offset:6656\n\tds_write_b64 %0, %7 offset:7168\n\t" \
        "ds_write_b64 %0, %8 offset:7680"                                                                         \
        :: "v"(addr), "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), "v"(d[6]), "v"(d[7]) : "memory")

#define LDSR16(addr, d)                                                                                          \
    asm volatile(                                                                                                \
        "ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:512\n\tds_read_b64 %2, %8 offset:1024\n\t"                \
        "ds_read_b64 %3, %8 offset:1536\n\tds_read_b64 %4, %8 offset:2048\n\tds_read_b64 %5, %8 offset:2560\n\t"   \
        "ds_read_b64 %6, %8 offset:3072\n\tds_read_b64 %7, %8 offset:3584\n\tds_read_b64 %0, %8 offset:4096\n\t"   \
        "ds_read_b64 %1, %8 offset:4608\n\tds_read_b64 %2, %8 offset:5120\n\tds_read_b64 %3, %8 offset:5632\n\t"   \
        "ds_read_b64 %4, %8 offset:6144\n\tds_read_b64 %5, %8 offset:6656\n\tds_read_b64 %6, %8 offset:7168\n\t"   \
        "ds_read_b64 %7, %8 offset:7680"                                                                          \
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7])  \
        : "v"(addr) : "memory")

// the same 8 KB per wave as 8 + 8 sixteen-byte accesses
#define LDSW8_128(addr, q)                                                                                       \
    asm volatile(                                                                                                \
        "ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:1024\n\tds_write_b128 %0, %3 offset:2048\n\t"        \
        "ds_write_b128 %0, %4 offset:3072\n\tds_write_b128 %0, %1 offset:4096\n\tds_write_b128 %0, %2 offset:5120\n\t" \
        "ds_write_b128 %0, %3 offset:6144\n\tds_write_b128 %0, %4 offset:7168"                                    \
        :: "v"(addr), "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]) : "memory")
#define LDSR8_128(addr, q)                                                                                       \
    asm volatile(                                                                                                \
        "ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"          \
        "ds_read_b128 %3, %4 offset:3072\n\tds_read_b128 %0, %4 offset:4096\n\tds_read_b128 %1, %4 offset:5120\n\t" \
        "ds_read_b128 %2, %4 offset:6144\n\tds_read_b128 %3, %4 offset:7168"                                      \
        : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]) : "v"(addr) : "memory")

#define WAIT_LGKM() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// mode 0: every wave runs   [nv x VALU16] [nw x LDSW16] (barrier) [nr x LDSR16] wait   per iteration
// mode 1: even workgroups run only the VALU part, odd workgroups only the LDS part (no barrier)
// mode 2: like 0, but the LDS instructions are issued BEFORE the arithmetic and waited for after it
// mode 3: LDS only, 16-byte accesses (8 writes + 8 reads of the same 8 KB); mode 4: the same behind nv x VALU16
// LDS_BYTES sets the residency: 40 KB -> 4 workgroups per CU, 80 KB -> 2, 160 KB -> 1
template <int LDS_BYTES, int nv, int nw, int nr, int barrier, int mode>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    __shared__ char lds[LDS_BYTES];
    const int t = threadIdx.x;
    const uint32_t addr = (uint32_t)(size_t)(lds) + (uint32_t)((t >> 6) * 8192 + (t & 63) * 8);
    f2 a[16], d[8];
    for (int i = 0; i < 16; i++) a[i] = (f2){(float)t, (float)i};
    for (int i = 0; i < 8; i++) d[i] = (f2){(float)i, (float)t};
    const f2 b = {0.999f, 1.001f}, c = {0.001f, -0.001f};
    const bool do_v = mode != 1 || (blockIdx.x & 1) == 0;
    if (mode == 3 || mode == 4) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 q[4];
        for (int i = 0; i < 4; i++) q[i] = (f4){(float)i, (float)t, 1.0f, 2.0f};
        const uint32_t addr16 = (uint32_t)(size_t)(lds) + (uint32_t)((t >> 6) * 8192 + (t & 63) * 16);
#pragma unroll 1
        for (int it = 0; it < iters; it++) {
            if (mode == 4) {
#pragma unroll
                for (int k = 0; k < nv; k++) VALU16(a, b, c);
            }
#pragma unroll
            for (int k = 0; k < nw; k++) LDSW8_128(addr16, q);
            if (barrier) __syncthreads();
#pragma unroll
            for (int k = 0; k < nr; k++) LDSR8_128(addr16, q);
            WAIT_LGKM();
        }
        for (int i = 0; i < 4; i++) d[i] = (f2){q[i].x + q[i].z, q[i].y + q[i].w};
    } else if (mode == 2) {
#pragma unroll 1
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int k = 0; k < nw; k++) LDSW16(addr, d);
#pragma unroll
            for (int k = 0; k < nr; k++) LDSR16(addr, d);
#pragma unroll
            for (int k = 0; k < nv; k++) VALU16(a, b, c);
            WAIT_LGKM();
        }
    } else if (mode == 1) {
        if (do_v) {
#pragma unroll 1
            for (int it = 0; it < iters; it++) {
#pragma unroll
                for (int k = 0; k < nv; k++) VALU16(a, b, c);
            }
        } else {
#pragma unroll 1
            for (int it = 0; it < iters; it++) {
#pragma unroll
                for (int k = 0; k < nw; k++) LDSW16(addr, d);
#pragma unroll
                for (int k = 0; k < nr; k++) LDSR16(addr, d);
                WAIT_LGKM();
            }
        }
    } else {
#pragma unroll 1
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int k = 0; k < nv; k++) VALU16(a, b, c);
#pragma unroll
            for (int k = 0; k < nw; k++) LDSW16(addr, d);
            if (barrier) __syncthreads();
#pragma unroll
            for (int k = 0; k < nr; k++) LDSR16(addr, d);
            WAIT_LGKM();
            if (barrier > 1) __syncthreads();
        }
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += a[i].x + a[i].y;
    for (int i = 0; i < 8; i++) s += d[i].x + d[i].y;
    if (s == 12345.678f) out[blockIdx.x * 256 + t] = s;
}

template <int LDS_BYTES, int nv, int nw, int nr, int barrier, int mode>
float run(float* d_out, int grid, int iters) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((probe<LDS_BYTES, nv, nw, nr, barrier, mode>), dim3(grid), dim3(256), 0, 0, d_out, iters);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const double ghz = p.clockRate * 1e-6;
    printf("device %s, %d CUs, %.2f GHz\n", p.name, cus, ghz);
    float* d_out;
    CHECK(hipMalloc(&d_out, 64 << 20));
    const int iters = 2000;
    // cycles per iteration = ms * 1e-3 * clock / iters, with `occ` workgroups resident per CU
#define ROW(name, nv, nw, nr, barrier, mode)                                                                      \
    {                                                                                                             \
        const float ms = occ == 4   ? run<40 * 1024, nv, nw, nr, barrier, mode>(d_out, grid, iters)               \
                         : occ == 2 ? run<80 * 1024, nv, nw, nr, barrier, mode>(d_out, grid, iters)               \
                                    : run<150 * 1024, nv, nw, nr, barrier, mode>(d_out, grid, iters);             \
        const double cyc = ms * 1e-3 * ghz * 1e9 / iters;                                                         \
        printf("%s  %8.3f ms  %9.1f cycles/iter  (%.1f per workgroup-iteration)\n", name, ms, cyc, cyc / occ);    \
    }
    for (int occ = 4; occ >= 1; occ /= 2) {
        const int grid = cus * occ;  // one round of resident workgroups
        printf("---- %d workgroup(s) of 256 threads per CU ----\n", occ);
        ROW("valu only            (6x16 / iter)", 6, 0, 0, 0, 0);
        ROW("valu only           (12x16 / iter)", 12, 0, 0, 0, 0);
        ROW("lds only  (16 w + 16 r / iter)    ", 0, 1, 1, 0, 0);
        ROW("lds only, barrier between w and r ", 0, 1, 1, 1, 0);
        ROW("valu + lds, same wave, in order   ", 6, 1, 1, 0, 0);
        ROW("valu + lds, same wave, lds first  ", 6, 1, 1, 0, 2);
        ROW("valu + lds, barrier (FFT phases)  ", 6, 1, 1, 1, 0);
        ROW("valu + lds, two barriers          ", 6, 1, 1, 2, 0);
        ROW("split: even WGs valu, odd WGs lds ", 6, 1, 1, 0, 1);
        ROW("split, twice the valu work        ", 12, 1, 1, 0, 1);
        ROW("lds only, b128 (8 w + 8 r / iter) ", 0, 1, 1, 0, 3);
        ROW("valu + lds b128, barrier          ", 6, 1, 1, 1, 4);
    }
    return 0;
}
