// dma_order_probe.hip -- does s_waitcnt vmcnt(N) order a global->LDS transfer (global_load_lds_dwordx4) against YOUNGER
// ordinary loads?  The hand-pipelined contraction of round 2 relied on it ("everything but the last 8 loads has landed")
// and returned a wrong chroma row once in ~7 000 songs; with s_waitcnt vmcnt(0) it is clean.
//
// Every wavefront: clear its LDS slot; issue ONE LDS-DMA from a COLD address (a fresh 2 MiB-strided line of a 16 GiB
// buffer: an HBM miss), then EIGHT ordinary 16-byte loads from a HOT line (L2 hits), then s_waitcnt vmcnt(8) -- by the
// in-order rule the DMA has landed -- and read the slot back at once.  A slot still holding the clear value is a DMA that
// the counted wait let through.  The same with vmcnt(0) as the control.
// Build: hipcc --offload-arch=gfx950 -O3 -o dma_order_probe dma_order_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int FULL_WAIT>
__global__ __launch_bounds__(256) void probe(const f4* __restrict__ cold, const f4* __restrict__ hot, unsigned long long* stale,
                                              unsigned long long* total, float* sink, int iters, size_t cold_vecs) {
    __shared__ f4 slot[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    unsigned long long my_stale = 0;
    f4 acc = {0, 0, 0, 0};
    size_t idx = ((size_t)blockIdx.x * 4 + wave) * 977u;
    for (int it = 0; it < iters; it++) {
        slot[wave][lane] = (f4){-1.0f, -1.0f, -1.0f, -1.0f};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        idx = (idx * 2862933555777941757ull + 3037000493ull);
        const f4* src = cold + ((idx >> 16) % (cold_vecs / 64)) * 64 + lane;   // 1 KiB run somewhere in 16 GiB: a miss
        const f4* h = hot + lane;
        __builtin_amdgcn_global_load_lds(src, &slot[wave_u][0], 16, 0, 0);
        f4 v[8];
        asm volatile(
            "global_load_dwordx4 %0, %8, off\n\tglobal_load_dwordx4 %1, %8, off offset:1024\n\t"
            "global_load_dwordx4 %2, %8, off offset:2048\n\tglobal_load_dwordx4 %3, %8, off offset:3072\n\t"
            "global_load_dwordx4 %4, %8, off\n\tglobal_load_dwordx4 %5, %8, off offset:1024\n\t"
            "global_load_dwordx4 %6, %8, off offset:2048\n\tglobal_load_dwordx4 %7, %8, off offset:3072"
            : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
            : "v"(h) : "memory");
        f4 got;
        if (FULL_WAIT)
            asm volatile("s_waitcnt vmcnt(0)\n\tds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(got) : "v"((uint32_t)(size_t)&slot[wave][lane]) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(8)\n\tds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(got) : "v"((uint32_t)(size_t)&slot[wave][lane]) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) :: "memory");
        my_stale += got.x == -1.0f;      // the cold buffer holds 1.0f everywhere
        for (int u = 0; u < 8; u++) acc += v[u];
    }
    if (my_stale) atomicAdd(stale, my_stale);
    if (threadIdx.x == 0) atomicAdd(total, (unsigned long long)iters * 256);
    if (acc.x == 12345.0f) sink[threadIdx.x] = acc.x;
}

// The contraction's own pattern: [3 DMA hot][8 loads cold] [3 DMA hot][8 loads cold]  s_waitcnt vmcnt(8)  -> the FIRST group
// (3 DMA + 8 loads) must have landed: the loads' registers are pre-set to a sentinel, the DMA slots too.
__global__ __launch_bounds__(256) void pattern(const f4* __restrict__ cold, const f4* __restrict__ hot, unsigned long long* cnt,
                                               float* sink, int iters, size_t cold_vecs) {
    __shared__ f4 slot[4][2][3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    unsigned long long stale_reg = 0, stale_lds = 0;
    f4 acc = {0, 0, 0, 0};
    size_t idx = ((size_t)blockIdx.x * 4 + wave) * 977u + 1;
    for (int it = 0; it < iters; it++) {
        for (int g = 0; g < 2; g++) for (int r = 0; r < 3; r++) slot[wave][g][r][lane] = (f4){-1.0f, -1.0f, -1.0f, -1.0f};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const f4* src[2];
        for (int g = 0; g < 2; g++) {
            idx = (idx * 2862933555777941757ull + 3037000493ull);
            src[g] = cold + ((idx >> 16) % (cold_vecs / 1024)) * 1024 + lane;   // 16 KiB run somewhere in 16 GiB
        }
        f4 v[8], w[8];
        for (int u = 0; u < 8; u++) { v[u] = (f4){-1.0f, -1.0f, -1.0f, -1.0f}; w[u] = v[u]; }
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
        for (int r = 0; r < 3; r++) __builtin_amdgcn_global_load_lds(hot + 64 * r + lane, &slot[wave_u][0][r][0], 16, 0, 0);
        asm volatile(
            "global_load_dwordx4 %0, %8, off\n\tglobal_load_dwordx4 %1, %8, off offset:1024\n\t"
            "global_load_dwordx4 %2, %8, off offset:2048\n\tglobal_load_dwordx4 %3, %8, off offset:3072\n\t"
            "global_load_dwordx4 %4, %8, off\n\tglobal_load_dwordx4 %5, %8, off offset:1024\n\t"
            "global_load_dwordx4 %6, %8, off offset:2048\n\tglobal_load_dwordx4 %7, %8, off offset:3072"
            : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
            : "v"(src[0]) : "memory");
        for (int r = 0; r < 3; r++) __builtin_amdgcn_global_load_lds(hot + 64 * (3 + r) + lane, &slot[wave_u][1][r][0], 16, 0, 0);
        asm volatile(
            "global_load_dwordx4 %0, %8, off\n\tglobal_load_dwordx4 %1, %8, off offset:1024\n\t"
            "global_load_dwordx4 %2, %8, off offset:2048\n\tglobal_load_dwordx4 %3, %8, off offset:3072\n\t"
            "global_load_dwordx4 %4, %8, off\n\tglobal_load_dwordx4 %5, %8, off offset:1024\n\t"
            "global_load_dwordx4 %6, %8, off offset:2048\n\tglobal_load_dwordx4 %7, %8, off offset:3072"
            : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7])
            : "v"(src[1]) : "memory");
        f4 a0, a1, a2;
        asm volatile("s_waitcnt vmcnt(8)\n\tds_read_b128 %0, %7\n\tds_read_b128 %1, %7 offset:1024\n\tds_read_b128 %2, %7 offset:2048\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(a0), "=&v"(a1), "=&v"(a2), "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])
                     : "v"((uint32_t)(size_t)&slot[wave][0][0][lane]) : "memory");
        // copy the first group's registers NOW (a late load return would overwrite them after this point)
        f4 c[8];
        for (int u = 0; u < 8; u++) c[u] = v[u];
        asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                     "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]) :: "memory");
        for (int u = 0; u < 8; u++) stale_reg += c[u].x == -1.0f;
        stale_lds += (a0.x == -1.0f) + (a1.x == -1.0f) + (a2.x == -1.0f);
        for (int u = 0; u < 8; u++) acc += v[u] + w[u];
    }
    if (stale_reg) atomicAdd(cnt, stale_reg);
    if (stale_lds) atomicAdd(cnt + 1, stale_lds);
    if (threadIdx.x == 0) atomicAdd(cnt + 2, (unsigned long long)iters * 256);
    if (acc.x == 12345.0f) sink[threadIdx.x] = acc.x;
}

// The other direction: EIGHT ordinary loads from cold addresses, then THREE LDS-DMA transfers from a hot line, then
// s_waitcnt vmcnt(3): in-order retirement would mean the eight loads have landed.
__global__ __launch_bounds__(256) void loads_then_dma(const f4* __restrict__ cold, const f4* __restrict__ hot, unsigned long long* cnt,
                                                      float* sink, int iters, size_t cold_vecs) {
    __shared__ f4 slot[4][3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    unsigned long long stale_reg = 0;
    f4 acc = {0, 0, 0, 0};
    size_t idx = ((size_t)blockIdx.x * 4 + wave) * 977u + 7;
    for (int it = 0; it < iters; it++) {
        idx = (idx * 2862933555777941757ull + 3037000493ull);
        const f4* src = cold + ((idx >> 16) % (cold_vecs / 1024)) * 1024 + lane;
        f4 v[8];
        for (int u = 0; u < 8; u++) v[u] = (f4){-1.0f, -1.0f, -1.0f, -1.0f};
        asm volatile(
            "global_load_dwordx4 %0, %8, off\n\tglobal_load_dwordx4 %1, %8, off offset:1024\n\t"
            "global_load_dwordx4 %2, %8, off offset:2048\n\tglobal_load_dwordx4 %3, %8, off offset:3072\n\t"
            "global_load_dwordx4 %4, %8, off\n\tglobal_load_dwordx4 %5, %8, off offset:1024\n\t"
            "global_load_dwordx4 %6, %8, off offset:2048\n\tglobal_load_dwordx4 %7, %8, off offset:3072"
            : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
            : "v"(src) : "memory");
        for (int r = 0; r < 3; r++) __builtin_amdgcn_global_load_lds(hot + 64 * r + lane, &slot[wave_u][r][0], 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(3)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) :: "memory");
        f4 c[8];
        for (int u = 0; u < 8; u++) c[u] = v[u];
        asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) :: "memory");
        for (int u = 0; u < 8; u++) stale_reg += c[u].x == -1.0f;
        for (int u = 0; u < 8; u++) acc += v[u];
        acc += slot[wave][0][lane];
    }
    if (stale_reg) atomicAdd(cnt, stale_reg);
    if (threadIdx.x == 0) atomicAdd(cnt + 2, (unsigned long long)iters * 256 * 8);
    if (acc.x == 12345.0f) sink[threadIdx.x] = acc.x;
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const size_t bytes = 16ull << 30, vecs = bytes / 16;
    f4 *cold, *hot; unsigned long long *cnt; float* sink;
    CHECK(hipMalloc(&cold, bytes)); CHECK(hipMalloc(&hot, 1 << 20)); CHECK(hipMalloc(&cnt, 64)); CHECK(hipMalloc(&sink, 4096));
    // fill with 1.0f (0x3f800000 is not a byte pattern: use a kernel-free trick, hipMemsetD32)
    CHECK(hipMemsetD32((hipDeviceptr_t)cold, 0x3f800000, bytes / 4));
    CHECK(hipMemsetD32((hipDeviceptr_t)hot, 0x3f800000, (1 << 20) / 4));
    for (int full = 0; full < 2; full++) {
        CHECK(hipMemset(cnt, 0, 64));
        const int blocks = p.multiProcessorCount * 8, iters = 4000;
        if (full) hipLaunchKernelGGL((probe<1>), dim3(blocks), dim3(256), 0, 0, cold, hot, cnt, cnt + 1, sink, iters, vecs);
        else hipLaunchKernelGGL((probe<0>), dim3(blocks), dim3(256), 0, 0, cold, hot, cnt, cnt + 1, sink, iters, vecs);
        CHECK(hipDeviceSynchronize());
        unsigned long long h[2];
        CHECK(hipMemcpy(h, cnt, 16, hipMemcpyDeviceToHost));
        printf("1 cold LDS-DMA, then 8 hot loads, then s_waitcnt vmcnt(%d), then ds_read of the DMA target: %llu of %llu lanes read the slot BEFORE the DMA landed\n",
               full ? 0 : 8, h[0], h[1]);
    }
    {
        CHECK(hipMemset(cnt, 0, 64));
        const int blocks = p.multiProcessorCount * 8, iters = 4000;
        hipLaunchKernelGGL(pattern, dim3(blocks), dim3(256), 0, 0, cold, hot, cnt, sink, iters, vecs);
        CHECK(hipDeviceSynchronize());
        unsigned long long h[3];
        CHECK(hipMemcpy(h, cnt, 24, hipMemcpyDeviceToHost));
        printf("[3 DMA hot][8 loads cold][3 DMA hot][8 loads cold] s_waitcnt vmcnt(8): of %llu lanes, %llu first-group LOAD registers and %llu first-group DMA slots had not landed\n",
               h[2], h[0], h[1]);
    }
    {
        CHECK(hipMemset(cnt, 0, 64));
        const int blocks = p.multiProcessorCount * 8, iters = 2000;
        hipLaunchKernelGGL(loads_then_dma, dim3(blocks), dim3(256), 0, 0, cold, hot, cnt, sink, iters, vecs);
        CHECK(hipDeviceSynchronize());
        unsigned long long h[3];
        CHECK(hipMemcpy(h, cnt, 24, hipMemcpyDeviceToHost));
        printf("[8 loads cold][3 DMA hot] s_waitcnt vmcnt(3): %llu of %llu load registers had NOT landed when the wait let the wavefront through\n", h[0], h[2]);
    }
    return 0;
}
