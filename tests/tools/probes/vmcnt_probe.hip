// vmcnt probe (developer tool): how many units of the vector-memory counter does one global->LDS transfer
// (global_load_lds_dwordx4) take, compared with one ordinary global_load_dwordx4?  The kernel issues operations on cold
// addresses and reads IB_STS.VM_CNT right behind them (before they can have completed).
#include <hip/hip_runtime.h>
#include <stdint.h>
__device__ __forceinline__ uint32_t vm_cnt_now() {
    const uint32_t v = __builtin_amdgcn_s_getreg(7 | (31 << 11));
    return (v & 0xF) | (((v >> 22) & 0x3) << 4);
}
typedef float f4_t __attribute__((ext_vector_type(4)));
__global__ void probe(const float* __restrict__ g, uint32_t* out, int stride) {
    __shared__ float lds[64 * 4 * 8];
    const float* p = g + (size_t)threadIdx.x * stride;
    uint32_t c0 = vm_cnt_now();
    __builtin_amdgcn_global_load_lds(p, &lds[0], 16, 0, 0);
    uint32_t c1 = vm_cnt_now();
    __builtin_amdgcn_global_load_lds(p + 1024 * 1024, &lds[256], 16, 0, 0);
    __builtin_amdgcn_global_load_lds(p + 2 * 1024 * 1024, &lds[512], 16, 0, 0);
    uint32_t c3 = vm_cnt_now();
    f4_t x = {0, 0, 0, 0}, y = {0, 0, 0, 0};
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16" : "=&v"(x), "=&v"(y) : "v"(p + 3 * 1024 * 1024) : "memory");
    uint32_t c5 = vm_cnt_now();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t c6 = vm_cnt_now();
    if (threadIdx.x == 0) { out[0] = c0; out[1] = c1; out[2] = c3; out[3] = c5; out[4] = c6; }
    if (x.x + y.y + lds[threadIdx.x] == 123.f) out[5] = 1;
}
#include <stdio.h>
int main() {
    float* g; uint32_t* out; uint32_t h[8] = {};
    (void)hipMalloc(&g, (size_t)64 << 20); (void)hipMalloc(&out, 64);
    (void)hipMemset(g, 0, (size_t)64 << 20); (void)hipMemset(out, 0, 64);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, g, out, 4096);
    (void)hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
    printf("vm_cnt: before %u, after 1 LDS-DMA %u, after 3 LDS-DMA %u, after +2 plain loads %u, after vmcnt(0) %u\n", h[0], h[1], h[2], h[3], h[4]);
    return 0;
}
