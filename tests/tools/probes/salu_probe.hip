// salu_probe.hip -- do s_nop / s_waitcnt issue beside VALU instructions of OTHER wavefronts of the same SIMD on gfx950,
// or does every instruction of any kind take one issue slot?  96 packed FMAs per iteration, alone and with 96 s_nop /
// s_waitcnt interleaved, at 1 and 4 wavefronts per SIMD.  (Answer: alone a wavefront pays ~3.4 cycles per s_nop; with four
// wavefronts per SIMD they cost nothing -- 2.2 vs 2.4 ns per VALU instruction.)  Time in s_memtime ticks of workgroup 0 and
// in kernel milliseconds.   Build: hipcc --offload-arch=gfx950 -O3 -o salu_probe salu_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

#define V(i) "v_pk_fma_f32 %" #i ", %" #i ", %17, %18\n\t"
#define S "s_add_u32 %16, %16, 1\n\t"
#define N "s_nop 0\n\t"
#define W "s_waitcnt lgkmcnt(0)\n\t"
#define ROW(X) V(0) X V(1) X V(2) X V(3) X V(4) X V(5) X V(6) X V(7) X V(8) X V(9) X V(10) X V(11) X V(12) X V(13) X V(14) X V(15) X
#define OPS : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), \
              "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]), "+s"(sacc) : "v"(b), "v"(c)

template <int KIND>
__global__ __launch_bounds__(256) void probe(float* out, int iters, unsigned long long* ticks) {
    f2 a[16];
    for (int i = 0; i < 16; i++) a[i] = (f2){(float)threadIdx.x, (float)i};
    const f2 b = {0.999f, 1.001f}, c = {0.001f, -0.001f};
    unsigned sacc = blockIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        for (int r = 0; r < 6; r++) {
            if (KIND == 0) asm volatile(ROW("") OPS);
            if (KIND == 2) asm volatile(ROW(N) OPS);
            if (KIND == 3) asm volatile(ROW(W) OPS);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = (float)sacc;
    for (int i = 0; i < 16; i++) s += a[i].x + a[i].y;
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int KIND>
void run(const char* name, float* out, unsigned long long* ticks, int cus) {
    const int iters = 2000;
    for (int occ = 1; occ <= 4; occ *= 4) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((probe<KIND>), dim3(cus * occ), dim3(256), 0, 0, out, iters, ticks);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("%-44s %d wave(s)/SIMD: kernel %.3f ms = %.2f ns per VALU instruction per SIMD\n", name, occ, ms, ms * 1e6 / (iters * 96.0 * occ));
    }
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    float* out; unsigned long long* ticks;
    CHECK(hipMalloc(&out, 64 << 20)); CHECK(hipMalloc(&ticks, 64));
    run<0>("96 v_pk_fma per iteration", out, ticks, p.multiProcessorCount);
    run<2>("+ 96 s_nop 0 interleaved", out, ticks, p.multiProcessorCount);
    run<3>("+ 96 s_waitcnt lgkmcnt(0) interleaved", out, ticks, p.multiProcessorCount);
    return 0;
}
