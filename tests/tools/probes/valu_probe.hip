// valu_probe.hip -- issue cost of the VALU instruction forms the FFT kernels are made of, on gfx950, in SHADER cycles
// (s_memtime), with independent operands and as a dependent chain, at 1 and 4 wavefronts per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_probe valu_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(e)                                                                      \
    do {                                                                              \
        hipError_t r_ = (e);                                                          \
        if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); exit(1); } \
    } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

// 16 instructions on 16 independent registers
#define REP16(INS)                                                                                                  \
    asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(8) INS(9) INS(10) INS(11) INS(12) INS(13) \
                     INS(14) INS(15)                                                                                \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),   \
                   "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) \
                 : "v"(b), "s"(sc))
// 16 instructions forming ONE dependent chain on register 0
#define CHAIN16(INS)                                                                                                \
    asm volatile(INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0)     \
                     INS(0) INS(0)                                                                                  \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),   \
                   "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) \
                 : "v"(b), "s"(sc))

#define REP16F(INS)                                                                                                 \
    asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(8) INS(9) INS(10) INS(11) INS(12) INS(13) \
                     INS(14) INS(15)                                                                                \
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),   \
                   "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]) \
                 : "v"(y), "s"(sc))
#define CHAIN16F(INS)                                                                                               \
    asm volatile(INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0) INS(0)     \
                     INS(0) INS(0)                                                                                  \
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),   \
                   "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]) \
                 : "v"(y), "s"(sc))
#define STR(x) #x
#define I_PK_ADD(i) "v_pk_add_f32 %" STR(i) ", %" STR(i) ", %16\n\t"
#define I_PK_ADD_MOD(i) "v_pk_add_f32 %" STR(i) ", %" STR(i) ", %16 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
#define I_PK_MUL(i) "v_pk_mul_f32 %" STR(i) ", %" STR(i) ", %16\n\t"
#define I_PK_MUL_S(i) "v_pk_mul_f32 %" STR(i) ", %" STR(i) ", %17 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]\n\t"
#define I_PK_FMA(i) "v_pk_fma_f32 %" STR(i) ", %" STR(i) ", %16, %16\n\t"
#define I_PK_FMA_S(i) "v_pk_fma_f32 %" STR(i) ", %" STR(i) ", %17, %16 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
#define I_ADD(i) "v_add_f32 %" STR(i) ", %" STR(i) ", %16\n\t"
#define I_FMA(i) "v_fma_f32 %" STR(i) ", %" STR(i) ", %16, %16\n\t"
#define I_SQRT(i) "v_sqrt_f32 %" STR(i) ", %" STR(i) "\n\t"
#define I_RCP(i) "v_rcp_f32 %" STR(i) ", %" STR(i) "\n\t"
#define I_MOV_DPP(i) "v_mov_b32_dpp %" STR(i) ", %" STR(i) " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_ADD_DPP(i) "v_add_f32_dpp %" STR(i) ", %" STR(i) ", %" STR(i) " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define I_PK_MOV(i) "v_pk_mov_b32 %" STR(i) ", %" STR(i) ", %16 op_sel:[1,0]\n\t"
#define I_FMA64(i) "v_fma_f64 %" STR(i) ", %" STR(i) ", %16, %16\n\t"

template <int KIND, int CHAIN>
__global__ __launch_bounds__(256) void probe(unsigned long long* out, float* sink, int iters) {
    f2 a[16];
    for (int i = 0; i < 16; i++) a[i] = (f2){1.0f + threadIdx.x * 1e-3f, 1.0f + i * 1e-3f};
    float x[16];
    for (int i = 0; i < 16; i++) x[i] = 1.0f + threadIdx.x * 1e-3f + i;
    float y = 1.0001f;
    f2 b = {1.0001f, 0.9999f};
    const f2 sc = {0.7071f, -0.7071f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
#define BODYF(INS) if (CHAIN) { CHAIN16F(INS); CHAIN16F(INS); CHAIN16F(INS); CHAIN16F(INS); } else { REP16F(INS); REP16F(INS); REP16F(INS); REP16F(INS); }
#define BODY(INS) if (CHAIN) { CHAIN16(INS); CHAIN16(INS); CHAIN16(INS); CHAIN16(INS); } else { REP16(INS); REP16(INS); REP16(INS); REP16(INS); }
        if (KIND == 0) BODY(I_PK_ADD)
        if (KIND == 1) BODY(I_PK_ADD_MOD)
        if (KIND == 2) BODY(I_PK_MUL)
        if (KIND == 3) BODY(I_PK_MUL_S)
        if (KIND == 4) BODY(I_PK_FMA)
        if (KIND == 5) BODY(I_PK_FMA_S)
        if (KIND == 6) BODYF(I_ADD)
        if (KIND == 7) BODYF(I_FMA)
        if (KIND == 8) BODYF(I_SQRT)
        if (KIND == 9) BODYF(I_RCP)
        if (KIND == 10) BODYF(I_MOV_DPP)
        if (KIND == 11) BODYF(I_ADD_DPP)
        if (KIND == 12) BODY(I_PK_MOV)
        if (KIND == 14) BODY(I_FMA64)
    }
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; i++) s += a[i].x + a[i].y + x[i];
    if (s == 12345.678f) sink[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int KIND, int CHAIN>
void run(const char* name, unsigned long long* d_out, float* sink, int cus) {
    const int iters = 2000;
    double res[2], mhz[2], kms[2];
    for (int o = 0; o < 2; o++) {
        const int occ = o == 0 ? 1 : 4;  // workgroups of 256 threads per CU = wavefronts per SIMD
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((probe<KIND, CHAIN>), dim3(cus * occ), dim3(256), 0, 0, d_out, sink, iters);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipDeviceSynchronize());
        }
        unsigned long long cyc;
        CHECK(hipMemcpy(&cyc, d_out, 8, hipMemcpyDeviceToHost));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        mhz[o] = (double)cyc / (ms * 1e3);
        kms[o] = ms;
        res[o] = (double)cyc / ((double)iters * 64.0) / occ;  // shader cycles per instruction per SIMD
    }
    printf("%-44s %s  %6.2f cycles/instr at 1 wave/SIMD   %6.2f at 4 waves/SIMD   (ticks / event us: %.0f, %.0f; kernel ms %.3f, %.3f)\n", name, CHAIN ? "chain" : "indep", res[0], res[1], mhz[0], mhz[1], kms[0], kms[1]);
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    unsigned long long* d_out;
    float* sink;
    CHECK(hipMalloc(&d_out, 64));
    CHECK(hipMalloc(&sink, 4096));
    const int cus = p.multiProcessorCount;
    printf("%d CUs; cycles = s_memtime ticks of workgroup 0 / instructions issued per SIMD\n", cus);
#define BOTH(K, name) run<K, 0>(name, d_out, sink, cus); run<K, 1>(name, d_out, sink, cus);
    BOTH(0, "v_pk_add_f32")
    BOTH(1, "v_pk_add_f32 op_sel/neg (the -i rotation)")
    BOTH(2, "v_pk_mul_f32")
    BOTH(3, "v_pk_mul_f32 SGPR operand + modifiers")
    BOTH(4, "v_pk_fma_f32")
    BOTH(5, "v_pk_fma_f32 SGPR operand + modifiers")
    BOTH(6, "v_add_f32")
    BOTH(7, "v_fma_f32")
    BOTH(8, "v_sqrt_f32")
    BOTH(9, "v_rcp_f32")
    BOTH(10, "v_mov_b32 dpp row_shr:1")
    BOTH(11, "v_add_f32 dpp quad_perm")
    BOTH(12, "v_pk_mov_b32")
    BOTH(14, "v_fma_f64")
    return 0;
}
