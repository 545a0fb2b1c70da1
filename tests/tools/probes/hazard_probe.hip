// hazard_probe.hip -- does gfx950 interlock back-to-back DEPENDENT packed-f32 VALU instructions?  (hipcc pads such
// pairs with s_nop -- and assumes every inline-asm result has the hazard -- which costs the exact-order distance kernel
// ~50 issue slots per row.)  Dependent chains written inside ONE asm block (the compiler cannot pad them) are checked
// against the same arithmetic done with padding; any stale read shows up as a mismatch.
// Build: hipcc --offload-arch=gfx950 -O3 -o hazard_probe hazard_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); exit(1); } } while (0)

__global__ void chain(f2* out, const f2* in, int padded) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    f2 a = in[3 * t], b = in[3 * t + 1], c = in[3 * t + 2];
    f2 r;
    if (padded) {
        asm volatile(
            "v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 4\n\t"  // (a.y, a.y) - b
            "v_pk_mul_f32 %0, %0, %0\n\ts_nop 4\n\t"                                                        // squared
            "v_pk_add_f32 %0, %0, %3\n\ts_nop 4\n\t"                                                        // + c
            "v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]\n\ts_nop 4\n\t"                           // * (a.y, a.x)
            "v_pk_fma_f32 %0, %0, %2, %0\n\ts_nop 4\n\t"                                                    // r * b + r
            "v_pk_add_f32 %0, %0, %0 op_sel:[1,0] op_sel_hi:[0,1]\n\ts_nop 4"                               // (r.y + r.x, r.x + r.y)
            : "=&v"(r) : "v"(a), "v"(b), "v"(c));
    } else {
        asm volatile(
            "v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_mul_f32 %0, %0, %0\n\t"
            "v_pk_add_f32 %0, %0, %3\n\t"
            "v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
            "v_pk_fma_f32 %0, %0, %2, %0\n\t"
            "v_pk_add_f32 %0, %0, %0 op_sel:[1,0] op_sel_hi:[0,1]"
            : "=&v"(r) : "v"(a), "v"(b), "v"(c));
    }
    out[t] = r;
}

int main() {
    const int n = 1 << 22;
    f2 *in, *o0, *o1;
    CHECK(hipMalloc(&in, 3 * n * sizeof(f2))); CHECK(hipMalloc(&o0, n * sizeof(f2))); CHECK(hipMalloc(&o1, n * sizeof(f2)));
    f2* h = (f2*)malloc(3 * n * sizeof(f2));
    unsigned s = 12345;
    for (int i = 0; i < 3 * n; i++) { s = s * 1664525u + 1013904223u; h[i].x = (float)(s >> 8) / 16777216.0f - 0.5f; s = s * 1664525u + 1013904223u; h[i].y = (float)(s >> 8) / 16777216.0f - 0.5f; }
    CHECK(hipMemcpy(in, h, 3 * n * sizeof(f2), hipMemcpyHostToDevice));
    long bad = 0;
    f2* a = (f2*)malloc(n * sizeof(f2)); f2* b = (f2*)malloc(n * sizeof(f2));
    for (int rep = 0; rep < 8; rep++) {
        hipLaunchKernelGGL(chain, dim3(n / 256), dim3(256), 0, 0, o0, in, 1);
        hipLaunchKernelGGL(chain, dim3(n / 256), dim3(256), 0, 0, o1, in, 0);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(a, o0, n * sizeof(f2), hipMemcpyDeviceToHost)); CHECK(hipMemcpy(b, o1, n * sizeof(f2), hipMemcpyDeviceToHost));
        for (int i = 0; i < n; i++) bad += (a[i].x != b[i].x) || (a[i].y != b[i].y);
    }
    printf("dependent packed-f32 chains, back to back vs padded with s_nop 4: %ld mismatches in 8 x %d lanes (sample %.9g %.9g | %.9g %.9g)\n", bad, n, a[5].x, a[5].y, b[5].x, b[5].y);
    return bad != 0;
}
