// f64 matrix-pipe probe (developer tool): the MFMA count of chroma_kernel for 1024 three-minute songs (29 696 wavefronts x
// 4128 v_mfma_f64_16x16x4_f64) with no memory traffic, and the same contraction as 3 x v_mfma_f64_4x4x4_4b_f64 per step.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void mfma16(double* out, int steps, double seed) {
    double4_t acc[8];
    for (int i = 0; i < 8; i++) acc[i] = double4_t{0, 0, 0, 0};
    double a = seed + threadIdx.x, b = seed * 0.5 + threadIdx.x;
#pragma unroll 1
    for (int s = 0; s < steps; s++) {
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i & 7] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i & 7], 0, 0, 0);
    }
    double4_t t = acc[0];
    for (int i = 1; i < 8; i++) t += acc[i];
    if (t.x == 1.2345) out[0] = t.x + t.y + t.z + t.w;
}
__global__ __launch_bounds__(256) void mfma4(double* out, int steps, double seed) {
    double acc[24];
    for (int i = 0; i < 24; i++) acc[i] = 0.0;
    double a = seed + threadIdx.x, b = seed * 0.5 + threadIdx.x;
#pragma unroll 1
    for (int s = 0; s < steps; s++) {
#pragma unroll
        for (int i = 0; i < 48; i++) acc[i % 24] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i % 24], 0, 0, 0);
    }
    double t = 0;
    for (int i = 0; i < 24; i++) t += acc[i];
    if (t == 1.2345) out[0] = t;
}
int main() {
    double* out; hipMalloc(&out, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int wgs = 29696 / 4, steps = 258;
    for (int rep = 0; rep < 3; rep++) {
        float ms;
        hipEventRecord(e0); hipLaunchKernelGGL(mfma16, dim3(wgs), dim3(256), 0, 0, out, steps, 1.0); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("16x16x4: %d x 16 MFMA per wave: %.3f ms  (%.1f TFLOP/s)\n", steps, ms, 29696.0 * steps * 16 * 2048 / ms * 1e-9);
        hipEventRecord(e0); hipLaunchKernelGGL(mfma4, dim3(wgs), dim3(256), 0, 0, out, steps, 1.0); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("4x4x4_4b: %d x 48 MFMA per wave: %.3f ms  (%.1f TFLOP/s)\n", steps, ms, 29696.0 * steps * 48 * 512 / ms * 1e-9);
    }
    return 0;
}
