// fft_r16.hpp -- register-resident radix-16 building block for the gfx950 FFT kernels.
//
// A lane keeps 16 complex values (32 VGPRs) and performs a 16-point DFT entirely in registers
// (two layers of radix-4 with the W16 twiddles as immediates).  Larger transforms are built as
// products of radix-16 passes with LDS transposes in between:
//   256-point  (FFT-512 real frames)  = 16 x 16, one transpose inside a 16-lane group
//   4096-point (FFT-8192 real frames) = 16 x 16 x 16, two transposes inside a 256-thread workgroup
#pragma once
#include "device_utils.hpp"

namespace bg {

// forward 16-point DFT, in place: v[k] <- sum_n v[n] * exp(-2*pi*i*n*k/16)
__device__ __forceinline__ void radix16(float2 (&v)[16]) {
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
    // layer 1: for each b, radix-4 over a on v[4a + b]  -> u[b][c] stored back at v[4c + b]
#pragma unroll
    for (int b = 0; b < 4; b++) radix4(v[b], v[4 + b], v[8 + b], v[12 + b]);
    // twiddle u[b][c] *= W16^(b*c)   (element index 4c + b)
    // c = 1: b = 1,2,3 -> W^1, W^2, W^3
    v[5] = cmul(v[5], make_float2(C1, -S1));
    v[6] = cmul(v[6], make_float2(R2, -R2));
    v[7] = cmul(v[7], make_float2(S1, -C1));
    // c = 2: b = 1,2,3 -> W^2, W^4, W^6
    v[9] = cmul(v[9], make_float2(R2, -R2));
    v[10] = cmul_mi(v[10]);
    v[11] = cmul(v[11], make_float2(-R2, -R2));
    // c = 3: b = 1,2,3 -> W^3, W^6, W^9
    v[13] = cmul(v[13], make_float2(S1, -C1));
    v[14] = cmul(v[14], make_float2(-R2, -R2));
    v[15] = cmul(v[15], make_float2(-C1, S1));
    // layer 2: for each c, radix-4 over b on v[4c + b] -> X[c + 4d] at v[4c + d]
#pragma unroll
    for (int c = 0; c < 4; c++) radix4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
    // reorder: X[k], k = c + 4d, currently at v[4c + d]  => transpose the 4x4 index grid
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int d = c + 1; d < 4; d++) {
            const float2 t = v[4 * c + d];
            v[4 * c + d] = v[4 * d + c];
            v[4 * d + c] = t;
        }
}

}  // namespace bg
