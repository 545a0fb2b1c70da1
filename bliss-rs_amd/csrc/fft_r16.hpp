// fft_r16.hpp -- register-resident radix-16 building block for the gfx950 FFT kernels.
//
// A lane keeps 16 complex values (32 VGPRs, each complex one aligned register pair) and performs a
// 16-point DFT entirely in registers.  Larger transforms are products of radix-16 passes with LDS
// transposes in between:
//   256-point  (FFT-512 real frames)  = 16 x 16, one transpose inside a 16-lane group
//   4096-point (FFT-8192 real frames) = 16 x 16 x 16, two transposes inside a 256-thread workgroup
//
// The kernels are VALU-issue bound (every wave64 VALU instruction occupies its SIMD for ~4 cycles, packed
// or not), so the complex arithmetic is written directly in packed VOP3P form: one v_pk_add_f32 per complex
// add, TWO instructions per complex multiply, and the multiplications by -i / conjugations folded into the
// op_sel / neg_lo / neg_hi modifiers of the consuming instruction.  hipcc reaches the first of these by
// itself but spends 4 instructions on a complex multiply and 3 v_mov per radix-4 butterfly.
//
// VOP3P modifier semantics used below (dst.lo / dst.hi computed independently):
//   op_sel[i]    : 1 => the LO result reads the HI half of source i
//   op_sel_hi[i] : 0 => the HI result reads the LO half of source i (default 1 = HI half)
//   neg_lo[i] / neg_hi[i] : negate source i for the LO / HI result
#pragma once
#include "device_utils.hpp"

namespace bg {

typedef float f2 __attribute__((ext_vector_type(2)));  // (re, im) in one 64-bit VGPR pair

__device__ __forceinline__ f2 mk(float x, float y) { f2 r; r.x = x; r.y = y; return r; }

// forward radix-4 butterfly (W4 = -i), in place, 8 instructions
__device__ __forceinline__ void radix4_pk(f2& v0, f2& v1, f2& v2, f2& v3) {
    f2 t0, t1, t2, d;
    asm("v_pk_add_f32 %4, %0, %2\n\t"                                            // t0 = v0 + v2
        "v_pk_add_f32 %5, %0, %2 neg_lo:[0,1] neg_hi:[0,1]\n\t"                   // t1 = v0 - v2
        "v_pk_add_f32 %6, %1, %3\n\t"                                            // t2 = v1 + v3
        "v_pk_add_f32 %7, %1, %3 neg_lo:[0,1] neg_hi:[0,1]\n\t"                   // d  = v1 - v3
        "v_pk_add_f32 %0, %4, %6\n\t"                                            // o0 = t0 + t2
        "v_pk_add_f32 %2, %4, %6 neg_lo:[0,1] neg_hi:[0,1]\n\t"                   // o2 = t0 - t2
        "v_pk_add_f32 %1, %5, %7 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"   // o1 = t1 + (-i) d
        "v_pk_add_f32 %3, %5, %7 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]"       // o3 = t1 - (-i) d
        : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(d));
}

// the same butterfly on (v0, v1, -i v2, v3): the multiplication by -i rides in the modifiers of the two instructions that read
// v2 (the W_16^4 twiddle of the radix-16 pass costs nothing), 8 instructions
__device__ __forceinline__ void radix4_pk_v2mi(f2& v0, f2& v1, f2& v2, f2& v3) {
    f2 t0, t1, t2, d;
    asm("v_pk_add_f32 %4, %0, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"   // t0 = v0 + (-i) v2
        "v_pk_add_f32 %5, %0, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"   // t1 = v0 - (-i) v2
        "v_pk_add_f32 %6, %1, %3\n\t"                                            // t2 = v1 + v3
        "v_pk_add_f32 %7, %1, %3 neg_lo:[0,1] neg_hi:[0,1]\n\t"                   // d  = v1 - v3
        "v_pk_add_f32 %0, %4, %6\n\t"                                            // o0 = t0 + t2
        "v_pk_add_f32 %2, %4, %6 neg_lo:[0,1] neg_hi:[0,1]\n\t"                   // o2 = t0 - t2
        "v_pk_add_f32 %1, %5, %7 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"   // o1 = t1 + (-i) d
        "v_pk_add_f32 %3, %5, %7 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]"       // o3 = t1 - (-i) d
        : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(d));
}

// a * w, 2 instructions
__device__ __forceinline__ f2 cmul_pk(f2 a, f2 w) {
    f2 t, r;
    asm("v_pk_mul_f32 %1, %2, %3 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]\n\t"   // t = (-a.y*w.y, a.y*w.x)
        "v_pk_fma_f32 %0, %2, %3, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1]"            // r = (a.x*w.x, a.x*w.y) + t
        : "=v"(r), "=&v"(t)
        : "v"(a), "v"(w));
    return r;
}

// a * w with a wave-uniform w held in an SGPR pair (compile-time twiddles: no v_mov per use)
__device__ __forceinline__ f2 cmul_pk_s(f2 a, f2 w) {
    f2 t, r;
    asm("v_pk_mul_f32 %1, %2, %3 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1]"
        : "=v"(r), "=&v"(t)
        : "v"(a), "s"(w));
    return r;
}

// (-i) * a = (a.y, -a.x), 1 instruction (ones = (1, 1))
__device__ __forceinline__ f2 mul_mi_pk(f2 a, f2 ones) {
    f2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0]" : "=v"(r) : "v"(a), "s"(ones));
    return r;
}

// Position of output X[k] after radix16(): the two radix-4 layers leave X[c + 4d] at v[4c + d]; callers
// index with R16(k) instead of paying register moves.
__host__ __device__ constexpr int R16(int k) { return 4 * (k & 3) + (k >> 2); }

// forward 16-point DFT, in place: v[R16(k)] <- sum_n v[n] * exp(-2*pi*i*n*k/16); 80 instructions
__device__ __forceinline__ void radix16(f2 (&v)[16]) {
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
    // layer 1: for each b, radix-4 over a on v[4a + b]  -> u[b][c] stored back at v[4c + b]
#pragma unroll
    for (int b = 0; b < 4; b++) radix4_pk(v[b], v[4 + b], v[8 + b], v[12 + b]);
    // twiddle u[b][c] *= W16^(b*c)   (element index 4c + b)
    v[5] = cmul_pk_s(v[5], mk(C1, -S1));     // W^1
    v[6] = cmul_pk_s(v[6], mk(R2, -R2));     // W^2
    v[7] = cmul_pk_s(v[7], mk(S1, -C1));     // W^3
    v[9] = cmul_pk_s(v[9], mk(R2, -R2));     // W^2
    // (v[10] *= W^4 = -i: folded into its butterfly below)
    v[11] = cmul_pk_s(v[11], mk(-R2, -R2));  // W^6
    v[13] = cmul_pk_s(v[13], mk(S1, -C1));   // W^3
    v[14] = cmul_pk_s(v[14], mk(-R2, -R2));  // W^6
    v[15] = cmul_pk_s(v[15], mk(-C1, S1));   // W^9
    // layer 2: for each c, radix-4 over b on v[4c + b] -> X[c + 4d] at v[4c + d]
    radix4_pk(v[0], v[1], v[2], v[3]);
    radix4_pk(v[4], v[5], v[6], v[7]);
    radix4_pk_v2mi(v[8], v[9], v[10], v[11]);
    radix4_pk(v[12], v[13], v[14], v[15]);
}

// Real-input split for a 2M-point real FFT packed as an M-point complex FFT (z[n] = x[2n] + i x[2n+1]):
// with zk = Z[k], zm = Z[(M-k) % M], w = exp(-2*pi*i*k/(2M)):
//     X[k] = (A + P) / 2,   X[M-k] = conj(A - P) / 2,   A = zk + conj(zm),  P = w * (-i) * (zk - conj(zm))
// Returns the squared magnitudes |A + P|^2 and |A - P|^2 (4 x the true ones); 9 packed instructions.  The two results are built
// side by side -- U = (Re(A + P), Re(A - P)), V = (Im(A + P), Im(A - P)), U U + V V -- so the sum of the two squares is one
// packed add for both bins (until round 6: X1 = A + P, X2 = A - P, their packed squares, and a scalar add each; the same
// operations on the same operands, one instruction more).
__device__ __forceinline__ void split_pair_sq(f2 zk, f2 zm, f2 w, float& sq_k, float& sq_mirror) {
    f2 a, b, t, p, u, v;
    asm("v_pk_add_f32 %0, %6, %7 neg_hi:[0,1]\n\t"                                   // A = zk + conj(zm)
        "v_pk_add_f32 %1, %6, %7 neg_lo:[0,1]\n\t"                                   // B = zk - conj(zm)
        "v_pk_mul_f32 %2, %1, %8 op_sel:[0,1] op_sel_hi:[0,0] neg_hi:[1,0]\n\t"      // t = (B.x*w.y, -B.x*w.x)
        "v_pk_fma_f32 %3, %1, %8, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"           // P = (B.y*w.x, B.y*w.y) + t
        "v_pk_add_f32 %4, %0, %3 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]\n\t"      // U = (A.x + P.x, A.x - P.x)
        "v_pk_add_f32 %5, %0, %3 op_sel:[1,1] op_sel_hi:[1,1] neg_hi:[0,1]\n\t"      // V = (A.y + P.y, A.y - P.y)
        "v_pk_mul_f32 %4, %4, %4\n\t"
        "v_pk_mul_f32 %5, %5, %5\n\t"
        "v_pk_add_f32 %4, %4, %5"                                                     // (|A + P|^2, |A - P|^2)
        : "=&v"(a), "=&v"(b), "=&v"(t), "=&v"(p), "=&v"(u), "=&v"(v)
        : "v"(zk), "v"(zm), "v"(w));
    sq_k = u.x;
    sq_mirror = u.y;
}

// single-bin form: |A + P|^2 only; 6 packed instructions + 1 add
__device__ __forceinline__ float split_one_sq(f2 zk, f2 zm, f2 w) {
    f2 a, b, t, p;
    asm("v_pk_add_f32 %0, %4, %5 neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %1, %4, %5 neg_lo:[0,1]\n\t"
        "v_pk_mul_f32 %2, %1, %6 op_sel:[0,1] op_sel_hi:[0,0] neg_hi:[1,0]\n\t"
        "v_pk_fma_f32 %3, %1, %6, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_add_f32 %0, %0, %3\n\t"
        "v_pk_mul_f32 %0, %0, %0"
        : "=&v"(a), "=&v"(b), "=&v"(t), "=&v"(p)
        : "v"(zk), "v"(zm), "v"(w));
    return a.x + a.y;
}

// magnitude from 4 x |X|^2 with v_sqrt_f32 (<= 1 ulp; scaling by powers of two stays exact)
__device__ __forceinline__ float mag_from_sq4(float sq4) { return 0.5f * __builtin_amdgcn_sqrtf(sq4); }
// the kernels feed the transform with HALF the window (see the table construction in blissgpu.hip): Z arrives halved,
// A + P is X itself and the magnitude needs no further scaling
__device__ __forceinline__ float mag_from_sq(float sq) { return __builtin_amdgcn_sqrtf(sq); }

}  // namespace bg
