// device_utils.hpp -- wavefront (64-lane) reductions/scans and complex helpers for gfx950.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace bg {

constexpr int WAVE = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, WAVE));
    return v;
}

__device__ __forceinline__ double wave_prod(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v *= __shfl_xor(v, off, WAVE);
    return v;
}

// inclusive scan across the wave
__device__ __forceinline__ float wave_scan_incl(float v) {
    const int l = lane_id();
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        const float o = __shfl_up(v, off, WAVE);
        if (l >= off) v += o;
    }
    return v;
}

__device__ __forceinline__ uint32_t wave_scan_incl_u32(uint32_t v) {
    const int l = lane_id();
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        const uint32_t o = __shfl_up(v, off, WAVE);
        if (l >= off) v += o;
    }
    return v;
}

// the same scan with DPP only (row_shr 1/2/4/8 inside the 16-lane rows, then row_bcast:15 / row_bcast:31 across them):
// six VALU instructions, no trip through the LDS crossbar and its latency (ds_bpermute)
__device__ __forceinline__ uint32_t wave_scan_incl_u32_dpp(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);  // row_shr:1, zero fill
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
    return v;
}

// wave-wide maximum with DPP row operations + four v_readlane (no ds_bpermute: its LDS-crossbar latency would be paid six
// times in a row); every lane gets the result
__device__ __forceinline__ float wave_max_dpp(float v) {
    auto dpp = [](float x, auto ctrl) {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, 0xF, 0xF, true));
    };
    v = fmaxf(v, dpp(v, std::integral_constant<int, 0xB1>{}));    // quad_perm xor 1
    v = fmaxf(v, dpp(v, std::integral_constant<int, 0x4E>{}));    // quad_perm xor 2
    v = fmaxf(v, dpp(v, std::integral_constant<int, 0x141>{}));   // row_half_mirror
    v = fmaxf(v, dpp(v, std::integral_constant<int, 0x140>{}));   // row_mirror: every lane of a row holds the row maximum
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// ---- complex helpers ----
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// multiply by -i
__device__ __forceinline__ float2 cmul_mi(float2 a) { return make_float2(a.y, -a.x); }

// forward radix-4 butterfly (W4 = -i), in place
__device__ __forceinline__ void radix4(float2& v0, float2& v1, float2& v2, float2& v3) {
    const float2 t0 = cadd(v0, v2), t1 = csub(v0, v2);
    const float2 t2 = cadd(v1, v3), t3 = cmul_mi(csub(v1, v3));
    v0 = cadd(t0, t2);
    v2 = csub(t0, t2);
    v1 = cadd(t1, t3);
    v3 = csub(t1, t3);
}

// One Stockham autosort radix-4 pass on an N-point sequence held in LDS.
//   j      : butterfly index in [0, N/4)
//   Ns     : product of the radices already applied (1, 4, 16, ...)
//   tw     : exp(-2*pi*i*m/TWN) table; tw_stride = TWN / N
// Reads x[j + r*N/4], writes y[(j/Ns)*4*Ns + (j%Ns) + r*Ns].
template <int N>
__device__ __forceinline__ void stockham_r4(const float2* __restrict__ x, float2* __restrict__ y, int j, int Ns,
                                            const float2* __restrict__ tw, int tw_stride) {
    constexpr int Q = N / 4;
    float2 v0 = x[j], v1 = x[j + Q], v2 = x[j + 2 * Q], v3 = x[j + 3 * Q];
    const int k = j & (Ns - 1);
    if (Ns > 1) {
        const int step = (Q / Ns) * tw_stride;  // N/(4*Ns) * tw_stride
        v1 = cmul(v1, tw[k * step]);
        v2 = cmul(v2, tw[2 * k * step]);
        v3 = cmul(v3, tw[3 * k * step]);
    }
    radix4(v0, v1, v2, v3);
    const int j0 = ((j - k) << 2) + k;
    y[j0] = v0;
    y[j0 + Ns] = v1;
    y[j0 + 2 * Ns] = v2;
    y[j0 + 3 * Ns] = v3;
}

// Real-input split: given Z = FFT_M(z), z[n] = x[2n] + i x[2n+1], return X[k] of the 2M-point real FFT.
//   zk = Z[k], zmk = Z[(M-k) % M], w = exp(-2*pi*i*k/(2M))
__device__ __forceinline__ float2 real_split(float2 zk, float2 zmk, float2 w) {
    const float2 b = make_float2(zmk.x, -zmk.y);
    const float2 e = make_float2(0.5f * (zk.x + b.x), 0.5f * (zk.y + b.y));
    const float2 d = make_float2(0.5f * (zk.x - b.x), 0.5f * (zk.y - b.y));
    const float2 o = make_float2(d.y, -d.x);  // -i * d
    return cadd(e, cmul(w, o));
}

__device__ __forceinline__ uint64_t f64_key(double v) {
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_f64(uint64_t k) {
    const uint64_t b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double((long long)b);
}

}  // namespace bg
