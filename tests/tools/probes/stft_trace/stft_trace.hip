// Probe translation unit: the library's kernels_chroma.hip with the per-phase cycle trace of stft8192_kernel switched on.
// Wall-clock shader cycles of wave 0 of every workgroup between the TRACE(k) marks of the frame loop, summed per mark over
// the launches (LDS counters, flushed once per workgroup); tests/tools/kbench prints the table when the library exports
// blissgpu_debug_stft_trace.  Build: tests/tools/variant.sh trace --tu tests/tools/probes/stft_trace/stft_trace.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
__device__ unsigned long long g_stft_trace[32];
#define TRACE_DECL __shared__ uint32_t trace_acc[32]; uint32_t trace_prev = 0;
#define TRACE_INIT() do { if (threadIdx.x < 32) trace_acc[threadIdx.x] = 0; } while (0)
#define TRACE_START() do { trace_prev = (uint32_t)__builtin_amdgcn_s_memtime(); } while (0)
#define TRACE(k) do { const uint32_t now_ = (uint32_t)__builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) atomicAdd(&trace_acc[k], now_ - trace_prev); trace_prev = now_; } while (0)
#define TRACE_FLUSH() do { __syncthreads(); if (threadIdx.x < 32 && trace_acc[threadIdx.x]) atomicAdd(&g_stft_trace[threadIdx.x], (unsigned long long)trace_acc[threadIdx.x]); } while (0)
#include "../../../../bliss-rs_amd/csrc/kernels_chroma.hip"

extern "C" int blissgpu_debug_stft_trace(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stft_trace), sizeof(g_stft_trace)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[32] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_stft_trace), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
