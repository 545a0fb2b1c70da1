// HBM read-bandwidth probe (developer tool, not part of the library): the same 30 GB of "spectrogram" read
//   (a) linearly, one float4 per lane, consecutive lanes on consecutive addresses;
//   (b) the way chroma_kernel reads it: a wavefront walks 64 rows (pitch 4104 floats), 64 contiguous bytes per row per
//       load instruction, KU = 4 instructions per row in flight;
//   (c) row-contiguous: a wavefront reads 1 KB of ONE row per load instruction (what an LDS-staged kernel would issue).
// build: hipcc --offload-arch=gfx950 -O3 -o bw_probe bw_probe.hip ; run: ./bw_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
constexpr int PITCH = 4104;
__global__ __launch_bounds__(256) void linear_read(const float4* __restrict__ p, size_t n4, float* out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    float acc = 0.f;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const float4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc += a.x + b.y + c.z + d.w;
    }
    if (acc == 123.456f) out[0] = acc;
}
// (b) one wave = 64 rows; lane (i16, g): row 16 q + i16, floats 4 g + 16 st
__global__ __launch_bounds__(256) void chroma_like(const float* __restrict__ spec, uint32_t n_rows, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x * 4 + wave;
    const uint32_t f0 = tile * 64;
    if (f0 >= n_rows) return;
    const int i16 = lane & 15, g = lane >> 4;
    const float* brow[4];
    for (int q = 0; q < 4; q++) {
        uint32_t fj = f0 + 16 * q + i16;
        if (fj >= n_rows) fj = n_rows - 1;
        brow[q] = spec + (size_t)fj * PITCH + 4 * g;
    }
    float acc = 0.f;
#pragma unroll 1
    for (int st = 0; st + 4 <= PITCH / 16; st += 4) {
        float4 b[4][4];
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int q = 0; q < 4; q++) b[u][q] = *reinterpret_cast<const float4*>(brow[q] + 16 * (st + u));
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int q = 0; q < 4; q++) acc += b[u][q].x + b[u][q].w;
    }
    if (acc == 123.456f) out[0] = acc;
}
// (e) chroma_like with the four lanes of a row spread over a whole 128-byte line: lane g reads 16 bytes at 32 g (+16 in
//     the next instruction), so that every instruction touches both 64-byte halves of the line
__global__ __launch_bounds__(256) void chroma_spread(const float* __restrict__ spec, uint32_t n_rows, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x * 4 + wave;
    const uint32_t f0 = tile * 64;
    if (f0 >= n_rows) return;
    const int i16 = lane & 15, g = lane >> 4;
    const float* brow[4];
    for (int q = 0; q < 4; q++) {
        uint32_t fj = f0 + 16 * q + i16;
        if (fj >= n_rows) fj = n_rows - 1;
        brow[q] = spec + (size_t)fj * PITCH + 8 * g;
    }
    float acc = 0.f;
#pragma unroll 1
    for (int st = 0; st + 4 <= PITCH / 16; st += 4) {
        float4 b[4][4];
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int u = 0; u < 4; u++) b[u][q] = *reinterpret_cast<const float4*>(brow[q] + 16 * st + 32 * (u >> 1) + 4 * (u & 1));
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int q = 0; q < 4; q++) acc += b[u][q].x + b[u][q].w;
    }
    if (acc == 123.456f) out[0] = acc;
}
// (c) one wave = 64 rows, visited one row at a time: 64 lanes x float4 = 1 KB contiguous per instruction, 4 rows in flight
__global__ __launch_bounds__(256) void row_contiguous(const float* __restrict__ spec, uint32_t n_rows, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x * 4 + wave;
    const uint32_t f0 = tile * 64;
    if (f0 >= n_rows) return;
    float acc = 0.f;
#pragma unroll 1
    for (int k0 = 0; k0 + 256 <= PITCH; k0 += 256) {       // 1 KB column block
#pragma unroll 1
        for (int r = 0; r < 64; r += 16) {
            float4 b[16];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                uint32_t fj = f0 + r + u;
                if (fj >= n_rows) fj = n_rows - 1;
                b[u] = *reinterpret_cast<const float4*>(spec + (size_t)fj * PITCH + k0 + 4 * lane);
            }
#pragma unroll
            for (int u = 0; u < 16; u++) acc += b[u].x + b[u].w;
        }
    }
    if (acc == 123.456f) out[0] = acc;
}
// (d) R rows per load instruction: lanes [j * 64/R, (j+1) * 64/R) read 1024/R contiguous bytes of row j; 16 instructions in flight
template <int R, int LDS_FLOATS = 1>
__global__ __launch_bounds__(256) void rows_per_instr(const float* __restrict__ spec, uint32_t n_rows, float* out) {
    __shared__ float occupancy_limiter[LDS_FLOATS];  // > 80 KB: one workgroup (four wavefronts) per CU
    if (LDS_FLOATS > 1 && n_rows == 7) occupancy_limiter[threadIdx.x] = 1.0f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x * 4 + wave;
    const uint32_t f0 = tile * 64;
    if (f0 >= n_rows) return;
    constexpr int LPR = 64 / R;          // lanes per row
    constexpr int CHUNK = LPR * 4;       // floats per row per instruction
    const int rsub = lane / LPR, col = (lane % LPR) * 4;
    float acc = 0.f;
#pragma unroll 1
    for (int k0 = 0; k0 + CHUNK <= PITCH; k0 += CHUNK) {
#pragma unroll 1
        for (int r = 0; r < 64; r += (R <= 4 ? 16 * R : 64)) {
            constexpr int NU = R <= 4 ? 16 : 64 / R;
            float4 b[16];
#pragma unroll
            for (int u = 0; u < 16; u++) b[u] = float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < NU; u++) {
                uint32_t fj = f0 + r + u * R + rsub;
                if (fj >= n_rows) fj = n_rows - 1;
                b[u] = *reinterpret_cast<const float4*>(spec + (size_t)fj * PITCH + k0 + col);
            }
#pragma unroll
            for (int u = 0; u < 16; u++) acc += b[u].x + b[u].w;
        }
    }
    if (acc == 123.456f) out[0] = acc;
}
int main() {
    const uint32_t n_rows = 1024u * 1797u;
    const size_t bytes = (size_t)n_rows * PITCH * 4;
    float *spec, *out;
    if (hipMalloc(&spec, bytes + 4096) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(spec, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto&& launch, double frac) {
        launch(); hipDeviceSynchronize();
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%-16s %8.3f ms  %7.1f GB/s\n", name, best, frac * bytes / best * 1e-6);
    };
    for (int wgs : {2048, 8192, 32768})
        timeit(wgs == 2048 ? "linear 2048wg" : wgs == 8192 ? "linear 8192wg" : "linear 32768wg",
               [&] { hipLaunchKernelGGL(linear_read, dim3(wgs), dim3(256), 0, 0, (const float4*)spec, bytes / 16, out); }, 1.0);
    const uint32_t tiles = (n_rows + 63) / 64;
    timeit("chroma-like", [&] { hipLaunchKernelGGL(chroma_like, dim3((tiles + 3) / 4), dim3(256), 0, 0, spec, n_rows, out); }, 4096.0 / PITCH);
    timeit("chroma-spread", [&] { hipLaunchKernelGGL(chroma_spread, dim3((tiles + 3) / 4), dim3(256), 0, 0, spec, n_rows, out); }, 4096.0 / PITCH);
    timeit("row-contiguous", [&] { hipLaunchKernelGGL(row_contiguous, dim3((tiles + 3) / 4), dim3(256), 0, 0, spec, n_rows, out); }, 4096.0 / PITCH);
    timeit("2 rows x 512B", [&] { hipLaunchKernelGGL(rows_per_instr<2>, dim3((tiles + 3) / 4), dim3(256), 0, 0, spec, n_rows, out); }, 4096.0 / PITCH);
    timeit("4 rows x 256B", [&] { hipLaunchKernelGGL(rows_per_instr<4>, dim3((tiles + 3) / 4), dim3(256), 0, 0, spec, n_rows, out); }, 4096.0 / PITCH);
    timeit("8 rows x 128B", [&] { hipLaunchKernelGGL(rows_per_instr<8>, dim3((tiles + 3) / 4), dim3(256), 0, 0, spec, n_rows, out); }, 4096.0 / PITCH);
    timeit("4x256B 1wg/CU", [&] { hipLaunchKernelGGL((rows_per_instr<4, 21000>), dim3((tiles + 3) / 4), dim3(256), 0, 0, spec, n_rows, out); }, 4096.0 / PITCH);
    timeit("4x256B 2wg/CU", [&] { hipLaunchKernelGGL((rows_per_instr<4, 12000>), dim3((tiles + 3) / 4), dim3(256), 0, 0, spec, n_rows, out); }, 4096.0 / PITCH);
    timeit("8x128B 2wg/CU", [&] { hipLaunchKernelGGL((rows_per_instr<8, 12000>), dim3((tiles + 3) / 4), dim3(256), 0, 0, spec, n_rows, out); }, 4096.0 / PITCH);
    return 0;
}
