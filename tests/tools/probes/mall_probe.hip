// mall_probe.hip -- does a freshly WRITTEN stretch of the size of a song group's spectrogram come back from the 256 MiB
// Infinity Cache (MALL) when it is read a moment later?  (Round-5 review item 6: a song-group pipeline stft8192 -> tune ->
// chroma whose group stays cache resident would spare the 65 GB spectrogram round trip of a 1024-song step.)
//   build: hipcc --offload-arch=gfx950 -O2 -o tests/tools/probes/mall_probe tests/tools/probes/mall_probe.hip
// For sizes S: write S with plain / non-temporal 16-byte stores (as stft8192_kernel writes its rows), optionally stream T MB
// of unrelated reads + writes in between (the other kernels of the step), then time a 16-byte-load read of S.
// Prints the read rate; "cold" = the same read after 2 GB of unrelated traffic.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void write_k(f4* p, size_t n4, float v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f4 q = {v, v + 1.0f, v + 2.0f, (float)i};
        if (NT) __builtin_nontemporal_store(q, p + i);
        else p[i] = q;
    }
}
__global__ __launch_bounds__(256) void read_k(const f4* p, size_t n4, float* sink) {
    f4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) acc += p[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}
__global__ __launch_bounds__(256) void stream_k(const f4* src, f4* dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = src[i] * 1.0001f;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const size_t MB = 1u << 20;
    f4 *a, *s1, *s2;
    float* sink;
    CK(hipMalloc(&a, 1024 * MB)); CK(hipMalloc(&s1, 1024 * MB)); CK(hipMalloc(&s2, 1024 * MB)); CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 8;
    auto flush = [&]() { stream_k<<<grid, 256>>>(s1, s2, 1024 * MB / 16); };
    printf("size_MB store   between_MB  read_GBps (median of 5)\n");
    for (size_t S : {16, 32, 64, 120, 200, 256, 400, 800}) {
        for (int nt = 0; nt < 2; nt++) {
            for (size_t T : {(size_t)0, (size_t)64, (size_t)250, (size_t)2048}) {
                std::vector<float> ms;
                for (int rep = 0; rep < 5; rep++) {
                    flush(); flush();
                    const size_t n4 = S * MB / 16;
                    if (nt) write_k<true><<<grid, 256>>>(a, n4, (float)rep); else write_k<false><<<grid, 256>>>(a, n4, (float)rep);
                    for (size_t done = 0; done < T; done += 512) {  // unrelated traffic: T MB read + T MB written
                        const size_t part = (T - done < 512 ? T - done : 512) * MB / 16;
                        stream_k<<<grid, 256>>>(s1 + (done % 512) * MB / 16, s2 + (done % 512) * MB / 16, part);
                    }
                    hipEventRecord(e0);
                    read_k<<<grid, 256>>>(a, n4, sink);
                    hipEventRecord(e1);
                    CK(hipEventSynchronize(e1));
                    float t;
                    hipEventElapsedTime(&t, e0, e1);
                    ms.push_back(t);
                }
                std::sort(ms.begin(), ms.end());
                printf("%7zu %-7s %10zu  %8.1f\n", S, nt ? "nt" : "plain", T, (double)S * MB / (ms[2] * 1e-3) / 1e9);
            }
        }
    }
    return 0;
}
