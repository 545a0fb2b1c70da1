// micro-benchmark: achievable HBM write bandwidth for the pairwise output pattern (developer aid)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
// each wave writes `chunk_lanes`*16 B contiguous per row; a WG covers ROWS rows x (waves_x*64*4) cols
template <int NT>
__global__ void wr(float* out, size_t n, size_t ld, int rows_per_wg, int waves_x) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wx = wave % waves_x, wy = wave / waves_x, waves_y = 4 / waves_x;
    const size_t j0 = ((size_t)blockIdx.x * waves_x + wx) * 256 + lane * 4;
    const size_t i0 = (size_t)blockIdx.y * rows_per_wg;
    if (j0 + 3 >= n) return;
    for (int r = wy; r < rows_per_wg; r += waves_y) {
        f4 v; v.x = r; v.y = lane; v.z = j0; v.w = 1.0f;
        f4* p = reinterpret_cast<f4*>(out + (i0 + r) * ld + j0);
        if (NT) __builtin_nontemporal_store(v, p); else *p = v;
    }
}
int main() {
    const size_t n = 100000; float* d; hipMalloc(&d, n * n * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    struct Cfg { int rows, wx, nt; } cfgs[] = {{128,1,0},{128,1,1},{32,1,1},{128,2,1},{128,4,1},{32,4,1},{8,4,1},{512,1,1}};
    for (auto c : cfgs) {
        dim3 grid((unsigned)((n + 256 * c.wx - 1) / (256 * c.wx)), (unsigned)(n / c.rows));
        for (int it = 0; it < 2; it++) {
            hipEventRecord(a);
            if (c.nt) wr<1><<<grid, 256>>>(d, n, n, c.rows, c.wx); else wr<0><<<grid, 256>>>(d, n, n, c.rows, c.wx);
            hipEventRecord(b); hipEventSynchronize(b);
        }
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("rows/WG %4d waves_x %d nt %d : %7.3f ms  %6.0f GB/s\n", c.rows, c.wx, c.nt, ms, 4.0 * n * (n / c.rows * c.rows) / ms / 1e6);
    }
    return 0;
}
