// Operand layout of v_mfma_f64_4x4x4_4b_f64 (developer tool): one wavefront multiplies random blocks; the host tries the
// candidate lane mappings (i + 4 k + 16 blk  vs  i + 4 blk + 16 k, for A, B and D independently) and prints the one that fits.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k(const double* a, const double* b, double* d) {
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}
static int lane_of(int g, int x, int kk, int blk) { return g == 0 ? x + 4 * kk + 16 * blk : x + 4 * blk + 16 * kk; }
int main() {
    double ha[64], hb[64], hd[64], *a, *b, *d;
    srand(7);
    for (int i = 0; i < 64; i++) { ha[i] = rand() % 17 - 8; hb[i] = rand() % 13 - 6; }
    hipMalloc(&a, 512); hipMalloc(&b, 512); hipMalloc(&d, 512);
    hipMemcpy(a, ha, 512, hipMemcpyHostToDevice); hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd, d, 512, hipMemcpyDeviceToHost);
    for (int ga = 0; ga < 2; ga++) for (int gb = 0; gb < 2; gb++) for (int gd = 0; gd < 2; gd++) {
        bool ok = true;
        for (int blk = 0; blk < 4 && ok; blk++) for (int i = 0; i < 4 && ok; i++) for (int j = 0; j < 4 && ok; j++) {
            double s = 0;
            for (int kk = 0; kk < 4; kk++) s += ha[lane_of(ga, i, kk, blk)] * hb[lane_of(gb, j, kk, blk)];
            // D: lane_of(gd, x, y, blk) with (x, y) = (i, j) or (j, i)
            if (hd[lane_of(gd, i, j, blk)] != s) ok = false;
        }
        if (ok) printf("A: %s   B: %s (x = column j)   D[i][j]: lane = %s with x = i, k-slot = j\n",
                       ga ? "i + 4 blk + 16 k" : "i + 4 k + 16 blk", gb ? "j + 4 blk + 16 k" : "j + 4 k + 16 blk",
                       gd ? "i + 4 blk + 16 j" : "i + 4 j + 16 blk");
        bool okT = true;
        for (int blk = 0; blk < 4 && okT; blk++) for (int i = 0; i < 4 && okT; i++) for (int j = 0; j < 4 && okT; j++) {
            double s = 0;
            for (int kk = 0; kk < 4; kk++) s += ha[lane_of(ga, i, kk, blk)] * hb[lane_of(gb, j, kk, blk)];
            if (hd[lane_of(gd, j, i, blk)] != s) okT = false;
        }
        if (okT) printf("A: %s   B: %s (x = column j)   D[i][j]: lane = %s with x = j, k-slot = i\n",
                        ga ? "i + 4 blk + 16 k" : "i + 4 k + 16 blk", gb ? "j + 4 blk + 16 k" : "j + 4 k + 16 blk",
                        gd ? "j + 4 blk + 16 i" : "j + 4 i + 16 blk");
    }
    printf("d[0..7] = %g %g %g %g %g %g %g %g\n", hd[0], hd[1], hd[2], hd[3], hd[4], hd[5], hd[6], hd[7]);
    return 0;
}
