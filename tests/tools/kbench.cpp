// kbench.cpp -- per-kernel timing of one analysis batch through the C ABI, without Python: the quick A/B loop for kernel
// work on the GPU box (a fresh box pays 1-2 minutes for its first `import torch`; this pays nothing).
//   build: g++ -std=c++17 -O1 -o tests/tools/kbench tests/tools/kbench.cpp -ldl
//   usage: kbench <libblissgpu.so> [songs=256] [seconds=180] [steps=3] [ragged=0]
//          kbench <libblissgpu.so> pairwise [n=100000] [reps=3]      the n x n euclidean matrix, self (A == B) and general
//          KBENCH_DETERMINISM=R kbench <lib> [songs] [seconds] ...   R extra runs, every row compared bit for bit with the
//                                                                    first run's (KBENCH_WS_LIMIT_MB cuts the batch into chunks)
// Prints ms per kernel per step (HIP events on the stream each kernel runs on; KBENCH_SERIAL=1 for every kernel alone,
// KBENCH_TAIL_MODE / KBENCH_PIPELINE_CHUNKS set the other scheduling options), the step's wall time, and an FNV-1a hash of the feature rows (two builds that agree bit for bit print the same hash).
#include <dlfcn.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/blissgpu.h"

#define SYM(name) auto p_##name = (decltype(&name))dlsym(h, #name); if (!p_##name) { std::fprintf(stderr, "missing %s\n", #name); return 2; }
#define OK(expr) do { int rc_ = (expr); if (rc_) { std::fprintf(stderr, "FAILED %s -> %d (%s)\n", #expr, rc_, p_blissgpu_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: kbench <lib> [songs] [seconds] [steps] [ragged]\n"); return 2; }
    void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { std::fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    if (argc > 2 && std::strcmp(argv[2], "pairwise") == 0) {
        const uint64_t m = argc > 3 ? (uint64_t)std::atoll(argv[3]) : 100000;
        const int reps = argc > 4 ? std::atoi(argv[4]) : 3;
        SYM(blissgpu_last_error) SYM(blissgpu_ctx_create) SYM(blissgpu_ctx_destroy) SYM(blissgpu_malloc) SYM(blissgpu_free)
        SYM(blissgpu_memcpy_h2d) SYM(blissgpu_memcpy_d2h) SYM(blissgpu_pairwise_device) SYM(blissgpu_ctx_synchronize)
        blissgpu_ctx* c = nullptr;
        OK(p_blissgpu_ctx_create(0, &c));
        std::vector<float> h(m * 23);
        uint32_t s = 1234567u;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) * (2.0f / 16777216.0f) - 1.0f; }
        float *dA = nullptr, *dB = nullptr, *dD = nullptr;
        OK(p_blissgpu_malloc((void**)&dA, m * 23 * 4)); OK(p_blissgpu_malloc((void**)&dB, m * 23 * 4)); OK(p_blissgpu_malloc((void**)&dD, m * m * 4));
        OK(p_blissgpu_memcpy_h2d(c, dA, h.data(), m * 23 * 4)); OK(p_blissgpu_memcpy_h2d(c, dB, h.data(), m * 23 * 4));
        for (int general = 0; general < 2; general++) {
            const float* rhs = general ? dB : dA;
            OK(p_blissgpu_pairwise_device(c, dA, m, rhs, m, 23, BLISSGPU_METRIC_EUCLIDEAN, nullptr, dD, m));
            OK(p_blissgpu_ctx_synchronize(c));
            auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < reps; r++) OK(p_blissgpu_pairwise_device(c, dA, m, rhs, m, 23, BLISSGPU_METRIC_EUCLIDEAN, nullptr, dD, m));
            OK(p_blissgpu_ctx_synchronize(c));
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
            // hash of 64 whole rows spread over the matrix (two builds that agree bit for bit print the same value)
            uint64_t hash = 1469598103934665603ull;
            std::vector<float> row(m);
            for (int q = 0; q < 64; q++) {
                const uint64_t i = (uint64_t)q * (m - 1) / 63;
                OK(p_blissgpu_memcpy_d2h(c, row.data(), dD + i * m, m * 4));
                const unsigned char* b = (const unsigned char*)row.data();
                for (size_t z = 0; z < m * 4; z++) { hash ^= b[z]; hash *= 1099511628211ull; }
            }
            std::printf("pairwise %s n=%llu ms=%.3f pairs_per_s=%.4g GBps=%.0f hash=%016llx\n", general ? "general(A!=B)" : "self(A==B)",
                        (unsigned long long)m, ms, (double)m * m / (ms * 1e-3), (4.0 * m * m + 8.0 * 23 * m) / (ms * 1e-3) / 1e9,
                        (unsigned long long)hash);
        }
        p_blissgpu_free(dA); p_blissgpu_free(dB); p_blissgpu_free(dD);
        p_blissgpu_ctx_destroy(c);
        return 0;
    }
    const uint32_t n = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 256;
    const double seconds = argc > 3 ? std::atof(argv[3]) : 180.0;
    const int steps = argc > 4 ? std::atoi(argv[4]) : 3;
    const int ragged = argc > 5 ? std::atoi(argv[5]) : 0;
    SYM(blissgpu_last_error) SYM(blissgpu_ctx_create) SYM(blissgpu_ctx_destroy) SYM(blissgpu_malloc) SYM(blissgpu_free)
    SYM(blissgpu_synth_white_noise_device) SYM(blissgpu_analyze_batch_device) SYM(blissgpu_ctx_synchronize)
    SYM(blissgpu_profile_enable) SYM(blissgpu_profile_reset) SYM(blissgpu_profile_kernel_count) SYM(blissgpu_profile_kernel_name)
    SYM(blissgpu_profile_get) SYM(blissgpu_memcpy_d2h) SYM(blissgpu_ctx_set_option)
    blissgpu_ctx* c = nullptr;
    OK(p_blissgpu_ctx_create(0, &c));
    SYM(blissgpu_ctx_set_workspace_limit)
    if (const char* e = std::getenv("KBENCH_WS_LIMIT_MB")) OK(p_blissgpu_ctx_set_workspace_limit(c, (uint64_t)std::atoll(e) << 20));
    if (const char* e = std::getenv("KBENCH_SERIAL")) OK(p_blissgpu_ctx_set_option(c, BLISSGPU_OPT_SERIAL, std::atoi(e)));
    if (const char* e = std::getenv("KBENCH_TAIL_MODE")) OK(p_blissgpu_ctx_set_option(c, BLISSGPU_OPT_TAIL_MODE, std::atoi(e)));
    if (const char* e = std::getenv("KBENCH_FLUX_ORDER")) OK(p_blissgpu_ctx_set_option(c, BLISSGPU_OPT_FLUX_ORDER, std::atoi(e)));
    if (const char* e = std::getenv("KBENCH_STFT_SHAPE")) OK(p_blissgpu_ctx_set_option(c, BLISSGPU_OPT_STFT_SHAPE, std::atoi(e)));
    if (const char* e = std::getenv("KBENCH_TAIL_SPLIT")) OK(p_blissgpu_ctx_set_option(c, BLISSGPU_OPT_TAIL_SPLIT, std::atoi(e)));
    if (const char* e = std::getenv("KBENCH_PIPELINE_CHUNKS")) OK(p_blissgpu_ctx_set_option(c, BLISSGPU_OPT_PIPELINE_CHUNKS, std::atoi(e)));
    std::vector<uint64_t> offs(n), lens(n);
    uint64_t total = 0;
    for (uint32_t i = 0; i < n; i++) {
        uint64_t len = (uint64_t)(seconds * 22050.0);
        if (ragged) len = 8192 + (uint64_t)((i * 2654435761u) % (uint32_t)(len - 8192 + 1));
        offs[i] = total; lens[i] = len; total += (len + 63) / 64 * 64;
    }
    float *d_pcm = nullptr, *d_out = nullptr;
    OK(p_blissgpu_malloc((void**)&d_pcm, total * 4));
    OK(p_blissgpu_malloc((void**)&d_out, (uint64_t)n * 23 * 4));
    OK(p_blissgpu_synth_white_noise_device(c, d_pcm, offs.data(), lens.data(), n, 0));
    OK(p_blissgpu_analyze_batch_device(c, d_pcm, offs.data(), lens.data(), n, 2, d_out, nullptr));  // warm-up (allocations)
    OK(p_blissgpu_ctx_synchronize(c));
    // plain steps: the wall time of a step as a caller sees it
    auto t0 = std::chrono::steady_clock::now();
    for (int s = 0; s < steps; s++) OK(p_blissgpu_analyze_batch_device(c, d_pcm, offs.data(), lens.data(), n, 2, d_out, nullptr));
    OK(p_blissgpu_ctx_synchronize(c));
    const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / steps;
    // profiled steps: HIP events around every launch
    OK(p_blissgpu_profile_enable(c, 1));
    OK(p_blissgpu_profile_reset(c));
    for (int s = 0; s < steps; s++) OK(p_blissgpu_analyze_batch_device(c, d_pcm, offs.data(), lens.data(), n, 2, d_out, nullptr));
    OK(p_blissgpu_ctx_synchronize(c));
    std::printf("%s songs=%u seconds=%g ragged=%d step_ms=%.3f", argv[1], n, seconds, ragged, wall_ms);
    for (int k = 0; k < p_blissgpu_profile_kernel_count(); k++) {
        double ms = 0; uint64_t launches = 0;
        OK(p_blissgpu_profile_get(c, k, &ms, &launches));
        if (launches) std::printf(" %s=%.3f", p_blissgpu_profile_kernel_name(k), ms / steps);
    }
    if (auto p_trace = (int (*)(unsigned long long*, int))dlsym(h, "blissgpu_debug_stft_trace")) {  // tests/tools/probes/stft_trace builds only
        unsigned long long tr[32];
        if (p_trace(tr, 1) == 0) {
            unsigned long long tot = 0;
            for (int k = 0; k < 32; k++) tot += tr[k];
            std::printf("\n  stft trace (%% of wave 0's cycles per mark):");
            for (int k = 0; k < 32; k++) if (tr[k]) std::printf(" [%d]=%.1f", k, 100.0 * (double)tr[k] / (double)tot);
            std::printf("  total cycles/frame-ish=%.0f", (double)tot);
        }
    }
    std::vector<float> rows((size_t)n * 23);
    OK(p_blissgpu_memcpy_d2h(c, rows.data(), d_out, rows.size() * 4));
    if (const char* e = std::getenv("KBENCH_DETERMINISM")) {
        const int runs = std::atoi(e);
        OK(p_blissgpu_profile_enable(c, 0));
        std::vector<float> again(rows.size());
        long bad_rows = 0, bad_runs = 0;
        for (int r = 0; r < runs; r++) {
            OK(p_blissgpu_analyze_batch_device(c, d_pcm, offs.data(), lens.data(), n, 2, d_out, nullptr));
            OK(p_blissgpu_ctx_synchronize(c));
            OK(p_blissgpu_memcpy_d2h(c, again.data(), d_out, again.size() * 4));
            long b = 0;
            for (uint32_t i = 0; i < n; i++)
                if (std::memcmp(&again[(size_t)i * 23], &rows[(size_t)i * 23], 23 * 4) != 0) {
                    if (b < 3) {
                        std::printf("\n  run %d song %u differs:", r, i);
                        for (int k = 0; k < 23; k++) if (again[(size_t)i * 23 + k] != rows[(size_t)i * 23 + k]) std::printf(" [%d] %.9g vs %.9g", k, again[(size_t)i * 23 + k], rows[(size_t)i * 23 + k]);
                    }
                    b++;
                }
            bad_rows += b; bad_runs += b != 0;
        }
        std::printf("\n  determinism: %d runs x %u songs, %ld differing rows in %ld runs", runs, n, bad_rows, bad_runs);
    }
    uint64_t hash = 1469598103934665603ull;
    const unsigned char* b = (const unsigned char*)rows.data();
    for (size_t i = 0; i < rows.size() * 4; i++) { hash ^= b[i]; hash *= 1099511628211ull; }
    std::printf(" hash=%016llx row0=[%.6f %.6f %.6f ... %.6f]\n", (unsigned long long)hash, rows[0], rows[1], rows[2], rows[22]);
    p_blissgpu_free(d_pcm); p_blissgpu_free(d_out);
    p_blissgpu_ctx_destroy(c);
    return 0;
}
