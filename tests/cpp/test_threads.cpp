// Re-entrancy test of the C ABI, shaped like the reference's bulk path (src/song/decoder.rs:299-329): N worker threads,
// each calling Song::analyze on one song at a time -- here blissgpu_analyze, whose concurrent callers are coalesced into
// device batches -- mixed with Song::distance and closest_to_songs calls on the same process-wide context.
// Every threaded result must be bit-identical to the serial run.
//   usage: test_threads [n_threads] [calls_per_thread]        prints timing lines "key value" for bench.py
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/blissgpu.h"

#define CHECK(cond) do { if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #cond, blissgpu_last_error()); std::exit(1); } } while (0)

static std::vector<float> noise(uint32_t seed, size_t n) {  // uniform [-0.5, 0.5), xorshift32
    std::vector<float> x(n);
    uint32_t s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; i++) {
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        x[i] = (float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f;
    }
    return x;
}

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const int n_threads = argc > 1 ? std::atoi(argv[1]) : 16;
    const int calls = argc > 2 ? std::atoi(argv[2]) : 32;
    const int n_songs = 32, d = 23;
    std::vector<std::vector<float>> songs;
    for (int i = 0; i < n_songs; i++) {
        size_t len = 22050 * (size_t)(3 + (i * 7) % 20) + (size_t)i * 131;  // 3 .. 22 s, ragged
        if (i == 5) len = 4000;                                             // too short
        songs.push_back(noise(100 + i, len));
    }
    // ---- serial reference rows (also: single-song latency) ----
    std::vector<float> ref((size_t)n_songs * d);
    std::vector<int32_t> ref_st(n_songs);
    CHECK(blissgpu_analyze(songs[0].data(), songs[0].size(), 2, ref.data(), &ref_st[0]) == BLISSGPU_OK);  // warm-up
    double t0 = now_s();
    for (int i = 0; i < n_songs; i++)
        CHECK(blissgpu_analyze(songs[i].data(), songs[i].size(), 2, ref.data() + (size_t)i * d, &ref_st[i]) == BLISSGPU_OK);
    const double serial_s = now_s() - t0;
    CHECK(ref_st[5] == BLISSGPU_SONG_TOO_SHORT && std::isnan(ref[5 * d]));
    for (int i = 0; i < n_songs; i++) CHECK(i == 5 || ref_st[i] == BLISSGPU_SONG_OK);
    // the bulk entry point agrees with the single-song one bit for bit
    {
        std::vector<float> pcm;
        std::vector<uint64_t> offs, lens;
        for (auto& s : songs) { offs.push_back(pcm.size()); lens.push_back(s.size()); pcm.insert(pcm.end(), s.begin(), s.end()); }
        std::vector<float> out((size_t)n_songs * d);
        std::vector<int32_t> st(n_songs);
        CHECK(blissgpu_analyze_batch(pcm.data(), offs.data(), lens.data(), n_songs, 2, out.data(), st.data()) == BLISSGPU_OK);
        CHECK(std::memcmp(out.data(), ref.data(), out.size() * sizeof(float)) == 0 && st == ref_st);
    }
    // one-pair distances for the distance threads
    float m[23 * 23];
    CHECK(blissgpu_feature_weights(2, m) == BLISSGPU_OK);
    float ref_d[3];
    CHECK(blissgpu_distance(ref.data(), ref.data() + d, d, BLISSGPU_METRIC_EUCLIDEAN, nullptr, &ref_d[0]) == BLISSGPU_OK);
    CHECK(blissgpu_distance(ref.data(), ref.data() + d, d, BLISSGPU_METRIC_COSINE, nullptr, &ref_d[1]) == BLISSGPU_OK);
    CHECK(blissgpu_distance(ref.data(), ref.data() + d, d, BLISSGPU_METRIC_MAHALANOBIS, m, &ref_d[2]) == BLISSGPU_OK);
    CHECK(ref_d[0] > 0.0f && ref_d[2] > 0.0f);
    std::vector<float> lib;  // the valid rows as a small library
    for (int i = 0; i < n_songs; i++)
        if (i != 5) lib.insert(lib.end(), ref.begin() + (size_t)i * d, ref.begin() + (size_t)(i + 1) * d);
    const uint64_t n_lib = lib.size() / d;
    std::vector<uint32_t> ref_order(n_lib);
    CHECK(blissgpu_closest_to_songs(lib.data(), 1, lib.data(), n_lib, d, BLISSGPU_METRIC_EUCLIDEAN, nullptr, ref_order.data(), nullptr) == BLISSGPU_OK);
    CHECK(ref_order[0] == 0);

    // ---- threaded: every thread walks the songs from its own starting point ----
    std::vector<int> bad(n_threads + 2, 0);
    t0 = now_s();
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++)
        th.emplace_back([&, t]() {
            std::vector<float> row(d);
            for (int k = 0; k < calls; k++) {
                const int i = (t * 5 + k) % n_songs;
                int32_t st = -1;
                if (blissgpu_analyze(songs[i].data(), songs[i].size(), 2, row.data(), &st) != BLISSGPU_OK) { bad[t]++; continue; }
                if (st != ref_st[i] || std::memcmp(row.data(), ref.data() + (size_t)i * d, d * sizeof(float)) != 0) bad[t]++;
            }
        });
    // two more threads hammer the distance / ordering entry points of the same default context meanwhile
    th.emplace_back([&]() {
        for (int k = 0; k < 200; k++) {
            float v[3];
            const bool ok = blissgpu_distance(ref.data(), ref.data() + d, d, BLISSGPU_METRIC_EUCLIDEAN, nullptr, &v[0]) == BLISSGPU_OK &&
                            blissgpu_distance(ref.data(), ref.data() + d, d, BLISSGPU_METRIC_COSINE, nullptr, &v[1]) == BLISSGPU_OK &&
                            blissgpu_distance(ref.data(), ref.data() + d, d, BLISSGPU_METRIC_MAHALANOBIS, m, &v[2]) == BLISSGPU_OK;
            if (!ok || std::memcmp(v, ref_d, sizeof(v)) != 0) bad[n_threads]++;
        }
    });
    th.emplace_back([&]() {
        std::vector<uint32_t> order(n_lib);
        for (int k = 0; k < 50; k++) {
            if (blissgpu_closest_to_songs(lib.data(), 1, lib.data(), n_lib, d, BLISSGPU_METRIC_EUCLIDEAN, nullptr, order.data(), nullptr) != BLISSGPU_OK ||
                order != ref_order) bad[n_threads + 1]++;
        }
    });
    for (auto& x : th) x.join();
    const double threaded_s = now_s() - t0;
    for (int t = 0; t < n_threads + 2; t++) CHECK(bad[t] == 0);

    // ---- per-call cost of Song::distance ----
    const int reps = 2000;
    float v = 0.0f;
    t0 = now_s();
    for (int k = 0; k < reps; k++) CHECK(blissgpu_distance(ref.data(), ref.data() + d, d, BLISSGPU_METRIC_EUCLIDEAN, nullptr, &v) == BLISSGPU_OK);
    const double dist_ns = (now_s() - t0) / reps * 1e9;
    t0 = now_s();
    for (int k = 0; k < reps; k++) CHECK(blissgpu_distance(ref.data(), ref.data() + d, d, BLISSGPU_METRIC_MAHALANOBIS, m, &v) == BLISSGPU_OK);
    const double maha_ns = (now_s() - t0) / reps * 1e9;

    double total_samples = 0;
    for (auto& s : songs) total_samples += (double)s.size();
    std::printf("single_song_latency_ms %.3f\n", serial_s / n_songs * 1e3);
    std::printf("single_song_mean_seconds_of_audio %.2f\n", total_samples / n_songs / 22050.0);
    std::printf("threads %d\n", n_threads);
    std::printf("threaded_calls %d\n", n_threads * calls);
    std::printf("threaded_songs_per_sec %.1f\n", n_threads * calls / threaded_s);
    std::printf("serial_songs_per_sec %.1f\n", n_songs / serial_s);
    std::printf("distance_euclidean_ns %.0f\n", dist_ns);
    std::printf("distance_mahalanobis_ns %.0f\n", maha_ns);
    // the default contexts behind the single-song front (one per visible device; BLISSGPU_DEFAULT_DEVICES repeats / restricts)
    std::printf("default_devices %d\n", blissgpu_default_device_count());
    for (int k = 0; k < blissgpu_default_device_count(); k++)
        std::printf("default_device_%d_hip_ordinal %d\ndefault_device_%d_batches %llu\n", k, blissgpu_default_device(k), k,
                    (unsigned long long)blissgpu_default_device_batches(k));
    std::printf("all checks passed\n");
    return 0;
}
