// CPU test of the coalescing front (bliss-rs_amd/csrc/coalescing_front.hpp): T threads x C calls against S seats with a
// batch runner that sleeps like a device batch.  Checks that every request is run exactly once, that no seat runs two
// batches at a time, that several seats are in use, and -- by finishing at all within the caller's timeout -- that no
// caller is left waiting.    usage: test_front [threads] [calls] [seats] [batch_us]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../../bliss-rs_amd/csrc/coalescing_front.hpp"

struct Req {
    int id;
    int runs = 0;
    bool done = false;
};

int main(int argc, char** argv) {
    const int T = argc > 1 ? std::atoi(argv[1]) : 32, C = argc > 2 ? std::atoi(argv[2]) : 2000;
    const int S = argc > 3 ? std::atoi(argv[3]) : 4, batch_us = argc > 4 ? std::atoi(argv[4]) : 300;
    bg::CoalescingFront<Req> front;
    std::vector<std::atomic<int>> busy(S);
    std::vector<std::atomic<long>> batches(S);
    for (int s = 0; s < S; s++) { busy[s] = 0; batches[s] = 0; }
    std::atomic<long> total{0}, bad{0}, biggest{0};
    auto run = [&](std::vector<Req*>& take, int seat) {
        if (busy[seat].fetch_add(1) != 0) bad++;                      // two batches on one seat
        batches[seat]++;
        long b = (long)take.size(), prev = biggest.load();
        while (b > prev && !biggest.compare_exchange_weak(prev, b)) {}
        // (a batch of one is shorter than a full one, like on the device)
        std::this_thread::sleep_for(std::chrono::microseconds(batch_us / 4 + (batch_us * 3 / 4) * (int)take.size() / T));
        for (Req* t : take) { t->runs++; total++; }
        busy[seat].fetch_sub(1);
    };
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t] {
            for (int c = 0; c < C; c++) {
                Req r{t * C + c};
                front.submit(r, S, run);
                if (!r.done || r.runs != 1) bad++;
                if ((c & 7) == (t & 7)) std::this_thread::sleep_for(std::chrono::microseconds(50 + 13 * (t % 5)));  // "decode"
            }
        });
    for (auto& x : th) x.join();
    int used = 0;
    for (int s = 0; s < S; s++) used += batches[s] > 0;
    std::printf("requests %ld of %ld, bad %ld, seats used %d of %d, biggest batch %ld\n", total.load(), (long)T * C, bad.load(), used, S,
                biggest.load());
    return (total == (long)T * C && bad == 0 && (S == 1 || used > 1)) ? 0 : 1;
}
