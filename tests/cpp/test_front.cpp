// CPU test of the coalescing front (bliss-rs_amd/csrc/coalescing_front.hpp): T threads x C calls against S seats with a
// batch runner that sleeps like a device batch -- for a random time --, sometimes throws in the middle of a batch, and may
// report a seat unusable.  Checks that every request is run exactly once (or refused with the outcome the scenario
// demands), that no seat runs two batches at a time, that several seats are in use, that a retired seat draws no more
// traffic, that a request nobody picks up fails at its deadline instead of blocking -- and, by finishing at all within the
// caller's timeout, that no caller is left waiting.  Built with -fsanitize=thread by the CPU suite.
//     usage: test_front [threads] [calls] [seats] [batch_us] [scenario]
//     scenario 0 = plain, 1 = one seat unusable, 2 = every seat unusable, 3 = a stuck leader and short deadlines,
//              4 = one seat unusable for the first 150 ms only: its rest (20 ms, doubling) ends and it serves batches again
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <thread>
#include <vector>

#include "../../bliss-rs_amd/csrc/coalescing_front.hpp"

struct Req {
    int id;
    int runs = 0;
    bool done = false;
    int front_outcome = 0;
};

static unsigned lcg(unsigned& s) { return s = s * 1664525u + 1013904223u; }

int main(int argc, char** argv) {
    const int T = argc > 1 ? std::atoi(argv[1]) : 32, C = argc > 2 ? std::atoi(argv[2]) : 2000;
    const int S = argc > 3 ? std::atoi(argv[3]) : 4, batch_us = argc > 4 ? std::atoi(argv[4]) : 300;
    const int scenario = argc > 5 ? std::atoi(argv[5]) : 0;
    bg::CoalescingFront<Req> front;
    // scenarios 1 - 3 count on a retired seat STAYING retired for the length of the run
    front.set_revive_base(scenario == 4 ? std::chrono::milliseconds(20) : std::chrono::milliseconds(3600 * 1000));
    const auto t_start = std::chrono::steady_clock::now();
    std::vector<std::atomic<int>> busy(S);
    std::vector<std::atomic<long>> batches(S), refused_calls(S);
    for (int s = 0; s < S; s++) { busy[s] = 0; batches[s] = 0; refused_calls[s] = 0; }
    std::atomic<long> total{0}, bad{0}, biggest{0}, thrown{0}, no_seat{0}, timed_out{0};
    std::atomic<bool> release_stuck{false};
    std::atomic<int> stuck_batches{0};
    auto run = [&](std::vector<Req*>& take, int seat) -> bool {
        const bool early = std::chrono::steady_clock::now() - t_start < std::chrono::milliseconds(150);
        if (scenario == 2 || (scenario == 1 && seat == S / 2) || (scenario == 4 && seat == S / 2 && early)) {  // this seat's device cannot give a context
            refused_calls[seat]++;
            return false;
        }
        if (busy[seat].fetch_add(1) != 0) bad++;                      // two batches on one seat
        batches[seat]++;
        long b = (long)take.size(), prev = biggest.load();
        while (b > prev && !biggest.compare_exchange_weak(prev, b)) {}
        unsigned rs = (unsigned)take[0]->id * 2654435761u + (unsigned)seat;
        if (scenario == 3 && stuck_batches.fetch_add(1) < S) {        // the first batch of every seat hangs "in the driver"
            while (!release_stuck.load()) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        } else {
            // (a batch of one is shorter than a full one, like on the device; the duration is random: 0.25x .. 1.75x)
            const int base = batch_us / 4 + (batch_us * 3 / 4) * (int)take.size() / T;
            std::this_thread::sleep_for(std::chrono::microseconds(base / 4 + (int)(lcg(rs) % (unsigned)(base * 3 / 2 + 1))));
        }
        const bool fail_midway = scenario == 0 && lcg(rs) % 61 == 0;    // a leader that fails in the middle of its batch
        size_t k = 0;
        for (Req* t : take) {
            t->runs++;
            total++;
            if (fail_midway && ++k == (take.size() + 1) / 2) {
                for (size_t q = k; q < take.size(); q++) { take[q]->runs++; total++; }   // (the real runner fails them with a code)
                busy[seat].fetch_sub(1);
                thrown++;
                throw std::runtime_error("out of host memory while gathering the batch");
            }
        }
        busy[seat].fetch_sub(1);
        return true;
    };
    const auto deadline = scenario == 3 ? std::chrono::milliseconds(150) : std::chrono::milliseconds(60000);
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t] {
            for (int c = 0; c < C; c++) {
                Req r{t * C + c};
                const bg::FrontOutcome o = front.submit(r, S, run, deadline);
                if (!r.done) bad++;
                if (o == bg::FRONT_SERVED && r.runs != 1) bad++;
                if (o != bg::FRONT_SERVED && r.runs != 0) bad++;
                if (o == bg::FRONT_NO_SEAT) no_seat++;
                if (o == bg::FRONT_TIMED_OUT) timed_out++;
                if ((c & 7) == (t & 7)) std::this_thread::sleep_for(std::chrono::microseconds(50 + 13 * (t % 5)));  // "decode"
                if (scenario == 4) std::this_thread::sleep_for(std::chrono::milliseconds(1));  // (the run must outlast the rests)
            }
        });
    if (scenario == 3) {
        // every seat's leader hangs; the queued callers must come back with TIMED_OUT by themselves; then the leaders return
        std::this_thread::sleep_for(std::chrono::milliseconds(600));
        std::printf("while the leaders hang: %s\n", front.describe().c_str());
        release_stuck = true;
    }
    for (auto& x : th) x.join();
    int used = 0;
    for (int s = 0; s < S; s++) used += batches[s] > 0;
    const long all = (long)T * C;
    std::printf("requests %ld of %ld, bad %ld, seats used %d of %d, biggest batch %ld, failed leaders %ld, no_seat %ld, timed_out %ld, "
                "retired %d; %s\n", total.load(), all, bad.load(), used, S, biggest.load(), thrown.load(), no_seat.load(), timed_out.load(),
                front.retired_seats(), front.describe().c_str());
    bool ok = bad == 0;
    if (scenario == 0) ok = ok && total == all && (S == 1 || used > 1) && no_seat == 0 && timed_out == 0;
    if (scenario == 1)   // the unusable seat was asked exactly once, ran nothing, and everything was served elsewhere
        ok = ok && total == all && front.retired_seats() == 1 && refused_calls[S / 2] == 1 && batches[S / 2] == 0 && no_seat == 0;
    if (scenario == 2) ok = ok && total == 0 && no_seat == all && front.retired_seats() == S;
    if (scenario == 3) ok = ok && timed_out > 0 && total + timed_out == all && no_seat == 0;
    if (scenario == 4)   // retired, tried again after each rest, and back in service once its device gives a context
        ok = ok && total == all && no_seat == 0 && refused_calls[S / 2] >= 1 && refused_calls[S / 2] <= 4 && batches[S / 2] > 0 &&
             front.retired_seats() == 0;
    return ok ? 0 : 1;
}
