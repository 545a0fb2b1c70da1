// coalescing_front.hpp -- the group-commit front of the single-song entry points, device-free (tests/cpp/test_front.cpp
// drives it on the CPU with a batch runner that sleeps, fails and retires seats; built with -fsanitize=thread in the CPU
// suite).
//
// The reference's bulk path is N worker threads each calling Song::analyze on its own song
// (src/song/decoder.rs:299-329); one song cannot fill a GPU, a batch can.  A caller queues its request; whoever finds a
// seat free becomes a leader and runs EVERYTHING that has queued up as one batch, the others sleep until their request
// is done.  There is one seat per default context (= per visible device), so on an 8-GPU node the worker threads keep
// all eight devices busy, each batch going to the device whose previous batch finished first; the lowest free seat is
// taken, so a lone caller always lands on the first device (whose context is warm) and pays no waiting window.
//
// A seat whose runner reports it unusable (its device cannot give a context: busy, full, another architecture, a bad
// ordinal) is RETIRED: the batch goes back to the head of the queue for another seat, and the seat draws no more traffic.
// Only when every seat is retired do requests fail (NO_SEAT).  Retirement is not for ever: a retired seat is offered one
// more batch after a rest that doubles from revive_base (1 s) to 64 x that -- the runner decides again whether the seat is
// usable (a device that was full when the process started may have room now) -- and revive_all() ends every rest at once.
//
// No caller blocks forever on the front's own account: a request that no leader has picked up by its deadline is withdrawn
// and fails with TIMED_OUT (describe() names the seats and the queue).  A request a leader HAS picked up is waited for
// without a deadline -- the leader is reading the caller's PCM and will write the caller's row, so it cannot be abandoned;
// that wait ends when the leader's device call returns, exactly like the leader's own.
//
// One mutex, two condition variables: `arrive` wakes a leader that is gathering its batch, `done` wakes the callers whose
// requests a leader has finished and the callers waiting for a free seat.
#pragma once
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <mutex>
#include <string>
#include <vector>

namespace bg {

enum FrontOutcome : int { FRONT_SERVED = 0, FRONT_NO_SEAT = 1, FRONT_TIMED_OUT = 2 };

// Req needs `bool done` (false when submitted) and `int front_outcome` (FRONT_SERVED when submitted); everything else
// belongs to the runner.
template <typename Req>
class CoalescingFront {
  public:
    // Blocks until the request is served, withdrawn or refused, and returns which.  run(batch, seat) is called WITHOUT the
    // mutex, by the leader, with every request it took.  It returns true when it has dealt with the batch (every request
    // carries its own result, failures included) and false when the SEAT is unusable and nothing was done.  An exception
    // out of run counts as "dealt with" (the front only guarantees `done` is set either way).
    template <typename Run>
    FrontOutcome submit(Req& r, int n_seats, Run&& run, std::chrono::milliseconds deadline = std::chrono::minutes(10)) {
        std::unique_lock<std::mutex> lk(mu_);
        if (seat_.empty()) {
            seat_.assign((size_t)n_seats, SEAT_FREE);
            seat_last_batch_.assign((size_t)n_seats, 1);
            seat_retired_until_.assign((size_t)n_seats, std::chrono::steady_clock::time_point{});
            seat_retirements_.assign((size_t)n_seats, 0);
        }
        const auto t_end = std::chrono::steady_clock::now() + deadline;
        queue_.push_back(&r);
        cv_arrive_.notify_one();
        // waits for a leader's report; false = the deadline passed with the request still in the queue (now withdrawn)
        auto wait_report = [&]() -> bool {
            if (std::find(queue_.begin(), queue_.end(), &r) == queue_.end()) {  // a leader holds it: see the header
                cv_done_.wait(lk);
                return true;
            }
            if (cv_done_.wait_until(lk, t_end) == std::cv_status::timeout && !r.done) {
                auto it = std::find(queue_.begin(), queue_.end(), &r);
                if (it != queue_.end()) {
                    queue_.erase(it);
                    r.done = true;
                    r.front_outcome = FRONT_TIMED_OUT;
                    return false;
                }
            }
            return true;
        };
        while (!r.done) {
            int seat = -1, alive = 0;
            {
                const auto now = std::chrono::steady_clock::now();
                for (int k = 0; k < n_seats; k++)  // a retired seat whose rest is over gets one more try
                    if (seat_[(size_t)k] == SEAT_RETIRED && now >= seat_retired_until_[(size_t)k]) seat_[(size_t)k] = SEAT_FREE;
            }
            for (int k = 0; k < n_seats; k++) {
                if (seat_[(size_t)k] != SEAT_RETIRED) alive++;
                if (seat < 0 && seat_[(size_t)k] == SEAT_FREE) seat = k;
            }
            if (alive == 0) {  // (set by the leader that retired the last seat; a request arriving later lands here)
                auto it = std::find(queue_.begin(), queue_.end(), &r);
                if (it != queue_.end()) queue_.erase(it);
                r.done = true;
                r.front_outcome = FRONT_NO_SEAT;
                break;
            }
            if (seat < 0) {  // every usable device is running a batch: the next leader will take this request along
                if (!wait_report()) break;
                continue;
            }
            seat_[(size_t)seat] = SEAT_TAKEN;
            // The callers the previous batch released are on their way back with their next song: when that batch showed
            // there is company, give them a moment (at most 200 us against a batch of milliseconds) instead of running a
            // batch of one.  A lone caller never waits.
            const bool waited = seat_last_batch_[(size_t)seat] > 1;  // the mutex is released while this thread holds the seat
            if (waited)
                cv_arrive_.wait_for(lk, std::chrono::microseconds(200),
                                    [&] { return queue_.size() >= seat_last_batch_[(size_t)seat]; });
            std::vector<Req*> take;
            take.swap(queue_);
            if (take.empty()) {
                // Another leader took everything, this caller's request included, while this thread was waiting (for a seat
                // or for company).  Give the seat back and SLEEP until a leader reports: going round again at once would
                // spin with the mutex held -- there is no wait in the loop when the seat's last batch was a single song --
                // and the leader that holds this request could never lock the mutex to mark it done.  (With one seat the
                // only leader always finds its own request in the queue.)
                seat_[(size_t)seat] = SEAT_FREE;
                if (waited) cv_done_.notify_all();  // someone may have found no seat free meanwhile
                if (!r.done) cv_done_.wait(lk);     // (a leader holds the request: no deadline, see the header)
                continue;
            }
            seat_last_batch_[(size_t)seat] = take.size();
            lk.unlock();
            // No exception may strand the followers (their `done` flags) or keep the seat.
            bool dealt_with = true;
            try {
                dealt_with = run(take, seat);
            } catch (...) {
            }
            lk.lock();
            if (dealt_with) {
                for (Req* t : take) t->done = true;
                seat_[(size_t)seat] = SEAT_FREE;
                seat_retirements_[(size_t)seat] = 0;
            } else {
                seat_[(size_t)seat] = SEAT_RETIRED;
                seat_last_batch_[(size_t)seat] = 1;
                seat_retired_until_[(size_t)seat] =
                    std::chrono::steady_clock::now() + revive_base_ * (1 << std::min(seat_retirements_[(size_t)seat], 6));
                seat_retirements_[(size_t)seat]++;
                bool any = false;
                for (char s : seat_) any = any || s != SEAT_RETIRED;
                if (any) {
                    // the batch goes back to the head of the queue, this thread's own request with it; the next free seat
                    // (maybe this very thread, going round the loop) takes it
                    queue_.insert(queue_.begin(), take.begin(), take.end());
                    cv_arrive_.notify_all();
                } else {
                    for (Req* t : take) { t->done = true; t->front_outcome = FRONT_NO_SEAT; }
                    for (Req* t : queue_) { t->done = true; t->front_outcome = FRONT_NO_SEAT; }
                    queue_.clear();
                }
            }
            cv_done_.notify_all();
        }
        return (FrontOutcome)r.front_outcome;
    }

    // "seats [busy, free, retired], 3 requests queued" -- for the error text of a request that was not served
    std::string describe() {
        std::lock_guard<std::mutex> lk(mu_);
        std::string s = "seats [";
        for (size_t k = 0; k < seat_.size(); k++)
            s += std::string(k ? ", " : "") + (seat_[k] == SEAT_FREE ? "free" : seat_[k] == SEAT_TAKEN ? "busy" : "retired");
        return s + "], " + std::to_string(queue_.size()) + " request(s) queued";
    }

    // every retired seat is offered traffic again (the runner decides anew whether it is usable)
    void revive_all() {
        std::lock_guard<std::mutex> lk(mu_);
        for (size_t k = 0; k < seat_.size(); k++)
            if (seat_[k] == SEAT_RETIRED) { seat_[k] = SEAT_FREE; seat_retirements_[k] = 0; }
        cv_done_.notify_all();
    }
    // the first rest of a retired seat (doubles with every further retirement, up to 64 x)
    void set_revive_base(std::chrono::milliseconds base) {
        std::lock_guard<std::mutex> lk(mu_);
        revive_base_ = base;
    }

    int retired_seats() {
        std::lock_guard<std::mutex> lk(mu_);
        int n = 0;
        for (char s : seat_) n += s == SEAT_RETIRED;
        return n;
    }

  private:
    enum : char { SEAT_FREE = 0, SEAT_TAKEN = 1, SEAT_RETIRED = 2 };
    std::mutex mu_;
    std::condition_variable cv_arrive_, cv_done_;
    std::vector<Req*> queue_;
    std::vector<char> seat_;
    std::vector<size_t> seat_last_batch_;
    std::vector<std::chrono::steady_clock::time_point> seat_retired_until_;
    std::vector<int> seat_retirements_;  // consecutive retirements of the seat
    std::chrono::milliseconds revive_base_{1000};
};

}  // namespace bg
