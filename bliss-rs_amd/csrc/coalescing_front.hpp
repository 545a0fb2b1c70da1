// coalescing_front.hpp -- the group-commit front of the single-song entry points, device-free (tests/cpp/test_front.cpp
// drives it on the CPU with a batch runner that sleeps).
//
// The reference's bulk path is N worker threads each calling Song::analyze on its own song
// (src/song/decoder.rs:299-329); one song cannot fill a GPU, a batch can.  A caller queues its request; whoever finds a
// seat free becomes a leader and runs EVERYTHING that has queued up as one batch, the others sleep until their request
// is done.  There is one seat per default context (= per visible device), so on an 8-GPU node the worker threads keep
// all eight devices busy, each batch going to the device whose previous batch finished first; the lowest free seat is
// taken, so a lone caller always lands on the first device (whose context is warm) and pays no waiting window.
//
// One mutex, two condition variables: `arrive` wakes a leader that is gathering its batch, `done` wakes the callers whose
// requests a leader has finished and the callers waiting for a free seat.
#pragma once
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <mutex>
#include <vector>

namespace bg {

template <typename Req>  // Req needs a member `bool done` (false when submitted); everything else belongs to the runner
class CoalescingFront {
  public:
    // Blocks until r.done.  run(batch, seat) is called WITHOUT the mutex, by the leader, with every request it took;
    // it must not throw past its own handling of the batch (the front only guarantees `done` is set either way).
    template <typename Run>
    void submit(Req& r, int n_seats, Run&& run) {
        std::unique_lock<std::mutex> lk(mu_);
        if (seat_taken_.empty()) { seat_taken_.assign((size_t)n_seats, 0); seat_last_batch_.assign((size_t)n_seats, 1); }
        queue_.push_back(&r);
        cv_arrive_.notify_one();
        while (!r.done) {
            int seat = -1;
            for (int k = 0; k < n_seats && seat < 0; k++)
                if (!seat_taken_[(size_t)k]) seat = k;
            if (seat < 0) {  // every device is running a batch: the next leader will take this request along
                cv_done_.wait(lk);
                continue;
            }
            seat_taken_[(size_t)seat] = 1;
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
                seat_taken_[(size_t)seat] = 0;
                if (waited) cv_done_.notify_all();  // someone may have found no seat free meanwhile
                if (!r.done) cv_done_.wait(lk);
                continue;
            }
            seat_last_batch_[(size_t)seat] = take.size();
            lk.unlock();
            // No exception may strand the followers (their `done` flags) or keep the seat.
            try {
                run(take, seat);
            } catch (...) {
            }
            lk.lock();
            for (Req* t : take) t->done = true;
            seat_taken_[(size_t)seat] = 0;
            cv_done_.notify_all();
        }
    }

  private:
    std::mutex mu_;
    std::condition_variable cv_arrive_, cv_done_;
    std::vector<Req*> queue_;
    std::vector<char> seat_taken_;
    std::vector<size_t> seat_last_batch_;
};

}  // namespace bg
