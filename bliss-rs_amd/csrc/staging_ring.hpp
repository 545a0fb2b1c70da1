// staging_ring.hpp -- the pinned staging ring of the host PCM feed, device-free (tests/cpp/test_staging.cpp drives it on the
// CPU with a device whose "DMA engine" is a thread; built with -fsanitize=thread in the CPU suite).
//
// What the reference's callers hand over is ordinary heap memory: a Rust Vec<f32> out of PreAnalyzedSong
// (src/song/decoder.rs:34-65, 85-101), the decoder's frame buffers (src/song/decoder/ffmpeg.rs:36-109).  A host -> device
// copy from pageable memory is staged by the HIP runtime on the calling thread, one bounce buffer at a time: 19 GB/s where
// the link moves 55 from page-locked memory.  The ring does that staging in the library, in parallel and ahead of the link:
//
//   * `lanes` worker threads, each with its own device copy queue (a HIP stream) and `slabs_per_lane` page-locked slabs;
//   * a TRANSFER is an ordered list of pieces (pageable source, device destination, <= one slab of bytes); the workers
//     claim pieces in order, memcpy the piece into a free slab and queue the slab's H2D copy on their lane, so that while
//     slab k is on the link the same worker fills slab k + 1 and the other lanes' slabs are in flight beside it;
//   * a slab is reused only after the event recorded behind its copy has completed (waited for on the host);
//   * transfers are served in the order they were posted; begin(lane) runs on every lane before its first piece of a
//     transfer (make the lane wait for the device buffer to be free), end(lane) after its last one (record the event the
//     consumer's stream waits for).  wait_enqueued() returns once every lane has run end(): every copy of the transfer is
//     queued, in order, on the lanes.
//
// Dev (the device side, a template parameter so that the CPU test can stand in for HIP):
//   void  thread_begin(int lane);                       a worker starts (hipSetDevice)
//   void* slab_alloc(size_t bytes);  void slab_free(void*);
//   void* event_create();            void event_destroy(void*);
//   int   copy_async(void* dst, const void* slab, size_t bytes, int lane);   0 = queued
//   int   event_record(void* ev, int lane);
//   int   event_wait(void* ev);                         host waits for the event
//   std::string error_string(int code);
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace bg {

struct StagePiece {
    const void* src;  // pageable host memory
    void* dst;        // device memory
    size_t bytes;     // <= slab_bytes
};

struct StageConfig {
    int lanes = 0;           // worker threads = copy queues
    int slabs_per_lane = 0;  // page-locked slabs each worker rotates through
    size_t slab_bytes = 0;
    bool operator==(const StageConfig& o) const { return lanes == o.lanes && slabs_per_lane == o.slabs_per_lane && slab_bytes == o.slab_bytes; }
};

template <class Dev>
class StagingRing {
  public:
    using LaneFn = std::function<int(int)>;  // (lane) -> 0 or a Dev error code

    explicit StagingRing(Dev dev) : dev_(std::move(dev)) {}
    StagingRing(const StagingRing&) = delete;
    StagingRing& operator=(const StagingRing&) = delete;
    ~StagingRing() { stop(); }

    bool running() const { return !workers_.empty(); }
    const StageConfig& config() const { return cfg_; }
    Dev& dev() { return dev_; }

    // Starts the workers; each allocates its own slabs and events once it has settled (thread_begin may have moved it next to
    // the device: page-locked memory is placed where the allocating thread runs).  false: an allocation failed (*err says
    // which); nothing is left behind.
    bool start(const StageConfig& cfg, std::string* err) {
        stop();
        if (cfg.lanes < 1 || cfg.slabs_per_lane < 1 || cfg.slab_bytes < 4096) { if (err) *err = "bad staging configuration"; return false; }
        cfg_ = cfg;
        lanes_.assign((size_t)cfg.lanes, Lane{});
        for (Lane& l : lanes_) {
            l.next_transfer = next_id_;
            l.slab.assign((size_t)cfg.slabs_per_lane, Slab{});
        }
        stop_ = false;
        ready_ = 0;
        start_error_.clear();
        pieces_staged_ = 0;
        bytes_staged_ = 0;
        for (int w = 0; w < cfg.lanes; w++) workers_.emplace_back([this, w] { work(w); });
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_done_.wait(lk, [&] { return ready_ == cfg_.lanes; });
        }
        if (!start_error_.empty()) {
            if (err) *err = start_error_;
            stop();
            return false;
        }
        return true;
    }

    // Ends the workers (after the transfers already posted) and frees the slabs.
    void stop() {
        if (workers_.empty()) { free_slabs(); return; }
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_work_.notify_all();
        for (std::thread& t : workers_) t.join();
        workers_.clear();
        // nothing is in flight from the ring's point of view once the slabs' events have completed
        for (Lane& l : lanes_)
            for (Slab& s : l.slab)
                if (s.busy) { (void)dev_.event_wait(s.ev); s.busy = false; }
        free_slabs();
        queue_.clear();
        first_id_ = next_id_;
    }

    // Queues a transfer; returns its ticket.  `pieces` are served in order; every piece must be <= slab_bytes.
    uint64_t post(std::vector<StagePiece>&& pieces, LaneFn begin, LaneFn end) {
        auto t = std::make_shared<Transfer>();
        t->pieces = std::move(pieces);
        t->begin = std::move(begin);
        t->end = std::move(end);
        uint64_t id;
        {
            std::lock_guard<std::mutex> lk(mu_);
            id = next_id_++;
            queue_.push_back(t);
        }
        cv_work_.notify_all();
        return id;
    }

    // Blocks until every lane has finished transfer `ticket` (all of its copies are queued on the lanes and end() has run on
    // each).  Returns 0 or the first Dev error any lane met in it (*err names the call).  Transfers must be waited for in the
    // order they were posted; a waited-for transfer is forgotten.
    int wait_enqueued(uint64_t ticket, std::string* err) {
        std::unique_lock<std::mutex> lk(mu_);
        if (ticket < first_id_ || ticket >= next_id_) return 0;  // already waited for (or never posted)
        std::shared_ptr<Transfer> t = queue_[(size_t)(ticket - first_id_)];
        cv_done_.wait(lk, [&] { return t->lanes_done == (int)lanes_.size(); });
        while (!queue_.empty() && queue_.front()->lanes_done == (int)lanes_.size() && first_id_ <= ticket) {
            queue_.pop_front();
            first_id_++;
        }
        if (t->error && err) *err = t->error_what + ": " + dev_.error_string(t->error);
        return t->error;
    }

    // Waits for every transfer posted so far and forgets them (the error paths: the workers read the caller's memory, so no
    // call may return while a transfer of its own is still being served).  Returns the first error met, or 0.
    int drain(std::string* err) {
        std::unique_lock<std::mutex> lk(mu_);
        int rc = 0;
        while (!queue_.empty()) {
            std::shared_ptr<Transfer> t = queue_.front();
            cv_done_.wait(lk, [&] { return t->lanes_done == (int)lanes_.size(); });
            if (t->error && !rc) {
                rc = t->error;
                if (err) *err = t->error_what + ": " + dev_.error_string(t->error);
            }
            queue_.pop_front();
            first_id_++;
        }
        return rc;
    }

    // pieces staged / bytes copied since start() (statistics for the tests and the bench)
    uint64_t pieces_staged() const { return pieces_staged_.load(); }
    uint64_t bytes_staged() const { return bytes_staged_.load(); }

  private:
    struct Slab { void* p = nullptr; void* ev = nullptr; bool busy = false; };
    struct Lane {
        std::vector<Slab> slab;
        size_t turn = 0;
        uint64_t next_transfer = 0;  // id of the next transfer this lane serves
    };
    struct Transfer {
        std::vector<StagePiece> pieces;
        LaneFn begin, end;
        std::atomic<size_t> next{0};
        int lanes_done = 0;  // guarded by mu_
        int error = 0;       // first error, guarded by mu_
        std::string error_what;
        std::atomic<bool> failed{false};
    };

    void free_slabs() {
        for (Lane& l : lanes_)
            for (Slab& s : l.slab) {
                if (s.ev) dev_.event_destroy(s.ev);
                if (s.p) dev_.slab_free(s.p);
                s = Slab{};
            }
        lanes_.clear();
    }

    void note_error(Transfer& t, int code, const char* what) {
        std::lock_guard<std::mutex> lk(mu_);
        if (!t.error) { t.error = code; t.error_what = what; }
        t.failed.store(true);
    }

    void work(int lane) {
        dev_.thread_begin(lane);
        Lane& me = lanes_[(size_t)lane];
        const char* bad = nullptr;
        for (Slab& s : me.slab) {
            s.p = dev_.slab_alloc(cfg_.slab_bytes);
            s.ev = s.p ? dev_.event_create() : nullptr;
            if (!s.p || !s.ev) { bad = s.p ? "cannot create a staging event" : "cannot allocate a page-locked staging slab"; break; }
        }
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (bad && start_error_.empty()) start_error_ = bad;
            ready_++;
        }
        cv_done_.notify_all();
        for (;;) {
            std::shared_ptr<Transfer> t;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_work_.wait(lk, [&] { return me.next_transfer < next_id_ || stop_; });
                if (me.next_transfer >= next_id_) return;  // stop_, and nothing left to serve
                t = queue_[(size_t)(me.next_transfer - first_id_)];
            }
            int rc = t->begin ? t->begin(lane) : 0;
            if (rc) note_error(*t, rc, "staging: begin of a transfer");
            for (;;) {
                const size_t i = t->next.fetch_add(1);
                if (i >= t->pieces.size()) break;
                if (t->failed.load()) continue;  // drain the indices, copy nothing more
                const StagePiece& pc = t->pieces[i];
                Slab& s = me.slab[me.turn++ % me.slab.size()];
                if (s.busy) {
                    rc = dev_.event_wait(s.ev);
                    s.busy = false;
                    if (rc) { note_error(*t, rc, "staging: waiting for a slab's copy"); continue; }
                }
                memcpy(s.p, pc.src, pc.bytes);
                rc = dev_.copy_async(pc.dst, s.p, pc.bytes, lane);
                if (rc) { note_error(*t, rc, "staging: host -> device copy of a slab"); continue; }
                rc = dev_.event_record(s.ev, lane);
                if (rc) { note_error(*t, rc, "staging: recording a slab's event"); continue; }
                s.busy = true;
                pieces_staged_.fetch_add(1, std::memory_order_relaxed);
                bytes_staged_.fetch_add(pc.bytes, std::memory_order_relaxed);
            }
            rc = t->end ? t->end(lane) : 0;
            if (rc) note_error(*t, rc, "staging: end of a transfer");
            {
                std::lock_guard<std::mutex> lk(mu_);
                t->lanes_done++;
                me.next_transfer++;
            }
            cv_done_.notify_all();
        }
    }

    Dev dev_;
    StageConfig cfg_{};
    std::vector<Lane> lanes_;
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_work_, cv_done_;
    std::deque<std::shared_ptr<Transfer>> queue_;  // transfers first_id_ .. next_id_ - 1 (waited-for ones popped from the front)
    uint64_t first_id_ = 0, next_id_ = 0;
    bool stop_ = false;
    int ready_ = 0;            // workers that have allocated their slabs (guarded by mu_)
    std::string start_error_;  // first allocation failure of this start (guarded by mu_)
    std::atomic<uint64_t> pieces_staged_{0}, bytes_staged_{0};
};

}  // namespace bg
