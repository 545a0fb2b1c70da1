// CPU test of the pinned staging ring (bliss-rs_amd/csrc/staging_ring.hpp) against a stand-in device whose copy queues are
// threads: a lane's copies and event records are executed in order, LATER and after a random delay, reading the slab at that
// moment -- so a slab refilled before its copy has completed corrupts the destination and the test sees it.
// Checks: every byte of every transfer arrives (random piece lists, several transfers posted ahead); begin / end run once per
// lane and transfer, begin before the lane's first piece, end after its last; an injected copy error comes back from
// wait_enqueued() of THAT transfer and the next one is clean; drain() on the error path; restart with another shape; a failed
// slab allocation fails start() cleanly; no
// copy of a transfer is still unqueued when wait_enqueued() returns.  Built with -fsanitize=thread by the CPU suite.
//     usage: test_staging [transfers] [lanes] [slabs_per_lane] [slab_kib] [seed]
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "../../bliss-rs_amd/csrc/staging_ring.hpp"

static unsigned lcg(unsigned& s) { return s = s * 1664525u + 1013904223u; }

struct FakeEvent {
    std::mutex mu;
    std::condition_variable cv;
    uint64_t recorded = 0, completed = 0;
};

struct FakeQueue {  // one lane's device copy queue
    struct Op { void* dst; const void* src; size_t bytes; FakeEvent* ev; uint64_t seq; };
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Op> ops;
    bool stop = false;
    std::thread engine;
    std::atomic<long> queued{0}, executed{0};
};

struct FakeShared {
    std::vector<FakeQueue> q;
    std::atomic<long> slabs_live{0}, events_live{0}, copies{0}, fail_at{-1}, allocs{0}, fail_alloc_at{-1};
    std::atomic<int> threads_begun{0};
    explicit FakeShared(int lanes) : q((size_t)lanes) {
        for (int l = 0; l < lanes; l++)
            q[(size_t)l].engine = std::thread([this, l] {
                FakeQueue& me = q[(size_t)l];
                unsigned rs = 977u * (unsigned)(l + 1);
                for (;;) {
                    FakeQueue::Op op;
                    {
                        std::unique_lock<std::mutex> lk(me.mu);
                        me.cv.wait(lk, [&] { return !me.ops.empty() || me.stop; });
                        if (me.ops.empty()) return;
                        op = me.ops.front();
                        me.ops.pop_front();
                    }
                    if (lcg(rs) % 4 == 0) std::this_thread::sleep_for(std::chrono::microseconds(lcg(rs) % 200));
                    if (op.dst) memcpy(op.dst, op.src, op.bytes);  // the "DMA": reads the slab NOW
                    if (op.ev) {
                        std::lock_guard<std::mutex> lk(op.ev->mu);
                        op.ev->completed = op.seq;
                        op.ev->cv.notify_all();
                    }
                    me.executed++;
                }
            });
    }
    ~FakeShared() {
        for (FakeQueue& me : q) {
            { std::lock_guard<std::mutex> lk(me.mu); me.stop = true; }
            me.cv.notify_all();
            me.engine.join();
        }
    }
    void idle() {  // every queued op has been executed
        for (FakeQueue& me : q)
            while (me.executed.load() < me.queued.load()) std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
};

struct FakeDev {
    FakeShared* sh;
    void thread_begin(int) { sh->threads_begun++; }
    void* slab_alloc(size_t bytes) {
        if (sh->allocs.fetch_add(1) == sh->fail_alloc_at.load()) return nullptr;  // injected: no page-locked memory left
        sh->slabs_live++;
        return malloc(bytes);
    }
    void slab_free(void* p) { sh->slabs_live--; free(p); }
    void* event_create() { sh->events_live++; return new FakeEvent; }
    void event_destroy(void* e) { sh->events_live--; delete (FakeEvent*)e; }
    int push(int lane, FakeQueue::Op op) {
        FakeQueue& me = sh->q[(size_t)lane];
        { std::lock_guard<std::mutex> lk(me.mu); me.ops.push_back(op); me.queued++; }
        me.cv.notify_one();
        return 0;
    }
    int copy_async(void* dst, const void* slab, size_t bytes, int lane) {
        if (sh->copies.fetch_add(1) == sh->fail_at.load()) return 700;  // injected
        return push(lane, {dst, slab, bytes, nullptr, 0});
    }
    int event_record(void* ev, int lane) {
        FakeEvent* e = (FakeEvent*)ev;
        uint64_t seq;
        { std::lock_guard<std::mutex> lk(e->mu); seq = ++e->recorded; }
        return push(lane, {nullptr, nullptr, 0, e, seq});
    }
    int event_wait(void* ev) {
        FakeEvent* e = (FakeEvent*)ev;
        std::unique_lock<std::mutex> lk(e->mu);
        const uint64_t want = e->recorded;
        e->cv.wait(lk, [&] { return e->completed >= want; });
        return 0;
    }
    std::string error_string(int code) { return "fake device error " + std::to_string(code); }
};

#define REQUIRE(cond)                                                          \
    do {                                                                       \
        if (!(cond)) { printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
    } while (0)

int main(int argc, char** argv) {
    const int transfers = argc > 1 ? std::atoi(argv[1]) : 60, lanes = argc > 2 ? std::atoi(argv[2]) : 4;
    const int spl = argc > 3 ? std::atoi(argv[3]) : 3;
    size_t slab = (size_t)(argc > 4 ? std::atoi(argv[4]) : 16) << 10;
    unsigned rs = argc > 5 ? (unsigned)std::atoi(argv[5]) : 1u;
    FakeShared sh(lanes);
    {
        bg::StagingRing<FakeDev> ring(FakeDev{&sh});
        std::string err;
        REQUIRE(!ring.start({0, spl, slab}, &err) && !ring.running());
        REQUIRE(ring.start({lanes, spl, slab}, &err));
        REQUIRE(sh.slabs_live.load() == (long)lanes * spl && sh.events_live.load() == (long)lanes * spl);

        struct Posted { uint64_t ticket; std::vector<uint8_t> src, dst; std::vector<std::atomic<int>> begun, ended; std::atomic<int> order_bad{0}; size_t n_pieces; };
        std::deque<std::unique_ptr<Posted>> inflight;
        long checked = 0;
        auto post_one = [&](size_t total) {
            auto p = std::make_unique<Posted>();
            p->src.resize(total);
            p->dst.assign(total, 0xEE);
            for (size_t i = 0; i < total; i++) p->src[i] = (uint8_t)(lcg(rs) >> 24);
            p->begun = std::vector<std::atomic<int>>((size_t)lanes);
            p->ended = std::vector<std::atomic<int>>((size_t)lanes);
            std::vector<bg::StagePiece> pieces;
            for (size_t off = 0; off < total;) {  // ragged pieces: "songs" cut at the slab size
                size_t len = 1 + lcg(rs) % slab;
                if (lcg(rs) % 3 == 0) len = slab;
                len = std::min(len, total - off);
                pieces.push_back({p->src.data() + off, p->dst.data() + off, len});
                off += len;
            }
            p->n_pieces = pieces.size();
            Posted* raw = p.get();
            p->ticket = ring.post(std::move(pieces),
                                  [raw](int lane) { if (raw->ended[(size_t)lane].load()) raw->order_bad++; raw->begun[(size_t)lane]++; return 0; },
                                  [raw](int lane) { if (raw->begun[(size_t)lane].load() != 1) raw->order_bad++; raw->ended[(size_t)lane]++; return 0; });
            inflight.push_back(std::move(p));
        };
        auto finish_one = [&](int expect_rc) -> int {
            std::unique_ptr<Posted> p = std::move(inflight.front());
            inflight.pop_front();
            std::string e;
            const int rc = ring.wait_enqueued(p->ticket, &e);
            if (rc != expect_rc) { printf("wait_enqueued: rc %d (%s), expected %d\n", rc, e.c_str(), expect_rc); return 1; }
            for (int l = 0; l < ring.config().lanes; l++)
                if (p->begun[(size_t)l].load() != 1 || p->ended[(size_t)l].load() != 1) { printf("begin/end not once per lane\n"); return 1; }
            if (p->order_bad.load()) { printf("begin/end out of order\n"); return 1; }
            sh.idle();  // (the product waits on the events end() recorded; the stand-in waits for the queues)
            if (!rc && memcmp(p->src.data(), p->dst.data(), p->src.size()) != 0) { printf("transfer %llu: bytes differ\n", (unsigned long long)p->ticket); return 1; }
            if (rc && !e.size()) { printf("an error without a message\n"); return 1; }
            checked++;
            return 0;
        };
        // 1. a stream of transfers, up to three posted ahead of the one being waited for
        for (int t = 0; t < transfers; t++) {
            post_one(1 + lcg(rs) % (slab * (size_t)lanes * (size_t)spl * 3));
            if (inflight.size() > (size_t)(lcg(rs) % 4)) REQUIRE(finish_one(0) == 0);
        }
        while (!inflight.empty()) REQUIRE(finish_one(0) == 0);
        REQUIRE(ring.pieces_staged() > 0 && ring.bytes_staged() > 0);
        // 2. an empty transfer still runs begin / end on every lane
        post_one(0);
        REQUIRE(finish_one(0) == 0);
        // 3. an injected copy failure: reported by that transfer, the next one is clean
        sh.fail_at = sh.copies.load() + 2;
        post_one(slab * 8);
        post_one(slab * 5);
        REQUIRE(finish_one(700) == 0);
        sh.fail_at = -1;
        REQUIRE(finish_one(0) == 0);
        // 4. drain() on an error path: two transfers posted, nobody waits for them one by one
        sh.fail_at = sh.copies.load() + 1;
        post_one(slab * 6);
        post_one(slab * 6);
        REQUIRE(ring.drain(&err) == 700 && !err.empty());
        sh.fail_at = -1;
        sh.idle();  // (the product synchronises the lanes' streams before it lets go of the buffers)
        inflight.clear();
        post_one(slab * 4);
        REQUIRE(finish_one(0) == 0);
        // 5. another shape: the ring restarts, old slabs are gone, tickets keep counting
        sh.idle();
        REQUIRE(ring.start({lanes > 1 ? lanes - 1 : 1, spl + 1, slab / 2 < 4096 ? 4096 : slab / 2}, &err));
        REQUIRE(sh.slabs_live.load() == (long)ring.config().lanes * ring.config().slabs_per_lane);
        slab = ring.config().slab_bytes;
        for (int t = 0; t < 6; t++) post_one(1 + lcg(rs) % (slab * 20));
        while (!inflight.empty()) REQUIRE(finish_one(0) == 0);
        // 6. a slab that cannot be allocated: start() fails, says why, leaves nothing behind, and a later start works
        sh.idle();
        sh.fail_alloc_at = sh.allocs.load() + 1;
        err.clear();
        REQUIRE(!ring.start({lanes, spl + 1, slab}, &err) && !ring.running() && !err.empty());
        REQUIRE(sh.slabs_live.load() == 0 && sh.events_live.load() == 0);
        sh.fail_alloc_at = -1;
        REQUIRE(ring.start({lanes, spl, slab}, &err));
        post_one(slab * 7 + 3);
        REQUIRE(finish_one(0) == 0);
    }
    // (scenario 5's ring has fewer lanes than the stand-in device has queues: harmless)
    if (sh.slabs_live.load() != 0 || sh.events_live.load() != 0) { printf("leaked %ld slabs, %ld events\n", sh.slabs_live.load(), sh.events_live.load()); return 1; }
    printf("staging ring ok: %d transfers, %d lanes x %d slabs of %zu KiB, %ld copies\n", transfers, lanes, spl, slab >> 10, sh.copies.load());
    return 0;
}
