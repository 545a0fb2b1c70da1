// node.hip -- one host process driving every GPU of the node: songs shard by song with no data-path
// collective, ONE RCCL all-gather over xGMI collects the feature rows so that every device holds the full n x d matrix,
// and the pairwise-distance kernel is then row-block sharded.  This is the C-ABI form of what bliss_rs_amd/shard.py does
// with torch.distributed (one process per GPU); a Rust / C host has no torch, so the library talks to RCCL itself.
//
// RCCL is loaded at run time (dlopen) the first time a node is created: libblissgpu.so keeps loading on machines without
// it, and inside a torch process the copy torch already mapped is reused.  Without RCCL blissgpu_node_create fails with
// BLISSGPU_ERR_RCCL -- there is no fallback path.
//
// Loopback ranks: a device list that names one ordinal more than once (several contexts sharing a GPU) cannot have an
// RCCL communicator (ncclCommInitAll rejects duplicate devices), and needs none: every rank's buffers are reachable
// from every other rank's stream, so the gather is R x R device-to-device copies ordered by events.  Everything else --
// the plan, the padded rank-major layout with perm = -1 slots, the scatter kernel, the row-block pairwise -- is the code
// the RCCL form runs, which makes the N > 1 paths testable on a one-GPU box.
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>  // types and prototypes only: every call goes through the pointers resolved below

#include <algorithm>
#include <numeric>
#include <thread>

#include "ctx.hpp"

using namespace bg;

namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

std::mutex g_rccl_mu;
Rccl g_rccl;

int load_rccl() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.handle) return BLISSGPU_OK;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names)  // a copy that is already mapped (torch ships one) wins
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    for (size_t i = 0; !h && i < sizeof(names) / sizeof(names[0]); i++) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(BLISSGPU_ERR_RCCL, "dlopen(librccl.so)", dlerror());
    Rccl r;
    r.handle = h;
    r.CommInitAll = (decltype(r.CommInitAll))dlsym(h, "ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
    r.GroupStart = (decltype(r.GroupStart))dlsym(h, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(h, "ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.CommInitAll || !r.CommDestroy || !r.AllGather || !r.GroupStart || !r.GroupEnd || !r.GetErrorString)
        return fail(BLISSGPU_ERR_RCCL, "dlsym(librccl.so)", "missing ncclCommInitAll / ncclAllGather / ncclGroup*");
    g_rccl = r;
    return BLISSGPU_OK;
}

#define RCCL_TRY(expr)                                                                         \
    do {                                                                                       \
        ncclResult_t r_ = (expr);                                                              \
        if (r_ != ncclSuccess) return fail(BLISSGPU_ERR_RCCL, #expr, g_rccl.GetErrorString(r_)); \
    } while (0)

// rows gathered rank-major ([rank][slot][d], slot < n_max, padding slots have perm = -1) -> their global rows
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ gathered, const int32_t* __restrict__ perm,
                                                           uint32_t n_slots, uint32_t d, float* __restrict__ full) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_slots * d) return;
    const uint32_t slot = i / d, k = i - slot * d;
    const int32_t row = perm[slot];
    if (row >= 0) full[(size_t)row * d + k] = gathered[i];
}

}  // namespace

struct blissgpu_node {
    std::mutex mu;
    int n = 0;
    bool loopback = false;              // some device ordinal appears twice: gather by device-to-device copies, no RCCL
    std::vector<hipEvent_t> ev_rows;    // per rank: its local rows are in send[r] (loopback gather)
    std::vector<hipEvent_t> ev_read;    // per rank: it has copied every rank's send buffer (loopback gather)
    std::vector<int> devices;
    std::vector<blissgpu_ctx*> ctx;
    std::vector<ncclComm_t> comm;
    // per rank
    std::vector<DevBuf<float>> send, recv, full, block;
    std::vector<DevBuf<int32_t>> perm;
    std::vector<DevBuf<float>> dm;
    // last analysis
    uint32_t n_songs = 0, d = 0;
    std::vector<std::vector<uint32_t>> shard;  // global song indices of each rank, ascending
};

namespace {

// bliss_rs_amd.shard.shard_songs: greedy longest-first assignment balancing the samples per rank; ties -> fewest songs ->
// lowest rank.  Deterministic, so every host computes the same plan.
void shard_plan(const uint64_t* lengths, uint32_t n_songs, int world, std::vector<std::vector<uint32_t>>& shards) {
    // (exported device-free as blissgpu_shard_plan)
    std::vector<uint32_t> order(n_songs);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return lengths[a] > lengths[b]; });
    std::vector<uint64_t> load(world, 0), count(world, 0);
    shards.assign(world, {});
    for (uint32_t i : order) {
        int best = 0;
        for (int r = 1; r < world; r++)
            if (load[r] < load[best] || (load[r] == load[best] && count[r] < count[best])) best = r;
        shards[best].push_back(i);
        load[best] += lengths[i];
        count[best]++;
    }
    for (auto& s : shards) std::sort(s.begin(), s.end());
}

// after every rank's local rows sit in send[r] (local order): one all-gather + a scatter to global rows on every device
int gather_rows(blissgpu_node* nd) {
    const int R = nd->n;
    const uint32_t d = nd->d;
    uint32_t n_max = 1;
    for (auto& s : nd->shard) n_max = std::max<uint32_t>(n_max, (uint32_t)s.size());
    std::vector<int32_t> perm((size_t)R * n_max, -1);
    for (int r = 0; r < R; r++)
        for (size_t j = 0; j < nd->shard[r].size(); j++) perm[(size_t)r * n_max + j] = (int32_t)nd->shard[r][j];
    // every rank's context is held while its stream is being fed (the header promises per-context serialisation);
    // always in rank order, so two nodes sharing contexts could not deadlock either
    std::vector<std::unique_lock<std::recursive_mutex>> held;
    for (int r = 0; r < R; r++) held.emplace_back(nd->ctx[r]->mu);
    for (int r = 0; r < R; r++) {
        blissgpu_ctx* c = nd->ctx[r];
        HIP_TRY(hipSetDevice(c->device));
        int rc = nd->recv[r].ensure((size_t)R * n_max * d);
        if (!rc) rc = nd->full[r].ensure(std::max<size_t>(1, (size_t)nd->n_songs * d));
        if (!rc) rc = nd->perm[r].ensure(perm.size());
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(nd->perm[r].p, perm.data(), perm.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));  // `perm` is a local: the copy must have left it
    }
    const size_t block = (size_t)n_max * d;  // padded fixed-size blocks (<= ~1 MB per rank at library sizes: latency-bound on xGMI)
    if (nd->loopback) {
        for (int q = 0; q < R; q++) {
            HIP_TRY(hipSetDevice(nd->ctx[q]->device));
            HIP_TRY(hipEventRecord(nd->ev_rows[q], nd->ctx[q]->stream));
        }
        for (int r = 0; r < R; r++) {
            blissgpu_ctx* c = nd->ctx[r];
            HIP_TRY(hipSetDevice(c->device));
            for (int q = 0; q < R; q++) {
                if (q != r) HIP_TRY(hipStreamWaitEvent(c->stream, nd->ev_rows[q], 0));
                HIP_TRY(hipMemcpyAsync(nd->recv[r].p + (size_t)q * block, nd->send[q].p, block * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
            }
            HIP_TRY(hipEventRecord(nd->ev_read[r], c->stream));  // rank r has read every send buffer
        }
        // the owner of a send buffer may only overwrite it (the next call's memset and row writes) once every OTHER rank's
        // copy out of it has run: without this the next node call raced with the previous gather unless the caller
        // synchronised in between
        for (int q = 0; q < R; q++) {
            HIP_TRY(hipSetDevice(nd->ctx[q]->device));
            for (int r = 0; r < R; r++)
                if (r != q) HIP_TRY(hipStreamWaitEvent(nd->ctx[q]->stream, nd->ev_read[r], 0));
        }
    } else {
        // single-process multi-device collectives must be issued inside one group; the group is closed on every path
        ncclResult_t nr = g_rccl.GroupStart();
        if (nr != ncclSuccess) return fail(BLISSGPU_ERR_RCCL, "ncclGroupStart", g_rccl.GetErrorString(nr));
        for (int r = 0; r < R && nr == ncclSuccess; r++)
            nr = g_rccl.AllGather(nd->send[r].p, nd->recv[r].p, block, ncclFloat, nd->comm[r], nd->ctx[r]->stream);
        const ncclResult_t ne = g_rccl.GroupEnd();
        if (nr == ncclSuccess) nr = ne;
        if (nr != ncclSuccess) return fail(BLISSGPU_ERR_RCCL, "ncclAllGather", g_rccl.GetErrorString(nr));
    }
    for (int r = 0; r < R; r++) {
        blissgpu_ctx* c = nd->ctx[r];
        HIP_TRY(hipSetDevice(c->device));
        const uint32_t n_slots = (uint32_t)R * n_max;
        hipLaunchKernelGGL(scatter_rows_kernel, dim3((n_slots * d + 255) / 256), dim3(256), 0, c->stream, nd->recv[r].p,
                           nd->perm[r].p, n_slots, d, nd->full[r].p);
        HIP_TRY(hipGetLastError());
    }
    return BLISSGPU_OK;
}

int prepare(blissgpu_node* nd, const uint64_t* lengths, const uint32_t* rank_of_song, uint32_t n_songs, uint32_t version) {
    nd->d = blissgpu_feature_count(version);
    nd->n_songs = n_songs;
    if (rank_of_song) {
        nd->shard.assign(nd->n, {});
        for (uint32_t i = 0; i < n_songs; i++) {
            if (rank_of_song[i] >= (uint32_t)nd->n) return fail(BLISSGPU_ERR_INVALID, "blissgpu_node", "rank_of_song out of range");
            nd->shard[rank_of_song[i]].push_back(i);
        }
    } else {
        shard_plan(lengths, n_songs, nd->n, nd->shard);
    }
    uint32_t n_max = 1;
    for (auto& s : nd->shard) n_max = std::max<uint32_t>(n_max, (uint32_t)s.size());
    for (int r = 0; r < nd->n; r++) {
        blissgpu_ctx* c = nd->ctx[r];
        std::lock_guard<std::recursive_mutex> lk(c->mu);
        HIP_TRY(hipSetDevice(c->device));
        int rc = nd->send[r].ensure((size_t)n_max * nd->d);
        if (rc) return rc;
        HIP_TRY(hipMemsetAsync(nd->send[r].p, 0, (size_t)n_max * nd->d * sizeof(float), c->stream));
    }
    return BLISSGPU_OK;
}

}  // namespace

extern "C" {

int blissgpu_shard_plan(const uint64_t* lengths, uint32_t n_songs, uint32_t world, uint32_t* rank_of_song) {
    if (world < 1 || world > 4096 || (n_songs && (!lengths || !rank_of_song)))
        return fail(BLISSGPU_ERR_INVALID, "blissgpu_shard_plan", "bad arguments");
    std::vector<std::vector<uint32_t>> shards;
    shard_plan(lengths, n_songs, (int)world, shards);
    for (uint32_t r = 0; r < world; r++)
        for (uint32_t i : shards[r]) rank_of_song[i] = r;
    return BLISSGPU_OK;
}

int blissgpu_node_create(int n_devices, const int* devices, blissgpu_node** out) {
    if (!out || n_devices < 1 || n_devices > 64) return fail(BLISSGPU_ERR_INVALID, "blissgpu_node_create", "bad arguments");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1)
        return fail(BLISSGPU_ERR_NO_DEVICE, "blissgpu_node_create", "no HIP device");
    std::vector<int> devs;
    bool loopback = false;
    for (int r = 0; r < n_devices; r++) {
        const int dv = devices ? devices[r] : r;
        if (dv < 0 || dv >= count) return fail(BLISSGPU_ERR_NO_DEVICE, "blissgpu_node_create", "fewer HIP devices than requested");
        loopback = loopback || std::find(devs.begin(), devs.end(), dv) != devs.end();
        devs.push_back(dv);
    }
    int rc = loopback ? BLISSGPU_OK : load_rccl();
    if (rc) return rc;
    blissgpu_node* nd = new blissgpu_node();
    nd->n = n_devices;
    nd->loopback = loopback;
    nd->devices = devs;
    nd->ev_rows.assign(n_devices, nullptr);
    nd->ev_read.assign(n_devices, nullptr);
    nd->ctx.assign(n_devices, nullptr);
    nd->comm.assign(n_devices, nullptr);
    nd->send.resize(n_devices); nd->recv.resize(n_devices); nd->full.resize(n_devices); nd->block.resize(n_devices);
    nd->perm.resize(n_devices); nd->dm.resize(n_devices);
    for (int r = 0; r < n_devices && !rc; r++) rc = blissgpu_ctx_create(nd->devices[r], &nd->ctx[r]);
    for (int r = 0; r < n_devices && !rc; r++) {
        hipError_t e = hipSetDevice(nd->devices[r]);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&nd->ev_rows[r], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&nd->ev_read[r], hipEventDisableTiming);
        if (e != hipSuccess) rc = fail(BLISSGPU_ERR_HIP, "hipEventCreate", hipGetErrorString(e));
    }
    if (!rc && !loopback) {
        ncclResult_t nr = g_rccl.CommInitAll(nd->comm.data(), n_devices, nd->devices.data());
        if (nr != ncclSuccess) rc = fail(BLISSGPU_ERR_RCCL, "ncclCommInitAll", g_rccl.GetErrorString(nr));
    }
    if (rc) { blissgpu_node_destroy(nd); return rc; }
    *out = nd;
    return BLISSGPU_OK;
}

int blissgpu_node_destroy(blissgpu_node* nd) {
    if (!nd) return BLISSGPU_OK;
    for (int r = 0; r < nd->n; r++) {
        if (nd->ctx[r]) { (void)hipSetDevice(nd->ctx[r]->device); (void)hipStreamSynchronize(nd->ctx[r]->stream); }
        if (nd->comm[r]) (void)g_rccl.CommDestroy(nd->comm[r]);
        if (nd->ev_rows[r]) (void)hipEventDestroy(nd->ev_rows[r]);
        if (nd->ev_read[r]) (void)hipEventDestroy(nd->ev_read[r]);
        nd->send[r].release(); nd->recv[r].release(); nd->full[r].release(); nd->block[r].release();
        nd->perm[r].release(); nd->dm[r].release();
        if (nd->ctx[r]) blissgpu_ctx_destroy(nd->ctx[r]);
    }
    delete nd;
    return BLISSGPU_OK;
}

int blissgpu_node_device_count(blissgpu_node* nd) { return nd ? nd->n : 0; }
blissgpu_ctx* blissgpu_node_ctx(blissgpu_node* nd, int rank) { return (nd && rank >= 0 && rank < nd->n) ? nd->ctx[rank] : nullptr; }

int blissgpu_node_shard(blissgpu_node* nd, const uint64_t* lengths, uint32_t n_songs, uint32_t* rank_of_song) {
    if (!nd) return fail(BLISSGPU_ERR_INVALID, "blissgpu_node_shard", "NULL argument");
    return blissgpu_shard_plan(lengths, n_songs, (uint32_t)nd->n, rank_of_song);
}

void blissgpu_row_block(uint64_t n_rows, uint32_t world, uint32_t rank, uint64_t* lo, uint64_t* hi) {
    if (world == 0) world = 1;
    if (rank >= world) {  // no such rank: an empty block behind the matrix
        if (lo) *lo = n_rows;
        if (hi) *hi = n_rows;
        return;
    }
    const uint64_t base = n_rows / world, rem = n_rows % world, r = rank;
    const uint64_t l = r * base + std::min(r, rem);
    if (lo) *lo = l;
    if (hi) *hi = l + base + (r < rem ? 1 : 0);
}

void blissgpu_node_row_block(blissgpu_node* nd, uint64_t n_rows, int rank, uint64_t* lo, uint64_t* hi) {
    blissgpu_row_block(n_rows, nd ? (uint32_t)nd->n : 1u, rank < 0 ? 0xFFFFFFFFu : (uint32_t)rank, lo, hi);
}

int blissgpu_node_analyze_device(blissgpu_node* nd, const float* const* d_pcm, const uint64_t* offsets, const uint64_t* lengths,
                                 const uint32_t* rank_of_song, uint32_t n_songs, uint32_t features_version) {
    if (!nd || (n_songs && (!d_pcm || !offsets || !lengths || !rank_of_song)))
        return fail(BLISSGPU_ERR_INVALID, "blissgpu_node_analyze_device", "NULL argument");
    if (!blissgpu_feature_count(features_version)) return fail(BLISSGPU_ERR_INVALID, "blissgpu_node_analyze_device", "features_version must be 1 or 2");
    std::lock_guard<std::mutex> lk(nd->mu);
    int rc = prepare(nd, lengths, rank_of_song, n_songs, features_version);
    if (rc) return rc;
    for (int r = 0; r < nd->n; r++) {  // enqueue only: the devices run concurrently
        const auto& mine = nd->shard[r];
        if (mine.empty()) continue;
        std::vector<uint64_t> offs(mine.size()), lens(mine.size());
        for (size_t j = 0; j < mine.size(); j++) { offs[j] = offsets[mine[j]]; lens[j] = lengths[mine[j]]; }
        rc = blissgpu_analyze_batch_device(nd->ctx[r], d_pcm[r], offs.data(), lens.data(), (uint32_t)mine.size(), features_version,
                                           nd->send[r].p, nullptr);
        if (rc) return rc;
    }
    return gather_rows(nd);
}

int blissgpu_node_analyze(blissgpu_node* nd, const float* pcm, const uint64_t* offsets, const uint64_t* lengths, uint32_t n_songs,
                          uint32_t features_version, float* out, int32_t* status) {
    if (!nd || (n_songs && (!pcm || !offsets || !lengths || !out)))
        return fail(BLISSGPU_ERR_INVALID, "blissgpu_node_analyze", "NULL argument");
    const uint32_t d = blissgpu_feature_count(features_version);
    if (!d) return fail(BLISSGPU_ERR_INVALID, "blissgpu_node_analyze", "features_version must be 1 or 2");
    std::lock_guard<std::mutex> lk(nd->mu);
    int rc = prepare(nd, lengths, nullptr, n_songs, features_version);
    if (rc) return rc;
    // one host thread per device feeds its shard (the H2D copies of the devices overlap; PCIe links are per device)
    std::vector<int> rcs(nd->n, BLISSGPU_OK);
    std::vector<std::string> errs(nd->n);
    std::vector<std::vector<float>> rows(nd->n);
    std::vector<std::thread> th;
    for (int r = 0; r < nd->n; r++)
        th.emplace_back([&, r]() {
            const auto& mine = nd->shard[r];
            if (mine.empty()) return;
            std::vector<const void*> ptrs(mine.size());
            std::vector<uint64_t> lens(mine.size());
            for (size_t j = 0; j < mine.size(); j++) { ptrs[j] = pcm + offsets[mine[j]]; lens[j] = lengths[mine[j]]; }
            rows[r].resize(mine.size() * (size_t)d);
            // the rows go to the host (the caller's `out`) AND stay on the device in send[r] for the gather
            rcs[r] = analyze_host_songs(nd->ctx[r], ptrs.data(), lens.data(), (uint32_t)mine.size(), BLISSGPU_SAMPLE_F32, 1, features_version,
                                        rows[r].data(), nullptr, "blissgpu_node_analyze", nd->send[r].p);
            if (rcs[r]) errs[r] = blissgpu_last_error();
        });
    for (auto& t : th) t.join();
    for (int r = 0; r < nd->n; r++)
        if (rcs[r]) return fail(rcs[r], "blissgpu_node_analyze", errs[r].c_str());
    for (int r = 0; r < nd->n; r++) {
        const auto& mine = nd->shard[r];
        for (size_t j = 0; j < mine.size(); j++) memcpy(out + (size_t)mine[j] * d, rows[r].data() + j * d, d * sizeof(float));
    }
    if (status)
        for (uint32_t i = 0; i < n_songs; i++) status[i] = lengths[i] >= (uint64_t)MIN_SAMPLES ? BLISSGPU_SONG_OK : BLISSGPU_SONG_TOO_SHORT;
    return gather_rows(nd);
}

const float* blissgpu_node_features(blissgpu_node* nd, int rank) {
    return (nd && rank >= 0 && rank < nd->n) ? nd->full[rank].p : nullptr;
}

int blissgpu_node_synchronize(blissgpu_node* nd) {
    if (!nd) return fail(BLISSGPU_ERR_INVALID, "blissgpu_node_synchronize", "node is NULL");
    for (int r = 0; r < nd->n; r++) {
        int rc = blissgpu_ctx_synchronize(nd->ctx[r]);
        if (rc) return rc;
    }
    return BLISSGPU_OK;
}

int blissgpu_node_pairwise(blissgpu_node* nd, int metric, const float* M, float* out) {
    if (!nd || !out) return fail(BLISSGPU_ERR_INVALID, "blissgpu_node_pairwise", "NULL argument");
    if (metric < 0 || metric > 2 || (metric == BLISSGPU_METRIC_MAHALANOBIS && !M))
        return fail(BLISSGPU_ERR_INVALID, "blissgpu_node_pairwise", "bad metric / M");
    std::lock_guard<std::mutex> lk(nd->mu);
    const uint64_t n = nd->n_songs;
    const uint32_t d = nd->d;
    if (n == 0) return BLISSGPU_OK;
    // rank r computes the rows [lo_r, hi_r) against every column of the matrix it holds: no exchange
    std::vector<uint64_t> lo(nd->n), hi(nd->n);
    for (int r = 0; r < nd->n; r++) {
        blissgpu_node_row_block(nd, n, r, &lo[r], &hi[r]);
        if (hi[r] == lo[r]) continue;
        blissgpu_ctx* c = nd->ctx[r];
        std::lock_guard<std::recursive_mutex> lk2(c->mu);
        HIP_TRY(hipSetDevice(c->device));
        int rc = nd->block[r].ensure((hi[r] - lo[r]) * n);
        if (!rc && metric == BLISSGPU_METRIC_MAHALANOBIS) {
            rc = nd->dm[r].ensure((size_t)d * d);
            if (!rc) {
                HIP_TRY(hipMemcpyAsync(nd->dm[r].p, M, (size_t)d * d * sizeof(float), hipMemcpyHostToDevice, c->stream));
                HIP_TRY(hipStreamSynchronize(c->stream));
            }
        }
        if (rc) return rc;
        rc = blissgpu_pairwise_device(c, nd->full[r].p + lo[r] * d, hi[r] - lo[r], nd->full[r].p, n, d, metric,
                                      metric == BLISSGPU_METRIC_MAHALANOBIS ? nd->dm[r].p : nullptr, nd->block[r].p, n);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(out + lo[r] * n, nd->block[r].p, (hi[r] - lo[r]) * n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    }
    return blissgpu_node_synchronize(nd);
}

}  // extern "C"
