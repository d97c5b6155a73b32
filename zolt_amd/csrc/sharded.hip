// sharded.hip — several GPUs driven by ONE host process, behind the same C ABI (include/zolt_gpu.h, "several GPUs").
//
// The reference's parallel MSM is threads inside the single `zolt prove` process (ParallelMSM.compute,
// /root/reference/src/msm/mod.zig:588-653; ParallelBatchMSM, :683-748): contiguous chunks of ceil(n / T) points, one
// Jacobian partial per worker, a serial combine and one toAffine. This file is that shape over GPUs: one shard of the
// bases resident per device, one host worker thread per shard issuing that device's launch set, the partials exchanged
// with ONE ncclAllGather over xGMI (RCCL has no user-defined reduction, so group elements are gathered, not reduced) and
// combined on the first device. k scalar vectors (HyperKZG.batchCommit, src/poly/commitment/mod.zig:558-570) travel as
// k partials per device in the same single gather. The affine result is the canonical representative of a unique group
// element, so it equals the one-GPU and the reference CPU bytes.
//
// RCCL is loaded at run time (dlopen) the first time more than one device is bound: the library has no link-time
// dependency on it, and in a process that already holds an RCCL (PyTorch's) that one is reused.
//
// Sumcheck tables shard the same way: LOW_PAIR tables by contiguous chunks, HIGH_HALF tables by residue
// class, so every fold of the first v - log2(S) rounds is local; a round's exchange is 64 bytes per device, read by the
// host from each session's pinned mailbox (the single process IS the meeting point — no collective is needed), and the
// last log2(S) rounds run on the gathered residuals on the first device.
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "common.hip.h"

namespace zg {

// ------------------------------------------------------------------ RCCL, loaded on demand
struct RcclApi {
    void *handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
static RcclApi g_rccl;
static std::mutex g_comm_mu;

static int rccl_load() {
    if (g_rccl.handle) return ZG_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);  // the host process (PyTorch) may hold one already
    for (size_t i = 0; !h && i < sizeof(names) / sizeof(names[0]); i++) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        set_error(std::string("RCCL not found (dlopen librccl.so.1): ") + dlerror());
        return ZG_ERR_HIP;
    }
    RcclApi a;
    a.handle = h;
    a.CommInitAll = reinterpret_cast<decltype(a.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(h, "ncclAllGather"));
    a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(dlsym(h, "ncclGroupStart"));
    a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!a.CommInitAll || !a.CommDestroy || !a.AllGather || !a.GroupStart || !a.GroupEnd || !a.GetErrorString) {
        set_error("RCCL library lacks a required symbol");
        return ZG_ERR_HIP;
    }
    g_rccl = a;
    return ZG_OK;
}

#define ZG_NCCL(expr)                                                                      \
    do {                                                                                   \
        ncclResult_t _r = (expr);                                                          \
        if (_r != ncclSuccess) {                                                           \
            zg::set_error(std::string(#expr) + ": " + zg::g_rccl.GetErrorString(_r));      \
            return ZG_ERR_HIP;                                                             \
        }                                                                                  \
    } while (0)

// One communicator per device over devices 0..ndev-1 (ncclCommInitAll order: rank i = device i). A set belongs to the sharded
// handles that were created while that many devices were bound and lives until the last of them is freed. When the bound set
// widens (zg_init_devices(n) after zg_init, or after a handle over fewer devices) the next handle gets a NEW set for the wider
// range; zg_shutdown only drops the library's own reference, so a handle that outlives it keeps working communicators.
// Communicators are destroyed only while the library is alive (the last handle freed, or zg_shutdown): at process exit, static
// destruction may run after librccl (dlopen'ed later than this library) and HIP have torn their own state down, so a set that is
// still referenced then is LEAKED on purpose (g_exiting, set by an atexit hook registered when the first set is created).
static std::atomic<bool> g_exiting{false};
struct CommSet {
    int ndev = 0;
    std::vector<ncclComm_t> comms;
    std::mutex mu;  // RCCL communicators are not thread-safe: every GroupStart .. GroupEnd section over this set holds it
    ~CommSet() {
        if (g_exiting.load()) return;
        for (ncclComm_t c : comms)
            if (c && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c);
    }
};
static std::shared_ptr<CommSet> g_commset;  // the set the next handle will share (guarded by g_comm_mu)
static std::atomic<int> g_commsets_created{0};

static int comms_acquire(int ndev, std::shared_ptr<CommSet> &out) {
    std::lock_guard<std::mutex> lk(g_comm_mu);
    if (g_commset && g_commset->ndev == ndev) {
        out = g_commset;
        return ZG_OK;
    }
    ZG_TRY(rccl_load());
    std::vector<int> devs(ndev);
    for (int i = 0; i < ndev; i++) devs[i] = i;
    auto cs = std::make_shared<CommSet>();
    cs->ndev = ndev;
    cs->comms.assign(ndev, nullptr);
    ZG_NCCL(g_rccl.CommInitAll(cs->comms.data(), ndev, devs.data()));
    if (g_commsets_created.fetch_add(1) == 0) (void)atexit([] { g_exiting.store(true); });  // runs before this library's static destructors
    g_commset = cs;  // an older set (other ndev) stays alive through the handles that hold it
    out = cs;
    return ZG_OK;
}

void sharded_shutdown() {
    std::lock_guard<std::mutex> lk(g_comm_mu);
    g_commset.reset();
}

// ------------------------------------------------------------------ one host worker per shard
// Issuing a launch set costs the host ~0.1 ms (about twenty launches); eight devices fed by one thread would be served one
// after the other. Each shard therefore has a persistent worker that runs the closures handed to it on ITS device.
struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = false, quit = false;
    int rc = ZG_OK;
    std::string err;

    void start(int device) {
        th = std::thread([this, device] {
            DeviceScope scope(device);
            std::unique_lock<std::mutex> lk(mu);
            for (;;) {
                cv.wait(lk, [this] { return has_job || quit; });
                if (quit) return;
                std::function<int()> j = std::move(job);
                has_job = false;
                lk.unlock();
                int r = j();
                std::string e = r != ZG_OK ? std::string(zg_last_error()) : std::string();
                lk.lock();
                rc = r;
                err = e;
                done = true;
                cv.notify_all();
            }
        });
    }
    void submit(std::function<int()> j) {
        std::lock_guard<std::mutex> lk(mu);
        job = std::move(j);
        has_job = true;
        done = false;
        cv.notify_all();
    }
    int wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this] { return done; });
        if (rc != ZG_OK) set_error(err);  // the error text belongs to the worker thread: hand it to the caller's
        return rc;
    }
    void stop() {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
            cv.notify_all();
        }
        if (th.joinable()) th.join();
    }
};

// How the partials of the shards meet. RCCL: one ncclAllGather (group call, one communicator per device). P2P: the first
// shard's stream copies every other shard's record into its gather buffer (hipMemcpyPeerAsync over xGMI) — used when
// several logical shards share a device (ZG_SHARDS, the single-GPU test configuration; RCCL refuses two ranks on one
// device) or when ZG_SHARD_EXCHANGE=p2p asks for it.
enum Exchange { EX_NONE, EX_RCCL, EX_P2P };

static int env_int_s(const char *name, int dflt) {
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

// logical shards and their devices: one per bound device, unless ZG_SHARDS (a test hook) asks for another number, in
// which case shard i sits on device i % bound
static int shard_layout(std::vector<int> &devs) {
    int nd = bound_devices();
    if (nd < 1) nd = 1;
    int ns = env_int_s("ZG_SHARDS", 0);
    if (ns <= 0) ns = nd;
    if (ns > 64) {
        set_error("sharded: at most 64 shards");
        return ZG_ERR_INVALID;
    }
    int base = primary_device();
    devs.resize(ns);
    for (int i = 0; i < ns; i++) devs[i] = nd > 1 ? i % nd : base;
    return ZG_OK;
}

static int pick_exchange(const std::vector<int> &devs, Exchange &ex) {
    const char *m = getenv("ZG_SHARD_EXCHANGE");
    bool one_per_device = (int)devs.size() == bound_devices();
    for (size_t i = 0; one_per_device && i < devs.size(); i++) one_per_device = devs[i] == (int)i;
    if (m && !strcmp(m, "rccl")) {
        if (!one_per_device) {
            set_error("ZG_SHARD_EXCHANGE=rccl needs exactly one shard per bound device (devices 0..n-1)");
            return ZG_ERR_INVALID;
        }
        ex = EX_RCCL;
    } else if (m && !strcmp(m, "p2p")) {
        ex = devs.size() > 1 ? EX_P2P : EX_NONE;
    } else {
        ex = devs.size() == 1 ? EX_NONE : (one_per_device ? EX_RCCL : EX_P2P);
    }
    return ZG_OK;
}

}  // namespace zg

using namespace zg;

// A sharded call in flight occupies one SLOT of the handle: a stream, staging and exchange buffers per shard, a result record
// and a completion event. The handle's slots rotate, so several MSMs overlap on every device — the latency-bound tail and the
// exchange of one under the ALU-bound accumulation of the next (the per-device handles rotate their own workspaces the same way).
struct zg_sbases_s {
    size_t n = 0;
    struct Shard {
        int device = 0;
        size_t start = 0, len = 0;
        zg_bases_t b = nullptr;
        Worker *worker = nullptr;
    };
    struct SlotShard {
        hipStream_t st = nullptr;
        uint64_t *d_scal = nullptr;    // staging for host scalars (len * 32 * vectors), grown on demand
        size_t scal_cap = 0;           // bytes
        uint64_t *d_send = nullptr;    // this shard's partial records (k * 12 words)
        uint64_t *d_gather = nullptr;  // all shards' records, shard-major (S * k * 12 words)
        hipEvent_t ev_part = nullptr;  // partial ready (P2P exchange)
        hipEvent_t ev_in = nullptr;    // the caller's scalars ready (zg_msm_g1_sharded_dev_async with ready_streams)
    };
    struct Slot {
        std::vector<SlotShard> ps;
        size_t xcap = 0;             // records per shard the exchange buffers hold
        uint64_t *d_out9 = nullptr;  // result records on shard 0's device (9 words each: xy[8], flag word)
        uint64_t *h_out9 = nullptr;  // pinned
        size_t out_cap = 0;
        hipEvent_t done = nullptr;   // recorded on shard 0's stream behind the copy into h_out9
        uint64_t ticket = 0;
        size_t k = 0;
        bool pending = false;
    };
    std::vector<Shard> shards;
    std::vector<Slot> slots;
    uint64_t next_ticket = 1;
    Exchange ex = EX_NONE;
    std::shared_ptr<CommSet> comms;  // EX_RCCL: the communicator set this handle was created with
    std::mutex mu;
};

namespace zg {

static void sbases_destroy(zg_sbases_s *sb) {
    if (!sb) return;
    for (auto &sh : sb->shards)
        if (sh.worker) {
            sh.worker->stop();
            delete sh.worker;
        }
    for (auto &sl : sb->slots) {
        for (size_t i = 0; i < sl.ps.size() && i < sb->shards.size(); i++) {
            auto &p = sl.ps[i];
            DeviceScope scope(sb->shards[i].device);
            if (p.st) (void)hipStreamSynchronize(p.st);
            void *ptrs[] = {p.d_scal, p.d_send, p.d_gather};
            for (void *q : ptrs)
                if (q) (void)hipFree(q);
            if (p.ev_part) (void)hipEventDestroy(p.ev_part);
            if (p.ev_in) (void)hipEventDestroy(p.ev_in);
            if (p.st) stream_release(p.st, sb->shards[i].device);
        }
        if (!sb->shards.empty()) {
            DeviceScope scope(sb->shards[0].device);
            if (sl.d_out9) (void)hipFree(sl.d_out9);
            if (sl.h_out9) (void)hipHostFree(sl.h_out9);
            if (sl.done) (void)hipEventDestroy(sl.done);
        }
    }
    for (auto &sh : sb->shards) {
        DeviceScope scope(sh.device);
        if (sh.b) (void)zg_g1_bases_free(sh.b);
    }
    sb->comms.reset();
    delete sb;
}

// exchange buffers of one (idle) slot for k records per shard
static int slot_reserve(zg_sbases_s *sb, zg_sbases_s::Slot &sl, size_t k) {
    const size_t S = sb->shards.size();
    if (sl.xcap < k && S > 1) {
        for (size_t i = 0; i < S; i++) {
            auto &p = sl.ps[i];
            DeviceScope scope(sb->shards[i].device);
            ZG_HIP(hipStreamSynchronize(p.st));
            if (p.d_send) (void)hipFree(p.d_send);
            if (p.d_gather) (void)hipFree(p.d_gather);
            p.d_send = p.d_gather = nullptr;
            sl.xcap = 0;
            ZG_HIP(hipMalloc((void **)&p.d_send, k * 12 * 8));
            if (i == 0 || sb->ex == EX_RCCL) ZG_HIP(hipMalloc((void **)&p.d_gather, S * k * 12 * 8));  // peer copies land on shard 0 only
        }
        sl.xcap = k;
    }
    if (sl.out_cap < k) {
        DeviceScope scope(sb->shards[0].device);
        ZG_HIP(hipStreamSynchronize(sl.ps[0].st));
        if (sl.d_out9) (void)hipFree(sl.d_out9);
        if (sl.h_out9) (void)hipHostFree(sl.h_out9);
        sl.d_out9 = sl.h_out9 = nullptr;
        sl.out_cap = 0;
        ZG_HIP(hipMalloc((void **)&sl.d_out9, k * 9 * 8));
        ZG_HIP(hipMemset(sl.d_out9, 0, k * 9 * 8));  // a record's flag is written as ONE byte of its ninth word
        ZG_HIP(hipHostMalloc((void **)&sl.h_out9, k * 9 * 8));
        sl.out_cap = k;
    }
    return ZG_OK;
}

// the part of shard sh that an MSM over bases[0, n) touches
static size_t shard_count(const zg_sbases_s::Shard &sh, size_t n) {
    if (n <= sh.start) return 0;
    size_t c = n - sh.start;
    return c < sh.len ? c : sh.len;
}

// Enqueue one sharded call on a free slot: `per_shard(i, slot)` issues shard i's launch set on the slot's stream of that shard
// (on the shard's worker thread, all shards concurrently) and leaves k records in the slot's send buffer — or, with a single
// shard, the k finished result records; then the exchange, the combine on shard 0 and the copy of the k result records into the
// slot's pinned buffer are enqueued behind them. Nothing is waited for: the ticket completes in sharded_wait. The caller holds
// sb->mu.
typedef std::function<int(size_t, zg_sbases_s::Slot &)> PerShard;
static int sharded_submit(zg_sbases_s *sb, size_t k, const PerShard &per_shard, uint64_t *ticket) {
    const size_t S = sb->shards.size(), R = sb->slots.size();
    zg_sbases_s::Slot *slp = nullptr;
    for (size_t t = 0; t < R && !slp; t++) {
        zg_sbases_s::Slot &c = sb->slots[(sb->next_ticket + t) % R];
        if (!c.pending) slp = &c;
    }
    if (!slp) {
        set_error("sharded: every in-flight slot of the handle holds an unfinished call (ZG_SHARDED_INFLIGHT): zg_sharded_wait for a ticket first");
        return ZG_ERR_INVALID;
    }
    zg_sbases_s::Slot &sl = *slp;
    ZG_TRY(slot_reserve(sb, sl, k));
    int rc = ZG_OK;
    if (S == 1) {  // no thread hop for a single shard
        DeviceScope scope(sb->shards[0].device);
        rc = per_shard(0, sl);
    } else {
        for (size_t i = 0; i < S; i++) sb->shards[i].worker->submit([&per_shard, &sl, i] { return per_shard(i, sl); });
        for (size_t i = 0; i < S; i++) {
            int r = sb->shards[i].worker->wait();
            if (r != ZG_OK && rc == ZG_OK) rc = r;
        }
    }
    auto drain = [&] {  // after an error: nothing of the slot may still be in flight when it is handed out again
        for (size_t i = 0; i < S; i++) {
            DeviceScope scope(sb->shards[i].device);
            (void)hipStreamSynchronize(sl.ps[i].st);
        }
    };
    if (rc != ZG_OK) {
        std::string keep = zg_last_error();
        drain();
        set_error(keep);
        return rc;
    }
    const int root_dev = sb->shards[0].device;
    hipStream_t root_st = sl.ps[0].st;
    const uint64_t *d_all = sl.ps[0].d_send;
    rc = [&]() -> int {
        if (sb->ex == EX_RCCL) {
            CommSet *cs = sb->comms.get();
            if (!cs || cs->comms.size() < S) {
                set_error("sharded: the handle has no communicator set");
                return ZG_ERR_INVALID;
            }
            std::lock_guard<std::mutex> comm_lk(cs->mu);  // two handles (two threads) share the set: one group section at a time
            ZG_NCCL(g_rccl.GroupStart());
            for (size_t i = 0; i < S; i++) {
                DeviceScope scope(sb->shards[i].device);
                ncclResult_t r = g_rccl.AllGather(sl.ps[i].d_send, sl.ps[i].d_gather, k * 12, ncclUint64, cs->comms[sb->shards[i].device], sl.ps[i].st);
                if (r != ncclSuccess) {
                    (void)g_rccl.GroupEnd();
                    set_error(std::string("ncclAllGather: ") + g_rccl.GetErrorString(r));
                    return ZG_ERR_HIP;
                }
            }
            ZG_NCCL(g_rccl.GroupEnd());
            d_all = sl.ps[0].d_gather;
        } else if (sb->ex == EX_P2P) {
            DeviceScope scope(root_dev);
            for (size_t i = 0; i < S; i++) {
                ZG_HIP(hipStreamWaitEvent(root_st, sl.ps[i].ev_part, 0));
                if (sb->shards[i].device == root_dev)
                    ZG_HIP(hipMemcpyAsync(sl.ps[0].d_gather + i * k * 12, sl.ps[i].d_send, k * 12 * 8, hipMemcpyDeviceToDevice, root_st));
                else
                    ZG_HIP(hipMemcpyPeerAsync(sl.ps[0].d_gather + i * k * 12, root_dev, sl.ps[i].d_send, sb->shards[i].device, k * 12 * 8, root_st));
            }
            d_all = sl.ps[0].d_gather;
        }
        DeviceScope scope(root_dev);
        if (S > 1) ZG_TRY(msm_combine_batch_enqueue(d_all, S, k * 12, k, root_st, sl.d_out9));
        ZG_HIP(hipMemcpyAsync(sl.h_out9, sl.d_out9, k * 9 * 8, hipMemcpyDeviceToHost, root_st));
        ZG_HIP(hipEventRecord(sl.done, root_st));
        return ZG_OK;
    }();
    if (rc != ZG_OK) {
        std::string keep = zg_last_error();
        drain();
        set_error(keep);
        return rc;
    }
    sl.pending = true;
    sl.k = k;
    sl.ticket = sb->next_ticket++;
    *ticket = sl.ticket;
    return ZG_OK;
}

// complete a ticket: wait for its slot's event, hand out the k result records, free the slot
static int sharded_wait(zg_sbases_s *sb, uint64_t ticket, uint64_t *out_xy, uint8_t *out_inf) {
    zg_sbases_s::Slot *slp = nullptr;
    {
        std::lock_guard<std::mutex> lk(sb->mu);
        for (auto &c : sb->slots)
            if (c.pending && c.ticket == ticket) slp = &c;
    }
    if (!slp) {
        set_error("zg_sharded_wait: unknown or already completed ticket");
        return ZG_ERR_INVALID;
    }
    hipError_t e;
    {
        DeviceScope scope(sb->shards[0].device);
        e = hipEventSynchronize(slp->done);  // the slot's buffers cannot change while it is pending
    }
    std::lock_guard<std::mutex> lk(sb->mu);
    if (e == hipSuccess && out_xy)
        for (size_t j = 0; j < slp->k; j++) {
            for (int l = 0; l < 8; l++) out_xy[8 * j + l] = slp->h_out9[9 * j + l];
            if (out_inf) out_inf[j] = (uint8_t)(slp->h_out9[9 * j + 8] & 0xff);
        }
    slp->pending = false;
    if (e != hipSuccess) {
        set_error(std::string("zg_sharded_wait: ") + hipGetErrorString(e));
        return ZG_ERR_HIP;
    }
    return ZG_OK;
}

// shard i's launch set for k host scalar vectors (each n scalars; the shard's chunk is copied to the slot's staging buffer)
static int shard_host_vectors(zg_sbases_s *sb, size_t i, zg_sbases_s::Slot &sl, size_t n, const uint64_t *const *batches, size_t k) {
    zg_sbases_s::Shard &sh = sb->shards[i];
    zg_sbases_s::SlotShard &p = sl.ps[i];
    const size_t cnt = shard_count(sh, n);
    if (cnt * 32 * k > p.scal_cap) {
        ZG_HIP(hipStreamSynchronize(p.st));
        if (p.d_scal) (void)hipFree(p.d_scal);
        p.d_scal = nullptr;
        p.scal_cap = 0;
        size_t want = sh.len * 32 * k;
        ZG_HIP(hipMalloc((void **)&p.d_scal, want));
        p.scal_cap = want;
    }
    for (size_t j = 0; j < k && cnt; j++)
        ZG_HIP(hipMemcpyAsync(p.d_scal + 4 * cnt * j, batches[j] + 4 * sh.start, cnt * 32, hipMemcpyHostToDevice, p.st));
    if (sb->shards.size() == 1) {  // a single shard finishes its own records: no partial, no combine launch
        if (k == 1) return zg_msm_g1_dev_async(sh.b, 0, cnt, p.d_scal, p.st, sl.d_out9, reinterpret_cast<uint8_t *>(sl.d_out9 + 8));
        return zg_msm_g1_batch_dev(sh.b, cnt, p.d_scal, k, p.st, sl.d_out9);
    }
    if (k == 1) ZG_TRY(zg_msm_g1_partial_fast_dev(sh.b, 0, cnt, p.d_scal, p.st, p.d_send));
    else ZG_TRY(msm_batch_partials_dev(sh.b, cnt, p.d_scal, k, p.st, p.d_send));
    if (sb->ex == EX_P2P) ZG_HIP(hipEventRecord(p.ev_part, p.st));
    return ZG_OK;
}

static int batch_args_ok(zg_sbases_t sb, size_t n, const uint64_t *const *batches, size_t k, const char *who) {
    if (!sb || (k && !batches)) {
        set_error(std::string(who) + ": invalid argument");
        return ZG_ERR_INVALID;
    }
    if (n > sb->n) {
        set_error("msm: range exceeds uploaded bases");
        return ZG_ERR_INVALID;
    }
    for (size_t j = 0; j < k; j++)
        if (n && !batches[j]) {
            set_error(std::string(who) + ": null scalar vector");
            return ZG_ERR_INVALID;
        }
    return ZG_OK;
}

}  // namespace zg

extern "C" {

int zg_shard_bounds(size_t n, int shards, int shard, size_t *start, size_t *len) {
    if (shards < 1 || shard < 0 || shard >= shards || !start || !len) {
        set_error("zg_shard_bounds: invalid argument");
        return ZG_ERR_INVALID;
    }
    const size_t per = (n + (size_t)shards - 1) / (size_t)shards;  // chunk_size = (n + T - 1) / T, src/msm/mod.zig:609
    size_t s0 = (size_t)shard * per;
    if (s0 > n) s0 = n;  // :620: start >= n -> an empty chunk
    *start = s0;
    *len = s0 + per <= n ? per : n - s0;  // :621: end = min(start + chunk_size, n)
    return ZG_OK;
}

int zg_g1_bases_upload_sharded(const uint64_t *xy, const uint8_t *inf, size_t n, const zg_msm_config *cfg, zg_sbases_t *out) {
    ZG_INIT();
    if (!out || (n && !xy)) {
        set_error("zg_g1_bases_upload_sharded: invalid argument");
        return ZG_ERR_INVALID;
    }
    std::vector<int> devs;
    ZG_TRY(shard_layout(devs));
    Exchange ex;
    ZG_TRY(pick_exchange(devs, ex));
    std::shared_ptr<CommSet> comms;
    if (ex == EX_RCCL) ZG_TRY(comms_acquire(bound_devices(), comms));
    const size_t S = devs.size();
    zg_sbases_s *sb = new zg_sbases_s();
    sb->n = n;
    sb->ex = ex;
    sb->comms = comms;
    sb->shards.resize(S);
    int R = env_int_s("ZG_SHARDED_INFLIGHT", 3);
    sb->slots.resize(R < 1 ? 1 : (R > 8 ? 8 : R));
    for (auto &sl : sb->slots) sl.ps.resize(S);
    int rc = ZG_OK;
    for (size_t i = 0; i < S && rc == ZG_OK; i++) {
        zg_sbases_s::Shard &sh = sb->shards[i];
        sh.device = devs[i];
        (void)zg_shard_bounds(n, (int)S, (int)i, &sh.start, &sh.len);
        DeviceScope scope(sh.device);
        hipError_t e = hipSuccess;
        for (auto &sl : sb->slots) {
            auto &p = sl.ps[i];
            p.st = stream_acquire();  // from the runtime's free list: creating a stream costs ~3 ms on this stack
            if (!p.st) e = hipErrorOutOfMemory;
            if (e == hipSuccess) e = hipEventCreateWithFlags(&p.ev_part, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&p.ev_in, hipEventDisableTiming);
            if (e == hipSuccess && i == 0) e = hipEventCreateWithFlags(&sl.done, hipEventDisableTiming);
        }
        if (e != hipSuccess) {
            set_error(std::string("sharded upload: ") + hipGetErrorString(e));
            rc = ZG_ERR_HIP;
            break;
        }
        rc = zg_g1_bases_upload(xy + 8 * sh.start, inf ? inf + sh.start : nullptr, sh.len, cfg, &sh.b);
        if (rc == ZG_OK && S > 1) {
            sh.worker = new Worker();
            sh.worker->start(sh.device);
        }
    }
    for (auto &sl : sb->slots)
        if (rc == ZG_OK) rc = slot_reserve(sb, sl, 1);
    if (rc != ZG_OK) {
        std::string keep = zg_last_error();
        sbases_destroy(sb);
        set_error(keep);
        return rc;
    }
    *out = sb;
    return ZG_OK;
}

int zg_g1_sbases_free(zg_sbases_t sb) {
    if (!sb) return ZG_OK;
    ZG_INIT();
    sbases_destroy(sb);
    return ZG_OK;
}

size_t zg_g1_sbases_len(zg_sbases_t sb) { return sb ? sb->n : 0; }

int zg_g1_sbases_shards(zg_sbases_t sb) { return sb ? (int)sb->shards.size() : 0; }

int zg_g1_sbases_shard(zg_sbases_t sb, int shard, int *device, size_t *start, size_t *len) {
    if (!sb || shard < 0 || shard >= (int)sb->shards.size()) {
        set_error("zg_g1_sbases_shard: invalid argument");
        return ZG_ERR_INVALID;
    }
    if (device) *device = sb->shards[shard].device;
    if (start) *start = sb->shards[shard].start;
    if (len) *len = sb->shards[shard].len;
    return ZG_OK;
}

int zg_g1_sbases_exchange(zg_sbases_t sb) { return sb ? (int)sb->ex : -1; }

int zg_g1_sbases_inflight(zg_sbases_t sb) { return sb ? (int)sb->slots.size() : 0; }

int zg_sharded_comm_sets_created(void) { return g_commsets_created.load(); }

int zg_msm_g1_batch_sharded_async(zg_sbases_t sb, size_t n, const uint64_t *const *batches, size_t k, uint64_t *ticket) {
    ZG_INIT();
    ZG_TRY(batch_args_ok(sb, n, batches, k, "zg_msm_g1_batch_sharded_async"));
    if (!ticket || k == 0) {
        set_error("zg_msm_g1_batch_sharded_async: invalid argument (k must be at least 1)");
        return ZG_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lk(sb->mu);
    PerShard per_shard = [&](size_t i, zg_sbases_s::Slot &sl) -> int { return shard_host_vectors(sb, i, sl, n, batches, k); };
    return sharded_submit(sb, k, per_shard, ticket);
}

int zg_msm_g1_sharded_dev_async(zg_sbases_t sb, size_t n, const uint64_t *const *d_scalars_per_shard, void *const *ready_streams,
                                uint64_t *ticket) {
    ZG_INIT();
    if (!sb || !ticket || (n && !d_scalars_per_shard)) {
        set_error("zg_msm_g1_sharded_dev_async: invalid argument");
        return ZG_ERR_INVALID;
    }
    if (n > sb->n) {
        set_error("msm: range exceeds uploaded bases");
        return ZG_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lk(sb->mu);
    PerShard per_shard = [&](size_t i, zg_sbases_s::Slot &sl) -> int {
        zg_sbases_s::Shard &sh = sb->shards[i];
        zg_sbases_s::SlotShard &p = sl.ps[i];
        size_t cnt = shard_count(sh, n);
        if (cnt && !d_scalars_per_shard[i]) {
            set_error("zg_msm_g1_sharded_dev: null shard pointer");
            return ZG_ERR_INVALID;
        }
        if (cnt && ready_streams && ready_streams[i]) {  // order the shard's work behind whatever fills its scalars
            ZG_HIP(hipEventRecord(p.ev_in, reinterpret_cast<hipStream_t>(ready_streams[i])));
            ZG_HIP(hipStreamWaitEvent(p.st, p.ev_in, 0));
        }
        if (sb->shards.size() == 1)
            return zg_msm_g1_dev_async(sh.b, 0, cnt, d_scalars_per_shard[i], p.st, sl.d_out9, reinterpret_cast<uint8_t *>(sl.d_out9 + 8));
        ZG_TRY(zg_msm_g1_partial_fast_dev(sh.b, 0, cnt, d_scalars_per_shard[i], p.st, p.d_send));
        if (sb->ex == EX_P2P) ZG_HIP(hipEventRecord(p.ev_part, p.st));
        return ZG_OK;
    };
    return sharded_submit(sb, 1, per_shard, ticket);
}

int zg_sharded_wait(zg_sbases_t sb, uint64_t ticket, uint64_t *out_xy, uint8_t *out_inf) {
    ZG_INIT();
    if (!sb) {
        set_error("zg_sharded_wait: invalid argument");
        return ZG_ERR_INVALID;
    }
    return sharded_wait(sb, ticket, out_xy, out_inf);
}

int zg_msm_g1_batch_sharded(zg_sbases_t sb, size_t n, const uint64_t *const *batches, size_t k, uint64_t *out_xy, uint8_t *out_inf) {
    ZG_INIT();
    ZG_TRY(batch_args_ok(sb, n, batches, k, "zg_msm_g1_batch_sharded"));
    if (k && !out_xy) {
        set_error("zg_msm_g1_batch_sharded: invalid argument");
        return ZG_ERR_INVALID;
    }
    if (k == 0) return ZG_OK;
    uint64_t ticket = 0;
    ZG_TRY(zg_msm_g1_batch_sharded_async(sb, n, batches, k, &ticket));
    return sharded_wait(sb, ticket, out_xy, out_inf);
}

int zg_msm_g1_sharded(zg_sbases_t sb, size_t n, const uint64_t *scalars, uint64_t out_xy[8], uint8_t *out_inf) {
    if (n && !scalars) {
        set_error("zg_msm_g1_sharded: invalid argument");
        return ZG_ERR_INVALID;
    }
    const uint64_t *one[1] = {scalars};
    return zg_msm_g1_batch_sharded(sb, n, one, 1, out_xy, out_inf);
}

int zg_msm_g1_sharded_dev(zg_sbases_t sb, size_t n, const uint64_t *const *d_scalars_per_shard, uint64_t out_xy[8], uint8_t *out_inf) {
    ZG_INIT();
    if (!sb || !out_xy || (n && !d_scalars_per_shard)) {
        set_error("zg_msm_g1_sharded_dev: invalid argument");
        return ZG_ERR_INVALID;
    }
    uint64_t ticket = 0;
    ZG_TRY(zg_msm_g1_sharded_dev_async(sb, n, d_scalars_per_shard, nullptr, &ticket));
    return sharded_wait(sb, ticket, out_xy, out_inf);
}

}  // extern "C"

// ------------------------------------------------------------------ sharded sumcheck session
struct zg_ssc_s {
    int layout = 0;
    size_t len = 0;      // current length of the WHOLE table
    size_t S = 1;        // shards (a power of two)
    std::vector<int> devs;
    std::vector<zg_sc_t> sess;  // one ordinary device session per shard (empty once the tail has started)
    zg_sc_t tail = nullptr;     // the last log2(S) rounds: the S residual elements, on the first device
    std::mutex mu;
};

namespace zg {

static void ssc_destroy(zg_ssc_s *s) {
    if (!s) return;
    for (size_t i = 0; i < s->sess.size(); i++)
        if (s->sess[i]) {
            DeviceScope scope(s->devs[i]);
            (void)zg_sumcheck_close(s->sess[i]);
        }
    if (s->tail) {
        DeviceScope scope(s->devs[0]);
        (void)zg_sumcheck_close(s->tail);
    }
    delete s;
}

// the S residual elements (one per shard, shard order = index order for both layouts) become the tail table on device 0
static int ssc_enter_tail(zg_ssc_s *s) {
    std::vector<uint64_t> res(4 * s->S);
    int rc = ZG_OK;
    for (size_t i = 0; i < s->S && rc == ZG_OK; i++) {
        DeviceScope scope(s->devs[i]);
        rc = zg_sumcheck_read(s->sess[i], res.data() + 4 * i);
    }
    for (size_t i = 0; i < s->S; i++) {  // every shard session is closed, also when a read failed
        DeviceScope scope(s->devs[i]);
        (void)zg_sumcheck_close(s->sess[i]);
        s->sess[i] = nullptr;
    }
    s->sess.clear();
    if (rc != ZG_OK) return rc;
    DeviceScope scope(s->devs[0]);
    return zg_sumcheck_open(res.data(), s->S, s->layout, &s->tail);
}

}  // namespace zg

extern "C" {

int zg_sumcheck_open_sharded(const uint64_t *evals, size_t len, int layout, zg_ssc_t *out) {
    ZG_INIT();
    if (!evals || !out || len == 0 || (len & (len - 1)) || (layout != ZG_SC_HIGH_HALF && layout != ZG_SC_LOW_PAIR)) {
        set_error("zg_sumcheck_open_sharded: len must be a power of two and layout valid");
        return ZG_ERR_INVALID;
    }
    std::vector<int> devs;
    ZG_TRY(shard_layout(devs));
    size_t S = 1;
    while (2 * S <= devs.size() && 2 * S <= len) S *= 2;  // shards: the largest power of two the devices and the table allow
    zg_ssc_s *s = new zg_ssc_s();
    s->layout = layout;
    s->len = len;
    s->S = S;
    s->devs.assign(devs.begin(), devs.begin() + S);
    s->sess.assign(S, nullptr);
    const size_t per = len / S;
    std::vector<uint64_t> tmp;
    int rc = ZG_OK;
    for (size_t i = 0; i < S && rc == ZG_OK; i++) {
        DeviceScope scope(s->devs[i]);
        const uint64_t *src = evals + 4 * per * i;  // LOW_PAIR: contiguous chunk (high index bits)
        if (layout == ZG_SC_HIGH_HALF && S > 1) {   // HIGH_HALF: residue class i mod S (low index bits)
            tmp.resize(4 * per);
            for (size_t j = 0; j < per; j++)
                for (int l = 0; l < 4; l++) tmp[4 * j + l] = evals[4 * (j * S + i) + l];
            src = tmp.data();
        }
        rc = zg_sumcheck_open(src, per, layout, &s->sess[i]);
    }
    if (rc == ZG_OK && per == 1 && S > 1) rc = ssc_enter_tail(s);
    if (rc != ZG_OK) {
        std::string keep = zg_last_error();
        ssc_destroy(s);
        set_error(keep);
        return rc;
    }
    *out = s;
    return ZG_OK;
}

int zg_sumcheck_shards(zg_ssc_t s) { return s ? (int)s->S : 0; }
size_t zg_sumcheck_len_sharded(zg_ssc_t s) { return s ? s->len : 0; }

int zg_sumcheck_round_sums_sharded(zg_ssc_t s, uint64_t g0[4], uint64_t g1[4]) {
    ZG_INIT();
    if (!s || !g0 || !g1) {
        set_error("zg_sumcheck_round_sums_sharded: invalid argument");
        return ZG_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lk(s->mu);
    if (s->tail) {
        DeviceScope scope(s->devs[0]);
        return zg_sumcheck_round_sums(s->tail, g0, g1);
    }
    // rounds whose sums the preceding fold did not leave behind (round 0, the first round after open): every shard's pass is
    // enqueued before the first mailbox is waited for, so the round costs one latency, not S
    for (size_t i = 0; i < s->S; i++) {
        DeviceScope scope(s->devs[i]);
        ZG_TRY(sc_round_sums_start(s->sess[i]));
    }
    uint64_t a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < s->S; i++) {
        uint64_t x[4], y[4];
        DeviceScope scope(s->devs[i]);
        ZG_TRY(zg_sumcheck_round_sums(s->sess[i], x, y));  // the shards computed their sums concurrently (fused into the last fold)
        fr_add_host(a, a, x);
        fr_add_host(b, b, y);
    }
    for (int l = 0; l < 4; l++) {
        g0[l] = a[l];
        g1[l] = b[l];
    }
    return ZG_OK;
}

int zg_sumcheck_bind_sharded(zg_ssc_t s, const uint64_t r[4]) {
    ZG_INIT();
    if (!s || !r) {
        set_error("zg_sumcheck_bind_sharded: invalid argument");
        return ZG_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lk(s->mu);
    if (s->len < 2) {
        set_error("zg_sumcheck_bind_sharded: the table is already a single element");
        return ZG_ERR_INVALID;
    }
    if (s->tail) {
        DeviceScope scope(s->devs[0]);
        ZG_TRY(zg_sumcheck_bind(s->tail, r));
        s->len /= 2;
        return ZG_OK;
    }
    for (size_t i = 0; i < s->S; i++) {  // asynchronous on each shard's device: the folds of the shards overlap
        DeviceScope scope(s->devs[i]);
        ZG_TRY(zg_sumcheck_bind(s->sess[i], r));
    }
    s->len /= 2;
    if (s->S > 1 && s->len == s->S) ZG_TRY(ssc_enter_tail(s));
    return ZG_OK;
}

int zg_sumcheck_final_sharded(zg_ssc_t s, uint64_t out[4]) {
    ZG_INIT();
    if (!s || !out) {
        set_error("zg_sumcheck_final_sharded: invalid argument");
        return ZG_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lk(s->mu);
    if (s->len != 1) {
        set_error("zg_sumcheck_final_sharded: the table is not a single element yet");
        return ZG_ERR_INVALID;
    }
    DeviceScope scope(s->devs[0]);
    return zg_sumcheck_final(s->tail ? s->tail : s->sess[0], out);
}

int zg_sumcheck_close_sharded(zg_ssc_t s) {
    if (!s) return ZG_OK;
    ZG_INIT();
    ssc_destroy(s);
    return ZG_OK;
}

}  // extern "C"
