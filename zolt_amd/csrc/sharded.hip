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

#include <condition_variable>
#include <functional>
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
static std::vector<ncclComm_t> g_comms;  // one per bound device, ncclCommInitAll order (rank i = device i)

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

// communicator over devices 0..ndev-1 (created once, on first need)
static int comms_ensure(int ndev) {
    std::lock_guard<std::mutex> lk(g_comm_mu);
    if ((int)g_comms.size() == ndev) return ZG_OK;
    if (!g_comms.empty()) {
        set_error("sharded: the set of bound devices changed after the communicator was created");
        return ZG_ERR_INVALID;
    }
    ZG_TRY(rccl_load());
    std::vector<int> devs(ndev);
    for (int i = 0; i < ndev; i++) devs[i] = i;
    std::vector<ncclComm_t> comms(ndev);
    ZG_NCCL(g_rccl.CommInitAll(comms.data(), ndev, devs.data()));
    g_comms = comms;
    return ZG_OK;
}

void sharded_shutdown() {
    std::lock_guard<std::mutex> lk(g_comm_mu);
    for (ncclComm_t c : g_comms)
        if (c && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c);
    g_comms.clear();
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

struct zg_sbases_s {
    size_t n = 0;
    struct Shard {
        int device = 0;
        size_t start = 0, len = 0;
        zg_bases_t b = nullptr;
        hipStream_t st = nullptr;
        uint64_t *d_scal = nullptr;    // staging for host scalars (len * 32 * vectors), grown on demand
        size_t scal_cap = 0;           // bytes
        uint64_t *d_send = nullptr;    // this shard's partial records (k * 12 words)
        uint64_t *d_gather = nullptr;  // all shards' records, shard-major (S * k * 12 words)
        size_t xcap = 0;               // k the exchange buffers hold
        hipEvent_t ev = nullptr;       // partial ready (P2P exchange)
        Worker *worker = nullptr;
    };
    std::vector<Shard> shards;
    Exchange ex = EX_NONE;
    uint64_t *d_out9 = nullptr;  // result records on shard 0's device
    size_t out_cap = 0;
    uint64_t *h_out9 = nullptr;  // pinned
    std::mutex mu;
};

namespace zg {

static void sbases_destroy(zg_sbases_s *sb) {
    if (!sb) return;
    for (auto &sh : sb->shards) {
        if (sh.worker) {
            sh.worker->stop();
            delete sh.worker;
        }
        DeviceScope scope(sh.device);
        if (sh.st) (void)hipStreamSynchronize(sh.st);
        if (sh.b) (void)zg_g1_bases_free(sh.b);
        void *ptrs[] = {sh.d_scal, sh.d_send, sh.d_gather};
        for (void *p : ptrs)
            if (p) (void)hipFree(p);
        if (sh.ev) (void)hipEventDestroy(sh.ev);
        if (sh.st) (void)hipStreamDestroy(sh.st);
    }
    if (!sb->shards.empty()) {
        DeviceScope scope(sb->shards[0].device);
        if (sb->d_out9) (void)hipFree(sb->d_out9);
        if (sb->h_out9) (void)hipHostFree(sb->h_out9);
    }
    delete sb;
}

// exchange buffers for k records per shard
static int shard_reserve_exchange(zg_sbases_s *sb, size_t k) {
    const size_t S = sb->shards.size();
    for (auto &sh : sb->shards) {
        if (sh.xcap >= k) continue;
        DeviceScope scope(sh.device);
        if (sh.st) ZG_HIP(hipStreamSynchronize(sh.st));
        if (sh.d_send) (void)hipFree(sh.d_send);
        if (sh.d_gather) (void)hipFree(sh.d_gather);
        sh.d_send = sh.d_gather = nullptr;
        sh.xcap = 0;
        ZG_HIP(hipMalloc((void **)&sh.d_send, k * 12 * 8));
        ZG_HIP(hipMalloc((void **)&sh.d_gather, S * k * 12 * 8));
        sh.xcap = k;
    }
    if (sb->out_cap < k) {
        DeviceScope scope(sb->shards[0].device);
        if (sb->d_out9) (void)hipFree(sb->d_out9);
        if (sb->h_out9) (void)hipHostFree(sb->h_out9);
        sb->d_out9 = sb->h_out9 = nullptr;
        sb->out_cap = 0;
        ZG_HIP(hipMalloc((void **)&sb->d_out9, k * 9 * 8));
        ZG_HIP(hipHostMalloc((void **)&sb->h_out9, k * 9 * 8));
        sb->out_cap = k;
    }
    return ZG_OK;
}

// the part of shard sh that an MSM over bases[0, n) touches
static size_t shard_count(const zg_sbases_s::Shard &sh, size_t n) {
    if (n <= sh.start) return 0;
    size_t c = n - sh.start;
    return c < sh.len ? c : sh.len;
}

// Run `per_shard(i)` on every shard's worker (each on its own device), then exchange k records per shard and combine them on
// shard 0; the k result records land in sb->h_out9.
static int sharded_finish(zg_sbases_s *sb, size_t k, const std::function<int(size_t)> &per_shard) {
    const size_t S = sb->shards.size();
    for (size_t i = 0; i < S; i++) sb->shards[i].worker->submit([&per_shard, i] { return per_shard(i); });
    int rc = ZG_OK;
    for (size_t i = 0; i < S; i++) {
        int r = sb->shards[i].worker->wait();
        if (r != ZG_OK && rc == ZG_OK) rc = r;
    }
    if (rc != ZG_OK) {
        for (auto &sh : sb->shards) {
            DeviceScope scope(sh.device);
            (void)hipStreamSynchronize(sh.st);
        }
        return rc;
    }
    zg_sbases_s::Shard &root = sb->shards[0];
    const uint64_t *d_all = root.d_send;
    if (sb->ex == EX_RCCL) {
        ZG_NCCL(g_rccl.GroupStart());
        for (size_t i = 0; i < S; i++) {
            zg_sbases_s::Shard &sh = sb->shards[i];
            DeviceScope scope(sh.device);
            ncclResult_t r = g_rccl.AllGather(sh.d_send, sh.d_gather, k * 12, ncclUint64, g_comms[sh.device], sh.st);
            if (r != ncclSuccess) {
                (void)g_rccl.GroupEnd();
                set_error(std::string("ncclAllGather: ") + g_rccl.GetErrorString(r));
                return ZG_ERR_HIP;
            }
        }
        ZG_NCCL(g_rccl.GroupEnd());
        d_all = root.d_gather;
    } else if (sb->ex == EX_P2P) {
        DeviceScope scope(root.device);
        for (size_t i = 0; i < S; i++) {
            zg_sbases_s::Shard &sh = sb->shards[i];
            ZG_HIP(hipStreamWaitEvent(root.st, sh.ev, 0));
            if (sh.device == root.device)
                ZG_HIP(hipMemcpyAsync(root.d_gather + i * k * 12, sh.d_send, k * 12 * 8, hipMemcpyDeviceToDevice, root.st));
            else
                ZG_HIP(hipMemcpyPeerAsync(root.d_gather + i * k * 12, root.device, sh.d_send, sh.device, k * 12 * 8, root.st));
        }
        d_all = root.d_gather;
    }
    DeviceScope scope(root.device);
    ZG_TRY(msm_combine_batch_enqueue(d_all, S, k * 12, k, root.st, sb->d_out9));
    ZG_HIP(hipMemcpyAsync(sb->h_out9, sb->d_out9, k * 9 * 8, hipMemcpyDeviceToHost, root.st));
    ZG_HIP(hipStreamSynchronize(root.st));
    return ZG_OK;
}

static void copy_records(const zg_sbases_s *sb, size_t k, uint64_t *out_xy, uint8_t *out_inf) {
    for (size_t j = 0; j < k; j++) {
        for (int l = 0; l < 8; l++) out_xy[8 * j + l] = sb->h_out9[9 * j + l];
        if (out_inf) out_inf[j] = (uint8_t)(sb->h_out9[9 * j + 8] & 0xff);
    }
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
    if (ex == EX_RCCL) ZG_TRY(comms_ensure(bound_devices()));
    const size_t S = devs.size();
    zg_sbases_s *sb = new zg_sbases_s();
    sb->n = n;
    sb->ex = ex;
    sb->shards.resize(S);
    int rc = ZG_OK;
    for (size_t i = 0; i < S && rc == ZG_OK; i++) {
        zg_sbases_s::Shard &sh = sb->shards[i];
        sh.device = devs[i];
        (void)zg_shard_bounds(n, (int)S, (int)i, &sh.start, &sh.len);
        DeviceScope scope(sh.device);
        hipError_t e = hipStreamCreateWithFlags(&sh.st, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sh.ev, hipEventDisableTiming);
        if (e != hipSuccess) {
            set_error(std::string("sharded upload: ") + hipGetErrorString(e));
            rc = ZG_ERR_HIP;
            break;
        }
        rc = zg_g1_bases_upload(xy + 8 * sh.start, inf ? inf + sh.start : nullptr, sh.len, cfg, &sh.b);
        if (rc == ZG_OK) {
            sh.worker = new Worker();
            sh.worker->start(sh.device);
        }
    }
    if (rc == ZG_OK) rc = shard_reserve_exchange(sb, 1);
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

int zg_msm_g1_batch_sharded(zg_sbases_t sb, size_t n, const uint64_t *const *batches, size_t k, uint64_t *out_xy, uint8_t *out_inf) {
    ZG_INIT();
    if (!sb || (k && (!batches || !out_xy))) {
        set_error("zg_msm_g1_batch_sharded: invalid argument");
        return ZG_ERR_INVALID;
    }
    if (n > sb->n) {
        set_error("msm: range exceeds uploaded bases");
        return ZG_ERR_INVALID;
    }
    if (k == 0) return ZG_OK;
    for (size_t j = 0; j < k; j++)
        if (n && !batches[j]) {
            set_error("zg_msm_g1_batch_sharded: null scalar vector");
            return ZG_ERR_INVALID;
        }
    std::lock_guard<std::mutex> lk(sb->mu);
    ZG_TRY(shard_reserve_exchange(sb, k));
    auto per_shard = [&](size_t i) -> int {
        zg_sbases_s::Shard &sh = sb->shards[i];
        size_t cnt = shard_count(sh, n);
        if (cnt * 32 * k > sh.scal_cap) {
            ZG_HIP(hipStreamSynchronize(sh.st));
            if (sh.d_scal) (void)hipFree(sh.d_scal);
            sh.d_scal = nullptr;
            sh.scal_cap = 0;
            size_t want = sh.len * 32 * k;
            ZG_HIP(hipMalloc((void **)&sh.d_scal, want));
            sh.scal_cap = want;
        }
        for (size_t j = 0; j < k && cnt; j++)
            ZG_HIP(hipMemcpyAsync(sh.d_scal + 4 * cnt * j, batches[j] + 4 * sh.start, cnt * 32, hipMemcpyHostToDevice, sh.st));
        if (k == 1) ZG_TRY(zg_msm_g1_partial_fast_dev(sh.b, 0, cnt, sh.d_scal, sh.st, sh.d_send));
        else ZG_TRY(msm_batch_partials_dev(sh.b, cnt, sh.d_scal, k, sh.st, sh.d_send));
        if (sb->ex == EX_P2P) ZG_HIP(hipEventRecord(sh.ev, sh.st));
        return ZG_OK;
    };
    ZG_TRY(sharded_finish(sb, k, per_shard));
    copy_records(sb, k, out_xy, out_inf);
    return ZG_OK;
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
    if (n > sb->n) {
        set_error("msm: range exceeds uploaded bases");
        return ZG_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lk(sb->mu);
    ZG_TRY(shard_reserve_exchange(sb, 1));
    auto per_shard = [&](size_t i) -> int {
        zg_sbases_s::Shard &sh = sb->shards[i];
        size_t cnt = shard_count(sh, n);
        if (cnt && !d_scalars_per_shard[i]) {
            set_error("zg_msm_g1_sharded_dev: null shard pointer");
            return ZG_ERR_INVALID;
        }
        ZG_TRY(zg_msm_g1_partial_fast_dev(sh.b, 0, cnt, d_scalars_per_shard[i], sh.st, sh.d_send));
        if (sb->ex == EX_P2P) ZG_HIP(hipEventRecord(sh.ev, sh.st));
        return ZG_OK;
    };
    ZG_TRY(sharded_finish(sb, 1, per_shard));
    copy_records(sb, 1, out_xy, out_inf);
    return ZG_OK;
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
    for (size_t i = 0; i < s->S; i++) {
        DeviceScope scope(s->devs[i]);
        ZG_TRY(zg_sumcheck_read(s->sess[i], res.data() + 4 * i));
    }
    for (size_t i = 0; i < s->S; i++) {
        DeviceScope scope(s->devs[i]);
        (void)zg_sumcheck_close(s->sess[i]);
        s->sess[i] = nullptr;
    }
    s->sess.clear();
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
