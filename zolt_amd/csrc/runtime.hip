// runtime.hip — lifecycle, error reporting, raw device memory and the elementwise field
// kernels of libzolt_gpu.so (C ABI: include/zolt_gpu.h).
#include <atomic>
#include <cstdlib>
#include <ctime>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "common.hip.h"
#include "field.hip.h"
#include "fp29.hip.h"

namespace zg {

static thread_local std::string t_err;
static std::mutex g_mu;
static std::atomic<bool> g_inited{false};  // read without the mutex on every entry point
static std::atomic<int> g_primary{-1};     // device bound by zg_init / zg_init_devices
static std::atomic<int> g_ndev{0};         // devices bound (1 after zg_init, n after zg_init_devices(n)); they are 0..n-1 then
static hipStream_t g_streams[ZG_MAX_DEVICES] = {};  // the library's own stream per device, created on first use

void set_error(const std::string &msg) { t_err = msg; }
static thread_local int t_dev_override = -1;
void set_device_override(int dev) { t_dev_override = dev; }
int device_override() { return t_dev_override; }
int primary_device() { return t_dev_override >= 0 ? t_dev_override : g_primary.load(std::memory_order_acquire); }
int bound_devices() { return g_ndev.load(std::memory_order_acquire); }
int current_device() {
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess) return -1;
    return d;
}

hipStream_t lib_stream() {
    int d = current_device();
    if (d < 0 || d >= ZG_MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_streams[d] && hipStreamCreateWithFlags(&g_streams[d], hipStreamNonBlocking) != hipSuccess) g_streams[d] = nullptr;
    return g_streams[d];
}

// Streams for sumcheck sessions: hipStreamCreate + hipStreamDestroy cost ~3 ms on this stack, a session open must not pay that.
// A session takes an idle stream from this per-device free list (or creates one) and hands it back when its buffers are freed;
// the streams live until zg_shutdown.
static std::vector<hipStream_t> g_idle_streams[ZG_MAX_DEVICES];
hipStream_t stream_acquire() {
    int d = current_device();
    if (d < 0 || d >= ZG_MAX_DEVICES) return nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_idle_streams[d].empty()) {
            hipStream_t st = g_idle_streams[d].back();
            g_idle_streams[d].pop_back();
            return st;
        }
    }
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return nullptr;
    return st;
}
void stream_release(hipStream_t st, int device) {
    if (!st || device < 0 || device >= ZG_MAX_DEVICES) return;
    std::lock_guard<std::mutex> lk(g_mu);
    g_idle_streams[device].push_back(st);
}

// Groups of three streams created back to back. HIP gives a new stream the next of its hardware queues in turn (four by default), so the
// members of a group sit on different queues and their kernels can overlap; two streams taken one by one from the free list above may
// share a queue, and then what was enqueued to overlap runs one after the other. Created under the lock, so that no other creation of
// this library falls in between.
struct StreamGroup { hipStream_t s[3]; };
static std::vector<StreamGroup> g_idle_groups[ZG_MAX_DEVICES];
bool stream_group_acquire(hipStream_t out[3]) {
    int d = current_device();
    out[0] = out[1] = out[2] = nullptr;
    if (d < 0 || d >= ZG_MAX_DEVICES) return false;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_idle_groups[d].empty()) {
        for (int i = 0; i < 3; i++) out[i] = g_idle_groups[d].back().s[i];
        g_idle_groups[d].pop_back();
        return true;
    }
    for (int i = 0; i < 3; i++)
        if (hipStreamCreateWithFlags(&out[i], hipStreamNonBlocking) != hipSuccess) {
            for (int j = 0; j < i; j++) (void)hipStreamDestroy(out[j]);
            out[0] = out[1] = out[2] = nullptr;
            return false;
        }
    return true;
}
void stream_group_release(const hipStream_t s[3], int device) {
    if (!s[0] || device < 0 || device >= ZG_MAX_DEVICES) return;
    std::lock_guard<std::mutex> lk(g_mu);
    g_idle_groups[device].push_back(StreamGroup{{s[0], s[1], s[2]}});
}

// device >= 0: bind that device as the primary; -1: keep the calling thread's current device. ndev: devices 0..ndev-1 are bound
// (zg_init_devices), 1 for the one-GPU-per-process model.
static int do_init(int device, int ndev) {
    std::lock_guard<std::mutex> lk(g_mu);
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        set_error("no HIP device available (libzolt_gpu has no CPU fallback)");
        return ZG_ERR_NO_DEVICE;
    }
    if (count > ZG_MAX_DEVICES) count = ZG_MAX_DEVICES;
    if (g_inited) {
        // idempotent; a later zg_init_devices may widen the set of bound devices but never moves the primary
        if (ndev > count) {
            set_error("zg_init_devices: more devices requested than are visible");
            return ZG_ERR_INVALID;
        }
        if (ndev > g_ndev) {
            if (g_primary != 0) {
                set_error("zg_init_devices: the process is already bound to a device other than 0 (one-GPU-per-process model)");
                return ZG_ERR_INVALID;
            }
            g_ndev = ndev;
        }
        return ZG_OK;
    }
    if (device >= count || ndev > count) {
        set_error("zg_init: device ordinal out of range");
        return ZG_ERR_INVALID;
    }
    if (device >= 0) ZG_HIP(hipSetDevice(device));
    int cur = 0;
    ZG_HIP(hipGetDevice(&cur));
    if (cur >= ZG_MAX_DEVICES) {
        set_error("zg_init: device ordinal beyond ZG_MAX_DEVICES");
        return ZG_ERR_INVALID;
    }
    ZG_HIP(hipStreamCreateWithFlags(&g_streams[cur], hipStreamNonBlocking));
    g_primary = cur;
    g_ndev = ndev;
    g_inited = true;
    return ZG_OK;
}

int ensure_init() {
    if (g_inited) return ZG_OK;
    return do_init(-1, 1);
}

// ------------------------------------------------------------------ the device-memory pool
// ONE pool of device memory per process behind every transient allocation of the library: the scratch buffers of the host-pointer entry
// points (Scratch), the session tables of zg_rrw_* / zg_rwc_*, and the caller's own tables (zg_dev_alloc / zg_dev_free). hipMalloc + hipFree
// of the hundreds of MB a prover stage holds cost 1-40 ms DEPENDING ON THE BOX (round 4: the same binary set a Stage-4 session up in 4.8 ms
// on one machine and 45 ms on the driver's; its ten tables were raw hipMalloc calls per open), so freed blocks are kept in size classes
// (the request rounded up to an eighth of its leading power of two: at most 12.5 % slack) and handed out again, whatever their size. The
// pool is sized for the part: by default it keeps up to a quarter of the device's memory (72 GB of the MI355X's 288 GB; ZG_DEV_ALLOC_CACHE_MB
// overrides, 0 = no caching). An allocation that fails anywhere in the library (pool_alloc, dev_malloc: handle tables, MSM workspaces)
// gives every idle block back to the driver and tries once more. A block is idle by contract when it is freed: the caller has synchronised
// the work that used it (Scratch / SyncGuard, session close; zg_dev_free synchronises the device itself).
struct PoolBlock { size_t bytes; int dev; };
static std::mutex g_pool_mu;
static std::unordered_map<void *, PoolBlock> g_pool_live;            // handed out (class size, device)
static std::multimap<std::pair<int, size_t>, void *> g_pool_idle;    // (device, class size) -> idle block
static size_t g_pool_cached = 0;                                     // bytes in g_pool_idle
static size_t pool_cap() {
    static const size_t cap = [] {
        const char *v = getenv("ZG_DEV_ALLOC_CACHE_MB");
        if (v && *v) {
            long mb = atol(v);
            return (size_t)(mb < 0 ? 0 : mb) << 20;
        }
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || total_b == 0) return (size_t)8 << 30;
        return total_b / 4;
    }();
    return cap;
}
static size_t pool_class(size_t bytes) {
    if (bytes <= 4096) return 4096;
    int lg = 63 - __builtin_clzll((unsigned long long)bytes);  // 2^lg <= bytes
    const size_t step = (size_t)1 << (lg - 3);
    return (bytes + step - 1) / step * step;
}
void pool_trim() {  // every idle block back to the driver (blocks in use stay with their owners)
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (auto &kv : g_pool_idle) (void)hipFree(kv.second);
    g_pool_idle.clear();
    g_pool_cached = 0;
}
hipError_t dev_malloc(void **p, size_t bytes) {  // hipMalloc for allocations that live with a handle; out of memory: trim the pool, retry once
    hipError_t e = hipMalloc(p, bytes ? bytes : 16);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        pool_trim();
        e = hipMalloc(p, bytes ? bytes : 16);
    }
    return e;
}

// ---- ZG_POOL_DEBUG: the check behind "a block is idle by contract when it is freed" (round-5 review: nothing checked it; a block freed
// with work in flight and handed to another stream is a silent wrong answer). MSM.compute is called from concurrent host threads
// (src/msm/mod.zig:355-372,637,732), so the pool is where two calls' memory can meet.
//   ZG_POOL_DEBUG=1  pool_free fills the block with a poison pattern (on a stream of the pool's own, ordered with nothing: work still in
//                    flight on the block reads poison and its parity test fails loudly instead of rarely); pool_alloc, before it hands an
//                    idle block out again, waits for the device and verifies that every word is still poison — a word that is not was
//                    WRITTEN AFTER THE FREE. Such a block is refused (nullptr, error text names it) and counted. Fresh blocks from the
//                    driver are poisoned too (a kernel that relied on zeroed scratch reads 0xDB...). Frees made while a library stream of
//                    the device still had work queued are counted as "suspect" (information: other calls' work is on those streams too).
//   ZG_POOL_DEBUG=2  the same, and a hit aborts the process on the spot.
// zg_pool_debug_stats (include/zolt_gpu_internal.h) reads the counters; zg_pool_debug_selftest breaks the contract on purpose.
static const uint32_t POOL_POISON = 0xDBDBDBDBu;
static int pool_debug() {
    static const int on = [] { const char *v = getenv("ZG_POOL_DEBUG"); return v && *v ? atoi(v) : 0; }();
    return on;
}
static std::atomic<uint64_t> g_pd_hits{0}, g_pd_suspect{0}, g_pd_checked{0}, g_pd_poisoned_bytes{0};
static hipStream_t g_pd_stream[ZG_MAX_DEVICES] = {};
static uint32_t *g_pd_count[ZG_MAX_DEVICES] = {};  // pinned: [0] words that are not poison, [1] index of the first one
static hipStream_t pd_stream(int dev) {            // under g_pool_mu
    if (dev < 0 || dev >= ZG_MAX_DEVICES) return nullptr;
    if (!g_pd_stream[dev]) {
        if (hipStreamCreateWithFlags(&g_pd_stream[dev], hipStreamNonBlocking) != hipSuccess) g_pd_stream[dev] = nullptr;
        if (hipHostMalloc((void **)&g_pd_count[dev], 16) != hipSuccess) g_pd_count[dev] = nullptr;
    }
    return g_pd_stream[dev];
}
__global__ void __launch_bounds__(256) pool_poison_check_kernel(const uint32_t *p, size_t words, uint32_t poison, uint32_t *count) {
    uint32_t bad = 0, first = 0xFFFFFFFFu;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256)
        if (p[i] != poison) {
            bad++;
            if (first == 0xFFFFFFFFu) first = (uint32_t)(i > 0xFFFFFFFEull ? 0xFFFFFFFEu : i);
        }
    if (bad) {
        atomicAdd(&count[0], bad);
        atomicMin(&count[1], first);
    }
}
static void pd_poison(void *p, size_t bytes, int dev) {  // under g_pool_mu; complete on return
    hipStream_t st = pd_stream(dev);
    if (!st) return;
    (void)hipMemsetD32Async((hipDeviceptr_t)p, (int)POOL_POISON, bytes / 4, st);
    (void)hipStreamSynchronize(st);
    g_pd_poisoned_bytes += bytes;
}
static bool pd_verify(void *p, size_t bytes, int dev) {  // under g_pool_mu; true = every word is still poison
    hipStream_t st = pd_stream(dev);
    if (!st || !g_pd_count[dev]) return true;
    (void)hipDeviceSynchronize();  // whatever was in flight when the block was freed has landed now
    g_pd_count[dev][0] = 0;
    g_pd_count[dev][1] = 0xFFFFFFFFu;
    const size_t words = bytes / 4;
    const unsigned grid = (unsigned)(words / 256 / 16 < 1 ? 1 : (words / 256 / 16 > 4096 ? 4096 : words / 256 / 16));
    hipLaunchKernelGGL(pool_poison_check_kernel, dim3(grid), dim3(256), 0, st, (const uint32_t *)p, words, POOL_POISON, g_pd_count[dev]);
    (void)hipStreamSynchronize(st);
    g_pd_checked++;
    if (g_pd_count[dev][0] == 0) return true;
    g_pd_hits++;
    char msg[256];
    snprintf(msg, sizeof msg, "ZG_POOL_DEBUG: block %p (class %zu bytes, device %d) was written after it was freed: %u words are not poison, first at word %u",
             p, bytes, dev, g_pd_count[dev][0], g_pd_count[dev][1]);
    fprintf(stderr, "%s\n", msg);
    set_error(msg);
    if (pool_debug() >= 2) abort();
    return false;
}

void *pool_alloc(size_t bytes) {
    const int dev = current_device();
    const size_t cls = pool_class(bytes ? bytes : 1);
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        auto it = g_pool_idle.find({dev, cls});
        if (it != g_pool_idle.end()) {
            void *p = it->second;
            g_pool_idle.erase(it);
            g_pool_cached -= cls;
            if (pool_debug() && !pd_verify(p, cls, dev)) {
                pd_poison(p, cls, dev);  // refused: back among the idle blocks, clean again; the caller sees an allocation failure
                g_pool_idle.insert({{dev, cls}, p});
                g_pool_cached += cls;
                return nullptr;
            }
            g_pool_live[p] = PoolBlock{cls, dev};
            return p;
        }
    }
    void *p = nullptr;
    hipError_t e = dev_malloc(&p, cls);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error(std::string("hipMalloc: ") + hipGetErrorString(e));
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (pool_debug()) pd_poison(p, cls, dev);
    g_pool_live[p] = PoolBlock{cls, dev};
    return p;
}
void pool_free(void *p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = g_pool_live.find(p);
    if (it == g_pool_live.end()) {  // not ours (or the pool was shut down under its owner): plain free
        (void)hipFree(p);
        return;
    }
    const PoolBlock b = it->second;
    g_pool_live.erase(it);
    if (g_pool_cached + b.bytes > pool_cap()) {
        (void)hipFree(p);
        return;
    }
    if (pool_debug()) {
        // (read without g_mu: zg_shutdown takes g_mu before the pool's mutex; a stale pointer only costs this counter)
        hipStream_t ls = b.dev >= 0 && b.dev < ZG_MAX_DEVICES ? g_streams[b.dev] : nullptr;
        if (ls && hipStreamQuery(ls) == hipErrorNotReady) g_pd_suspect++;
        (void)hipGetLastError();
        int cur = current_device();
        if (cur != b.dev) (void)hipSetDevice(b.dev);
        pd_poison(p, b.bytes, b.dev);
        if (cur != b.dev && cur >= 0) (void)hipSetDevice(cur);
    }
    g_pool_idle.insert({{b.dev, b.bytes}, p});
    g_pool_cached += b.bytes;
}
// zg_shutdown: idle blocks are freed; blocks still held by a live session or Scratch are forgotten, so that their owner's later
// pool_free takes the plain hipFree path (round-4 advisor finding: the old scratch_trim freed buffers under their owners)
static void pool_shutdown() {
    pool_trim();
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool_live.clear();
}
void *scratch_get(size_t bytes) { return pool_alloc(bytes); }
void scratch_put(void *p) { pool_free(p); }

// Pinned staging buffers for the host-pointer entry points that move hundreds of MB (zg_fr_rows_from_columns): hipHostMalloc of 32 MB costs
// milliseconds, so the few buffers ever asked for are kept until zg_shutdown. Best fit among the idle ones, otherwise a new one.
struct PinBuf { void *p; size_t bytes; bool used; };
static std::mutex g_pin_mu;
static std::vector<PinBuf> g_pin;
void *pinned_get(size_t bytes) {
    if (bytes == 0) bytes = 16;
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        size_t best = (size_t)-1;
        for (size_t i = 0; i < g_pin.size(); i++)
            if (!g_pin[i].used && g_pin[i].bytes >= bytes && g_pin[i].bytes <= 4 * bytes + 4096 && (best == (size_t)-1 || g_pin[i].bytes < g_pin[best].bytes)) best = i;
        if (best != (size_t)-1) {
            g_pin[best].used = true;
            return g_pin[best].p;
        }
    }
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes) != hipSuccess) {
        (void)hipGetLastError();
        set_error("hipHostMalloc(staging): out of memory");
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(g_pin_mu);
    g_pin.push_back({p, bytes, true});
    return p;
}
void pinned_put(void *p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (auto &b : g_pin)
        if (b.p == p) {
            b.used = false;
            return;
        }
    (void)hipHostFree(p);
}
static void pinned_shutdown() {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    size_t keep = 0;
    for (size_t i = 0; i < g_pin.size(); i++) {
        if (g_pin[i].used) g_pin[keep++] = g_pin[i];  // still held: its owner's pinned_put finds it
        else (void)hipHostFree(g_pin[i].p);
    }
    g_pin.resize(keep);
}

// ------------------------------------------------------------------ set-up phase split
static thread_local SetupTimes t_setup;
SetupTimes &setup_times() { return t_setup; }
bool setup_times_enabled() {
    static const bool on = [] { const char *v = getenv("ZG_SETUP_TIMES"); return v && *v && *v != '0'; }();
    return on;
}
double now_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

// ------------------------------------------------------------------ profiling
struct ProfRec { int id; hipEvent_t e0, e1; };
static std::vector<ProfRec> g_prof;
static size_t g_prof_used = 0;
static std::atomic<bool> g_prof_on{false};
static std::mutex g_prof_mu;                            // record allocation; entry points may run on several host threads
static thread_local int t_prof_open[ZG_PROF_NKERNELS];  // record index + 1 of this thread's open bracket per kernel id (0 = none)

void prof_begin(int id, hipStream_t st) {
    t_prof_open[id] = 0;
    if (!g_prof_on.load(std::memory_order_acquire)) return;
    ProfRec *r = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (!g_prof_on.load(std::memory_order_relaxed) || g_prof_used >= g_prof.size()) return;
        r = &g_prof[g_prof_used];
        r->id = id;
        t_prof_open[id] = (int)++g_prof_used;
    }
    (void)hipEventRecord(r->e0, st);
}
void prof_end(int id, hipStream_t st) {
    int k = t_prof_open[id];
    if (k == 0 || !g_prof_on.load(std::memory_order_acquire)) return;
    t_prof_open[id] = 0;
    (void)hipEventRecord(g_prof[k - 1].e1, st);
}

// ------------------------------------------------------------------ field op kernel
// self-test hooks for the lazy 29-bit-limb arithmetic: results come back in canonical form
__global__ void __launch_bounds__(256) fp29_op_kernel(int op, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        F29 x = f29_from_fp(fe_load<FpParams>(a + 4 * i));
        F29 y = f29_from_fp(fe_load<FpParams>(b + 4 * i));
        F29 r;
        if (op == ZG_OP_MUL29) r = f29_mul(x, y);
        else if (op == ZG_OP_SQR29) r = f29_sqr(x);
        else {
            // ((x - y) [bias 2p] ... ) chain touching every biased subtraction and the zero test:
            //   t = x^2 + 5p - y - 2*(x*y);  u = (t + 7p - t') with t' = sub4(x, y);  r = u * (2p - y) , zeroed if y == 0 mod p
            F29 xy = f29_mul(x, y);
            F29 t = f29_x3(f29_sqr(x), y, xy);
            F29 t2 = f29_sub4(x, y);
            F29 u = f29_sub7(t, f29_mul(t2, t2));
            r = f29_mul(u, f29_neg2(y));
            r = f29_sub2(r, f29_mul(f29_times2(x), f29_times3(y)));
            if (f29_is_zero_modp(f29_sub2(y, f29_mul(y, f29_from_fp(Fp::one()))))) r = f29_sub4_2c(r, xy);
        }
        fe_store(out + 4 * i, f29_to_fp(r));
    }
}

template <class P>
__global__ void __launch_bounds__(256) field_op_kernel(int op, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        Fe<P> x = fe_load<P>(a + 4 * i);
        Fe<P> y = Fe<P>::zero();
        if (op <= ZG_OP_SUB) y = fe_load<P>(b + 4 * i);
        Fe<P> r;
        switch (op) {
            case ZG_OP_MUL: r = fe_mul(x, y); break;
            case ZG_OP_ADD: r = fe_add(x, y); break;
            case ZG_OP_SUB: r = fe_sub(x, y); break;
            case ZG_OP_NEG: r = fe_neg(x); break;
            case ZG_OP_SQR: r = fe_sqr(x); break;
            case ZG_OP_INV: r = fe_inv(x); break;
            case ZG_OP_FROM_MONT: r = fe_from_mont(x); break;
            case ZG_OP_INV_XGCD: r = fe_inv_fast(x); break;
            case ZG_OP_INV_SAFEGCD: r = fe_inv_safegcd(x); break;
            default: r = fe_to_mont(x); break;
        }
        fe_store(out + 4 * i, r);
    }
}

}  // namespace zg

using namespace zg;

extern "C" {

int zg_init(int device) { return do_init(device, 1); }

int zg_init_devices(int n_devices) {
    int count = zg_device_count();
    if (count > ZG_MAX_DEVICES) count = ZG_MAX_DEVICES;
    if (n_devices <= 0) n_devices = count;
    if (count == 0) {
        set_error("no HIP device available (libzolt_gpu has no CPU fallback)");
        return ZG_ERR_NO_DEVICE;
    }
    return do_init(0, n_devices);
}

int zg_n_devices(void) { return bound_devices(); }

void zg_shutdown(void) {
    sharded_shutdown();  // communicators and per-device exchange buffers (sharded.hip)
    sc_shutdown();       // pooled sumcheck sessions
    psc_shutdown();
    rwc_shutdown();
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_inited) return;
    int prev = current_device();
    for (int d = 0; d < ZG_MAX_DEVICES; d++) {
        if (!g_streams[d]) continue;
        (void)hipSetDevice(d);
        (void)hipStreamSynchronize(g_streams[d]);
        (void)hipStreamDestroy(g_streams[d]);
        g_streams[d] = nullptr;
    }
    for (int d = 0; d < ZG_MAX_DEVICES; d++) {
        for (hipStream_t st : g_idle_streams[d]) {
            (void)hipSetDevice(d);
            (void)hipStreamDestroy(st);
        }
        g_idle_streams[d].clear();
        for (const StreamGroup &g : g_idle_groups[d])
            for (int i = 0; i < 3; i++) {
                (void)hipSetDevice(d);
                (void)hipStreamDestroy(g.s[i]);
            }
        g_idle_groups[d].clear();
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    pool_shutdown();
    pinned_shutdown();
    g_primary = -1;
    g_ndev = 0;
    g_inited = false;
}

uint32_t zg_abi_version(void) { return ((uint32_t)ZG_ABI_MAJOR << 16) | (uint32_t)ZG_ABI_MINOR; }
uint32_t zg_abi_features(void) { return ZG_FEATURE_PROTOCOL_SESSIONS | ZG_FEATURE_RCCL | ZG_FEATURE_COLUMN_INGEST; }
const char *zg_last_error(void) { return t_err.c_str(); }
const char *zg_version(void) { return "zolt-gfx950 0.1 (BN254 G1 MSM / eq-table / sumcheck fold; gfx950 HIP)"; }

int zg_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}

int zg_dev_alloc(size_t bytes, void **dptr) {
    ZG_INIT();
    if (!dptr) return ZG_ERR_INVALID;
    *dptr = pool_alloc(bytes);
    return *dptr ? ZG_OK : ZG_ERR_NOMEM;
}
int zg_dev_free(void *dptr) {
    ZG_INIT();
    if (!dptr) return ZG_OK;
    int dev = -1;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        auto it = g_pool_live.find(dptr);
        if (it != g_pool_live.end()) dev = it->second.dev;
    }
    if (dev < 0) {  // not from zg_dev_alloc
        ZG_HIP(hipFree(dptr));
        return ZG_OK;
    }
    {  // hipFree's guarantee: nothing on the block's device still uses it when it is handed out again
        DeviceGuard dg(dev);
        ZG_HIP(hipDeviceSynchronize());
    }
    pool_free(dptr);
    return ZG_OK;
}
int zg_dev_trim(void) {
    ZG_INIT();
    pool_trim();
    return ZG_OK;
}
// Both copies run ON the library stream and wait for it: they are ordered after every call that was given stream = NULL (the
// asynchronous zg_fr_eq_table_dev in particular — a plain hipMemcpy is not ordered with a non-blocking stream, and a zero tail written
// "after" a table build could land before it).
int zg_memcpy_h2d(void *dst, const void *src, size_t bytes) {
    ZG_INIT();
    ZG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, lib_stream()));
    ZG_HIP(hipStreamSynchronize(lib_stream()));
    return ZG_OK;
}
int zg_dev_memset(void *dst, int byte_value, size_t bytes) {
    ZG_INIT();
    if (bytes == 0) return ZG_OK;
    ZG_HIP(hipMemsetAsync(dst, byte_value, bytes, lib_stream()));
    ZG_HIP(hipStreamSynchronize(lib_stream()));
    return ZG_OK;
}
int zg_memcpy_d2h(void *dst, const void *src, size_t bytes) {
    ZG_INIT();
    ZG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, lib_stream()));
    ZG_HIP(hipStreamSynchronize(lib_stream()));
    return ZG_OK;
}
int zg_sync(void) {
    ZG_INIT();
    ZG_HIP(hipStreamSynchronize(lib_stream()));
    return ZG_OK;
}

int zg_pool_debug_stats(uint64_t out[4]) {
    if (!out) return ZG_ERR_INVALID;
    out[0] = g_pd_hits.load();
    out[1] = g_pd_suspect.load();
    out[2] = g_pd_checked.load();
    out[3] = g_pd_poisoned_bytes.load();
    return pool_debug();
}

// Breaks the pool's contract on purpose: a block is freed while a kernel that will still write it is queued behind a slow one on the
// library's stream, and the same size class is asked for again. Returns 1 when the debug mode caught the write (the allocation was refused
// and counted), 0 when the block came back unchecked (ZG_POOL_DEBUG unset), a negative error code otherwise.
__global__ void pool_selftest_late_writer(uint32_t *p, size_t words, long long spin_cycles) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin_cycles) {}
    for (size_t i = threadIdx.x; i < words; i += blockDim.x) p[i] = 0x600DF00Du;
}
int zg_pool_debug_selftest(void) {
    ZG_INIT();
    if (pool_debug() >= 2) return ZG_ERR_INVALID;  // would abort by design
    if (pool_cap() == 0) return 0;                  // ZG_DEV_ALLOC_CACHE_MB=0: freed blocks go straight back to the driver, nothing is ever reused
    const size_t bytes = 3 * 4096 + 512;            // a class of its own in practice
    hipStream_t st = lib_stream();
    void *p = pool_alloc(bytes);
    if (!p) return -ZG_ERR_NOMEM;
    hipLaunchKernelGGL(pool_selftest_late_writer, dim3(1), dim3(64), 0, st, (uint32_t *)p, bytes / 4, (long long)2000000);  // ~20 ms at 100 MHz
    pool_free(p);  // the violation: the writer is still running
    const uint64_t before = g_pd_hits.load();
    void *q = pool_alloc(bytes);
    (void)hipStreamSynchronize(st);
    const bool caught = q == nullptr && g_pd_hits.load() == before + 1;
    if (q) pool_free(q);
    if (!caught && pool_debug()) return -ZG_ERR_HIP;
    if (pool_debug()) {  // the refused block sits clean among the idle ones: the pool stays usable after a hit
        void *r = pool_alloc(bytes);
        if (!r) return -ZG_ERR_NOMEM;
        pool_free(r);
        g_pd_hits--;  // the hit was the self-test's own
    }
    return caught ? 1 : 0;
}

int zg_last_setup_times(double out[4]) {
    if (!out) return ZG_ERR_INVALID;
    const SetupTimes &t = setup_times();
    out[0] = t.alloc_ms; out[1] = t.h2d_ms; out[2] = t.kernel_ms; out[3] = t.other_ms;
    return ZG_OK;
}

int zg_profile_begin(int max_records) {
    ZG_INIT();
    if (max_records < 1) return ZG_ERR_INVALID;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    while ((int)g_prof.size() < max_records) {
        ProfRec r;
        r.id = -1;
        ZG_HIP(hipEventCreate(&r.e0));
        ZG_HIP(hipEventCreate(&r.e1));
        g_prof.push_back(r);
    }
    g_prof_used = 0;
    g_prof_on = true;
    return ZG_OK;
}

int zg_profile_end(double ms_out[ZG_PROF_NKERNELS], uint64_t count_out[ZG_PROF_NKERNELS]) {
    ZG_INIT();
    if (!ms_out || !count_out) {
        set_error("zg_profile_end: invalid argument");
        return ZG_ERR_INVALID;
    }
    g_prof_on = false;
    for (int i = 0; i < ZG_PROF_NKERNELS; i++) { ms_out[i] = 0.0; count_out[i] = 0; }
    ZG_HIP(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (size_t k = 0; k < g_prof_used; k++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_prof[k].e0, g_prof[k].e1) == hipSuccess) {
            ms_out[g_prof[k].id] += ms;
            count_out[g_prof[k].id] += 1;
        }
    }
    g_prof_used = 0;
    return ZG_OK;
}

int zg_field_op(int field, int op, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n) {
    ZG_INIT();
    if (op < 0 || op > ZG_OP_INV_SAFEGCD || op == 8 /* retired */ || (op >= ZG_OP_MUL29 && op <= ZG_OP_X3_29 && field != ZG_FIELD_FP) || (field != ZG_FIELD_FR && field != ZG_FIELD_FP) || !a || !out ||
        ((op <= ZG_OP_SUB || (op >= ZG_OP_MUL29 && op <= ZG_OP_X3_29)) && !b)) {
        set_error("zg_field_op: invalid argument");
        return ZG_ERR_INVALID;
    }
    if (n == 0) return ZG_OK;
    size_t bytes = n * 32;
    bool two = op <= ZG_OP_SUB || (op >= ZG_OP_MUL29 && op <= ZG_OP_X3_29);
    hipStream_t st = lib_stream();
    Scratch sa(bytes), sout(bytes), sb;
    if (!sa.p || !sout.p || (two && !sb.alloc(bytes))) return ZG_ERR_NOMEM;
    SyncGuard sync(st);
    uint64_t *da = sa.as<uint64_t>(), *db = sb.as<uint64_t>(), *dout = sout.as<uint64_t>();
    ZG_HIP(hipMemcpyAsync(da, a, bytes, hipMemcpyHostToDevice, st));
    if (two) ZG_HIP(hipMemcpyAsync(db, b, bytes, hipMemcpyHostToDevice, st));
    unsigned blocks = div_up(n, 256);
    if (blocks > 4096) blocks = 4096;
    if (op >= ZG_OP_MUL29 && op <= ZG_OP_X3_29)
        hipLaunchKernelGGL(fp29_op_kernel, dim3(blocks), dim3(256), 0, st, op, da, db, dout, n);
    else if (field == ZG_FIELD_FR)
        hipLaunchKernelGGL(field_op_kernel<FrParams>, dim3(blocks), dim3(256), 0, st, op, da, db, dout, n);
    else
        hipLaunchKernelGGL(field_op_kernel<FpParams>, dim3(blocks), dim3(256), 0, st, op, da, db, dout, n);
    ZG_HIP(hipGetLastError());
    ZG_HIP(hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, st));
    ZG_HIP(hipStreamSynchronize(st));
    sync.dismiss();
    return ZG_OK;
}

}  // extern "C"
