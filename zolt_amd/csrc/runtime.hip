// runtime.hip — lifecycle, error reporting, raw device memory and the elementwise field
// kernels of libzolt_gpu.so (C ABI: include/zolt_gpu.h).
#include <atomic>
#include <cstdlib>
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

// ------------------------------------------------------------------ scratch cache
struct ScratchBuf { void *p; size_t bytes; bool used; int dev; };
static std::mutex g_scratch_mu;
static std::vector<ScratchBuf> g_scratch;
static size_t g_scratch_total = 0;
static constexpr size_t SCRATCH_CAP = (size_t)4 << 30;  // cached bytes kept at most

void *scratch_get(size_t bytes) {
    if (bytes == 0) bytes = 16;
    const int dev = current_device();
    {
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        size_t best = (size_t)-1;
        for (size_t i = 0; i < g_scratch.size(); i++)
            if (!g_scratch[i].used && g_scratch[i].dev == dev && g_scratch[i].bytes >= bytes && g_scratch[i].bytes <= 4 * bytes + 4096 &&
                (best == (size_t)-1 || g_scratch[i].bytes < g_scratch[best].bytes))
                best = i;
        if (best != (size_t)-1) {
            g_scratch[best].used = true;
            return g_scratch[best].p;
        }
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        set_error(std::string("hipMalloc(scratch): ") + hipGetErrorString(e));
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    g_scratch.push_back({p, bytes, true, dev});
    g_scratch_total += bytes;
    return p;
}

void scratch_put(void *p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    for (size_t i = 0; i < g_scratch.size(); i++) {
        if (g_scratch[i].p == p) {
            if (g_scratch_total > SCRATCH_CAP) {  // over the cap: really free it
                g_scratch_total -= g_scratch[i].bytes;
                (void)hipFree(p);
                g_scratch.erase(g_scratch.begin() + i);
            } else {
                g_scratch[i].used = false;
            }
            return;
        }
    }
    (void)hipFree(p);
}

static void scratch_trim() {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    for (auto &b : g_scratch) (void)hipFree(b.p);
    g_scratch.clear();
    g_scratch_total = 0;
}

// ------------------------------------------------------------------ zg_dev_alloc / zg_dev_free
// A host that drives a prover through the C ABI allocates and frees its tables per proof (the C++ / Python mirrors: a few 32 MB tables per
// stage): hipMalloc + hipFree of those cost more than the rounds they serve. Freed blocks are kept per device in size classes (the request
// rounded up to an eighth of its leading power of two: at most 12.5 % slack) and handed out again. zg_dev_free keeps hipFree's guarantee —
// it returns after all device work has finished (hipDeviceSynchronize), so a block that comes back from the cache is idle. Blocks above
// 512 MiB bypass the cache; at most 2 GiB stay cached (ZG_DEV_ALLOC_CACHE_MB, 0 = off); an allocation that fails empties the cache and
// tries again.
struct DevBlock { size_t bytes; int dev; };
static std::mutex g_da_mu;
static std::unordered_map<void *, DevBlock> g_da_live;             // blocks handed out by zg_dev_alloc (class size, device)
static std::multimap<std::pair<int, size_t>, void *> g_da_free;    // (device, class size) -> idle block
static size_t g_da_cached = 0;
static constexpr size_t DA_MAX_BLOCK = (size_t)512 << 20;
static size_t da_cache_cap() {
    static const size_t cap = [] {
        const char *v = getenv("ZG_DEV_ALLOC_CACHE_MB");
        long mb = v && *v ? atol(v) : 2048;
        return (size_t)(mb < 0 ? 0 : mb) << 20;
    }();
    return cap;
}
static size_t da_class(size_t bytes) {
    if (bytes <= 4096) return 4096;
    int lg = 63 - __builtin_clzll((unsigned long long)bytes);  // 2^lg <= bytes
    const size_t step = (size_t)1 << (lg - 3);
    return (bytes + step - 1) / step * step;
}
static void da_trim_locked() {
    for (auto &kv : g_da_free) (void)hipFree(kv.second);
    g_da_free.clear();
    g_da_cached = 0;
}
static void da_trim() {
    std::lock_guard<std::mutex> lk(g_da_mu);
    da_trim_locked();
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
    scratch_trim();
    da_trim();
    g_primary = -1;
    g_ndev = 0;
    g_inited = false;
}

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
    const int dev = current_device();
    const size_t cls = da_class(bytes ? bytes : 1);
    const bool cached = cls <= DA_MAX_BLOCK && da_cache_cap() > 0;
    if (cached) {
        std::lock_guard<std::mutex> lk(g_da_mu);
        auto it = g_da_free.find({dev, cls});
        if (it != g_da_free.end()) {
            *dptr = it->second;
            g_da_free.erase(it);
            g_da_cached -= cls;
            g_da_live[*dptr] = DevBlock{cls, dev};
            return ZG_OK;
        }
    }
    hipError_t e = hipMalloc(dptr, cached ? cls : (bytes ? bytes : 1));
    if (e == hipErrorOutOfMemory) {  // give the cached blocks (this cache and the scratch cache) back and try once more
        (void)hipGetLastError();
        da_trim();
        e = hipMalloc(dptr, cached ? cls : (bytes ? bytes : 1));
    }
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        set_error("hipMalloc: out of memory");
        return ZG_ERR_NOMEM;
    }
    ZG_HIP(e);
    if (cached) {
        std::lock_guard<std::mutex> lk(g_da_mu);
        g_da_live[*dptr] = DevBlock{cls, dev};
    }
    return ZG_OK;
}
int zg_dev_free(void *dptr) {
    ZG_INIT();
    if (!dptr) return ZG_OK;
    DevBlock b{0, -1};
    {
        std::lock_guard<std::mutex> lk(g_da_mu);
        auto it = g_da_live.find(dptr);
        if (it != g_da_live.end()) {
            b = it->second;
            g_da_live.erase(it);
        }
    }
    if (b.dev < 0 || g_da_cached + b.bytes > da_cache_cap()) {  // not from the cache's classes, or the cache is full
        ZG_HIP(hipFree(dptr));
        return ZG_OK;
    }
    {  // hipFree's guarantee: nothing on the block's device still uses it when it is handed out again
        DeviceGuard dg(b.dev);
        ZG_HIP(hipDeviceSynchronize());
    }
    std::lock_guard<std::mutex> lk(g_da_mu);
    g_da_free.insert({{b.dev, b.bytes}, dptr});
    g_da_cached += b.bytes;
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
