// common.hip.h — host-side plumbing shared by the translation units of libzolt_gpu.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <mutex>
#include <string>

#include "../../include/zolt_gpu_internal.h"

namespace zg {

static constexpr int ZG_MAX_DEVICES = 16;

void set_error(const std::string &msg);
int ensure_init();          // ZG_OK or ZG_ERR_NO_DEVICE / ZG_ERR_HIP
int primary_device();       // the device zg_init / zg_init_devices bound this process to (valid after ensure_init)
int current_device();       // the calling thread's HIP device
hipStream_t lib_stream();   // the library's own stream on the calling thread's CURRENT device (created on first use)
hipStream_t stream_acquire();                     // an idle stream of the current device (created if none): sumcheck sessions
void stream_release(hipStream_t st, int device);  // back to the free list (streams live until zg_shutdown)
// three streams created back to back (= on three different hardware queues), as a unit: launch sets that are meant to overlap
bool stream_group_acquire(hipStream_t out[3]);
void stream_group_release(const hipStream_t s[3], int device);

// HIP's current device is per host thread (a fresh std.Thread worker starts on device 0) and a handle's memory lives on
// the device it was created on: every entry point pins the calling thread to the right device for its duration.
struct DeviceGuard {
    int prev = -1, dev = -1;
    explicit DeviceGuard(int d) : dev(d) {
        if (d < 0) return;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != d) (void)hipSetDevice(d);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
    ~DeviceGuard() {
        if (dev >= 0 && prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
};

inline hipStream_t pick_stream(void *s) { return s ? reinterpret_cast<hipStream_t>(s) : lib_stream(); }

#define ZG_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            zg::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                    \
            return ZG_ERR_HIP;                                                                    \
        }                                                                                         \
    } while (0)

#define ZG_TRY(expr)                  \
    do {                              \
        int _r = (expr);              \
        if (_r != ZG_OK) return _r;   \
    } while (0)

// Multi-device code (sharded.hip) runs whole public entry points on another device: inside a DeviceScope the calling thread
// sits on `dev` AND primary_device() reports `dev`, so the ZG_INIT() of every nested entry point keeps it there.
void set_device_override(int dev);  // thread-local; -1 = none
int device_override();
struct DeviceScope {
    int prev_override;
    DeviceGuard guard;
    explicit DeviceScope(int d) : prev_override(device_override()), guard(d) { set_device_override(d); }
    ~DeviceScope() { set_device_override(prev_override); }
};

// every extern "C" compute entry point starts with this: library initialised, thread on the primary device
#define ZG_INIT()                \
    ZG_TRY(zg::ensure_init());   \
    zg::DeviceGuard _zg_primary_guard(zg::primary_device())

// a function attribute (dynamic LDS limit) belongs to the kernel's code object on ONE device: set it once per device
struct PerDeviceOnce {
    std::once_flag flag[ZG_MAX_DEVICES];
    hipError_t err[ZG_MAX_DEVICES];
    template <class F> hipError_t run(F f) {
        int d = current_device();
        if (d < 0 || d >= ZG_MAX_DEVICES) return hipErrorInvalidDevice;
        std::call_once(flag[d], [&] { err[d] = f(); });
        return err[d];
    }
};

// optional per-kernel HIP-event timing (bench.py's roofline leg); ids are ZG_PROF_*
void prof_begin(int id, hipStream_t st);
void prof_end(int id, hipStream_t st);
struct ProfScope {
    int id;
    hipStream_t st;
    ProfScope(int i, hipStream_t s) : id(i), st(s) { prof_begin(id, st); }
    ~ProfScope() { prof_end(id, st); }
};

// Cached device scratch for the host-pointer entry points: hipMalloc/hipFree per call cost more than the
// kernels they serve (hipFree also synchronises the device, which would stall other streams' work).
// A buffer is returned to the cache only after the work using it has been synchronised.
void *scratch_get(size_t bytes);  // nullptr on allocation failure (error set)
void scratch_put(void *p);
// the pool underneath (runtime.hip): size-classed blocks of device memory kept for reuse, for session tables as well as scratch.
// pool_free: the caller has synchronised the work that used the block. dev_malloc: hipMalloc for handle-lifetime allocations, with
// the pool trimmed and one retry when the device is out of memory.
void *pool_alloc(size_t bytes);
void pool_free(void *p);
void pool_trim();
hipError_t dev_malloc(void **p, size_t bytes);
// pinned host staging buffers, kept until zg_shutdown (nullptr + error on failure)
void *pinned_get(size_t bytes);
void pinned_put(void *p);
struct Scratch {
    void *p = nullptr;
    Scratch() = default;
    explicit Scratch(size_t bytes) : p(scratch_get(bytes)) {}
    Scratch(const Scratch &) = delete;
    Scratch &operator=(const Scratch &) = delete;
    ~Scratch() { if (p) scratch_put(p); }
    bool alloc(size_t bytes) { p = scratch_get(bytes); return p != nullptr; }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Declared AFTER the Scratch objects of a host-pointer entry point (so it is destroyed BEFORE them): an early error return
// then waits for the work already enqueued on `st` before the scratch buffers go back to the shared cache.
struct SyncGuard {
    hipStream_t st;
    bool armed = true;
    explicit SyncGuard(hipStream_t s) : st(s) {}
    SyncGuard(const SyncGuard &) = delete;
    SyncGuard &operator=(const SyncGuard &) = delete;
    ~SyncGuard() { if (armed) (void)hipStreamSynchronize(st); }
    void dismiss() { armed = false; }
};

// set-up phase split for the bench (include/zolt_gpu_internal.h: zg_last_setup_times); phases are only separated by synchronisations
// when ZG_SETUP_TIMES is set
struct SetupTimes { double alloc_ms = 0, h2d_ms = 0, kernel_ms = 0, other_ms = 0; };
SetupTimes &setup_times();   // the calling thread's record
bool setup_times_enabled();
double now_ms();

static inline unsigned div_up(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// r = a + b mod the BN254 scalar modulus on the host, canonical inputs (src/field/mod.zig:782-798)
static inline void fr_add_host(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
    static const uint64_t M[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
    uint64_t t[4];
    unsigned __int128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (unsigned __int128)a[i] + b[i];
        t[i] = (uint64_t)c;
        c >>= 64;
    }
    uint64_t carry = (uint64_t)c, d[4];
    unsigned __int128 br = 0;
    for (int i = 0; i < 4; i++) {
        unsigned __int128 x = (unsigned __int128)t[i] - M[i] - (uint64_t)br;
        d[i] = (uint64_t)x;
        br = (x >> 64) & 1;
    }
    bool ge = carry || !br;
    for (int i = 0; i < 4; i++) r[i] = ge ? d[i] : t[i];
}

int bound_devices();      // devices 0..n-1 bound by zg_init_devices (1 in the one-GPU-per-process model)
void sharded_shutdown();  // sharded.hip: drop communicators / exchange buffers (called by zg_shutdown)
void sc_shutdown();       // poly.hip / psc.hip: drop the pooled sumcheck sessions (called by zg_shutdown)
void psc_shutdown();
void rwc_shutdown();      // rwc.hip: free the pinned-buffer pool (called by zg_shutdown)

// msm.hip: zg_msm_g1_batch_dev that also fuses zero-padded rows on wide-window handles (HyperKZG.open's long levels)
// row_len (optional, k entries): row j holds zeros from row_len[j] on — the digit and sort kernels then walk the live entries only
int msm_batch_dev_wide(zg_bases_t b, size_t n, const uint64_t *d_scalars, size_t k, hipStream_t st, uint64_t *d_out9, const size_t *row_len = nullptr);
// msm.hip, for sharded.hip: k un-normalised Jacobian partials of this device's shard; the combine of gathered partials
int msm_batch_partials_dev(zg_bases_t b, size_t n, const uint64_t *d_scalars, size_t k, hipStream_t st, uint64_t *d_out12);
int msm_combine_batch_enqueue(const uint64_t *d_partials, size_t ranks, size_t rank_stride, size_t k, hipStream_t st, uint64_t *d_out9);
int bases_device(zg_bases_t b);
// ingest.hip: host columns widened into the cycle-major matrix at d_rows on st (synchronous: zg_fr_rows_from_columns, zg_sumcheck_open_column)
int rows_from_host_columns(const zg_col_t *cols, size_t n_cols, size_t n_rows, uint64_t *d_rows, hipStream_t st);
// ingest.hip: d_out[i] = F.fromU64(d_vals[i]) as canonical Montgomery elements, one launch on st (zg_msm_g1_u64's widening step)
int ingest_u64_to_fr(const uint64_t *d_vals, size_t n, uint64_t *d_out, hipStream_t st);
// poly.hip, for sharded.hip: enqueue a session's round-sums pass without waiting for its mailbox
int sc_round_sums_start(zg_sc_t s);
// poly.hip: out_host[i] = d_table[idx_host[i]] (32-byte elements; every index < len) through one gather launch on st, synchronous
int gather_to_host(const uint64_t *d_table, size_t len, const uint64_t *idx_host, size_t n, uint64_t *out_host, hipStream_t st);

}  // namespace zg
