// common.hip.h — host-side plumbing shared by the translation units of libzolt_gpu.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/zolt_gpu.h"

namespace zg {

void set_error(const std::string &msg);
int ensure_init();          // ZG_OK or ZG_ERR_NO_DEVICE / ZG_ERR_HIP
hipStream_t lib_stream();   // the library's own stream (valid after ensure_init)

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

#define ZG_INIT() ZG_TRY(zg::ensure_init())

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

static inline unsigned div_up(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// msm.hip: zg_msm_g1_batch_dev that also fuses zero-padded rows on wide-window handles (HyperKZG.open's long levels)
int msm_batch_dev_wide(zg_bases_t b, size_t n, const uint64_t *d_scalars, size_t k, hipStream_t st, uint64_t *d_out9);

}  // namespace zg
