// zolt_host.hpp — C++ host-side mirror of the reference's module API for the hot path, layered
// on the C ABI (include/zolt_gpu.h). The reference is Zig and no Zig toolchain exists in this
// image, so this header plays the role the patched Zig modules play in the real integration
// (INTEGRATION.md): same names, argument meaning and error behaviour as
//
//   zolt.msm.{AffinePoint, MSM, BatchMSM, ParallelMSM}          src/msm/mod.zig
//   zolt.poly.{DensePolynomial, EqPolynomial, UniPoly}          src/poly/mod.zig
//   zolt.poly.commitment.HyperKZG.{setup, commit, batchCommit}  src/poly/commitment/mod.zig
//   zolt.subprotocols.{Sumcheck, runSumcheck}                   src/subprotocols/mod.zig
//
// All heavy arithmetic runs in libzolt_gpu.so. The host keeps what the Zig host keeps: a few
// scalar Fr operations per sumcheck round for the toy verifier (the unchanged `field` module's
// job above the FFI seam) — implemented below with unsigned __int128 CIOS.
#pragma once
#define ZOLT_HOST_UMBRELLA 1
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <condition_variable>
#include <map>
#include <mutex>
#include <unordered_map>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/zolt_gpu.h"

// The mirror is split per family (round-3 review): each header below is written to be included from here, in this order,
// inside no namespace of its own choosing other than zolt — include zolt_host.hpp, not the parts.
namespace zolt {

struct GpuError : std::runtime_error {
    int code;
    GpuError(int c, const std::string &where) : std::runtime_error(where + ": " + zg_last_error()), code(c) {}
};
inline void check(int rc, const char *where) {
    if (rc != ZG_OK) throw GpuError(rc, where);
}

// raw device memory through the C ABI, released when the owner goes away (also when a constructor throws half-way)
struct DeviceMem {
    void *p = nullptr;
    DeviceMem() = default;
    explicit DeviceMem(size_t bytes) { alloc(bytes); }
    DeviceMem(const DeviceMem &) = delete;
    DeviceMem &operator=(const DeviceMem &) = delete;
    ~DeviceMem() { if (p) zg_dev_free(p); }
    void alloc(size_t bytes) { check(zg_dev_alloc(bytes ? bytes : 1, &p), "zg_dev_alloc"); }
    uint64_t *u64() const { return static_cast<uint64_t *>(p); }
};

}  // namespace zolt

#include "field.hpp"
#include "msm.hpp"
#include "poly.hpp"
#include "hyperkzg.hpp"
#include "sumcheck.hpp"
#include "product_provers.hpp"
#include "wire.hpp"
#include "witness.hpp"
#include "stage_provers.hpp"
#include "lasso.hpp"
