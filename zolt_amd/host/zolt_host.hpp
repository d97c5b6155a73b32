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

// A pinned host buffer per thread for inputs that are assembled per call and uploaded at once (lookup indices, word vectors): memory
// from zg_host_alloc is neither zero-filled nor page-faulted on every use, and copies from it run at link rate whatever the page state
// of the process (a fresh pageable vector per call is pinned on the fly by the HIP runtime: usually fast, 16-24 ms now and then —
// profiles/r5s_h2d_pageable.txt). The buffer grows to the largest request and lives until the thread ends; valid until the next get().
struct PinnedStage {
    static void *get(size_t bytes) {
        struct Buf { void *p = nullptr; size_t cap = 0; ~Buf() { if (p) zg_host_free(p); } };
        static thread_local Buf b;
        if (b.cap < bytes) {
            if (b.p) zg_host_free(b.p);
            b.p = nullptr;
            b.cap = 0;
            check(zg_host_alloc(bytes + bytes / 4, &b.p), "zg_host_alloc");
            b.cap = bytes + bytes / 4;
        }
        return b.p;
    }
};

// A vector of machine words in pinned memory (a polynomial's evaluations before they are committed: commitBytecode / commitMemory /
// commitRegisters fill such a vector per proof, src/zkvm/mod.zig:1518-1617). The memory comes from a small per-thread stock of
// zg_host_alloc blocks and goes back to it: no pageable 8 MB vector is born, zero-filled, pinned on the fly by the HIP runtime and
// unpinned again per proof (what that costs and when it lands: INTEGRATION.md "Host memory").
class PinnedWords {
public:
    explicit PinnedWords(size_t n, bool zeroed = true) : n_(n) {
        Stock &st = stock();
        size_t best = (size_t)-1;
        for (size_t i = 0; i < st.free.size(); i++)
            if (st.free[i].cap >= n && (best == (size_t)-1 || st.free[i].cap < st.free[best].cap)) best = i;
        if (best != (size_t)-1) {
            blk_ = st.free[best];
            st.free.erase(st.free.begin() + best);
        } else {
            blk_.cap = n + n / 8 + 64;
            check(zg_host_alloc(blk_.cap * 8, &blk_.p), "zg_host_alloc");
        }
        if (zeroed && n) std::memset(blk_.p, 0, n * 8);
    }
    ~PinnedWords() {
        Stock &st = stock();
        if (st.free.size() < 6) st.free.push_back(blk_);
        else zg_host_free(blk_.p);
    }
    PinnedWords(const PinnedWords &) = delete;
    PinnedWords &operator=(const PinnedWords &) = delete;
    uint64_t *data() { return static_cast<uint64_t *>(blk_.p); }
    const uint64_t *data() const { return static_cast<const uint64_t *>(blk_.p); }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    uint64_t &operator[](size_t i) { return data()[i]; }
    const uint64_t &operator[](size_t i) const { return data()[i]; }

private:
    struct Block { void *p = nullptr; size_t cap = 0; };
    struct Stock {
        std::vector<Block> free;
        ~Stock() { for (auto &b : free) zg_host_free(b.p); }
    };
    static Stock &stock() { static thread_local Stock s; return s; }
    Block blk_;
    size_t n_;
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
