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
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <unordered_map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/zolt_gpu.h"

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

// ---------------------------------------------------------------- host Fr (scalar use only)
struct Fr {
    uint64_t limbs[4];

    static constexpr uint64_t MOD[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    static constexpr uint64_t R[4] = {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL};
    static constexpr uint64_t R2[4] = {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL};
    static constexpr uint64_t INV = 0xc2e1f593efffffffULL;

    static Fr zero() { return Fr{{0, 0, 0, 0}}; }
    static Fr one() { return Fr{{R[0], R[1], R[2], R[3]}}; }
    bool isZero() const { return (limbs[0] | limbs[1] | limbs[2] | limbs[3]) == 0; }
    bool eql(const Fr &o) const { return std::memcmp(limbs, o.limbs, 32) == 0; }

    static bool geMod(const uint64_t *a) {
        for (int i = 3; i >= 0; i--) {
            if (a[i] < MOD[i]) return false;
            if (a[i] > MOD[i]) return true;
        }
        return true;
    }
    static void subMod(uint64_t *a) {
        unsigned __int128 borrow = 0;
        for (int i = 0; i < 4; i++) {
            unsigned __int128 d = (unsigned __int128)a[i] - MOD[i] - borrow;
            a[i] = (uint64_t)d;
            borrow = (d >> 64) & 1;
        }
    }
    Fr mul(const Fr &o) const {  // src/field/mod.zig:735-779
        uint64_t t[5] = {0, 0, 0, 0, 0};
        for (int i = 0; i < 4; i++) {
            uint64_t carry = 0;
            for (int j = 0; j < 4; j++) {
                unsigned __int128 s = (unsigned __int128)t[j] + (unsigned __int128)limbs[i] * o.limbs[j] + carry;
                t[j] = (uint64_t)s;
                carry = (uint64_t)(s >> 64);
            }
            t[4] += carry;
            uint64_t m = t[0] * INV;
            unsigned __int128 s0 = (unsigned __int128)t[0] + (unsigned __int128)m * MOD[0];
            carry = (uint64_t)(s0 >> 64);
            for (int j = 1; j < 4; j++) {
                unsigned __int128 s = (unsigned __int128)t[j] + (unsigned __int128)m * MOD[j] + carry;
                t[j - 1] = (uint64_t)s;
                carry = (uint64_t)(s >> 64);
            }
            unsigned __int128 fs = (unsigned __int128)t[4] + carry;
            t[3] = (uint64_t)fs;
            t[4] = (uint64_t)(fs >> 64);
        }
        Fr r{{t[0], t[1], t[2], t[3]}};
        if (t[4] != 0 || geMod(r.limbs)) subMod(r.limbs);
        return r;
    }
    Fr add(const Fr &o) const {  // :782-798
        Fr r;
        unsigned __int128 carry = 0;
        for (int i = 0; i < 4; i++) {
            unsigned __int128 s = (unsigned __int128)limbs[i] + o.limbs[i] + carry;
            r.limbs[i] = (uint64_t)s;
            carry = s >> 64;
        }
        if (carry || geMod(r.limbs)) subMod(r.limbs);
        return r;
    }
    Fr sub(const Fr &o) const {  // :801-816
        Fr r;
        unsigned __int128 borrow = 0;
        for (int i = 0; i < 4; i++) {
            unsigned __int128 d = (unsigned __int128)limbs[i] - o.limbs[i] - borrow;
            r.limbs[i] = (uint64_t)d;
            borrow = (d >> 64) & 1;
        }
        if (borrow) {
            unsigned __int128 carry = 0;
            for (int i = 0; i < 4; i++) {
                unsigned __int128 s = (unsigned __int128)r.limbs[i] + MOD[i] + carry;
                r.limbs[i] = (uint64_t)s;
                carry = s >> 64;
            }
        }
        return r;
    }
    bool inverse(Fr &out) const {  // :955-983 — Fermat, a^(p-2); false for zero (Zig: null)
        if (isZero()) return false;
        uint64_t e[4] = {MOD[0] - 2, MOD[1], MOD[2], MOD[3]};
        Fr result = one(), base = *this;
        for (int i = 0; i < 256; i++) {
            if ((e[i / 64] >> (i % 64)) & 1) result = result.mul(base);
            base = base.mul(base);
        }
        out = result;
        return true;
    }
    static Fr fromU64(uint64_t n) {  // :617-622
        Fr a{{n, 0, 0, 0}}, r2{{R2[0], R2[1], R2[2], R2[3]}};
        return a.mul(r2);
    }
    static Fr fromBytes(const uint8_t *bytes) {  // :625-639: 32 little-endian bytes (may exceed the modulus), times R^2
        Fr a, r2{{R2[0], R2[1], R2[2], R2[3]}};
        for (int i = 0; i < 4; i++) {
            uint64_t v = 0;
            for (int b = 7; b >= 0; b--) v = (v << 8) | bytes[8 * i + b];
            a.limbs[i] = v;
        }
        return a.mul(r2);
    }
};

// Fp values cross the host only as opaque limbs (coordinates of points)
struct Fp {
    uint64_t limbs[4];
    static constexpr uint64_t ONE[4] = {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL};
    static constexpr uint64_t TWO[4] = {0xa6ba871b8b1e1b3aULL, 0x14f1d651eb8e167bULL, 0xccdd46def0f28c58ULL, 0x1c14ef83340fbe5eULL};
};

// ---------------------------------------------------------------- msm
struct AffinePoint {  // src/msm/mod.zig:15-49
    Fp x, y;
    bool infinity;
    static AffinePoint identity() { return AffinePoint{{{0, 0, 0, 0}}, {{0, 0, 0, 0}}, true}; }
    static AffinePoint generator() {
        AffinePoint g;
        std::memcpy(g.x.limbs, Fp::ONE, 32);
        std::memcpy(g.y.limbs, Fp::TWO, 32);
        g.infinity = false;
        return g;
    }
    bool isIdentity() const { return infinity; }
    bool eql(const AffinePoint &o) const {
        if (infinity && o.infinity) return true;
        if (infinity || o.infinity) return false;
        return std::memcmp(x.limbs, o.x.limbs, 32) == 0 && std::memcmp(y.limbs, o.y.limbs, 32) == 0;
    }
    // add (:74-103) / double (:118-138): lambda formulas, one inversion; on the device (zg_g1_affine_add_batch)
    AffinePoint add(const AffinePoint &o) const {
        uint64_t a[8], b[8], out[8];
        uint8_t ai = infinity ? 1 : 0, bi = o.infinity ? 1 : 0, oi = 0;
        std::memcpy(a, x.limbs, 32); std::memcpy(a + 4, y.limbs, 32);
        std::memcpy(b, o.x.limbs, 32); std::memcpy(b + 4, o.y.limbs, 32);
        check(zg_g1_affine_add_batch(a, &ai, b, &bi, 1, out, &oi), "zg_g1_affine_add_batch");
        AffinePoint r;
        std::memcpy(r.x.limbs, out, 32); std::memcpy(r.y.limbs, out + 4, 32);
        r.infinity = oi != 0;
        return r;
    }
    AffinePoint dbl() const { return add(*this); }  // `double` is a C++ keyword
};

inline void pack_points(const std::vector<AffinePoint> &pts, std::vector<uint64_t> &xy, std::vector<uint8_t> &inf) {
    xy.resize(pts.size() * 8);
    inf.resize(pts.size());
    for (size_t i = 0; i < pts.size(); i++) {
        std::memcpy(&xy[8 * i], pts[i].x.limbs, 32);
        std::memcpy(&xy[8 * i + 4], pts[i].y.limbs, 32);
        inf[i] = pts[i].infinity ? 1 : 0;
    }
}
inline AffinePoint unpack_point(const uint64_t *xy, uint8_t inf) {
    AffinePoint p;
    std::memcpy(p.x.limbs, xy, 32);
    std::memcpy(p.y.limbs, xy + 4, 32);
    p.infinity = inf != 0;
    return p;
}

// device-resident bases: the GPU image of SetupParams.powers_of_tau_g1
class DeviceBases {
public:
    explicit DeviceBases(const std::vector<AffinePoint> &pts, const zg_msm_config *cfg = nullptr) : n_(pts.size()) {
        std::vector<uint64_t> xy;
        std::vector<uint8_t> inf;
        pack_points(pts, xy, inf);
        check(zg_g1_bases_upload(xy.data(), inf.data(), n_, cfg, &h_), "zg_g1_bases_upload");
    }
    ~DeviceBases() { zg_g1_bases_free(h_); }
    DeviceBases(const DeviceBases &) = delete;
    DeviceBases &operator=(const DeviceBases &) = delete;
    size_t len() const { return n_; }
    AffinePoint msm(const Fr *scalars, size_t n, size_t off = 0) const {
        uint64_t out[8];
        uint8_t inf = 0;
        check(zg_msm_g1(h_, off, n, reinterpret_cast<const uint64_t *>(scalars), out, &inf), "zg_msm_g1");
        return unpack_point(out, inf);
    }
    zg_bases_t handle() const { return h_; }

private:
    zg_bases_t h_ = nullptr;
    size_t n_;
};

struct MSM {  // MSM(Fr, Fp), src/msm/mod.zig:345-542
    // compute(bases, scalars) — :355-372. Lengths must match (std.debug.assert :359).
    static AffinePoint compute(const std::vector<AffinePoint> &bases, const std::vector<Fr> &scalars) {
        if (bases.size() != scalars.size()) throw std::invalid_argument("MSM.compute: bases.len != scalars.len");
        if (bases.empty()) return AffinePoint::identity();
        DeviceBases d(bases);
        return d.msm(scalars.data(), scalars.size());
    }
    // scalarMul(base, scalar).toAffine() — :503-540
    static AffinePoint scalarMul(const AffinePoint &base, const Fr &scalar) {
        uint64_t xy[8], out[8];
        uint8_t inf = base.infinity ? 1 : 0, oinf = 0;
        std::memcpy(xy, base.x.limbs, 32);
        std::memcpy(xy + 4, base.y.limbs, 32);
        check(zg_g1_scalar_mul_batch(xy, &inf, scalar.limbs, 1, out, &oinf), "zg_g1_scalar_mul_batch");
        return unpack_point(out, oinf);
    }
};

struct BatchMSM {  // :545-565 (ParallelBatchMSM :683-748 returns the same values)
    static std::vector<AffinePoint> compute(const std::vector<AffinePoint> &bases, const std::vector<std::vector<Fr>> &batches) {
        std::vector<AffinePoint> out;
        if (batches.empty()) return out;
        DeviceBases d(bases);
        for (const auto &b : batches) out.push_back(d.msm(b.data(), b.size()));
        return out;
    }
};

// Dory's data-parallel G1 / Fr pieces (src/poly/commitment/dory.zig; pairings and GT arithmetic stay the reference's)
struct Dory {
    // computeRowCommitments (:646-670): row r = MSM(g1_vec[0..len(row)], row r); full rows in one fused launch set, a shorter last row after
    static std::vector<AffinePoint> computeRowCommitments(const DeviceBases &g1_vec, const std::vector<Fr> &evals, size_t num_columns) {
        const size_t full = evals.size() / num_columns, rest = evals.size() % num_columns;
        std::vector<AffinePoint> out;
        if (full) {
            std::vector<const uint64_t *> ptrs;
            for (size_t r = 0; r < full; r++) ptrs.push_back(reinterpret_cast<const uint64_t *>(evals.data() + r * num_columns));
            std::vector<uint64_t> xy(8 * full);
            std::vector<uint8_t> inf(full);
            check(zg_msm_g1_batch(g1_vec.handle(), num_columns, ptrs.data(), full, xy.data(), inf.data()), "zg_msm_g1_batch");
            for (size_t r = 0; r < full; r++) out.push_back(unpack_point(xy.data() + 8 * r, inf[r]));
        }
        if (rest) out.push_back(g1_vec.msm(evals.data() + full * num_columns, rest));
        return out;
    }
    // multilinearLagrangeBasis (:544-588): the eq table with the index's LOW bit on point[0] = the device's eq table of the reversed point;
    // a shorter output is its first entries
    static std::vector<Fr> multilinearLagrangeBasis(const std::vector<Fr> &point, size_t out_len = 0) {
        std::vector<Fr> full(size_t(1) << point.size(), Fr::one());
        if (!point.empty()) {
            std::vector<Fr> rev(point.rbegin(), point.rend());
            check(zg_fr_eq_table(reinterpret_cast<const uint64_t *>(rev.data()), rev.size(), nullptr, reinterpret_cast<uint64_t *>(full.data())), "zg_fr_eq_table");
        }
        if (out_len && out_len < full.size()) full.resize(out_len);
        return full;
    }
    // computeEvaluationVectors (:590-620) -> (left_vec of 2^nu, right_vec of 2^sigma entries)
    static std::pair<std::vector<Fr>, std::vector<Fr>> computeEvaluationVectors(const std::vector<Fr> &point, unsigned nu, unsigned sigma) {
        std::vector<Fr> left(size_t(1) << nu, Fr::zero()), right(size_t(1) << sigma, Fr::zero());
        const size_t d = point.size();
        auto put = [](std::vector<Fr> &dst, const std::vector<Fr> &src) { std::copy(src.begin(), src.end(), dst.begin()); };
        if (d <= sigma) {
            put(right, multilinearLagrangeBasis(point));
            left[0] = Fr::one();
        } else {
            put(right, multilinearLagrangeBasis(std::vector<Fr>(point.begin(), point.begin() + sigma)));
            put(left, multilinearLagrangeBasis(std::vector<Fr>(point.begin() + sigma, point.end()), d <= nu + sigma ? 0 : left.size()));
        }
        return {left, right};
    }
    // computeVectorMatrixProduct (:622-642): v[col] = sum_row left_vec[row] * evals[row * 2^sigma + col]
    static std::vector<Fr> computeVectorMatrixProduct(const std::vector<Fr> &evals, const std::vector<Fr> &left_vec, unsigned nu, unsigned sigma) {
        const size_t rows = size_t(1) << nu, cols = size_t(1) << sigma;
        std::vector<Fr> m(rows * cols, Fr::zero()), w(rows, Fr::zero()), out(cols);
        std::copy(evals.begin(), evals.begin() + std::min(evals.size(), rows * cols), m.begin());
        std::copy(left_vec.begin(), left_vec.begin() + std::min(left_vec.size(), rows), w.begin());
        check(zg_fr_weighted_colsum(reinterpret_cast<const uint64_t *>(m.data()), rows, cols, reinterpret_cast<const uint64_t *>(w.data()), 1,
                                    reinterpret_cast<uint64_t *>(out.data())), "zg_fr_weighted_colsum");
        return out;
    }
};

// the SRS sharded over the devices bound by zg_init_devices (one resident table per GPU)
class ShardedDeviceBases {
public:
    explicit ShardedDeviceBases(const std::vector<AffinePoint> &pts, const zg_msm_config *cfg = nullptr) : n_(pts.size()) {
        std::vector<uint64_t> xy;
        std::vector<uint8_t> inf;
        pack_points(pts, xy, inf);
        check(zg_g1_bases_upload_sharded(xy.data(), inf.data(), n_, cfg, &h_), "zg_g1_bases_upload_sharded");
    }
    ~ShardedDeviceBases() { zg_g1_sbases_free(h_); }
    ShardedDeviceBases(const ShardedDeviceBases &) = delete;
    ShardedDeviceBases &operator=(const ShardedDeviceBases &) = delete;
    size_t len() const { return n_; }
    int shards() const { return zg_g1_sbases_shards(h_); }
    AffinePoint msm(const Fr *scalars, size_t n) const {
        uint64_t out[8];
        uint8_t inf = 0;
        check(zg_msm_g1_sharded(h_, n, reinterpret_cast<const uint64_t *>(scalars), out, &inf), "zg_msm_g1_sharded");
        return unpack_point(out, inf);
    }
    std::vector<AffinePoint> msmBatch(const std::vector<std::vector<Fr>> &batches, size_t n) const {
        std::vector<const uint64_t *> ptrs;
        for (const auto &b : batches) ptrs.push_back(reinterpret_cast<const uint64_t *>(b.data()));
        std::vector<uint64_t> xy(8 * batches.size());
        std::vector<uint8_t> inf(batches.size());
        check(zg_msm_g1_batch_sharded(h_, n, ptrs.data(), batches.size(), xy.data(), inf.data()), "zg_msm_g1_batch_sharded");
        std::vector<AffinePoint> out;
        for (size_t i = 0; i < batches.size(); i++) out.push_back(unpack_point(&xy[8 * i], inf[i]));
        return out;
    }

private:
    zg_sbases_t h_ = nullptr;
    size_t n_;
};

struct ParallelMSM {  // :572-680 — contiguous chunks of ceil(n / T), one partial per worker, serial combine: one worker = one GPU
    static AffinePoint compute(const std::vector<AffinePoint> &bases, const std::vector<Fr> &scalars, size_t /*num_threads*/) {
        if (bases.size() != scalars.size()) throw std::invalid_argument("ParallelMSM.compute: bases.len != scalars.len");
        if (bases.empty()) return AffinePoint::identity();
        zg_msm_config cfg{0, 0, 1};  // a one-shot slice: no precompute table
        ShardedDeviceBases d(bases, &cfg);
        return d.msm(scalars.data(), scalars.size());
    }
};

struct ParallelBatchMSM {  // :683-748 — k vectors, k partials per GPU, one exchange
    static std::vector<AffinePoint> compute(const std::vector<AffinePoint> &bases, const std::vector<std::vector<Fr>> &batches) {
        if (batches.empty()) return {};
        zg_msm_config cfg{0, 0, 1};
        ShardedDeviceBases d(bases, &cfg);
        return d.msmBatch(batches, batches[0].size());
    }
};

// ---------------------------------------------------------------- poly
struct UniPoly {  // src/poly/mod.zig:584-624
    std::vector<Fr> coeffs;
    Fr evaluate(const Fr &x) const {
        if (coeffs.empty()) return Fr::zero();
        Fr r = coeffs.back();
        for (size_t i = coeffs.size() - 1; i-- > 0;) r = r.mul(x).add(coeffs[i]);
        return r;
    }
};

struct DensePolynomial {  // src/poly/mod.zig:23-182
    std::vector<Fr> evaluations;
    size_t num_vars;
    explicit DensePolynomial(const std::vector<Fr> &evals) : evaluations(evals), num_vars(0) {
        size_t n = evals.size();
        if (n == 0 || (n & (n - 1))) throw std::invalid_argument("DensePolynomial.init: length must be a power of two");
        while ((size_t(1) << num_vars) < n) num_vars++;
    }
    size_t len() const { return evaluations.size(); }
    Fr evaluate(const std::vector<Fr> &point) const {  // :73-92, index bit j <-> point[j]
        if (point.size() != num_vars) throw std::invalid_argument("evaluate: point length != num_vars");
        Fr out;
        check(zg_fr_dense_evaluate(reinterpret_cast<const uint64_t *>(evaluations.data()), num_vars,
                                   reinterpret_cast<const uint64_t *>(point.data()), out.limbs), "zg_fr_dense_evaluate");
        return out;
    }
    DensePolynomial bindFirst(const Fr &value) const {  // :128-149
        if (num_vars == 0) throw std::invalid_argument("bindFirst: num_vars == 0");
        std::vector<Fr> out(evaluations.size() / 2);
        check(zg_fr_bind_high(reinterpret_cast<const uint64_t *>(evaluations.data()), evaluations.size(), value.limbs,
                              reinterpret_cast<uint64_t *>(out.data())), "zg_fr_bind_high");
        return DensePolynomial(out);
    }
    DensePolynomial add(const DensePolynomial &other) const {  // :94-110
        if (num_vars != other.num_vars) throw std::invalid_argument("add: num_vars differ");
        std::vector<Fr> out(evaluations.size());
        check(zg_field_op(ZG_FIELD_FR, ZG_OP_ADD, reinterpret_cast<const uint64_t *>(evaluations.data()),
                          reinterpret_cast<const uint64_t *>(other.evaluations.data()), reinterpret_cast<uint64_t *>(out.data()), out.size()),
              "zg_field_op");
        return DensePolynomial(out);
    }
    DensePolynomial scale(const Fr &scalar) const {  // :112-126
        std::vector<Fr> out(evaluations.size());
        check(zg_fr_scale(reinterpret_cast<const uint64_t *>(evaluations.data()), out.size(), scalar.limbs, reinterpret_cast<uint64_t *>(out.data())),
              "zg_fr_scale");
        return DensePolynomial(out);
    }
    void bindLow(const Fr &value) {  // :160-175, in place
        if (num_vars == 0) throw std::invalid_argument("bindLow: num_vars == 0");
        check(zg_fr_bind_low(reinterpret_cast<uint64_t *>(evaluations.data()), evaluations.size(), value.limbs), "zg_fr_bind_low");
        evaluations.resize(evaluations.size() / 2);
        num_vars -= 1;
    }
};

struct EqPolynomial {  // src/poly/mod.zig:190-323
    std::vector<Fr> r;
    explicit EqPolynomial(const std::vector<Fr> &point) : r(point) {}
    std::vector<Fr> evals() const { return evalsSliceWithScaling(r, nullptr); }
    // evaluate (:214-227) / mle (:311-321): prod_i (r_i x_i + (1 - r_i)(1 - x_i)) — host scalar code, v products
    Fr evaluate(const std::vector<Fr> &x) const { return mle(r, x); }
    static Fr mle(const std::vector<Fr> &r, const std::vector<Fr> &x) {
        if (r.size() != x.size()) throw std::invalid_argument("EqPolynomial.mle: r.len != x.len");
        Fr result = Fr::one();
        for (size_t i = 0; i < r.size(); i++) {
            Fr ri_xi = r[i].mul(x[i]);
            Fr one_minus_ri = Fr::one().sub(r[i]), one_minus_xi = Fr::one().sub(x[i]);
            result = result.mul(ri_xi.add(one_minus_ri.mul(one_minus_xi)));
        }
        return result;
    }
    static std::vector<Fr> evalsSliceWithScaling(const std::vector<Fr> &r, const Fr *scaling_factor) {  // :252-290
        std::vector<Fr> out(size_t(1) << r.size());
        check(zg_fr_eq_table(reinterpret_cast<const uint64_t *>(r.data()), r.size(), scaling_factor ? scaling_factor->limbs : nullptr,
                             reinterpret_cast<uint64_t *>(out.data())), "zg_fr_eq_table");
        return out;
    }
};

// GruenSplitEqPolynomial (src/poly/split_eq.zig:22-514): the prefix-table set comes from the device in one launch per half;
// bind / computeCubicRoundPoly are the reference's host scalar algebra.
struct GruenSplitEqPolynomial {
    size_t current_index = 0;
    Fr current_scalar = Fr::one();
    std::vector<Fr> tau;
    std::vector<std::vector<Fr>> E_out_vec, E_in_vec;
    size_t num_x_out = 0, num_x_in = 0;

    static std::vector<std::vector<Fr>> prefixTables(const Fr *w, size_t v) {  // :122-171, every level kept
        std::vector<Fr> flat((size_t(2) << v) - 1);
        check(zg_fr_eq_prefix_tables(reinterpret_cast<const uint64_t *>(w), v, reinterpret_cast<uint64_t *>(flat.data())), "zg_fr_eq_prefix_tables");
        std::vector<std::vector<Fr>> tabs(v + 1);
        for (size_t k = 0; k <= v; k++) tabs[k].assign(flat.begin() + ((size_t(1) << k) - 1), flat.begin() + ((size_t(2) << k) - 1));
        return tabs;
    }
    explicit GruenSplitEqPolynomial(const std::vector<Fr> &t, const Fr *scaling_factor = nullptr) : tau(t) {  // init / initWithScaling :51-183
        current_index = tau.size();
        if (scaling_factor) current_scalar = *scaling_factor;
        if (tau.empty()) return;
        size_t m = tau.size() / 2;
        num_x_out = m;
        num_x_in = tau.size() > 1 ? std::min(tau.size() - 1 - m, tau.size() - 1) : 0;
        E_out_vec = prefixTables(tau.data(), m);
        E_in_vec = prefixTables(tau.data() + m, num_x_in);
    }
    void bind(const Fr &r) {  // :213-248
        if (current_index == 0) return;
        const Fr &tau_i = tau[current_index - 1];
        Fr eq_val = tau_i.mul(r).add(Fr::one().sub(tau_i).mul(Fr::one().sub(r)));
        current_scalar = current_scalar.mul(eq_val);
        current_index -= 1;
        size_t m = tau.size() / 2;
        if (m < current_index) {
            if (E_in_vec.size() > 1) E_in_vec.pop_back();
        } else if (current_index > 0) {
            if (E_out_vec.size() > 1) E_out_vec.pop_back();
        }
    }
    std::vector<Fr> getFullEqTable() const {  // :254-285
        std::vector<Fr> head(tau.begin(), tau.begin() + current_index);
        return EqPolynomial::evalsSliceWithScaling(head, &current_scalar);
    }
    Fr getTauHigh() const { return tau.empty() ? Fr::zero() : tau.back(); }  // :291-294
    struct Window { const std::vector<Fr> *E_out, *E_in; size_t head_in_bits; };
    Window getWindowEqTables(size_t /* num_unbound_vars: ignored, as in the reference */, size_t window_size) const {  // :312-343
        size_t num_unbound = current_index, actual = std::min(window_size, num_unbound), head_len = num_unbound - actual;
        size_t m = tau.size() / 2, head_out_bits = std::min(head_len, m), head_in_bits = head_len - head_out_bits;
        const std::vector<Fr> &eo = head_out_bits < E_out_vec.size() ? E_out_vec[head_out_bits] : E_out_vec.back();
        const std::vector<Fr> &ei = head_in_bits < E_in_vec.size() ? E_in_vec[head_in_bits] : E_in_vec.back();
        return Window{&eo, &ei, head_in_bits};
    }
    std::array<Fr, 2> getCurrentEqFactors() const {  // :441-452
        if (current_index == 0) return {current_scalar, current_scalar};
        const Fr &tc = tau[current_index - 1];
        return {current_scalar.mul(Fr::one().sub(tc)), current_scalar.mul(tc)};
    }
    std::array<Fr, 4> computeCubicRoundPoly(const Fr &q_constant, const Fr &q_quadratic_coeff, const Fr &previous_claim) const {  // :353-434
        if (current_index == 0) return {previous_claim, Fr::zero(), Fr::zero(), Fr::zero()};
        auto f = getCurrentEqFactors();
        Fr l_slope = f[1].sub(f[0]);
        Fr l_2 = f[0].add(l_slope.mul(Fr::fromU64(2))), l_3 = f[0].add(l_slope.mul(Fr::fromU64(3)));
        Fr l0_q0 = f[0].mul(q_constant), inv, q_1 = Fr::zero();
        if (f[1].inverse(inv)) q_1 = previous_claim.sub(l0_q0).mul(inv);
        Fr e2 = q_quadratic_coeff.add(q_quadratic_coeff);
        Fr q_2 = q_1.add(q_1).sub(q_constant).add(e2);
        Fr q_3 = q_2.add(q_1).sub(q_constant).add(e2).add(e2);
        return {l0_q0, f[1].mul(q_1), l_2.mul(q_2), l_3.mul(q_3)};
    }
    std::vector<Fr> getEActiveForWindow(size_t window_size) const {  // :466-514
        if (window_size <= 1 || window_size > current_index) return {Fr::one()};
        size_t ws = current_index - window_size;
        std::vector<Fr> w(tau.begin() + ws, tau.begin() + ws + window_size - 1);
        return EqPolynomial::evalsSliceWithScaling(w, nullptr);
    }
};

// ---------------------------------------------------------------- HyperKZG (commit side)
struct HyperKZG {
    struct SetupParams {  // src/poly/commitment/mod.zig:122-140
        std::vector<AffinePoint> powers_of_tau_g1;
        AffinePoint g1;
        size_t max_degree;
        std::unique_ptr<DeviceBases> device;  // uploaded once, reused by every commit
    };
    struct Commitment {
        AffinePoint point;
        bool eql(const Commitment &o) const { return point.eql(o.point) && point.infinity == o.point.infinity; }
    };
    static SetupParams setup(size_t max_degree) {  // :174-213, tau = 0x12345678
        SetupParams p;
        p.g1 = AffinePoint::generator();
        p.max_degree = max_degree;
        std::vector<uint64_t> sc(max_degree * 4), out(max_degree * 8);
        std::vector<uint8_t> oinf(max_degree, 0);
        uint64_t g[8];
        std::memcpy(g, p.g1.x.limbs, 32);
        std::memcpy(g + 4, p.g1.y.limbs, 32);
        Fr tau = Fr::fromU64(0x12345678), tp = Fr::one();
        for (size_t i = 0; i < max_degree; i++) {
            std::memcpy(&sc[4 * i], tp.limbs, 32);
            tp = tp.mul(tau);
        }
        // :194-199: every product has the same base -> the fixed-base batch kernel
        check(zg_g1_fixed_base_mul_batch(g, 0, sc.data(), max_degree, out.data(), oinf.data()), "zg_g1_fixed_base_mul_batch");
        for (size_t i = 0; i < max_degree; i++) p.powers_of_tau_g1.push_back(unpack_point(&out[8 * i], oinf[i]));
        p.device.reset(new DeviceBases(p.powers_of_tau_g1));
        return p;
    }
    static Commitment commit(const SetupParams &params, const std::vector<Fr> &evals) {  // :239-255
        if (evals.empty()) return Commitment{AffinePoint::identity()};
        size_t n = evals.size() < params.powers_of_tau_g1.size() ? evals.size() : params.powers_of_tau_g1.size();
        return Commitment{params.device->msm(evals.data(), n)};
    }
    struct Proof {  // :155-167
        std::vector<Commitment> quotient_commitments;
        Fr final_eval;
    };
    // open(params, evals, point, value) — :261-324, the whole fold/commit loop stays on the device
    static Proof open(const SetupParams &params, const std::vector<Fr> &evals, const std::vector<Fr> &point, const Fr &value) {
        Proof pr;
        size_t v = point.size();
        std::vector<uint64_t> q(8 * v);
        std::vector<uint8_t> qi(v);
        check(zg_hyperkzg_open(params.device->handle(), reinterpret_cast<const uint64_t *>(evals.data()), evals.size(),
                               reinterpret_cast<const uint64_t *>(point.data()), v, value.limbs, q.data(), qi.data(), pr.final_eval.limbs),
              "zg_hyperkzg_open");
        for (size_t i = 0; i < v; i++) pr.quotient_commitments.push_back(Commitment{unpack_point(&q[8 * i], qi[i])});
        return pr;
    }
    struct BatchProof {  // :577-596
        std::vector<Commitment> quotient_commitments;
        std::vector<Fr> evaluations;
        Fr final_eval;
        Fr batching_challenge;
    };
    // batchOpen(params, polys, point) — :607-732: combination, evaluations and the fold/commit loop on the device
    static BatchProof batchOpen(const SetupParams &params, const std::vector<std::vector<Fr>> &polys, const std::vector<Fr> &point) {
        BatchProof pr;
        size_t k = polys.size(), v = point.size(), nq = 0;
        std::vector<const uint64_t *> ptrs(k ? k : 1, nullptr);
        std::vector<size_t> lens(k ? k : 1, 0);
        for (size_t i = 0; i < k; i++) {
            ptrs[i] = reinterpret_cast<const uint64_t *>(polys[i].data());
            lens[i] = polys[i].size();
        }
        std::vector<uint64_t> q(8 * (v ? v : 1)), ev(4 * (k ? k : 1));
        std::vector<uint8_t> qi(v ? v : 1);
        check(zg_hyperkzg_batch_open(params.device->handle(), ptrs.data(), lens.data(), k, reinterpret_cast<const uint64_t *>(point.data()), v,
                                     q.data(), qi.data(), &nq, ev.data(), pr.final_eval.limbs, pr.batching_challenge.limbs),
              "zg_hyperkzg_batch_open");
        for (size_t i = 0; i < nq; i++) pr.quotient_commitments.push_back(Commitment{unpack_point(&q[8 * i], qi[i])});
        for (size_t i = 0; i < k; i++) {
            Fr e;
            std::memcpy(e.limbs, &ev[4 * i], 32);
            pr.evaluations.push_back(e);
        }
        return pr;
    }
    static std::vector<Commitment> batchCommit(const SetupParams &params, const std::vector<std::vector<Fr>> &polys) {  // :558-570
        // polynomials of equal (clamped) length share one zg_msm_g1_batch call: short vectors are fused into one launch set
        std::vector<Commitment> out(polys.size(), Commitment{AffinePoint::identity()});
        std::vector<bool> done(polys.size(), false);
        size_t srs = params.powers_of_tau_g1.size();
        for (size_t i = 0; i < polys.size(); i++) {
            if (done[i]) continue;
            size_t n = polys[i].size() < srs ? polys[i].size() : srs;
            std::vector<size_t> idx;
            for (size_t j = i; j < polys.size(); j++)
                if (!done[j] && (polys[j].size() < srs ? polys[j].size() : srs) == n) idx.push_back(j);
            for (size_t j : idx) done[j] = true;
            if (n == 0 || idx.size() == 1) {
                for (size_t j : idx) out[j] = commit(params, polys[j]);
                continue;
            }
            std::vector<const uint64_t *> ptrs;
            for (size_t j : idx) ptrs.push_back(reinterpret_cast<const uint64_t *>(polys[j].data()));
            std::vector<uint64_t> xy(8 * idx.size());
            std::vector<uint8_t> inf(idx.size());
            check(zg_msm_g1_batch(params.device->handle(), n, ptrs.data(), idx.size(), xy.data(), inf.data()), "zg_msm_g1_batch");
            for (size_t t = 0; t < idx.size(); t++) out[idx[t]] = Commitment{unpack_point(&xy[8 * t], inf[t])};
        }
        return out;
    }
};

// ---------------------------------------------------------------- sumcheck
// ---------------------------------------------------------------- transcript (host side, between rounds)
// Transcript(F) — the reference's Keccak Fiat-Shamir transcript, src/transcripts/mod.zig:49-221: bytes XORed into a 200-byte
// state at `position`, Keccak-f[1600] every 136 bytes, challengeScalar = label, one Keccak-f, F.fromBytes(state[0..32]).
class Transcript {
public:
    explicit Transcript(const std::string &domain = "Jolt") { appendBytes(reinterpret_cast<const uint8_t *>(domain.data()), domain.size()); }
    void appendBytes(const uint8_t *data, size_t n) {  // :88-98
        for (size_t i = 0; i < n; i++) {
            state_[position_] ^= data[i];
            position_ += 1;
            if (position_ >= 136) {
                keccakF();
                position_ = 0;
            }
        }
    }
    void appendBytes(const std::string &s) { appendBytes(reinterpret_cast<const uint8_t *>(s.data()), s.size()); }
    void appendScalar(const std::string &label, const Fr &scalar) {  // :100-110: raw Montgomery limbs, little-endian
        appendBytes(label);
        uint8_t buf[32];
        for (int i = 0; i < 4; i++)
            for (int b = 0; b < 8; b++) buf[8 * i + b] = (uint8_t)(scalar.limbs[i] >> (8 * b));
        appendBytes(buf, 32);
    }
    Fr challengeScalar(const std::string &label) {  // :116-130
        appendBytes(label);
        keccakF();
        return Fr::fromBytes(state_);
    }
    const uint8_t *state() const { return state_; }

private:
    uint8_t state_[200] = {0};
    size_t position_ = 0;
    static uint64_t rotl(uint64_t x, unsigned n) { return (x << n) | (x >> (64 - n)); }
    void keccakF() {  // :163-213
        static const uint64_t RC[24] = {
            0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL, 0x0000000080000001ULL,
            0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
            0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL,
            0x000000000000800aULL, 0x800000008000000aULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
        static const unsigned ROTC[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
        static const unsigned PILN[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
        uint64_t st[25];
        for (int i = 0; i < 25; i++) {
            uint64_t v = 0;
            for (int b = 7; b >= 0; b--) v = (v << 8) | state_[8 * i + b];
            st[i] = v;
        }
        for (int round = 0; round < 24; round++) {
            uint64_t bc[5];
            for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
            for (int i = 0; i < 5; i++) {
                uint64_t t = bc[(i + 4) % 5] ^ rotl(bc[(i + 1) % 5], 1);
                for (int j = i; j < 25; j += 5) st[j] ^= t;
            }
            uint64_t t = st[1];
            for (int i = 0; i < 24; i++) {
                unsigned j = PILN[i];
                uint64_t tmp = st[j];
                st[j] = rotl(t, ROTC[i]);
                t = tmp;
            }
            for (int row = 0; row < 25; row += 5) {
                for (int i = 0; i < 5; i++) bc[i] = st[row + i];
                for (int i = 0; i < 5; i++) st[row + i] = bc[i] ^ (~bc[(i + 1) % 5] & bc[(i + 2) % 5]);
            }
            st[0] ^= RC[round];
        }
        for (int i = 0; i < 25; i++)
            for (int b = 0; b < 8; b++) state_[8 * i + b] = (uint8_t)(st[i] >> (8 * b));
    }
};

// Blake2b-256 (RFC 7693, unkeyed) for the Jolt-compatible transcript
inline void blake2b256(const uint8_t *in, size_t inlen, uint8_t out[32]) {
    static const uint64_t IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                   0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
    static const uint8_t SIGMA[12][16] = {
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
        {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
        {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
        {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
        {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
    uint64_t h[8];
    for (int i = 0; i < 8; i++) h[i] = IV[i];
    h[0] ^= 0x01010000ULL ^ 32ULL;
    auto rotr = [](uint64_t x, unsigned n) { return (x >> n) | (x << (64 - n)); };
    auto compress = [&](const uint8_t *block, uint64_t t, bool last) {
        uint64_t m[16], v[16];
        for (int i = 0; i < 16; i++) {
            uint64_t w = 0;
            for (int b = 7; b >= 0; b--) w = (w << 8) | block[8 * i + b];
            m[i] = w;
        }
        for (int i = 0; i < 8; i++) {
            v[i] = h[i];
            v[i + 8] = IV[i];
        }
        v[12] ^= t;
        if (last) v[14] = ~v[14];
        auto G = [&](int a, int b, int c, int d, uint64_t x, uint64_t y) {
            v[a] = v[a] + v[b] + x; v[d] = rotr(v[d] ^ v[a], 32);
            v[c] = v[c] + v[d];     v[b] = rotr(v[b] ^ v[c], 24);
            v[a] = v[a] + v[b] + y; v[d] = rotr(v[d] ^ v[a], 16);
            v[c] = v[c] + v[d];     v[b] = rotr(v[b] ^ v[c], 63);
        };
        for (int r = 0; r < 12; r++) {
            const uint8_t *sg = SIGMA[r];
            G(0, 4, 8, 12, m[sg[0]], m[sg[1]]);   G(1, 5, 9, 13, m[sg[2]], m[sg[3]]);
            G(2, 6, 10, 14, m[sg[4]], m[sg[5]]);  G(3, 7, 11, 15, m[sg[6]], m[sg[7]]);
            G(0, 5, 10, 15, m[sg[8]], m[sg[9]]);  G(1, 6, 11, 12, m[sg[10]], m[sg[11]]);
            G(2, 7, 8, 13, m[sg[12]], m[sg[13]]); G(3, 4, 9, 14, m[sg[14]], m[sg[15]]);
        }
        for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
    };
    size_t off = 0;
    while (inlen - off > 128) {
        compress(in + off, off + 128, false);
        off += 128;
    }
    uint8_t blk[128] = {0};
    std::memcpy(blk, in + off, inlen - off);
    compress(blk, inlen, true);
    for (int i = 0; i < 4; i++)
        for (int b = 0; b < 8; b++) out[8 * i + b] = (uint8_t)(h[i] >> (8 * b));
}

// Blake2bTranscript(F) — the Jolt-compatible transcript of the reference's proving path (src/transcripts/blake2b.zig:25-545): a 32-byte
// state and a round counter; every operation hashes state || [0u8; 28] || n_rounds_be32 || payload, the digest is the new state.
class Blake2bTranscript {
public:
    uint8_t state[32];
    uint32_t n_rounds = 0;
    explicit Blake2bTranscript(const std::string &label = "Jolt") {  // :39-69
        uint8_t padded[32] = {0};
        std::memcpy(padded, label.data(), label.size() < 32 ? label.size() : 32);
        blake2b256(padded, 32, state);
    }
    void appendMessage(const std::string &msg) {  // :96-120: right-padded to 32 bytes
        uint8_t padded[32] = {0};
        std::memcpy(padded, msg.data(), msg.size() < 32 ? msg.size() : 32);
        hashWith(padded, 32, nullptr);
    }
    void appendBytes(const uint8_t *data, size_t n) { hashWith(data, n, nullptr); }  // :123-156
    void appendU64(uint64_t x) {  // :160-176: [0u8; 24] ++ x.to_be_bytes()
        uint8_t buf[32] = {0};
        for (int b = 0; b < 8; b++) buf[24 + b] = (uint8_t)(x >> (8 * (7 - b)));
        hashWith(buf, 32, nullptr);
    }
    void appendScalar(const Fr &scalar) {  // :182-200: the canonical value, big-endian
        Fr one_raw{{1, 0, 0, 0}};
        Fr canon = scalar.mul(one_raw);  // fromMontgomery
        uint8_t buf[32];
        for (int i = 0; i < 4; i++)
            for (int b = 0; b < 8; b++) buf[31 - (8 * i + b)] = (uint8_t)(canon.limbs[i] >> (8 * b));
        hashWith(buf, 32, nullptr);
    }
    void challenge16(uint8_t out16[16]) {  // challengeBytes(16) (:215-240)
        uint8_t d[32];
        hashWith(nullptr, 0, d);
        std::memcpy(out16, d, 16);
    }
    Fr challengeScalarFull() {  // :279-312: the 16 bytes reversed, read little-endian = the digest prefix as a big-endian u128, to Montgomery
        uint8_t b[16];
        challenge16(b);
        uint64_t hi = 0, lo = 0;
        for (int i = 0; i < 8; i++) hi = (hi << 8) | b[i];
        for (int i = 8; i < 16; i++) lo = (lo << 8) | b[i];
        Fr raw{{lo, hi, 0, 0}}, r2{{Fr::R2[0], Fr::R2[1], Fr::R2[2], Fr::R2[3]}};
        return raw.mul(r2);
    }
    Fr challengeScalar() {  // :264-266,332-390: 125-bit mask, stored as RAW Montgomery limbs [0, 0, lo, hi] (MontU128Challenge)
        uint8_t b[16];
        challenge16(b);
        uint64_t hi = 0, lo = 0;  // the reversed buffer read big-endian = the digest prefix as a LITTLE-endian u128 (unlike challengeScalarFull)
        for (int i = 7; i >= 0; i--) lo = (lo << 8) | b[i];
        for (int i = 15; i >= 8; i--) hi = (hi << 8) | b[i];
        hi &= (1ULL << 61) - 1;
        return Fr{{0, 0, lo, hi}};
    }

private:
    void hashWith(const uint8_t *payload, size_t n, uint8_t *digest_out) {  // hasher() (:76-87) + payload, updateState (:90-93)
        std::vector<uint8_t> buf(64 + n, 0);
        std::memcpy(buf.data(), state, 32);
        buf[60] = (uint8_t)(n_rounds >> 24); buf[61] = (uint8_t)(n_rounds >> 16); buf[62] = (uint8_t)(n_rounds >> 8); buf[63] = (uint8_t)n_rounds;
        if (n) std::memcpy(buf.data() + 64, payload, n);
        blake2b256(buf.data(), buf.size(), state);
        n_rounds += 1;
        if (digest_out) std::memcpy(digest_out, state, 32);
    }
};

struct SumcheckVerificationFailed : std::runtime_error {
    SumcheckVerificationFailed() : std::runtime_error("SumcheckVerificationFailed") {}
};

struct Sumcheck {
    struct Round {
        UniPoly poly;
    };
    class Prover {  // src/subprotocols/mod.zig:50-134 — the polynomial lives on the GPU
    public:
        explicit Prover(const DensePolynomial &p) : round(0) {
            check(zg_sumcheck_open(reinterpret_cast<const uint64_t *>(p.evaluations.data()), p.evaluations.size(), ZG_SC_HIGH_HALF, &s_),
                  "zg_sumcheck_open");
        }
        ~Prover() { zg_sumcheck_close(s_); }
        Prover(const Prover &) = delete;
        Round nextRound() {  // :69-109 -> coefficients [g(0), g(1) - g(0)]
            Fr g0, g1;
            check(zg_sumcheck_round_sums(s_, g0.limbs, g1.limbs), "zg_sumcheck_round_sums");
            Round r;
            r.poly.coeffs = {g0, g1.sub(g0)};
            return r;
        }
        void receiveChallenge(const Fr &c) {  // :112-122
            check(zg_sumcheck_bind(s_, c.limbs), "zg_sumcheck_bind");
            round++;
        }
        bool isComplete() const { return zg_sumcheck_len(s_) == 1; }
        Fr getFinalEval() const {  // :130-133
            Fr f;
            check(zg_sumcheck_final(s_, f.limbs), "zg_sumcheck_final");
            return f;
        }
        size_t round;

    private:
        zg_sc_t s_ = nullptr;
    };
    struct Verifier {  // :137-244 (toy Fiat-Shamir mixer, host side as in the reference)
        Fr claim;
        size_t round = 0;
        std::vector<Fr> challenges;
        explicit Verifier(const Fr &c) : claim(c) {}
        Fr deriveChallenge(const Round &rd) const {  // :211-243
            uint64_t h = 0x9e3779b97f4a7c15ULL;
            h ^= (uint64_t)round;
            h *= 0xff51afd7ed558ccdULL;
            for (uint64_t limb : claim.limbs) { h ^= limb; h *= 0xc4ceb9fe1a85ec53ULL; }
            for (const Fr &c : rd.poly.coeffs)
                for (uint64_t limb : c.limbs) { h ^= limb; h *= 0xff51afd7ed558ccdULL; h ^= h >> 33; }
            h ^= h >> 33; h *= 0xff51afd7ed558ccdULL; h ^= h >> 33;
            return Fr::fromU64(h);
        }
        Fr verifyRound(const Round &rd) {  // :165-207
            Fr sum = rd.poly.evaluate(Fr::zero()).add(rd.poly.evaluate(Fr::one()));
            if (!sum.eql(claim)) throw SumcheckVerificationFailed();
            Fr ch = deriveChallenge(rd);
            challenges.push_back(ch);
            claim = rd.poly.evaluate(ch);
            round++;
            return ch;
        }
    };
    struct Proof {
        Fr claim;
        std::vector<Round> rounds;
        std::vector<Fr> final_point;
        Fr final_eval;
    };
};

struct SumcheckResult {
    Sumcheck::Proof proof;
    bool result;
};

// runSumcheck with the verifier on the host, one device round trip per round: the shape every prover with a real
// (Keccak/Blake2b) transcript has. Same outputs as runSumcheck below.
inline SumcheckResult runSumcheckInteractive(const DensePolynomial &polynomial) {  // src/subprotocols/mod.zig:302-354
    SumcheckResult out;
    Fr claim = Fr::zero();
    if (polynomial.num_vars == 0) {
        claim = polynomial.evaluations[0];
    } else {  // claim = sum of all evaluations (:306-309) = g0 + g1 of round 0
        zg_sc_t s = nullptr;
        check(zg_sumcheck_open(reinterpret_cast<const uint64_t *>(polynomial.evaluations.data()), polynomial.evaluations.size(),
                               ZG_SC_HIGH_HALF, &s), "zg_sumcheck_open");
        Fr g0, g1;
        int rc = zg_sumcheck_round_sums(s, g0.limbs, g1.limbs);
        zg_sumcheck_close(s);
        check(rc, "zg_sumcheck_round_sums");
        claim = g0.add(g1);
    }
    Sumcheck::Prover prover(polynomial);
    Sumcheck::Verifier verifier(claim);
    for (size_t i = 0; i < polynomial.num_vars; i++) {
        Sumcheck::Round rd = prover.nextRound();
        Fr ch = verifier.verifyRound(rd);
        prover.receiveChallenge(ch);
        out.proof.rounds.push_back(rd);
    }
    out.proof.claim = claim;
    out.proof.final_point = verifier.challenges;
    out.proof.final_eval = prover.getFinalEval();
    out.result = verifier.claim.eql(out.proof.final_eval);
    return out;
}

// runSumcheck (src/subprotocols/mod.zig:302-354): prover AND toy verifier on the device (zg_run_sumcheck), no PCIe
// crossing between rounds.
inline SumcheckResult runSumcheck(const DensePolynomial &polynomial) {
    SumcheckResult out;
    size_t v = polynomial.num_vars;
    std::vector<uint64_t> rounds(8 * v + 1), chal(4 * v + 1);
    uint8_t result = 0;
    int rc = zg_run_sumcheck(reinterpret_cast<const uint64_t *>(polynomial.evaluations.data()), polynomial.evaluations.size(),
                             out.proof.claim.limbs, rounds.data(), chal.data(), out.proof.final_eval.limbs, &result);
    if (rc == ZG_ERR_VERIFY) throw SumcheckVerificationFailed();
    check(rc, "zg_run_sumcheck");
    for (size_t i = 0; i < v; i++) {
        Sumcheck::Round rd;
        Fr c0, c1, ch;
        std::memcpy(c0.limbs, &rounds[8 * i], 32);
        std::memcpy(c1.limbs, &rounds[8 * i + 4], 32);
        std::memcpy(ch.limbs, &chal[4 * i], 32);
        rd.poly.coeffs = {c0, c1};
        out.proof.rounds.push_back(rd);
        out.proof.final_point.push_back(ch);
    }
    out.result = result != 0;
    return out;
}

// ---------------------------------------------------------------- product-form provers (zg_psc_*)
// Lagrange interpolation through evals at 0,1,2,3 evaluated at x — the claim update of every cubic prover
// (val_evaluation.zig:630-660, instruction_lookups.zig:250-270, product_remainder.zig:534-559)
inline Fr cubicAtPoint(const std::array<Fr, 4> &evals, const Fr &x) {
    Fr x1 = x.sub(Fr::one()), x2 = x.sub(Fr::fromU64(2)), x3 = x.sub(Fr::fromU64(3));
    // the four constant inverses are computed once (a Fermat inversion is ~380 products: four per call would dominate a round)
    static const std::array<Fr, 4> inv = [] {
        std::array<Fr, 4> r;
        Fr::zero().sub(Fr::fromU64(6)).inverse(r[0]);
        Fr::fromU64(2).inverse(r[1]);
        Fr::zero().sub(Fr::fromU64(2)).inverse(r[2]);
        Fr::fromU64(6).inverse(r[3]);
        return r;
    }();
    const Fr &i6n = inv[0], &i2 = inv[1], &i2n = inv[2], &i6 = inv[3];
    Fr L0 = x1.mul(x2).mul(x3).mul(i6n), L1 = x.mul(x2).mul(x3).mul(i2), L2 = x.mul(x1).mul(x3).mul(i2n), L3 = x.mul(x1).mul(x2).mul(i6);
    return evals[0].mul(L0).add(evals[1].mul(L1)).add(evals[2].mul(L2)).add(evals[3].mul(L3));
}
// UniPoly.interpolateDegree3 / evalsToCompressed (src/poly/mod.zig:632-685)
inline std::array<Fr, 4> interpolateDegree3(const std::array<Fr, 4> &p) {
    static const std::array<Fr, 2> inv = [] {
        std::array<Fr, 2> r;
        Fr::fromU64(6).inverse(r[0]);
        Fr::fromU64(2).inverse(r[1]);
        return r;
    }();
    const Fr &inv6 = inv[0], &inv2 = inv[1];
    Fr c1 = Fr::zero().sub(Fr::fromU64(11).mul(p[0])).add(Fr::fromU64(18).mul(p[1])).sub(Fr::fromU64(9).mul(p[2])).add(Fr::fromU64(2).mul(p[3])).mul(inv6);
    Fr c2 = Fr::fromU64(2).mul(p[0]).sub(Fr::fromU64(5).mul(p[1])).add(Fr::fromU64(4).mul(p[2])).sub(p[3]).mul(inv2);
    Fr c3 = Fr::zero().sub(p[0]).add(Fr::fromU64(3).mul(p[1])).sub(Fr::fromU64(3).mul(p[2])).add(p[3]).mul(inv6);
    return {p[0], c1, c2, c3};
}
inline std::array<Fr, 3> evalsToCompressed(const std::array<Fr, 4> &evals) {
    auto c = interpolateDegree3(evals);
    return {c[0], c[2], c[3]};
}

// k tables folded together in one device session
class ProductSumcheckSession {
public:
    explicit ProductSumcheckSession(const std::vector<const std::vector<Fr> *> &tables) {
        std::vector<const uint64_t *> ptrs;
        for (auto *t : tables) ptrs.push_back(reinterpret_cast<const uint64_t *>(t->data()));
        check(zg_psc_open(ptrs.data(), ptrs.size(), tables.empty() ? 0 : tables[0]->size(), &s_), "zg_psc_open");
    }
    struct OnDevice {};  // tables already in HBM (the session copies them): zg_psc_open_dev
    ProductSumcheckSession(OnDevice, const std::vector<const uint64_t *> &d_tables, size_t n) {
        check(zg_psc_open_dev(d_tables.data(), d_tables.size(), n, nullptr, &s_), "zg_psc_open_dev");
    }
    ~ProductSumcheckSession() { zg_psc_close(s_); }
    ProductSumcheckSession(const ProductSumcheckSession &) = delete;
    size_t len() const { return zg_psc_len(s_); }
    std::array<Fr, 4> roundEvals(const std::vector<int> &prod, const std::vector<int> &lin = {}, const std::vector<Fr> &coeff = {}) {
        std::array<Fr, 4> out;
        check(zg_psc_round_evals(s_, prod.data(), prod.size(), lin.data(), reinterpret_cast<const uint64_t *>(coeff.data()), lin.size(),
                                 reinterpret_cast<uint64_t *>(out.data())), "zg_psc_round_evals");
        return out;
    }
    std::array<Fr, 2> roundGruen(const std::vector<int> &prod, const uint64_t *d_e_out, size_t n_out, const uint64_t *d_e_in, size_t n_in) {
        std::array<Fr, 2> out;
        check(zg_psc_round_gruen(s_, prod.data(), prod.size(), d_e_out, n_out, d_e_in, n_in, out[0].limbs, out[1].limbs), "zg_psc_round_gruen");
        return out;
    }
    // a SUM of product terms in one pass (zg_psc_round_expr)
    struct Term {
        std::vector<int> prod, lin;
        std::vector<Fr> coeff;
        bool pair_sum = false;  // ZG_PSC_PAIR_SUM: (T[prod0] T[prod1] + T[prod2] T[prod3]) * L
    };
    std::array<Fr, 4> roundExpr(const std::vector<Term> &terms) {
        std::vector<zg_psc_term> t(terms.size());
        for (size_t i = 0; i < terms.size(); i++) {
            std::memset(&t[i], 0, sizeof(zg_psc_term));
            t[i].n_prod = (int)terms[i].prod.size() | (terms[i].pair_sum ? ZG_PSC_PAIR_SUM : 0);
            t[i].n_lin = (int)terms[i].lin.size();
            for (size_t j = 0; j < terms[i].prod.size() && j < 4; j++) t[i].prod[j] = terms[i].prod[j];
            for (size_t m = 0; m < terms[i].lin.size() && m < 4; m++) {
                t[i].lin[m] = terms[i].lin[m];
                std::memcpy(&t[i].lin_coeff[4 * m], terms[i].coeff[m].limbs, 32);
            }
        }
        std::array<Fr, 4> out;
        check(zg_psc_round_expr(s_, t.data(), t.size(), reinterpret_cast<uint64_t *>(out.data())), "zg_psc_round_expr");
        return out;
    }
    // bit t of `points`: the round calls compute p(t); the other slots come back as zero
    void setPoints(unsigned points) { check(zg_psc_set_points(s_, points), "zg_psc_set_points"); }
    void bind(const Fr &r) { check(zg_psc_bind(s_, r.limbs), "zg_psc_bind"); }
    std::vector<Fr> read(size_t table) {  // the whole current table
        std::vector<Fr> out(len());
        check(zg_psc_read(s_, table, reinterpret_cast<uint64_t *>(out.data())), "zg_psc_read");
        return out;
    }
    std::vector<Fr> gather(size_t table, const std::vector<uint64_t> &idx) {  // T[table][idx[i]] of the current tables
        std::vector<Fr> out(idx.size());
        check(zg_psc_gather(s_, table, idx.data(), idx.size(), reinterpret_cast<uint64_t *>(out.data())), "zg_psc_gather");
        return out;
    }
    const uint64_t *tableDev(size_t table) {  // where the folded table lies in HBM, pending folds completed (zg_psc_table_dev)
        const uint64_t *p = nullptr;
        check(zg_psc_table_dev(s_, table, &p), "zg_psc_table_dev");
        return p;
    }
    std::vector<Fr> final() {
        std::vector<Fr> out(zg_psc_tables(s_));
        check(zg_psc_final(s_, reinterpret_cast<uint64_t *>(out.data())), "zg_psc_final");
        return out;
    }

private:
    zg_psc_t s_ = nullptr;
};

// ValEvaluationProver's loop (src/zkvm/ram/val_evaluation.zig:545-700); lt == nullptr: ValFinalProver (ram/val_final.zig:144-230)
class ValEvaluationProver {
public:
    Fr current_claim;
    size_t round = 0;
    ValEvaluationProver(const std::vector<Fr> &inc, const std::vector<Fr> &wa, const std::vector<Fr> *lt, const Fr &claim)
        : current_claim(claim), s_(lt ? std::vector<const std::vector<Fr> *>{&inc, &wa, lt} : std::vector<const std::vector<Fr> *>{&inc, &wa}),
          factors_(lt ? std::vector<int>{0, 1, 2} : std::vector<int>{0, 1}) {}
    // three tables already in HBM (n entries each; the session copies them)
    ValEvaluationProver(ProductSumcheckSession::OnDevice, const uint64_t *d_inc, const uint64_t *d_wa, const uint64_t *d_lt, size_t n, const Fr &claim)
        : current_claim(claim), s_(ProductSumcheckSession::OnDevice{}, {d_inc, d_wa, d_lt}, n), factors_{0, 1, 2} {}
    std::array<Fr, 4> computeRoundPolynomial() {  // :554-603
        if (s_.len() < 2) {
            Fr acc = Fr::one();
            for (const Fr &v : s_.final()) acc = acc.mul(v);
            return {acc, Fr::zero(), Fr::zero(), Fr::zero()};
        }
        return s_.roundEvals(factors_);
    }
    void bindChallengeWithPoly(const Fr &r, const std::array<Fr, 4> &round_poly) {  // :609-660
        if (s_.len() >= 2) {
            s_.bind(r);
            current_claim = cubicAtPoint(round_poly, r);
        }
        round++;
    }
    std::vector<Fr> getFinalClaims() { return s_.final(); }

private:
    ProductSumcheckSession s_;
    std::vector<int> factors_;
};

// ProductVirtualRemainderProver's loop (src/zkvm/spartan/product_remainder.zig:269-394): Gruen's (t0, t_inf) on the device under
// split-eq prefix tables resident in HBM, the cubic on the host
class ProductVirtualRemainderProver {
public:
    Fr current_claim;
    size_t current_round = 0;
    GruenSplitEqPolynomial split_eq;
    ProductVirtualRemainderProver(const std::vector<Fr> &left, const std::vector<Fr> &right, const std::vector<Fr> &tau_low, const Fr &lagrange_kernel,
                                  const Fr &uni_skip_claim)
        : current_claim(uni_skip_claim), split_eq(tau_low, &lagrange_kernel), s_({&left, &right}) {
        size_t m = tau_low.size() / 2;
        d_out_.alloc(((size_t(2) << m) - 1) * 32);
        d_in_.alloc(((size_t(2) << split_eq.num_x_in) - 1) * 32);
        check(zg_fr_eq_prefix_tables_dev(reinterpret_cast<const uint64_t *>(tau_low.data()), m, d_out_.u64(), nullptr), "prefix");
        check(zg_fr_eq_prefix_tables_dev(reinterpret_cast<const uint64_t *>(tau_low.data() + m), split_eq.num_x_in, d_in_.u64(), nullptr), "prefix");
        check(zg_sync(), "zg_sync");  // the session reads the tables on its own stream
    }
    bool roundEvals(std::array<Fr, 4> &evals) {
        if (s_.len() < 2) return false;
        auto w = split_eq.getWindowEqTables(current_round, 1);  // sizes; the same tables sit at element 2^k - 1 of the device buffers
        size_t n_out = w.E_out->size(), n_in = w.E_in->size();
        auto t = s_.roundGruen({0, 1}, d_out_.u64() + 4 * (n_out - 1), n_out, d_in_.u64() + 4 * (n_in - 1), n_in);
        evals = split_eq.computeCubicRoundPoly(t[0], t[1], current_claim);
        return true;
    }
    std::array<Fr, 3> computeRoundPolynomial() {  // compressed [c0, c2, c3]; [claim, 0, 0] without groups (:274-276)
        std::array<Fr, 4> ev;
        if (!roundEvals(ev)) return {current_claim, Fr::zero(), Fr::zero()};
        return evalsToCompressed(ev);
    }
    void bindChallenge(const Fr &challenge) {
        s_.bind(challenge);
        split_eq.bind(challenge);
        current_round++;
    }
    void updateClaim(const std::array<Fr, 4> &round_evals, const Fr &challenge) { current_claim = cubicAtPoint(round_evals, challenge); }
    Fr getFinalClaim() {
        auto f = s_.final();
        return f[0].mul(f[1]);
    }

private:
    ProductSumcheckSession s_;
    DeviceMem d_out_, d_in_;
};

// InstructionInputProver's loop (src/zkvm/spartan/stage3_prover.zig:2029-2150): tables left_is_rs1, rs1_value, left_is_pc, unexpanded_pc,
// right_is_rs2, rs2_value, right_is_imm, imm, eq_outer, eq_product; f = (eq_outer + g^2 eq_product) * (is_rs2*rs2 + is_imm*imm +
// g (is_rs1*rs1 + is_pc*pc)) as four product terms of one multi-term round
class InstructionInputProver {
public:
    InstructionInputProver(const std::vector<const std::vector<Fr> *> &tables, const Fr &gamma) : s_(tables) {
        Fr g2 = gamma.mul(gamma);
        std::vector<Fr> w_right = {Fr::one(), g2}, w_left = {gamma, g2.mul(gamma)};
        terms_ = {{{4, 5, 6, 7}, {8, 9}, w_right, true}, {{0, 1, 2, 3}, {8, 9}, w_left, true}};  // two pair-sum terms
        s_.setPoints(0b1101);  // p(1) comes from the claim
    }
    std::array<Fr, 4> computeRoundEvals(const Fr &previous_claim) {  // [p(0), claim - p(0), p(2), p(3)] (:2029-2100)
        auto ev = s_.roundExpr(terms_);
        return {ev[0], previous_claim.sub(ev[0]), ev[2], ev[3]};
    }
    void bind(const Fr &r_j) { s_.bind(r_j); }
    std::vector<Fr> finalClaims() { return s_.final(); }

private:
    ProductSumcheckSession s_;
    std::vector<ProductSumcheckSession::Term> terms_;
};

// R1CSInputEvaluator.computeClaimedInputs (src/zkvm/r1cs/evaluation.zig:55-122): witness = cycle-major matrix, k values per cycle
inline std::vector<Fr> computeClaimedInputs(const std::vector<Fr> &cycle_witnesses, size_t k, const std::vector<Fr> &r_cycle) {
    size_t num_cycles = k ? cycle_witnesses.size() / k : 0;
    std::vector<Fr> out(k, Fr::zero());
    if (num_cycles == 0) return out;
    size_t log_n = 0;
    while ((size_t(2) << log_n) <= num_cycles) log_n++;
    size_t padded_len = size_t(1) << log_n, effective_len = std::min(r_cycle.size(), log_n);
    if (effective_len == 0) {  // :75-83
        for (size_t i = 0; i < k; i++) out[i] = cycle_witnesses[i];
        return out;
    }
    if (effective_len < log_n) throw std::out_of_range("computeClaimedInputs: r_cycle shorter than log2 of the cycle count");
    check(zg_fr_rows_mle(reinterpret_cast<const uint64_t *>(cycle_witnesses.data()), std::min(num_cycles, padded_len), k,
                         reinterpret_cast<const uint64_t *>(r_cycle.data()), effective_len, reinterpret_cast<uint64_t *>(out.data())), "zg_fr_rows_mle");
    return out;
}

// computeEqPlusOneEvals (src/poly/mod.zig:530-548; src/zkvm/spartan/stage3_prover.zig:1878-1894): eq+1(r, j) over the cube
inline std::vector<Fr> eqPlusOneEvals(const std::vector<Fr> &r) {
    std::vector<Fr> out(size_t(1) << r.size());
    check(zg_fr_eq_plus_one_table(reinterpret_cast<const uint64_t *>(r.data()), r.size(), reinterpret_cast<uint64_t *>(out.data())), "zg_fr_eq_plus_one_table");
    return out;
}

// ---------------------------------------------------------------- wire / disk formats around the path
// G1 coordinates travel as big-endian canonical bytes (commitments, raw SRS) or little-endian canonical bytes (ptau); the conversion
// to Montgomery limbs and the curve check run on the device (zg_field_op, zg_g1_is_on_curve_batch).
namespace wire {
struct SRSError : std::runtime_error { using std::runtime_error::runtime_error; };  // TruncatedData, InvalidFileFormat, UnsupportedFormat, PointNotOnCurve
struct G1Points {
    std::vector<uint64_t> xy;  // n * 8 Montgomery limbs (x | y), zeros at infinity
    std::vector<uint8_t> inf;
    size_t size() const { return inf.size(); }
};
inline uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
inline uint64_t le64(const uint8_t *p) { return (uint64_t)le32(p) | (uint64_t)le32(p + 4) << 32; }
// `count` records of 64 bytes (x | y), big- or little-endian integers; all-zero = infinity; every other point checked on the curve
// (parseG1Uncompressed, src/poly/commitment/srs.zig:65-99; parseG1LE, :616-660)
inline G1Points g1FromBytes(const uint8_t *rec, size_t count, bool big_endian) {
    G1Points out;
    out.xy.assign(count * 8, 0);
    out.inf.assign(count, 0);
    if (!count) return out;
    std::vector<uint64_t> raw(count * 8);
    for (size_t i = 0; i < count; i++) {
        bool any = false;
        for (size_t b = 0; b < 64; b++) any = any || rec[64 * i + b] != 0;
        out.inf[i] = any ? 0 : 1;
        for (size_t c = 0; c < 2; c++)
            for (size_t l = 0; l < 4; l++) {
                uint64_t v = 0;
                for (size_t b = 0; b < 8; b++) {
                    const size_t byte_le = 8 * l + b;  // byte index counted from the least significant end
                    v |= (uint64_t)rec[64 * i + 32 * c + (big_endian ? 31 - byte_le : byte_le)] << (8 * b);
                }
                raw[8 * i + 4 * c + l] = v;
            }
    }
    check(zg_field_op(ZG_FIELD_FP, ZG_OP_TO_MONT, raw.data(), nullptr, out.xy.data(), 2 * count), "zg_field_op");  // reduces like Fp.fromBytes
    for (size_t i = 0; i < count; i++)
        if (out.inf[i]) std::fill(out.xy.begin() + 8 * i, out.xy.begin() + 8 * i + 8, 0);
    std::vector<uint8_t> ok(count);
    check(zg_g1_is_on_curve_batch(out.xy.data(), out.inf.data(), count, ok.data()), "zg_g1_is_on_curve_batch");
    for (uint8_t v : ok)
        if (!v) throw SRSError("PointNotOnCurve");
    return out;
}
// G1 part of loadFromRawBinary (srs.zig:256-306): u32 n (LE) | n x (x BE | y BE) | 128 B tau G2 | 64 B G1 | 128 B G2 (the trailer comes back raw)
inline G1Points srsG1FromRaw(const std::vector<uint8_t> &data, std::vector<uint8_t> *trailer = nullptr) {
    if (data.size() < 4) throw SRSError("TruncatedData");
    const size_t n = le32(data.data());
    if (data.size() < 4 + 64 * n + 128 + 64 + 128) throw SRSError("TruncatedData");
    if (trailer) trailer->assign(data.begin() + 4 + 64 * n, data.end());
    return g1FromBytes(data.data() + 4, n, true);
}
// serializeToRawBinary's G1 section (srs.zig:358-408): toBytesBE of x and y; identity = 64 zero bytes
inline std::vector<uint8_t> g1ToBytesBE(const G1Points &pts) {
    const size_t n = pts.size();
    std::vector<uint64_t> canon(n * 8);
    if (n) check(zg_field_op(ZG_FIELD_FP, ZG_OP_FROM_MONT, pts.xy.data(), nullptr, canon.data(), 2 * n), "zg_field_op");
    std::vector<uint8_t> out(64 * n, 0);
    for (size_t i = 0; i < n; i++) {
        if (pts.inf[i]) continue;
        for (size_t c = 0; c < 2; c++)
            for (size_t l = 0; l < 4; l++)
                for (size_t b = 0; b < 8; b++) out[64 * i + 32 * c + 31 - (8 * l + b)] = (uint8_t)(canon[8 * i + 4 * c + l] >> (8 * b));
    }
    return out;
}
inline std::vector<uint8_t> srsG1ToRaw(const G1Points &pts, const std::vector<uint8_t> &trailer = std::vector<uint8_t>(128 + 64 + 128, 0)) {
    std::vector<uint8_t> out(4);
    for (int b = 0; b < 4; b++) out[b] = (uint8_t)(pts.size() >> (8 * b));
    auto body = g1ToBytesBE(pts);
    out.insert(out.end(), body.begin(), body.end());
    out.insert(out.end(), trailer.begin(), trailer.end());
    return out;
}
// PolyCommitment.toBytes / fromBytes (src/zkvm/commitment_types.zig:49-65): x || y big-endian canonical, identity = 64 zero bytes
inline std::array<uint8_t, 64> commitmentToBytes(const AffinePoint &p) {
    G1Points one;
    one.xy.assign(8, 0);
    one.inf.assign(1, p.infinity ? 1 : 0);
    if (!p.infinity) {
        std::memcpy(one.xy.data(), p.x.limbs, 32);
        std::memcpy(one.xy.data() + 4, p.y.limbs, 32);
    }
    auto v = g1ToBytesBE(one);
    std::array<uint8_t, 64> out;
    std::copy(v.begin(), v.end(), out.begin());
    return out;
}
// G1 side of loadFromPtau (srs.zig:733-900, snarkjs powers-of-tau container): "ptau" | u32 version (= 1) | u32 sections | sections (u32 type,
// u64 size, payload); header payload: u32 field size (= 32) | 32-byte prime | u32 power | u32 ceremony power
struct Ptau {
    uint32_t power = 0, ceremony_power = 0;
    G1Points powers_of_tau_g1, alpha_tau_g1, beta_tau_g1;
    bool has_alpha = false, has_beta = false;
    std::vector<uint8_t> tau_g2_raw, beta_g2_raw;  // pairing side: out of scope, returned untouched
};
inline Ptau srsG1FromPtau(const std::vector<uint8_t> &data) {
    if (data.size() < 12) throw SRSError("TruncatedData");
    if (std::memcmp(data.data(), "ptau", 4) != 0) throw SRSError("InvalidFileFormat");
    if (le32(data.data() + 4) != 1) throw SRSError("UnsupportedFormat");
    const uint32_t nsec = le32(data.data() + 8);
    size_t off = 12;
    std::map<uint32_t, std::pair<size_t, size_t>> secs;  // type -> (offset, size); a later section of the same type wins
    for (uint32_t i = 0; i < nsec; i++) {
        if (off + 12 > data.size()) throw SRSError("TruncatedData");
        const uint32_t typ = le32(data.data() + off);
        const uint64_t size = le64(data.data() + off + 4);
        off += 12;
        if (size > data.size() - off) throw SRSError("TruncatedData");
        secs[typ] = {off, (size_t)size};
        off += (size_t)size;
    }
    if (!secs.count(1)) throw SRSError("InvalidFileFormat");
    const auto hdr = secs[1];
    if (hdr.second < 8) throw SRSError("TruncatedData");
    if (le32(data.data() + hdr.first) != 32) throw SRSError("UnsupportedFormat");
    if (hdr.second < 44) throw SRSError("TruncatedData");
    Ptau out;
    out.power = le32(data.data() + hdr.first + 36);
    out.ceremony_power = le32(data.data() + hdr.first + 40);
    auto points = [&](uint32_t typ, size_t most) {
        const auto sec = secs[typ];
        return g1FromBytes(data.data() + sec.first, std::min(most, sec.second / 64), false);
    };
    if (secs.count(2)) out.powers_of_tau_g1 = points(2, (size_t(1) << out.power) * 2 - 1);
    if (secs.count(4)) { out.alpha_tau_g1 = points(4, size_t(1) << out.power); out.has_alpha = true; }
    if (secs.count(5)) { out.beta_tau_g1 = points(5, size_t(1) << out.power); out.has_beta = true; }
    if (secs.count(3)) out.tau_g2_raw.assign(data.begin() + secs[3].first, data.begin() + secs[3].first + secs[3].second);
    if (secs.count(6)) out.beta_g2_raw.assign(data.begin() + secs[6].first, data.begin() + secs[6].first + secs[6].second);
    return out;
}
// Header of serializeProof (src/zkvm/serialization.zig:283-306): "ZOLT" | u32 version 1 | bytecode proof {commitment, read_ts, write_ts,
// 32-byte legacy field element} | memory proof {commitment, final_state, read_ts, write_ts} | register proof {same four}: the eleven
// commitments this backend produces, in file order
static constexpr const char *PROOF_COMMITMENT_NAMES[11] = {
    "bytecode.commitment", "bytecode.read_ts_commitment", "bytecode.write_ts_commitment", "memory.commitment", "memory.final_state_commitment",
    "memory.read_ts_commitment", "memory.write_ts_commitment", "register.commitment", "register.final_state_commitment", "register.read_ts_commitment",
    "register.write_ts_commitment"};
inline std::array<std::array<uint8_t, 64>, 11> parseZoltProofCommitments(const std::vector<uint8_t> &data) {
    if (data.size() < 8 + 3 * 64 + 32 + 8 * 64 || std::memcmp(data.data(), "ZOLT", 4) != 0) throw std::invalid_argument("not a ZOLT proof");
    if (le32(data.data() + 4) != 1) throw std::invalid_argument("unsupported ZOLT proof version");
    std::array<std::array<uint8_t, 64>, 11> out;
    size_t off = 8;
    for (size_t i = 0; i < 11; i++) {
        if (i == 3) off += 32;  // bytecode._legacy_commitment
        std::memcpy(out[i].data(), data.data() + off, 64);
        off += 64;
    }
    return out;
}
inline std::vector<uint8_t> serializeZoltProofHeader(const std::array<std::array<uint8_t, 64>, 11> &commitments) {
    std::vector<uint8_t> out = {'Z', 'O', 'L', 'T', 1, 0, 0, 0};
    for (size_t i = 0; i < 11; i++) {
        if (i == 3) out.insert(out.end(), 32, 0);  // F.zero()
        out.insert(out.end(), commitments[i].begin(), commitments[i].end());
    }
    return out;  // the first 744 bytes of the proof
}
}  // namespace wire

// ---- the remaining fold sites: SpartanOuterProver's standard rounds (src/zkvm/spartan/outer.zig:364-407), Phase1Prover
// (src/zkvm/spartan/prefix_suffix.zig:35-147), the Lasso PrefixPolynomial (src/zkvm/lasso/prefix_suffix.zig:133-231)
class SpartanOuterProver {
public:
    explicit SpartanOuterProver(const std::vector<Fr> &working_vals) : current_len(working_vals.size()) {
        if (current_len) check(zg_sumcheck_open(reinterpret_cast<const uint64_t *>(working_vals.data()), current_len, ZG_SC_LOW_PAIR, &s_), "zg_sumcheck_open");
    }
    ~SpartanOuterProver() { if (s_) zg_sumcheck_close(s_); }
    SpartanOuterProver(const SpartanOuterProver &) = delete;
    std::array<Fr, 3> computeStandardRoundPoly() {  // [p(0), p(1), 2 p(1) - p(0)]; a single entry left: [it, 0, 0] (:364-388)
        if (current_len <= 1) {
            Fr v = Fr::zero();
            if (current_len == 1) check(zg_sumcheck_final(s_, v.limbs), "zg_sumcheck_final");
            return {v, Fr::zero(), Fr::zero()};
        }
        Fr p0, p1;
        check(zg_sumcheck_round_sums(s_, p0.limbs, p1.limbs), "zg_sumcheck_round_sums");
        return {p0, p1, p1.add(p1).sub(p0)};
    }
    void bindChallenge(const Fr &challenge) {  // :391-407
        challenges.push_back(challenge);
        if (current_len <= 1) return;
        check(zg_sumcheck_bind(s_, challenge.limbs), "zg_sumcheck_bind");
        current_len /= 2;
    }
    size_t current_len;
    std::vector<Fr> challenges;

private:
    zg_sc_t s_ = nullptr;
};

class Phase1Prover {
public:
    void addPair(const std::vector<Fr> &P, const std::vector<Fr> &Q) {
        if (P.size() != Q.size() || (current_size && P.size() != current_size) || s_ || tabs_.size() >= 12)
            throw std::invalid_argument("Phase1Prover.addPair: equal lengths, at most six pairs, before the first round");
        current_size = P.size();
        tabs_.push_back(P);
        tabs_.push_back(Q);
    }
    bool shouldTransition() const { return current_size <= 2; }
    std::array<Fr, 2> computeRoundEvals() {  // g(0), g(1) (:95-112)
        open();
        auto ev = s_->roundExpr(terms_);
        return {ev[0], ev[1]};
    }
    void bind(const Fr &r) {  // :114-132
        open();
        challenges.push_back(r);
        s_->bind(r);
        current_size /= 2;
    }
    std::vector<Fr> buffer(size_t pair, bool q) {  // P (q = false) or Q of a pair, as folded so far
        open();
        return stage3_readTable(*s_, 2 * pair + (q ? 1 : 0));
    }
    size_t current_size = 0;
    std::vector<Fr> challenges;

private:
    static std::vector<Fr> stage3_readTable(ProductSumcheckSession &s, size_t table) { return s.read(table); }
    void open() {
        if (s_) return;
        std::vector<const std::vector<Fr> *> tp;
        for (auto &t : tabs_) tp.push_back(&t);
        s_.reset(new ProductSumcheckSession(tp));
        const int k = (int)tabs_.size() / 2;
        for (int t = 0; t < k / 2; t++) terms_.push_back({{4 * t, 4 * t + 1, 4 * t + 2, 4 * t + 3}, {}, {}, true});
        if (k % 2) terms_.push_back({{2 * k - 2, 2 * k - 1}, {}, {}, false});
        s_->setPoints(0b0011);
        tabs_.clear();
    }
    std::vector<std::vector<Fr>> tabs_;
    std::unique_ptr<ProductSumcheckSession> s_;
    std::vector<ProductSumcheckSession::Term> terms_;
};

// ExpandingTable (src/zkvm/lasso/expanding_table.zig:27-190): after k binds the table IS the eq table of the challenges (first challenge on
// the index's top bit) times the initial value — rebuilt by the device's eq-table kernel; condense (:144-161) = products and sums over runs
class ExpandingTable {
public:
    explicit ExpandingTable(size_t max_rounds, const Fr &initial = Fr::one()) : max_rounds_(max_rounds), initial_(initial), values_{initial} {}
    size_t size() const { return values_.size(); }
    size_t round() const { return r_.size(); }
    void bind(const Fr &r) {  // :83-99
        if (r_.size() >= max_rounds_) throw std::invalid_argument("ExpandingTable.bind: past max_rounds");
        r_.push_back(r);
        values_ = EqPolynomial::evalsSliceWithScaling(r_, &initial_);
    }
    const Fr &get(size_t i) const { return values_.at(i); }
    const std::vector<Fr> &getAll() const { return values_; }
    Fr sum() const {
        Fr s = Fr::zero();
        for (const Fr &v : values_) s = s.add(v);
        return s;
    }
    std::vector<Fr> condense(const std::vector<Fr> &weights, size_t out_bits) const {  // out[i / chunk] += values[i] * weights[i]
        if (weights.size() != values_.size() || out_bits > r_.size()) throw std::invalid_argument("ExpandingTable.condense: weights.len == size, out_bits <= round");
        const size_t out_size = size_t(1) << out_bits, chunk = size_t(1) << (r_.size() - out_bits), n = values_.size();
        std::vector<Fr> prod(n), t(n), ones(chunk, Fr::one()), out(out_size);
        check(zg_field_op(ZG_FIELD_FR, ZG_OP_MUL, reinterpret_cast<const uint64_t *>(values_.data()), reinterpret_cast<const uint64_t *>(weights.data()),
                          reinterpret_cast<uint64_t *>(prod.data()), n), "zg_field_op");
        for (size_t g = 0; g < out_size; g++)  // chunk rows of out_size columns: a column sum per output
            for (size_t c = 0; c < chunk; c++) t[c * out_size + g] = prod[g * chunk + c];
        check(zg_fr_weighted_colsum(reinterpret_cast<const uint64_t *>(t.data()), chunk, out_size, reinterpret_cast<const uint64_t *>(ones.data()), 1,
                                    reinterpret_cast<uint64_t *>(out.data())), "zg_fr_weighted_colsum");
        return out;
    }

private:
    size_t max_rounds_;
    Fr initial_;
    std::vector<Fr> r_, values_;
};

struct LassoPrefixPolynomial {
    std::vector<Fr> evaluations;
    size_t num_vars;
    explicit LassoPrefixPolynomial(std::vector<Fr> evals) : evaluations(std::move(evals)), num_vars(0) {
        while ((size_t(2) << num_vars) <= evaluations.size()) num_vars++;
    }
    LassoPrefixPolynomial bind(const Fr &challenge) const {  // new[i] = old[i] (1 - c) + old[i + half] c (:175-196)
        if (num_vars == 0) throw std::invalid_argument("PrefixPolynomial.bind: no variable left");
        std::vector<Fr> out(evaluations.size() / 2);
        check(zg_fr_bind_high(reinterpret_cast<const uint64_t *>(evaluations.data()), evaluations.size(), challenge.limbs,
                              reinterpret_cast<uint64_t *>(out.data())), "zg_fr_bind_high");
        return LassoPrefixPolynomial(std::move(out));
    }
    Fr evaluate(const std::vector<Fr> &point) const {  // the index's low bit on point[0] (:198-216)
        if (point.size() != num_vars) throw std::invalid_argument("PrefixPolynomial.evaluate: point.len != num_vars");
        if (num_vars == 0) return evaluations[0];
        Fr out;
        check(zg_fr_dense_evaluate(reinterpret_cast<const uint64_t *>(evaluations.data()), num_vars, reinterpret_cast<const uint64_t *>(point.data()), out.limbs),
              "zg_fr_dense_evaluate");
        return out;
    }
};

// ---- Stage 3 as a whole (src/zkvm/spartan/stage3_prover.zig). The witness matrix (cycle-major, 43 elements per padded cycle) is read
// in HBM; cycle-length tables are affine maps of its rows (zg_fr_rows_affine_dev), the Q tables weighted column sums
// (zg_fr_weighted_colsum_dev); prefix / suffix tables have sqrt(T) entries.
namespace stage3 {
constexpr size_t NUM_INPUTS = 43;
// R1CSInputIndex (src/zkvm/r1cs/constraints.zig:39-92), the columns Stage 3 reads
enum Input : size_t { PC = 6, UnexpandedPC = 7, Imm = 8, Rs1Value = 10, Rs2Value = 11, RdWriteValue = 12, FlagVirtualInstruction = 30,
                      FlagIsFirstInSequence = 35, FlagIsNoop = 38, FlagLeftOperandIsRs1 = 39, FlagLeftOperandIsPC = 40,
                      FlagRightOperandIsRs2 = 41, FlagRightOperandIsImm = 42 };
using Map = std::vector<std::pair<size_t, Fr>>;  // (column, coefficient) terms; the constant rides at column NUM_INPUTS

// tables (n entries each, back to back in `out`) = the maps applied to every row; at most 16 per launch
inline void witnessMaps(const uint64_t *d_rows, size_t n, const std::vector<Map> &maps, DeviceMem &out, std::vector<const uint64_t *> &ptrs) {
    out.alloc(maps.size() * n * 32);
    ptrs.clear();
    for (size_t i = 0; i < maps.size(); i++) ptrs.push_back(out.u64() + 4 * i * n);
    for (size_t a = 0; a < maps.size(); a += 16) {
        size_t cnt = std::min<size_t>(16, maps.size() - a);
        std::vector<Fr> coeffs(cnt * (NUM_INPUTS + 1), Fr::zero());
        for (size_t i = 0; i < cnt; i++)
            for (auto &t : maps[a + i]) coeffs[i * (NUM_INPUTS + 1) + t.first] = t.second;
        std::vector<uint64_t *> tabs;
        for (size_t i = 0; i < cnt; i++) tabs.push_back(const_cast<uint64_t *>(ptrs[a + i]));
        check(zg_fr_rows_affine_dev(d_rows, n, NUM_INPUTS, 0, reinterpret_cast<const uint64_t *>(coeffs.data()), cnt, 1, n, tabs.data(), nullptr),
              "zg_fr_rows_affine_dev");
    }
}
// out[k][c] = sum_r weights[k][r] * table[r * cols + c]; weights and sums travel through the host (sqrt(T) entries)
inline std::vector<std::vector<Fr>> colsum(const uint64_t *d_table, size_t rows, size_t cols, const std::vector<const std::vector<Fr> *> &weights) {
    const size_t m = weights.size();
    std::vector<Fr> w(m * rows);
    for (size_t k = 0; k < m; k++) std::copy(weights[k]->begin(), weights[k]->end(), w.begin() + k * rows);
    DeviceMem d_w(m * rows * 32), d_o(m * cols * 32);
    check(zg_memcpy_h2d(d_w.p, w.data(), m * rows * 32), "zg_memcpy_h2d");
    check(zg_fr_weighted_colsum_dev(d_table, rows, cols, d_w.u64(), m, d_o.u64(), nullptr), "zg_fr_weighted_colsum_dev");
    check(zg_sync(), "zg_sync");
    std::vector<Fr> flat(m * cols);
    check(zg_memcpy_d2h(flat.data(), d_o.p, m * cols * 32), "zg_memcpy_d2h");
    std::vector<std::vector<Fr>> out(m);
    for (size_t k = 0; k < m; k++) out[k].assign(flat.begin() + k * cols, flat.begin() + (k + 1) * cols);
    return out;
}
inline Fr evaluateMle(std::vector<Fr> t, const std::vector<Fr> &point) {  // :1820-1838: the point's first entry binds the LOW index bit
    for (const Fr &r : point) {
        if (t.size() == 1) break;
        std::vector<Fr> n(t.size() / 2);
        for (size_t i = 0; i < n.size(); i++) n[i] = t[2 * i].add(r.mul(t[2 * i + 1].sub(t[2 * i])));
        t.swap(n);
    }
    return t[0];
}
inline std::vector<Fr> readTable(ProductSumcheckSession &s, size_t table) { return s.read(table); }
inline std::array<Fr, 4> evalsToCoeffs(const std::vector<Fr> &ev) {  // :846-901, degree 2 (three evaluations) or 3 (four)
    static const std::array<Fr, 2> inv = [] {  // 1/2 and 1/6, once: an inversion costs more than the rest of a round's host algebra
        std::array<Fr, 2> r;
        Fr::fromU64(2).inverse(r[0]);
        Fr::fromU64(6).inverse(r[1]);
        return r;
    }();
    const Fr &two_inv = inv[0], &six_inv = inv[1];
    if (ev.size() == 3) {
        Fr c2 = ev[2].sub(ev[1].add(ev[1])).add(ev[0]).mul(two_inv);
        return {ev[0], ev[1].sub(ev[0]).sub(c2), c2, Fr::zero()};
    }
    Fr d1 = ev[1].sub(ev[0]), d2 = ev[2].sub(ev[1]), d3 = ev[3].sub(ev[2]);
    Fr dd1 = d2.sub(d1), dd2 = d3.sub(d2), c3 = dd2.sub(dd1).mul(six_inv);
    Fr c2 = dd1.mul(two_inv).sub(c3.mul(Fr::fromU64(3)));
    return {ev[0], d1.sub(c2).sub(c3), c2, c3};
}
inline Fr polyAt(const std::array<Fr, 4> &c, const Fr &x) { return c[0].add(x.mul(c[1].add(x.mul(c[2].add(x.mul(c[3])))))); }
}  // namespace stage3

class ShiftPrefixSuffixProver {  // :928-1919
public:
    ShiftPrefixSuffixProver(const uint64_t *d_rows, const std::vector<Fr> &r_outer, const std::vector<Fr> &r_product, const std::vector<Fr> &gamma_powers)
        : g_(gamma_powers) {
        using namespace stage3;
        const size_t n = r_outer.size(), split = n / 2, N = size_t(1) << n;
        if (n < 2 || r_product.size() != n || g_.size() != 5) throw std::invalid_argument("ShiftPrefixSuffixProver: n >= 2, five gamma powers");
        prefix_size_ = size_t(1) << (n - split);
        const size_t ss = size_t(1) << split;
        for (const auto *r : {&r_outer, &r_product}) {  // EqPlusOnePrefixSuffixPoly: PREFIX uses r_lo, SUFFIX uses r_hi
            std::vector<Fr> hi(r->begin(), r->begin() + split), lo(r->begin() + split, r->end());
            Fr is_max = Fr::one();
            for (const Fr &x : lo) is_max = is_max.mul(x);
            std::vector<Fr> p1(prefix_size_, Fr::zero());
            p1[0] = is_max;
            prefix_0_.push_back(eqPlusOneEvals(lo));
            prefix_1_.push_back(p1);
            suffix_0_.push_back(EqPolynomial(hi).evals());
            suffix_1_.push_back(eqPlusOneEvals(hi));
        }
        std::vector<Map> maps = {{{UnexpandedPC, Fr::one()}, {PC, g_[1]}, {FlagVirtualInstruction, g_[2]}, {FlagIsFirstInSequence, g_[3]}},
                                 {{FlagIsNoop, Fr::zero().sub(g_[4])}, {NUM_INPUTS, g_[4]}}};
        for (size_t c : {UnexpandedPC, PC, FlagVirtualInstruction, FlagIsFirstInSequence, FlagIsNoop}) maps.push_back({{c, Fr::one()}});
        DeviceMem buf;
        std::vector<const uint64_t *> ptrs;
        witnessMaps(d_rows, N, maps, buf, ptrs);
        auto qo = colsum(ptrs[0], ss, prefix_size_, {&suffix_0_[0], &suffix_1_[0]});
        auto qp = colsum(ptrs[1], ss, prefix_size_, {&suffix_0_[1], &suffix_1_[1]});
        rounds_.reset(new ProductSumcheckSession({&prefix_0_[0], &qo[0], &prefix_1_[0], &qo[1], &prefix_0_[1], &qp[0], &prefix_1_[1], &qp[1]}));
        rounds_->setPoints(0b0111);
        wit_.reset(new ProductSumcheckSession(ProductSumcheckSession::OnDevice{}, std::vector<const uint64_t *>(ptrs.begin() + 2, ptrs.end()), N));
        // the suffix tables stay resident for the transition: [suffix_0[k]; suffix_1[k]] as a 2 x S matrix per k
        d_suffix_.alloc(4 * ss * 32);
        d_t_.alloc(2 * ss * 32);  // the transition's buffers are allocated here: a hipMalloc / hipFree pair inside the round loop costs more than the round
        d_w_.alloc(4 * 32);
        for (size_t k = 0; k < 2; k++) {
            check(zg_memcpy_h2d(d_suffix_.u64() + 4 * (2 * k) * ss, suffix_0_[k].data(), ss * 32), "zg_memcpy_h2d");
            check(zg_memcpy_h2d(d_suffix_.u64() + 4 * (2 * k + 1) * ss, suffix_1_[k].data(), ss * 32), "zg_memcpy_h2d");
        }
        check(zg_sync(), "zg_sync");
    }
    std::array<Fr, 3> computeRoundEvals(const Fr &previous_claim) {
        if (!in_phase2_) {  // :1351-1392: p(0), p(1), p(2) all from the tables
            auto ev = rounds_->roundExpr({{{0, 1, 2, 3}, {}, {}, true}, {{4, 5, 6, 7}, {}, {}, true}});
            return {ev[0], ev[1], ev[2]};
        }
        Fr neg_g4 = Fr::zero().sub(g_[4]);  // :1399-1455: eq_outer * val + gamma^4 eq_prod - gamma^4 noop eq_prod
        auto ev = rounds_->roundExpr({{{0}, {2, 3, 4, 5}, {Fr::one(), g_[1], g_[2], g_[3]}, false}, {{}, {1}, {g_[4]}, false}, {{6}, {1}, {neg_g4}, false}});
        return {ev[0], previous_claim.sub(ev[0]), ev[2]};
    }
    void bind(const Fr &r_j) {  // :1458-1472
        rounds_->bind(r_j);
        if (in_phase2_) return;
        const bool transition = prefix_size_ == 2;
        wit_->bind(r_j);
        challenges_.push_back(r_j);
        prefix_size_ /= 2;
        if (!transition) return;
        // transitionToPhase2 (:1506-1700). The prefix tables evaluated at the phase-1 challenges are what the round session's P tables
        // have been folded down to (the same LowToHigh steps as stage3::evaluateMle), so they are read from it; the witness columns,
        // folded on the device since round 0, go to the new session inside HBM
        const std::vector<Fr> f = rounds_->final();
        const size_t S = suffix_0_[0].size();
        // t_k[j] = e0 suffix_0[k][j] + e1 suffix_1[k][j]: a 2-row weighted column sum of the resident suffix matrix on the device
        // (round 3 formed the 2 S products on the host: 0.46 ms of a 2.9 ms stage at 2^20 cycles)
        const Fr w[4] = {f[0], f[2], f[4], f[6]};
        check(zg_memcpy_h2d(d_w_.p, w, sizeof(w)), "zg_memcpy_h2d");
        for (size_t k = 0; k < 2; k++)
            check(zg_fr_weighted_colsum_dev(d_suffix_.u64() + 4 * (2 * k) * S, 2, S, d_w_.u64() + 8 * k, 1, d_t_.u64() + 4 * k * S, nullptr), "zg_fr_weighted_colsum_dev");
        check(zg_sync(), "zg_sync");
        std::vector<const uint64_t *> ptrs = {d_t_.u64(), d_t_.u64() + 4 * S};
        for (size_t c = 0; c < 5; c++) ptrs.push_back(wit_->tableDev(c));
        rounds_.reset(new ProductSumcheckSession(ProductSumcheckSession::OnDevice{}, ptrs, S));  // the copies are complete on return
        wit_.reset();
        rounds_->setPoints(0b0101);
        in_phase2_ = true;
    }
    std::vector<Fr> finalClaims() {  // unexpanded_pc, pc, is_virtual, is_first_in_sequence, is_noop (:1860-1876)
        auto f = rounds_->final();
        return std::vector<Fr>(f.begin() + 2, f.end());
    }

private:
    std::vector<Fr> g_, challenges_;
    std::vector<std::vector<Fr>> prefix_0_, prefix_1_, suffix_0_, suffix_1_;
    DeviceMem d_suffix_, d_t_, d_w_;
    std::unique_ptr<ProductSumcheckSession> rounds_, wit_;
    size_t prefix_size_ = 0;
    bool in_phase2_ = false;
};

class RegistersPrefixSuffixProver {  // :2156-2495
public:
    RegistersPrefixSuffixProver(const uint64_t *d_rows, const std::vector<Fr> &r_spartan, const Fr &gamma) : gamma_(gamma) {
        using namespace stage3;
        const size_t n = r_spartan.size(), split = n / 2, N = size_t(1) << n;
        if (n < 2) throw std::invalid_argument("RegistersPrefixSuffixProver: n >= 2");
        r_hi_.assign(r_spartan.begin(), r_spartan.begin() + split);
        r_lo_.assign(r_spartan.begin() + split, r_spartan.end());
        prefix_size_ = size_t(1) << (n - split);
        std::vector<Map> maps = {{{RdWriteValue, Fr::one()}, {Rs1Value, gamma}, {Rs2Value, gamma.mul(gamma)}}};
        for (size_t c : {RdWriteValue, Rs1Value, Rs2Value}) maps.push_back({{c, Fr::one()}});
        DeviceMem buf;
        std::vector<const uint64_t *> ptrs;
        witnessMaps(d_rows, N, maps, buf, ptrs);
        std::vector<Fr> suffix = EqPolynomial(r_hi_).evals(), P = EqPolynomial(r_lo_).evals();
        auto q = colsum(ptrs[0], suffix.size(), prefix_size_, {&suffix});
        rounds_.reset(new ProductSumcheckSession({&P, &q[0]}));
        rounds_->setPoints(0b0101);
        wit_.reset(new ProductSumcheckSession(ProductSumcheckSession::OnDevice{}, std::vector<const uint64_t *>(ptrs.begin() + 1, ptrs.end()), N));
        d_eq_.alloc((size_t(1) << r_hi_.size()) * 32);
        check(zg_sync(), "zg_sync");
    }
    std::array<Fr, 3> computeRoundEvals(const Fr &previous_claim) {  // [p(0), claim - p(0), p(2)] (:2334-2389)
        auto ev = in_phase2_ ? rounds_->roundEvals({0}, {1, 2, 3}, {Fr::one(), gamma_, gamma_.mul(gamma_)}) : rounds_->roundEvals({0, 1});
        return {ev[0], previous_claim.sub(ev[0]), ev[2]};
    }
    void bind(const Fr &r_j) {  // :2388-2398
        rounds_->bind(r_j);
        if (in_phase2_) return;
        const bool transition = prefix_size_ == 2;
        wit_->bind(r_j);
        challenges_.push_back(r_j);
        prefix_size_ /= 2;
        if (!transition) return;
        std::vector<Fr> rev(challenges_.rbegin(), challenges_.rend());  // :2427-2466
        Fr e = EqPolynomial::mle(r_lo_, rev);
        const size_t n_hi = size_t(1) << r_hi_.size();
        // e * eq(r_hi, .) straight from the eq-table kernel (its scale argument): no host table, no upload; buffer allocated at construction
        check(zg_fr_eq_table_dev(r_hi_.empty() ? nullptr : r_hi_[0].limbs, r_hi_.size(), e.limbs, d_eq_.u64(), nullptr), "zg_fr_eq_table_dev");
        check(zg_sync(), "zg_sync");
        std::vector<const uint64_t *> ptrs = {d_eq_.u64()};
        for (size_t c = 0; c < 3; c++) ptrs.push_back(wit_->tableDev(c));  // folded on the device since round 0: handed over inside HBM
        rounds_.reset(new ProductSumcheckSession(ProductSumcheckSession::OnDevice{}, ptrs, n_hi));  // the copies are complete on return
        wit_.reset();
        rounds_->setPoints(0b0101);
        in_phase2_ = true;
    }
    std::vector<Fr> finalClaims() {  // rd_write_value, rs1_value, rs2_value (:2483-2494)
        auto f = rounds_->final();
        return std::vector<Fr>(f.begin() + 1, f.end());
    }

private:
    Fr gamma_;
    std::vector<Fr> r_hi_, r_lo_, challenges_;
    DeviceMem d_eq_;
    std::unique_ptr<ProductSumcheckSession> rounds_, wit_;
    size_t prefix_size_ = 0;
    bool in_phase2_ = false;
};

// the round loop of Stage3Prover.generateStage3Proof (:327-560) over the three instances; the transcript stays the caller's
class Stage3Prover {
public:
    Stage3Prover(const uint64_t *d_rows, const std::vector<Fr> &r_outer, const std::vector<Fr> &r_product, const std::vector<Fr> &shift_gamma_powers,
                 const Fr &instr_gamma, const Fr &reg_gamma, const std::array<Fr, 3> &input_claims, const std::array<Fr, 3> &batching_coeffs)
        : shift(d_rows, r_outer, r_product, shift_gamma_powers), reg(d_rows, r_outer, reg_gamma), claims(input_claims), coeffs_(batching_coeffs) {
        using namespace stage3;
        const size_t N = size_t(1) << r_outer.size();
        std::vector<Map> maps;
        for (size_t c : {FlagLeftOperandIsRs1, Rs1Value, FlagLeftOperandIsPC, UnexpandedPC, FlagRightOperandIsRs2, Rs2Value, FlagRightOperandIsImm, Imm})
            maps.push_back({{c, Fr::one()}});
        DeviceMem buf, d_eq(2 * N * 32);
        std::vector<const uint64_t *> ptrs;
        witnessMaps(d_rows, N, maps, buf, ptrs);
        check(zg_fr_eq_table_dev(reinterpret_cast<const uint64_t *>(r_outer.data()), r_outer.size(), nullptr, d_eq.u64(), nullptr), "zg_fr_eq_table_dev");
        check(zg_fr_eq_table_dev(reinterpret_cast<const uint64_t *>(r_product.data()), r_product.size(), nullptr, d_eq.u64() + 4 * N, nullptr), "zg_fr_eq_table_dev");
        check(zg_sync(), "zg_sync");
        ptrs.push_back(d_eq.u64());
        ptrs.push_back(d_eq.u64() + 4 * N);
        instr_.reset(new ProductSumcheckSession(ProductSumcheckSession::OnDevice{}, ptrs, N));
        check(zg_sync(), "zg_sync");
        Fr g2 = instr_gamma.mul(instr_gamma);
        instr_terms_ = {{{4, 5, 6, 7}, {8, 9}, {Fr::one(), g2}, true}, {{0, 1, 2, 3}, {8, 9}, {instr_gamma, g2.mul(instr_gamma)}, true}};
        instr_->setPoints(0b1101);
        combined_claim = claims[0].mul(coeffs_[0]).add(claims[1].mul(coeffs_[1])).add(claims[2].mul(coeffs_[2]));
    }
    std::array<Fr, 3> computeRoundPolynomial() {  // (c0, c2, c3) of the combined cubic (:333-445)
        auto s = shift.computeRoundEvals(claims[0]);
        auto iv = instr_->roundExpr(instr_terms_);
        auto r = reg.computeRoundEvals(claims[2]);
        evals_[0] = {s[0], s[1], s[2]};
        evals_[1] = {iv[0], claims[1].sub(iv[0]), iv[2], iv[3]};
        evals_[2] = {r[0], r[1], r[2]};
        std::array<Fr, 4> comb;
        Fr three = Fr::fromU64(3);
        for (size_t i = 0; i < 4; i++) {
            comb[i] = Fr::zero();
            for (size_t k = 0; k < 3; k++) {
                const auto &e = evals_[k];
                Fr v = i < e.size() ? e[i] : e[2].mul(three).sub(e[1].mul(three)).add(e[0]);  // a quadratic at 3 (:415-417)
                comb[i] = comb[i].add(v.mul(coeffs_[k]));
            }
        }
        combined_coeffs_ = stage3::evalsToCoeffs(std::vector<Fr>(comb.begin(), comb.end()));
        return {combined_coeffs_[0], combined_coeffs_[2], combined_coeffs_[3]};
    }
    void bindChallenge(const Fr &r_j) {  // :458-490
        combined_claim = stage3::polyAt(combined_coeffs_, r_j);
        for (size_t k = 0; k < 3; k++) claims[k] = stage3::polyAt(stage3::evalsToCoeffs(evals_[k]), r_j);
        shift.bind(r_j);
        instr_->bind(r_j);
        reg.bind(r_j);
    }
    const std::vector<Fr> &roundEvals(size_t k) const { return evals_[k]; }
    ShiftPrefixSuffixProver shift;
    RegistersPrefixSuffixProver reg;
    std::array<Fr, 3> claims;  // shift, instruction input, registers
    Fr combined_claim;

private:
    std::array<Fr, 3> coeffs_;
    std::unique_ptr<ProductSumcheckSession> instr_;
    std::vector<ProductSumcheckSession::Term> instr_terms_;
    std::array<std::vector<Fr>, 3> evals_;
    std::array<Fr, 4> combined_coeffs_;
};

// OutputSumcheckProver's loop (src/zkvm/ram/output_check.zig:375-499): eq * io_mask * (val_final - val_io); val_init folded alongside
class OutputSumcheckProver {
public:
    Fr current_claim;
    OutputSumcheckProver(const std::vector<Fr> &eq_r_address, const std::vector<Fr> &io_mask, const std::vector<Fr> &val_final,
                         const std::vector<Fr> &val_io, const std::vector<Fr> &val_init, const Fr &claim)
        : current_claim(claim), s_({&eq_r_address, &io_mask, &val_final, &val_io, &val_init}), coeff_{Fr::one(), Fr::zero().sub(Fr::one())} {}
    std::array<Fr, 4> roundEvals() { return s_.roundEvals({0, 1}, {2, 3}, coeff_); }               // s(0..3) (:378-430)
    std::array<Fr, 3> computeRoundPolynomial() { return evalsToCompressed(roundEvals()); }           // :445
    void bindChallenge(const Fr &r) { s_.bind(r); }                                                  // :449-480
    void updateClaim(const std::array<Fr, 4> &evals, const Fr &r) {                                  // :482-499
        auto c = interpolateDegree3(evals);
        Fr c1 = evals[1].sub(c[0]).sub(c[2]).sub(c[3]);
        Fr r2 = r.mul(r);
        current_claim = c[0].add(c1.mul(r)).add(c[2].mul(r2)).add(c[3].mul(r2.mul(r)));
    }
    std::vector<Fr> finalValues() { return s_.final(); }  // eq_r_address, io_mask, val_final, val_io, val_init

private:
    ProductSumcheckSession s_;
    std::vector<Fr> coeff_;
};

// RamReadWriteCheckingProver (src/zkvm/ram/read_write_checking.zig:160-1323): the three-phase sumcheck over a sparse access matrix.
// The dense side tables are folded on the device — eq_evals and inc in one two-table LowToHigh session, val_init in a LOW_PAIR
// session — and read back only at the rows / columns the entries touch (zg_psc_gather / zg_sumcheck_gather); the entry algebra
// (pair merges with checkpoints, Gruen's cubic) is host scalar code, as in the reference.
struct MemoryAccess {  // one element of MemoryTrace.accesses
    uint64_t timestamp, address;
    bool is_write;
    uint64_t value;
};
class RamReadWriteCheckingProver {
public:
    // RamReadWriteCheckingProver (src/zkvm/ram/read_write_checking.zig:160-1323) over one device session (zg_rwc_*): the library walks the
    // entry list's integer fields on the host once per round, the coefficients and the dense tables (eq_evals, inc, val_init) live in
    // HBM; here: the trace decoding of init, the split-eq structure, the cubic, the claim.
    struct Entry {  // CycleMajorEntry (:91-157), as read back from the session
        size_t cycle, address;
        Fr ra_coeff, val_coeff;
        uint64_t prev_val, next_val;
    };
    Fr current_claim;
    size_t round = 0;
    std::vector<Fr> challenges;
    Fr last_q_constant = Fr::zero(), last_q_quadratic = Fr::zero();

    RamReadWriteCheckingProver(const std::vector<MemoryAccess> &accesses, const Fr &gamma, const std::vector<Fr> &r_cycle, size_t log_k, size_t log_t,
                               size_t phase1_num_rounds, uint64_t start_address, const Fr &initial_claim,
                               const std::vector<std::pair<uint64_t, uint64_t>> &initial_ram = {})
        : current_claim(initial_claim), gamma_(gamma), log_k_(log_k), log_t_(log_t), p1_(phase1_num_rounds), gruen_(r_cycle) {
        const size_t K = size_t(1) << log_k, T = size_t(1) << log_t;
        std::vector<Fr> val_init(K, Fr::zero());
        std::vector<uint64_t> cur(K, 0);  // the reference's address -> value map (an absent address reads 0), flat: val_init is K elements already
        for (auto &kv : initial_ram)  // :212-231, :253-267
            if (kv.first >= start_address && (kv.first - start_address) / 8 < K) {
                size_t idx = (kv.first - start_address) / 8;
                val_init[idx] = Fr::fromU64(kv.second);
                cur[idx] = kv.second;
            }
        struct Raw { uint32_t cycle, address; uint64_t val, prev, next; bool is_write; };
        std::vector<Raw> raw;
        raw.reserve(accesses.size());
        for (auto &a : accesses) {  // :269-330; inc[timestamp] of a write = F(value) - F(prev) is formed on the device from these entries
            if (a.timestamp >= T || a.address < start_address || (a.address - start_address) / 8 >= K) continue;
            size_t idx = (a.address - start_address) / 8;
            uint64_t prev = cur[idx];
            if (a.is_write) cur[idx] = a.value;
            raw.push_back(Raw{(uint32_t)a.timestamp, (uint32_t)idx, a.is_write ? prev : a.value, prev, a.value, a.is_write});
        }
        // two writes in one cycle: the reference keeps the later one in ACCESS order, so inc is built here (before the sort) and handed over
        std::vector<Fr> inc;
        {
            std::vector<uint32_t> wc;
            for (auto &e : raw) if (e.is_write) wc.push_back(e.cycle);
            if (!std::is_sorted(wc.begin(), wc.end())) std::sort(wc.begin(), wc.end());
            if (std::adjacent_find(wc.begin(), wc.end()) != wc.end()) {
                inc.assign(T, Fr::zero());
                for (auto &e : raw)
                    if (e.is_write) inc[e.cycle] = e.next >= e.prev ? Fr::fromU64(e.next - e.prev) : Fr::zero().sub(Fr::fromU64(e.prev - e.next));
            }
        }
        auto by_cycle_then_address = [](const Raw &x, const Raw &y) { return x.cycle != y.cycle ? x.cycle < y.cycle : x.address < y.address; };
        if (!std::is_sorted(raw.begin(), raw.end(), by_cycle_then_address)) std::stable_sort(raw.begin(), raw.end(), by_cycle_then_address);  // a trace arrives in order
        std::vector<uint32_t> cyc(raw.size()), adr(raw.size());
        std::vector<uint64_t> val(raw.size()), prev(raw.size()), next(raw.size());
        std::vector<uint8_t> wr(raw.size());
        for (size_t i = 0; i < raw.size(); i++) { cyc[i] = raw[i].cycle; adr[i] = raw[i].address; val[i] = raw[i].val; prev[i] = raw[i].prev; next[i] = raw[i].next; wr[i] = raw[i].is_write; }
        if (inc.empty())
            check(zg_rwc_open_writes(log_k, log_t, raw.size(), cyc.data(), adr.data(), val.data(), prev.data(), next.data(), wr.data(),
                                     reinterpret_cast<const uint64_t *>(val_init.data()), reinterpret_cast<const uint64_t *>(r_cycle.data()), &s_), "zg_rwc_open_writes");
        else
            check(zg_rwc_open(log_k, log_t, raw.size(), cyc.data(), adr.data(), val.data(), prev.data(), next.data(), reinterpret_cast<const uint64_t *>(inc.data()),
                              reinterpret_cast<const uint64_t *>(val_init.data()), reinterpret_cast<const uint64_t *>(r_cycle.data()), &s_), "zg_rwc_open");
        eq_size_ = T;
        const size_t m = r_cycle.size() / 2;
        try {  // the two prefix-table sets of the split-eq structure in HBM (table k starts at element 2^k - 1)
            d_out_.alloc(((size_t(2) << m) - 1) * 32);
            d_in_.alloc(((size_t(2) << gruen_.num_x_in) - 1) * 32);
            check(zg_fr_eq_prefix_tables_dev(reinterpret_cast<const uint64_t *>(gruen_.tau.data()), m, d_out_.u64(), nullptr), "zg_fr_eq_prefix_tables_dev");
            check(zg_fr_eq_prefix_tables_dev(reinterpret_cast<const uint64_t *>(gruen_.tau.data() + m), gruen_.num_x_in, d_in_.u64(), nullptr), "zg_fr_eq_prefix_tables_dev");
        } catch (...) {
            zg_rwc_close(s_);
            throw;
        }
    }
    ~RamReadWriteCheckingProver() { zg_rwc_close(s_); }
    RamReadWriteCheckingProver(const RamReadWriteCheckingProver &) = delete;
    RamReadWriteCheckingProver &operator=(const RamReadWriteCheckingProver &) = delete;
    size_t numRounds() const { return log_k_ + log_t_; }
    bool isComplete() const { return round >= numRounds(); }
    size_t numEntries() const { return zg_rwc_entries(s_); }

    std::array<Fr, 4> computeRoundPolynomialCubic() {  // :391-408
        if (inCyclePhase()) {  // computePhase1Polynomial (:410-536) + Gruen's cubic
            size_t head_len = gruen_.current_index - std::min<size_t>(1, gruen_.current_index), m = gruen_.tau.size() / 2;
            size_t ho = std::min(head_len, m), hi = head_len - ho;
            size_t ko = gruen_.E_out_vec.empty() ? 0 : std::min(ho, gruen_.E_out_vec.size() - 1), ki = gruen_.E_in_vec.empty() ? 0 : std::min(hi, gruen_.E_in_vec.size() - 1);
            check(zg_rwc_round_cycle(s_, d_out_.u64() + 4 * ((size_t(1) << ko) - 1), size_t(1) << ko, d_in_.u64() + 4 * ((size_t(1) << ki) - 1), size_t(1) << ki,
                                     gamma_.limbs, last_q_constant.limbs, last_q_quadratic.limbs), "zg_rwc_round_cycle");
            return gruen_.computeCubicRoundPoly(last_q_constant, last_q_quadratic, current_claim);
        }
        const size_t addr_round = round - p1_;  // computePhase2Polynomial (:538-769)
        Fr s0, s2;
        check(zg_rwc_round_address(s_, addr_round, addr_round ? reinterpret_cast<const uint64_t *>(challenges.data() + p1_) : nullptr, gamma_.limbs, s0.limbs, s2.limbs),
              "zg_rwc_round_address");
        Fr s1 = current_claim.sub(s0), three = Fr::fromU64(3);
        return {s0, s1, s2, s2.mul(three).sub(s1.mul(three)).add(s0)};
    }
    void bindChallenge(const Fr &r) {  // :902-970
        challenges.push_back(r);
        if (inCyclePhase() && eq_size_ > 1) {
            check(zg_rwc_bind_cycle(s_, r.limbs), "zg_rwc_bind_cycle");  // eq_evals, inc and the entry list
            eq_size_ /= 2;
            gruen_.bind(r);
        }
        if (round >= p1_ && round < p1_ + log_k_) check(zg_rwc_bind_address(s_, round - p1_, r.limbs), "zg_rwc_bind_address");  // val_init and the list
        round++;
    }
    void updateClaim(const std::array<Fr, 4> &evals, const Fr &challenge) { current_claim = cubicAtPoint(evals, challenge); }  // :1187-1204
    struct OpeningClaims { Fr ra_claim, val_claim, inc_claim; };
    OpeningClaims getOpeningClaims(const std::vector<Fr> &r_sumcheck) {  // :1210-1322
        const size_t p2 = p1_ + log_k_, p3 = log_t_ - p1_;
        std::vector<Fr> r_address(log_k_, Fr::zero()), r_cyc(log_t_, Fr::zero());
        for (size_t i = 0; i < log_k_ && p1_ + i < r_sumcheck.size(); i++) r_address[log_k_ - 1 - i] = r_sumcheck[p1_ + i];
        for (size_t i = 0; i < p1_ && i < r_sumcheck.size(); i++)
            if (p3 + (p1_ - 1 - i) < log_t_) r_cyc[p3 + (p1_ - 1 - i)] = r_sumcheck[i];
        for (size_t i = 0; i < p3 && p2 + i < r_sumcheck.size(); i++) r_cyc[p3 - 1 - i] = r_sumcheck[p2 + i];
        Fr out[3];
        check(zg_rwc_opening(s_, reinterpret_cast<const uint64_t *>(r_address.data()), reinterpret_cast<const uint64_t *>(r_cyc.data()), reinterpret_cast<uint64_t *>(out)),
              "zg_rwc_opening");
        return OpeningClaims{out[0], out[1], out[2]};
    }
    std::vector<Entry> entries() {  // the current list, coefficients from the device
        const size_t n = numEntries();
        std::vector<uint32_t> cyc(n), adr(n);
        std::vector<Fr> ra(n), val(n);
        std::vector<uint64_t> prev(n), next(n);
        check(zg_rwc_read_entries(s_, cyc.data(), adr.data(), reinterpret_cast<uint64_t *>(ra.data()), reinterpret_cast<uint64_t *>(val.data()), prev.data(), next.data()),
              "zg_rwc_read_entries");
        std::vector<Entry> out(n);
        for (size_t i = 0; i < n; i++) out[i] = Entry{cyc[i], adr[i], ra[i], val[i], prev[i], next[i]};
        return out;
    }

private:
    Fr gamma_;
    size_t log_k_, log_t_, p1_, eq_size_ = 0;
    GruenSplitEqPolynomial gruen_;
    zg_rwc_t s_ = nullptr;
    DeviceMem d_out_, d_in_;
    bool inCyclePhase() const { return round < p1_ || round >= p1_ + log_k_; }
};

// InstructionLookupsClaimReductionProver's loop (src/zkvm/claim_reductions/instruction_lookups.zig:146-284)
// ---------------------------------------------------------------- MultiStageProver stages 5 and 6 (src/zkvm/prover.zig:818-1112)
struct StageRoundsResult {
    Fr initial_claim = Fr::zero(), final_claim = Fr::zero();
    std::vector<std::array<Fr, 2>> round_polys;  // [p(0), p(2)] (:925-927)
    std::vector<Fr> challenges, claims;
    bool skipped = false;  // empty trace (:859-863, 1003-1007)
};
inline Fr computeRegEq(const std::vector<Fr> &r, unsigned reg) {  // :961-972
    Fr acc = Fr::one();
    for (size_t i = 0; i < r.size(); i++) acc = acc.mul(((reg >> i) & 1) ? r[i] : Fr::one().sub(r[i]));
    return acc;
}
// the round loop the two stages share (:902-944, 1055-1097) over a HIGH_HALF device session
inline void highHalfRounds(const std::vector<Fr> &evals, size_t num_rounds, Transcript &transcript, const std::string &label, StageRoundsResult &out) {
    zg_sc_t s = nullptr;
    check(zg_sumcheck_open(reinterpret_cast<const uint64_t *>(evals.data()), evals.size(), ZG_SC_HIGH_HALF, &s), "zg_sumcheck_open");
    try {
        for (size_t rd = 0; rd < num_rounds; rd++) {
            Fr p0, p1;
            check(zg_sumcheck_round_sums(s, p0.limbs, p1.limbs), "zg_sumcheck_round_sums");
            if (rd == 0) out.initial_claim = p0.add(p1);
            out.round_polys.push_back({p0, p1.add(p1).sub(p0)});
            Fr ch = transcript.challengeScalar(label);
            out.challenges.push_back(ch);
            check(zg_sumcheck_bind(s, ch.limbs), "zg_sumcheck_bind");
            out.claims.push_back(Fr::one().sub(ch).mul(p0).add(ch.mul(p1)));
        }
        check(zg_sumcheck_final(s, out.final_claim.limbs), "zg_sumcheck_final");
        if (num_rounds == 0) out.initial_claim = out.final_claim;
    } catch (...) {
        zg_sumcheck_close(s);
        throw;
    }
    check(zg_sumcheck_close(s), "zg_sumcheck_close");
}
inline size_t log2Ceil(size_t n) {
    size_t k = 0;
    while ((size_t(1) << k) < n) k++;
    return k;
}
// What ValEvaluationProver.init tabulates (src/zkvm/ram/val_evaluation.zig:423-470): inc from the writes of the trace (IncPolynomial.fromTrace,
// :92-165), wa[j] = eq(r_address, address written in cycle j) (WaPolynomial, :208-262: a gather from the device's eq table of the reversed
// point — index bit i belongs to r_address[i]), lt = LtPolynomial over the cube (:289-330, zg_fr_lt_table); n = ceilPow2(max(trace_len, 1))
struct ValEvaluationTables { std::vector<Fr> inc, wa, lt; };
// inc and wa on the host (one element per write of the trace), n = ceilPow2(max(trace_len, 1)); lt stays out (valEvaluationTables adds it)
inline void valEvaluationIncWa(const std::vector<MemoryAccess> &accesses, const std::vector<std::pair<uint64_t, uint64_t>> &initial_ram, size_t trace_len,
                               size_t k, const std::vector<Fr> &r_address, uint64_t start_address, std::vector<Fr> &inc, std::vector<Fr> &wa) {
    size_t n = 1;
    while (n < std::max<size_t>(trace_len, 1)) n <<= 1;
    inc.assign(n, Fr::zero());
    wa.assign(n, Fr::zero());
    std::map<uint64_t, uint64_t> last;
    for (auto &kv : initial_ram)
        if (kv.first >= start_address && (kv.first - start_address) / 8 < k) last[kv.first] = kv.second;
    std::vector<Fr> eq = EqPolynomial(std::vector<Fr>(r_address.rbegin(), r_address.rend())).evals();
    for (const MemoryAccess &a : accesses) {
        if (!a.is_write || a.address < start_address || (a.address - start_address) / 8 >= k || a.timestamp >= trace_len) continue;
        auto it = last.find(a.address);
        const uint64_t old = it == last.end() ? 0 : it->second;
        inc[a.timestamp] = a.value >= old ? Fr::fromU64(a.value - old) : Fr::zero().sub(Fr::fromU64(old - a.value));
        last[a.address] = a.value;
        wa[a.timestamp] = eq[((a.address - start_address) / 8) % eq.size()];
    }
}
// the same two tables as the list of their non-zero entries, for zg_fr_write_tables_dev: (cycle, word, old value, new value) per write; a
// cycle written twice keeps its later write, as the loop above does by overwriting
struct ValEvaluationWrites {
    size_t n = 1;
    std::vector<uint32_t> cycle, word;
    std::vector<uint64_t> pre, post;
};
inline ValEvaluationWrites valEvaluationWrites(const std::vector<MemoryAccess> &accesses, const std::vector<std::pair<uint64_t, uint64_t>> &initial_ram,
                                               size_t trace_len, size_t k, uint64_t start_address) {
    ValEvaluationWrites w;
    while (w.n < std::max<size_t>(trace_len, 1)) w.n <<= 1;
    std::unordered_map<uint64_t, uint64_t> last;
    last.reserve(initial_ram.size() + 1024);
    for (auto &kv : initial_ram)
        if (kv.first >= start_address && (kv.first - start_address) / 8 < k) last[kv.first] = kv.second;
    std::vector<uint32_t> slot(w.n, ~0u);
    for (const MemoryAccess &a : accesses) {
        if (!a.is_write || a.address < start_address || (a.address - start_address) / 8 >= k || a.timestamp >= trace_len) continue;
        uint64_t &cur = last[a.address];  // an address not seen before reads 0
        uint32_t &sl = slot[a.timestamp];
        if (sl == ~0u) {
            sl = (uint32_t)w.cycle.size();
            w.cycle.push_back((uint32_t)a.timestamp);
            w.word.push_back(0);
            w.pre.push_back(0);
            w.post.push_back(0);
        }
        w.word[sl] = (uint32_t)((a.address - start_address) / 8);
        w.pre[sl] = cur;
        w.post[sl] = a.value;
        cur = a.value;
    }
    return w;
}
inline ValEvaluationTables valEvaluationTables(const std::vector<MemoryAccess> &accesses, const std::vector<std::pair<uint64_t, uint64_t>> &initial_ram,
                                               size_t trace_len, size_t k, const std::vector<Fr> &r_address, const std::vector<Fr> &r_cycle, uint64_t start_address) {
    ValEvaluationTables t;
    valEvaluationIncWa(accesses, initial_ram, trace_len, k, r_address, start_address, t.inc, t.wa);
    const size_t n = t.inc.size();
    t.lt.resize(n);
    std::vector<Fr> full(size_t(1) << r_cycle.size());
    check(zg_fr_lt_table(reinterpret_cast<const uint64_t *>(r_cycle.data()), r_cycle.size(), reinterpret_cast<uint64_t *>(full.data())), "zg_fr_lt_table");
    for (size_t j = 0; j < n; j++) t.lt[j] = full[j % full.size()];  // evaluateAtIndex reads len(r_cycle) index bits
    return t;
}
// proveStage4 (:713-828): Val evaluation — challenges, the prover over the memory trace (init_eval = 0), cubic rounds under "val_eval_round"
struct Stage4Result {
    std::vector<Fr> r_address, r_cycle, challenges;
    std::vector<std::array<Fr, 4>> round_polys;
    Fr initial_claim = Fr::zero(), final_claim = Fr::zero();
    bool skipped = false;
};
inline Stage4Result proveStage4(const std::vector<MemoryAccess> &accesses, const std::vector<std::pair<uint64_t, uint64_t>> &initial_ram, size_t trace_len,
                                size_t log_k, size_t log_t, uint64_t start_address, Transcript &transcript) {
    Stage4Result out;
    for (size_t i = 0; i < log_k; i++) out.r_address.push_back(transcript.challengeScalar("r_address"));
    for (size_t i = 0; i < log_t; i++) out.r_cycle.push_back(transcript.challengeScalar("r_cycle_val"));
    if (trace_len == 0) { out.skipped = true; return out; }
    // the three tables are built in HBM: inc and wa scattered from the list of writes (24 bytes per write cross the boundary), lt by its
    // table kernel (when the cube of r_cycle is at least n entries; else tiled on the host)
    ValEvaluationWrites w = valEvaluationWrites(accesses, initial_ram, trace_len, size_t(1) << log_k, start_address);
    const size_t n = w.n;
    DeviceMem d(3 * n * 32);
    {
        std::vector<Fr> r_eq(out.r_address.rbegin(), out.r_address.rend());
        check(zg_fr_write_tables_dev(n, w.cycle.size(), w.cycle.data(), w.word.data(), w.pre.data(), w.post.data(), reinterpret_cast<const uint64_t *>(r_eq.data()),
                                     log_k, d.u64(), d.u64() + 4 * n, nullptr), "zg_fr_write_tables_dev");
    }
    if ((size_t(1) << log_t) == n) {
        check(zg_fr_lt_table_dev(reinterpret_cast<const uint64_t *>(out.r_cycle.data()), log_t, d.u64() + 8 * n, nullptr), "zg_fr_lt_table_dev");
        check(zg_sync(), "zg_sync");
    } else {
        std::vector<Fr> full(size_t(1) << log_t), lt(n);
        check(zg_fr_lt_table(reinterpret_cast<const uint64_t *>(out.r_cycle.data()), log_t, reinterpret_cast<uint64_t *>(full.data())), "zg_fr_lt_table");
        for (size_t j = 0; j < n; j++) lt[j] = full[j % full.size()];
        check(zg_memcpy_h2d(d.u64() + 8 * n, lt.data(), n * 32), "zg_memcpy_h2d");
    }
    ValEvaluationProver pr(ProductSumcheckSession::OnDevice{}, d.u64(), d.u64() + 4 * n, d.u64() + 8 * n, n, Fr::zero());
    check(zg_sync(), "zg_sync");
    // the initial claim is p(0) + p(1) of the first round (a single entry: the product); that round's evaluations are kept
    std::array<Fr, 4> first = pr.computeRoundPolynomial();
    pr.current_claim = n >= 2 ? first[0].add(first[1]) : first[0];
    out.initial_claim = pr.current_claim;
    const size_t num_rounds = trace_len <= 1 ? 0 : log2Ceil(trace_len);
    for (size_t rd = 0; rd < num_rounds; rd++) {
        auto rp = rd == 0 ? first : pr.computeRoundPolynomial();
        out.round_polys.push_back(rp);
        Fr ch = transcript.challengeScalar("val_eval_round");
        out.challenges.push_back(ch);
        pr.bindChallengeWithPoly(ch, rp);
    }
    auto f = pr.getFinalClaims();
    out.final_claim = f[0].mul(f[1]).mul(f[2]);
    return out;
}
// proveStage5 (:829-958): register value evaluation — eq(r_register, rd(j)) over the trace steps
inline StageRoundsResult proveStage5(const std::vector<uint32_t> &instructions, size_t log_t, Transcript &transcript, std::vector<Fr> *r_register_out = nullptr) {
    std::vector<Fr> r_register(5);
    for (auto &x : r_register) x = transcript.challengeScalar("r_register");
    for (size_t i = 0; i < log_t; i++) (void)transcript.challengeScalar("r_cycle_reg");
    if (r_register_out) *r_register_out = r_register;
    StageRoundsResult out;
    if (instructions.empty()) { out.skipped = true; return out; }
    const size_t num_rounds = instructions.size() <= 1 ? 0 : log2Ceil(instructions.size());
    Fr table[32];
    for (unsigned reg = 0; reg < 32; reg++) table[reg] = computeRegEq(r_register, reg);
    std::vector<Fr> eq_evals(size_t(1) << num_rounds, Fr::zero());
    for (size_t j = 0; j < instructions.size(); j++) eq_evals[j] = table[(instructions[j] >> 7) & 31];
    highHalfRounds(eq_evals, num_rounds, transcript, "reg_eval_round", out);
    return out;
}
// proveStage6 (:990-1112): booleanity — violation_evals = 0 for every step of a valid trace (:1024-1033)
inline StageRoundsResult proveStage6(size_t trace_len, Transcript &transcript, Fr *bool_challenge_out = nullptr) {
    Fr bc = transcript.challengeScalar("booleanity");
    if (bool_challenge_out) *bool_challenge_out = bc;
    StageRoundsResult out;
    if (trace_len == 0) { out.skipped = true; return out; }
    const size_t num_rounds = trace_len <= 1 ? 0 : log2Ceil(trace_len);
    std::vector<Fr> viol(size_t(1) << num_rounds, Fr::zero());
    highHalfRounds(viol, num_rounds, transcript, "bool_round", out);
    return out;
}

// ---------------------------------------------------------------- Spartan outer sumcheck, remaining rounds
// The 19 uniform R1CS constraints (src/zkvm/r1cs/constraints.zig:248-531, the published Jolt R1CS): condition * (left - right) = 0 with
// each side a linear combination of the 43 per-cycle inputs (R1CSInputIndex, :39-92) plus a constant.
namespace r1cs {
constexpr size_t NUM_INPUTS = 43;
enum In : int {
    LeftInstructionInput, RightInstructionInput, Product, WriteLookupOutputToRD, WritePCtoRD, ShouldBranch, PC, UnexpandedPC, Imm, RamAddress,
    Rs1Value, Rs2Value, RdWriteValue, RamReadValue, RamWriteValue, LeftLookupOperand, RightLookupOperand, NextUnexpandedPC, NextPC, NextIsVirtual,
    NextIsFirstInSequence, LookupOutput, ShouldJump, FlagAddOperands, FlagSubtractOperands, FlagMultiplyOperands, FlagLoad, FlagStore, FlagJump,
    FlagWriteLookupOutputToRD, FlagVirtualInstruction, FlagAssert, FlagDoNotUpdateUnexpandedPC, FlagAdvice, FlagIsCompressed, FlagIsFirstInSequence
};
struct Term { int input; int coeff; };
struct LC {
    std::vector<Term> terms;
    bool two_pow_64 = false;  // the one constant that does not fit an int (constraint 8)
    int constant = 0;
};
struct Constraint { LC condition, left, right; };
inline LC lc(std::vector<Term> t, int c = 0) { return LC{std::move(t), false, c}; }
inline const std::vector<Constraint> &uniformConstraints() {
    static const std::vector<Constraint> k = [] {
        LC sub_rhs = lc({{LeftInstructionInput, 1}, {RightInstructionInput, -1}});
        sub_rhs.two_pow_64 = true;
        return std::vector<Constraint>{
            {lc({{FlagLoad, 1}, {FlagStore, 1}}), lc({{RamAddress, 1}}), lc({{Rs1Value, 1}, {Imm, 1}})},
            {lc({{FlagLoad, -1}, {FlagStore, -1}}, 1), lc({{RamAddress, 1}}), lc({})},
            {lc({{FlagLoad, 1}}), lc({{RamReadValue, 1}}), lc({{RamWriteValue, 1}})},
            {lc({{FlagLoad, 1}}), lc({{RamReadValue, 1}}), lc({{RdWriteValue, 1}})},
            {lc({{FlagStore, 1}}), lc({{Rs2Value, 1}}), lc({{RamWriteValue, 1}})},
            {lc({{FlagAddOperands, 1}, {FlagSubtractOperands, 1}, {FlagMultiplyOperands, 1}}), lc({{LeftLookupOperand, 1}}), lc({})},
            {lc({{FlagAddOperands, -1}, {FlagSubtractOperands, -1}, {FlagMultiplyOperands, -1}}, 1), lc({{LeftLookupOperand, 1}}), lc({{LeftInstructionInput, 1}})},
            {lc({{FlagAddOperands, 1}}), lc({{RightLookupOperand, 1}}), lc({{LeftInstructionInput, 1}, {RightInstructionInput, 1}})},
            {lc({{FlagSubtractOperands, 1}}), lc({{RightLookupOperand, 1}}), sub_rhs},
            {lc({{FlagMultiplyOperands, 1}}), lc({{RightLookupOperand, 1}}), lc({{Product, 1}})},
            {lc({{FlagAddOperands, -1}, {FlagSubtractOperands, -1}, {FlagMultiplyOperands, -1}, {FlagAdvice, -1}}, 1), lc({{RightLookupOperand, 1}}),
             lc({{RightInstructionInput, 1}})},
            {lc({{FlagAssert, 1}}), lc({{LookupOutput, 1}}), lc({}, 1)},
            {lc({{WriteLookupOutputToRD, 1}}), lc({{RdWriteValue, 1}}), lc({{LookupOutput, 1}})},
            {lc({{WritePCtoRD, 1}}), lc({{RdWriteValue, 1}}), lc({{UnexpandedPC, 1}, {FlagIsCompressed, -2}}, 4)},
            {lc({{ShouldJump, 1}}), lc({{NextUnexpandedPC, 1}}), lc({{LookupOutput, 1}})},
            {lc({{ShouldBranch, 1}}), lc({{NextUnexpandedPC, 1}}), lc({{UnexpandedPC, 1}, {Imm, 1}})},
            {lc({{ShouldBranch, -1}, {FlagJump, -1}}, 1), lc({{NextUnexpandedPC, 1}}),
             lc({{UnexpandedPC, 1}, {FlagDoNotUpdateUnexpandedPC, -4}, {FlagIsCompressed, -2}}, 4)},
            {lc({{FlagVirtualInstruction, 1}}), lc({{NextPC, 1}}), lc({{PC, 1}}, 1)},
            {lc({{NextIsVirtual, 1}, {NextIsFirstInSequence, -1}}), lc({}, 1), lc({{FlagDoNotUpdateUnexpandedPC, 1}})},
        };
    }();
    return k;
}
constexpr int FIRST_GROUP[10] = {1, 2, 3, 4, 5, 6, 11, 14, 17, 18};  // :537-548
constexpr int SECOND_GROUP[9] = {0, 7, 8, 9, 10, 12, 13, 15, 16};    // :553-563
inline Fr fromInt(int v) { return v >= 0 ? Fr::fromU64((uint64_t)v) : Fr::zero().sub(Fr::fromU64((uint64_t)(-(int64_t)v))); }
}  // namespace r1cs

// L_i(r) over the symmetric domain {-(size-1)/2, ...} (LagrangePoly.evals; computeLagrangeEvalsAtR0, streaming_outer.zig:1157-1213)
inline std::vector<Fr> lagrangeEvals(const Fr &r, size_t size = 10) {
    const int start = -(int)((size - 1) / 2);
    std::vector<Fr> out(size);
    for (size_t i = 0; i < size; i++) {
        Fr num = Fr::one(), den = Fr::one(), inv;
        for (size_t j = 0; j < size; j++) {
            if (j == i) continue;
            num = num.mul(r.sub(r1cs::fromInt(start + (int)j)));
            den = den.mul(r1cs::fromInt((int)i - (int)j));
        }
        out[i] = den.inverse(inv) ? num.mul(inv) : Fr::zero();
    }
    return out;
}
// LagrangePoly.lagrangeKernel (src/zkvm/r1cs/univariate_skip.zig:296-312)
inline Fr lagrangeKernel(const Fr &x, const Fr &y, size_t size = 10) {
    auto a = lagrangeEvals(x, size), b = lagrangeEvals(y, size);
    Fr acc = Fr::zero();
    for (size_t i = 0; i < size; i++) acc = acc.add(a[i].mul(b[i]));
    return acc;
}

// StreamingOuterProver's remaining rounds (src/zkvm/spartan/streaming_outer.zig: :120-212, 1135-1155, 258-372, 1215-1281, 1681-1737): the
// cycle witnesses are uploaded once, Az / Bz of both constraint groups are ONE affine-map launch over them (zg_fr_rows_affine_dev) into a
// two-table product session; a round is zg_psc_round_gruen + zg_psc_bind, the split-eq scalar / cubic / claim are host algebra.
class StreamingOuterProver {
public:
    using CycleInputs = std::array<Fr, r1cs::NUM_INPUTS>;  // R1CSCycleInputs.values
    Fr current_claim = Fr::zero(), last_t_zero = Fr::zero(), last_t_infinity = Fr::zero();
    size_t current_round = 0, num_cycle_vars = 0, padded_trace_len = 1;
    GruenSplitEqPolynomial split_eq;
    std::vector<Fr> challenges, lagrange_evals_r0;

    StreamingOuterProver(const std::vector<CycleInputs> &cycle_witnesses, const std::vector<Fr> &tau, const Fr *lagrange_tau_r0 = nullptr)
        : split_eq(std::vector<Fr>(tau.begin(), tau.end() - (tau.empty() ? 0 : 1)), lagrange_tau_r0), num_cycles_(cycle_witnesses.size()),
          tau_high_(tau.empty() ? Fr::zero() : tau.back()) {
        if (cycle_witnesses.empty()) throw std::invalid_argument("StreamingOuterProver: empty trace");  // error.EmptyTrace
        while (padded_trace_len < num_cycles_) padded_trace_len <<= 1, num_cycle_vars++;
        if (tau.size() != num_cycle_vars + 2) throw std::invalid_argument("StreamingOuterProver: tau has num_cycle_vars + 2 challenges");
        d_rows_.alloc(num_cycles_ * r1cs::NUM_INPUTS * 32);
        check(zg_memcpy_h2d(d_rows_.p, cycle_witnesses.data(), num_cycles_ * r1cs::NUM_INPUTS * 32), "zg_memcpy_h2d");
        const size_t m = split_eq.tau.size() / 2;
        d_out_.alloc(((size_t(2) << m) - 1) * 32);
        d_in_.alloc(((size_t(2) << split_eq.num_x_in) - 1) * 32);
        check(zg_fr_eq_prefix_tables_dev(reinterpret_cast<const uint64_t *>(split_eq.tau.data()), m, d_out_.u64(), nullptr), "zg_fr_eq_prefix_tables_dev");
        check(zg_fr_eq_prefix_tables_dev(reinterpret_cast<const uint64_t *>(split_eq.tau.data() + m), split_eq.num_x_in, d_in_.u64(), nullptr), "zg_fr_eq_prefix_tables_dev");
        check(zg_sync(), "zg_sync");
    }
    size_t numRounds() const { return 1 + num_cycle_vars; }
    // uniskipTargets / COEFFS_PER_J of the outer sumcheck (src/zkvm/r1cs/univariate_skip.zig:188-225, 398-476): -5, 6, -6, ... and, per target,
    // the Lagrange basis of the base window {-4..5} at it (integers)
    static std::array<int, 9> uniskipTargets() { return {-5, 6, -6, 7, -7, 8, -8, 9, -9}; }
    static std::array<long long, 10> shiftCoeffs(int target) {
        std::array<long long, 10> out;
        for (int i = 0; i < 10; i++) {  // L_i(target) = prod_{j != i} (target - x_j) / (x_i - x_j), x_k = -4 + k: exact integer division
            long long num = 1, den = 1;
            for (int j = 0; j < 10; j++)
                if (j != i) { num *= target - (-4 + j); den *= i - j; }
            out[i] = num / den;
        }
        return out;
    }
    // coefficients (ascending) of the polynomial through (left + i, vals[i]) (lagrangeInterpolate, streaming_outer.zig:728-799)
    static std::vector<Fr> interpolateIntDomain(const std::vector<Fr> &vals, int left) {
        const size_t n = vals.size();
        std::vector<Fr> coeffs(n, Fr::zero());
        for (size_t i = 0; i < n; i++) {
            if (vals[i].isZero()) continue;
            Fr den = Fr::one(), inv;
            std::vector<Fr> basis(n, Fr::zero());
            basis[0] = Fr::one();
            size_t deg = 0;
            for (size_t j = 0; j < n; j++) {
                if (j == i) continue;
                den = den.mul(r1cs::fromInt((int)i - (int)j));
                const Fr neg_xj = r1cs::fromInt(-(left + (int)j));
                for (size_t k = deg + 1; k > 0; k--) basis[k] = k <= deg ? basis[k - 1].add(neg_xj.mul(basis[k])) : basis[k - 1];
                basis[0] = neg_xj.mul(basis[0]);
                deg++;
            }
            den.inverse(inv);
            const Fr scale = vals[i].mul(inv);
            for (size_t k = 0; k < n; k++) coeffs[k] = coeffs[k].add(basis[k].mul(scale));
        }
        return coeffs;
    }
    // computeFirstRoundPoly (:523-597): t1 at the nine targets by ONE launch over the resident witnesses (zg_fr_rows_affine_prodsum_dev), then
    // s1 = L(tau_high, .) * t1 as 28 coefficients on the host
    std::vector<Fr> last_extended_evals;
    std::array<Fr, 28> computeFirstRoundPoly() {
        const size_t W = r1cs::NUM_INPUTS + 1;
        std::vector<Fr> m(36 * W, Fr::zero());  // rows 2 p, 2 p + 1 = A_p, B_p for pair p = 2 j + group
        const auto &cs = r1cs::uniformConstraints();
        const auto targets = uniskipTargets();
        auto add = [&](size_t row, const r1cs::LC &l, const Fr &w, bool negate) {
            for (const auto &t : l.terms) m[row * W + t.input] = m[row * W + t.input].add(w.mul(r1cs::fromInt(negate ? -t.coeff : t.coeff)));
            Fr c = r1cs::fromInt(l.constant);
            if (l.two_pow_64) c = c.add(Fr::fromU64(uint64_t(1) << 32).mul(Fr::fromU64(uint64_t(1) << 32)));
            c = w.mul(c);
            m[row * W + r1cs::NUM_INPUTS] = negate ? m[row * W + r1cs::NUM_INPUTS].sub(c) : m[row * W + r1cs::NUM_INPUTS].add(c);
        };
        for (size_t j = 0; j < 9; j++) {
            const auto alpha = shiftCoeffs(targets[j]);
            for (size_t g = 0; g < 2; g++) {
                const size_t p = 2 * j + g, gs = g == 0 ? 10 : 9;  // the second group uses the first nine of the ten coefficients (:631-657)
                for (size_t i = 0; i < gs; i++) {
                    const Fr a = alpha[i] >= 0 ? Fr::fromU64((uint64_t)alpha[i]) : Fr::zero().sub(Fr::fromU64((uint64_t)(-alpha[i])));
                    const auto &c = cs[g == 0 ? r1cs::FIRST_GROUP[i] : r1cs::SECOND_GROUP[i]];
                    add(2 * p, c.condition, a, false);
                    add(2 * p + 1, c.left, a, false);
                    add(2 * p + 1, c.right, a, true);
                }
            }
        }
        DeviceMem d_w((size_t(1) << split_eq.tau.size()) * 32);  // eq(tau_low, .): index = cycle * 2 + group (:541-566)
        check(zg_fr_eq_table_dev(reinterpret_cast<const uint64_t *>(split_eq.tau.data()), split_eq.tau.size(), nullptr, d_w.u64(), nullptr), "zg_fr_eq_table_dev");
        Fr out[18];
        check(zg_fr_rows_affine_prodsum_dev(d_rows_.u64(), std::min(num_cycles_, padded_trace_len), r1cs::NUM_INPUTS, 0, reinterpret_cast<const uint64_t *>(m.data()), 18,
                                            d_w.u64(), 2, reinterpret_cast<uint64_t *>(out), nullptr), "zg_fr_rows_affine_prodsum_dev");
        std::vector<Fr> t1(19, Fr::zero());
        last_extended_evals.assign(9, Fr::zero());
        for (size_t j = 0; j < 9; j++) {
            last_extended_evals[j] = out[2 * j].add(out[2 * j + 1]);
            t1[(size_t)(targets[j] + 9)] = last_extended_evals[j];
        }
        const std::vector<Fr> t1c = interpolateIntDomain(t1, -9), lc = interpolateIntDomain(lagrangeEvals(tau_high_, 10), -4);
        std::array<Fr, 28> s1;
        for (auto &x : s1) x = Fr::zero();
        for (size_t i = 0; i < 10; i++)
            for (size_t j = 0; j < 19; j++) s1[i + j] = s1[i + j].add(lc[i].mul(t1c[j]));
        return s1;
    }
    void bindFirstRoundChallenge(const Fr &r0, const Fr &uni_skip_claim) {  // r0 is not bound in split_eq (:1135-1155)
        current_round = 1;
        current_claim = uni_skip_claim;
        lagrange_evals_r0 = lagrangeEvals(r0, 10);
    }
    // rows az(group 0), az(group 1), bz(group 0), bz(group 1) as affine maps of a cycle's inputs, the constant last (:300-345)
    std::vector<Fr> constraintMatrix() const {
        const size_t W = r1cs::NUM_INPUTS + 1;
        std::vector<Fr> m(4 * W, Fr::zero());
        const Fr two64 = Fr::fromU64(uint64_t(1) << 32).mul(Fr::fromU64(uint64_t(1) << 32));
        auto add = [&](size_t row, const r1cs::LC &l, const Fr &w, bool negate) {
            for (const auto &t : l.terms) {
                Fr v = w.mul(r1cs::fromInt(negate ? -t.coeff : t.coeff));
                m[row * W + t.input] = m[row * W + t.input].add(v);
            }
            Fr c = r1cs::fromInt(l.constant);
            if (l.two_pow_64) c = c.add(two64);
            c = w.mul(c);
            m[row * W + r1cs::NUM_INPUTS] = negate ? m[row * W + r1cs::NUM_INPUTS].sub(c) : m[row * W + r1cs::NUM_INPUTS].add(c);
        };
        const auto &cs = r1cs::uniformConstraints();
        for (size_t t = 0; t < 10; t++) {
            const auto &c0 = cs[r1cs::FIRST_GROUP[t]];
            add(0, c0.condition, lagrange_evals_r0[t], false);
            add(2, c0.left, lagrange_evals_r0[t], false);
            add(2, c0.right, lagrange_evals_r0[t], true);
            if (t < 9) {
                const auto &c1 = cs[r1cs::SECOND_GROUP[t]];
                add(1, c1.condition, lagrange_evals_r0[t], false);
                add(3, c1.left, lagrange_evals_r0[t], false);
                add(3, c1.right, lagrange_evals_r0[t], true);
            }
        }
        return m;
    }
    void materializeLinearPhasePolynomials() {  // Az[2 i + group], Bz[2 i + group], zero past the trace (:258-372)
        const size_t n2 = 2 * padded_trace_len;
        DeviceMem d_az(n2 * 32), d_bz(n2 * 32);
        std::vector<Fr> m = constraintMatrix();
        uint64_t *tabs[2] = {d_az.u64(), d_bz.u64()};
        check(zg_fr_rows_affine_dev(d_rows_.u64(), std::min(num_cycles_, padded_trace_len), r1cs::NUM_INPUTS, 0, reinterpret_cast<const uint64_t *>(m.data()), 2, 2,
                                    padded_trace_len, tabs, nullptr), "zg_fr_rows_affine_dev");
        s_.reset(new ProductSumcheckSession(ProductSumcheckSession::OnDevice{}, {d_az.u64(), d_bz.u64()}, n2));
        check(zg_sync(), "zg_sync");  // the session holds its own copies before the two buffers are released
    }
    std::array<Fr, 4> computeRemainingRoundPoly() {  // :1215-1281
        if (current_round == 1 && !s_) materializeLinearPhasePolynomials();
        auto w = split_eq.getWindowEqTables(0, 1);
        size_t n_out = w.E_out->size(), n_in = w.E_in->size();
        auto t = s_->roundGruen({0, 1}, d_out_.u64() + 4 * (n_out - 1), n_out, d_in_.u64() + 4 * (n_in - 1), n_in);
        last_t_zero = t[0];
        last_t_infinity = t[1];
        return split_eq.computeCubicRoundPoly(t[0], t[1], current_claim);
    }
    void bindRemainingRoundChallenge(const Fr &r) {  // split_eq first, then Az / Bz low-to-high (:1681-1717)
        challenges.push_back(r);
        split_eq.bind(r);
        s_->bind(r);
        current_round++;
    }
    void updateClaim(const std::array<Fr, 4> &round_poly, const Fr &challenge) { current_claim = cubicAtPoint(round_poly, challenge); }
    Fr getFinalEval() const { return current_claim; }
    std::array<Fr, 2> finalAzBz() {
        auto f = s_->final();
        return {f[0], f[1]};
    }

private:
    size_t num_cycles_;
    Fr tau_high_;
    DeviceMem d_rows_, d_out_, d_in_;
    std::unique_ptr<ProductSumcheckSession> s_;
};

// Stage4GruenProver (src/zkvm/spartan/stage4_gruen_prover.zig:65-1240), RegistersReadWriteChecking: the five dense K = 128 x T tables
// and inc[T] are built on the device from the per-cycle trace columns and stay in HBM (zg_rrw_*); the eq structure (its prefix tables in
// device buffers as well), Gruen's cubic and the claim algebra stay on the host, as in the reference.
struct TraceStep {  // what the prover reads of ExecutionTrace.steps (:196-246)
    uint32_t instruction;
    uint64_t rd_value;
    bool is_noop;
};
// what initWithPhaseConfig / initWithClaims read of the trace (stage4_gruen_prover.zig:183-258 = stage4_prover.zig:183-277), as the columns
// zg_rrw_open_trace takes: the register a cycle reads / writes (0xFF: none) and the value it writes; the register file before every cycle
// and inc of the written register are rebuilt from them on the device
inline zg_rrw_t openRegistersSession(const std::vector<TraceStep> &steps, size_t log_T, const Fr &gamma) {
    const size_t T = size_t(1) << log_T;
    std::vector<uint8_t> rs1(T, 0xFF), rs2(T, 0xFF), rd(T, 0xFF);
    std::vector<uint64_t> rd_value(T, 0);
    for (size_t j = 0; j < T && j < steps.size(); j++) {
        if (steps[j].is_noop) continue;
        const uint32_t w = steps[j].instruction, op = w & 0x7F, f_rd = (w >> 7) & 31, f_rs1 = (w >> 15) & 31, f_rs2 = (w >> 20) & 31;
        const bool two = op == 0x33 || op == 0x3B || op == 0x23 || op == 0x63;
        if (two || op == 0x13 || op == 0x03 || op == 0x67 || op == 0x1B) rs1[j] = (uint8_t)f_rs1;
        if (two) rs2[j] = (uint8_t)f_rs2;
        if (op != 0x23 && op != 0x63 && f_rd != 0) {
            rd[j] = (uint8_t)f_rd;
            rd_value[j] = steps[j].rd_value;
        }
    }
    zg_rrw_t s = nullptr;
    check(zg_rrw_open_trace(log_T, rs1.data(), rs2.data(), rd.data(), rd_value.data(), gamma.limbs, &s), "zg_rrw_open_trace");
    return s;
}
class Stage4GruenProver {
public:
    static constexpr size_t LOG_K = 7, K = 128;
    size_t T = 1, log_T = 0, current_T = 0, current_K = K, num_rounds = 0;
    Fr last_q_constant = Fr::zero(), last_q_quadratic = Fr::zero();

    // r_cycle in ROUND order (r_cycle[0] is bound first); the split-eq structure takes it big-endian (:283-288)
    Stage4GruenProver(const std::vector<TraceStep> &steps, const Fr &gamma, const std::vector<Fr> &r_cycle, size_t phase1_num_rounds, size_t phase2_num_rounds)
        : p1_(phase1_num_rounds), p2_(phase2_num_rounds), gruen_(std::vector<Fr>(r_cycle.rbegin(), r_cycle.rend())) {
        while (T < steps.size()) T <<= 1, log_T++;
        if (r_cycle.size() != log_T || log_T < 1 || p1_ < 1 || p1_ > log_T || p2_ != LOG_K) throw std::invalid_argument("Stage4GruenProver: configuration");
        current_T = T;
        num_rounds = LOG_K + log_T;
        s_ = openRegistersSession(steps, log_T, gamma);
        // the two prefix-table sets of the split-eq structure, in HBM for the phase-1 rounds (table k starts at element 2^k - 1)
        const size_t m = log_T / 2;
        try {
            d_out_.alloc(((size_t(2) << m) - 1) * 32);
            d_in_.alloc(((size_t(2) << gruen_.num_x_in) - 1) * 32);
            check(zg_fr_eq_prefix_tables_dev(reinterpret_cast<const uint64_t *>(gruen_.tau.data()), m, d_out_.u64(), nullptr), "zg_fr_eq_prefix_tables_dev");
            check(zg_fr_eq_prefix_tables_dev(reinterpret_cast<const uint64_t *>(gruen_.tau.data() + m), gruen_.num_x_in, d_in_.u64(), nullptr), "zg_fr_eq_prefix_tables_dev");
            check(zg_sync(), "zg_sync");  // the session reads the tables on its own stream
        } catch (...) {
            zg_rrw_close(s_);
            throw;
        }
    }
    Stage4GruenProver(const Stage4GruenProver &) = delete;
    Stage4GruenProver &operator=(const Stage4GruenProver &) = delete;
    ~Stage4GruenProver() { zg_rrw_close(s_); }

    std::array<Fr, 4> computeRoundEvals(size_t round, const Fr &current_claim) {  // :1165-1190
        if (round < p1_) {  // phase1ComputeMessage (:561-741)
            size_t head_len = gruen_.current_index - std::min<size_t>(1, gruen_.current_index), m = gruen_.tau.size() / 2;
            size_t ho = std::min(head_len, m), hi = head_len - ho;
            size_t ko = std::min(ho, gruen_.E_out_vec.size() - 1), ki = std::min(hi, gruen_.E_in_vec.size() - 1);
            check(zg_rrw_round_cycle_gruen(s_, d_out_.u64() + 4 * ((size_t(1) << ko) - 1), size_t(1) << ko, d_in_.u64() + 4 * ((size_t(1) << ki) - 1), size_t(1) << ki,
                                           last_q_constant.limbs, last_q_quadratic.limbs), "zg_rrw_round_cycle_gruen");
            return gruen_.computeCubicRoundPoly(last_q_constant, last_q_quadratic, current_claim);
        }
        if (round < p1_ + p2_ || current_T == 1) {  // phase2ComputeMessage (:764-852); phase 3 with a single cycle left (:955-1013)
            Fr e0, e2;
            check(zg_rrw_round_address(s_, e0.limbs, nullptr, e2.limbs), "zg_rrw_round_address");
            Fr e1 = current_claim.sub(e0), three = Fr::fromU64(3);
            return {e0, e1, e2, e0.sub(three.mul(e1)).add(three.mul(e2))};  // the quadratic's p(3) (:841-850)
        }
        Fr e0, e2, e3;  // phase3ComputeMessage (:854-953)
        check(zg_rrw_round_cycle(s_, e0.limbs, nullptr, e2.limbs, e3.limbs), "zg_rrw_round_cycle");
        return {e0, current_claim.sub(e0), e2, e3};
    }
    void bindChallenge(size_t round, const Fr &challenge) {  // :1047-1163, 1192-1216
        if (round < p1_ || round >= p1_ + p2_) {
            check(zg_rrw_bind_cycle(s_, challenge.limbs), "zg_rrw_bind_cycle");
            current_T /= 2;
            if (round < p1_) {
                gruen_.bind(challenge);
                if (round == p1_ - 1) {  // gruen_eq.merge (gruen_eq.zig:119-146)
                    std::vector<Fr> eq = gruen_.getFullEqTable();
                    check(zg_rrw_set_eq(s_, reinterpret_cast<const uint64_t *>(eq.data()), eq.size()), "zg_rrw_set_eq");
                }
            }
        } else {
            check(zg_rrw_bind_address(s_, challenge.limbs), "zg_rrw_bind_address");
            current_K /= 2;
        }
    }
    struct FinalClaims { Fr val_claim, rs1_ra_claim, rs2_ra_claim, rd_wa_claim, inc_claim; };
    FinalClaims getFinalClaims() {  // :1219-1236
        Fr f[7];
        check(zg_rrw_final(s_, reinterpret_cast<uint64_t *>(f)), "zg_rrw_final");
        return FinalClaims{f[0], f[3], f[4], f[1], f[5]};
    }
    std::array<Fr, 3> finalCheck() {  // (eq_scalar, combined, expected) as printed after the last round (:1196-1210)
        Fr f[7];
        check(zg_rrw_final(s_, reinterpret_cast<uint64_t *>(f)), "zg_rrw_final");
        Fr comb = f[2].mul(f[0]).add(f[1].mul(f[0].add(f[5])));
        return {f[6], comb, f[6].mul(comb)};
    }

private:
    size_t p1_, p2_;
    GruenSplitEqPolynomial gruen_;
    zg_rrw_t s_ = nullptr;
    DeviceMem d_out_, d_in_;
};

// the original Stage4Prover (src/zkvm/spartan/stage4_prover.zig:74-865) on the same device session: dense eq table from the start, every
// cycle variable first, all four evaluations from the tables (:601-723), full-coefficient round polynomial (:731-758)
class Stage4Prover {
public:
    static constexpr size_t LOG_K = 7, K = 128;
    size_t T = 1, log_T = 0, current_T = 0, current_K = K, num_rounds = 0;
    Stage4Prover(const std::vector<TraceStep> &steps, const Fr &gamma, const std::vector<Fr> &r_cycle) {
        if (steps.empty()) throw std::invalid_argument("Stage4Prover: empty trace");  // error.EmptyTrace
        while (T < steps.size()) T <<= 1, log_T++;
        if (log_T < 1) throw std::invalid_argument("Stage4Prover: at least two cycles");  // (the device session holds cycle pairs)
        if (r_cycle.size() != log_T) throw std::invalid_argument("Stage4Prover: r_cycle length");  // error.InvalidRCycleLength
        current_T = T;
        num_rounds = LOG_K + log_T;
        s_ = openRegistersSession(steps, log_T, gamma);
        std::vector<Fr> be(r_cycle.rbegin(), r_cycle.rend());  // :279-292: computeEqEvalsBE of the reversed point
        std::vector<Fr> eq = EqPolynomial::evalsSliceWithScaling(be, nullptr);
        int rc = zg_rrw_set_eq(s_, reinterpret_cast<const uint64_t *>(eq.data()), eq.size());
        if (rc != ZG_OK) { zg_rrw_close(s_); check(rc, "zg_rrw_set_eq"); }
    }
    Stage4Prover(const Stage4Prover &) = delete;
    Stage4Prover &operator=(const Stage4Prover &) = delete;
    ~Stage4Prover() { zg_rrw_close(s_); }
    std::array<Fr, 4> computeRoundEvals(size_t round, const Fr & /* current_claim: not read, p(1) comes from the tables */) {
        std::array<Fr, 4> e;
        if (round < log_T) {
            check(zg_rrw_round_cycle(s_, e[0].limbs, e[1].limbs, e[2].limbs, e[3].limbs), "zg_rrw_round_cycle");
        } else {
            check(zg_rrw_round_address(s_, e[0].limbs, e[1].limbs, e[2].limbs), "zg_rrw_round_address");
            Fr three = Fr::fromU64(3);
            e[3] = e[0].sub(three.mul(e[1])).add(three.mul(e[2]));  // quadratic in the register variable
        }
        return e;
    }
    std::array<Fr, 4> computeRoundPolynomial(size_t round, const Fr &current_claim) {  // :731-758 -> c0..c3
        auto e = computeRoundEvals(round, current_claim);
        static const std::array<Fr, 2> inv = [] {  // 1/6 and 1/2, once
            std::array<Fr, 2> r;
            Fr::fromU64(6).inverse(r[0]);
            Fr::fromU64(2).inverse(r[1]);
            return r;
        }();
        const Fr &six_inv = inv[0], &two_inv = inv[1];
        Fr three = Fr::fromU64(3);
        Fr c3 = Fr::zero().sub(e[0]).add(e[1].mul(three)).sub(e[2].mul(three)).add(e[3]).mul(six_inv);
        Fr c2 = e[0].mul(Fr::fromU64(2)).sub(e[1].mul(Fr::fromU64(5))).add(e[2].mul(Fr::fromU64(4))).sub(e[3]).mul(two_inv);
        return {e[0], e[1].sub(e[0]).sub(c2).sub(c3), c2, c3};
    }
    void bindChallenge(size_t round, const Fr &challenge) {  // :779-839
        if (round < log_T) {
            check(zg_rrw_bind_cycle(s_, challenge.limbs), "zg_rrw_bind_cycle");
            current_T /= 2;
        } else {
            check(zg_rrw_bind_address(s_, challenge.limbs), "zg_rrw_bind_address");
            current_K /= 2;
        }
    }
    Stage4GruenProver::FinalClaims getFinalClaims() {  // :845-863
        Fr f[7];
        check(zg_rrw_final(s_, reinterpret_cast<uint64_t *>(f)), "zg_rrw_final");
        return Stage4GruenProver::FinalClaims{f[0], f[3], f[4], f[1], f[5]};
    }

private:
    zg_rrw_t s_ = nullptr;
};

class InstructionLookupsClaimReductionProver {
public:
    Fr current_claim;
    InstructionLookupsClaimReductionProver(const std::vector<Fr> &eq_evals, const std::vector<Fr> &lookup_outputs, const std::vector<Fr> &left_operands,
                                           const std::vector<Fr> &right_operands, const Fr &gamma, const Fr &claim)
        : current_claim(claim), s_({&eq_evals, &lookup_outputs, &left_operands, &right_operands}), coeff_{Fr::one(), gamma, gamma.mul(gamma)} {
        s_.setPoints(0b0101);  // only s(0) and s(2) are read
    }
    std::array<Fr, 4> computeRoundPolynomialCubic() {  // :146-200: s0, s2 from the tables; s1 = claim - s0; s3 = s0 - 3 s1 + 3 s2
        auto ev = s_.roundEvals({0}, {1, 2, 3}, coeff_);
        Fr s1 = current_claim.sub(ev[0]), three = Fr::fromU64(3);
        return {ev[0], s1, ev[2], ev[0].sub(s1.mul(three)).add(ev[2].mul(three))};
    }
    void bindChallenge(const Fr &c) { s_.bind(c); }
    void updateClaim(const std::array<Fr, 4> &evals, const Fr &c) { current_claim = cubicAtPoint(evals, c); }
    std::vector<Fr> finalValues() { return s_.final(); }  // eq, lookup_output, left_operand, right_operand

private:
    ProductSumcheckSession s_;
    std::vector<Fr> coeff_;
};

// RafEvaluationProver's loop (src/zkvm/ram/raf_checking.zig:262-470) over RaPolynomial's table in a LOW_PAIR session
class RafEvaluationProver {
public:
    Fr current_claim;
    RafEvaluationProver(const std::vector<Fr> &ra_evals, uint64_t start_address, const Fr &initial_claim)
        : current_claim(initial_claim), base_(Fr::fromU64(start_address)) {
        check(zg_sumcheck_open(reinterpret_cast<const uint64_t *>(ra_evals.data()), ra_evals.size(), ZG_SC_LOW_PAIR, &s_), "zg_sumcheck_open");
    }
    ~RafEvaluationProver() { zg_sumcheck_close(s_); }
    RafEvaluationProver(const RafEvaluationProver &) = delete;
    std::array<Fr, 4> computeRoundPolynomialCubic() {  // :335-410: s(0), s(2) in one pass on the device
        Fr s0, s2;
        check(zg_sumcheck_raf_round(s_, base_.limbs, power_, s0.limbs, s2.limbs), "zg_sumcheck_raf_round");
        Fr s1 = current_claim.sub(s0), three = Fr::fromU64(3);
        return {s0, s1, s2, s0.sub(s1.mul(three)).add(s2.mul(three))};
    }
    void updateClaim(const std::array<Fr, 4> &evals, const Fr &c) { current_claim = cubicAtPoint(evals, c); }  // :420-445
    void bindChallenge(const Fr &c) {  // RaPolynomial.bind (:162-174) + the bound-address bookkeeping (:413-417)
        check(zg_sumcheck_bind(s_, c.limbs), "zg_sumcheck_bind");
        base_ = base_.add(c.mul(Fr::fromU64(power_)));
        power_ *= 2;
    }

private:
    zg_sc_t s_ = nullptr;
    Fr base_;
    uint64_t power_ = 8;
};

// SumcheckInstance / BatchedSumcheckProver / generateBatchedProof (src/zkvm/batched_sumcheck.zig:34-430)
struct SumcheckInstance {
    size_t num_rounds, degree;
    Fr input_claim;
    std::function<std::array<Fr, 4>(size_t)> computeRoundPoly;
    std::function<void(const Fr &)> bindChallenge;
};

class BatchedSumcheckProver {
public:
    std::vector<SumcheckInstance> instances;
    std::vector<Fr> batching_coeffs, challenges;
    size_t max_num_rounds = 0, current_round = 0;
    Fr current_claim = Fr::zero();
    // The constant an instance contributes before its first round: coeff * claim * 2^(start - round - 1) is what the loop `zolt prove`
    // runs uses (src/zkvm/proof_converter.zig:3330-3343, Jolt's rule — twice the constant is the instance's share of the claim, so
    // s(0) + s(1) = claim in every round); batched_sumcheck.zig:208-212 itself writes 2^(start - round), which no caller in the
    // reference reaches and which breaks that identity. false selects the file's own formula.
    bool proof_converter_scaling = true;

    void addInstance(SumcheckInstance inst) {  // :115-121
        max_num_rounds = std::max(max_num_rounds, inst.num_rounds);
        instances.push_back(std::move(inst));
    }
    void setupBatching(Blake2bTranscript &transcript) {  // :127-186
        for (auto &inst : instances) transcript.appendScalar(inst.input_claim);
        for (size_t i = 0; i < instances.size(); i++) batching_coeffs.push_back(transcript.challengeScalarFull());
        Fr batched = Fr::zero();
        for (size_t i = 0; i < instances.size(); i++)
            batched = batched.add(scaled(instances[i].input_claim, max_num_rounds - instances[i].num_rounds).mul(batching_coeffs[i]));
        current_claim = batched;
    }
    std::array<Fr, 4> combinedEvals() {  // :193-222
        std::array<Fr, 4> comb = {Fr::zero(), Fr::zero(), Fr::zero(), Fr::zero()};
        for (size_t i = 0; i < instances.size(); i++) {
            size_t start = max_num_rounds - instances[i].num_rounds;
            if (current_round >= start) {
                auto ev = instances[i].computeRoundPoly(current_round - start);
                for (int j = 0; j < 4; j++) comb[j] = comb[j].add(ev[j].mul(batching_coeffs[i]));
            } else {
                Fr w = scaled(instances[i].input_claim, start - current_round - (proof_converter_scaling ? 1 : 0)).mul(batching_coeffs[i]);
                for (int j = 0; j < 4; j++) comb[j] = comb[j].add(w);
            }
        }
        return comb;
    }
    std::array<Fr, 3> computeRoundPolynomial() { return evalsToCompressed(combinedEvals()); }
    void bindChallenge(const Fr &challenge) {  // :229-241
        challenges.push_back(challenge);
        for (auto &inst : instances)
            if (current_round >= max_num_rounds - inst.num_rounds) inst.bindChallenge(challenge);
        current_round++;
    }
    void updateClaim(const std::array<Fr, 4> &round_evals, const Fr &challenge) { current_claim = cubicAtPoint(round_evals, challenge); }

private:
    static Fr scaled(Fr v, size_t doublings) {
        for (size_t k = 0; k < doublings; k++) v = v.add(v);
        return v;
    }
};

struct BatchedSumcheckProof {
    std::vector<std::array<Fr, 3>> round_polys;
    std::vector<Fr> challenges;
    Fr final_claim;
};

// [s(0), s(1), s(2), s(3)] from the compressed [c0, c2, c3] and the claim (:380-400)
inline std::array<Fr, 4> decompressRoundPoly(const std::array<Fr, 3> &c, const Fr &claim) {
    Fr c1 = claim.sub(c[0]).sub(c[0]).sub(c[1]).sub(c[2]);
    return {c[0], c[0].add(c1).add(c[1]).add(c[2]),
            c[0].add(c1.mul(Fr::fromU64(2))).add(c[1].mul(Fr::fromU64(4))).add(c[2].mul(Fr::fromU64(8))),
            c[0].add(c1.mul(Fr::fromU64(3))).add(c[1].mul(Fr::fromU64(9))).add(c[2].mul(Fr::fromU64(27)))};
}

inline BatchedSumcheckProof generateBatchedProof(BatchedSumcheckProver &prover, Blake2bTranscript &transcript) {  // :306-430
    BatchedSumcheckProof proof;
    for (size_t k = 0; k < prover.max_num_rounds; k++) {
        auto comp = prover.computeRoundPolynomial();
        proof.round_polys.push_back(comp);
        transcript.appendMessage("UniPoly_begin");
        for (const Fr &c : comp) transcript.appendScalar(c);
        transcript.appendMessage("UniPoly_end");
        Fr challenge = transcript.challengeScalar();
        proof.challenges.push_back(challenge);
        prover.updateClaim(decompressRoundPoly(comp, prover.current_claim), challenge);
        prover.bindChallenge(challenge);
    }
    proof.final_claim = prover.current_claim;
    return proof;
}

// ---------------------------------------------------------------- LassoProver (src/zkvm/lasso/prover.zig:80-551)
// The sumcheck over eq_evals on ONE device session: address rounds = zg_sumcheck_bit_round / bit_bind, cycle rounds = the session's
// HIGH_HALF round_sums / bind. The prefix-suffix structures the reference binds alongside (:402-404) do not enter the round
// polynomials and are not mirrored. Lookup indices: u128 as two little-endian u64 words.
class LassoProver {
public:
    size_t log_T, log_K, round = 0, eq_evals_len;
    Fr current_claim;
    std::vector<Fr> challenges;

    LassoProver(const std::vector<unsigned __int128> &lookup_indices, size_t log_T_, size_t log_K_, const std::vector<Fr> &r_reduction)
        : log_T(log_T_), log_K(log_K_), eq_evals_len(size_t(1) << log_T_), n_(lookup_indices.size()) {
        if (r_reduction.size() != log_T || n_ > eq_evals_len) throw std::invalid_argument("LassoProver: r_reduction.len != log_T");
        // SplitEqPolynomial.getEq (src/zkvm/lasso/split_eq.zig:113-168): both halves are built LSB-first = the MSB-first eq table of
        // each half reversed; eq_evals[j] for j >= num_cycles is zero (prover.zig:160-164)
        size_t outer = log_T / 2;
        std::vector<Fr> point;
        for (size_t i = outer; i-- > 0;) point.push_back(r_reduction[i]);
        for (size_t i = log_T; i-- > outer;) point.push_back(r_reduction[i]);
        {
            DeviceMem d_tab(eq_evals_len * 32);
            check(zg_fr_eq_table_dev(reinterpret_cast<const uint64_t *>(point.data()), log_T, nullptr, d_tab.u64(), nullptr), "zg_fr_eq_table_dev");
            if (n_ < eq_evals_len) {
                std::vector<uint64_t> zeros((eq_evals_len - n_) * 4, 0);
                check(zg_memcpy_h2d(d_tab.u64() + 4 * n_, zeros.data(), zeros.size() * 8), "zg_memcpy_h2d");
            }
            check(zg_sumcheck_open_dev(d_tab.u64(), eq_evals_len, ZG_SC_HIGH_HALF, nullptr, &s_), "zg_sumcheck_open_dev");
            check(zg_sync(), "zg_sync");  // the session copied the table: the staging buffer may go
        }
        std::vector<uint64_t> words(2 * (n_ ? n_ : 1), 0);
        for (size_t j = 0; j < n_; j++) {
            words[2 * j] = (uint64_t)lookup_indices[j];
            words[2 * j + 1] = (uint64_t)(lookup_indices[j] >> 64);
        }
        try {
            d_idx_.alloc(words.size() * 8);
            check(zg_memcpy_h2d(d_idx_.p, words.data(), words.size() * 8), "zg_memcpy_h2d");
            current_claim = total();  // :166-171
        } catch (...) {
            zg_sumcheck_close(s_);
            throw;
        }
    }
    ~LassoProver() { zg_sumcheck_close(s_); }
    LassoProver(const LassoProver &) = delete;
    bool isAddressPhase() const { return round < log_K; }
    bool isComplete() const { return round >= log_K + log_T; }
    UniPoly computeRoundPolynomial() {  // :262-345 -> [sum_0, sum_1 - sum_0, 0]
        Fr s0, s1;
        if (isAddressPhase()) {
            check(zg_sumcheck_bit_round(s_, d_idx_.u64(), n_, (unsigned)round, s0.limbs, s1.limbs), "zg_sumcheck_bit_round");
        } else if (eq_evals_len <= 1) {
            check(zg_sumcheck_final(s_, s0.limbs), "zg_sumcheck_final");
            return UniPoly{{s0, Fr::zero(), Fr::zero()}};
        } else {
            check(zg_sumcheck_round_sums(s_, s0.limbs, s1.limbs), "zg_sumcheck_round_sums");
        }
        return UniPoly{{s0, s1.sub(s0), Fr::zero()}};
    }
    void receiveChallenge(const Fr &challenge) {  // :352-453
        challenges.push_back(challenge);
        if (isAddressPhase()) {
            check(zg_sumcheck_bit_bind(s_, d_idx_.u64(), n_, (unsigned)round, challenge.limbs, current_claim.limbs),
                  "zg_sumcheck_bit_bind");
        } else if (eq_evals_len > 1) {
            check(zg_sumcheck_bind(s_, challenge.limbs), "zg_sumcheck_bind");
            eq_evals_len /= 2;
            current_claim = total();
        }
        round++;
    }
    Fr getFinalEval() const {  // :458-462: expanding_v.get(0) = prod over the address challenges of (1 - r) (expanding_table.zig:83-99)
        Fr acc = Fr::one();
        for (size_t i = 0; i < log_K && i < challenges.size(); i++) acc = acc.mul(Fr::one().sub(challenges[i]));
        return acc;
    }
    static Fr deriveChallenge(const UniPoly &round_poly, size_t round_index) {  // :533-551
        uint64_t hash = 0x9e3779b97f4a7c15ULL;
        hash ^= (uint64_t)round_index;
        hash *= 0xff51afd7ed558ccdULL;
        for (const Fr &c : round_poly.coeffs)
            for (int l = 0; l < 4; l++) {
                hash ^= c.limbs[l];
                hash *= 0xc4ceb9fe1a85ec53ULL;
            }
        hash ^= hash >> 33;
        return Fr::fromU64(hash);
    }

private:
    Fr total() {
        Fr a, b;
        if (zg_sumcheck_len(s_) >= 2) {
            check(zg_sumcheck_round_sums(s_, a.limbs, b.limbs), "zg_sumcheck_round_sums");
            return a.add(b);
        }
        check(zg_sumcheck_final(s_, a.limbs), "zg_sumcheck_final");
        return a;
    }
    size_t n_;
    zg_sc_t s_ = nullptr;
    DeviceMem d_idx_;
};

struct LassoProof {  // :470-492
    std::vector<UniPoly> round_polys;
    Fr final_eval;
    std::vector<Fr> challenges;
};

inline LassoProof runLassoProver(const std::vector<unsigned __int128> &lookup_indices, size_t log_T, size_t log_K,
                                 const std::vector<Fr> &r_reduction) {  // :495-530
    LassoProver prover(lookup_indices, log_T, log_K, r_reduction);
    LassoProof proof;
    size_t round = 0;
    while (!prover.isComplete()) {
        proof.round_polys.push_back(prover.computeRoundPolynomial());
        prover.receiveChallenge(LassoProver::deriveChallenge(proof.round_polys.back(), round));
        round++;
    }
    proof.final_eval = prover.getFinalEval();
    proof.challenges = prover.challenges;
    return proof;
}

}  // namespace zolt
