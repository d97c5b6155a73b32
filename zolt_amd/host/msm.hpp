// msm.hpp — zolt.msm: AffinePoint, MSM, BatchMSM, ParallelMSM, Dory row commitments, device / sharded base handles.
// Part of zolt_host.hpp (the C++ host mirror over include/zolt_gpu.h); included by it, after the parts it depends on.
#pragma once
#ifndef ZOLT_HOST_UMBRELLA
#error "include zolt_host.hpp"
#endif
namespace zolt {

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
    // HyperKZG.setup's G1 side built on the device (zg_hyperkzg_setup): tau^i * base for i < n; the points come back only when asked for
    DeviceBases(const AffinePoint &base, const Fr &tau, size_t n, std::vector<uint64_t> *xy_out = nullptr, std::vector<uint8_t> *inf_out = nullptr,
                const zg_msm_config *cfg = nullptr) : n_(n) {
        uint64_t b[8];
        std::memcpy(b, base.x.limbs, 32);
        std::memcpy(b + 4, base.y.limbs, 32);
        if (xy_out) xy_out->resize(8 * n);
        if (inf_out) inf_out->resize(n);
        check(zg_hyperkzg_setup(b, tau.limbs, n, cfg, xy_out ? xy_out->data() : nullptr, inf_out ? inf_out->data() : nullptr, &h_), "zg_hyperkzg_setup");
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
    // scalars given as u64 machine words: = msm(F.fromU64 of every word), 8 bytes per scalar across PCIe (zg_msm_g1_u64)
    AffinePoint msmU64(const uint64_t *values, size_t n, size_t off = 0) const {
        uint64_t out[8];
        uint8_t inf = 0;
        check(zg_msm_g1_u64(h_, off, n, values, out, &inf), "zg_msm_g1_u64");
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

}  // namespace zolt
