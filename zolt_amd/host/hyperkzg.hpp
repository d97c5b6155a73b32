// hyperkzg.hpp — zolt.poly.commitment.HyperKZG: setup, commit, batchCommit, open, batchOpen.
// Part of zolt_host.hpp (the C++ host mirror over include/zolt_gpu.h); included by it, after the parts it depends on.
#pragma once
#ifndef ZOLT_HOST_UMBRELLA
#error "include zolt_host.hpp"
#endif
namespace zolt {

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
    // setup(max_degree) — :174-213, tau = 0x12345678. The powers of tau, the fixed-base batch (:194-199: every product has the same base)
    // and the MSM handle with its table of multiples are built ON THE DEVICE (zg_hyperkzg_setup); host_points = false leaves
    // powers_of_tau_g1 empty for a prover that only commits and opens (the points are never read on the host then: 64 bytes per power
    // stay off PCIe and out of the struct conversions, 131 -> ~35 ms at 2^20 powers)
    // cfg: the MSM plan of the key's handle — expected_uses = 1..15 skips the table of multiples (a key that serves ONE proof: three
    // commits and an open are fewer MSMs than the table's break-even, bench.py config.breakeven_msms); nullptr = an SRS that lives on
    static SetupParams setup(size_t max_degree, bool host_points = true, const zg_msm_config *cfg = nullptr) {
        SetupParams p;
        p.g1 = AffinePoint::generator();
        p.max_degree = max_degree;
        std::vector<uint64_t> xy;
        std::vector<uint8_t> inf;
        p.device.reset(new DeviceBases(p.g1, Fr::fromU64(0x12345678), max_degree, host_points ? &xy : nullptr, host_points ? &inf : nullptr, cfg));
        if (host_points) {
            p.powers_of_tau_g1.reserve(max_degree);
            for (size_t i = 0; i < max_degree; i++) p.powers_of_tau_g1.push_back(unpack_point(&xy[8 * i], inf[i]));
        }
        return p;
    }
    static size_t srsLen(const SetupParams &params) { return params.powers_of_tau_g1.empty() ? params.max_degree : params.powers_of_tau_g1.size(); }
    static Commitment commit(const SetupParams &params, const std::vector<Fr> &evals) {  // :239-255
        if (evals.empty()) return Commitment{AffinePoint::identity()};
        size_t n = evals.size() < srsLen(params) ? evals.size() : srsLen(params);
        return Commitment{params.device->msm(evals.data(), n)};
    }
    // commit to a polynomial whose evaluations are F.fromU64 of machine words (commitBytecode / commitMemory / commitRegisters,
    // src/zkvm/mod.zig:1518-1617 build exactly such vectors): the words cross as they are
    static Commitment commitU64(const SetupParams &params, const uint64_t *values, size_t count) {
        if (count == 0) return Commitment{AffinePoint::identity()};
        size_t n = count < srsLen(params) ? count : srsLen(params);
        return Commitment{params.device->msmU64(values, n)};
    }
    static Commitment commitU64(const SetupParams &params, const std::vector<uint64_t> &values) { return commitU64(params, values.data(), values.size()); }
    static Commitment commitU64(const SetupParams &params, const PinnedWords &values) { return commitU64(params, values.data(), values.size()); }
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
        size_t srs = srsLen(params);
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

}  // namespace zolt
