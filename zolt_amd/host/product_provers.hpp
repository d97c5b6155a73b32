// product_provers.hpp — product-form sessions (zg_psc_*) and the provers built on them.
// Part of zolt_host.hpp (the C++ host mirror over include/zolt_gpu.h); included by it, after the parts it depends on.
#pragma once
#ifndef ZOLT_HOST_UMBRELLA
#error "include zolt_host.hpp"
#endif
namespace zolt {

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

}  // namespace zolt
