// stage_provers.hpp — the prover fold sites of stages 1-6 (Spartan outer, Stage 3, RAM read/write checking, Stage 4, batched driver).
// Part of zolt_host.hpp (the C++ host mirror over include/zolt_gpu.h); included by it, after the parts it depends on.
#pragma once
#ifndef ZOLT_HOST_UMBRELLA
#error "include zolt_host.hpp"
#endif
namespace zolt {

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
inline void highHalfRounds(zg_sc_t s, size_t num_rounds, Transcript &transcript, const std::string &label, StageRoundsResult &out) {
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
inline void highHalfRounds(const std::vector<Fr> &evals, size_t num_rounds, Transcript &transcript, const std::string &label, StageRoundsResult &out) {
    zg_sc_t s = nullptr;
    check(zg_sumcheck_open(reinterpret_cast<const uint64_t *>(evals.data()), evals.size(), ZG_SC_HIGH_HALF, &s), "zg_sumcheck_open");
    highHalfRounds(s, num_rounds, transcript, label, out);
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
    // last value per ADDRESS (the reference keys its map by address, not by word): every address that passes the range filter lies in
    // [start, start + 8k), so the map is a flat array over byte offsets — a hash map cost 2 ms per 10^5 accesses here
    // 64 * k bytes, value-initialised on every call: 4 MB at log_k = 16 (the captured run) but 256 MB of memset and page faults at 2^22,
    // far more than the map costs a sparse trace — flat only while the array stays within 16 MB or the trace really fills it
    const bool flat = k <= (size_t(1) << 18) || (k <= (size_t(1) << 22) && accesses.size() >= k);
    std::vector<uint64_t> last_flat(flat ? 8 * k : 0, 0);
    std::unordered_map<uint64_t, uint64_t> last_map;
    auto last = [&](uint64_t address) -> uint64_t & { return flat ? last_flat[address - start_address] : last_map[address]; };
    for (auto &kv : initial_ram)
        if (kv.first >= start_address && (kv.first - start_address) / 8 < k) last(kv.first) = kv.second;
    std::vector<uint32_t> slot(w.n, ~0u);
    w.cycle.reserve(accesses.size()); w.word.reserve(accesses.size()); w.pre.reserve(accesses.size()); w.post.reserve(accesses.size());
    for (const MemoryAccess &a : accesses) {
        if (!a.is_write || a.address < start_address || (a.address - start_address) / 8 >= k || a.timestamp >= trace_len) continue;
        uint64_t &cur = last(a.address);  // an address not seen before reads 0
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
// rd: the destination register of every cycle, (instruction >> 7) & 31 — what the stage reads of the trace
inline StageRoundsResult proveStage5(const uint8_t *rd, size_t cycles, size_t log_t, Transcript &transcript, std::vector<Fr> *r_register_out = nullptr) {
    std::vector<Fr> r_register(5);
    for (auto &x : r_register) x = transcript.challengeScalar("r_register");
    for (size_t i = 0; i < log_t; i++) (void)transcript.challengeScalar("r_cycle_reg");
    if (r_register_out) *r_register_out = r_register;
    StageRoundsResult out;
    if (cycles == 0) { out.skipped = true; return out; }
    const size_t num_rounds = cycles <= 1 ? 0 : log2Ceil(cycles);
    Fr table[32];
    for (unsigned reg = 0; reg < 32; reg++) table[reg] = computeRegEq(r_register, reg);
    // eq_evals[j] = table[rd of cycle j] (:880-900), zero past the trace: one byte per cycle crosses, the 32-entry table is looked up on the
    // device (ZG_COL_LUT) — the 2^log_t-element table of field elements is never built on the host
    const zg_col_t col{ZG_COL_LUT, 1, 32, rd, table};
    zg_sc_t s = nullptr;
    check(zg_sumcheck_open_column(&col, cycles, size_t(1) << num_rounds, ZG_SC_HIGH_HALF, &s), "zg_sumcheck_open_column");
    highHalfRounds(s, num_rounds, transcript, "reg_eval_round", out);
    return out;
}
inline StageRoundsResult proveStage5(const std::vector<uint32_t> &instructions, size_t log_t, Transcript &transcript, std::vector<Fr> *r_register_out = nullptr) {
    std::vector<uint8_t> rd(instructions.size());
    for (size_t j = 0; j < instructions.size(); j++) rd[j] = (uint8_t)((instructions[j] >> 7) & 31);
    return proveStage5(rd.data(), rd.size(), log_t, transcript, r_register_out);
}
// proveStage6 (:990-1112): booleanity — violation_evals = 0 for every step of a valid trace (:1024-1033)
inline StageRoundsResult proveStage6(size_t trace_len, Transcript &transcript, Fr *bool_challenge_out = nullptr) {
    Fr bc = transcript.challengeScalar("booleanity");
    if (bool_challenge_out) *bool_challenge_out = bc;
    StageRoundsResult out;
    if (trace_len == 0) { out.skipped = true; return out; }
    const size_t num_rounds = trace_len <= 1 ? 0 : log2Ceil(trace_len);
    // violation_evals = 0 for every step of a valid trace (:1024-1033): the session's table is cleared on the device, nothing is uploaded
    const zg_col_t col{ZG_COL_ZERO, 0, 0, nullptr, nullptr};
    zg_sc_t s = nullptr;
    check(zg_sumcheck_open_column(&col, 0, size_t(1) << num_rounds, ZG_SC_HIGH_HALF, &s), "zg_sumcheck_open_column");
    highHalfRounds(s, num_rounds, transcript, "bool_round", out);
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

    // ready rows of field elements (1376 bytes per cycle cross PCIe) ...
    StreamingOuterProver(const std::vector<CycleInputs> &cycle_witnesses, const std::vector<Fr> &tau, const Fr *lagrange_tau_r0 = nullptr)
        : StreamingOuterProver(cycle_witnesses.empty() ? std::shared_ptr<CycleWitnessMatrix>() : CycleWitnessMatrix::fromWitnesses(cycle_witnesses.data(), cycle_witnesses.size()),
                               tau, lagrange_tau_r0) {}
    // ... or the shared device-resident matrix (witness.hpp: widened on the device from integer trace columns, 156 bytes per cycle)
    StreamingOuterProver(std::shared_ptr<CycleWitnessMatrix> rows, const std::vector<Fr> &tau, const Fr *lagrange_tau_r0 = nullptr)
        : split_eq(std::vector<Fr>(tau.begin(), tau.end() - (tau.empty() ? 0 : 1)), lagrange_tau_r0), num_cycles_(rows ? rows->num_cycles : 0),
          tau_high_(tau.empty() ? Fr::zero() : tau.back()), d_rows_(std::move(rows)) {
        if (num_cycles_ == 0) throw std::invalid_argument("StreamingOuterProver: empty trace");  // error.EmptyTrace
        while (padded_trace_len < num_cycles_) padded_trace_len <<= 1, num_cycle_vars++;
        if (tau.size() != num_cycle_vars + 2) throw std::invalid_argument("StreamingOuterProver: tau has num_cycle_vars + 2 challenges");
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
        check(zg_fr_rows_affine_prodsum_dev(d_rows_->u64(), std::min(num_cycles_, padded_trace_len), r1cs::NUM_INPUTS, 0, reinterpret_cast<const uint64_t *>(m.data()), 18,
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
        check(zg_fr_rows_affine_dev(d_rows_->u64(), std::min(num_cycles_, padded_trace_len), r1cs::NUM_INPUTS, 0, reinterpret_cast<const uint64_t *>(m.data()), 2, 2,
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
    std::shared_ptr<CycleWitnessMatrix> d_rows_;  // shared with the other stages that read the witness matrix
    DeviceMem d_out_, d_in_;
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
        : current_claim(initial_claim), base_(Fr::fromU64(start_address)), start_(start_address) {
        check(zg_sumcheck_open(reinterpret_cast<const uint64_t *>(ra_evals.data()), ra_evals.size(), ZG_SC_LOW_PAIR, &s_), "zg_sumcheck_open");
    }
    // the claim is computeInitialClaim() of the table (what prover.zig's Stage 2 passes in, :312-321)
    RafEvaluationProver(const std::vector<Fr> &ra_evals, uint64_t start_address) : current_claim(Fr::zero()), base_(Fr::fromU64(start_address)), start_(start_address) {
        check(zg_sumcheck_open(reinterpret_cast<const uint64_t *>(ra_evals.data()), ra_evals.size(), ZG_SC_LOW_PAIR, &s_), "zg_sumcheck_open");
        current_claim = computeInitialClaim();
    }
    ~RafEvaluationProver() { zg_sumcheck_close(s_); }
    RafEvaluationProver(const RafEvaluationProver &) = delete;
    Fr computeInitialClaim() {  // sum_k ra(k) * F.fromU64(start + 8 k), one pass over the resident table (before the first bind)
        Fr c;
        check(zg_sumcheck_raf_claim(s_, start_, 8, c.limbs), "zg_sumcheck_raf_claim");
        return c;
    }
    std::array<Fr, 4> computeRoundPolynomialCubic() {  // :335-410: s(0), s(2) in one pass on the device
        Fr s0, s2;
        check(zg_sumcheck_raf_round(s_, base_.limbs, power_, s0.limbs, s2.limbs), "zg_sumcheck_raf_round");
        Fr s1 = current_claim.sub(s0), three = Fr::fromU64(3);
        return {s0, s1, s2, s0.sub(s1.mul(three)).add(s2.mul(three))};
    }
    void updateClaim(const std::array<Fr, 4> &evals, const Fr &c) { current_claim = cubicAtPoint(evals, c); }  // :420-445
    Fr getFinalClaim() {  // :448-450 -> RaPolynomial.finalClaim (:179-185): evals[0] at ANY point of the protocol, zero only for an empty table
        Fr v = Fr::zero();
        const size_t len = zg_sumcheck_len(s_);
        if (len == 1) check(zg_sumcheck_final(s_, v.limbs), "zg_sumcheck_final");
        else if (len > 1) {  // read before the last bind: entry 0 of the current table (32 bytes over PCIe)
            const uint64_t zero = 0;
            check(zg_sumcheck_gather(s_, &zero, 1, v.limbs), "zg_sumcheck_gather");
        }
        return v;
    }
    void bindChallenge(const Fr &c) {  // RaPolynomial.bind (:162-174) + the bound-address bookkeeping (:413-417)
        check(zg_sumcheck_bind(s_, c.limbs), "zg_sumcheck_bind");
        base_ = base_.add(c.mul(Fr::fromU64(power_)));
        power_ *= 2;
    }

private:
    zg_sc_t s_ = nullptr;
    Fr base_;
    uint64_t start_ = 0, power_ = 8;
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

}  // namespace zolt
