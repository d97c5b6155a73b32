// lasso.hpp — LassoProver.
// Part of zolt_host.hpp (the C++ host mirror over include/zolt_gpu.h); included by it, after the parts it depends on.
#pragma once
#ifndef ZOLT_HOST_UMBRELLA
#error "include zolt_host.hpp"
#endif
namespace zolt {

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
        // (low word, high word) per index, assembled in the thread's pinned staging buffer: no 16 MB vector to zero-fill and page in
        const size_t n_words = 2 * (n_ ? n_ : 1);
        uint64_t *words = static_cast<uint64_t *>(PinnedStage::get(n_words * 8));
        words[0] = words[1] = 0;
        for (size_t j = 0; j < n_; j++) {
            words[2 * j] = (uint64_t)lookup_indices[j];
            words[2 * j + 1] = (uint64_t)(lookup_indices[j] >> 64);
        }
        try {
            d_idx_.alloc(n_words * 8);
            check(zg_memcpy_h2d(d_idx_.p, words, n_words * 8), "zg_memcpy_h2d");
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
