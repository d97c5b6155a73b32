// poly.hpp — zolt.poly: UniPoly, DensePolynomial, EqPolynomial, GruenSplitEqPolynomial.
// Part of zolt_host.hpp (the C++ host mirror over include/zolt_gpu.h); included by it, after the parts it depends on.
#pragma once
#ifndef ZOLT_HOST_UMBRELLA
#error "include zolt_host.hpp"
#endif
namespace zolt {

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

}  // namespace zolt
