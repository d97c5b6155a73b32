// psc.hip — product-form sumcheck sessions on gfx950: k multilinear evaluation tables that are folded together (LowToHigh, adjacent
// pairs) and whose round polynomial is a sum over the pairs g of
//       prod_{j<p} A_j(t)  *  L(t),        A(t) = lo + t (hi - lo),   L(t) = sum_{m<q} c_m B_m(t)   (q = 0: L = 1)
// evaluated at t = 0, 1, 2, 3, or — Gruen's form — as (t0, t_inf) = sum_g w(g) (prod lo, prod (hi - lo)) with w(g) the split-eq
// weight E_out[g >> bits] * E_in[g & mask].
//
// Reference loops replaced (paths under /root/reference), all the same two shapes with different table counts:
//   ValEvaluationProver.computeRoundPolynomial / bindChallengeWithPoly       src/zkvm/ram/val_evaluation.zig:554-628   (inc * wa * lt)
//   ValFinalProver.computeRoundPolynomial / bindChallengeWithPoly            src/zkvm/ram/val_final.zig:149-200        (inc * wa)
//   OutputSumcheckProver.computeRoundPolynomial / bind                       src/zkvm/ram/output_check.zig:375-470     (eq * io * (vf - vio))
//   InstructionLookupsClaimReduction.computeRoundPolynomialCubic / bind      src/zkvm/claim_reductions/instruction_lookups.zig:146-240
//                                                                            (eq * (out + gamma left + gamma^2 right))
//   ProductVirtualRemainderProver.computeRoundPolynomial / bindChallenge     src/zkvm/spartan/product_remainder.zig:269-420 (Gruen, left * right)
// Exact modular arithmetic with canonical outputs: 2*f1 - f0, f0 + 2 (f1 - f0) and f(1) + (f(1) - f(0)) are the same field value, sums
// commute, so every evaluation order gives the reference's bytes.
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "common.hip.h"
#include "field.hip.h"
#include "fp29.hip.h"
#include "sc_common.hip.h"

#define ZG_PSC_MAX_TABLES 12
#define ZG_PSC_MAX_TERMS 4
#define ZG_PSC_MAX_FACTORS 4

namespace zg {

constexpr unsigned PSC_MAX_BLOCKS = 4096;
constexpr size_t PSC_COUNTER_OFF = 16 * (size_t)PSC_MAX_BLOCKS;     // u64 words: block quadruples, then the arrival counters (sc_arrive)
constexpr size_t PSC_MISC_BYTES = PSC_COUNTER_OFF * 8 + SC_COUNTER_BYTES;
constexpr int PSC_FLAG = 24;                                        // h_pin: 16 words of values, sequence word at 24

// compile-time loop: the table loops carry register arrays (prescaled coefficients) and must not stay loops
template <int I, int N, class F>
ZG_DEV void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Four chain values (t = 0..3) summed lazily over the pairs a thread owns, limb-wise in 32-bit words (fp29.hip.h: fr29_sum_reduce): a
// carry pass every four additions, one reduction into the canonical accumulators every 64 — before the sum could outgrow the
// multiplier's input range. 36 registers instead of 72 for the 64-bit limb sums: these kernels live on the 256-register line.
struct ChainAcc4 {
    F29 a[4];
    unsigned cnt;
    ZG_DEV void init() {
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int i = 0; i < 9; i++) a[t].l[i] = 0;
        cnt = 0;
    }
    ZG_DEV void add(const F29 (&w)[4], Fr (&e)[4]) {
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int i = 0; i < 9; i++) a[t].l[i] += w[t].l[i];
        ++cnt;
        if ((cnt & 3u) == 0) {
#pragma unroll
            for (int t = 0; t < 4; t++) a[t] = f29_carry(a[t]);
            if (cnt == FR29_ACC_MAX) flush(e);
        }
    }
    ZG_DEV void flush(Fr (&e)[4]) {
        if (cnt == 0) return;
        if (cnt == 1) {  // a single chain value: exact limbs, < 2 r — no product needed (the one-pair-per-thread regime of short tables)
#pragma unroll
            for (int t = 0; t < 4; t++) e[t] = fe_add(e[t], fr29_out(a[t]));
        } else {
#pragma unroll
            for (int t = 0; t < 4; t++) e[t] = fe_add(e[t], fr29_sum_reduce(f29_carry(a[t])));
        }
        init();
    }
};

// two waves per SIMD (at most 256 registers): these kernels issue their loads right before the products that use them, and a second
// resident wave is what overlaps one wave's HBM latency with the other's arithmetic
#ifndef PSC_OCC
#define PSC_OCC __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
struct PscSpec {
    uint32_t prod[ZG_PSC_MAX_FACTORS];  // table indices of the plain factors
    uint32_t lin[ZG_PSC_MAX_FACTORS];   // table indices of the linear combination
    FrArg coeff[ZG_PSC_MAX_FACTORS];    // its coefficients (Montgomery)
    uint32_t points;                    // bit t: the evaluation at t is wanted (zg_psc_set_points; the others are not multiplied out)
};

// The running products at t = 0..3 times the next factor's values lo, hi, 2 hi - lo, 3 hi - 2 lo — only at the points the caller
// reads: several of the reference's provers derive p(1) from the previous claim and p(3) from the others (stage3_prover.zig:2029-2100,
// 2334-2389; claim_reductions/instruction_lookups.zig:146-200), and these kernels are bound by the field products.
ZG_DEV void psc_chain4(F29 (&w)[4], uint32_t pm, const Fr &lo, const Fr &hi) {
    if (pm & 1u) w[0] = fr29_chain_mul(w[0], fr29_in_shift(lo));
    if (pm & 2u) w[1] = fr29_chain_mul(w[1], fr29_in_shift(hi));
    if (pm & 12u) {
        Fr d = fe_sub(hi, lo), f2 = fe_add(hi, d);
        if (pm & 4u) w[2] = fr29_chain_mul(w[2], fr29_in_shift(f2));
        if (pm & 8u) w[3] = fr29_chain_mul(w[3], fr29_in_shift(fe_add(f2, d)));
    }
}
// A coefficient of the linear combination, prepared once per launch: 1 and -1 (OutputSumcheck's vf - vio, the leading 1 of every
// gamma-power combination) cost an addition instead of a field product.
struct LinCoeff {
    F29 p;
    uint32_t kind;  // 0: general, 1: one, 2: minus one
};
ZG_DEV uint32_t lin_kind(const FrArg &a) {  // scalar work: the coefficient is a kernel argument
    Fr one = Fr::one(), mone = fe_neg(Fr::one());
    uint32_t d1 = 0, d2 = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        d1 |= a.l[i] ^ one.l[i];
        d2 |= a.l[i] ^ mone.l[i];
    }
    return d1 == 0 ? 1u : (d2 == 0 ? 2u : 0u);
}
ZG_DEV LinCoeff lin_prepare(const FrArg &a) {
    Fr c;
#pragma unroll
    for (int i = 0; i < 8; i++) c.l[i] = a.l[i];
    LinCoeff r;
    r.kind = lin_kind(a);
    r.p = fr29_prescale(c);
    return r;
}
ZG_DEV Fr lin_add(const Fr &acc, const Fr &v, const LinCoeff &c) {
    if (c.kind == 1u) return fe_add(acc, v);
    if (c.kind == 2u) return fe_sub(acc, v);
    return fe_add(acc, fr_mul29(v, c.p));
}
ZG_DEV void psc_first4(F29 (&w)[4], const Fr &lo, const Fr &hi) {
    Fr d = fe_sub(hi, lo), f2 = fe_add(hi, d);
    w[0] = fr29_in(lo);
    w[1] = fr29_in(hi);
    w[2] = fr29_in(f2);
    w[3] = fr29_in(fe_add(f2, d));
}

// End of a round inside the producing kernel: every block leaves NP pairs in `partials` (write-through stores, drained, one relaxed
// arrival: the hand-off of sc_common.hip.h); the block that arrives last adds them up (sc1 loads, lazy sums), writes the 2*NP values
// to the pinned mailbox and publishes the sequence word. v[]: the block's values, valid in thread 0.
template <int NP>
__device__ __forceinline__ void psc_finish(Fr (&v)[2 * NP], uint4 *sh4, uint64_t *partials, uint64_t *sums, uint32_t *counter, uint64_t *flag,
                                           uint64_t seq) {
    const uint32_t tid = threadIdx.x, nb = gridDim.x, lane = tid & 63u;
    u32 *sh = reinterpret_cast<u32 *>(sh4);
    if (nb == 1) {
        if (tid == 0) {
#pragma unroll
            for (int a = 0; a < 2 * NP; a++) mailbox_store_fr(sums + 4 * a, v[a], flag);
            publish_seq(flag, seq);
        }
        return;
    }
    __shared__ uint32_t last;
    if (tid == 0) {
        uint64_t *dst = partials + 16 * (size_t)blockIdx.x;
#pragma unroll
        for (int a = 0; a < 2 * NP; a++) sc1_store_fr(dst + 4 * a, v[a]);
        sc_drain_stores();
        last = sc_arrive(counter, nb) ? 1u : 0u;
    }
    __syncthreads();
    if (!last) return;
    Acc9 acc[2 * NP];
#pragma unroll
    for (int a = 0; a < 2 * NP; a++) acc[a] = acc9_zero();
    for (uint32_t k = tid; k < nb; k += blockDim.x) {
        const uint64_t *src = partials + 16 * (size_t)k;
#pragma unroll
        for (int a = 0; a < 2 * NP; a++) acc9_add(acc[a], sc1_load_fr(src + 4 * a));
    }
    Fr t01 = block_sum_pair9(acc[0], acc[1], sh), t23 = Fr::zero();
    if constexpr (NP == 2) {
        __syncthreads();
        t23 = block_sum_pair9(acc[2], acc[3], sh);
    }
    if (tid < 64) {  // wave 0: lane SC_LANE_G0 holds values 0 and 2, lane SC_LANE_G1 values 1 and 3
        if (lane == SC_LANE_G0 || lane == SC_LANE_G1) {
            const int odd = lane == SC_LANE_G1 ? 1 : 0;
            mailbox_store_fr(sums + 4 * odd, t01, flag);
            if constexpr (NP == 2) mailbox_store_fr(sums + 4 * (2 + odd), t23, flag);
        }
        if (lane == SC_LANE_G0) publish_seq(flag, seq);  // its s_waitcnt covers the wave's stores, the other lane's included
    }
}

// round evaluations at t = 0..3; tables at base + table * stride (elements), pair g = entries 2g, 2g + 1
template <int P, int Q>
__global__ void __launch_bounds__(256) PSC_OCC psc_evals_kernel(const uint64_t *base, size_t stride, size_t half, PscSpec spec, uint64_t *partials,
                                                        uint64_t *sums, uint32_t *counter, uint64_t *flag, uint64_t seq) {
    __shared__ uint4 sh[256 * 4];
    LinCoeff cp[Q > 0 ? Q : 1];
    if constexpr (Q > 0) {
        static_for<0, Q>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            cp[m] = lin_prepare(spec.coeff[m]);
        });
    }
    Fr e[4] = {Fr::zero(), Fr::zero(), Fr::zero(), Fr::zero()};
    ChainAcc4 acc;
    acc.init();
    size_t step = (size_t)gridDim.x * 256;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < half; g += step) {
        F29 w[4];  // the running products at t = 0..3, lazy 29-bit-limb values (fp29.hip.h: fr29_chain_mul)
        if constexpr (Q > 0) {  // L(0), L(1) from the tables, L(2), L(3) by linearity
            Fr l0 = Fr::zero(), l1 = Fr::zero();
            static_for<0, Q>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                const uint64_t *t = base + 4 * ((size_t)spec.lin[m] * stride + 2 * g);
                l0 = lin_add(l0, fe_load<FrParams>(t), cp[m]);
                l1 = lin_add(l1, fe_load<FrParams>(t + 4), cp[m]);
            });
            Fr d = fe_sub(l1, l0), l2 = fe_add(l1, d);
            w[0] = fr29_in(l0);
            w[1] = fr29_in(l1);
            w[2] = fr29_in(l2);
            w[3] = fr29_in(fe_add(l2, d));
        }
        if constexpr (P > 0) {
        static_for<0, P>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const uint64_t *t = base + 4 * ((size_t)spec.prod[j] * stride + 2 * g);
            Fr lo = fe_load<FrParams>(t), hi = fe_load<FrParams>(t + 4);
            if (Q == 0 && j == 0) psc_first4(w, lo, hi);
            else psc_chain4(w, spec.points, lo, hi);
        });
        }
        acc.add(w, e);
    }
    acc.flush(e);
    block_sum_pair(e[0], e[1], sh);
    __syncthreads();
    block_sum_pair(e[2], e[3], sh);
    psc_finish<2>(e, sh, partials, sums, counter, flag, seq);
}

// Gruen's two sums with the split-eq weights (product_remainder.zig:281-330): E tables in device memory, 2^in_bits = |E_in|
template <int P>
__global__ void __launch_bounds__(256) psc_gruen_kernel(const uint64_t *base, size_t stride, size_t half, PscSpec spec, const uint64_t *e_out,
                                                        size_t n_out, const uint64_t *e_in, uint32_t in_bits, uint64_t *partials, uint64_t *sums,
                                                        uint32_t *counter, uint64_t *flag, uint64_t seq) {
    __shared__ uint4 sh[256 * 4];
    Fr e[2] = {Fr::zero(), Fr::zero()};
    Acc29 a0 = acc29_zero(), a1 = acc29_zero();
    unsigned cnt = 0;
    size_t step = (size_t)gridDim.x * 256;
    const size_t mask = ((size_t)1 << in_bits) - 1;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < half; g += step) {
        size_t x_out = g >> in_bits;
        if (x_out >= n_out) continue;  // the reference's loops only reach g = (x_out << bits) | x_in with x_out < |E_out|
        // weight and both products as one lazy chain each (fp29.hip.h): w = E_out * E_in, then the P table factors
        F29 w = fr29_chain_mul(fr29_in(fe_load<FrParams>(e_out + 4 * x_out)), fr29_in_shift(fe_load<FrParams>(e_in + 4 * (g & mask))));
        F29 t0 = w, ti = w;
#pragma unroll
        for (int j = 0; j < P; j++) {
            const uint64_t *t = base + 4 * ((size_t)spec.prod[j] * stride + 2 * g);
            Fr lo = fe_load<FrParams>(t), hi = fe_load<FrParams>(t + 4);
            t0 = fr29_chain_mul(t0, fr29_in_shift(lo));
            ti = fr29_chain_mul(ti, fr29_in_shift(fe_sub(hi, lo)));
        }
        acc29_add(a0, t0);
        acc29_add(a1, ti);
        if (++cnt == FR29_ACC_MAX) {
            e[0] = fe_add(e[0], acc29_reduce(a0));
            e[1] = fe_add(e[1], acc29_reduce(a1));
            a0 = acc29_zero();
            a1 = acc29_zero();
            cnt = 0;
        }
    }
    if (cnt) {
        e[0] = fe_add(e[0], acc29_reduce(a0));
        e[1] = fe_add(e[1], acc29_reduce(a1));
    }
    block_sum_pair(e[0], e[1], sh);
    psc_finish<1>(e, sh, partials, sums, counter, flag, seq);
}

struct PscTables {
    uint32_t t[ZG_PSC_MAX_TABLES];
};

// tables which[0..gridDim.y) folded by r in one launch: out[i] = (1 - r) t[2i] + r t[2i+1] = t[2i] + r (t[2i+1] - t[2i])
__global__ void __launch_bounds__(256) psc_fold_kernel(const uint64_t *base, size_t stride, size_t half, FrArg r, uint64_t *out, size_t ostride,
                                                       PscTables which) {
    Fr rv;
#pragma unroll
    for (int i = 0; i < 8; i++) rv.l[i] = r.l[i];
    FrMul rp = frmul_prepare(rv);
    const uint32_t table = which.t[blockIdx.y];
    const uint64_t *t = base + 4 * (size_t)table * stride;
    uint64_t *o = out + 4 * (size_t)table * ostride;
    size_t step = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < half; i += step) {
        Fr lo = fe_load<FrParams>(t + 8 * i), hi = fe_load<FrParams>(t + 8 * i + 4);
        fe_store(o + 4 * i, fe_add(lo, frmul_apply(fe_sub(hi, lo), rp)));
    }
}

// The fold of a round and the NEXT round's evaluations in one pass (the spec of the previous psc_round_evals call): a thread folds
// the two old pairs 4g..4g+3 of every table the spec names into the new pair (2g, 2g+1), writes it, and evaluates the product form
// on the values it still holds — one launch per round, and the folded tables are not read back.
template <int P, int Q>
__global__ void __launch_bounds__(256) PSC_OCC psc_fold_evals_kernel(const uint64_t *base, size_t stride, size_t quarter, FrArg r, uint64_t *out,
                                                             size_t ostride, PscSpec spec, uint64_t *partials, uint64_t *sums, uint32_t *counter,
                                                             uint64_t *flag, uint64_t seq) {
    __shared__ uint4 sh[256 * 4];
    Fr rv;
#pragma unroll
    for (int i = 0; i < 8; i++) rv.l[i] = r.l[i];
    FrMul rp = frmul_prepare(rv);
    LinCoeff cp[Q > 0 ? Q : 1];
    if constexpr (Q > 0) {
        static_for<0, Q>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            cp[m] = lin_prepare(spec.coeff[m]);
        });
    }
    Fr e[4] = {Fr::zero(), Fr::zero(), Fr::zero(), Fr::zero()};
    ChainAcc4 acc;
    acc.init();
    size_t step = (size_t)gridDim.x * 256;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < quarter; g += step) {
        F29 w[4];
        if constexpr (Q > 0) {
            Fr l0 = Fr::zero(), l1 = Fr::zero();
            static_for<0, Q>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                const uint64_t *t = base + 4 * ((size_t)spec.lin[m] * stride + 4 * g);
                Fr a0 = fe_load<FrParams>(t), a1 = fe_load<FrParams>(t + 4), a2 = fe_load<FrParams>(t + 8), a3 = fe_load<FrParams>(t + 12);
                Fr lo = fe_add(a0, frmul_apply(fe_sub(a1, a0), rp)), hi = fe_add(a2, frmul_apply(fe_sub(a3, a2), rp));
                uint64_t *o = out + 4 * ((size_t)spec.lin[m] * ostride + 2 * g);
                fe_store(o, lo);
                fe_store(o + 4, hi);
                l0 = lin_add(l0, lo, cp[m]);
                l1 = lin_add(l1, hi, cp[m]);
            });
            Fr d = fe_sub(l1, l0), l2 = fe_add(l1, d);
            w[0] = fr29_in(l0);
            w[1] = fr29_in(l1);
            w[2] = fr29_in(l2);
            w[3] = fr29_in(fe_add(l2, d));
        }
        if constexpr (P > 0) {
        static_for<0, P>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const uint64_t *t = base + 4 * ((size_t)spec.prod[j] * stride + 4 * g);
            Fr a0 = fe_load<FrParams>(t), a1 = fe_load<FrParams>(t + 4), a2 = fe_load<FrParams>(t + 8), a3 = fe_load<FrParams>(t + 12);
            Fr lo = fe_add(a0, frmul_apply(fe_sub(a1, a0), rp)), hi = fe_add(a2, frmul_apply(fe_sub(a3, a2), rp));
            uint64_t *o = out + 4 * ((size_t)spec.prod[j] * ostride + 2 * g);
            fe_store(o, lo);
            fe_store(o + 4, hi);
            if (Q == 0 && j == 0) psc_first4(w, lo, hi);
            else psc_chain4(w, spec.points, lo, hi);
        });
        }
        acc.add(w, e);
    }
    acc.flush(e);
    block_sum_pair(e[0], e[1], sh);
    __syncthreads();
    block_sum_pair(e[2], e[3], sh);
    psc_finish<2>(e, sh, partials, sums, counter, flag, seq);
}

// A SUM of up to four product terms (each: up to four plain factors times an optional linear combination), evaluated at t = 0..3 —
// the round polynomials of stage3_prover.zig (ShiftSumcheck phase 1: four P*Q pairs, :1351-1392; phase 2: eq_out * (upc + g pc + g^2 virt
// + g^3 first) + g^4 (1 - noop) eq_prod, :1399-1455; InstructionInput: (eq_out + g^2 eq_prod) * (is_rs2 * rs2 + is_imm * imm +
// g (is_rs1 * rs1 + is_pc * pc)), :2029-2100). Runtime loops over the description (uniform across the grid), no register arrays.
struct PscExprTerm {
    uint32_t np, nq;
    uint32_t pair_sum;  // ZG_PSC_PAIR_SUM: (T[prod0] * T[prod1] + T[prod2] * T[prod3]) * L instead of the product of the four
    uint32_t prod[ZG_PSC_MAX_FACTORS], lin[ZG_PSC_MAX_FACTORS];
    FrArg coeff[ZG_PSC_MAX_FACTORS];
};
struct PscExpr {
    uint32_t n_terms;
    uint32_t points;  // as PscSpec::points
    PscExprTerm t[ZG_PSC_MAX_TERMS];
};

// FOLD: fold the two old pairs 4g..4g+3 of every named table by r into the new pair first (and write it), as psc_fold_evals_kernel
template <bool FOLD>
__global__ void __launch_bounds__(256) PSC_OCC psc_expr_kernel(const uint64_t *base, size_t stride, size_t n_pairs, FrArg r, uint64_t *out, size_t ostride,
                                                       PscExpr ex, uint64_t *partials, uint64_t *sums, uint32_t *counter, uint64_t *flag, uint64_t seq) {
    __shared__ uint4 sh[256 * 4];
    FrMul rp;
    rp.narrow = false;
    if (FOLD) {
        Fr rv;
#pragma unroll
        for (int i = 0; i < 8; i++) rv.l[i] = r.l[i];
        rp = frmul_prepare(rv);
    }
    Fr e[4] = {Fr::zero(), Fr::zero(), Fr::zero(), Fr::zero()};
    ChainAcc4 acc;
    acc.init();
    size_t step = (size_t)gridDim.x * 256;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < n_pairs; g += step) {
        Fr l0 = Fr::zero(), l1 = Fr::zero();  // the linear combination of the current term; kept while the next terms name the same one
        for (uint32_t ti = 0; ti < ex.n_terms; ti++) {
            const PscExprTerm &tm = ex.t[ti];
            // consecutive terms with the same combination (InstructionInput: two terms per weight) share its two values — and, when
            // folding, the fold of its tables. Uniform: the description is a kernel argument.
            bool same_lin = ti > 0 && tm.nq == ex.t[ti - 1].nq;
            if (same_lin)
                for (uint32_t m = 0; m < tm.nq; m++) {
                    same_lin = same_lin && tm.lin[m] == ex.t[ti - 1].lin[m];
                    for (int i = 0; i < 8; i++) same_lin = same_lin && tm.coeff[m].l[i] == ex.t[ti - 1].coeff[m].l[i];
                }
            F29 w[4];
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int i = 0; i < 9; i++) w[t].l[i] = 0;
            bool have = false;
            auto pair_of = [&](uint32_t table, Fr &lo, Fr &hi) {
                if (FOLD) {
                    const uint64_t *t = base + 4 * ((size_t)table * stride + 4 * g);
                    Fr a0 = fe_load<FrParams>(t), a1 = fe_load<FrParams>(t + 4), a2 = fe_load<FrParams>(t + 8), a3 = fe_load<FrParams>(t + 12);
                    lo = fe_add(a0, frmul_apply(fe_sub(a1, a0), rp));
                    hi = fe_add(a2, frmul_apply(fe_sub(a3, a2), rp));
                    uint64_t *o = out + 4 * ((size_t)table * ostride + 2 * g);
                    fe_store(o, lo);
                    fe_store(o + 4, hi);
                } else {
                    const uint64_t *t = base + 4 * ((size_t)table * stride + 2 * g);
                    lo = fe_load<FrParams>(t);
                    hi = fe_load<FrParams>(t + 4);
                }
            };
            if (tm.nq) {
                if (!same_lin) {
                l0 = Fr::zero();
                l1 = Fr::zero();
                for (uint32_t m = 0; m < tm.nq; m++) {
                    Fr lo, hi;
                    pair_of(tm.lin[m], lo, hi);
                    const uint32_t kind = lin_kind(tm.coeff[m]);
                    if (kind == 1u) {
                        l0 = fe_add(l0, lo);
                        l1 = fe_add(l1, hi);
                    } else if (kind == 2u) {
                        l0 = fe_sub(l0, lo);
                        l1 = fe_sub(l1, hi);
                    } else {
                        Fr c;
#pragma unroll
                        for (int i = 0; i < 8; i++) c.l[i] = tm.coeff[m].l[i];
                        l0 = fe_add(l0, fr_mul29v(lo, c));
                        l1 = fe_add(l1, fr_mul29v(hi, c));
                    }
                }
                }
                if (!tm.pair_sum) {
                    Fr d = fe_sub(l1, l0), l2 = fe_add(l1, d);
                    w[0] = fr29_in(l0);
                    w[1] = fr29_in(l1);
                    w[2] = fr29_in(l2);
                    w[3] = fr29_in(fe_add(l2, d));
                    have = true;
                }
            }
            if (tm.pair_sum) {
                // (a b + c d) L: the two pair products are added as lazy values (limbs < 2^30, value < 2.4 r: still a valid left operand of
                // the multiplier, product < (2.4 * 32 / 168.9 + 1) r) and the sum takes ONE product by L — three per point instead of four
                for (uint32_t q = 0; q < 2; q++) {
                    Fr alo, ahi, blo, bhi;
                    pair_of(tm.prod[2 * q], alo, ahi);
                    pair_of(tm.prod[2 * q + 1], blo, bhi);
                    Fr da = fe_sub(ahi, alo), db = fe_sub(bhi, blo);
                    Fr at = alo, bt = blo;
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        if ((ex.points >> t) & 1u) {
                            F29 v = fr29_chain_mul(fr29_in(at), fr29_in_shift(bt));
#pragma unroll
                            for (int i = 0; i < 9; i++) w[t].l[i] += v.l[i];
                        }
                        if (t < 3) {
                            at = fe_add(at, da);
                            bt = fe_add(bt, db);
                        }
                    }
                }
                if (tm.nq) {
                    Fr dl = fe_sub(l1, l0), lt = l0;
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        if ((ex.points >> t) & 1u) w[t] = fr29_chain_mul(w[t], fr29_in_shift(lt));
                        if (t < 3) lt = fe_add(lt, dl);
                    }
                } else {  // no weight: one product by 1 brings the sum of two lazy values back under 2 r with exact limbs (what the sums expect)
                    const F29 one_s = fr29_in_shift(Fr::one());
#pragma unroll
                    for (int t = 0; t < 4; t++)
                        if ((ex.points >> t) & 1u) w[t] = fr29_chain_mul(w[t], one_s);
                }
            } else
            for (uint32_t j = 0; j < tm.np; j++) {
                Fr lo, hi;
                pair_of(tm.prod[j], lo, hi);
                if (!have) {
                    psc_first4(w, lo, hi);
                    have = true;
                } else {  // the running products stay lazy 29-bit-limb values (fp29.hip.h: fr29_chain_mul)
                    psc_chain4(w, ex.points, lo, hi);
                }
            }
            acc.add(w, e);
        }
    }
    acc.flush(e);
    block_sum_pair(e[0], e[1], sh);
    __syncthreads();
    block_sum_pair(e[2], e[3], sh);
    psc_finish<2>(e, sh, partials, sums, counter, flag, seq);
}

// ---- the same round evaluations for SHORT tables: one lane per (pair, term, point). In psc_expr_kernel a thread walks every term and
// every point of its pair — with a few pairs left that is ONE chain of up to ~60 dependent field products per round (35-65 us at
// 2.2 ns per instruction of a lone wave, whatever the table length: the floor of Stage 3's last dozen rounds). Here lane
// u = 16 g + 4 term + point evaluates its term at its point only: the folds of the term's tables, its linear combination at that point
// (one product per weighted table instead of two) and the chain of its factors — a quarter to a sixth of the dependent products.
// Values: the same field elements (a term's value at a point is one product of canonical factors however it is scheduled), summed
// by the same canonical reduction, so every round's evaluations are the reference's bytes as before.
ZG_DEV Fr psc_at_point(const Fr &lo, const Fr &hi, uint32_t t) {  // lo + t (hi - lo), t = 0..3, without lane divergence
    const Fr d = fe_sub(hi, lo);
    Fr v, a1, a2;
    const u32 m0 = t == 0 ? ~0u : 0u, m2 = t >= 2 ? ~0u : 0u, m3 = t == 3 ? ~0u : 0u;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        v.l[i] = (lo.l[i] & m0) | (hi.l[i] & ~m0);
        a1.l[i] = d.l[i] & m2;
        a2.l[i] = d.l[i] & m3;
    }
    return fe_add(fe_add(v, a1), a2);
}
constexpr size_t PSC_SPREAD_MAX_PAIRS = 1024;
template <bool FOLD>
__global__ void __launch_bounds__(256) psc_expr_spread_kernel(const uint64_t *base, size_t stride, size_t n_pairs, FrArg r, uint64_t *out, size_t ostride,
                                                              PscExpr ex, uint64_t *partials, uint64_t *sums, uint32_t *counter, uint64_t *flag, uint64_t seq) {
    __shared__ uint4 sh[256 * 4];
    FrMul rp;
    rp.narrow = false;
    if (FOLD) {
        Fr rv;
#pragma unroll
        for (int i = 0; i < 8; i++) rv.l[i] = r.l[i];
        rp = frmul_prepare(rv);
    }
    const size_t u = (size_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t t = (uint32_t)u & 3u, ti = ((uint32_t)u >> 2) & 3u;
    const size_t g = u >> 4;
    const bool live = g < n_pairs && ti < ex.n_terms && ((ex.points >> t) & 1u);
    const bool writer = g < n_pairs && ti < ex.n_terms && t == 0;  // the lane that stores a folded pair (a table named by several terms is stored by each: same bytes)
    Fr val = Fr::zero();
    if (g < n_pairs && ti < ex.n_terms) {
        const PscExprTerm &tm = ex.t[ti];
        auto pair_of = [&](uint32_t table, Fr &lo, Fr &hi) {
            if (FOLD) {
                const uint64_t *p4 = base + 4 * ((size_t)table * stride + 4 * g);
                Fr a0 = fe_load<FrParams>(p4), a1 = fe_load<FrParams>(p4 + 4), a2 = fe_load<FrParams>(p4 + 8), a3 = fe_load<FrParams>(p4 + 12);
                lo = fe_add(a0, frmul_apply(fe_sub(a1, a0), rp));
                hi = fe_add(a2, frmul_apply(fe_sub(a3, a2), rp));
                if (writer) {
                    uint64_t *o = out + 4 * ((size_t)table * ostride + 2 * g);
                    fe_store(o, lo);
                    fe_store(o + 4, hi);
                }
            } else {
                const uint64_t *p2 = base + 4 * ((size_t)table * stride + 2 * g);
                lo = fe_load<FrParams>(p2);
                hi = fe_load<FrParams>(p2 + 4);
            }
        };
        Fr L = Fr::zero();  // the term's linear combination at this lane's point
        for (uint32_t m = 0; m < tm.nq; m++) {
            Fr lo, hi;
            pair_of(tm.lin[m], lo, hi);
            const Fr x = psc_at_point(lo, hi, t);
            const uint32_t kind = lin_kind(tm.coeff[m]);
            if (kind == 1u) {
                L = fe_add(L, x);
            } else if (kind == 2u) {
                L = fe_sub(L, x);
            } else {
                Fr c;
#pragma unroll
                for (int i = 0; i < 8; i++) c.l[i] = tm.coeff[m].l[i];
                L = fe_add(L, fr_mul29v(x, c));
            }
        }
        F29 w;
#pragma unroll
        for (int i = 0; i < 9; i++) w.l[i] = 0;
        bool have = false;
        if (tm.pair_sum) {
            for (uint32_t q = 0; q < 2; q++) {
                Fr alo, ahi, blo, bhi;
                pair_of(tm.prod[2 * q], alo, ahi);
                pair_of(tm.prod[2 * q + 1], blo, bhi);
                F29 v = fr29_chain_mul(fr29_in(psc_at_point(alo, ahi, t)), fr29_in_shift(psc_at_point(blo, bhi, t)));
#pragma unroll
                for (int i = 0; i < 9; i++) w.l[i] += v.l[i];
            }
            w = fr29_chain_mul(w, fr29_in_shift(tm.nq ? L : Fr::one()));  // (a b + c d) L; without a weight the product by 1 restores exact limbs
        } else {
            if (tm.nq) {
                w = fr29_in(L);
                have = true;
            }
            for (uint32_t j = 0; j < tm.np; j++) {
                Fr lo, hi;
                pair_of(tm.prod[j], lo, hi);
                const Fr f = psc_at_point(lo, hi, t);
                if (!have) {
                    w = fr29_in(f);
                    have = true;
                } else {
                    w = fr29_chain_mul(w, fr29_in_shift(f));
                }
            }
        }
        if (live) val = fr29_out(w);
    }
    Fr e[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const u32 mk = (t == (uint32_t)k) ? ~0u : 0u;
#pragma unroll
        for (int i = 0; i < 8; i++) e[k].l[i] = val.l[i] & mk;
    }
    block_sum_pair(e[0], e[1], sh);
    __syncthreads();
    block_sum_pair(e[2], e[3], sh);
    psc_finish<2>(e, sh, partials, sums, counter, flag, seq);
}

// one lane per (pair, term, point) from this many pairs down (ZG_PSC_SPREAD_MAX_PAIRS; 0 = never): measured, tools/bench_sumcheck Stage 3
static bool psc_spread(size_t pairs) {
    static const size_t lim = [] {
        const char *e = getenv("ZG_PSC_SPREAD_MAX_PAIRS");
        long v = e && *e ? atol(e) : (long)PSC_SPREAD_MAX_PAIRS;
        return (size_t)(v < 0 ? 0 : v);
    }();
    return pairs >= 1 && pairs <= lim && pairs * 16 <= (size_t)PSC_MAX_BLOCKS * 256;
}
static unsigned psc_blocks(size_t half) {
    static const unsigned cap = [] {
        const char *e = getenv("ZG_PSC_BLOCKS");
        int v = e && *e ? atoi(e) : 512;
        return (unsigned)(v < 1 ? 1 : (v > (int)PSC_MAX_BLOCKS ? (int)PSC_MAX_BLOCKS : v));
    }();
    size_t nb = (half + 255) / 256;
    if (nb < 1) nb = 1;
    return (unsigned)(nb > cap ? cap : nb);
}

}  // namespace zg

struct zg_psc_s {
    int device = -1;
    size_t k = 0, len = 0, cap = 0;
    size_t alloc_k = 0;  // tables the buffers were sized for (a pooled session serves any k <= alloc_k)
    uint64_t *buf[2] = {nullptr, nullptr};  // buf[0]: k tables of cap entries; buf[1]: k tables of cap/2 (a fold cannot run in place)
    int cur = 0;
    uint64_t *d_misc = nullptr;
    uint64_t *h_pin = nullptr;  // pinned, device-visible mailbox: up to 16 words of values, sequence word at PSC_FLAG
    hipStream_t st = nullptr;
    hipStream_t own_st = nullptr;  // created with the session, kept in the pool (see psc_open_stream)
    uint64_t seq = 0;
    // the spec of the last zg_psc_round_evals call: zg_psc_bind then folds AND evaluates in one launch, and the next
    // zg_psc_round_evals with the same spec only collects the mailbox
    bool have_spec = false, evals_pending = false;
    zg::PscSpec spec;
    size_t spec_p = 0, spec_q = 0;
    uint32_t points = 0xF;      // zg_psc_set_points: evaluation points the round calls compute
    bool spec_is_expr = false;  // the cached description is `expr` (zg_psc_round_expr) instead of spec / spec_p / spec_q
    zg::PscExpr expr;
    std::mutex mu;
    size_t stride() const { return cur == 0 ? cap : (cap / 2 ? cap / 2 : 1); }
};

using namespace zg;

static void psc_free(zg_psc_s *s) {
    if (!s) return;
    if (s->own_st) (void)hipStreamSynchronize(s->own_st);  // the tables go back to the device pool: nothing of this session may still run
    if (s->st && s->st != s->own_st) (void)hipDeviceSynchronize();  // (a caller's stream may be gone by now: wait for the device instead)
    pool_free(s->buf[0]);
    pool_free(s->buf[1]);
    pool_free(s->d_misc);
    if (s->h_pin) (void)hipHostFree(s->h_pin);
    stream_release(s->own_st, s->device);
    delete s;
}

// A session opened without a caller stream runs on a stream of its own, so the independent provers of a batched sumcheck overlap;
// the stream is created with the session and STAYS with it in the pool. Measured (tools/bench_sumcheck, Stage-2-shaped proof: four of
// these sessions + one single-table session, same box, A/B): rounds of the proof 1.42 ms with pooled streams, 1.15 ms with a stream
// created per open (streams created together land on consecutive hardware queues; long-lived ones can share a queue with another busy
// session), 1.40 / 1.55 ms with a per-device ring of eight / four shared streams, 2.2 ms with everything on the library stream —
// but hipStreamCreate + hipStreamDestroy cost 2.9 ms per open, against 45 us for a pooled open: per proof the pooled stream wins by far.
static hipError_t psc_open_stream(zg_psc_s *s, hipStream_t caller) {
    if (!s->own_st) s->own_st = stream_acquire();  // from the runtime's free list: creating one costs ~3 ms
    s->st = caller ? caller : s->own_st;
    return s->own_st ? hipSuccess : hipErrorOutOfMemory;
}

// closed sessions kept for reuse (allocations and the pinned mailbox cost more than a proof's worth of rounds)
static std::mutex g_psc_pool_mu;
static std::vector<zg_psc_s *> g_psc_pool;

static int psc_create(size_t k, size_t len, hipStream_t st, zg_psc_s **out) {
    if (k == 0 || k > ZG_PSC_MAX_TABLES || len == 0 || (len & (len - 1))) {
        set_error("zg_psc_open: 1..12 tables, len a power of two");
        return ZG_ERR_INVALID;
    }
    {
        std::lock_guard<std::mutex> lk(g_psc_pool_mu);
        for (size_t i = 0; i < g_psc_pool.size(); i++) {
            zg_psc_s *c = g_psc_pool[i];
            if (c->device == current_device() && c->alloc_k >= k && c->cap >= len && c->cap <= 4 * len) {
                g_psc_pool.erase(g_psc_pool.begin() + i);
                c->k = k; c->len = len; c->cur = 0; c->seq = 0; c->h_pin[PSC_FLAG] = 0;
                c->have_spec = c->evals_pending = false;
                c->points = 0xF;
                if (psc_open_stream(c, st) != hipSuccess) {
                    set_error("zg_psc_open: hipStreamCreate failed");
                    psc_free(c);
                    return ZG_ERR_HIP;
                }
                *out = c;
                return ZG_OK;
            }
        }
    }
    zg_psc_s *s = new zg_psc_s();
    s->device = current_device();
    s->k = s->alloc_k = k;
    s->len = s->cap = len;
    size_t half = len / 2 ? len / 2 : 1;
    hipError_t e = psc_open_stream(s, st);
    auto grab = [&](uint64_t *&ptr, size_t bytes) {  // from the device pool (runtime.hip)
        if (e == hipSuccess && !(ptr = reinterpret_cast<uint64_t *>(pool_alloc(bytes)))) e = hipErrorOutOfMemory;
    };
    grab(s->buf[0], k * len * 32);
    grab(s->buf[1], k * half * 32);
    grab(s->d_misc, PSC_MISC_BYTES);
    if (e == hipSuccess) e = hipMemset(s->d_misc, 0, PSC_MISC_BYTES);
    if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_pin, 256, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) s->h_pin[PSC_FLAG] = 0;
    if (e != hipSuccess) {
        set_error(std::string("zg_psc_open: ") + hipGetErrorString(e));
        psc_free(s);
        return e == hipErrorOutOfMemory ? ZG_ERR_NOMEM : ZG_ERR_HIP;
    }
    *out = s;
    return ZG_OK;
}

// host side of a round that ended inside its kernel (same protocol as the single-table sessions of poly.hip)
static int psc_wait(zg_psc_s *s, uint64_t *out, int words) {
    ZG_HIP(hipGetLastError());
    volatile uint64_t *flag = s->h_pin + PSC_FLAG;
    bool got = false;
    for (uint64_t spin = 0; spin < (1ull << 22); spin++) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == s->seq) { got = true; break; }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    if (!got) ZG_HIP(hipStreamSynchronize(s->st));
    for (int i = 0; i < words; i++) out[i] = s->h_pin[i];
    return ZG_OK;
}

// the evaluations that were not asked for come back as zero
static int psc_collect(zg_psc_s *s, uint32_t points, uint64_t *out) {
    int rc = psc_wait(s, out, 16);
    for (int t = 0; t < 4; t++)
        if (!((points >> t) & 1u)) memset(out + 4 * t, 0, 32);
    return rc;
}

static int psc_spec(zg_psc_s *s, const int *prod_idx, size_t p, const int *lin_idx, const uint64_t *lin_coeff, size_t q, PscSpec *spec,
                    const char *who) {
    bool bad = !s || p > ZG_PSC_MAX_FACTORS || q > ZG_PSC_MAX_FACTORS || (p + q) == 0 || (p && !prod_idx) || (q && (!lin_idx || !lin_coeff));
    for (size_t j = 0; !bad && j < p; j++) bad = prod_idx[j] < 0 || (size_t)prod_idx[j] >= s->k;
    for (size_t m = 0; !bad && m < q; m++) bad = lin_idx[m] < 0 || (size_t)lin_idx[m] >= s->k;
    if (bad) {
        set_error(std::string(who) + ": at most 4 product tables and 4 linear-combination tables, indices below the session's table count");
        return ZG_ERR_INVALID;
    }
    spec->points = 0xF;
    for (size_t j = 0; j < ZG_PSC_MAX_FACTORS; j++) {
        spec->prod[j] = j < p ? (uint32_t)prod_idx[j] : 0;
        spec->lin[j] = j < q ? (uint32_t)lin_idx[j] : 0;
        for (int i = 0; i < 4; i++) {
            uint64_t w = j < q ? lin_coeff[4 * j + i] : 0;
            spec->coeff[j].l[2 * i] = (uint32_t)w;
            spec->coeff[j].l[2 * i + 1] = (uint32_t)(w >> 32);
        }
    }
    return ZG_OK;
}

template <int P>
static void psc_launch_evals_q(size_t q, unsigned nb, hipStream_t st, const uint64_t *base, size_t stride, size_t half, const PscSpec &spec,
                               uint64_t *partials, uint64_t *sums, uint32_t *counter, uint64_t *flag, uint64_t seq) {
    switch (q) {
    case 0: hipLaunchKernelGGL((psc_evals_kernel<P, 0>), dim3(nb), dim3(256), 0, st, base, stride, half, spec, partials, sums, counter, flag, seq); break;
    case 1: hipLaunchKernelGGL((psc_evals_kernel<P, 1>), dim3(nb), dim3(256), 0, st, base, stride, half, spec, partials, sums, counter, flag, seq); break;
    case 2: hipLaunchKernelGGL((psc_evals_kernel<P, 2>), dim3(nb), dim3(256), 0, st, base, stride, half, spec, partials, sums, counter, flag, seq); break;
    case 3: hipLaunchKernelGGL((psc_evals_kernel<P, 3>), dim3(nb), dim3(256), 0, st, base, stride, half, spec, partials, sums, counter, flag, seq); break;
    default: hipLaunchKernelGGL((psc_evals_kernel<P, 4>), dim3(nb), dim3(256), 0, st, base, stride, half, spec, partials, sums, counter, flag, seq); break;
    }
}

template <int P>
static void psc_launch_fold_evals_q(size_t q, unsigned nb, hipStream_t st, const uint64_t *base, size_t stride, size_t quarter, const FrArg &r,
                                    uint64_t *out, size_t ostride, const PscSpec &spec, uint64_t *partials, uint64_t *sums, uint32_t *counter,
                                    uint64_t *flag, uint64_t seq) {
    switch (q) {
    case 0: hipLaunchKernelGGL((psc_fold_evals_kernel<P, 0>), dim3(nb), dim3(256), 0, st, base, stride, quarter, r, out, ostride, spec, partials, sums, counter, flag, seq); break;
    case 1: hipLaunchKernelGGL((psc_fold_evals_kernel<P, 1>), dim3(nb), dim3(256), 0, st, base, stride, quarter, r, out, ostride, spec, partials, sums, counter, flag, seq); break;
    case 2: hipLaunchKernelGGL((psc_fold_evals_kernel<P, 2>), dim3(nb), dim3(256), 0, st, base, stride, quarter, r, out, ostride, spec, partials, sums, counter, flag, seq); break;
    case 3: hipLaunchKernelGGL((psc_fold_evals_kernel<P, 3>), dim3(nb), dim3(256), 0, st, base, stride, quarter, r, out, ostride, spec, partials, sums, counter, flag, seq); break;
    default: hipLaunchKernelGGL((psc_fold_evals_kernel<P, 4>), dim3(nb), dim3(256), 0, st, base, stride, quarter, r, out, ostride, spec, partials, sums, counter, flag, seq); break;
    }
}

static bool psc_same_spec(const zg_psc_s *s, const PscSpec &spec, size_t p, size_t q) {
    if (!s->have_spec || s->spec_is_expr || s->spec_p != p || s->spec_q != q || s->spec.points != spec.points) return false;
    for (size_t j = 0; j < p; j++)
        if (s->spec.prod[j] != spec.prod[j]) return false;
    for (size_t m = 0; m < q; m++) {
        if (s->spec.lin[m] != spec.lin[m]) return false;
        for (int i = 0; i < 8; i++)
            if (s->spec.coeff[m].l[i] != spec.coeff[m].l[i]) return false;
    }
    return true;
}

extern "C" {

int zg_psc_open(const uint64_t *const *tables, size_t k, size_t len, zg_psc_t *out) {
    ZG_INIT();
    if (!tables || !out) {
        set_error("zg_psc_open: invalid argument");
        return ZG_ERR_INVALID;
    }
    for (size_t j = 0; j < k && j < ZG_PSC_MAX_TABLES; j++)
        if (!tables[j]) {
            set_error("zg_psc_open: null table");
            return ZG_ERR_INVALID;
        }
    zg_psc_s *s = nullptr;
    ZG_TRY(psc_create(k, len, nullptr, &s));
    for (size_t j = 0; j < k; j++) {
        hipError_t e = hipMemcpyAsync(s->buf[0] + 4 * j * s->cap, tables[j], len * 32, hipMemcpyHostToDevice, s->st);
        if (e != hipSuccess) {
            (void)hipStreamSynchronize(s->st);
            set_error(hipGetErrorString(e));
            psc_free(s);
            return ZG_ERR_HIP;
        }
    }
    hipError_t e = hipStreamSynchronize(s->st);  // the host tables may be released on return
    if (e != hipSuccess) {
        set_error(hipGetErrorString(e));
        psc_free(s);
        return ZG_ERR_HIP;
    }
    *out = s;
    return ZG_OK;
}

int zg_psc_open_dev(const uint64_t *const *d_tables, size_t k, size_t len, void *stream, zg_psc_t *out) {
    ZG_INIT();
    if (!d_tables || !out) {
        set_error("zg_psc_open_dev: invalid argument");
        return ZG_ERR_INVALID;
    }
    for (size_t j = 0; j < k && j < ZG_PSC_MAX_TABLES; j++)
        if (!d_tables[j]) {
            set_error("zg_psc_open_dev: null table");
            return ZG_ERR_INVALID;
        }
    zg_psc_s *s = nullptr;
    // stream == NULL: the session works on its OWN pooled stream (sessions of one prover overlap instead of queueing on the library
    // stream); the copies are ordered after what the library stream holds — where the *_dev table builders put their work by default —
    // and have completed on return, so the caller may release or overwrite the sources at once
    hipStream_t caller = reinterpret_cast<hipStream_t>(stream);
    ZG_TRY(psc_create(k, len, caller, &s));
    hipError_t e = hipSuccess;
    if (!caller) {
        hipEvent_t ev = nullptr;
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(ev, lib_stream());
        if (e == hipSuccess) e = hipStreamWaitEvent(s->st, ev, 0);
        if (ev) (void)hipEventDestroy(ev);
    }
    for (size_t j = 0; j < k && e == hipSuccess; j++)
        e = hipMemcpyAsync(s->buf[0] + 4 * j * s->cap, d_tables[j], len * 32, hipMemcpyDeviceToDevice, s->st);
    if (e == hipSuccess && !caller) e = hipStreamSynchronize(s->st);
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(s->st);
        set_error(hipGetErrorString(e));
        psc_free(s);
        return ZG_ERR_HIP;
    }
    *out = s;
    return ZG_OK;
}

size_t zg_psc_len(zg_psc_t s) { return s ? s->len : 0; }
size_t zg_psc_tables(zg_psc_t s) { return s ? s->k : 0; }

int zg_psc_round_evals(zg_psc_t s, const int *prod_idx, size_t p, const int *lin_idx, const uint64_t *lin_coeff, size_t q, uint64_t out[16]) {
    ZG_INIT();
    PscSpec spec;
    ZG_TRY(psc_spec(s, prod_idx, p, lin_idx, lin_coeff, q, &spec, "zg_psc_round_evals"));
    if (!out || s->len < 2) {
        set_error("zg_psc_round_evals: invalid argument or a single entry left");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    spec.points = s->points;
    if (s->evals_pending && psc_same_spec(s, spec, p, q)) return psc_collect(s, spec.points, out);  // produced by the fold of the last bind
    const size_t half = s->len / 2;
    unsigned nb = psc_blocks(half);
    uint32_t *counter = reinterpret_cast<uint32_t *>(s->d_misc + PSC_COUNTER_OFF);
    uint64_t *flag = s->h_pin + PSC_FLAG;
    s->seq++;
    s->spec = spec;
    s->spec_p = p;
    s->spec_q = q;
    s->spec_is_expr = false;
    s->have_spec = true;
    s->evals_pending = true;  // the mailbox holds this spec's evaluations of the current tables until the next bind / Gruen round
    const uint64_t *base = s->buf[s->cur];
    switch (p) {
    case 0: psc_launch_evals_q<0>(q, nb, s->st, base, s->stride(), half, spec, s->d_misc, s->h_pin, counter, flag, s->seq); break;
    case 1: psc_launch_evals_q<1>(q, nb, s->st, base, s->stride(), half, spec, s->d_misc, s->h_pin, counter, flag, s->seq); break;
    case 2: psc_launch_evals_q<2>(q, nb, s->st, base, s->stride(), half, spec, s->d_misc, s->h_pin, counter, flag, s->seq); break;
    case 3: psc_launch_evals_q<3>(q, nb, s->st, base, s->stride(), half, spec, s->d_misc, s->h_pin, counter, flag, s->seq); break;
    default: psc_launch_evals_q<4>(q, nb, s->st, base, s->stride(), half, spec, s->d_misc, s->h_pin, counter, flag, s->seq); break;
    }
    return psc_collect(s, spec.points, out);
}

int zg_psc_set_points(zg_psc_t s, unsigned points) {
    ZG_INIT();
    if (!s || points == 0 || points > 0xFu) {
        set_error("zg_psc_set_points: a non-empty subset of the points 0..3 as a bit mask");
        return ZG_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lk(s->mu);
    s->points = points;
    return ZG_OK;
}

int zg_psc_round_expr(zg_psc_t s, const zg_psc_term *terms, size_t n_terms, uint64_t out[16]) {
    ZG_INIT();
    if (!s || !terms || !out || n_terms == 0 || n_terms > ZG_PSC_MAX_TERMS || s->len < 2) {
        set_error("zg_psc_round_expr: 1..4 terms, a session with at least two entries");
        return ZG_ERR_INVALID;
    }
    PscExpr ex;
    memset(&ex, 0, sizeof(ex));
    ex.n_terms = (uint32_t)n_terms;
    for (size_t ti = 0; ti < n_terms; ti++) {
        const zg_psc_term &t = terms[ti];
        const bool pair_sum = t.n_prod >= 0 && (t.n_prod & ZG_PSC_PAIR_SUM);
        const int n_prod = pair_sum ? (t.n_prod & ~ZG_PSC_PAIR_SUM) : t.n_prod;
        bool bad = n_prod < 0 || n_prod > ZG_PSC_MAX_FACTORS || t.n_lin < 0 || t.n_lin > ZG_PSC_MAX_FACTORS || n_prod + t.n_lin == 0 ||
                   (pair_sum && n_prod != 4);
        for (int j = 0; !bad && j < n_prod; j++) bad = t.prod[j] < 0 || (size_t)t.prod[j] >= s->k;
        for (int m = 0; !bad && m < t.n_lin; m++) bad = t.lin[m] < 0 || (size_t)t.lin[m] >= s->k;
        if (bad) {
            set_error("zg_psc_round_expr: a term has at most 4 product tables and 4 linear-combination tables, indices below the table count; "
                      "ZG_PSC_PAIR_SUM needs exactly 4 product tables");
            return ZG_ERR_INVALID;
        }
        ex.t[ti].np = (uint32_t)n_prod;
        ex.t[ti].nq = (uint32_t)t.n_lin;
        ex.t[ti].pair_sum = pair_sum ? 1u : 0u;
        for (int j = 0; j < n_prod; j++) ex.t[ti].prod[j] = (uint32_t)t.prod[j];
        for (int m = 0; m < t.n_lin; m++) {
            ex.t[ti].lin[m] = (uint32_t)t.lin[m];
            for (int i = 0; i < 4; i++) {
                ex.t[ti].coeff[m].l[2 * i] = (uint32_t)t.lin_coeff[4 * m + i];
                ex.t[ti].coeff[m].l[2 * i + 1] = (uint32_t)(t.lin_coeff[4 * m + i] >> 32);
            }
        }
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    ex.points = s->points;
    if (s->evals_pending && s->have_spec && s->spec_is_expr && memcmp(&s->expr, &ex, sizeof(ex)) == 0) return psc_collect(s, ex.points, out);
    const size_t half = s->len / 2;
    unsigned nb = psc_blocks(half);
    uint32_t *counter = reinterpret_cast<uint32_t *>(s->d_misc + PSC_COUNTER_OFF);
    s->seq++;
    s->expr = ex;
    s->spec_is_expr = true;
    s->have_spec = true;
    s->evals_pending = true;
    FrArg none;
    memset(&none, 0, sizeof(none));
    if (psc_spread(half))
        hipLaunchKernelGGL((psc_expr_spread_kernel<false>), dim3(div_up(half * 16, 256)), dim3(256), 0, s->st, s->buf[s->cur], s->stride(), half, none,
                           (uint64_t *)nullptr, (size_t)0, ex, s->d_misc, s->h_pin, counter, s->h_pin + PSC_FLAG, s->seq);
    else
    hipLaunchKernelGGL((psc_expr_kernel<false>), dim3(nb), dim3(256), 0, s->st, s->buf[s->cur], s->stride(), half, none, (uint64_t *)nullptr, (size_t)0, ex,
                       s->d_misc, s->h_pin, counter, s->h_pin + PSC_FLAG, s->seq);
    return psc_collect(s, ex.points, out);
}

int zg_psc_round_gruen(zg_psc_t s, const int *prod_idx, size_t p, const uint64_t *d_e_out, size_t n_out, const uint64_t *d_e_in, size_t n_in,
                       uint64_t t0[4], uint64_t t_inf[4]) {
    ZG_INIT();
    PscSpec spec;
    ZG_TRY(psc_spec(s, prod_idx, p, nullptr, nullptr, 0, &spec, "zg_psc_round_gruen"));
    if (!t0 || !t_inf || !d_e_out || !d_e_in || n_out == 0 || n_in == 0 || (n_in & (n_in - 1)) || s->len < 2) {
        set_error("zg_psc_round_gruen: invalid argument (|E_in| must be a power of two) or a single entry left");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    const size_t half = s->len / 2;
    uint32_t in_bits = 0;
    while (((size_t)1 << in_bits) < n_in) in_bits++;
    s->have_spec = s->evals_pending = false;  // the E tables differ from round to round: nothing to prepare at the bind
    unsigned nb = psc_blocks(half);
    uint32_t *counter = reinterpret_cast<uint32_t *>(s->d_misc + PSC_COUNTER_OFF);
    uint64_t *flag = s->h_pin + PSC_FLAG;
    s->seq++;
    const uint64_t *base = s->buf[s->cur];
    switch (p) {
    case 1: hipLaunchKernelGGL((psc_gruen_kernel<1>), dim3(nb), dim3(256), 0, s->st, base, s->stride(), half, spec, d_e_out, n_out, d_e_in, in_bits, s->d_misc, s->h_pin, counter, flag, s->seq); break;
    case 2: hipLaunchKernelGGL((psc_gruen_kernel<2>), dim3(nb), dim3(256), 0, s->st, base, s->stride(), half, spec, d_e_out, n_out, d_e_in, in_bits, s->d_misc, s->h_pin, counter, flag, s->seq); break;
    case 3: hipLaunchKernelGGL((psc_gruen_kernel<3>), dim3(nb), dim3(256), 0, s->st, base, s->stride(), half, spec, d_e_out, n_out, d_e_in, in_bits, s->d_misc, s->h_pin, counter, flag, s->seq); break;
    default: hipLaunchKernelGGL((psc_gruen_kernel<4>), dim3(nb), dim3(256), 0, s->st, base, s->stride(), half, spec, d_e_out, n_out, d_e_in, in_bits, s->d_misc, s->h_pin, counter, flag, s->seq); break;
    }
    uint64_t h[8];
    ZG_TRY(psc_wait(s, h, 8));
    for (int i = 0; i < 4; i++) {
        t0[i] = h[i];
        t_inf[i] = h[4 + i];
    }
    return ZG_OK;
}

int zg_psc_bind(zg_psc_t s, const uint64_t r[4]) {
    ZG_INIT();
    if (!s || !r || s->len < 2) {
        set_error("zg_psc_bind: invalid session or a single entry left");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    FrArg ra;
    for (int i = 0; i < 4; i++) {
        ra.l[2 * i] = (uint32_t)r[i];
        ra.l[2 * i + 1] = (uint32_t)(r[i] >> 32);
    }
    const size_t half = s->len / 2;
    const int nxt = s->cur ^ 1;
    const size_t ostride = nxt == 0 ? s->cap : (s->cap / 2 ? s->cap / 2 : 1);
    unsigned nbx = (unsigned)((half + 255) / 256);
    if (nbx > 1024) nbx = 1024;
    PscTables rest;
    unsigned n_rest = 0;
    const bool fuse = s->have_spec && half >= 2;  // a next round exists and its product form is known
    s->evals_pending = false;
    if (fuse) {
        bool in_spec[ZG_PSC_MAX_TABLES] = {false};
        if (s->spec_is_expr) {
            for (uint32_t ti = 0; ti < s->expr.n_terms; ti++) {
                for (uint32_t j = 0; j < s->expr.t[ti].np; j++) in_spec[s->expr.t[ti].prod[j]] = true;
                for (uint32_t m = 0; m < s->expr.t[ti].nq; m++) in_spec[s->expr.t[ti].lin[m]] = true;
            }
        } else {
            for (size_t j = 0; j < s->spec_p; j++) in_spec[s->spec.prod[j]] = true;
            for (size_t m = 0; m < s->spec_q; m++) in_spec[s->spec.lin[m]] = true;
        }
        for (size_t t = 0; t < s->k; t++)
            if (!in_spec[t]) rest.t[n_rest++] = (uint32_t)t;
        const size_t quarter = half / 2;
        unsigned nb = psc_blocks(quarter);
        uint32_t *counter = reinterpret_cast<uint32_t *>(s->d_misc + PSC_COUNTER_OFF);
        s->seq++;
        const uint64_t *base = s->buf[s->cur];
        if (s->spec_is_expr) {
            if (psc_spread(quarter))
                hipLaunchKernelGGL((psc_expr_spread_kernel<true>), dim3(div_up(quarter * 16, 256)), dim3(256), 0, s->st, base, s->stride(), quarter, ra,
                                   s->buf[nxt], ostride, s->expr, s->d_misc, s->h_pin, counter, s->h_pin + PSC_FLAG, s->seq);
            else
            hipLaunchKernelGGL((psc_expr_kernel<true>), dim3(nb), dim3(256), 0, s->st, base, s->stride(), quarter, ra, s->buf[nxt], ostride, s->expr,
                               s->d_misc, s->h_pin, counter, s->h_pin + PSC_FLAG, s->seq);
        } else
        switch (s->spec_p) {
        case 0: psc_launch_fold_evals_q<0>(s->spec_q, nb, s->st, base, s->stride(), quarter, ra, s->buf[nxt], ostride, s->spec, s->d_misc, s->h_pin, counter, s->h_pin + PSC_FLAG, s->seq); break;
        case 1: psc_launch_fold_evals_q<1>(s->spec_q, nb, s->st, base, s->stride(), quarter, ra, s->buf[nxt], ostride, s->spec, s->d_misc, s->h_pin, counter, s->h_pin + PSC_FLAG, s->seq); break;
        case 2: psc_launch_fold_evals_q<2>(s->spec_q, nb, s->st, base, s->stride(), quarter, ra, s->buf[nxt], ostride, s->spec, s->d_misc, s->h_pin, counter, s->h_pin + PSC_FLAG, s->seq); break;
        case 3: psc_launch_fold_evals_q<3>(s->spec_q, nb, s->st, base, s->stride(), quarter, ra, s->buf[nxt], ostride, s->spec, s->d_misc, s->h_pin, counter, s->h_pin + PSC_FLAG, s->seq); break;
        default: psc_launch_fold_evals_q<4>(s->spec_q, nb, s->st, base, s->stride(), quarter, ra, s->buf[nxt], ostride, s->spec, s->d_misc, s->h_pin, counter, s->h_pin + PSC_FLAG, s->seq); break;
        }
        s->evals_pending = true;
    } else {
        for (size_t t = 0; t < s->k; t++) rest.t[n_rest++] = (uint32_t)t;
    }
    if (n_rest)  // the tables the product form does not name (or all of them), off the round's critical path
        hipLaunchKernelGGL(psc_fold_kernel, dim3(nbx, n_rest), dim3(256), 0, s->st, s->buf[s->cur], s->stride(), half, ra, s->buf[nxt], ostride, rest);
    ZG_HIP(hipGetLastError());  // asynchronous: the next round's kernel follows in stream order
    s->cur = nxt;
    s->len = half;
    return ZG_OK;
}

int zg_psc_read(zg_psc_t s, size_t table, uint64_t *out) {
    ZG_INIT();
    if (!s || !out || table >= s->k) {
        set_error("zg_psc_read: invalid argument");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    ZG_HIP(hipMemcpyAsync(out, s->buf[s->cur] + 4 * table * s->stride(), s->len * 32, hipMemcpyDeviceToHost, s->st));
    ZG_HIP(hipStreamSynchronize(s->st));
    return ZG_OK;
}

int zg_psc_table_dev(zg_psc_t s, size_t table, const uint64_t **d_ptr) {
    ZG_INIT();
    if (!s || !d_ptr || table >= s->k) {
        set_error("zg_psc_table_dev: invalid argument");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    ZG_HIP(hipStreamSynchronize(s->st));  // the folds enqueued so far have landed: any stream may read the table now
    *d_ptr = s->buf[s->cur] + 4 * table * s->stride();
    return ZG_OK;
}

int zg_psc_gather(zg_psc_t s, size_t table, const uint64_t *idx, size_t n, uint64_t *out) {
    ZG_INIT();
    if (!s || table >= s->k || (n && (!idx || !out))) {
        set_error("zg_psc_gather: invalid argument");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    return gather_to_host(s->buf[s->cur] + 4 * table * s->stride(), s->len, idx, n, out, s->st);
}

int zg_psc_final(zg_psc_t s, uint64_t *out) {
    ZG_INIT();
    if (!s || !out || s->len != 1) {
        set_error("zg_psc_final: protocol not complete");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    ZG_HIP(hipMemcpy2DAsync(out, 32, s->buf[s->cur], s->stride() * 32, 32, s->k, hipMemcpyDeviceToHost, s->st));
    ZG_HIP(hipStreamSynchronize(s->st));
    return ZG_OK;
}

int zg_psc_close(zg_psc_t s) {
    if (!s) return ZG_OK;
    ZG_INIT();
    DeviceGuard dg(s->device);
    (void)hipStreamSynchronize(s->st);
    static const bool pool_on = [] { const char *e = getenv("ZG_PSC_POOL"); return !(e && *e == '0'); }();
    if (pool_on) {
        std::lock_guard<std::mutex> lk(g_psc_pool_mu);
        if (g_psc_pool.size() < 8) {
            g_psc_pool.push_back(s);
            return ZG_OK;
        }
    }
    psc_free(s);
    return ZG_OK;
}

}  // extern "C"

namespace zg {
void psc_shutdown() {  // zg_shutdown: drop the pooled sessions
    std::lock_guard<std::mutex> lk(g_psc_pool_mu);
    for (zg_psc_s *s : g_psc_pool) {
        DeviceGuard dg(s->device);
        psc_free(s);
    }
    g_psc_pool.clear();
}
}  // namespace zg
