// test_host_mirror.cpp — the reference's inline tests for the hot path, restated against the C++ host
// mirror (zolt_amd/host/zolt_host.hpp) over libzolt_gpu.so. Each TEST names the Zig test it follows.
// Runs on the GPU box (tests/test_gpu_cpp_host.py); exit code 0 = all passed.
#include <cstdio>
#include <cstdlib>

#include "../../zolt_amd/host/zolt_host.hpp"

using namespace zolt;

static int g_failed = 0, g_run = 0;
#define EXPECT(cond)                                                                  \
    do {                                                                              \
        if (!(cond)) { std::printf("  FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); g_failed++; } \
    } while (0)
#define TEST(name) static void name(); static struct Reg_##name { Reg_##name() { tests().push_back({#name, name}); } } reg_##name; static void name()
struct T { const char *n; void (*f)(); };
static std::vector<T> &tests() { static std::vector<T> v; return v; }

static std::vector<AffinePoint> generator_multiples(size_t n) {  // points (i+1)G as in src/bench.zig:261-268
    std::vector<AffinePoint> pts;
    for (size_t i = 0; i < n; i++) pts.push_back(MSM::scalarMul(AffinePoint::generator(), Fr::fromU64(i + 1)));
    return pts;
}

// src/msm/mod.zig:802-825 "scalar mul by zero/one"
TEST(scalar_mul_by_zero_and_one) {
    AffinePoint g = AffinePoint::generator();
    EXPECT(MSM::scalarMul(g, Fr::zero()).isIdentity());
    EXPECT(MSM::scalarMul(g, Fr::one()).eql(g));
    EXPECT(MSM::scalarMul(AffinePoint::identity(), Fr::fromU64(5)).isIdentity());
}

// src/msm/mod.zig:875-891 "msm with zero scalars", :938-947 "parallel msm empty"
TEST(msm_zero_scalars_and_empty) {
    auto pts = generator_multiples(10);
    std::vector<Fr> zeros(10, Fr::zero());
    EXPECT(MSM::compute(pts, zeros).isIdentity());
    EXPECT(MSM::compute({}, {}).isIdentity());
    EXPECT(ParallelMSM::compute(pts, zeros, 4).isIdentity());
}

// src/integration_tests.zig:145-161 "msm scalar multiply consistency": 2*P == P + P
TEST(msm_scalar_multiply_consistency) {
    AffinePoint g = AffinePoint::generator();
    AffinePoint twice = MSM::scalarMul(g, Fr::fromU64(2));
    AffinePoint sum = MSM::compute({g, g}, {Fr::one(), Fr::one()});
    EXPECT(twice.eql(sum) && !twice.isIdentity());
    // identity bases are skipped (src/msm/mod.zig:407; src/integration_tests.zig:124-143)
    AffinePoint r = MSM::compute({AffinePoint::identity(), g, AffinePoint::identity()}, {Fr::fromU64(5), Fr::fromU64(3), Fr::fromU64(7)});
    EXPECT(r.eql(MSM::scalarMul(g, Fr::fromU64(3))));
}

// src/bench.zig:243-287 benchMSM inputs: points (i+1)G, scalars 7i+13, n = 16, 64, 256 — closed form
TEST(bench_msm_family_closed_form) {
    for (size_t n : {16u, 64u, 256u}) {
        auto pts = generator_multiples(n);
        std::vector<Fr> sc;
        uint64_t k = 0;
        for (size_t i = 0; i < n; i++) { sc.push_back(Fr::fromU64(7 * i + 13)); k += (7 * i + 13) * (i + 1); }
        AffinePoint got = MSM::compute(pts, sc);
        EXPECT(got.eql(MSM::scalarMul(AffinePoint::generator(), Fr::fromU64(k))));
    }
}

// src/poly/commitment/mod.zig:1392-1420 "hyperkzg batch commit", :239-243 empty -> identity
TEST(hyperkzg_batch_commit) {
    auto params = HyperKZG::setup(16);
    EXPECT(params.powers_of_tau_g1[0].eql(AffinePoint::generator()));
    std::vector<std::vector<Fr>> polys(3);
    for (int p = 0; p < 3; p++)
        for (int i = 0; i < 16; i++) polys[p].push_back(Fr::fromU64((uint64_t)(p * 100 + i * 7 + 1)));
    auto batch = HyperKZG::batchCommit(params, polys);
    for (int p = 0; p < 3; p++) EXPECT(batch[p].eql(HyperKZG::commit(params, polys[p])));
    EXPECT(HyperKZG::commit(params, {}).point.isIdentity());
    // commit(e_1) = tau*G = powers[1]
    std::vector<Fr> e1(16, Fr::zero());
    e1[1] = Fr::one();
    EXPECT(HyperKZG::commit(params, e1).point.eql(params.powers_of_tau_g1[1]));
}

// batchOpen of ONE polynomial is open() of that polynomial (gamma^0 = 1); evaluations are the multilinear evaluations;
// the batching challenge follows :633-640
TEST(hyperkzg_batch_open_single_equals_open) {
    auto params = HyperKZG::setup(8);
    std::vector<Fr> e, pt;
    for (int i = 0; i < 8; i++) e.push_back(Fr::fromU64((uint64_t)(3 * i + 2)));
    for (int j = 0; j < 3; j++) pt.push_back(Fr::fromU64((uint64_t)(5 + j)));
    auto single = HyperKZG::open(params, e, pt, Fr::zero());
    auto batch = HyperKZG::batchOpen(params, {e}, pt);
    EXPECT(batch.quotient_commitments.size() == 3 && batch.evaluations.size() == 1);
    for (size_t i = 0; i < 3; i++) EXPECT(batch.quotient_commitments[i].eql(single.quotient_commitments[i]));
    EXPECT(batch.final_eval.eql(single.final_eval));
    EXPECT(batch.evaluations[0].eql(DensePolynomial(e).evaluate(pt)));
    Fr g = Fr::fromU64(0x9a8b7c6dULL);
    for (auto &r : pt) g = g.mul(r.add(Fr::fromU64(11)));
    EXPECT(batch.batching_challenge.eql(g));
}

// src/poly/commitment/mod.zig:1448-1472 "hyperkzg multilinear evaluation" + open(): the final evaluation of open() at a
// boolean point is the table entry it selects (high variable first), and quotient 0 is commit(hi - lo)
TEST(hyperkzg_open_corner_points) {
    auto params = HyperKZG::setup(8);
    std::vector<Fr> evals;
    for (int i = 0; i < 8; i++) evals.push_back(Fr::fromU64(100 + 7 * i));
    for (int idx = 0; idx < 8; idx++) {
        std::vector<Fr> pt = {Fr::fromU64((idx >> 2) & 1), Fr::fromU64((idx >> 1) & 1), Fr::fromU64(idx & 1)};
        auto pr = HyperKZG::open(params, evals, pt, Fr::zero());
        EXPECT(pr.final_eval.eql(evals[idx]));
        EXPECT(pr.quotient_commitments.size() == 3);
    }
    std::vector<Fr> q0;
    for (int j = 0; j < 4; j++) q0.push_back(evals[j + 4].sub(evals[j]));
    auto pr = HyperKZG::open(params, evals, {Fr::fromU64(3), Fr::fromU64(5), Fr::fromU64(9)}, Fr::zero());
    EXPECT(pr.quotient_commitments[0].eql(HyperKZG::commit(params, q0)));
    EXPECT(HyperKZG::open(params, evals, {}, Fr::fromU64(42)).final_eval.eql(Fr::fromU64(42)));
}

// src/poly/mod.zig:816-888 "dense polynomial bindLow"
TEST(dense_polynomial_bind_low) {
    DensePolynomial p({Fr::fromU64(1), Fr::fromU64(2), Fr::fromU64(3), Fr::fromU64(4)});
    p.bindLow(Fr::fromU64(3));
    EXPECT(p.num_vars == 1 && p.evaluations[0].eql(Fr::fromU64(4)) && p.evaluations[1].eql(Fr::fromU64(6)));
    std::vector<Fr> e;
    for (int i = 1; i <= 8; i++) e.push_back(Fr::fromU64(10 * i));
    DensePolynomial q(e);
    q.bindLow(Fr::fromU64(5));
    uint64_t want[4] = {60, 80, 100, 120};
    for (int i = 0; i < 4; i++) EXPECT(q.evaluations[i].eql(Fr::fromU64(want[i])));
}

// src/poly/mod.zig:689-753 "EqPolynomial partition of unity"
TEST(eq_polynomial_partition_of_unity) {
    std::vector<Fr> r = {Fr::fromU64(12345), Fr::fromU64(67890), Fr::fromU64(0x123456789abcdefULL)};
    auto ev = EqPolynomial(r).evals();
    EXPECT(ev.size() == 8);
    Fr sum = Fr::zero();
    for (auto &x : ev) sum = sum.add(x);
    EXPECT(sum.eql(Fr::one()));
    // index 0 = prod (1 - r_i); index 2^n - 1 = prod r_i; MSB <-> r[0]
    Fr one = Fr::one();
    EXPECT(ev[0].eql(one.sub(r[0]).mul(one.sub(r[1])).mul(one.sub(r[2]))));
    EXPECT(ev[7].eql(r[0].mul(r[1]).mul(r[2])));
    EXPECT(ev[4].eql(r[0].mul(one.sub(r[1])).mul(one.sub(r[2]))));
}

// src/subprotocols/mod.zig:366-439 "sumcheck prover round generation"
TEST(sumcheck_prover_round_generation) {
    DensePolynomial poly({Fr::fromU64(1), Fr::fromU64(2), Fr::fromU64(3), Fr::fromU64(4)});
    Sumcheck::Prover prover(poly);
    auto round1 = prover.nextRound();
    EXPECT(round1.poly.evaluate(Fr::zero()).eql(Fr::fromU64(3)));
    EXPECT(round1.poly.evaluate(Fr::one()).eql(Fr::fromU64(7)));
    Fr r0 = Fr::fromU64(2);
    prover.receiveChallenge(r0);
    auto round2 = prover.nextRound();
    Fr a = round2.poly.evaluate(Fr::zero()), b = round2.poly.evaluate(Fr::one());
    EXPECT(a.eql(Fr::fromU64(5)) && b.eql(Fr::fromU64(6)));
    EXPECT(a.add(b).eql(round1.poly.evaluate(r0)));
}

// src/subprotocols/mod.zig:441-461 "sumcheck complete protocol"
TEST(sumcheck_complete_protocol) {
    std::vector<Fr> e;
    for (int i = 1; i <= 8; i++) e.push_back(Fr::fromU64(i));
    auto res = runSumcheck(DensePolynomial(e));
    EXPECT(res.result);
    EXPECT(res.proof.claim.eql(Fr::fromU64(36)));
    EXPECT(res.proof.rounds.size() == 3 && res.proof.final_point.size() == 3);
    // the host-verifier form (one device round trip per round) produces the same transcript
    auto inter = runSumcheckInteractive(DensePolynomial(e));
    EXPECT(inter.result && inter.proof.claim.eql(res.proof.claim) && inter.proof.final_eval.eql(res.proof.final_eval));
    for (size_t i = 0; i < 3; i++) {
        EXPECT(inter.proof.final_point[i].eql(res.proof.final_point[i]));
        EXPECT(inter.proof.rounds[i].poly.coeffs[0].eql(res.proof.rounds[i].poly.coeffs[0]));
        EXPECT(inter.proof.rounds[i].poly.coeffs[1].eql(res.proof.rounds[i].poly.coeffs[1]));
    }
}

// src/poly/commitment/mod.zig:1240-1258 "generator affine double == jacobian double"; AffinePoint.add (src/msm/mod.zig:74-103):
// P + P takes the doubling branch, P + (-P) is the identity, identity operands pass the other point through
TEST(affine_point_add_and_double) {
    AffinePoint g = AffinePoint::generator();
    AffinePoint two = MSM::scalarMul(g, Fr::fromU64(2)), three = MSM::scalarMul(g, Fr::fromU64(3));
    EXPECT(g.dbl().eql(two));
    EXPECT(g.add(g).eql(two));
    EXPECT(g.add(two).eql(three) && two.add(g).eql(three));
    AffinePoint neg = MSM::scalarMul(g, Fr::zero().sub(Fr::one()));  // (r - 1) G = -G
    EXPECT(g.add(neg).isIdentity());
    EXPECT(g.add(AffinePoint::identity()).eql(g) && AffinePoint::identity().add(g).eql(g));
    EXPECT(AffinePoint::identity().dbl().isIdentity());
}

// src/poly/mod.zig:94-126 DensePolynomial.add / scale; :214-227,311-321 EqPolynomial.evaluate / mle
TEST(dense_polynomial_add_scale_and_eq_mle) {
    DensePolynomial a({Fr::fromU64(1), Fr::fromU64(2), Fr::fromU64(3), Fr::fromU64(4)});
    DensePolynomial b({Fr::fromU64(10), Fr::fromU64(20), Fr::fromU64(30), Fr::fromU64(40)});
    auto s = a.add(b);
    auto t = a.scale(Fr::fromU64(7));
    for (int i = 0; i < 4; i++) {
        EXPECT(s.evaluations[i].eql(Fr::fromU64(11 * (i + 1))));
        EXPECT(t.evaluations[i].eql(Fr::fromU64(7 * (i + 1))));
    }
    std::vector<Fr> r = {Fr::fromU64(12345), Fr::fromU64(67890), Fr::fromU64(0x123456789abcdefULL)};
    auto ev = EqPolynomial(r).evals();
    for (int idx = 0; idx < 8; idx++) {  // mle at a boolean point = the table entry (index MSB <-> r[0])
        std::vector<Fr> x = {Fr::fromU64((idx >> 2) & 1), Fr::fromU64((idx >> 1) & 1), Fr::fromU64(idx & 1)};
        EXPECT(EqPolynomial::mle(r, x).eql(ev[idx]));
        EXPECT(EqPolynomial(r).evaluate(x).eql(ev[idx]));
    }
    EXPECT(EqPolynomial::mle({}, {}).eql(Fr::one()));
}

// src/msm/mod.zig:949-966 "parallel msm" through the several-GPU entry points (one shard per bound device); :683-748 batch form
TEST(parallel_msm_equals_msm) {
    auto pts = generator_multiples(64);
    std::vector<Fr> sc;
    for (size_t i = 0; i < 64; i++) sc.push_back(Fr::fromU64(1000003 * i + 17));
    EXPECT(ParallelMSM::compute(pts, sc, 8).eql(MSM::compute(pts, sc)));
    std::vector<std::vector<Fr>> batches(3, sc);
    batches[1][5] = Fr::fromU64(99);
    auto out = ParallelBatchMSM::compute(pts, batches);
    EXPECT(out.size() == 3);
    for (size_t j = 0; j < 3; j++) EXPECT(out[j].eql(MSM::compute(pts, batches[j])));
}

// src/transcripts/mod.zig:49-161 — the host Keccak transcript. Expected limbs generated with the C oracle's restatement
// (oracle.binding.Transcript: init "Jolt", appendScalar("round_poly_0", fromU64(7)), challengeScalar("spartan_round"); then
// 200 bytes 'q' and challengeScalar("x")), itself pinned against hashlib.sha3_256 (tests/test_transcript_host.py).
TEST(keccak_transcript_kat) {
    Transcript t("Jolt");
    t.appendScalar("round_poly_0", Fr::fromU64(7));
    Fr c = t.challengeScalar("spartan_round");
    Fr want{{0x1719faabca06f284ULL, 0x61cd547aaa4e2a6fULL, 0xee71ee5ce38605c0ULL, 0x01db5fa4e23c262bULL}};
    EXPECT(c.eql(want));
    t.appendBytes(std::string(200, 'q'));
    Fr c2 = t.challengeScalar("x");
    Fr want2{{0x769c3385e142fb3cULL, 0x9a870a5588cbfe6eULL, 0x201d82b576506a20ULL, 0x16cd0b4530a610dfULL}};
    EXPECT(c2.eql(want2));
}

// RaPolynomial.finalClaim (src/zkvm/ram/raf_checking.zig:179-185) returns evals[0] at ANY point of the protocol; the mirror used to answer
// zero until the table had one entry (round-5 advisor)
TEST(raf_final_claim_before_the_last_bind) {
    std::vector<Fr> ra;
    for (uint64_t i = 0; i < 8; i++) ra.push_back(Fr::fromU64(5 + 3 * i));
    RafEvaluationProver p(ra, 0x7fff8000ULL);
    EXPECT(p.getFinalClaim().eql(ra[0]));
    Fr c = Fr::fromU64(11);
    p.bindChallenge(c);  // LowToHigh: new[0] = ra[0] + c (ra[1] - ra[0])
    EXPECT(p.getFinalClaim().eql(ra[0].add(c.mul(ra[1].sub(ra[0])))));
    p.bindChallenge(c);
    p.bindChallenge(c);  // one entry left: the session's final value
    Fr a = ra[0].add(c.mul(ra[1].sub(ra[0]))), b = ra[2].add(c.mul(ra[3].sub(ra[2])));
    Fr e = ra[4].add(c.mul(ra[5].sub(ra[4]))), f = ra[6].add(c.mul(ra[7].sub(ra[6])));
    Fr ab = a.add(c.mul(b.sub(a))), ef = e.add(c.mul(f.sub(e)));
    EXPECT(p.getFinalClaim().eql(ab.add(c.mul(ef.sub(ab)))));
}

// src/poly/split_eq.zig:525-733 — the reference's six GruenSplitEqPolynomial tests, restated
TEST(gruen_split_eq_polynomial) {
    auto F = [](uint64_t v) { return Fr::fromU64(v); };
    {  // "initialization" (:525-553) and "prefix tables correctness" (:578-611)
        GruenSplitEqPolynomial p({F(2), F(3), F(5)});
        EXPECT(p.current_index == 3 && p.current_scalar.eql(Fr::one()));
        EXPECT(p.num_x_in == 1 && p.num_x_out == 1 && p.E_in_vec.size() == 2 && p.E_out_vec.size() == 2);
        EXPECT(p.E_out_vec[1].size() == 2 && p.E_out_vec[1][0].eql(Fr::one().sub(F(2))) && p.E_out_vec[1][1].eql(F(2)));
        EXPECT(p.E_in_vec[1].size() == 2 && p.E_in_vec[1][0].eql(Fr::one().sub(F(3))) && p.E_in_vec[1][1].eql(F(3)));
        EXPECT(p.E_out_vec[0].size() == 1 && p.E_out_vec[0][0].eql(Fr::one()));
    }
    {  // "bind updates scalar" (:555-576): eq(3, 5) = 15 + (-2)(-4) = 23
        GruenSplitEqPolynomial p({F(2), F(3)});
        p.bind(F(5));
        EXPECT(p.current_index == 1 && p.current_scalar.eql(F(23)));
    }
    {  // "cubic round poly basic" (:613-633)
        GruenSplitEqPolynomial p({F(1), F(2)});
        auto rp = p.computeCubicRoundPoly(F(10), F(3), F(100));
        EXPECT(rp[0].add(rp[1]).eql(F(100)));
    }
    {  // "big-endian eq table correctness" (:635-681) and "getEActiveForWindow" (:683-733)
        GruenSplitEqPolynomial p({F(3), F(5), F(7), F(11)});
        auto t = p.getFullEqTable();
        Fr o = Fr::one(), m3 = o.sub(F(3)), m5 = o.sub(F(5)), m7 = o.sub(F(7)), m11 = o.sub(F(11));
        EXPECT(t.size() == 16);
        EXPECT(t[0].eql(m3.mul(m5).mul(m7).mul(m11)) && t[15].eql(F(3 * 5 * 7 * 11)));
        EXPECT(t[5].eql(m3.mul(F(5)).mul(m7).mul(F(11))) && t[10].eql(F(3).mul(m5).mul(F(7)).mul(m11)));
        auto a1 = p.getEActiveForWindow(1), a2 = p.getEActiveForWindow(2), a3 = p.getEActiveForWindow(3);
        EXPECT(a1.size() == 1 && a1[0].eql(o));
        EXPECT(a2.size() == 2 && a2[0].eql(m7) && a2[1].eql(F(7)));
        EXPECT(a3.size() == 4 && a3[0].eql(m5.mul(m7)) && a3[1].eql(m5.mul(F(7))) && a3[2].eql(F(5).mul(m7)) && a3[3].eql(F(35)));
        auto w = p.getWindowEqTables(0, 1);  // head_len 3, m 2: E_out over 2 bits, E_in over 1
        EXPECT(w.E_out->size() == 4 && w.E_in->size() == 2 && w.head_in_bits == 1);
    }
}

// src/zkvm/lasso/prover.zig:553-688 — "lasso prover basic" / "rounds" / "claim tracking": p(0) + p(1) = current_claim before every
// round, current_claim = p(challenge) after it; then runLassoProver's invariants over a ragged lookup set
TEST(lasso_prover_claim_tracking) {
    std::vector<unsigned __int128> idx = {0, 1, 2, 3};
    std::vector<Fr> r_reduction = {Fr::fromU64(2), Fr::fromU64(3)};
    {
        LassoProver p(idx, 2, 3, r_reduction);
        EXPECT(p.round == 0 && p.isAddressPhase() && !p.isComplete());
        EXPECT(!p.computeRoundPolynomial().coeffs.empty());
        p.receiveChallenge(Fr::fromU64(7));
        EXPECT(p.round == 1);
    }
    LassoProver p(idx, 2, 3, r_reduction);
    for (size_t round = 0; round < 5; round++) {
        Fr claim = p.current_claim;
        UniPoly rp = p.computeRoundPolynomial();
        Fr p1 = rp.coeffs[0].add(rp.coeffs[1]).add(rp.coeffs[2]);
        EXPECT(rp.coeffs[0].add(p1).eql(claim));
        Fr ch = Fr::fromU64(round + 10);
        p.receiveChallenge(ch);
        EXPECT(p.current_claim.eql(rp.evaluate(ch)));
    }
    EXPECT(p.isComplete());
    std::vector<unsigned __int128> big;
    for (unsigned i = 0; i < 37; i++) big.push_back(((unsigned __int128)(i * 0x9e3779b97f4a7c15ULL) << 17) ^ (i * 77u));
    std::vector<Fr> w;
    for (unsigned i = 0; i < 6; i++) w.push_back(Fr::fromU64(1000003 * i + 5));
    LassoProof proof = runLassoProver(big, 6, 81, w);
    EXPECT(proof.round_polys.size() == 87 && proof.challenges.size() == 87);
    for (size_t i = 0; i + 1 < proof.round_polys.size(); i++) {  // p_i(r_i) = p_{i+1}(0) + p_{i+1}(1)
        const UniPoly &a = proof.round_polys[i], &b = proof.round_polys[i + 1];
        Fr next = b.coeffs[0].add(b.coeffs[0].add(b.coeffs[1]).add(b.coeffs[2]));
        EXPECT(a.evaluate(proof.challenges[i]).eql(next));
    }
}

// ValEvaluationProver (src/zkvm/ram/val_evaluation.zig:545-700) and ProductVirtualRemainderProver (src/zkvm/spartan/product_remainder.zig:269-394)
// as sumchecks: with the claim = sum over the hypercube every round has s(0) + s(1) = claim and the last claim is the product of the
// tables' final values (times the split-eq scalar)
TEST(product_form_provers_are_sound) {
    const size_t v = 6, n = size_t(1) << v;
    auto mk = [&](uint64_t seed) {
        std::vector<Fr> t(n);
        uint64_t x = seed;
        for (auto &e : t) {
            x = x * 6364136223846793005ULL + 1442695040888963407ULL;
            e = Fr::fromU64(x >> 7).mul(Fr::fromU64(x | 1));
        }
        return t;
    };
    auto inc = mk(1), wa = mk(2), lt = mk(3);
    Fr claim = Fr::zero();
    for (size_t i = 0; i < n; i++) claim = claim.add(inc[i].mul(wa[i]).mul(lt[i]));
    ValEvaluationProver p(inc, wa, &lt, claim);
    for (size_t r = 0; r < v; r++) {
        auto ev = p.computeRoundPolynomial();
        EXPECT(ev[0].add(ev[1]).eql(p.current_claim));
        p.bindChallengeWithPoly(Fr::fromU64(1000 + r).mul(inc[r]), ev);
    }
    auto f = p.getFinalClaims();
    EXPECT(f.size() == 3 && f[0].mul(f[1]).mul(f[2]).eql(p.current_claim));

    auto left = mk(4), right = mk(5);
    std::vector<Fr> tau;
    for (size_t i = 0; i < v; i++) tau.push_back(wa[i]);
    Fr kernel = lt[0];
    auto eq = EqPolynomial::evalsSliceWithScaling(tau, &kernel);
    Fr c2 = Fr::zero();
    for (size_t i = 0; i < n; i++) c2 = c2.add(eq[i].mul(left[i]).mul(right[i]));
    ProductVirtualRemainderProver q(left, right, tau, kernel, c2);
    for (size_t r = 0; r < v; r++) {
        std::array<Fr, 4> ev;
        EXPECT(q.roundEvals(ev));
        EXPECT(ev[0].add(ev[1]).eql(q.current_claim));
        auto comp = q.computeRoundPolynomial();
        EXPECT(comp[0].eql(ev[0]));
        Fr ch = Fr::fromU64(77 + r).mul(inc[r + 8]);
        q.bindChallenge(ch);
        q.updateClaim(ev, ch);
    }
    EXPECT(q.getFinalClaim().mul(q.split_eq.current_scalar).eql(q.current_claim));
}

// src/transcripts/blake2b.zig — the C++ Blake2b transcript against values of the Python mirror (api.Blake2bTranscript), which is held
// against the states the reference printed (tests/golden/blake2b_transcript_preamble.json): init("Jolt"), appendU64, appendMessage,
// appendScalar, a 200-byte appendBytes, then challengeScalarFull and challengeScalar
TEST(blake2b_transcript_kat) {
    auto hex = [](const uint8_t *b, size_t n) {
        std::string s;
        char t[3];
        for (size_t i = 0; i < n; i++) { std::snprintf(t, 3, "%02x", b[i]); s += t; }
        return s;
    };
    Blake2bTranscript t("Jolt");
    EXPECT(hex(t.state, 32) == "06ce2c10d1d2801c48c859d7cb16510476b0d48667d9562ed021b20d9a05e547");
    t.appendU64(0x1122334455667788ULL);
    t.appendMessage("UniPoly_begin");
    t.appendScalar(Fr::fromU64(123456789));
    uint8_t data[200];
    for (int i = 0; i < 200; i++) data[i] = (uint8_t)i;
    t.appendBytes(data, 200);
    EXPECT(hex(t.state, 32) == "99d58a4ad34a2480b75e52df4cd885908f0cd089674e56f5fdd656ee62450ce0" && t.n_rounds == 4);
    Fr full = t.challengeScalarFull();
    Fr want_full{{0xcd111d86aac77a34ULL, 0x84437048f6f30ccfULL, 0x3dd77a39079b4474ULL, 0x8e52df5bb84bdbULL}};
    EXPECT(full.eql(want_full));
    Fr ch = t.challengeScalar();
    Fr want_ch{{0, 0, 0xcc31546f28f7fe6cULL, 0x5540a1100b32435ULL}};
    EXPECT(ch.eql(want_ch));
}

// src/zkvm/batched_sumcheck.zig:77-430 — five device-backed instances of different length under one Blake2b transcript (the shape of
// Stage 2): with consistent input claims every combined round satisfies s(0) + s(1) = claim, and the final claim is the
// coefficient-weighted sum of the instances' own final claims
TEST(batched_sumcheck_stage2_shape) {
    auto mk = [&](size_t n, uint64_t seed) {
        std::vector<Fr> t(n);
        uint64_t x = seed;
        for (auto &e : t) {
            x = x * 6364136223846793005ULL + 1442695040888963407ULL;
            e = Fr::fromU64(x >> 7).mul(Fr::fromU64(x | 1));
        }
        return t;
    };
    // ProductVirtualRemainder, 5 rounds
    auto left = mk(32, 1), right = mk(32, 2);
    std::vector<Fr> tau = mk(5, 3);
    Fr kernel = Fr::fromU64(77);
    auto eq = EqPolynomial::evalsSliceWithScaling(tau, &kernel);
    Fr c_pv = Fr::zero();
    for (size_t i = 0; i < 32; i++) c_pv = c_pv.add(eq[i].mul(left[i]).mul(right[i]));
    ProductVirtualRemainderProver pv(left, right, tau, kernel, c_pv);
    // ValEvaluation, 9 rounds
    auto inc = mk(512, 4), wa = mk(512, 5), lt = mk(512, 6);
    Fr c_ve = Fr::zero();
    for (size_t i = 0; i < 512; i++) c_ve = c_ve.add(inc[i].mul(wa[i]).mul(lt[i]));
    ValEvaluationProver ve(inc, wa, &lt, c_ve);
    // OutputSumcheck, 7 rounds
    auto e2 = mk(128, 7), io = mk(128, 8), vf = mk(128, 9), vio = mk(128, 10), vinit = mk(128, 11);
    Fr c_op = Fr::zero();
    for (size_t i = 0; i < 128; i++) c_op = c_op.add(e2[i].mul(io[i]).mul(vf[i].sub(vio[i])));
    OutputSumcheckProver op(e2, io, vf, vio, vinit, c_op);
    // InstructionLookups claim reduction, 5 rounds
    auto e3 = mk(32, 12), lo = mk(32, 13), lf = mk(32, 14), rt = mk(32, 15);
    Fr gamma = Fr::fromU64(991), c_il = Fr::zero();
    for (size_t i = 0; i < 32; i++) c_il = c_il.add(e3[i].mul(lo[i].add(gamma.mul(lf[i])).add(gamma.mul(gamma).mul(rt[i]))));
    InstructionLookupsClaimReductionProver il(e3, lo, lf, rt, gamma, c_il);

    std::array<Fr, 4> last_pv, last_ve, last_op, last_il;
    BatchedSumcheckProver p;
    p.addInstance({5, 3, c_pv, [&](size_t) { pv.roundEvals(last_pv); return last_pv; }, [&](const Fr &c) { pv.updateClaim(last_pv, c); pv.bindChallenge(c); }});
    p.addInstance({9, 3, c_ve, [&](size_t) { last_ve = ve.computeRoundPolynomial(); return last_ve; }, [&](const Fr &c) { ve.bindChallengeWithPoly(c, last_ve); }});
    p.addInstance({7, 3, c_op, [&](size_t) { last_op = op.roundEvals(); return last_op; }, [&](const Fr &c) { op.updateClaim(last_op, c); op.bindChallenge(c); }});
    p.addInstance({5, 2, c_il, [&](size_t) { last_il = il.computeRoundPolynomialCubic(); return last_il; }, [&](const Fr &c) { il.updateClaim(last_il, c); il.bindChallenge(c); }});
    Blake2bTranscript tr("Jolt");
    p.setupBatching(tr);
    EXPECT(p.max_num_rounds == 9);
    for (size_t k = 0; k < 9; k++) {
        Fr claim = p.current_claim;
        auto ev = p.combinedEvals();
        EXPECT(ev[0].add(ev[1]).eql(claim));
        auto comp = evalsToCompressed(ev);
        auto back = decompressRoundPoly(comp, claim);
        for (int j = 0; j < 4; j++) EXPECT(back[j].eql(ev[j]));
        tr.appendMessage("UniPoly_begin");
        for (const Fr &c : comp) tr.appendScalar(c);
        tr.appendMessage("UniPoly_end");
        Fr ch = tr.challengeScalar();
        p.updateClaim(ev, ch);
        p.bindChallenge(ch);
    }
    Fr want = pv.current_claim.mul(p.batching_coeffs[0]).add(ve.current_claim.mul(p.batching_coeffs[1]))
                  .add(op.current_claim.mul(p.batching_coeffs[2])).add(il.current_claim.mul(p.batching_coeffs[3]));
    EXPECT(p.current_claim.eql(want));
    // and each instance ended on the product form of its tables' final values
    auto f = ve.getFinalClaims();
    EXPECT(f[0].mul(f[1]).mul(f[2]).eql(ve.current_claim));
    auto g = op.finalValues();
    EXPECT(g[0].mul(g[1]).mul(g[2].sub(g[3])).eql(op.current_claim));
    auto h = il.finalValues();
    EXPECT(h[0].mul(h[1].add(gamma.mul(h[2])).add(gamma.mul(gamma).mul(h[3]))).eql(il.current_claim));
    EXPECT(pv.getFinalClaim().mul(pv.split_eq.current_scalar).eql(pv.current_claim));
}

// InstructionInputProver (src/zkvm/spartan/stage3_prover.zig:2029-2150) as a sumcheck, computeClaimedInputs (src/zkvm/r1cs/evaluation.zig:55-122)
// against the eq-table definition, and the eq+1 table (src/poly/mod.zig:530-548) as the eq table moved up one entry
TEST(stage3_and_opening_claim_sites) {
    const size_t v = 6, n = size_t(1) << v;
    auto mk = [&](uint64_t seed, size_t len) {
        std::vector<Fr> t(len);
        uint64_t x = seed;
        for (auto &e : t) {
            x = x * 6364136223846793005ULL + 1442695040888963407ULL;
            e = Fr::fromU64(x >> 9).mul(Fr::fromU64(x | 1));
        }
        return t;
    };
    std::vector<std::vector<Fr>> T;
    for (int j = 0; j < 10; j++) T.push_back(mk(100 + j, n));
    Fr gamma = Fr::fromU64(12345), g2 = gamma.mul(gamma);
    Fr claim = Fr::zero();
    for (size_t i = 0; i < n; i++) {
        Fr left = T[0][i].mul(T[1][i]).add(T[2][i].mul(T[3][i])), right = T[4][i].mul(T[5][i]).add(T[6][i].mul(T[7][i]));
        claim = claim.add(T[8][i].add(g2.mul(T[9][i])).mul(right.add(gamma.mul(left))));
    }
    std::vector<const std::vector<Fr> *> ptrs;
    for (auto &t : T) ptrs.push_back(&t);
    InstructionInputProver p(ptrs, gamma);
    for (size_t r = 0; r < v; r++) {
        auto ev = p.computeRoundEvals(claim);  // p(1) is derived, so check the cubic through p(0), p(2), p(3) against the next round instead
        Fr ch = Fr::fromU64(900 + r).mul(T[0][r]);
        claim = cubicAtPoint(ev, ch);
        p.bind(ch);
    }
    auto f = p.finalClaims();
    Fr left = f[0].mul(f[1]).add(f[2].mul(f[3])), right = f[4].mul(f[5]).add(f[6].mul(f[7]));
    EXPECT(f[8].add(g2.mul(f[9])).mul(right.add(gamma.mul(left))).eql(claim));

    const size_t k = 36, cycles = 64;
    auto W = mk(7, cycles * k);
    std::vector<Fr> rc = mk(8, 6);
    auto eq = EqPolynomial(rc).evals();
    auto got = computeClaimedInputs(W, k, rc);
    for (size_t i : {size_t(0), size_t(17), k - 1}) {
        Fr want = Fr::zero();
        for (size_t t = 0; t < cycles; t++) want = want.add(eq[t].mul(W[t * k + i]));
        EXPECT(got[i].eql(want));
    }
    auto e1 = eqPlusOneEvals(rc);
    EXPECT(e1.size() == 64 && e1[0].isZero());
    for (size_t j = 1; j < 64; j++) EXPECT(e1[j].eql(eq[j - 1]));
}

// proveStage5 / proveStage6 (src/zkvm/prover.zig:829-1112) on the device against the same loops written out with host scalars
TEST(stage_4_val_evaluation) {
    // LtPolynomial at a boolean point is the indicator j < r (the reference's "lt polynomial basic", all points of three variables)
    for (unsigned r = 0; r < 8; r++) {
        std::vector<Fr> bits = {Fr::fromU64(r & 1), Fr::fromU64((r >> 1) & 1), Fr::fromU64((r >> 2) & 1)}, tab(8);
        check(zg_fr_lt_table(reinterpret_cast<const uint64_t *>(bits.data()), 3, reinterpret_cast<uint64_t *>(tab.data())), "zg_fr_lt_table");
        for (unsigned j = 0; j < 8; j++) EXPECT(tab[j].eql(j < r ? Fr::one() : Fr::zero()));
    }
    for (size_t trace_len : {size_t(1), size_t(37), size_t(1000)}) {
        const size_t log_k = 5, log_t = 10, K = size_t(1) << log_k;
        const uint64_t start = 0x80000000ULL;
        std::vector<MemoryAccess> acc;
        uint64_t z = 0xabc + trace_len;
        auto next = [&]() { z = z * 6364136223846793005ULL + 1442695040888963407ULL; return z >> 20; };
        for (size_t ts = 0; ts < trace_len; ts++)
            if (next() % 3) {
                acc.push_back(MemoryAccess{ts, start + 8 * (next() % (K + 2)), (next() & 1) != 0, next()});
                if (next() % 5 == 0) acc.push_back(MemoryAccess{ts, start + 8 * (next() % K) + (next() % 3), true, next()});  // a second, unaligned write in the cycle: the later one stays
            }
        std::vector<std::pair<uint64_t, uint64_t>> init = {{start + 8, 77}, {start + 8 * 3, 1234567}};
        Transcript ta("Jolt"), tb("Jolt");
        ta.appendBytes("stages 1-3"); tb.appendBytes("stages 1-3");
        auto got = proveStage4(acc, init, trace_len, log_k, log_t, start, ta);
        // host restatement: tables by the reference's formulas, rounds by plain loops
        std::vector<Fr> ra, rc;
        for (size_t i = 0; i < log_k; i++) ra.push_back(tb.challengeScalar("r_address"));
        for (size_t i = 0; i < log_t; i++) rc.push_back(tb.challengeScalar("r_cycle_val"));
        size_t n = 1;
        while (n < trace_len) n <<= 1;
        std::vector<Fr> inc(n, Fr::zero()), wa(n, Fr::zero()), lt(n, Fr::zero());
        std::map<uint64_t, uint64_t> last;
        for (auto &kv : init) last[kv.first] = kv.second;
        for (auto &a : acc) {
            if (!a.is_write || (a.address - start) / 8 >= K || a.timestamp >= trace_len) continue;
            uint64_t old = last.count(a.address) ? last[a.address] : 0;
            inc[a.timestamp] = a.value >= old ? Fr::fromU64(a.value - old) : Fr::zero().sub(Fr::fromU64(old - a.value));
            last[a.address] = a.value;
            Fr e = Fr::one();
            for (size_t i = 0; i < log_k; i++) e = e.mul((((a.address - start) / 8) >> i) & 1 ? ra[i] : Fr::one().sub(ra[i]));
            wa[a.timestamp] = e;
        }
        for (size_t j = 0; j < n; j++)
            for (size_t i = 0; i < log_t; i++)
                if (!((j >> i) & 1)) {
                    Fr c = rc[i];
                    for (size_t k2 = i + 1; k2 < log_t; k2++) c = c.mul((j >> k2) & 1 ? rc[k2] : Fr::one().sub(rc[k2]));
                    lt[j] = lt[j].add(c);
                }
        Fr claim = Fr::zero();
        for (size_t j = 0; j < n; j++) claim = claim.add(inc[j].mul(wa[j]).mul(lt[j]));
        EXPECT(got.initial_claim.eql(claim));
        const size_t rounds = trace_len <= 1 ? 0 : log2Ceil(trace_len);
        EXPECT(got.round_polys.size() == rounds);
        for (size_t rd = 0; rd < rounds; rd++) {
            const size_t half = n >> (rd + 1);
            std::array<Fr, 4> ev = {Fr::zero(), Fr::zero(), Fr::zero(), Fr::zero()};
            for (size_t j = 0; j < half; j++)
                for (uint64_t x = 0; x < 4; x++) {
                    auto at = [&](const std::vector<Fr> &t) { return t[2 * j].add(Fr::fromU64(x).mul(t[2 * j + 1].sub(t[2 * j]))); };
                    ev[x] = ev[x].add(at(inc).mul(at(wa)).mul(at(lt)));
                }
            for (int x = 0; x < 4; x++) EXPECT(got.round_polys[rd][x].eql(ev[x]));
            EXPECT(ev[0].add(ev[1]).eql(claim));
            Fr ch = tb.challengeScalar("val_eval_round");
            EXPECT(ch.eql(got.challenges[rd]));
            for (auto *t : {&inc, &wa, &lt})
                for (size_t j = 0; j < half; j++) (*t)[j] = (*t)[2 * j].add(ch.mul((*t)[2 * j + 1].sub((*t)[2 * j])));
            claim = cubicAtPoint(ev, ch);
        }
        EXPECT(got.final_claim.eql(inc[0].mul(wa[0]).mul(lt[0])) && (rounds == 0 || claim.eql(got.final_claim)));
    }
    Transcript te("Jolt");
    EXPECT(proveStage4({}, {}, 0, 2, 2, 0, te).skipped);
}

TEST(stages_5_and_6) {
    for (size_t n_steps : {size_t(1), size_t(2), size_t(200), size_t(4096)}) {
        std::vector<uint32_t> instr(n_steps);
        uint64_t z = 0x5eed + n_steps;
        for (auto &w : instr) { z = z * 6364136223846793005ULL + 1442695040888963407ULL; w = (uint32_t)(z >> 29); }
        Transcript ta("Jolt"), tb("Jolt");
        ta.appendBytes("stages 1-4"); tb.appendBytes("stages 1-4");
        std::vector<Fr> r_reg;
        auto got = proveStage5(instr, 4, ta, &r_reg);
        // host restatement
        std::vector<Fr> rr(5);
        for (auto &x : rr) x = tb.challengeScalar("r_register");
        for (int i = 0; i < 4; i++) (void)tb.challengeScalar("r_cycle_reg");
        for (int i = 0; i < 5; i++) EXPECT(rr[i].eql(r_reg[i]));
        size_t rounds = n_steps <= 1 ? 0 : log2Ceil(n_steps), n = size_t(1) << rounds;
        std::vector<Fr> ev(n, Fr::zero());
        Fr claim = Fr::zero();
        for (size_t j = 0; j < n_steps; j++) { ev[j] = computeRegEq(rr, (instr[j] >> 7) & 31); claim = claim.add(ev[j]); }
        EXPECT(got.initial_claim.eql(claim) && got.round_polys.size() == rounds);
        for (size_t rd = 0; rd < rounds; rd++) {
            size_t half = n >> (rd + 1);
            Fr s0 = Fr::zero(), s1 = Fr::zero();
            for (size_t j = 0; j < half; j++) { s0 = s0.add(ev[j]); s1 = s1.add(ev[j + half]); }
            EXPECT(got.round_polys[rd][0].eql(s0) && got.round_polys[rd][1].eql(s1.add(s1).sub(s0)));
            Fr ch = tb.challengeScalar("reg_eval_round");
            EXPECT(ch.eql(got.challenges[rd]));
            Fr omr = Fr::one().sub(ch);
            for (size_t j = 0; j < half; j++) ev[j] = omr.mul(ev[j]).add(ch.mul(ev[j + half]));
            claim = omr.mul(s0).add(ch.mul(s1));
            EXPECT(claim.eql(got.claims[rd]));
        }
        EXPECT(got.final_claim.eql(ev[0]) && (rounds == 0 || claim.eql(ev[0])));
        Fr bc;
        auto g6 = proveStage6(n_steps, ta, &bc);
        EXPECT(bc.eql(tb.challengeScalar("booleanity")) && g6.round_polys.size() == rounds && g6.final_claim.isZero() && g6.initial_claim.isZero());
        for (size_t rd = 0; rd < rounds; rd++) EXPECT(g6.round_polys[rd][0].isZero() && g6.round_polys[rd][1].isZero() && g6.challenges[rd].eql(tb.challengeScalar("bool_round")));
    }
    Transcript te("Jolt");
    EXPECT(proveStage5({}, 3, te).skipped && proveStage6(0, te).skipped);
}

// the reference's own vectors for the last fold sites: src/zkvm/spartan/prefix_suffix.zig "Phase1Prover basic" (P = [1,2,3,4], Q = [5,6,7,8]:
// evaluations (26, 44), bind 2 -> P = [3, 5]) and src/zkvm/lasso/prefix_suffix.zig "prefix polynomial bind" ([1,2,3,4] at 2 -> [5, 6])
TEST(remaining_fold_sites_on_the_references_vectors) {
    auto f = [](std::initializer_list<uint64_t> v) { std::vector<Fr> o; for (uint64_t x : v) o.push_back(Fr::fromU64(x)); return o; };
    Phase1Prover p;
    p.addPair(f({1, 2, 3, 4}), f({5, 6, 7, 8}));
    auto ev = p.computeRoundEvals();
    EXPECT(ev[0].eql(Fr::fromU64(26)) && ev[1].eql(Fr::fromU64(44)) && !p.shouldTransition());
    p.bind(Fr::fromU64(2));
    auto P = p.buffer(0, false);
    EXPECT(p.current_size == 2 && P.size() == 2 && P[0].eql(Fr::fromU64(3)) && P[1].eql(Fr::fromU64(5)) && p.shouldTransition());
    // three pairs: two in one pair-sum term, one as a plain product
    Phase1Prover p3;
    p3.addPair(f({1, 2, 3, 4}), f({5, 6, 7, 8}));
    p3.addPair(f({2, 0, 1, 3}), f({1, 1, 2, 2}));
    p3.addPair(f({7, 1, 0, 9}), f({3, 4, 5, 6}));
    auto e3 = p3.computeRoundEvals();  // g0 = 26 + (2 + 2) + (21 + 0), g1 = 44 + (0 + 6) + (4 + 54)
    EXPECT(e3[0].eql(Fr::fromU64(51)) && e3[1].eql(Fr::fromU64(108)));
    LassoPrefixPolynomial lp(f({1, 2, 3, 4}));
    auto b = lp.bind(Fr::fromU64(2));
    EXPECT(b.num_vars == 1 && b.evaluations[0].eql(Fr::fromU64(5)) && b.evaluations[1].eql(Fr::fromU64(6)));
    EXPECT(lp.evaluate({Fr::one(), Fr::zero()}).eql(Fr::fromU64(2)) && lp.evaluate({Fr::zero(), Fr::one()}).eql(Fr::fromU64(3)));
    // src/zkvm/lasso/expanding_table.zig tests: bind 3 -> [1 - 3, 3]; binds 2, 3, 5 -> sum 1, entry 0 = -8, entry 7 = 30; condense
    ExpandingTable et(4);
    for (uint64_t c : {2, 3, 5}) et.bind(Fr::fromU64(c));
    EXPECT(et.size() == 8 && et.sum().eql(Fr::one()) && et.get(0).eql(Fr::zero().sub(Fr::fromU64(8))) && et.get(7).eql(Fr::fromU64(30)) &&
           et.get(1).eql(Fr::fromU64(10)));
    ExpandingTable e2(4);
    e2.bind(Fr::fromU64(2));
    e2.bind(Fr::fromU64(3));
    auto cd = e2.condense(f({1, 2, 3, 4}), 1);
    EXPECT(cd.size() == 2 && cd[0].eql(e2.get(0).add(e2.get(1).mul(Fr::fromU64(2)))) &&
           cd[1].eql(e2.get(2).mul(Fr::fromU64(3)).add(e2.get(3).mul(Fr::fromU64(4)))));
    // Dory's evaluation vectors (src/poly/commitment/dory.zig:544-620): the basis at (3, 5) is [(1-3)(1-5), 3 (1-5), (1-3) 5, 15]
    auto lb = Dory::multilinearLagrangeBasis(f({3, 5}));
    EXPECT(lb.size() == 4 && lb[0].eql(Fr::fromU64(8)) && lb[1].eql(Fr::zero().sub(Fr::fromU64(12))) && lb[2].eql(Fr::zero().sub(Fr::fromU64(10))) &&
           lb[3].eql(Fr::fromU64(15)));
    auto lr = Dory::computeEvaluationVectors(f({3, 5, 7}), 2, 2);  // two column variables, one row variable: left = [1 - 7, 7, 0, 0]
    EXPECT(lr.second.size() == 4 && lr.second[3].eql(Fr::fromU64(15)) && lr.first[0].eql(Fr::zero().sub(Fr::fromU64(6))) && lr.first[1].eql(Fr::fromU64(7)) &&
           lr.first[2].isZero() && lr.first[3].isZero());
    SpartanOuterProver o(f({1, 2, 3, 4}));
    auto r0 = o.computeStandardRoundPoly();
    EXPECT(r0[0].eql(Fr::fromU64(4)) && r0[1].eql(Fr::fromU64(6)) && r0[2].eql(Fr::fromU64(8)));
    o.bindChallenge(Fr::fromU64(3));  // [(1 - 3) * 1 + 3 * 2, (1 - 3) * 3 + 3 * 4] = [4, 6]
    auto r1 = o.computeStandardRoundPoly();
    EXPECT(r1[0].eql(Fr::fromU64(4)) && r1[1].eql(Fr::fromU64(6)));
    o.bindChallenge(Fr::fromU64(2));  // 4 * (1 - 2) + 6 * 2 = 8
    auto r2 = o.computeStandardRoundPoly();
    EXPECT(r2[0].eql(Fr::fromU64(8)) && r2[1].isZero() && r2[2].isZero());
}

// `test_host_mirror rwc <file>`: runs zolt::RamReadWriteCheckingProver on the instance the file describes (written by
// tests/test_gpu_cpp_host.py: the reference's captured run and random traces) and prints every round polynomial, claim, entry count and
// the opening claims as hex limbs; the Python test compares the lines with the oracle's.
static Fr read_fr(std::FILE *f) {
    Fr x;
    for (int i = 0; i < 4; i++) {
        unsigned long long v = 0;
        if (std::fscanf(f, "%llx", &v) != 1) throw std::runtime_error("rwc input: field element expected");
        x.limbs[i] = v;
    }
    return x;
}
static void print_fr(const Fr &x) { std::printf(" %016llx %016llx %016llx %016llx", (unsigned long long)x.limbs[0], (unsigned long long)x.limbs[1], (unsigned long long)x.limbs[2], (unsigned long long)x.limbs[3]); }
static int rwc_main(const char *path) {
    std::FILE *f = std::fopen(path, "r");
    if (!f) { std::printf("cannot open %s\n", path); return 2; }
    unsigned long long log_k, log_t, p1, start, n;
    if (std::fscanf(f, "%llu %llu %llu %llu", &log_k, &log_t, &p1, &start) != 4) return 2;
    Fr gamma = read_fr(f), claim = read_fr(f);
    std::vector<Fr> r_cycle;
    for (size_t i = 0; i < log_t; i++) r_cycle.push_back(read_fr(f));
    std::vector<std::pair<uint64_t, uint64_t>> init;
    if (std::fscanf(f, "%llu", &n) != 1) return 2;
    for (size_t i = 0; i < n; i++) {
        unsigned long long a, v;
        if (std::fscanf(f, "%llu %llu", &a, &v) != 2) return 2;
        init.emplace_back(a, v);
    }
    std::vector<MemoryAccess> acc;
    if (std::fscanf(f, "%llu", &n) != 1) return 2;
    for (size_t i = 0; i < n; i++) {
        unsigned long long ts, a, w, v;
        if (std::fscanf(f, "%llu %llu %llu %llu", &ts, &a, &w, &v) != 4) return 2;
        acc.push_back(MemoryAccess{ts, a, w != 0, v});
    }
    std::vector<Fr> ch;
    for (size_t i = 0; i < log_k + log_t; i++) ch.push_back(read_fr(f));
    std::fclose(f);
    RamReadWriteCheckingProver p(acc, gamma, r_cycle, log_k, log_t, p1, start, claim, init);
    for (size_t rd = 0; rd < log_k + log_t; rd++) {
        auto ev = p.computeRoundPolynomialCubic();
        std::printf("E");
        for (const Fr &x : ev) print_fr(x);
        std::printf("\n");
        p.updateClaim(ev, ch[rd]);
        p.bindChallenge(ch[rd]);
        std::printf("C");
        print_fr(p.current_claim);
        std::printf(" %zu\n", p.numEntries());
    }
    auto oc = p.getOpeningClaims(ch);
    std::printf("O");
    print_fr(oc.ra_claim);
    print_fr(oc.val_claim);
    print_fr(oc.inc_claim);
    std::printf("\n%d\n", p.isComplete() ? 1 : 0);
    return 0;
}

// `test_host_mirror stage4 <file>`: zolt::Stage4GruenProver on the instance the file describes (written by tests/test_gpu_cpp_host.py):
// log_t, phase-1 rounds, gamma, the claim, r_cycle, the steps, the challenges -> every round's four evaluations and the final values.
static int stage4_main(const char *path) {
    std::FILE *f = std::fopen(path, "r");
    if (!f) { std::printf("cannot open %s\n", path); return 2; }
    unsigned long long log_t, p1, n;
    if (std::fscanf(f, "%llu %llu", &log_t, &p1) != 2) return 2;
    Fr gamma = read_fr(f), claim = read_fr(f);
    std::vector<Fr> r_cycle;
    for (size_t i = 0; i < log_t; i++) r_cycle.push_back(read_fr(f));
    if (std::fscanf(f, "%llu", &n) != 1) return 2;
    std::vector<TraceStep> steps;
    for (size_t i = 0; i < n; i++) {
        unsigned long long w, v, z;
        if (std::fscanf(f, "%llu %llu %llu", &w, &v, &z) != 3) return 2;
        steps.push_back(TraceStep{(uint32_t)w, v, z != 0});
    }
    std::vector<Fr> ch;
    for (size_t i = 0; i < 7 + log_t; i++) ch.push_back(read_fr(f));
    std::fclose(f);
    if (p1 == 0) {  // phase-1 length 0 in the file: the ORIGINAL Stage4Prover (cycle variables first, four evaluations from the tables)
        Stage4Prover q(steps, gamma, r_cycle);
        for (size_t rd = 0; rd < q.num_rounds; rd++) {
            auto ev = q.computeRoundEvals(rd, claim);
            auto co = q.computeRoundPolynomial(rd, claim);
            std::printf("E");
            for (const Fr &x : ev) print_fr(x);
            std::printf("\nP");
            for (const Fr &x : co) print_fr(x);
            std::printf("\n");
            claim = cubicAtPoint(ev, ch[rd]);
            q.bindChallenge(rd, ch[rd]);
        }
        auto fc = q.getFinalClaims();
        std::printf("O");
        for (const Fr &x : {fc.val_claim, fc.rs1_ra_claim, fc.rs2_ra_claim, fc.rd_wa_claim, fc.inc_claim, claim}) print_fr(x);
        std::printf("\n");
        return 0;
    }
    Stage4GruenProver p(steps, gamma, r_cycle, p1, 7);
    for (size_t rd = 0; rd < p.num_rounds; rd++) {
        auto ev = p.computeRoundEvals(rd, claim);
        std::printf("E");
        for (const Fr &x : ev) print_fr(x);
        std::printf("\n");
        claim = cubicAtPoint(ev, ch[rd]);
        p.bindChallenge(rd, ch[rd]);
    }
    auto fc = p.getFinalClaims();
    auto chk = p.finalCheck();
    std::printf("O");
    for (const Fr &x : {fc.val_claim, fc.rs1_ra_claim, fc.rs2_ra_claim, fc.rd_wa_claim, fc.inc_claim, chk[0], chk[1], chk[2], claim}) print_fr(x);
    std::printf("\n");
    return 0;
}

// `test_host_mirror outer <file>`: zolt::StreamingOuterProver's remaining rounds on the instance the file describes (cycles, scaling, r0,
// claim, tau, the 43 inputs of every cycle, the challenges) -> per round (t'(0), t'(inf)) and the four evaluations, then the final values.
static int outer_main(const char *path) {
    std::FILE *f = std::fopen(path, "r");
    if (!f) { std::printf("cannot open %s\n", path); return 2; }
    unsigned long long n, nv;
    if (std::fscanf(f, "%llu %llu", &n, &nv) != 2) return 2;
    Fr scale = read_fr(f), r0 = read_fr(f), claim = read_fr(f);
    std::vector<Fr> tau;
    for (size_t i = 0; i < nv + 2; i++) tau.push_back(read_fr(f));
    std::vector<StreamingOuterProver::CycleInputs> w(n);
    for (size_t i = 0; i < n; i++)
        for (size_t k = 0; k < r1cs::NUM_INPUTS; k++) w[i][k] = read_fr(f);
    std::vector<Fr> ch;
    for (size_t i = 0; i < nv + 1; i++) ch.push_back(read_fr(f));
    std::fclose(f);
    StreamingOuterProver p(w, tau, &scale);
    {  // the UniSkip first round: t1 at the nine targets and the 28 coefficients of s1
        auto s1 = p.computeFirstRoundPoly();
        std::printf("X");
        for (const Fr &x : p.last_extended_evals) print_fr(x);
        std::printf("\nS");
        for (const Fr &x : s1) print_fr(x);
        std::printf("\n");
    }
    p.bindFirstRoundChallenge(r0, claim);
    for (size_t rd = 0; rd < p.numRounds(); rd++) {
        auto ev = p.computeRemainingRoundPoly();
        std::printf("T");
        print_fr(p.last_t_zero);
        print_fr(p.last_t_infinity);
        std::printf("\nE");
        for (const Fr &x : ev) print_fr(x);
        std::printf("\n");
        p.updateClaim(ev, ch[rd]);
        p.bindRemainingRoundChallenge(ch[rd]);
    }
    auto fin = p.finalAzBz();
    std::printf("O");
    for (const Fr &x : {fin[0], fin[1], p.getFinalEval(), p.split_eq.current_scalar}) print_fr(x);
    std::printf("\n");
    return 0;
}

// `test_host_mirror witness <file>`: zolt::CycleColumns::fromTrace + CycleWitnessMatrix on the trace the file describes (n, then per step:
// instruction pc unexpanded_pc rs1 rs2 rd has_memory memory is_compressed is_noop) -> per cycle "W" and its 43 inputs as the device widened
// them, then "B" bytes per cycle; StreamingOuterProver's first round from the shared matrix and from ready rows must agree ("S" lines)
static int witness_main(const char *path) {
    std::FILE *f = std::fopen(path, "r");
    if (!f) { std::printf("cannot open %s\n", path); return 2; }
    unsigned long long n;
    if (std::fscanf(f, "%llu", &n) != 1) return 2;
    std::vector<R1CSTraceStep> steps(n);
    for (auto &st : steps) {
        unsigned long long w, pc, upc, a, b, rd, hm, mem, comp, noop;
        if (std::fscanf(f, "%llx %llx %llx %llx %llx %llx %llu %llx %llu %llu", &w, &pc, &upc, &a, &b, &rd, &hm, &mem, &comp, &noop) != 10) return 2;
        st.instruction = (uint32_t)w; st.pc = pc; st.unexpanded_pc = upc; st.rs1_value = a; st.rs2_value = b; st.rd_value = rd;
        st.has_memory_value = hm != 0; st.memory_value = mem; st.is_compressed = comp != 0; st.is_noop = noop != 0;
    }
    std::fclose(f);
    CycleColumns cols = CycleColumns::fromTrace(steps);
    auto m = CycleWitnessMatrix::fromColumns(cols);
    std::vector<Fr> rows = m->toHost();
    for (size_t i = 0; i < n; i++) {
        std::printf("W");
        for (size_t k = 0; k < CycleColumns::NUM_INPUTS; k++) print_fr(rows[i * CycleColumns::NUM_INPUTS + k]);
        std::printf("\n");
    }
    std::printf("B %zu\n", cols.bytesPerCycle());
    {  // the streamed construction (slices of cycles decoded while the previous slice uploads) must give the same matrix
        setenv("ZOLT_WITNESS_SLICES", "3", 1);
        auto m2 = CycleWitnessMatrix::fromTrace(steps);
        unsetenv("ZOLT_WITNESS_SLICES");
        std::vector<Fr> rows2 = m2->toHost();
        std::printf("T %d\n", rows2.size() == rows.size() && std::memcmp(rows2.data(), rows.data(), rows.size() * sizeof(Fr)) == 0);
    }
    size_t nv = 0;
    while ((size_t(1) << nv) < n) nv++;
    std::vector<Fr> tau(nv + 2);
    for (size_t i = 0; i < tau.size(); i++) tau[i] = Fr::fromU64(0x9E3779B97F4A7C15ULL * (i + 1));
    std::vector<StreamingOuterProver::CycleInputs> w(n);
    for (size_t i = 0; i < n; i++)
        for (size_t k = 0; k < r1cs::NUM_INPUTS; k++) w[i][k] = rows[i * r1cs::NUM_INPUTS + k];
    StreamingOuterProver from_matrix(m, tau), from_rows(w, tau);
    for (auto *p : {&from_matrix, &from_rows}) {
        auto s1 = p->computeFirstRoundPoly();
        std::printf("S");
        for (const Fr &x : s1) print_fr(x);
        std::printf("\n");
    }
    return 0;
}

// `test_host_mirror stage3 <file>`: zolt::Stage3Prover on the instance the file describes (written by tests/test_gpu_cpp_host.py): n, the
// five shift gamma powers, the two other gammas, three input claims, three batching coefficients, r_outer, r_product, the 2^n x 43 witness
// matrix, n challenges -> per round ShiftSumcheck's three evaluations (S), the compressed combined polynomial (P), the claims after the
// challenge (C: shift, instruction input, registers, combined); then the final claims of ShiftSumcheck (F) and RegistersClaimReduction (G)
static int stage3_main(const char *path) {
    std::FILE *f = std::fopen(path, "r");
    if (!f) { std::printf("cannot open %s\n", path); return 2; }
    unsigned long long n;
    if (std::fscanf(f, "%llu", &n) != 1) return 2;
    std::vector<Fr> sg;
    for (int i = 0; i < 5; i++) sg.push_back(read_fr(f));
    Fr ig = read_fr(f), rg = read_fr(f);
    std::array<Fr, 3> claims, coeffs;
    for (auto &x : claims) x = read_fr(f);
    for (auto &x : coeffs) x = read_fr(f);
    std::vector<Fr> ro, rp, w, ch;
    for (size_t i = 0; i < n; i++) ro.push_back(read_fr(f));
    for (size_t i = 0; i < n; i++) rp.push_back(read_fr(f));
    const size_t N = size_t(1) << n;
    w.reserve(N * 43);
    for (size_t i = 0; i < N * 43; i++) w.push_back(read_fr(f));
    for (size_t i = 0; i < n; i++) ch.push_back(read_fr(f));
    std::fclose(f);
    DeviceMem d_rows(N * 43 * 32);
    check(zg_memcpy_h2d(d_rows.p, w.data(), N * 43 * 32), "zg_memcpy_h2d");
    Stage3Prover p(d_rows.u64(), ro, rp, sg, ig, rg, claims, coeffs);
    for (size_t rd = 0; rd < n; rd++) {
        auto comp = p.computeRoundPolynomial();
        std::printf("S");
        for (const Fr &x : p.roundEvals(0)) print_fr(x);
        std::printf("\nP");
        for (const Fr &x : comp) print_fr(x);
        p.bindChallenge(ch[rd]);
        std::printf("\nC");
        for (const Fr &x : p.claims) print_fr(x);
        print_fr(p.combined_claim);
        std::printf("\n");
    }
    std::printf("F");
    for (const Fr &x : p.shift.finalClaims()) print_fr(x);
    std::printf("\nG");
    for (const Fr &x : p.reg.finalClaims()) print_fr(x);
    std::printf("\n");
    return 0;
}

// `test_host_mirror wire <kind> <in> <out>`: the wire / disk formats of zolt::wire on a file written by tests/test_gpu_cpp_host.py.
//   raw    : raw SRS file  -> prints n, then re-serialises it (out must equal in); the Montgomery limbs of every point as hex lines
//   ptau   : ptau container -> power, ceremony power, counts; limbs of the three G1 sections
//   proof  : ZOLT proof     -> the eleven 64-byte commitments as hex, and the re-serialised 744-byte header to <out>
static std::vector<uint8_t> read_file(const char *path) {
    std::FILE *f = std::fopen(path, "rb");
    if (!f) throw std::runtime_error("cannot open input");
    std::vector<uint8_t> d;
    uint8_t buf[65536];
    size_t k;
    while ((k = std::fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + k);
    std::fclose(f);
    return d;
}
static void write_file(const char *path, const std::vector<uint8_t> &d) {
    std::FILE *f = std::fopen(path, "wb");
    if (!f) throw std::runtime_error("cannot open output");
    std::fwrite(d.data(), 1, d.size(), f);
    std::fclose(f);
}
static void print_points(const char *tag, const wire::G1Points &p) {
    for (size_t i = 0; i < p.size(); i++) {
        std::printf("%s %d", tag, (int)p.inf[i]);
        for (size_t l = 0; l < 8; l++) std::printf(" %016llx", (unsigned long long)p.xy[8 * i + l]);
        std::printf("\n");
    }
}
static int wire_main(const char *kind, const char *in, const char *out) {
    auto data = read_file(in);
    try {
        if (!std::strcmp(kind, "raw")) {
            std::vector<uint8_t> trailer;
            auto pts = wire::srsG1FromRaw(data, &trailer);
            std::printf("N %zu\n", pts.size());
            print_points("P", pts);
            write_file(out, wire::srsG1ToRaw(pts, trailer));
        } else if (!std::strcmp(kind, "ptau")) {
            auto pt = wire::srsG1FromPtau(data);
            std::printf("H %u %u %zu %zu %zu %zu %zu\n", pt.power, pt.ceremony_power, pt.powers_of_tau_g1.size(), pt.has_alpha ? pt.alpha_tau_g1.size() : 0,
                        pt.has_beta ? pt.beta_tau_g1.size() : 0, pt.tau_g2_raw.size(), pt.beta_g2_raw.size());
            print_points("T", pt.powers_of_tau_g1);
            print_points("A", pt.alpha_tau_g1);
            print_points("B", pt.beta_tau_g1);
        } else {
            auto cs = wire::parseZoltProofCommitments(data);
            for (size_t i = 0; i < 11; i++) {
                std::printf("C %s ", wire::PROOF_COMMITMENT_NAMES[i]);
                for (uint8_t b : cs[i]) std::printf("%02x", b);
                std::printf("\n");
            }
            write_file(out, wire::serializeZoltProofHeader(cs));
            // a commitment produced here: the generator, as PolyCommitment.toBytes writes it
            auto g = wire::commitmentToBytes(AffinePoint::generator());
            std::printf("G ");
            for (uint8_t b : g) std::printf("%02x", b);
            std::printf("\n");
        }
    } catch (const wire::SRSError &e) {
        std::printf("SRSError %s\n", e.what());
        return 0;
    }
    return 0;
}

// `test_host_mirror dory <file>`: zolt::Dory on the instance the file describes (written by tests/test_gpu_cpp_host.py): n bases (packed
// xy limbs + infinity flags), num_columns, the evaluations, nu, sigma, left_vec -> the row commitments (R) and the vector-matrix product (V)
static int dory_main(const char *path) {
    std::FILE *f = std::fopen(path, "r");
    if (!f) { std::printf("cannot open %s\n", path); return 2; }
    unsigned long long n, cols, ne, nu, sigma, nl;
    if (std::fscanf(f, "%llu", &n) != 1) return 2;
    std::vector<AffinePoint> bases(n);
    for (auto &b : bases) {
        for (int i = 0; i < 4; i++) { unsigned long long v; if (std::fscanf(f, "%llx", &v) != 1) return 2; b.x.limbs[i] = v; }
        for (int i = 0; i < 4; i++) { unsigned long long v; if (std::fscanf(f, "%llx", &v) != 1) return 2; b.y.limbs[i] = v; }
        unsigned long long inf;
        if (std::fscanf(f, "%llu", &inf) != 1) return 2;
        b.infinity = inf != 0;
    }
    if (std::fscanf(f, "%llu %llu", &cols, &ne) != 2) return 2;
    std::vector<Fr> evals;
    for (size_t i = 0; i < ne; i++) evals.push_back(read_fr(f));
    if (std::fscanf(f, "%llu %llu %llu", &nu, &sigma, &nl) != 3) return 2;
    std::vector<Fr> left;
    for (size_t i = 0; i < nl; i++) left.push_back(read_fr(f));
    std::fclose(f);
    DeviceBases g1(bases);
    for (const AffinePoint &p : Dory::computeRowCommitments(g1, evals, cols)) {
        std::printf("R %d", p.infinity ? 1 : 0);
        for (int i = 0; i < 4; i++) std::printf(" %016llx", (unsigned long long)p.x.limbs[i]);
        for (int i = 0; i < 4; i++) std::printf(" %016llx", (unsigned long long)p.y.limbs[i]);
        std::printf("\n");
    }
    std::printf("V");
    for (const Fr &x : Dory::computeVectorMatrixProduct(evals, left, (unsigned)nu, (unsigned)sigma)) print_fr(x);
    std::printf("\n");
    return 0;
}

// BASELINE config 5 from compiled host code: the stage records of the reference's captured proof file (src/zkvm/serialization.zig:186-343)
// produced by the C++ mirrors — Keccak transcript, LassoProver, proveStage4 / 5 / 6 on device sessions, Stage 1's Az(r) / Bz(r) by
// thirteen LowToHigh folds — from inputs the Python side regenerates from the ELF (tests/util.py). Input file: five absorbed
// commitments (64 bytes each, hex), the padded trace's instruction words, the lookup indices, Az and Bz (row = cycle * 19 + constraint).
// Output: per stage "S k", then "P" lines (round polynomials), "H" (challenges), "C" (claims), raw Montgomery limbs.
static int proof_main(const char *path) {
    std::FILE *f = std::fopen(path, "r");
    if (!f) { std::printf("cannot open %s\n", path); return 2; }
    Transcript T("Jolt");
    for (int k = 0; k < 5; k++) {
        char hex[129];
        if (std::fscanf(f, "%128s", hex) != 1) return 2;
        uint8_t b[64];
        for (int i = 0; i < 64; i++) { unsigned v; std::sscanf(hex + 2 * i, "%2x", &v); b[i] = (uint8_t)v; }
        T.appendBytes(b, 64);
    }
    unsigned long long log_t, log_k, n_instr, n_look, n_rows;
    if (std::fscanf(f, "%llu %llu %llu", &log_t, &log_k, &n_instr) != 3) return 2;
    std::vector<uint32_t> instr(n_instr);
    for (auto &w : instr) { unsigned long long v; if (std::fscanf(f, "%llu", &v) != 1) return 2; w = (uint32_t)v; }
    if (std::fscanf(f, "%llu", &n_look) != 1) return 2;
    std::vector<unsigned __int128> idx(n_look);
    for (auto &v : idx) { unsigned long long lo, hi; if (std::fscanf(f, "%llu %llu", &lo, &hi) != 2) return 2; v = ((unsigned __int128)hi << 64) | lo; }
    if (std::fscanf(f, "%llu", &n_rows) != 1) return 2;
    std::vector<Fr> az(n_rows), bz(n_rows);
    for (auto &x : az) x = read_fr(f);
    for (auto &x : bz) x = read_fr(f);
    std::fclose(f);
    auto line = [](const char *tag, const std::vector<Fr> &v) { std::printf("%s", tag); for (const Fr &x : v) print_fr(x); std::printf("\n"); };
    // ---- stage 1: tau, thirteen zero round polynomials (absorbed), Az(r) / Bz(r)
    size_t rounds1 = 0;
    while ((size_t(1) << rounds1) < n_rows) rounds1++;
    for (size_t i = 0; i < rounds1; i++) (void)T.challengeScalar("spartan_tau");
    std::vector<Fr> ch;
    std::printf("S 1\n");
    for (size_t k = 0; k < rounds1; k++) {
        T.appendScalar("round_poly_0", Fr::zero());
        T.appendScalar("round_poly_1", Fr::zero());
        T.appendScalar("round_poly_2", Fr::zero());
        ch.push_back(T.challengeScalar("spartan_round"));
        line("P", {Fr::zero(), Fr::zero(), Fr::zero()});
    }
    for (const Fr &c : ch) {
        check(zg_fr_bind_low(reinterpret_cast<uint64_t *>(az.data()), az.size(), c.limbs), "zg_fr_bind_low");
        check(zg_fr_bind_low(reinterpret_cast<uint64_t *>(bz.data()), bz.size(), c.limbs), "zg_fr_bind_low");
        az.resize(az.size() / 2);
        bz.resize(bz.size() / 2);
    }
    line("H", ch);
    line("C", {Fr::zero(), Fr::zero(), az[0], bz[0], Fr::zero()});
    // ---- stage 2: RAF over an empty memory trace
    for (size_t i = 0; i < log_t; i++) (void)T.challengeScalar("r_cycle");
    ch.clear();
    std::printf("S 2\n");
    for (size_t k = 0; k < log_k; k++) { ch.push_back(T.challengeScalar("raf_round")); line("P", {Fr::zero(), Fr::zero()}); }
    line("H", ch);
    line("C", {Fr::zero(), Fr::zero()});
    // ---- stage 3: Lasso
    (void)T.challengeScalar("lasso_gamma");
    std::vector<Fr> r_red;
    for (size_t i = 0; i < log_t; i++) r_red.push_back(T.challengeScalar("r_reduction"));
    LassoProver lp(idx, log_t, 16, r_red);
    ch.clear();
    std::printf("S 3\n");
    Fr init3 = Fr::zero();
    for (size_t k = 0; k < 16 + log_t; k++) {
        UniPoly rp = lp.computeRoundPolynomial();
        if (k == 0) init3 = lp.current_claim;
        line("P", rp.coeffs);
        Fr c = T.challengeScalar("lasso_round");
        ch.push_back(c);
        lp.receiveChallenge(c);
    }
    line("H", ch);
    line("C", {init3, lp.getFinalEval()});
    // ---- stages 4, 5, 6
    Stage4Result r4 = proveStage4({}, {}, size_t(1) << log_t, log_k, log_t, 0x80000000ull, T);
    std::printf("S 4\n");
    for (const auto &rp : r4.round_polys) line("P", {rp[0], rp[1], rp[2], rp[3]});
    line("H", r4.challenges);
    line("C", {r4.initial_claim, r4.final_claim});
    StageRoundsResult r5 = proveStage5(instr, log_t, T), r6 = proveStage6(size_t(1) << log_t, T);
    int k = 5;
    for (const StageRoundsResult *r : {&r5, &r6}) {
        std::printf("S %d\n", k++);
        for (const auto &rp : r->round_polys) line("P", {rp[0], rp[1]});
        line("H", r->challenges);
        line("C", {r->initial_claim, r->final_claim});
    }
    return 0;
}

int main(int argc, char **argv) {
    if (zg_init(0) != ZG_OK) { std::printf("zg_init failed: %s\n", zg_last_error()); return 2; }
    if (argc >= 3 && !std::strcmp(argv[1], "proof")) {
        int rc;
        try { check(zg_init(0), "zg_init"); rc = proof_main(argv[2]); } catch (const std::exception &e) { std::printf("EXCEPTION: %s\n", e.what()); rc = 3; }
        return rc;
    }
    if (argc >= 3 && !std::strcmp(argv[1], "outer")) {
        int rc;
        try { rc = outer_main(argv[2]); } catch (const std::exception &e) { std::printf("EXCEPTION: %s\n", e.what()); rc = 3; }
        zg_shutdown();
        return rc;
    }
    if (argc >= 3 && !std::strcmp(argv[1], "witness")) {
        int rc;
        try { rc = witness_main(argv[2]); } catch (const std::exception &e) { std::printf("EXCEPTION: %s\n", e.what()); rc = 3; }
        zg_shutdown();
        return rc;
    }
    if (argc >= 5 && !std::strcmp(argv[1], "wire")) {
        int rc;
        try { rc = wire_main(argv[2], argv[3], argv[4]); } catch (const std::exception &e) { std::printf("EXCEPTION: %s\n", e.what()); rc = 3; }
        zg_shutdown();
        return rc;
    }
    if (argc >= 3 && !std::strcmp(argv[1], "dory")) {
        int rc;
        try { rc = dory_main(argv[2]); } catch (const std::exception &e) { std::printf("EXCEPTION: %s\n", e.what()); rc = 3; }
        zg_shutdown();
        return rc;
    }
    if (argc >= 3 && !std::strcmp(argv[1], "stage3")) {
        int rc;
        try { rc = stage3_main(argv[2]); } catch (const std::exception &e) { std::printf("EXCEPTION: %s\n", e.what()); rc = 3; }
        zg_shutdown();
        return rc;
    }
    if (argc >= 3 && !std::strcmp(argv[1], "stage4")) {
        int rc;
        try { rc = stage4_main(argv[2]); } catch (const std::exception &e) { std::printf("EXCEPTION: %s\n", e.what()); rc = 3; }
        zg_shutdown();
        return rc;
    }
    if (argc >= 3 && !std::strcmp(argv[1], "rwc")) {
        int rc;
        try { rc = rwc_main(argv[2]); } catch (const std::exception &e) { std::printf("EXCEPTION: %s\n", e.what()); rc = 3; }
        zg_shutdown();
        return rc;
    }
    for (auto &t : tests()) {
        int before = g_failed;
        try { t.f(); } catch (const std::exception &e) { std::printf("  EXCEPTION in %s: %s\n", t.n, e.what()); g_failed++; }
        std::printf("[%s] %s\n", g_failed == before ? " OK " : "FAIL", t.n);
        g_run++;
    }
    std::printf("%d tests, %d failures\n", g_run, g_failed);
    zg_shutdown();
    return g_failed ? 1 : 0;
}
