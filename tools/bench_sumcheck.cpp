// bench_sumcheck.cpp — BASELINE config 3 with a compiled host loop: 20-variable sumcheck
// (runSumcheck, src/subprotocols/mod.zig:302-354) over a table resident in HBM, toy verifier on the host,
// plus the eq-table build and Spartan combine that precede it (src/zkvm/spartan/mod.zig:182-206).
// Prints one JSON object. Usage: bench_sumcheck [v=20] [reps=20]
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "../zolt_amd/host/zolt_host.hpp"
#include "../include/zolt_gpu_internal.h"

using namespace zolt;
using clk = std::chrono::steady_clock;

static uint64_t sm_state = 0x53554D43;
static uint64_t splitmix() {
    uint64_t z = (sm_state += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

int main(int argc, char **argv) {
    int v = argc > 1 ? atoi(argv[1]) : 20, reps = argc > 2 ? atoi(argv[2]) : 20;
    size_t n = size_t(1) << v;
    check(zg_init(0), "zg_init");
    // random table: raw 256-bit words -> Montgomery (reduces mod r), on the GPU
    std::vector<uint64_t> raw(4 * n), tab(4 * n);
    for (auto &w : raw) w = splitmix();
    check(zg_field_op(ZG_FIELD_FR, ZG_OP_TO_MONT, raw.data(), nullptr, tab.data(), n), "to_mont");
    void *d_tab = nullptr, *d_eq = nullptr, *d_f = nullptr;
    check(zg_dev_alloc(n * 32, &d_tab), "alloc");
    check(zg_dev_alloc(n * 32, &d_eq), "alloc");
    check(zg_dev_alloc(n * 32, &d_f), "alloc");
    check(zg_memcpy_h2d(d_tab, tab.data(), n * 32), "h2d");
    std::vector<Fr> r(v);
    for (auto &x : r) x = Fr::fromU64(splitmix());

    double t_sc = 0, t_eq = 0, t_comb = 0, t_dev = 0;
    bool ok = true;
    Fr last_fin = Fr::zero();
    for (int rep = -2; rep < reps; rep++) {  // two warm-up passes
        auto t0 = clk::now();
        check(zg_fr_eq_table_dev(reinterpret_cast<const uint64_t *>(r.data()), v, nullptr, (uint64_t *)d_eq, nullptr), "eq");
        check(zg_sync(), "sync");
        auto t1 = clk::now();
        check(zg_fr_spartan_combine_dev((uint64_t *)d_eq, (uint64_t *)d_tab, (uint64_t *)d_tab, (uint64_t *)d_tab, n, (uint64_t *)d_f, nullptr), "comb");
        check(zg_sync(), "sync");
        auto t2 = clk::now();
        zg_sc_t s = nullptr;
        check(zg_sumcheck_open_dev((uint64_t *)d_f, n, ZG_SC_HIGH_HALF, nullptr, &s), "open");
        Fr g0, g1;
        check(zg_sumcheck_round_sums(s, g0.limbs, g1.limbs), "sums");
        Sumcheck::Verifier ver(g0.add(g1));
        for (int k = 0; k < v; k++) {
            check(zg_sumcheck_round_sums(s, g0.limbs, g1.limbs), "sums");
            Sumcheck::Round rd;
            rd.poly.coeffs = {g0, g1.sub(g0)};
            Fr ch = ver.verifyRound(rd);
            check(zg_sumcheck_bind(s, ch.limbs), "bind");
        }
        Fr fin;
        check(zg_sumcheck_final(s, fin.limbs), "final");
        ok = ok && fin.eql(ver.claim);
        last_fin = fin;
        zg_sumcheck_close(s);
        auto t3 = clk::now();
        if (rep >= 0) {
            t_eq += std::chrono::duration<double>(t1 - t0).count();
            t_comb += std::chrono::duration<double>(t2 - t1).count();
            t_sc += std::chrono::duration<double>(t3 - t2).count();
        }
    }
    // the whole Spartan instance with the fused opening (zg_sumcheck_open_spartan_dev: no eq table in HBM, no table copy, round 0's
    // sums from the same pass), verifier on the host
    double t_fused = 0;
    for (int rep = -2; rep < reps; rep++) {
        auto t0 = clk::now();
        zg_sc_t s = nullptr;
        check(zg_sumcheck_open_spartan_dev(reinterpret_cast<const uint64_t *>(r.data()), v, nullptr, (uint64_t *)d_tab, (uint64_t *)d_tab,
                                           (uint64_t *)d_tab, ZG_SC_HIGH_HALF, nullptr, &s), "open_spartan");
        Fr g0, g1;
        check(zg_sumcheck_round_sums(s, g0.limbs, g1.limbs), "sums");
        Sumcheck::Verifier ver(g0.add(g1));
        for (int k = 0; k < v; k++) {
            check(zg_sumcheck_round_sums(s, g0.limbs, g1.limbs), "sums");
            Sumcheck::Round rd;
            rd.poly.coeffs = {g0, g1.sub(g0)};
            Fr ch = ver.verifyRound(rd);
            check(zg_sumcheck_bind(s, ch.limbs), "bind");
        }
        Fr fin;
        check(zg_sumcheck_final(s, fin.limbs), "final");
        ok = ok && fin.eql(ver.claim) && fin.eql(last_fin);
        zg_sumcheck_close(s);
        if (rep >= 0) t_fused += std::chrono::duration<double>(clk::now() - t0).count();
    }
    // the same protocol with the toy verifier on the device too (zg_run_sumcheck_dev): no PCIe crossing per round
    {
        std::vector<uint64_t> rounds(8 * v + 1), chal(4 * v + 1);
        Fr claim, fin;
        uint8_t res = 0;
        for (int rep = -2; rep < reps; rep++) {
            auto t0 = clk::now();
            check(zg_run_sumcheck_dev((uint64_t *)d_f, n, nullptr, claim.limbs, rounds.data(), chal.data(), fin.limbs, &res), "run");
            auto t1 = clk::now();
            if (rep >= 0) t_dev += std::chrono::duration<double>(t1 - t0).count();
            ok = ok && res == 1 && fin.eql(last_fin);
        }
    }
    // Stage 1 as the reference's prover runs it (src/zkvm/prover.zig:397-432): LowToHigh table in a LOW_PAIR session, per round
    // [p0, p1, 2p1 - p0] absorbed into the Keccak transcript (src/transcripts/mod.zig), challengeScalar("spartan_round"), fold —
    // and Stage 2's RAF cubic rounds (raf_checking.zig:335-445: s(0), s(2) in one pass, transcript challenge, fold)
    double t_s1 = 0, t_raf = 0;
    {
        for (int rep = -2; rep < reps; rep++) {
            auto t0 = clk::now();
            zg_sc_t s = nullptr;
            check(zg_sumcheck_open_dev((uint64_t *)d_f, n, ZG_SC_LOW_PAIR, nullptr, &s), "open");
            Transcript tr("Jolt");
            for (int k = 0; k < v; k++) {
                Fr p0, p1;
                check(zg_sumcheck_round_sums(s, p0.limbs, p1.limbs), "sums");
                Fr p2 = p1.add(p1).sub(p0);
                tr.appendScalar("round_poly_0", p0);
                tr.appendScalar("round_poly_1", p1);
                tr.appendScalar("round_poly_2", p2);
                Fr ch = tr.challengeScalar("spartan_round");
                check(zg_sumcheck_bind(s, ch.limbs), "bind");
            }
            Fr fin;
            check(zg_sumcheck_final(s, fin.limbs), "final");
            zg_sumcheck_close(s);
            auto t1 = clk::now();
            if (rep >= 0) t_s1 += std::chrono::duration<double>(t1 - t0).count();
            // RAF: same table as RaPolynomial, start address 0x7fff8000
            check(zg_sumcheck_open_dev((uint64_t *)d_f, n, ZG_SC_LOW_PAIR, nullptr, &s), "open");
            Fr base = Fr::fromU64(0x7fff8000ULL), claim = Fr::fromU64(12345);
            uint64_t power = 8;
            for (int k = 0; k < v; k++) {
                Fr s0, s2;
                check(zg_sumcheck_raf_round(s, base.limbs, power, s0.limbs, s2.limbs), "raf_round");
                Fr s1 = claim.sub(s0);
                Fr ch = tr.challengeScalar("raf_round");
                // the claim update is a handful of host products (Lagrange through s(0..3)); here: keep the chain data-dependent
                claim = s0.add(ch.mul(s1.sub(s0))).add(s2.mul(ch));
                check(zg_sumcheck_bind(s, ch.limbs), "bind");
                base = base.add(ch.mul(Fr::fromU64(power)));
                power *= 2;
            }
            check(zg_sumcheck_final(s, fin.limbs), "final");
            zg_sumcheck_close(s);
            if (rep >= 0) t_raf += std::chrono::duration<double>(clk::now() - t1).count();
        }
    }
    // runLassoProver (src/zkvm/lasso/prover.zig:495-551): 2^v cycles, log_K = 16 address rounds + v cycle rounds on one session,
    // the reference's 64-bit challenge mixer between rounds
    double t_lasso = 0;
    {
        std::vector<unsigned __int128> lk(n);
        uint64_t x = 0x4c4153534fULL;
        for (size_t j = 0; j < n; j++) {
            x = x * 6364136223846793005ULL + 1442695040888963407ULL;
            lk[j] = (x >> 33) & 0xffff;
        }
        std::vector<Fr> w;
        for (int i = 0; i < v; i++) w.push_back(Fr::fromU64(1000003ULL * i + 17));
        for (int rep = -1; rep < reps; rep++) {
            auto t0 = clk::now();
            LassoProof pr = runLassoProver(lk, v, 16, w);
            if (rep >= 0) t_lasso += std::chrono::duration<double>(clk::now() - t0).count();
            ok = ok && pr.round_polys.size() == (size_t)(16 + v);
        }
    }
    // ValEvaluationProver (inc * wa * lt, val_evaluation.zig:554-660) and ProductVirtualRemainderProver (Gruen, product_remainder.zig:269-394):
    // one product-form session each, Keccak transcript between rounds
    double t_val = 0, t_prod = 0;
    {
        std::vector<Fr> a(n), b(n), c(n), tau;
        uint64_t x = 0x50534321ULL;
        for (size_t j = 0; j < n; j++) {
            x = x * 6364136223846793005ULL + 1442695040888963407ULL;
            a[j] = Fr::fromU64(x >> 3);
            b[j] = Fr::fromU64(x ^ (x >> 29));
            c[j] = a[j].add(b[j]);
        }
        for (int i = 0; i < v; i++) tau.push_back(Fr::fromU64(1000003ULL * i + 29));
        for (int rep = -1; rep < reps; rep++) {
            auto t0 = clk::now();
            double dt_val = 0, dt_prod = 0;
            {
                ValEvaluationProver p(a, b, &c, Fr::fromU64(5));
                t0 = clk::now();  // rounds only: the session set-up (three uploads, allocations) is a one-off per proof
                Transcript tr("Jolt");
                for (int k = 0; k < v; k++) {
                    auto ev = p.computeRoundPolynomial();
                    for (auto &e : ev) tr.appendScalar("val_eval", e);
                    p.bindChallengeWithPoly(tr.challengeScalar("val_eval_r"), ev);
                }
                ok = ok && p.getFinalClaims().size() == 3;
                dt_val = std::chrono::duration<double>(clk::now() - t0).count();
            }
            auto t1 = clk::now();
            {
                ProductVirtualRemainderProver q(a, b, tau, Fr::fromU64(3), Fr::fromU64(9));
                t1 = clk::now();
                Transcript tr("Jolt");
                for (int k = 0; k < v; k++) {
                    std::array<Fr, 4> ev;
                    q.roundEvals(ev);
                    for (auto &e : ev) tr.appendScalar("prod", e);
                    Fr ch = tr.challengeScalar("prod_r");
                    q.bindChallenge(ch);
                    q.updateClaim(ev, ch);
                }
                dt_prod = std::chrono::duration<double>(clk::now() - t1).count();
            }
            if (rep >= 0) {
                t_val += dt_val;
                t_prod += dt_prod;
            }
        }
    }
    // A Stage-2-shaped batched sumcheck (src/zkvm/batched_sumcheck.zig:1-21, proof_converter.zig:3026-3400): ProductVirtualRemainder (v rounds),
    // RamRafEvaluation (16), OutputSumcheck (16), InstructionLookupsClaimReduction (v) and a ValEvaluation of v + 8 rounds in the place of
    // RamReadWriteChecking (log_K + v rounds in the reference), one Blake2b transcript, generateBatchedProof
    double t_batch = 0;
    size_t batch_rounds = 0;
    if (v <= 16) {
        auto mkt = [&](size_t len, uint64_t seed) {
            std::vector<Fr> t(len);
            uint64_t x = seed;
            for (auto &e : t) {
                x = x * 6364136223846793005ULL + 1442695040888963407ULL;
                e = Fr::fromU64(x >> 5);
            }
            return t;
        };
        const size_t nk = size_t(1) << 16, nl = n << 8;
        auto left = mkt(n, 1), right = mkt(n, 2), ra = mkt(nk, 3), e2 = mkt(nk, 4), io = mkt(nk, 5), vf = mkt(nk, 6), vio = mkt(nk, 7), vin = mkt(nk, 8);
        auto e3 = mkt(n, 9), lo = mkt(n, 10), lf = mkt(n, 11), rt = mkt(n, 12), inc = mkt(nl, 13), wa = mkt(nl, 14), lt = mkt(nl, 15);
        std::vector<Fr> tau;
        for (int i = 0; i < v; i++) tau.push_back(Fr::fromU64(1000003ULL * i + 31));
        for (int rep = -1; rep < reps; rep++) {
            ProductVirtualRemainderProver pv(left, right, tau, Fr::fromU64(3), Fr::fromU64(11));
            RafEvaluationProver raf(ra, 0x7fff8000ULL, Fr::fromU64(12));
            ValEvaluationProver ve(inc, wa, &lt, Fr::fromU64(13));
            OutputSumcheckProver op(e2, io, vf, vio, vin, Fr::fromU64(14));
            InstructionLookupsClaimReductionProver il(e3, lo, lf, rt, Fr::fromU64(7), Fr::fromU64(15));
            std::array<Fr, 4> l0, l1, l2, l3, l4;
            BatchedSumcheckProver p;
            p.addInstance({(size_t)v, 3, pv.current_claim, [&](size_t) { pv.roundEvals(l0); return l0; }, [&](const Fr &c) { pv.updateClaim(l0, c); pv.bindChallenge(c); }});
            p.addInstance({16, 2, raf.current_claim, [&](size_t) { l1 = raf.computeRoundPolynomialCubic(); return l1; }, [&](const Fr &c) { raf.updateClaim(l1, c); raf.bindChallenge(c); }});
            p.addInstance({(size_t)v + 8, 3, ve.current_claim, [&](size_t) { l2 = ve.computeRoundPolynomial(); return l2; }, [&](const Fr &c) { ve.bindChallengeWithPoly(c, l2); }});
            p.addInstance({16, 3, op.current_claim, [&](size_t) { l3 = op.roundEvals(); return l3; }, [&](const Fr &c) { op.updateClaim(l3, c); op.bindChallenge(c); }});
            p.addInstance({(size_t)v, 2, il.current_claim, [&](size_t) { l4 = il.computeRoundPolynomialCubic(); return l4; }, [&](const Fr &c) { il.updateClaim(l4, c); il.bindChallenge(c); }});
            Blake2bTranscript tr("Jolt");
            auto t0 = clk::now();
            p.setupBatching(tr);
            BatchedSumcheckProof proof = generateBatchedProof(p, tr);
            if (rep >= 0) t_batch += std::chrono::duration<double>(clk::now() - t0).count();
            batch_rounds = proof.round_polys.size();
        }
        std::printf("{\"stage2_shaped_batched_proof_ms\": %.4f, \"stage2_shaped_rounds\": %zu, \"stage2_shaped_instance_rounds_per_s\": %.1f, ", t_batch / reps * 1e3,
                    batch_rounds, reps * (double)(2 * v + 32 + v + 8) / t_batch);
    } else {
        std::printf("{");
    }
    // RamReadWriteCheckingProver (src/zkvm/ram/read_write_checking.zig): 2^v cycles, 2^16 words, a memory access in one cycle of four
    // over 64 hot addresses, three phases (v / 2 cycle, 16 address, v / 2 cycle variables), Keccak transcript between rounds. The dense
    // tables (eq_evals, inc, val_init) fold on the device, the sparse entries on the host.
    double t_rwc = 0, t_rwc_setup = 0;
    size_t rwc_rounds = 0;
    {
        const size_t log_k = 16, T = n;
        std::vector<MemoryAccess> acc;
        std::vector<std::pair<uint64_t, uint64_t>> init;
        std::map<uint64_t, uint64_t> mem;
        const uint64_t start = 0x80000000ULL;
        for (size_t i = 0; i < 13; i++) { init.emplace_back(start + 8 * (4096 + i), splitmix() >> 8); mem[init.back().first] = init.back().second; }
        for (size_t ts = 0; ts < T; ts++) {
            uint64_t z = splitmix();
            if (z & 3) continue;
            uint64_t a = start + 8 * (4096 + ((z >> 8) & 63));
            if ((z >> 4) & 1) { uint64_t val = splitmix(); acc.push_back(MemoryAccess{ts, a, true, val}); mem[a] = val; }
            else acc.push_back(MemoryAccess{ts, a, false, mem.count(a) ? mem[a] : 0});
        }
        std::vector<Fr> rc(r.begin(), r.begin() + v);
        Fr gamma = Fr::fromU64(splitmix());
        for (int rep = -1; rep < (reps > 5 ? 5 : reps); rep++) {
            auto t0 = clk::now();
            RamReadWriteCheckingProver p(acc, gamma, rc, log_k, (size_t)v, (size_t)v / 2, start, Fr::zero());
            if (rep >= 0) t_rwc_setup += std::chrono::duration<double>(clk::now() - t0).count();
            Transcript tr("Jolt");
            const bool split = rep == 0 && std::getenv("ZOLT_RWC_SPLIT");  // per-round times of one run, to stderr
            if (split) std::fprintf(stderr, "rwc setup %.1f us\n", std::chrono::duration<double>(clk::now() - t0).count() * 1e6);
            size_t rd = 0;
            while (!p.isComplete()) {
                auto ta = clk::now();
                auto ev = p.computeRoundPolynomialCubic();
                auto tb = clk::now();
                for (auto &e : ev) tr.appendScalar("rwc", e);
                Fr ch = tr.challengeScalar("rwc_r");
                p.updateClaim(ev, ch);
                p.bindChallenge(ch);
                if (split)
                    std::fprintf(stderr, "rwc round %2zu: poly %8.1f us, bind %8.1f us\n", rd, std::chrono::duration<double>(tb - ta).count() * 1e6,
                                 std::chrono::duration<double>(clk::now() - tb).count() * 1e6);
                rd++;
            }
            if (rep >= 0) { t_rwc += std::chrono::duration<double>(clk::now() - t0).count(); rwc_rounds += p.numRounds(); }
        }
        std::printf("\"ram_read_write_checking_rounds_per_s\": %.1f, \"ram_read_write_checking_ms_incl_setup\": %.4f, \"ram_read_write_checking_setup_ms\": %.4f, "
                    "\"ram_read_write_checking_accesses\": %zu, ", rwc_rounds / t_rwc, t_rwc / (reps > 5 ? 5 : reps) * 1e3,
                    t_rwc_setup / (reps > 5 ? 5 : reps) * 1e3, acc.size());
    }
    // proveStage4 of the standard path (src/zkvm/prover.zig:713-828): Val evaluation over a memory trace of 2^v cycles, 2^16 words, a write in
    // one cycle of eight; tables (inc from the trace, wa = eq gather, lt = zg_fr_lt_table) + log T cubic rounds with a Keccak transcript
    {
        const size_t log_k = 16, T = n;
        const uint64_t start = 0x80000000ULL;
        std::vector<MemoryAccess> acc;
        for (size_t ts = 0; ts < T; ts++) {
            uint64_t z = splitmix();
            if (z & 7) continue;
            acc.push_back(MemoryAccess{ts, start + 8 * ((z >> 8) & 0xFFFF), true, splitmix() >> 2});
        }
        double t_s4 = 0;
        const int n4 = reps > 3 ? 3 : reps;
        for (int rep = -1; rep < n4; rep++) {
            Transcript tr("Jolt");
            auto t0 = clk::now();
            auto res = proveStage4(acc, {}, T, log_k, (size_t)v, start, tr);
            if (rep >= 0) t_s4 += std::chrono::duration<double>(clk::now() - t0).count();
            ok = ok && res.round_polys.size() == (size_t)v;
        }
        std::printf("\"stage4_val_evaluation_ms_incl_tables\": %.4f, \"stage4_val_evaluation_writes\": %zu, ", t_s4 / n4 * 1e3, acc.size());
    }
    // Stage4GruenProver (src/zkvm/spartan/stage4_gruen_prover.zig), RegistersReadWriteChecking: 128 registers x 2^min(v, 18) cycles (five dense
    // tables of 128 * T elements built and folded on the device), phases T/2 cycle, 7 register, T/2 cycle variables, Keccak transcript
    {
        const size_t lt = v > 18 ? 18 : (size_t)v, T4 = size_t(1) << lt;
        std::vector<TraceStep> steps(T4);
        static const uint32_t ops[11] = {0x13, 0x03, 0x67, 0x1B, 0x33, 0x3B, 0x23, 0x63, 0x37, 0x6F, 0x17};
        for (auto &st : steps) {
            uint64_t z = splitmix();
            st.instruction = ops[z % 11] | (uint32_t)(((z >> 8) & 31) << 7) | (uint32_t)(((z >> 16) & 31) << 15) | (uint32_t)(((z >> 24) & 31) << 20);
            st.rd_value = splitmix();
            st.is_noop = false;
        }
        std::vector<Fr> rc(r.begin(), r.begin() + lt);
        Fr gamma = Fr::fromU64(splitmix());
        setenv("ZG_SETUP_TIMES", "1", 1);
        double t_s4 = 0, t_s4_rounds = 0, ph4[4] = {0, 0, 0, 0};
        const int n4 = reps > 3 ? 3 : reps;
        for (int rep = -1; rep < n4; rep++) {
            auto t0 = clk::now();
            Stage4GruenProver p(steps, gamma, rc, lt / 2 ? lt / 2 : 1, 7);
            auto t1 = clk::now();
            if (rep >= 0) {
                double p4[4];
                zg_last_setup_times(p4);
                for (int k = 0; k < 4; k++) ph4[k] += p4[k];
            }
            Transcript tr("Jolt");
            Fr claim = Fr::zero();
            for (size_t rd = 0; rd < p.num_rounds; rd++) {
                auto ev = p.computeRoundEvals(rd, claim);
                for (auto &e : ev) tr.appendScalar("s4", e);
                Fr ch = tr.challengeScalar("s4_r");
                claim = cubicAtPoint(ev, ch);
                p.bindChallenge(rd, ch);
            }
            (void)p.getFinalClaims();
            if (rep >= 0) { t_s4 += std::chrono::duration<double>(clk::now() - t0).count(); t_s4_rounds += std::chrono::duration<double>(clk::now() - t1).count(); }
        }
        std::printf("\"stage4_registers_log_t\": %zu, \"stage4_registers_ms_incl_setup\": %.4f, \"stage4_registers_setup_ms\": %.4f, "
                    "\"stage4_registers_setup_split_ms\": {\"alloc\": %.4f, \"h2d\": %.4f, \"kernels\": %.4f}, \"stage4_registers_rounds_ms\": %.4f, "
                    "\"stage4_registers_rounds_per_s\": %.1f, ",
                    lt, t_s4 / n4 * 1e3, (t_s4 - t_s4_rounds) / n4 * 1e3, ph4[0] / n4, ph4[1] / n4, ph4[2] / n4, t_s4_rounds / n4 * 1e3, n4 * (double)(7 + lt) / t_s4_rounds);
    }
    // StreamingOuterProver (src/zkvm/spartan/streaming_outer.zig) over 2^min(v, 20) cycles. The witness matrix is built ON THE DEVICE from the
    // integer columns of a synthetic trace (zolt::CycleColumns::fromTrace + zg_fr_rows_from_columns: 156 bytes per cycle cross PCIe instead
    // of 43 field elements = 1376); Az / Bz are materialised by one launch, then 1 + log T Gruen rounds with a Keccak transcript. The set-up
    // is reported phase by phase: the host's column decode (the reference spends this loop building field-element rows), the device allocation,
    // the copies, the widening kernel (ZG_SETUP_TIMES=1 separates the last three with synchronisations), the prover's own tables.
    std::shared_ptr<CycleWitnessMatrix> shared_matrix;  // Stage 1's matrix stays resident: Stage 3 below reads the same one
    {
        const size_t lt = v > 20 ? 20 : (size_t)v, To = size_t(1) << lt;
        static const uint32_t ops[13] = {0x33, 0x13, 0x03, 0x23, 0x63, 0x37, 0x17, 0x6F, 0x67, 0x1B, 0x3B, 0x73, 0x0F};
        std::vector<R1CSTraceStep> trace(To);
        for (size_t i = 0; i < To; i++) {
            auto &st = trace[i];
            if (i >= To - To / 16) { st.is_noop = true; continue; }  // NoOp padding at the end, as padWithNoop leaves it
            uint64_t z = splitmix();
            st.instruction = ((uint32_t)(z >> 32) & ~0x7Fu) | ops[z % 13];
            st.pc = st.unexpanded_pc = 0x80000000ULL + 4 * (splitmix() & 0xFFFFF);
            st.rs1_value = splitmix();
            st.rs2_value = splitmix();
            st.rd_value = splitmix();
            st.has_memory_value = (z >> 8) & 1;
            st.memory_value = splitmix();
        }
        std::vector<Fr> tau(lt + 2);
        for (auto &x : tau) x = Fr::fromU64(splitmix());
        Fr r0 = Fr::fromU64(splitmix()), scale = Fr::fromU64(splitmix());
        setenv("ZG_SETUP_TIMES", "1", 1);  // read once by the library: set before the first call that looks at it
        double t_cols = 0, t_matrix = 0, t_ctor = 0, t_mat = 0, t_rounds = 0, t_first = 0, ph[4] = {0, 0, 0, 0};
        size_t bytes_per_cycle = 0;
        const int no = reps > 3 ? 3 : reps;
        for (int rep = -1; rep < no; rep++) {
            shared_matrix.reset();  // (the previous repetition's matrix goes back to the pool before the next one is built)
            auto t0 = clk::now();
            CycleColumns cols = CycleColumns::fromTrace(trace);
            auto ta = clk::now();
            auto matrix = CycleWitnessMatrix::fromColumns(cols);
            auto tb = clk::now();
            double p4[4];
            zg_last_setup_times(p4);
            StreamingOuterProver p(matrix, tau, &scale);
            auto tf = clk::now();
            (void)p.computeFirstRoundPoly();  // the UniSkip first round: t1 at nine targets, 28 coefficients
            auto tg = clk::now();
            p.bindFirstRoundChallenge(r0, Fr::zero());
            auto t1 = clk::now();
            p.materializeLinearPhasePolynomials();
            auto t2 = clk::now();
            Transcript tr("Jolt");
            for (size_t rd = 0; rd < p.numRounds(); rd++) {
                auto ev = p.computeRemainingRoundPoly();
                for (auto &e : ev) tr.appendScalar("so", e);
                Fr ch = tr.challengeScalar("so_r");
                p.updateClaim(ev, ch);
                p.bindRemainingRoundChallenge(ch);
            }
            (void)p.finalAzBz();
            auto t3 = clk::now();
            bytes_per_cycle = cols.bytesPerCycle();
            shared_matrix = matrix;
            if (rep >= 0) {
                auto d = [](clk::time_point x, clk::time_point y) { return std::chrono::duration<double>(y - x).count(); };
                t_cols += d(t0, ta); t_matrix += d(ta, tb); t_ctor += d(tb, tf); t_first += d(tf, tg); t_mat += d(t1, t2); t_rounds += d(t2, t3);
                for (int k = 0; k < 4; k++) ph[k] += p4[k];
            }
        }
        // the rows-of-field-elements upload of rounds 3 / 4 beside it, once (1376 bytes per cycle through zg_memcpy_h2d)
        double t_fr_rows = 0;
        {
            auto m0 = CycleWitnessMatrix::fromColumns(CycleColumns::fromTrace(trace));
            std::vector<Fr> rows = m0->toHost();
            m0.reset();
            for (int rep = -1; rep < 2; rep++) {
                auto t0 = clk::now();
                auto m1 = CycleWitnessMatrix::fromWitnesses(rows.data(), To);
                if (rep >= 0) t_fr_rows += std::chrono::duration<double>(clk::now() - t0).count() / 2;
            }
        }
        // the same matrix through CycleWitnessMatrix::fromTrace: slices of cycles, the decode of one beside the upload of the previous
        // (what the prove path calls); the slice count is swept once so the default can be read against its neighbours
        {
            std::printf("\"outer_trace_to_matrix_streamed_ms\": {");
            bool first = true;
            for (const char *sl : {"", "1", "2", "4", "8", "16"}) {
                if (*sl) setenv("ZOLT_WITNESS_SLICES", sl, 1);
                double t = 0;
                for (int rep = -1; rep < 3; rep++) {
                    shared_matrix.reset();
                    auto t0 = clk::now();
                    shared_matrix = CycleWitnessMatrix::fromTrace(trace);
                    if (rep >= 0) t += std::chrono::duration<double>(clk::now() - t0).count() / 3;
                }
                std::printf("%s\"%s\": %.4f", first ? "" : ", ", *sl ? sl : "default", t * 1e3);
                first = false;
            }
            unsetenv("ZOLT_WITNESS_SLICES");
            std::printf("}, ");
        }
        std::printf("\"outer_log_t\": %zu, \"outer_column_bytes_per_cycle\": %zu, \"outer_host_columns_ms\": %.4f, \"outer_upload_ms\": %.4f, "
                    "\"outer_upload_split_ms\": {\"alloc\": %.4f, \"h2d\": %.4f, \"widen_kernel\": %.4f}, \"outer_prover_tables_ms\": %.4f, "
                    "\"outer_uniskip_first_round_ms\": %.4f, \"outer_materialize_ms\": %.4f, \"outer_upload_plus_first_round_plus_materialize_ms\": %.4f, "
                    "\"outer_upload_fr_rows_ms\": %.4f, \"outer_rounds_ms\": %.4f, \"outer_rounds_per_s\": %.1f, ",
                    lt, bytes_per_cycle, t_cols / no * 1e3, t_matrix / no * 1e3, ph[0] / no, ph[1] / no, ph[2] / no, t_ctor / no * 1e3, t_first / no * 1e3, t_mat / no * 1e3,
                    (t_matrix + t_ctor + t_first + t_mat) / no * 1e3, t_fr_rows * 1e3, t_rounds / no * 1e3, no * (double)(lt + 1) / t_rounds);
    }
    // Stage 3 (src/zkvm/spartan/stage3_prover.zig): ShiftSumcheck + InstructionInput + RegistersClaimReduction over 2^min(v, 20) padded cycles,
    // the witness matrix resident in HBM (Stage 1's upload): build of the three instances, then log T batched rounds with a Keccak transcript
    {
        const size_t lt = v > 20 ? 20 : (v < 2 ? 2 : (size_t)v), T3 = size_t(1) << lt;
        const bool shared = shared_matrix && shared_matrix->num_cycles == T3;
        DeviceMem d_own;
        if (!shared) {  // sizes where no Stage-1 matrix of this length exists: a matrix of its own
            std::vector<Fr> w(T3 * 43);
            for (auto &x : w) x = Fr::fromU64(splitmix());
            d_own.alloc(T3 * 43 * 32);
            check(zg_memcpy_h2d(d_own.p, w.data(), T3 * 43 * 32), "h2d");
        }
        const uint64_t *d_rows_ptr = shared ? shared_matrix->u64() : d_own.u64();
        std::vector<Fr> ro(lt), rp(lt), sg(5);
        for (auto &x : ro) x = Fr::fromU64(splitmix());
        for (auto &x : rp) x = Fr::fromU64(splitmix());
        Fr g = Fr::fromU64(splitmix());
        sg[0] = Fr::one();
        for (int i = 1; i < 5; i++) sg[i] = sg[i - 1].mul(g);
        std::array<Fr, 3> cl = {Fr::fromU64(splitmix()), Fr::fromU64(splitmix()), Fr::fromU64(splitmix())}, co = cl;
        double t_build = 0, t_rounds = 0;
        const int n3 = reps > 3 ? 3 : reps;
        for (int rep = -1; rep < n3; rep++) {
            auto t0 = clk::now();
            Stage3Prover p(d_rows_ptr, ro, rp, sg, g, sg[2], cl, co);
            auto t1 = clk::now();
            Transcript tr("Jolt");
            for (size_t rd = 0; rd < lt; rd++) {
                auto c = p.computeRoundPolynomial();
                for (auto &e : c) tr.appendScalar("s3", e);
                p.bindChallenge(tr.challengeScalar("s3_r"));
            }
            (void)p.shift.finalClaims();
            (void)p.reg.finalClaims();
            auto t2 = clk::now();
            if (rep >= 0) {
                t_build += std::chrono::duration<double>(t1 - t0).count();
                t_rounds += std::chrono::duration<double>(t2 - t1).count();
            }
        }
        std::printf("\"stage3_log_t\": %zu, \"stage3_reads_stage1_matrix\": %s, \"stage3_build_ms\": %.4f, \"stage3_rounds_ms\": %.4f, \"stage3_rounds_per_s\": %.1f, ", lt,
                    shared ? "true" : "false", t_build / n3 * 1e3, t_rounds / n3 * 1e3, n3 * (double)lt / t_rounds);
        shared_matrix.reset();
    }
    std::printf("\"val_evaluation_rounds_per_s\": %.1f, \"val_evaluation_ms\": %.4f, \"product_remainder_rounds_per_s\": %.1f, "
                "\"product_remainder_ms\": %.4f, ", reps * v / t_val, t_val / reps * 1e3, reps * v / t_prod, t_prod / reps * 1e3);
    std::printf("\"lasso_log_K16_rounds_per_s\": %.1f, \"lasso_ms_whole_protocol_incl_setup\": %.4f, ", reps * (16 + v) / t_lasso, t_lasso / reps * 1e3);
    std::printf("\"v\": %d, \"reps\": %d, \"verified\": %s, \"stage1_keccak_rounds_per_s\": %.1f, \"stage1_ms\": %.4f, "
                "\"raf_cubic_rounds_per_s\": %.1f, \"raf_ms\": %.4f, \"device_resident_rounds_per_s\": %.1f, "
                "\"device_resident_ms_runSumcheck\": %.4f, ", v, reps, ok ? "true" : "false", reps * v / t_s1, t_s1 / reps * 1e3, reps * v / t_raf,
                t_raf / reps * 1e3, reps * v / t_dev, t_dev / reps * 1e3);
    std::printf("\"rounds_per_s\": %.1f, \"us_per_round\": %.2f, "
                "\"ms_runSumcheck\": %.4f, \"ms_eq_table\": %.4f, \"ms_spartan_combine\": %.4f, "
                "\"rounds_per_s_incl_eq_and_combine\": %.1f, \"fused_open_ms_whole_instance\": %.4f, "
                "\"fused_open_rounds_per_s_incl_eq_and_combine\": %.1f}\n",
                reps * v / t_sc, t_sc / (reps * v) * 1e6, t_sc / reps * 1e3, t_eq / reps * 1e3,
                t_comb / reps * 1e3, reps * v / (t_sc + t_eq + t_comb), t_fused / reps * 1e3, reps * v / t_fused);
    zg_dev_free(d_tab); zg_dev_free(d_eq); zg_dev_free(d_f);
    zg_shutdown();
    return ok ? 0 : 1;
}
