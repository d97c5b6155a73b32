// bench_sumcheck.cpp — BASELINE config 3 with a compiled host loop: 20-variable sumcheck
// (runSumcheck, src/subprotocols/mod.zig:302-354) over a table resident in HBM, toy verifier on the host,
// plus the eq-table build and Spartan combine that precede it (src/zkvm/spartan/mod.zig:182-206).
// Prints one JSON object. Usage: bench_sumcheck [v=20] [reps=20]
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "../zolt_amd/host/zolt_host.hpp"

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
    std::printf("{\"v\": %d, \"reps\": %d, \"verified\": %s, \"device_resident_rounds_per_s\": %.1f, "
                "\"device_resident_ms_runSumcheck\": %.4f, ", v, reps, ok ? "true" : "false", reps * v / t_dev, t_dev / reps * 1e3);
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
