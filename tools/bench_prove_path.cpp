// bench_prove_path.cpp — the hot path of ONE proof as a single sequence, from compiled host code over the C ABI.
//
// `zolt prove` is one sequence (JoltProver.prove, src/zkvm/mod.zig:366-452): proving key (HyperKZG.setup) -> commitBytecode / commitMemory /
// commitRegisters (:1518-1617) -> transcript absorbs the commitments -> MultiStageProver.prove (src/zkvm/prover.zig:302-1110): stage 1
// (JoltR1CS.fromTrace, Az / Bz, eq(tau) * Az * Bz, log2(19 T) LowToHigh rounds, Az(r) / Bz(r)), stage 2 (RAF over the memory trace), stage 3
// (Lasso), stages 4-6 — and a HyperKZG.open of a committed polynomial at the sumcheck point is what the verifier's opening check needs next
// (src/poly/commitment/mod.zig:261-324). Rounds 1-4 reported some thirty per-site figures; this tool runs the sequence ONCE per repetition,
// uploads included, and prints one JSON object with the cost of every call, so that the top costs of a proof are named, not guessed.
//
//   bench_prove_path synth <log_t> [reps]     a synthetic trace of 2^log_t cycles (random RV64 instruction mix, one memory access in
//                                             eight cycles, 16-bit lookup indices, 64 KiB of program bytes)
//   bench_prove_path file <case> [reps]       the inputs of a real run (written by tests/test_gpu_cpp_host.py from the ELF): the stage
//                                             records are printed as `test_host_mirror proof` prints them (S / P / H / C lines) together
//                                             with the three header commitments (K lines), so the bytes can be held against the captured
//                                             proof file; the JSON line comes last
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../zolt_amd/host/zolt_host.hpp"
#include "../include/zolt_gpu_internal.h"

using namespace zolt;
using zolt::wire::commitmentToBytes;
using clk = std::chrono::steady_clock;

static uint64_t sm_state = 0x50524F5645ULL;
static uint64_t splitmix() {
    uint64_t z = (sm_state += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

struct ProveCase {
    size_t log_t = 0, log_k = 16, srs_size = 0;
    uint64_t start_address = 0x80000000ULL;
    std::vector<R1CSTraceStep> trace;               // NoOp-padded to 2^log_t
    std::vector<MemoryAccess> accesses;             // MemoryTrace.accesses
    std::vector<unsigned __int128> lookup_indices;  // LookupTraceCollector entries (one per executed cycle)
    std::vector<uint8_t> bytecode;
};

static ProveCase synthetic_case(size_t log_t) {
    ProveCase c;
    c.log_t = log_t;
    const size_t T = size_t(1) << log_t;
    c.srs_size = T;
    static const uint32_t ops[13] = {0x33, 0x13, 0x03, 0x23, 0x63, 0x37, 0x17, 0x6F, 0x67, 0x1B, 0x3B, 0x73, 0x0F};
    c.trace.resize(T);
    const size_t executed = T - T / 16;  // NoOp padding at the end, as padWithNoop leaves it
    for (size_t i = 0; i < executed; i++) {
        auto &st = c.trace[i];
        uint64_t z = splitmix();
        st.instruction = ((uint32_t)(z >> 32) & ~0x7Fu) | ops[z % 13];
        st.pc = st.unexpanded_pc = 0x80000000ULL + 4 * (splitmix() & 0xFFFFF);
        st.rs1_value = splitmix();
        st.rs2_value = splitmix();
        st.rd_value = splitmix();
        st.has_memory_value = (z >> 8) & 1;
        st.memory_value = splitmix();
        if ((z & 7) == 0) c.accesses.push_back(MemoryAccess{i, c.start_address + 8 * ((z >> 8) & 0xFFFF), true, splitmix() >> 2});
    }
    for (size_t i = executed; i < T; i++) c.trace[i].is_noop = true;
    c.lookup_indices.resize(executed);
    for (auto &v : c.lookup_indices) v = splitmix() & 0xFFFF;
    c.bytecode.resize(size_t(1) << 16);
    for (auto &b : c.bytecode) b = (uint8_t)splitmix();
    return c;
}

static Fr read_fr_hex(std::FILE *f) {
    Fr x;
    for (int i = 0; i < 4; i++) {
        unsigned long long v = 0;
        if (std::fscanf(f, "%llx", &v) != 1) throw std::runtime_error("case file: truncated field element");
        x.limbs[i] = v;
    }
    return x;
}
static ProveCase file_case(const char *path) {
    std::FILE *f = std::fopen(path, "r");
    if (!f) throw std::runtime_error(std::string("cannot open ") + path);
    ProveCase c;
    unsigned long long a, b, n;
    if (std::fscanf(f, "%llu %llu %llu %llx", &a, &b, &n, (unsigned long long *)&c.start_address) != 4) throw std::runtime_error("case file: header");
    c.log_t = a;
    c.log_k = b;
    c.srs_size = n;
    if (std::fscanf(f, "%llu", &n) != 1) throw std::runtime_error("case file: trace length");
    c.trace.resize(n);
    for (auto &st : c.trace) {
        unsigned long long w, pc, upc, r1, r2, rd, hm, mem, comp, noop;
        if (std::fscanf(f, "%llx %llx %llx %llx %llx %llx %llu %llx %llu %llu", &w, &pc, &upc, &r1, &r2, &rd, &hm, &mem, &comp, &noop) != 10) throw std::runtime_error("case file: trace step");
        st.instruction = (uint32_t)w; st.pc = pc; st.unexpanded_pc = upc; st.rs1_value = r1; st.rs2_value = r2; st.rd_value = rd;
        st.has_memory_value = hm != 0; st.memory_value = mem; st.is_compressed = comp != 0; st.is_noop = noop != 0;
    }
    if (std::fscanf(f, "%llu", &n) != 1) throw std::runtime_error("case file: access count");
    for (size_t i = 0; i < n; i++) {
        unsigned long long ts, ad, wr, val;
        if (std::fscanf(f, "%llu %llx %llu %llx", &ts, &ad, &wr, &val) != 4) throw std::runtime_error("case file: access");
        c.accesses.push_back(MemoryAccess{ts, ad, wr != 0, val});
    }
    if (std::fscanf(f, "%llu", &n) != 1) throw std::runtime_error("case file: lookup count");
    c.lookup_indices.resize(n);
    for (auto &v : c.lookup_indices) {
        unsigned long long lo, hi;
        if (std::fscanf(f, "%llu %llu", &lo, &hi) != 2) throw std::runtime_error("case file: lookup index");
        v = ((unsigned __int128)hi << 64) | lo;
    }
    if (std::fscanf(f, "%llu", &n) != 1) throw std::runtime_error("case file: bytecode length");
    c.bytecode.resize(n);
    for (auto &x : c.bytecode) { unsigned v; if (std::fscanf(f, "%2x", &v) != 1) throw std::runtime_error("case file: bytecode"); x = (uint8_t)v; }
    std::fclose(f);
    (void)read_fr_hex;
    return c;
}

struct StepTime { std::string name, kind; double ms = 0; };
struct Timeline {
    std::vector<StepTime> steps;
    clk::time_point t0;
    void start() { t0 = clk::now(); }
    void lap(const char *name, const char *kind) {  // everything since the last lap belongs to `name`
        auto t1 = clk::now();
        steps.push_back({name, kind, std::chrono::duration<double>(t1 - t0).count() * 1e3});
        t0 = t1;
    }
};

static size_t ceil_pow2(size_t n) { size_t p = 1; while (p < n) p <<= 1; return p; }
static void print_fr(const Fr &x) { std::printf(" %llx %llx %llx %llx", (unsigned long long)x.limbs[0], (unsigned long long)x.limbs[1], (unsigned long long)x.limbs[2], (unsigned long long)x.limbs[3]); }
static void line(bool on, const char *tag, const std::vector<Fr> &v) {
    if (!on) return;
    std::printf("%s", tag);
    for (const Fr &x : v) print_fr(x);
    std::printf("\n");
}

// the 19 uniform constraints as affine maps of a cycle's 43 inputs: rows 0..18 = condition (Az), 19..37 = left - right (Bz); constant last
static std::vector<Fr> constraint_maps() {
    const size_t W = r1cs::NUM_INPUTS + 1;
    std::vector<Fr> m(38 * W, Fr::zero());
    const Fr two64 = Fr::fromU64(uint64_t(1) << 32).mul(Fr::fromU64(uint64_t(1) << 32));
    const auto &cs = r1cs::uniformConstraints();
    auto add = [&](size_t row, const r1cs::LC &l, bool negate) {
        for (const auto &t : l.terms) m[row * W + t.input] = m[row * W + t.input].add(r1cs::fromInt(negate ? -t.coeff : t.coeff));
        Fr c = r1cs::fromInt(l.constant);
        if (l.two_pow_64) c = c.add(two64);
        m[row * W + r1cs::NUM_INPUTS] = negate ? m[row * W + r1cs::NUM_INPUTS].sub(c) : m[row * W + r1cs::NUM_INPUTS].add(c);
    };
    for (size_t i = 0; i < 19; i++) {
        add(i, cs[i].condition, false);
        add(19 + i, cs[i].left, false);
        add(19 + i, cs[i].right, true);
    }
    return m;
}

// 0: the key's handle is planned as an SRS that lives on (table of multiples built with the key); k > 0: planned for k MSMs — a key that
// serves one proof (`zolt prove` builds its mock SRS in-process, src/main.zig:271-696) does three commits and an open, fewer MSMs than the
// table's break-even (bench.py config.breakeven_msms), so the table-less plan is the faster whole
static int key_uses = 0, key_levels = 0;  // key_levels: zg_msm_config.precompute_levels (0 = the plan picks)

// one proof; `emit`: print the stage records (file mode)
static Timeline prove_once(const ProveCase &pc, bool emit) {
    Timeline tl;
    const size_t T = size_t(1) << pc.log_t;
    tl.start();
    // ---- proving key: HyperKZG.setup (mock SRS: tau^i * G by the fixed-base kernel), uploaded with its table of multiples
    // the prover commits and opens: the points never come back to the host. key_uses > 0: the key is planned for that many MSMs (no table)
    const zg_msm_config key_cfg{0, key_levels, key_uses};
    HyperKZG::SetupParams pk = HyperKZG::setup(pc.srs_size, false, key_uses > 0 || key_levels > 0 ? &key_cfg : nullptr);
    tl.lap("proving key: HyperKZG.setup on the device (powers of tau, fixed-base batch, table of multiples)", "once per key");
    // ---- the three commitments, from machine words (zg_msm_g1_u64)
    // (in pinned memory, as a shim would hold them: PinnedWords — a pageable vector per proof is pinned on the fly and unpinned again by
    // the HIP runtime, usually cheaply, by 10-15 ms when the process's address reuse changes)
    PinnedWords bc(pc.bytecode.size() < 2 ? 2 : ceil_pow2(pc.bytecode.size()), false), mem(pc.accesses.size() < 2 ? 2 : ceil_pow2(pc.accesses.size()), false),
        reg(pc.trace.size() < 2 ? 2 : ceil_pow2(pc.trace.size()), false);
    {  // every word is written exactly once (value or zero padding), long vectors by several host threads
        auto fill = [](uint64_t *dst, size_t total, size_t live, auto value) {
            const size_t T = total >= (size_t(1) << 16) ? std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), 8) : 1;
            std::vector<std::thread> pool;
            for (size_t t = 0; t < T; t++) {
                const size_t a = total * t / T, b = total * (t + 1) / T;
                auto work = [=] { for (size_t i = a; i < b; i++) dst[i] = i < live ? value(i) : 0; };
                if (T == 1) work(); else pool.emplace_back(work);
            }
            for (auto &th : pool) th.join();
        };
        fill(bc.data(), bc.size(), pc.bytecode.size(), [&](size_t i) { return (uint64_t)pc.bytecode[i]; });
        fill(mem.data(), mem.size(), pc.accesses.size(), [&](size_t i) { return pc.accesses[i].value; });
        fill(reg.data(), reg.size(), pc.trace.size(), [&](size_t i) { return pc.trace[i].rd_value; });
    }
    tl.lap("commit: host builds the three u64 polynomials", "host");
    HyperKZG::Commitment c_bc = HyperKZG::commitU64(pk, bc);
    tl.lap("commitBytecode (MSM over machine words)", "h2d+kernels");
    HyperKZG::Commitment c_mem = pc.accesses.empty() ? HyperKZG::Commitment{AffinePoint::identity()} : HyperKZG::commitU64(pk, mem);
    tl.lap("commitMemory", "h2d+kernels");
    HyperKZG::Commitment c_reg = HyperKZG::commitU64(pk, reg);
    tl.lap("commitRegisters", "h2d+kernels");
    Transcript tr("Jolt");
    const auto zero64 = commitmentToBytes(AffinePoint::identity());
    for (const auto &b : {commitmentToBytes(c_bc.point), commitmentToBytes(c_mem.point), zero64, commitmentToBytes(c_reg.point), zero64}) tr.appendBytes(b.data(), 64);
    if (emit) {
        for (const auto &kv : {std::make_pair("bytecode", c_bc), std::make_pair("memory", c_mem), std::make_pair("register", c_reg)}) {
            std::printf("K %s ", kv.first);
            for (uint8_t x : commitmentToBytes(kv.second.point)) std::printf("%02x", x);
            std::printf("\n");
        }
    }
    tl.lap("transcript absorbs the commitments", "host");
    // ---- stage 1 (prover.zig:344-450): witness, Az / Bz, eq(tau) * Az * Bz, LowToHigh rounds, Az(r) / Bz(r)
    auto matrix = CycleWitnessMatrix::fromTrace(pc.trace);
    tl.lap("stage 1: trace -> 156-byte integer columns -> witness matrix in HBM (slices of cycles: decode on host threads beside the upload + widening of the previous slice)", "host+h2d+kernels");
    const size_t n_constraints = pc.trace.size() * 19, padded = n_constraints ? ceil_pow2(n_constraints) : 1;
    size_t rounds1 = 0;
    while ((size_t(1) << rounds1) < padded) rounds1++;
    DeviceMem d_az(padded * 32), d_bz(padded * 32);
    // (the constraint rows of every cycle are written below: only the padding behind them is cleared)
    check(zg_dev_memset(d_az.u64() + 4 * n_constraints, 0, (padded - n_constraints) * 32), "zg_dev_memset");
    check(zg_dev_memset(d_bz.u64() + 4 * n_constraints, 0, (padded - n_constraints) * 32), "zg_dev_memset");
    const std::vector<Fr> maps = constraint_maps();
    const size_t W = r1cs::NUM_INPUTS + 1;
    for (int t = 0; t < 2; t++) {
        uint64_t *dst = t ? d_bz.u64() : d_az.u64();
        const uint64_t *m = reinterpret_cast<const uint64_t *>(maps.data() + (size_t)t * 19 * W);
        check(zg_fr_rows_affine_records_dev(matrix->u64(), pc.trace.size(), r1cs::NUM_INPUTS, 0, m, 16, 19, 0, dst, nullptr), "zg_fr_rows_affine_records_dev");
        check(zg_fr_rows_affine_records_dev(matrix->u64(), pc.trace.size(), r1cs::NUM_INPUTS, 0, m + 4 * 16 * W, 3, 19, 16, dst, nullptr), "zg_fr_rows_affine_records_dev");
    }
    check(zg_sync(), "zg_sync");
    tl.lap("stage 1: Az / Bz materialised in HBM (JoltR1CS layout, cycle * 19 + constraint)", "kernels");
    std::vector<Fr> tau(rounds1), ch;
    for (auto &x : tau) x = tr.challengeScalar("spartan_tau");
    zg_sc_t s1 = nullptr, sa = nullptr, sb = nullptr;
    check(zg_sumcheck_open_spartan_dev(reinterpret_cast<const uint64_t *>(tau.data()), rounds1, nullptr, d_az.u64(), d_bz.u64(), nullptr, ZG_SC_LOW_PAIR, nullptr, &s1),
          "zg_sumcheck_open_spartan_dev");
    // Az and Bz are bound alongside for Az(r), Bz(r): their sessions read the materialised tables in place (no second copy of 1 GB each)
    check(zg_sumcheck_open_dev_borrowed(d_az.u64(), padded, ZG_SC_LOW_PAIR, nullptr, &sa), "zg_sumcheck_open_dev_borrowed");
    check(zg_sumcheck_open_dev_borrowed(d_bz.u64(), padded, ZG_SC_LOW_PAIR, nullptr, &sb), "zg_sumcheck_open_dev_borrowed");
    if (emit) std::printf("S 1\n");
    Fr initial1 = Fr::zero();
    for (size_t k = 0; k < rounds1; k++) {
        Fr p0 = Fr::zero(), p1 = Fr::zero();
        if (padded >> k >= 2) check(zg_sumcheck_round_sums(s1, p0.limbs, p1.limbs), "zg_sumcheck_round_sums");
        if (k == 0) initial1 = p0.add(p1);
        const Fr p2 = p1.add(p1).sub(p0);
        tr.appendScalar("round_poly_0", p0);
        tr.appendScalar("round_poly_1", p1);
        tr.appendScalar("round_poly_2", p2);
        const Fr c = tr.challengeScalar("spartan_round");
        ch.push_back(c);
        line(emit, "P", {p0, p1, p2});
        check(zg_sumcheck_bind(s1, c.limbs), "zg_sumcheck_bind");
        check(zg_sumcheck_bind(sa, c.limbs), "zg_sumcheck_bind");
        check(zg_sumcheck_bind(sb, c.limbs), "zg_sumcheck_bind");
    }
    Fr fin1 = Fr::zero(), az_r = Fr::zero(), bz_r = Fr::zero();
    check(zg_sumcheck_final(s1, fin1.limbs), "zg_sumcheck_final");
    check(zg_sumcheck_final(sa, az_r.limbs), "zg_sumcheck_final");
    check(zg_sumcheck_final(sb, bz_r.limbs), "zg_sumcheck_final");
    zg_sumcheck_close(s1);
    zg_sumcheck_close(sa);
    zg_sumcheck_close(sb);
    line(emit, "H", ch);
    line(emit, "C", {initial1, fin1, az_r, bz_r, Fr::zero()});
    tl.lap("stage 1: eq(tau) * Az * Bz fused open + log2(19 T) rounds (three LowToHigh sessions, Keccak transcript)", "kernels+host");
    // ---- stage 2: RAF over the memory trace (prover.zig:452-560): ra(k) = sum over the accesses of slot k of eq(r_cycle, timestamp)
    std::vector<Fr> r_cycle(pc.log_t);
    for (auto &x : r_cycle) x = tr.challengeScalar("r_cycle");
    const size_t K = size_t(1) << pc.log_k;
    std::vector<Fr> ra(K, Fr::zero());
    if (!pc.accesses.empty()) {
        // RaPolynomial.fromTrace (src/zkvm/ram/raf_checking.zig:88-132): ra(k) += eq(r_cycle, j) for ACCESS j of slot k, the eq table over
        // ceil(log2(#accesses)) variables, little-endian (computeEqEvals :535-567: index bit i <-> r[i]) = the device's big-endian table of
        // the reversed point; missing challenges read as zero
        size_t log_a = 0;
        while ((size_t(1) << log_a) < pc.accesses.size()) log_a++;
        std::vector<Fr> r_rev(log_a, Fr::zero());
        for (size_t i = 0; i < log_a; i++)
            if (i < r_cycle.size()) r_rev[log_a - 1 - i] = r_cycle[i];
        std::vector<Fr> eq(size_t(1) << log_a);
        check(zg_fr_eq_table(reinterpret_cast<const uint64_t *>(r_rev.data()), log_a, nullptr, reinterpret_cast<uint64_t *>(eq.data())), "zg_fr_eq_table");
        for (size_t j = 0; j < pc.accesses.size(); j++) {
            const uint64_t a = pc.accesses[j].address;
            if (a < pc.start_address) continue;
            const size_t k = (a - pc.start_address) / 8;
            if (k < K) ra[k] = ra[k].add(eq[j]);
        }
    }
    tl.lap("stage 2: RaPolynomial.fromTrace (eq table on the device, per-slot sums on the host)", "kernels+d2h+host");
    {
        RafEvaluationProver raf(ra, pc.start_address);  // the claim = computeInitialClaim(): sum_k ra(k) * unmap(k) over the resident table
        const Fr claim2 = raf.current_claim;
        std::vector<Fr> ch2;
        if (emit) std::printf("S 2\n");
        for (size_t k = 0; k < pc.log_k; k++) {
            auto ev = raf.computeRoundPolynomialCubic();
            const Fr c = tr.challengeScalar("raf_round");
            ch2.push_back(c);
            line(emit, "P", {ev[0], ev[2]});
            raf.updateClaim(ev, c);
            raf.bindChallenge(c);
        }
        line(emit, "H", ch2);
        line(emit, "C", {claim2, raf.getFinalClaim()});
    }
    tl.lap("stage 2: RAF initial claim + cubic rounds (log K)", "h2d+kernels+host");
    // ---- stage 3: Lasso (prover.zig:562-700)
    (void)tr.challengeScalar("lasso_gamma");
    std::vector<Fr> r_red(pc.log_t);
    for (auto &x : r_red) x = tr.challengeScalar("r_reduction");
    {
        LassoProver lp(pc.lookup_indices, pc.log_t, 16, r_red);
        std::vector<Fr> ch3;
        Fr init3 = Fr::zero();
        if (emit) std::printf("S 3\n");
        for (size_t k = 0; k < 16 + pc.log_t; k++) {
            UniPoly rp = lp.computeRoundPolynomial();
            if (k == 0) init3 = lp.current_claim;
            line(emit, "P", rp.coeffs);
            const Fr c = tr.challengeScalar("lasso_round");
            ch3.push_back(c);
            lp.receiveChallenge(c);
        }
        line(emit, "H", ch3);
        line(emit, "C", {init3, lp.getFinalEval()});
    }
    tl.lap("stage 3: LassoProver (16 address + log T cycle rounds, index upload)", "h2d+kernels+host");
    // ---- stages 4, 5, 6
    Stage4Result r4 = proveStage4(pc.accesses, {}, T, pc.log_k, pc.log_t, pc.start_address, tr);
    tl.lap("stage 4: Val evaluation (inc / wa / lt tables on the device, log T cubic rounds)", "h2d+kernels+host");
    std::vector<uint8_t> rd(pc.trace.size());  // the destination register of every cycle: all the stage reads of the trace
    for (size_t i = 0; i < rd.size(); i++) rd[i] = (uint8_t)((pc.trace[i].instruction >> 7) & 31);
    StageRoundsResult r5 = proveStage5(rd.data(), rd.size(), pc.log_t, tr);
    tl.lap("stage 5: register evaluation", "h2d+kernels+host");
    StageRoundsResult r6 = proveStage6(T, tr);
    tl.lap("stage 6: booleanity", "h2d+kernels+host");
    if (emit) {
        std::printf("S 4\n");
        for (const auto &rp : r4.round_polys) line(true, "P", {rp[0], rp[1], rp[2], rp[3]});
        line(true, "H", r4.challenges);
        line(true, "C", {r4.initial_claim, r4.final_claim});
        int k = 5;
        for (const StageRoundsResult *r : {&r5, &r6}) {
            std::printf("S %d\n", k++);
            for (const auto &rp : r->round_polys) line(true, "P", {rp[0], rp[1]});
            line(true, "H", r->challenges);
            line(true, "C", {r->initial_claim, r->final_claim});
        }
    }
    // ---- the opening the verifier's check consumes: HyperKZG.open of the register polynomial at a transcript point. The polynomial is
    // widened on the device from its machine words (8 bytes per evaluation cross PCIe), the fold / quotient / commit loop stays resident.
    size_t reg_vars = 0;
    while ((size_t(1) << reg_vars) < reg.size()) reg_vars++;
    std::vector<Fr> point;
    for (size_t i = 0; i < reg_vars; i++) point.push_back(tr.challengeScalar("opening_point"));
    DeviceMem d_reg(reg.size() * 32);
    const zg_col_t reg_col{ZG_COL_U64, 0, 0, reg.data(), nullptr};
    check(zg_fr_rows_from_columns(&reg_col, 1, reg.size(), d_reg.u64()), "zg_fr_rows_from_columns");
    std::vector<uint64_t> q(8 * (reg_vars ? reg_vars : 1));
    std::vector<uint8_t> qi(reg_vars ? reg_vars : 1);
    Fr value = Fr::zero(), final_eval;
    const size_t n_open = reg.size() < HyperKZG::srsLen(pk) ? reg.size() : HyperKZG::srsLen(pk);
    check(zg_hyperkzg_open_dev(pk.device->handle(), d_reg.u64(), n_open, reinterpret_cast<const uint64_t *>(point.data()), reg_vars, value.limbs, nullptr, q.data(),
                               qi.data(), final_eval.limbs), "zg_hyperkzg_open_dev");
    if (emit) {
        std::printf("O");
        print_fr(final_eval);
        std::printf("\n");
    }
    tl.lap("HyperKZG.open of the register polynomial (log T quotient commitments)", "h2d+kernels");
    return tl;
}

int main(int argc, char **argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: bench_prove_path synth <log_t> [reps [key_uses [key_table_levels]]] | file <case> [reps [key_uses [key_table_levels]]]\n"); return 2; }
    const bool from_file = !std::strcmp(argv[1], "file");
    const int reps = argc > 3 ? atoi(argv[3]) : 3;
    key_uses = argc > 4 ? atoi(argv[4]) : 0;
    key_levels = argc > 5 ? atoi(argv[5]) : 0;
    int rc = 0;
    try {
        auto t_init = clk::now();
        check(zg_init(0), "zg_init");
        void *warm = nullptr;
        check(zg_dev_alloc(1 << 20, &warm), "zg_dev_alloc");
        check(zg_dev_free(warm), "zg_dev_free");
        const double init_ms = std::chrono::duration<double>(clk::now() - t_init).count() * 1e3;
        ProveCase pc = from_file ? file_case(argv[2]) : synthetic_case((size_t)atoi(argv[2]));
        // repetition -1 warms the library's pools (device pool, pinned slab, session pools, streams) and, in file mode, prints the records
        std::vector<Timeline> runs;
        for (int rep = -1; rep < reps; rep++) {
            Timeline tl = prove_once(pc, from_file && rep == -1);
            if (rep >= 0) runs.push_back(tl);
            else runs.insert(runs.begin(), tl);  // runs[0] = the cold proof
        }
        const Timeline &cold = runs[0];
        const size_t ns = cold.steps.size();
        std::vector<double> avg(ns, 0.0);
        for (size_t r = 1; r < runs.size(); r++)
            for (size_t i = 0; i < ns; i++) avg[i] += runs[r].steps[i].ms / (double)(runs.size() - 1);
        double total = 0, total_cold = 0, total_wo_key = 0;
        for (size_t i = 0; i < ns; i++) {
            total += avg[i];
            total_cold += cold.steps[i].ms;
            if (cold.steps[i].kind != "once per key") total_wo_key += avg[i];
        }
        std::vector<size_t> order(ns);
        for (size_t i = 0; i < ns; i++) order[i] = i;
        std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return avg[a] > avg[b]; });
        std::printf("{\"prove_path\": {\"log_t\": %zu, \"cycles\": %zu, \"memory_accesses\": %zu, \"lookups\": %zu, \"srs_size\": %zu, \"reps\": %d, \"key_expected_uses\": %d, \"key_table_levels\": %d, "
                    "\"library_init_ms\": %.3f, \"total_ms\": %.3f, \"total_ms_without_proving_key\": %.3f, \"first_proof_ms_cold_pools\": %.3f, \"steps\": [",
                    pc.log_t, pc.trace.size(), pc.accesses.size(), pc.lookup_indices.size(), pc.srs_size, reps, key_uses, key_levels, init_ms, total, total_wo_key, total_cold);
        for (size_t i = 0; i < ns; i++)
            std::printf("%s{\"call\": \"%s\", \"what\": \"%s\", \"ms\": %.4f, \"ms_cold\": %.4f, \"share\": %.4f}", i ? ", " : "", cold.steps[i].name.c_str(),
                        cold.steps[i].kind.c_str(), avg[i], cold.steps[i].ms, total > 0 ? avg[i] / total : 0.0);
        std::printf("], \"top3\": [");
        for (size_t k = 0; k < 3 && k < ns; k++) std::printf("%s{\"call\": \"%s\", \"ms\": %.4f}", k ? ", " : "", cold.steps[order[k]].name.c_str(), avg[order[k]]);
        std::printf("], \"note\": \"one sequence per repetition, host-timed per call (every call returns a host value or synchronises); what: which of "
                    "host work / PCIe copies / kernels the call contains; repetition -1 (cold pools) is reported beside the average of the others\"}}\n");
    } catch (const std::exception &e) {
        std::printf("{\"error\": \"%s\"}\n", e.what());
        rc = 3;
    }
    zg_shutdown();
    return rc;
}
