// stage3_round_split.cpp — where a batched Stage-3 round waits: host clock around each instance's round read and bind
// (profiles/r3final_stage3_round_split.txt). Build from the repo root, run on the GPU box:
//   g++ -O2 -std=c++17 -Iinclude -I/opt/rocm/include tools/stage3_round_split.cpp -o tools/stage3_round_split \
//       -Lzolt_amd -lzolt_gpu -Wl,-rpath,'$ORIGIN/../zolt_amd' && tools/stage3_round_split 20
// The instruction-input session is a private member of zolt::Stage3Prover; this measuring tool opens the class up instead of widening
// the mirror's interface.
#define private public
#include "../zolt_amd/host/zolt_host.hpp"
#undef private
#include <chrono>
#include <cstdio>
using namespace zolt;
using clk = std::chrono::steady_clock;
static uint64_t sm_s = 7;
static uint64_t splitmix() { sm_s += 0x9E3779B97F4A7C15ull; uint64_t z = sm_s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
int main(int argc, char **argv) {
    size_t lt = argc > 1 ? atoi(argv[1]) : 13, T3 = size_t(1) << lt;
    check(zg_init(0), "init");
    std::vector<Fr> w(T3 * 43);
    for (auto &x : w) x = Fr::fromU64(splitmix());
    DeviceMem d_rows(T3 * 43 * 32);
    check(zg_memcpy_h2d(d_rows.p, w.data(), T3 * 43 * 32), "h2d");
    std::vector<Fr> ro(lt), rp(lt), sg(5);
    for (auto &x : ro) x = Fr::fromU64(splitmix());
    for (auto &x : rp) x = Fr::fromU64(splitmix());
    Fr g = Fr::fromU64(splitmix());
    sg[0] = Fr::one();
    for (int i = 1; i < 5; i++) sg[i] = sg[i - 1].mul(g);
    std::array<Fr, 3> cl = {Fr::fromU64(splitmix()), Fr::fromU64(splitmix()), Fr::fromU64(splitmix())}, co = cl;
    for (int rep = 0; rep < 3; rep++) {
        Stage3Prover p(d_rows.u64(), ro, rp, sg, g, sg[2], cl, co);
        Transcript tr("Jolt");
        double t[8] = {};
        auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        for (size_t rd = 0; rd < lt; rd++) {
            auto a0 = clk::now();
            auto s = p.shift.computeRoundEvals(p.claims[0]);
            auto a1 = clk::now();
            auto iv = p.instr_->roundExpr(p.instr_terms_);
            auto a2 = clk::now();
            auto r = p.reg.computeRoundEvals(p.claims[2]);
            auto a3 = clk::now();
            Fr ch = Fr::fromU64(splitmix());
            p.shift.bind(ch);
            auto a4 = clk::now();
            p.instr_->bind(ch);
            auto a5 = clk::now();
            p.reg.bind(ch);
            auto a6 = clk::now();
            t[0] += us(a0, a1); t[1] += us(a1, a2); t[2] += us(a2, a3); t[3] += us(a3, a4); t[4] += us(a4, a5); t[5] += us(a5, a6);
            if (rep == 2) printf("round %zu: shift %.0f instr %.0f reg %.0f | bind shift %.0f instr %.0f reg %.0f us\n", rd, us(a0, a1), us(a1, a2), us(a2, a3), us(a3, a4), us(a4, a5), us(a5, a6));
        }
        printf("rep %d totals: shift %.0f instr %.0f reg %.0f | bind shift %.0f instr %.0f reg %.0f us\n", rep, t[0], t[1], t[2], t[3], t[4], t[5]);
    }
    return 0;
}
