// poly.hip — Fr evaluation-table kernels on gfx950: eq-table build, Spartan combine,
// sumcheck folds with fused next-round sums, and the device-resident sumcheck session.
//
// Reference functions replaced (paths under /root/reference):
//   EqPolynomial.evalsSliceWithScaling   src/poly/mod.zig:252-290
//   GruenSplitEqPolynomial tables        src/poly/split_eq.zig:122-171 (same values)
//   DensePolynomial.bindFirst / bindLow  src/poly/mod.zig:128-149 / :160-175
//   Sumcheck.Prover.nextRound sums       src/subprotocols/mod.zig:79-93
//   LowToHigh round sums / fold          src/zkvm/r1cs/jolt_r1cs.zig:436-444,470-477
//   Spartan combine                      src/zkvm/spartan/mod.zig:191-199
// All of these are exact modular arithmetic with canonical outputs, so any evaluation order
// gives the reference's bytes (sums of field elements commute; eq(r,x) is one field value
// however its factors are grouped).
#include <mutex>
#include <vector>

#include "common.hip.h"
#include "field.hip.h"
#include "fp29.hip.h"
#include "sc_common.hip.h"

namespace zg {

// ------------------------------------------------------------------ eq table
// A factor product of eq(r, .) for one index (r[0] <-> MSB of the index). Used where a launch is pure dependency latency (the
// prefix-table set of GruenSplitEqPolynomial): FOUR adjacent lanes share one output — lane q multiplies the factors
// j = q (mod 4), then two shuffle steps combine the four partial products: a chain of ceil(v/4) + 2 products instead of v.
ZG_DEV Fr eq_factor_product4(const uint64_t *r, int v, uint32_t idx, uint32_t q, const Fr *init) {
    Fr one = Fr::one();
    Fr c = (q == 0 && init) ? *init : one;
    bool used = q == 0 && init;
    for (int j = (int)q; j < v; j += 4) {
        Fr rj = fe_load<FrParams>(r + 4 * j);
        Fr f = ((idx >> (v - 1 - j)) & 1u) ? rj : fe_sub(one, rj);
        c = used ? fr_mul29v(c, f) : f;
        used = true;
    }
    // lanes q and q+1, then q and q+2 (all four lanes of a group execute the same products; lane 0 holds the result)
    Fr o = fr_shfl_down(c, 1);
    c = fr_mul29v(c, o);
    o = fr_shfl_down(c, 2);
    return fr_mul29v(c, o);
}

// GruenSplitEqPolynomial's prefix tables: entry idx of table k = prod_{j<k} (bit_j(idx) ? tau[j] : 1 - tau[j]) at element
// 2^k - 1 + idx of out. Element p-1 (p = 1 .. 2^(v+1)-1) therefore has k = floor(log2 p), idx = p - 2^k: one launch, four
// lanes per output as above (the chain is at most ceil(v/4) + 2 products; Montgomery products are exact, so the product
// order does not change the value the reference's level-by-level build produces).
__global__ void __launch_bounds__(256) eq_prefix_kernel(const uint64_t *tau, int v, uint64_t *out) {
    uint32_t q = threadIdx.x & 3;
    uint64_t p = (uint64_t)blockIdx.x * 64 + (threadIdx.x >> 2) + 1;
    bool live = p < (2ull << v);
    uint32_t pp = live ? (uint32_t)p : 1u;
    int k = 31 - __clz(pp);
    Fr e = eq_factor_product4(tau, k, pp - (1u << k), q, nullptr);
    if (live && q == 0) fe_store(out + 4 * (size_t)(p - 1), e);
}

// ---- eq table, one launch. The challenge vector travels as a KERNEL ARGUMENT (<= 34 x 32 bytes): no H2D copy, no staging
// buffer, no factor-table launch in front — building a 2^20-entry table was 8 us of factor tables + 17 us of stream behind a
// pageable copy; now the call is one asynchronous launch.
struct EqArgs {
    uint32_t r[34][8];  // r[0] <-> MSB of the index
    uint32_t scale[8];
    int v, has_scale;
};
static constexpr int EQ_MAX_ROWS = 256;  // rows of 2^v_lo entries per workgroup
struct EqShared {
    uint4 small[9][16][2];        // factor tables of 4 variables each: groups 0, 1 = the low 8 index bits, 2.. = the row index h
    uint4 hi_row[EQ_MAX_ROWS][3];  // hi[h0 + k] as nine 29-bit limbs (the form every product of the main loop needs: unpacked once per row, not per entry)
};

// Factor tables in LDS, then thread t's lo[t] (returned, canonical) and the block's hi[h0 .. h0 + rows) in sh.hi_row.
//   lo[t]  = prod_{j < v_lo} (bit_j(t) ? r[v_hi + j] : 1 - r[v_hi + j]),  bit_j(t) = bit (v_lo - 1 - j) of t
//   hi[h]  = scale * prod_{j < v_hi} (bit_j(h) ? r[j] : 1 - r[j]),          bit_j(h) = bit (v_hi - 1 - j) of h
// Index bits are taken four at a time: 16-entry tables of <= 3 products each (144 threads at most), then lo[t] is one product
// of two table entries and hi[h] at most ceil(v_hi / 4) products by one thread per row — a chain of ~10 products (~2 us)
// in front of the stream instead of a launch of its own. Exact products: any grouping gives the reference's bytes.
// lo_var0: the variable the MOST significant of the v_lo low index bits belongs to (v_hi when the two parts are adjacent)
ZG_DEV Fr eq_block_factors(const EqArgs &a, int v_lo, int v_hi, uint32_t h0, uint32_t rows, EqShared &sh, int lo_var0 = -1) {
    if (lo_var0 < 0) lo_var0 = v_hi;
    const uint32_t tid = threadIdx.x;
    const int ng_hi = (v_hi + 3) / 4, ng = 2 + ng_hi;
    const Fr one = Fr::one();
    if (tid < 16u * (uint32_t)ng) {
        const int g = (int)(tid >> 4);
        const uint32_t e = tid & 15u;
        // group g covers index bits [b0, b0 + nb) of the low part (g < 2) or of h (g >= 2); bit b <-> variable var(b)
        const int part_bits = g < 2 ? v_lo : v_hi, b0 = g < 2 ? 4 * g : 4 * (g - 2);
        // the group's <= 4 factors as a balanced tree, (f0 f1)(f2 f3): two independent products, then one — a chain of two instead of three.
        // A group with fewer than four variables takes the Montgomery one for the missing factors (x * 1 is x, exactly): every index below
        // is a compile-time constant, so the four factors live in registers. (Round 4 filled `f[nf++]` in a loop with a break: a dynamically
        // indexed local array, 144 bytes of SCRATCH per lane — for 144 threads of a prologue, but a kernel that uses scratch at all pays
        // for its set-up in every wave of every launch.)
        Fr f0 = one, f1 = one, f2 = one, f3 = one;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int bit = b0 + b;
            Fr fb = one;
            if (bit < part_bits) {
                const int var = g < 2 ? lo_var0 + (v_lo - 1 - bit) : (v_hi - 1 - bit);
                Fr rj;
#pragma unroll
                for (int i = 0; i < 8; i++) rj.l[i] = a.r[var][i];
                fb = ((e >> b) & 1u) ? rj : fe_sub(one, rj);
            }
            if (b == 0) f0 = fb;
            else if (b == 1) f1 = fb;
            else if (b == 2) f2 = fb;
            else f3 = fb;
        }
        const Fr acc = fr_mul29v(fr_mul29v(f0, f1), fr_mul29v(f2, f3));
        fe_store(&sh.small[g][e][0], acc);
    }
    __syncthreads();
    Fr lo = fe_load<FrParams>(&sh.small[0][tid & 15u][0]);
    if (v_lo > 4) lo = fr_mul29v(lo, fe_load<FrParams>(&sh.small[1][(tid >> 4) & 15u][0]));
    if (tid < rows) {
        const uint32_t h = h0 + tid;
        Fr hv = one;
        bool used = false;
        if (a.has_scale) {
#pragma unroll
            for (int i = 0; i < 8; i++) hv.l[i] = a.scale[i];
            used = true;
        }
        for (int g = 0; g < ng_hi; g++) {
            Fr f = fe_load<FrParams>(&sh.small[2 + g][(h >> (4 * g)) & 15u][0]);
            hv = used ? fr_mul29v(hv, f) : f;
            used = true;
        }
        {
            const F29 hu = f29_unpack(hv.l);
            sh.hi_row[tid][0] = make_uint4(hu.l[0], hu.l[1], hu.l[2], hu.l[3]);
            sh.hi_row[tid][1] = make_uint4(hu.l[4], hu.l[5], hu.l[6], hu.l[7]);
            sh.hi_row[tid][2] = make_uint4(hu.l[8], 0u, 0u, 0u);
        }
    }
    __syncthreads();
    return lo;
}

ZG_DEV F29 eq_hi_row(const EqShared &sh, uint32_t k) {
    const uint4 a = sh.hi_row[k][0], b = sh.hi_row[k][1];
    F29 r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = sh.hi_row[k][2].x;
    return r;
}
// out[(h << v_lo) | t] = hi[h] * lo[t]; one 32-byte store per thread and row, 8 KiB contiguous per (block, h): the
// kernel is an HBM write stream with one mixed-format product (fr_mul29) per element.
// WG groups of 256 threads per block: thread (grp, lo) writes rows grp, grp + WG, ... — the factor prologue (a latency chain of ~5
// products) is paid once per block, and a block of 1024 threads keeps four waves per SIMD in the product stream behind it.
template <int WG>
__global__ void __launch_bounds__(256 * WG) eq_main_kernel(EqArgs a, int v_lo, int v_hi, uint32_t hi_per_block, uint64_t *out) {
    __shared__ EqShared sh;
    const uint32_t n_hi = 1u << v_hi, h0 = blockIdx.x * hi_per_block;
    const uint32_t rows = n_hi - h0 < hi_per_block ? n_hi - h0 : hi_per_block;
    Fr lov = eq_block_factors(a, v_lo, v_hi, h0, rows, sh);
    const uint32_t lo = threadIdx.x & 255u, grp = threadIdx.x >> 8;
    if (lo >= (1u << v_lo)) return;
    F29 tp = fr29_prescale(lov);  // shared factor of this thread's products
    // two rows per trip: the two products are independent chains of 162 multiply-adds each, so a wave has an instruction to issue while
    // the other chain's accumulator is in flight (two waves per SIMD alone do not cover the multiplier's latency)
    uint32_t k = grp;
    for (; k + WG < rows; k += 2 * WG) {
        const F29 t0 = f29t_mul<Fr29>(eq_hi_row(sh, k), tp), t1 = f29t_mul<Fr29>(eq_hi_row(sh, k + WG), tp);
        fe_store(out + 4 * (((size_t)(h0 + k) << v_lo) | lo), fr29_out(t0));
        fe_store(out + 4 * (((size_t)(h0 + k + WG) << v_lo) | lo), fr29_out(t1));
    }
    if (k < rows) fe_store(out + 4 * (((size_t)(h0 + k) << v_lo) | lo), fr29_out(f29t_mul<Fr29>(eq_hi_row(sh, k), tp)));
}

// The same table when the three variables in the MIDDLE of the index are 128-bit challenges (the reference's transcript challenges are:
// MontU128Challenge, stored [0, 0, lo, hi] — fp29.hip.h FrMul): index = (h << 11) | (e << 8) | t with h <-> r[0 .. v_hi), e <-> the three
// variables r[v_hi .. v_hi + 3), t <-> the last eight. A thread forms B = hi[h] * lo[t] (one full product) and EXPANDS it over the three
// variables exactly as the reference builds its table (evalsSliceWithScaling, src/poly/mod.zig:252-290: y = x * r_j, x = x - y): seven
// products by a 128-bit factor (9 x 5 limbs, 90 multiply-adds instead of 162) and seven subtractions for eight entries. 2^20 entries:
// 5.4 M -> 3.8 M wave instructions (the kernel is bound by its instruction count, profiles/r4final_kernel_table.md). For a fixed e the
// lanes of a wave write consecutive t: every store instruction is a contiguous 2 KiB.
constexpr int EQ_XL = 3;
template <int WG>
__global__ void __launch_bounds__(256 * WG) eq_expand_kernel(EqArgs a, int v_hi, uint32_t hi_per_block, uint64_t *out) {
    __shared__ EqShared sh;
    const uint32_t n_hi = 1u << v_hi, h0 = blockIdx.x * hi_per_block;
    const uint32_t rows = n_hi - h0 < hi_per_block ? n_hi - h0 : hi_per_block;
    const Fr lov = eq_block_factors(a, 8, v_hi, h0, rows, sh, v_hi + EQ_XL);
    const uint32_t t = threadIdx.x & 255u, grp = threadIdx.x >> 8;
    F29 rp[EQ_XL];  // H * 2^17 of the three expansion challenges (frmul_prepare's narrow form), opaque to the optimiser
#pragma unroll
    for (int j = 0; j < EQ_XL; j++) {
        const uint32_t *y = a.r[v_hi + j];
        u32 w[8] = {y[4] << 17, (y[5] << 17) | (y[4] >> 15), (y[6] << 17) | (y[5] >> 15), (y[7] << 17) | (y[6] >> 15), y[7] >> 15, 0u, 0u, 0u};
        rp[j] = f29_unpack(w);
#pragma unroll
        for (int i = 0; i < 9; i++) asm volatile("" : "+v"(rp[j].l[i]));
    }
    const F29 tp = fr29_prescale(lov);
    for (uint32_t k = grp; k < rows; k += WG) {
        Fr v[8];
        v[0] = fr29_out(f29t_mul<Fr29>(eq_hi_row(sh, k), tp));
#pragma unroll
        for (int j = 0; j < EQ_XL; j++) {
            const int stride = 4 >> j;  // variable v_hi + j <-> bit (2 - j) of e
#pragma unroll
            for (int base = 0; base < 8; base += 2 * stride) {
                const Fr y = fr29_out(f29t_mul_short<Fr29, 5>(f29_unpack(v[base].l), rp[j]));
                v[base + stride] = y;
                v[base] = fe_sub(v[base], y);
            }
        }
        uint64_t *row = out + 4 * (((size_t)(h0 + k) << (8 + EQ_XL)) | t);
#pragma unroll
        for (int e = 0; e < 8; e++) fe_store(row + 4 * ((size_t)e << 8), v[e]);
    }
}

// f[i] = eq[i] * (Az[i]*Bz[i] - Cz[i])
__global__ void __launch_bounds__(256) spartan_combine_kernel(const uint64_t *eq, const uint64_t *az, const uint64_t *bz,
                                                              const uint64_t *cz, size_t n, uint64_t *out) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        Fr a = fe_load<FrParams>(az + 4 * i), b = fe_load<FrParams>(bz + 4 * i), c = fe_load<FrParams>(cz + 4 * i);
        Fr e = fe_load<FrParams>(eq + 4 * i);
        fe_store(out + 4 * i, fr_mul29v(e, fe_sub(fr_mul29v(a, b), c)));
    }
}

// dot product sum_i a[i]*b[i] over Fr: per-block partials, finished by sc_finish_kernel's first slot
__global__ void __launch_bounds__(256) fr_dot_kernel(const uint64_t *a, const uint64_t *b, size_t n, uint64_t *partials);

// ------------------------------------------------------------------ sums / folds
__global__ void __launch_bounds__(256) fr_dot_kernel(const uint64_t *a, const uint64_t *b, size_t n, uint64_t *partials) {
    __shared__ uint4 sh[256 * 4];
    Fr g0 = Fr::zero(), g1 = Fr::zero();
    size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        g0 = fe_add(g0, fr_mul29v(fe_load<FrParams>(a + 4 * i), fe_load<FrParams>(b + 4 * i)));
    block_sum_pair(g0, g1, sh);
    if (threadIdx.x == 0) {
        fe_store(partials + 8 * (size_t)blockIdx.x, g0);
        fe_store(partials + 8 * (size_t)blockIdx.x + 4, g1);
    }
}

// round sums of a table: HIGH: g0 = sum t[0..h), g1 = sum t[h..2h);  LOW: g0 = sum t[2i], g1 = sum t[2i+1]
// (publication of a round's pair to the pinned host mailbox: publish_seq, sc_common.hip.h)

// ---- runSumcheck resident on the device (src/subprotocols/mod.zig:302-354): the toy verifier (:165-243) runs on the device (until
// round 5 as the last step of the kernel that produced a round's sums, since then as the first step of the next one), so no round needs
// the host. `res` (u64 words):
//   [0,4) claim | [4, 4+8v) rounds: c0 || c1 | [.., +4v) challenges | +4 final_eval | +1 status | +4 running claim before an even round |
//   +4 running claim before an odd round.   status: bit 0 = result (verifier.claim == final_eval), bits 8.. = 1 + first failed round
// Round 5: the verifier step of round k no longer closes the kernel that produced round k's sums (block pairs -> arrival counter ->
// the last block adds them up -> verifier step: three dependent round trips, ~4 us at the end of every launch) — it OPENS the next
// kernel: a producer leaves its block pairs in `partials` with plain stores and ends; every block of the consumer adds the nb_prev pairs
// up for itself (32 KiB from L2, under the latency of its own first table loads), runs the same verifier step and gets the same
// challenge; block 0 alone records it in `res`. The pairs alternate between two halves of the array and the running claim between two
// slots (a consumer's early blocks write while its late blocks still read).
struct ScRunArg {
    uint64_t *res;  // nullptr: an ordinary session launch (sums go to `sums`, sequence word published)
    uint32_t v, round, init;
    uint32_t nb_prev = 0;  // device-resident protocol: workgroups of the launch that produced the sums this launch's verifier step reads
                           // (0: this launch starts the protocol — it only produces); `round` = the round of THOSE sums
};
constexpr uint32_t SC_RUN_PAIRS = 1024;  // block pairs per half of the partials array in the device-resident protocol
ZG_DEV size_t run_off_chal(uint32_t v) { return 4 + 8 * (size_t)v; }
ZG_DEV size_t run_off_final(uint32_t v) { return 4 + 12 * (size_t)v; }
ZG_DEV size_t run_off_status(uint32_t v) { return 8 + 12 * (size_t)v; }
ZG_DEV size_t run_off_claim(uint32_t v) { return 9 + 12 * (size_t)v; }
ZG_DEV size_t run_off_cur(uint32_t v) { return 13 + 12 * (size_t)v; }
// the running claim BEFORE round `round`'s verifier step: two slots in turn (the second one is the word block that held the running
// challenge until round 5 — every consumer now derives the challenge itself)
ZG_DEV size_t run_off_claim_of(uint32_t v, uint32_t round) { return (round & 1u) ? run_off_cur(v) : run_off_claim(v); }
ZG_DEV uint64_t fr_limb64(const Fr &a, int i) { return (uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32); }
ZG_DEV bool fr_eq(const Fr &a, const Fr &b) {
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d |= a.l[i] ^ b.l[i];
    return d == 0;
}

// Verifier.verifyRound for the round polynomial [g0, g1 - g0]: check, derive the challenge, update the claim.
// One thread. With `init` the claim is first set to g0 + g1 (runSumcheck's initial sum over the hypercube).
// `claim_reg` (the LDS tail): the running claim is taken from and left in the caller's registers instead of a dependent global load.
// `defer_claim`: the claim update g0 + c1 * ch is left to the caller (the single-wave tail folds it with the table: same formula).
// `record`: this caller writes the round into `res` (one workgroup per launch does; the others only need the challenge and the claim).
ZG_DEV Fr sc_verifier_step(const ScRunArg &a, const Fr &g0, const Fr &g1, F29 *ch_pre = nullptr, Fr *claim_reg = nullptr, bool defer_claim = false,
                           bool record = true) {
    uint64_t *res = a.res;
    Fr sum = fe_add(g0, g1);
    Fr claim;
    if (a.init) {
        claim = sum;
        if (record) {
            fe_store(res, claim);
            res[run_off_status(a.v)] = 0;
        }
    } else {
        claim = claim_reg ? *claim_reg : fe_load<FrParams>(res + run_off_claim_of(a.v, a.round));
    }
    Fr c1 = fe_sub(g1, g0);
    if (record && !fr_eq(sum, claim) && (a.init || (res[run_off_status(a.v)] >> 8) == 0)) res[run_off_status(a.v)] = ((uint64_t)(a.round + 1)) << 8;
    uint64_t h = 0x9e3779b97f4a7c15ull;
    h ^= (uint64_t)a.round;
    h *= 0xff51afd7ed558ccdull;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        h ^= fr_limb64(claim, i);
        h *= 0xc4ceb9fe1a85ec53ull;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        h ^= i < 4 ? fr_limb64(g0, i) : fr_limb64(c1, i - 4);
        h *= 0xff51afd7ed558ccdull;
        h ^= h >> 33;
    }
    h ^= h >> 33;
    h *= 0xff51afd7ed558ccdull;
    h ^= h >> 33;
    Fr ch = fr_from_u64_29(h);                   // F.fromU64
    F29 chp = fr29_prescale(ch);                 // also the shared factor of the fold that follows
    if (ch_pre) *ch_pre = chp;
    if (record) {
        fe_store(res + 4 + 8 * (size_t)a.round, g0);
        fe_store(res + 8 + 8 * (size_t)a.round, c1);
        fe_store(res + run_off_chal(a.v) + 4 * (size_t)a.round, ch);
    }
    if (!defer_claim) {
        Fr next = fe_add(fr_mul29(c1, chp), g0);  // UniPoly.evaluate by Horner: c1 * x + c0
        if (record) fe_store(res + run_off_claim_of(a.v, a.round + 1), next);
        if (claim_reg) *claim_reg = next;
    }
    return ch;
}

// The opening of a consumer launch of the device-resident protocol (see ScRunArg): every thread of every workgroup calls it. Adds the
// producer's nb_prev block pairs up, runs round run.round's verifier step in one lane and returns the prescaled challenge to every thread
// (through `sh`, which is free again on return); *claim_out (if given) receives the claim after the step in EVERY thread.
ZG_DEV F29 sc_run_open(const ScRunArg &run, const uint64_t *partials, u32 *sh, Fr *claim_out = nullptr) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint64_t *src = partials + 8 * (size_t)SC_RUN_PAIRS * (run.round & 1u);
    Acc9 a0 = acc9_zero(), a1 = acc9_zero();
    for (uint32_t k = tid; k < run.nb_prev; k += blockDim.x) {
        acc9_add(a0, fe_load<FrParams>(src + 8 * (size_t)k));
        acc9_add(a1, fe_load<FrParams>(src + 8 * (size_t)k + 4));
    }
    const Fr tot = block_sum_pair9(a0, a1, sh);  // wave 0: lane SC_LANE_G0 / SC_LANE_G1
    __syncthreads();                             // the reduction's reads of `sh` are done: it now carries the challenge and the claim
    if (tid < 64) {
        const Fr second = pair_second_to_first(tot);
        if (lane == SC_LANE_G0) {
            F29 cp;
            Fr claim = Fr::zero();
            Fr *creg = nullptr;
            if (!run.init) {
                claim = fe_load<FrParams>(run.res + run_off_claim_of(run.v, run.round));
                creg = &claim;
            }
            ScRunArg a = run;
            (void)sc_verifier_step(a, tot, second, &cp, run.init ? &claim : creg, false, blockIdx.x == 0);
#pragma unroll
            for (int i = 0; i < 9; i++) sh[i] = cp.l[i];
#pragma unroll
            for (int i = 0; i < 8; i++) sh[9 + i] = claim.l[i];
        }
    }
    __syncthreads();
    F29 rp;
#pragma unroll
    for (int i = 0; i < 9; i++) rp.l[i] = sh[i];
    if (claim_out) {
#pragma unroll
        for (int i = 0; i < 8; i++) claim_out->l[i] = sh[9 + i];
    }
    __syncthreads();  // `sh` may be reused by the caller's own reduction
    return rp;
}

// End of a round inside the producing kernel (no second launch): every thread hands in its lazy sums; the block's canonical
// pair goes to `partials` (write-through stores by the two lanes of wave 0 that hold the totals, drained, then one relaxed arrival:
// sc_common.hip.h), and the block that arrives last adds all pairs up (sc1 loads), writes the round's pair to `sums` (the pinned
// host mailbox) and publishes the sequence word — or runs the device-resident protocol's verifier step. `sh`: SC_RED_WORDS u32.
__device__ __forceinline__ void finish_round(const Acc9 &g0, const Acc9 &g1, u32 *sh, uint64_t *partials, uint64_t *sums, uint32_t *counter,
                                             uint64_t *flag, uint64_t seq, const ScRunArg &run) {
    const uint32_t tid = threadIdx.x, nb = gridDim.x, lane = tid & 63u;
    Fr tot = block_sum_pair9(g0, g1, sh);  // wave 0: lane SC_LANE_G0 / SC_LANE_G1
    if (run.res) {  // device-resident protocol: the block pair stays in `partials` (plain stores, complete when the launch is); the
                    // next launch's workgroups add the pairs up and run the verifier step (sc_run_open)
        const uint32_t out_round = run.nb_prev ? run.round + 1 : run.round;  // the round these sums belong to
        if (tid < 64 && (lane == SC_LANE_G0 || lane == SC_LANE_G1))
            fe_store(partials + 8 * ((size_t)SC_RUN_PAIRS * (out_round & 1u) + blockIdx.x) + (lane == SC_LANE_G1 ? 4 : 0), tot);
        return;
    }
    if (nb > 1) {
        __shared__ uint32_t last;
        if (tid < 64) {
            if (lane == SC_LANE_G0 || lane == SC_LANE_G1) sc1_store_fr(partials + 8 * (size_t)blockIdx.x + (lane == SC_LANE_G1 ? 4 : 0), tot);
            sc_drain_stores();  // the one storing wave, before its lane signals
            if (lane == SC_LANE_G0) last = sc_arrive(counter, nb) ? 1u : 0u;
        }
        __syncthreads();  // (also orders the reduction's LDS reads before its reuse below)
        if (!last) return;
        Acc9 a0 = acc9_zero(), a1 = acc9_zero();
        for (uint32_t k = tid; k < nb; k += blockDim.x) {
            const uint64_t *src = partials + 8 * (size_t)k;
            acc9_add(a0, sc1_load_fr(src));
            acc9_add(a1, sc1_load_fr(src + 4));
        }
        tot = block_sum_pair9(a0, a1, sh);
    }
    if (tid < 64) {
        Fr second = pair_second_to_first(tot);
        if (lane == SC_LANE_G0) {
            mailbox_store_fr(sums, tot, flag);
            mailbox_store_fr(sums + 4, second, flag);
            publish_seq(flag, seq);
        }
    }
}

// the block's canonical pair to partials[blockIdx] with plain stores (finished by a later launch: sc_finish_kernel)
__device__ __forceinline__ void store_block_pair(const Acc9 &g0, const Acc9 &g1, u32 *sh, uint64_t *partials) {
    Fr tot = block_sum_pair9(g0, g1, sh);
    const uint32_t lane = threadIdx.x & 63u;
    if (threadIdx.x < 64 && (lane == SC_LANE_G0 || lane == SC_LANE_G1))
        fe_store(partials + 8 * (size_t)blockIdx.x + (lane == SC_LANE_G1 ? 4 : 0), tot);
}

// ---- zg_selftest_handoff: the hand-off of finish_round under the conditions that expose a wrong one (MI355X_MICROARCH.md: "test every
// hand-off under UNEVEN load, consumer L1-warm, checking every word"). Same primitives, same lanes, same order as finish_round.
ZG_DEV uint32_t handoff_word(uint32_t epoch, uint32_t block, uint32_t w) { return (epoch * 0x9e3779b1u) ^ (block * 0x85ebca6bu) ^ (w * 0xc2b2ae35u) ^ 0x27d4eb2fu; }
__global__ void __launch_bounds__(1024) handoff_stress_kernel(uint64_t *partials, uint32_t *counter, uint32_t epoch, uint64_t *res) {
    const uint32_t tid = threadIdx.x, nb = gridDim.x, lane = tid & 63u;
    // 1. warm this CU's L1 with the previous launch's partials: plain loads of every line
    uint64_t warm = 0;
    for (uint32_t k = tid; k < nb * 8; k += blockDim.x) warm ^= partials[k];
    if (warm == 0x0123456789abcdefull) res[7] = warm;  // keeps the loads
    // 2. uneven arrival: a block- and launch-dependent wait of 0 .. ~8 us
    const uint32_t spins = ((blockIdx.x * 2654435761u + epoch * 40503u) >> 7) % 37u;
    for (uint32_t k = 0; k < spins * 8; k++) __builtin_amdgcn_s_sleep(32);
    __syncthreads();
    // 3. the hand-off, exactly as in finish_round
    __shared__ uint32_t last;
    if (tid < 64) {
        if (lane == SC_LANE_G0 || lane == SC_LANE_G1) {
            const uint32_t half = lane == SC_LANE_G1 ? 1u : 0u;
            Fr v;
#pragma unroll
            for (int i = 0; i < 8; i++) v.l[i] = handoff_word(epoch, blockIdx.x, 8 * half + i);
            sc1_store_fr(partials + 8 * (size_t)blockIdx.x + 4 * half, v);
        }
        sc_drain_stores();
        if (lane == SC_LANE_G0) last = sc_arrive(counter, nb) ? 1u : 0u;
    }
    __syncthreads();
    if (!last) return;
    // 4. the last arriver checks every word of every partial
    uint32_t bad = 0;
    for (uint32_t k = tid; k < nb; k += blockDim.x) {
        Fr a = sc1_load_fr(partials + 8 * (size_t)k), b = sc1_load_fr(partials + 8 * (size_t)k + 4);
#pragma unroll
        for (int i = 0; i < 8; i++) bad += (a.l[i] != handoff_word(epoch, k, i)) + (b.l[i] != handoff_word(epoch, k, 8 + i));
    }
    if (bad) atomicAdd((unsigned long long *)res, (unsigned long long)bad);
    if (tid == 0) atomicAdd((unsigned long long *)(res + 1), 1ull);
}
__global__ void __launch_bounds__(256) handoff_busy_kernel(uint4 *buf, size_t n) {  // read-modify-write stream beside the test
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        uint4 v = buf[i];
        v.x += 1;
        buf[i] = v;
    }
}

// Sums of a table (a session's first round). SC_SUMS_U pairs per thread are requested before the first addition; 1024-thread
// workgroups, at most one per CU: 2^20 entries 25 -> 9 us, 2^24 202 -> 90 us (6 TB/s) against round 3's one-pair-per-trip loop of
// modular additions (tools/exp/fold_ab.hip, profiles/r4a_fold_ab_*_first_pass.txt, r4final_fold_ab_*).
constexpr int SC_SUMS_U = 2;
template <int LAYOUT>
__global__ void __launch_bounds__(1024) sc_sums_kernel(const uint64_t *t, size_t half, uint64_t *partials, uint64_t *sums,
                                                       uint32_t *counter, uint64_t *flag, uint64_t seq, ScRunArg run) {
    __shared__ u32 sh[SC_RED_WORDS];
    Acc9 g0 = acc9_zero(), g1 = acc9_zero();
    const size_t tile = (size_t)blockDim.x * SC_SUMS_U, stride = (size_t)gridDim.x * tile;
    for (size_t b = (size_t)blockIdx.x * tile + threadIdx.x; b < half; b += stride) {
        Fr lo[SC_SUMS_U], hi[SC_SUMS_U];
#pragma unroll
        for (int k = 0; k < SC_SUMS_U; k++) {
            const size_t i = b + (size_t)k * blockDim.x;
            lo[k] = Fr::zero();
            hi[k] = Fr::zero();
            if (i < half) {
                lo[k] = fe_load<FrParams>(t + 4 * (LAYOUT == ZG_SC_HIGH_HALF ? i : 2 * i));
                hi[k] = fe_load<FrParams>(t + 4 * (LAYOUT == ZG_SC_HIGH_HALF ? i + half : 2 * i + 1));
            }
        }
#pragma unroll
        for (int k = 0; k < SC_SUMS_U; k++) {
            acc9_add(g0, lo[k]);
            acc9_add(g1, hi[k]);
        }
    }
    finish_round(g0, g1, sh, partials, sums, counter, flag, seq, run);
}

// fold by r and produce the NEXT round's two sums from the values just written:
//   HIGH: out[i] = (1-r)*t[i] + r*t[i+half]     next g0 over i < half/2, g1 over i >= half/2
//   LOW : out[i] = t[2i] + r*(t[2i+1]-t[2i])    next g0 over even i,     g1 over odd i

// One level of HyperKZG.open (src/poly/commitment/mod.zig:296-310) in one pass over the table: the quotient
// q[j] = t[j + half] - t[j] (the first q_count entries are kept: commit() uses min(half, srs_len) of them) and the fold
// out[j] = (1 - r) t[j] + r t[j + half] = t[j] + r q[j]. It replaces a quotient launch plus a fold launch that also produced round
// sums nobody reads, and stays below the 112 registers per SIMD that a resident bucket accumulation leaves free: the open's serial
// quotient -> fold chain then runs beside the commits of the earlier levels instead of waiting for their workgroups to retire.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(56))) hk_quot_fold_kernel(const uint64_t *t, size_t half, FrArg r, uint64_t *q,
                                                                                               size_t q_count, uint64_t *out) {
    __builtin_amdgcn_s_setprio(3);
    Fr rv;
#pragma unroll
    for (int i = 0; i < 8; i++) rv.l[i] = r.l[i];
    FrMul rp = frmul_prepare(rv);
    size_t stride = (size_t)gridDim.x * 256;
    for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < half; j += stride) {
        Fr lo = fe_load<FrParams>(t + 4 * j), hi = fe_load<FrParams>(t + 4 * (j + half));
        Fr d = fe_sub(hi, lo);
        if (j < q_count) fe_store(q + 4 * j, d);
        fe_store(out + 4 * j, fe_add(lo, frmul_apply(d, rp)));
    }
}

// 512-thread workgroups, at most 512 of them (two per CU = four waves per SIMD), grid-stride with the next pair requested before the
// current product. Round 3 ran one wave per SIMD (every trip a full memory latency) and ended each block with a shuffle tree of modular
// additions plus an ACQ_REL arrival. Steady state (tools/exp/fold_ab.hip; profiles/r4a_fold_ab_*_first_pass.txt, r4final_fold_ab_*):
// 2^16 entries 15.7 -> 7.8 us, 2^20 30.0 -> 15.7 us, 2^24 245 -> 168 us (4.8 TB/s). What is left at 2^20 is issue, not traffic: ~2 us
// launch boundary, ~2 us until the first loads land, ~8 us of VALU issue (532 executed instructions per pair: SQ_INSTS_VALU 4.1 M wave
// instructions per launch, profiles/r4final_kernel_table.md), then ~4 us of hand-off for the last workgroup (three dependent round trips).
template <int LAYOUT>
__global__ void __launch_bounds__(512) sc_fold_kernel(const uint64_t *t, size_t half, FrArg r, uint64_t *out,
                                                     uint64_t *partials, uint64_t *sums, uint32_t *counter, uint64_t *flag,
                                                     uint64_t seq, ScRunArg run) {
    __shared__ u32 sh[SC_RED_WORDS];
    const size_t stride = (size_t)gridDim.x * blockDim.x, quarter = half / 2;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fr lo = Fr::zero(), hi = Fr::zero();
    if (i < half) {  // requested before the challenge is prepared
        lo = fe_load<FrParams>(LAYOUT == ZG_SC_HIGH_HALF ? t + 4 * i : t + 8 * i);
        hi = fe_load<FrParams>(LAYOUT == ZG_SC_HIGH_HALF ? t + 4 * (i + half) : t + 8 * i + 4);
    }
    FrMul rp;  // the challenge is the shared factor of every product of this launch (narrow form for a 128-bit one)
    Fr run_claim = Fr::zero();
    if (run.res) {  // device-resident protocol: this launch opens with the verifier step of the round whose sums the previous launch left
        rp.p = sc_run_open(run, partials, sh, &run_claim);
        rp.narrow = false;
    } else {
        Fr rv;
#pragma unroll
        for (int k = 0; k < 8; k++) rv.l[k] = r.l[k];
        rp = frmul_prepare(rv);
    }
    Acc9 g0 = acc9_zero(), g1 = acc9_zero();
    while (i < half) {
        const size_t ni = i + stride;
        Fr nlo = lo, nhi = hi;
        if (ni < half) {
            nlo = fe_load<FrParams>(LAYOUT == ZG_SC_HIGH_HALF ? t + 4 * ni : t + 8 * ni);
            nhi = fe_load<FrParams>(LAYOUT == ZG_SC_HIGH_HALF ? t + 4 * (ni + half) : t + 8 * ni + 4);
        }
        Fr v = fe_add(lo, frmul_apply(fe_sub(hi, lo), rp));  // (1-r)*lo + r*hi = lo + r*(hi - lo): one product, same value
        fe_store(out + 4 * i, v);
        const bool second = LAYOUT == ZG_SC_HIGH_HALF ? (i >= quarter) : (i & 1);
        acc9_add_if(g0, v, !second);
        acc9_add_if(g1, v, second);
        lo = nlo;
        hi = nhi;
        i = ni;
    }
    if (run.res && half == 1) {  // the table is down to one element: getFinalEval + the verifier's last comparison
        if (threadIdx.x == 0 && blockIdx.x == 0) {  // (a one-pair fold is one workgroup; its store above is this thread's own)
            Fr fin = fe_load<FrParams>(out);
            fe_store(run.res + run_off_final(run.v), fin);
            if (fr_eq(fin, run_claim)) run.res[run_off_status(run.v)] |= 1;
        }
        return;
    }
    finish_round(g0, g1, sh, partials, sums, counter, flag, seq, run);
}

// Last rounds of the device-resident protocol in ONE single-block launch: once the table fits in LDS (<= 4096
// entries = 128 KiB) every remaining round — fold by the challenge the previous verifier step left, sums of the folded
// table, verifier step, next challenge — stays inside the workgroup; a round costs a block reduction and two
// dependent field products instead of a kernel launch. run.round = index of the next verifier step.
// End of the device-resident protocol: the result block goes to a pinned host buffer straight from the last kernel (write-through
// system-scope stores, drained, then the flag word) — the host spins on the flag instead of paying a device-to-host copy and a stream
// synchronisation (~15 us of a 0.24 ms protocol). Called by every thread of ONE block, behind a barrier that follows the block's own
// stores to `res` (sc1 loads: the verifier lane's stores are then read from L2 whatever this CU's L1 holds).
__device__ __forceinline__ void run_publish(const uint64_t *res, uint32_t words, uint64_t *host, uint64_t *hflag) {
    if (!host) return;
    for (uint32_t k = threadIdx.x; k < words; k += blockDim.x) {
        uint64_t w = __hip_atomic_load((const sc_gu64 *)res + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store((sc_gu64 *)host + k, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave, before the barrier behind which one lane signals
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store((sc_gu64 *)hflag, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void __launch_bounds__(256) sc_run_publish_kernel(const uint64_t *res, uint32_t words, uint64_t *host, uint64_t *hflag) {
    run_publish(res, words, host, hflag);
}

constexpr uint32_t SC_TAIL_MAX = 4096, SC_TAIL_THREADS = 1024;
constexpr size_t SC_TAIL_LDS_EXTRA = (SC_RED_WORDS + 12) * 4;  // reduction scratch + the prescaled challenge, after the table
__global__ void __launch_bounds__(SC_TAIL_THREADS) sc_tail_run_kernel(const uint64_t *t, uint32_t len, ScRunArg run, const uint64_t *partials,
                                                                       uint32_t res_words, uint64_t *host, uint64_t *hflag) {
    extern __shared__ uint4 lds_tab[];  // len entries of 2 x uint4, then SC_RED_WORDS u32 of reduction scratch + 9 for the challenge
    u32 *sh = reinterpret_cast<u32 *>(lds_tab + 2 * (size_t)len);
    u32 *sh_rp = sh + SC_RED_WORDS;  // the prescaled challenge (9 limbs), written by the lane that ran the verifier step
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    for (uint32_t i = tid; i < len; i += SC_TAIL_THREADS) fe_store(&lds_tab[2 * i], fe_load<FrParams>(t + 4 * (size_t)i));
    Fr claim;  // stays in registers from here on
    F29 rp = sc_run_open(run, partials, sh, &claim);  // the verifier step of the round whose sums the last fold (or the sums launch) left
    run.round++;
    run.init = 0;
    __syncthreads();
    // ---- phase A: rounds with at least 64 pairs, all waves. A round = fold in LDS, block reduction, verifier step in lane
    // SC_LANE_G0 of wave 0 (which keeps the claim), challenge through LDS.
    while (len > 64) {
        const uint32_t half = len / 2, quarter = half / 2;
        Acc9 g0 = acc9_zero(), g1 = acc9_zero();
        for (uint32_t i = tid; i < half; i += SC_TAIL_THREADS) {
            Fr lo = fe_load<FrParams>(&lds_tab[2 * i]), hi = fe_load<FrParams>(&lds_tab[2 * (i + half)]);
            Fr v = fe_add(lo, fr_mul29(fe_sub(hi, lo), rp));
            fe_store(&lds_tab[2 * i], v);  // in place: entry i is read and written by this thread only
            acc9_add_if(g0, v, i < quarter);
            acc9_add_if(g1, v, i >= quarter);
        }
        len = half;
        Fr tot = block_sum_pair9(g0, g1, sh);
        if (tid < 64) {
            Fr second = pair_second_to_first(tot);
            if (lane == SC_LANE_G0) {
                F29 cp;
                (void)sc_verifier_step(run, tot, second, &cp, &claim);
#pragma unroll
                for (int i = 0; i < 9; i++) sh_rp[i] = cp.l[i];
            }
        }
        run.round++;
        __syncthreads();  // the folded table and the challenge are visible; the reduction scratch is free again
#pragma unroll
        for (int i = 0; i < 9; i++) rp.l[i] = sh_rp[i];
    }
    // ---- phase B: at most 32 pairs — wave 0 alone, no barriers, no LDS hops for the sums or the challenge. The claim update
    // g0 + (g1 - g0) * ch is the fold formula applied to the previous round's pair: lane 63 (never a folding lane here) carries it
    // through the same instruction stream as the table's pairs, so the verifier lane's serial work is the check, the hash and the
    // conversion of the challenge only.
    if (tid < 64) {
#pragma unroll
        for (int i = 0; i < 8; i++) claim.l[i] = __shfl(claim.l[i], (int)SC_LANE_G0, 64);  // phase A's verifier lane (every lane holds the loaded one if it never ran)
        Fr pg0 = Fr::zero(), pg1 = Fr::zero();
        bool have_pg = false;
        while (len > 1) {
            const uint32_t half = len / 2, quarter = half / 2;
            Fr lo = Fr::zero(), hi = Fr::zero();
            if (lane < half) {
                lo = fe_load<FrParams>(&lds_tab[2 * lane]);
                hi = fe_load<FrParams>(&lds_tab[2 * (lane + half)]);
            } else if (lane == 63 && have_pg) {
                lo = pg0;
                hi = pg1;
            }
            Fr v = fe_add(lo, fr_mul29(fe_sub(hi, lo), rp));
            if (lane < half) fe_store(&lds_tab[2 * lane], v);
            if (lane == 63 && have_pg) claim = v;
            len = half;
            if (len == 1) break;
            Acc9 g0 = acc9_zero(), g1 = acc9_zero();
            acc9_add_if(g0, v, lane < quarter);
            acc9_add_if(g1, v, lane >= quarter && lane < half);
            wave_sum_pair9(g0, g1);          // lane 31: the g0 total, lane 63: the g1 total
            const Fr tot = acc9_reduce(g0);
            Fr G0, G1, cl;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                G0.l[i] = __shfl(tot.l[i], 31, 64);
                G1.l[i] = __shfl(tot.l[i], 63, 64);
                cl.l[i] = __shfl(claim.l[i], 63, 64);
            }
            F29 cp;
#pragma unroll
            for (int i = 0; i < 9; i++) cp.l[i] = 0;
            if (lane == 31) (void)sc_verifier_step(run, G0, G1, &cp, &cl, true);
#pragma unroll
            for (int i = 0; i < 9; i++) rp.l[i] = __shfl(cp.l[i], 31, 64);
            pg0 = G0;
            pg1 = G1;
            have_pg = true;
            run.round++;
        }
        if (lane == 63) {
            Fr fin = fe_load<FrParams>(&lds_tab[0]);
            fe_store(run.res + run_off_final(run.v), fin);
            if (fr_eq(fin, claim)) run.res[run_off_status(run.v)] |= 1;
        }
    }
    __syncthreads();  // (waits for the stores above: the workgroup-scope release of the barrier)
    run_publish(run.res, res_words, host, hflag);
}

// R1CSInputEvaluator.computeClaimedInputs (src/zkvm/r1cs/evaluation.zig:55-122): out[i] = sum_t eq[t] * rows[t][i] for the k columns of a
// cycle-major matrix (k field elements per cycle, the layout of R1CSCycleInputs.values). The matrix is read as ONE flat stream of
// n_rows * k elements with every lane busy: the grid has a multiple of k threads, so thread g always meets column g mod k and its
// cycle advances by (threads / k) per iteration — no division in the loop, coalesced 32-byte loads, one general product per element
// (the eq value of an element's cycle is a second, mostly shared, 32-byte load). partials[thread] = that thread's column sum.
constexpr unsigned ROWS_MLE_MAX_K = 64;
__global__ void __launch_bounds__(256) rows_mle_kernel(const uint64_t *rows, size_t n_rows, uint32_t k, const uint64_t *eq, uint64_t *partials) {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x, threads = (size_t)gridDim.x * 256;
    const size_t t_step = threads / k;  // threads is a multiple of k
    Fr acc = Fr::zero();
    Acc29 lazy = acc29_zero();  // products summed limb-wise in 64-bit words, reduced every FR29_ACC_MAX terms (fp29.hip.h)
    unsigned cnt = 0;
    size_t e = g;
    for (size_t t = g / k; t < n_rows; t += t_step, e += threads) {
        acc29_add(lazy, fr29_chain_mul(fr29_in(fe_load<FrParams>(rows + 4 * e)), fr29_in_shift(fe_load<FrParams>(eq + 4 * t))));
        if (++cnt == FR29_ACC_MAX) {
            acc = fe_add(acc, acc29_reduce(lazy));
            lazy = acc29_zero();
            cnt = 0;
        }
    }
    if (cnt) acc = fe_add(acc, acc29_reduce(lazy));
    fe_store(partials + 4 * g, acc);
}

// out[column] = sum of the partials of the threads g with g mod k == column; one block per column
__global__ void __launch_bounds__(256) rows_mle_finish_kernel(const uint64_t *partials, size_t threads, uint32_t k, uint64_t *out) {
    __shared__ uint4 sh[256 * 4];
    Fr g0 = Fr::zero(), g1 = Fr::zero();
    for (size_t g = blockIdx.x + (size_t)threadIdx.x * k; g < threads; g += (size_t)256 * k) g0 = fe_add(g0, fe_load<FrParams>(partials + 4 * g));
    block_sum_pair(g0, g1, sh);
    if (threadIdx.x == 0) fe_store(out + 4 * (size_t)blockIdx.x, g0);
}

// Affine maps of a cycle-major matrix: out[c][i] = coeff[c][k] + sum_{col < k} coeff[c][col] * rows[i][col] for nout <= 16 outputs —
// what StreamingOuterProver.materializeLinearPhasePolynomials computes per cycle for (Az, Bz) x (first, second constraint group)
// (src/zkvm/spartan/streaming_outer.zig:258-372): the Lagrange-weighted sums of the constraints' condition / left - right linear
// combinations ARE one affine map of the cycle's R1CS inputs each. Wave w of a block <-> output w, lane <-> cycle: the column list of
// an output (its non-zero coefficients only) and the prescaled coefficients are wave-uniform, a lane reads just the elements of its
// row that the output uses; the terms are summed limb-wise and reduced once. Output c goes to table c / g, element i * g + c % g
// (g interleaved outputs per table: Az[2 i + group]); rows i in [n_rows, n_pad) are written as zero.
constexpr unsigned ROWS_AFFINE_MAX_OUT = 16, ROWS_AFFINE_MAX_K = 128, ROWS_AFFINE_MAX_NNZ = 64;
struct RowsAffineArgs {
    uint64_t *tab[ROWS_AFFINE_MAX_OUT];  // per OUTPUT: its table's base
    uint8_t nnz[ROWS_AFFINE_MAX_OUT];
};
// coeff: nout x (k + 1) canonical elements -> pre: the same entries as prescaled 29-bit limbs (9 words each); small (optional): per entry
// ROWS_SMALL | sign << 31 | magnitude when the coefficient is +-m with 0 < m < 2^24 as an INTEGER (its Montgomery form says nothing about
// that: the entry is taken out of it first), else 0
constexpr uint32_t ROWS_SMALL = 0x40000000u, ROWS_SMALL_MAG = 0x00FFFFFFu;
__global__ void __launch_bounds__(256) rows_affine_prep_kernel(const uint64_t *coeff, uint32_t n, uint32_t *pre, uint32_t *small = nullptr) {
    uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const Fr cm = fe_load<FrParams>(coeff + 4 * (size_t)e);
    F29 f = fr29_prescale(cm);
#pragma unroll
    for (int i = 0; i < 9; i++) pre[9 * (size_t)e + i] = f.l[i];
    if (small) {
        const Fr c = fr_from_mont29(cm), nc = fe_sub(Fr::zero(), c);
        uint32_t hi_c = 0, hi_n = 0;
#pragma unroll
        for (int i = 1; i < 8; i++) {
            hi_c |= c.l[i];
            hi_n |= nc.l[i];
        }
        uint32_t w = 0;
        if (hi_c == 0 && c.l[0] != 0 && c.l[0] <= ROWS_SMALL_MAG) w = ROWS_SMALL | c.l[0];
        else if (hi_n == 0 && nc.l[0] != 0 && nc.l[0] <= ROWS_SMALL_MAG) w = ROWS_SMALL | 0x80000000u | nc.l[0];
        small[e] = w;
    }
}
__global__ void __launch_bounds__(1024) rows_affine_kernel(const uint64_t *rows, size_t n_rows, uint32_t k, uint32_t stride, const uint64_t *coeff, const uint32_t *pre,
                                                           const uint8_t *cols /* nout x 64 */, RowsAffineArgs a, uint32_t g, size_t n_pad) {
    const uint32_t c = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t i = (size_t)blockIdx.x * 64 + lane;
    if (i >= n_pad) return;
    Fr val = Fr::zero();
    if (i < n_rows) {
        const uint64_t *row = rows + 4 * i * stride;  // k >= stride: a map may read into the following rows (a sliding window)
        const uint32_t *pc = pre + 9 * (size_t)c * (k + 1);
        const uint8_t *cl = cols + 64 * c;
        Acc29 lazy = acc29_zero();
        const uint32_t nnz = a.nnz[c];  // <= 64 = FR29_ACC_MAX: one reduction
        for (uint32_t j = 0; j < nnz; j++) {
            const uint32_t col = cl[j];
            F29 y;
#pragma unroll
            for (int w = 0; w < 9; w++) y.l[w] = pc[9 * col + w];
            acc29_add(lazy, fr29_chain_mul(fr29_in(fe_load<FrParams>(row + 4 * col)), y));
        }
        val = fe_load<FrParams>(coeff + 4 * ((size_t)c * (k + 1) + k));  // the constant
        if (nnz) val = fe_add(val, acc29_reduce(lazy));
    }
    fe_store(a.tab[c] + 4 * (i * g + c % g), val);
}

// Weighted sums of PRODUCTS of two affine maps of the rows: out[p] = sum_i W[i * G + p % G] * A_p(row_i) * B_p(row_i) — the extended
// evaluations of StreamingOuterProver.computeFirstRoundPoly (src/zkvm/spartan/streaming_outer.zig:523-597): for each of the nine UniSkip
// targets and each constraint group, Az and Bz at the target are Lagrange extrapolations of the group's constraint values, i.e. affine
// maps of the cycle's inputs; their product is summed under eq(tau_low, (cycle, group)). Wave w of a block <-> pair (blockIdx.y * waves
// + w), lane <-> cycle (grid-stride over the cycles): the two affine maps as in rows_affine_kernel (wave-uniform column lists, lazy
// limb-wise sums), one product, one product by the weight, a lane-wise running sum; a shuffle tree per wave at the end.
constexpr unsigned ROWS_PS_MAX_PAIRS = 32, ROWS_PS_WAVES = 8;
struct RowsProdSumArgs {
    uint8_t nnz[2 * ROWS_PS_MAX_PAIRS];
};
// Small coefficients (round 4): the nine targets' maps are Lagrange-weighted sums of constraint rows, i.e. their coefficients are small
// INTEGERS (a few thousand at most) — a term is then an 8 x 1-limb product of the stored element, summed as a 288-bit integer with
// carries (an integer combination of Montgomery forms is the Montgomery form of the combination) and reduced once per map by
// acc9_reduce: ~25 instructions per term instead of ~200 for the general product. +-m with m < 2^24 and <= 64 terms keep a sum below
// 2^30 r; anything else takes the general path into the lazy limb sum, as before.
// STAGED: the workgroup's tile of 64 rows is copied to LDS once (column-major, a column every 65 elements: the transposing writes of
// consecutive lanes fall on different banks) and all its waves' maps read the elements from there — without it every map re-reads its
// ~25 elements of a row through L1 / L2, 40 GB at 2^20 cycles for 1.4 GB of witness, which bound the kernel once the terms were cheap.
constexpr uint32_t ROWS_TILE_COL = 65 * 2;  // uint4 units per staged column (64 rows x 32 bytes + one element of padding)
// A staged workgroup serves ALL pairs of its tile — wave w takes pairs w, w + waves, ... (gridDim.y = 1, up to ROWS_PS_STAGED_WAVES
// waves): the tile is copied once per 64 rows instead of once per eight pairs, and no wave idles beside a copy made for two pairs.
constexpr unsigned ROWS_PS_STAGED_WAVES = 10;
template <bool STAGED>
__global__ void __launch_bounds__(64 * (STAGED ? ROWS_PS_STAGED_WAVES : ROWS_PS_WAVES)) rows_affine_prodsum_kernel(const uint64_t *rows, size_t n_rows, uint32_t k, uint32_t stride, const uint64_t *coeff,
                                                                                 const uint32_t *pre, const uint32_t *small, const uint8_t *cols,
                                                                                 RowsProdSumArgs a, const uint64_t *w, uint32_t G, uint32_t npairs,
                                                                                 uint64_t *partials) {
    extern __shared__ uint4 tile[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6, p_first = STAGED ? wave : blockIdx.y * ROWS_PS_WAVES + wave;
    if (!STAGED && p_first >= npairs) return;  // (a staged workgroup keeps every wave for the copy and the barriers)
    Fr accs[4];  // a wave's pairs: at most four (32 pairs over eight waves)
#pragma unroll
    for (int q = 0; q < 4; q++) accs[q] = Fr::zero();
    for (size_t i0 = (size_t)blockIdx.x * 64; i0 < n_rows; i0 += (size_t)gridDim.x * 64) {
        const size_t i = i0 + lane;
        if (STAGED) {
            __syncthreads();  // the previous tile has been read
            const uint32_t live = n_rows - i0 < 64 ? (uint32_t)(n_rows - i0) : 64u;
            for (uint32_t e = threadIdx.x; e < live * k; e += blockDim.x) {
                const uint32_t r = e / k, c = e - r * k;
                const uint4 *src = reinterpret_cast<const uint4 *>(rows + 4 * ((i0 + r) * stride + c));
                tile[c * ROWS_TILE_COL + 2 * r] = src[0];
                tile[c * ROWS_TILE_COL + 2 * r + 1] = src[1];
            }
            __syncthreads();
        }
        if (i >= n_rows) continue;
        const uint64_t *row = rows + 4 * i * stride;
        for (int q = 0; q < 4; q++) {  // (rolled: the four running sums live in scratch, two accesses per pair and row)
        const uint32_t p = p_first + (uint32_t)q * (STAGED ? nw : 0u);
        if ((q && !STAGED) || p >= npairs) continue;
        Fr ab[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t c = 2 * p + h;
            const uint32_t *pc = pre + 9 * (size_t)c * (k + 1);
            const uint32_t *sm = small + (size_t)c * (k + 1);
            const uint8_t *cl = cols + 64 * c;
            Acc29 lazy = acc29_zero();
            Acc9 pos = acc9_zero(), neg = acc9_zero();
            bool any_gen = false, any_pos = false, any_neg = false;  // wave-uniform: every lane walks the same map
            const uint32_t nnz = a.nnz[c];
            for (uint32_t j = 0; j < nnz; j++) {
                const uint32_t col = cl[j], sw = sm[col];
                Fr x;
                if (STAGED) {
                    const uint4 lo = tile[col * ROWS_TILE_COL + 2 * lane], hi = tile[col * ROWS_TILE_COL + 2 * lane + 1];
                    x.l[0] = lo.x; x.l[1] = lo.y; x.l[2] = lo.z; x.l[3] = lo.w;
                    x.l[4] = hi.x; x.l[5] = hi.y; x.l[6] = hi.z; x.l[7] = hi.w;
                } else {
                    x = fe_load<FrParams>(row + 4 * col);
                }
                if (sw) {
                    const uint32_t mag = sw & ROWS_SMALL_MAG;
                    Acc9 prod;
                    uint64_t carry = 0;
#pragma unroll
                    for (int t = 0; t < 8; t++) {
                        carry += (uint64_t)x.l[t] * mag;
                        prod.l[t] = (uint32_t)carry;
                        carry >>= 32;
                    }
                    prod.l[8] = (uint32_t)carry;
                    if (sw >> 31) {
                        acc9_add_acc(neg, prod);
                        any_neg = true;
                    } else {
                        acc9_add_acc(pos, prod);
                        any_pos = true;
                    }
                } else {
                    F29 y;
#pragma unroll
                    for (int t = 0; t < 9; t++) y.l[t] = pc[9 * col + t];
                    acc29_add(lazy, fr29_chain_mul(fr29_in(x), y));
                    any_gen = true;
                }
            }
            ab[h] = fe_load<FrParams>(coeff + 4 * ((size_t)c * (k + 1) + k));
            if (any_gen) ab[h] = fe_add(ab[h], acc29_reduce(lazy));
            if (any_pos) ab[h] = fe_add(ab[h], acc9_reduce(pos));
            if (any_neg) ab[h] = fe_sub(ab[h], acc9_reduce(neg));
        }
        if (ab[0].is_zero() || ab[1].is_zero()) continue;
        accs[q] = fe_add(accs[q], fr_mul29v(fr_mul29v(ab[0], ab[1]), fe_load<FrParams>(w + 4 * (i * G + p % G))));
        }
    }
    for (int q = 0; q < 4; q++) {
        const uint32_t p = p_first + (uint32_t)q * (STAGED ? nw : 0u);
        if ((q && !STAGED) || p >= npairs) continue;
        Fr acc = accs[q];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) acc = fe_add(acc, fr_shfl_down(acc, d));
        if (lane == 0) fe_store(partials + 4 * ((size_t)blockIdx.x * npairs + p), acc);
    }
}
// (Round 4 also tried the ROW as the outer loop — a wave owning two or four pairs walks the columns, one load per element per wave, the
// small-coefficient sums of its four or eight maps in registers: 5.9 / 7.8 ms against 4.1 ms for the kernel above at 2^20 cycles. The
// column loop is a chain of dependent scalar loads, a row load and up to eight carry chains per column at one or two waves per SIMD;
// the kernel above re-reads the row through L1 / L2 but keeps eight independent waves per workgroup busy. Taken out again.)
// out[p] = sum over the blocks' partials; one block per pair
__global__ void __launch_bounds__(256) rows_prodsum_finish_kernel(const uint64_t *partials, uint32_t nblocks, uint32_t npairs, uint64_t *out) {
    __shared__ uint4 sh[256 * 4];
    Fr g0 = Fr::zero(), g1 = Fr::zero();
    for (uint32_t b = threadIdx.x; b < nblocks; b += 256) g0 = fe_add(g0, fe_load<FrParams>(partials + 4 * ((size_t)b * npairs + blockIdx.x)));
    block_sum_pair(g0, g1, sh);
    if (threadIdx.x == 0) fe_store(out + 4 * (size_t)blockIdx.x, g0);
}

// reduce the per-block partial pairs to sums[0..8)
__global__ void __launch_bounds__(256) sc_finish_kernel(const uint64_t *partials, uint32_t nblocks, uint64_t *sums, uint64_t *flag,
                                                        uint64_t seq) {
    __shared__ uint4 sh[256 * 4];
    Fr g0 = Fr::zero(), g1 = Fr::zero();
    for (uint32_t k = threadIdx.x; k < nblocks; k += 256) {
        g0 = fe_add(g0, fe_load<FrParams>(partials + 8 * (size_t)k));
        g1 = fe_add(g1, fe_load<FrParams>(partials + 8 * (size_t)k + 4));
    }
    block_sum_pair(g0, g1, sh);
    if (threadIdx.x == 0) {
        mailbox_store_fr(sums, g0, flag);
        mailbox_store_fr(sums + 4, g1, flag);
        publish_seq(flag, seq);
    }
}

// Spartan's first sumcheck instance in one pass (src/zkvm/spartan/mod.zig:182-206 followed by Sumcheck.Prover.init): the table
//   f[i] = eq(r, i) * (Az[i]*Bz[i] - Cz[i]),   eq(r, i) = hi[i >> v_lo] * lo[i & (2^v_lo - 1)]
// is written straight into the session's buffer and round 0's pair of sums is accumulated on the way (finished by the last
// block to arrive, like every other round): the eq table is never materialised, the 2^v-entry copy into the session and the
// separate first sums pass disappear.
// WG groups of 256 threads per block (thread (grp, lo) takes rows grp, grp + WG, ...), the next row's three operands requested before
// the current row's three products: round 3 ran 512 x 256 threads with the loads inside the trip (two waves per SIMD, each trip a
// memory latency plus three dependent products) — 78 us at 2^20 for 134 MB and 3 x 2^20 products (~20 us of multiplier throughput).
template <int LAYOUT, int WG>
__global__ void __launch_bounds__(256 * WG) eq_spartan_kernel(EqArgs ea, int v_lo, int v_hi, uint32_t hi_per_block, const uint64_t *az,
                                                              const uint64_t *bz, const uint64_t *cz, uint64_t *out, uint64_t *partials,
                                                              uint64_t *sums, uint32_t *counter, uint64_t *flag, uint64_t seq) {
    __shared__ u32 sh[SC_RED_WORDS];
    __shared__ EqShared fs;
    const uint32_t n_hi = 1u << v_hi, h0 = blockIdx.x * hi_per_block;
    const uint32_t rows = n_hi - h0 < hi_per_block ? n_hi - h0 : hi_per_block;
    const uint32_t lo = threadIdx.x & 255u, grp = threadIdx.x >> 8;
    const bool live = lo < (1u << v_lo);
    uint32_t k = grp;
    Fr a = Fr::zero(), b = Fr::zero(), c = Fr::zero();
    if (live && k < rows) {  // requested before the factor prologue
        const size_t i = ((size_t)(h0 + k) << v_lo) | lo;
        a = fe_load<FrParams>(az + 4 * i);
        b = fe_load<FrParams>(bz + 4 * i);
        if (cz) c = fe_load<FrParams>(cz + 4 * i);  // cz == nullptr: Cz is identically zero (the uniform R1CS: condition * (left - right) = 0)
    }
    Fr lov = eq_block_factors(ea, v_lo, v_hi, h0, rows, fs);
    Acc9 g0 = acc9_zero(), g1 = acc9_zero();
    if (live) {
        F29 tp = fr29_prescale(lov);
        while (k < rows) {
            const uint32_t nk = k + WG, h = h0 + k;
            const size_t i = ((size_t)h << v_lo) | lo;
            Fr na = a, nb = b, nc = c;
            if (nk < rows) {
                const size_t ni = ((size_t)(h0 + nk) << v_lo) | lo;
                na = fe_load<FrParams>(az + 4 * ni);
                nb = fe_load<FrParams>(bz + 4 * ni);
                if (cz) nc = fe_load<FrParams>(cz + 4 * ni);
            }
            // f = e (a b - c) with e = hi lo, as ONE lazy chain (no canonical value between the three products): with every
            // 2^-261 of the lazy multiplier paid for by a 5-bit shift of one operand,
            //   E  = hi * (32 lo)          * 2^-261 = e                      (< 1.2 r)
            //   AB = (32 a) * (32 b)       * 2^-261 = 32 a b R^-1            (< 7.1 r)
            //   W  = AB - 32 c + 64 r                = 32 (a b R^-1 - c) mod r  (limb-wise with a pre-biased 64 r, < 71.1 r < 2^261)
            //   f  = E * W                 * 2^-261 = e (a b R^-1 - c) R^-1   (< (1.2 * 71.1 / 168.9 + 1) r = 1.5 r: one conditional subtraction)
            // — the reference's montgomeryMul(e, sub(montgomeryMul(a, b), c)), since Montgomery products are exact. Saves two
            // canonicalisations, a modular subtraction and two unpacks per entry of a kernel that is bound by its instruction count.
            constexpr u32 BIAS64R[9] = {0x40000040u, 0x43eb27deu, 0x5709143cu, 0x54243cdau, 0x4174a0cdu, 0x56d03029u, 0x49b85043u, 0x57098cffu, 0x0c19139au};
            const F29 E = f29t_mul<Fr29>(eq_hi_row(fs, k), tp);
            const F29 AB = f29t_mul<Fr29>(fr29_in_shift(a), fr29_in_shift(b));
            const F29 cs = fr29_in_shift(c);
            F29 W;
#pragma unroll
            for (int q = 0; q < 9; q++) W.l[q] = AB.l[q] + BIAS64R[q] - cs.l[q];
            Fr f = fr29_out(f29t_mul<Fr29>(E, f29_carry(W)));
            fe_store(out + 4 * i, f);
            const bool second = LAYOUT == ZG_SC_HIGH_HALF ? (n_hi > 1 ? h >= n_hi / 2 : lo >= (1u << v_lo) / 2) : (lo & 1u);
            acc9_add_if(g0, f, !second);
            acc9_add_if(g1, f, second);
            a = na;
            b = nb;
            c = nc;
            k = nk;
        }
    }
    finish_round(g0, g1, sh, partials, sums, counter, flag, seq, ScRunArg{nullptr, 0, 0, 0});
}

// RafEvaluationProver.computeRoundPolynomialCubic's two sums (src/zkvm/ram/raf_checking.zig:335-410) over a LowToHigh table:
//   s(0) = sum_i ra[2i] * u0(i),   s(2) = sum_i (2 ra[2i+1] - ra[2i]) * u2(i),
//   u0(i) = base + F(rem(i)),  u2(i) = u0(i) + F(2 * current_power),  rem(i) = sum_j bit_j(i) * current_power * 2^(j+1) = step * i
// (`base` = start_address + 8 * sum_j bound_j 2^j, a handful of host scalar operations; step = 2 * current_power; the host has
// checked that step * half fits 64 bits, so F.fromU64 of the sum equals the reference's sum of F.fromU64 terms).
__global__ void __launch_bounds__(256) raf_round_kernel(const uint64_t *t, size_t half, FrArg base, uint64_t step, uint64_t *partials,
                                                        uint64_t *sums, uint32_t *counter, uint64_t *flag, uint64_t seq) {
    __shared__ u32 sh[SC_RED_WORDS];
    Fr bv;
#pragma unroll
    for (int i = 0; i < 8; i++) bv.l[i] = base.l[i];
    const Fr cp2 = fr_from_u64_29(step);  // F.fromU64(2 * current_power)
    Acc9 g0 = acc9_zero(), g1 = acc9_zero();
    size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < half; i += stride) {
        Fr lo = fe_load<FrParams>(t + 8 * i), hi = fe_load<FrParams>(t + 8 * i + 4);
        Fr u0 = fe_add(bv, fr_from_u64_29(step * (uint64_t)i));  // base + F.fromU64(rem)
        Fr u2 = fe_add(u0, cp2);
        Fr ra2 = fe_sub(fe_add(hi, hi), lo);
        acc9_add(g0, fr_mul29v(lo, u0));
        acc9_add(g1, fr_mul29v(ra2, u2));
    }
    finish_round(g0, g1, sh, partials, sums, counter, flag, seq, ScRunArg{nullptr, 0, 0, 0});  // the round ends inside this launch
}

// RafEvaluationProver.computeInitialClaim (src/zkvm/ram/raf_checking.zig:312-321): sum_k ra(k) * F.fromU64(start_address + 8 k) over the
// table the session holds (base + step * k < 2^64 for every k: checked by the host) — the pair is (claim, 0)
__global__ void __launch_bounds__(256) raf_claim_kernel(const uint64_t *t, size_t n, uint64_t base, uint64_t step, uint64_t *partials, uint64_t *sums,
                                                        uint32_t *counter, uint64_t *flag, uint64_t seq) {
    __shared__ u32 sh[SC_RED_WORDS];
    Acc9 g0 = acc9_zero(), g1 = acc9_zero();
    size_t stride = (size_t)gridDim.x * 256;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += stride)
        acc9_add(g0, fr_mul29v(fe_load<FrParams>(t + 4 * k), fr_from_u64_29(base + step * (uint64_t)k)));
    finish_round(g0, g1, sh, partials, sums, counter, flag, seq, ScRunArg{nullptr, 0, 0, 0});
}

// LassoProver.computeAddressRoundPoly's two sums (src/zkvm/lasso/prover.zig:283-293): the eq values split by bit `bit` of the
// u128 lookup index (two little-endian u64 words per entry)
// `sums` == nullptr: the block pairs stay in `partials` (finished by sc_finish_kernel); otherwise the round ends inside this launch
// (finish_round: pinned mailbox + sequence word), as in the fold kernels.
__global__ void __launch_bounds__(256) bit_split_sums_kernel(const uint64_t *vals, const uint64_t *idx, size_t n, uint32_t bit,
                                                             uint64_t *partials, uint64_t *sums, uint32_t *counter, uint64_t *flag,
                                                             uint64_t seq) {
    __shared__ u32 sh[SC_RED_WORDS];
    Acc9 g0 = acc9_zero(), g1 = acc9_zero();
    size_t stride = (size_t)gridDim.x * 256;
    const uint32_t word = bit >> 6, sh_bits = bit & 63u;
    for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < n; j += stride) {
        Fr v = fe_load<FrParams>(vals + 4 * j);
        const bool one = (idx[2 * j + word] >> sh_bits) & 1ull;
        acc9_add_if(g0, v, !one);
        acc9_add_if(g1, v, one);
    }
    if (sums) {
        finish_round(g0, g1, sh, partials, sums, counter, flag, seq, ScRunArg{nullptr, 0, 0, 0});
    } else {
        store_block_pair(g0, g1, sh, partials);
    }
}

// LassoProver.receiveChallenge, address phase (src/zkvm/lasso/prover.zig:375-399): t[j] *= (bit `bit` of idx[j]) ? r : 1 - r for the
// n lookups, in place, and — in the same pass — the NEXT address round's two sums (the scaled values split by bit `next_bit`,
// :283-293); their total is the new current_claim (:394-399; entries past the lookups are not touched by the reference either).
__global__ void __launch_bounds__(256) bit_bind_kernel(uint64_t *t, const uint64_t *idx, size_t n, uint32_t bit, uint32_t next_bit,
                                                       FrArg r, uint64_t *partials, uint64_t *sums, uint32_t *counter, uint64_t *flag,
                                                       uint64_t seq) {
    __shared__ u32 sh[SC_RED_WORDS];
    Fr rv;
#pragma unroll
    for (int i = 0; i < 8; i++) rv.l[i] = r.l[i];
    // a narrow challenge (FrMul): v * r by the short product and v * (1 - r) = v - v * r — the same canonical value, 90 multiply-adds
    FrMul rm = frmul_prepare(rv);
    F29 omrp = fr29_prescale(fe_sub(Fr::one(), rv));
    Acc9 g0 = acc9_zero(), g1 = acc9_zero();
    size_t stride = (size_t)gridDim.x * 256;
    const uint32_t w0 = bit >> 6, s0 = bit & 63u, w1 = next_bit >> 6, s1 = next_bit & 63u;
    for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < n; j += stride) {
        Fr v = fe_load<FrParams>(t + 4 * j);
        uint64_t a = idx[2 * j + w0], b = w1 == w0 ? a : idx[2 * j + w1];
        if (rm.narrow) {
            Fr vr = frmul_apply(v, rm);
            v = ((a >> s0) & 1ull) ? vr : fe_sub(v, vr);
        } else {
            v = ((a >> s0) & 1ull) ? fr_mul29(v, rm.p) : fr_mul29(v, omrp);
        }
        fe_store(t + 4 * j, v);
        const bool one = (b >> s1) & 1ull;
        acc9_add_if(g0, v, !one);
        acc9_add_if(g1, v, one);
    }
    if (sums) {
        finish_round(g0, g1, sh, partials, sums, counter, flag, seq, ScRunArg{nullptr, 0, 0, 0});
    } else {
        store_block_pair(g0, g1, sh, partials);
    }
}

// sum of t[from, to) as the pair (sum, 0): the padding entries of LassoProver.eq_evals (:164-171) that `current_claim` includes
__global__ void __launch_bounds__(256) range_sum_kernel(const uint64_t *t, size_t from, size_t to, uint64_t *partials, uint64_t *sums,
                                                        uint32_t *counter, uint64_t *flag, uint64_t seq) {
    __shared__ u32 sh[SC_RED_WORDS];
    Acc9 g0 = acc9_zero(), g1 = acc9_zero();
    size_t stride = (size_t)gridDim.x * 256;
    for (size_t j = from + (size_t)blockIdx.x * 256 + threadIdx.x; j < to; j += stride) acc9_add(g0, fe_load<FrParams>(t + 4 * j));
    if (sums) {
        finish_round(g0, g1, sh, partials, sums, counter, flag, seq, ScRunArg{nullptr, 0, 0, 0});
    } else {
        store_block_pair(g0, g1, sh, partials);
    }
}

static unsigned env_uint(const char *name, unsigned dflt, unsigned lo, unsigned hi) {
    const char *e = getenv(name);
    unsigned v = e && *e ? (unsigned)atoi(e) : dflt;
    return v < lo ? lo : (v > hi ? hi : v);
}

// scratch layout of a fold/sums launch (u64 words): [0, 8*SC_MAX_BLOCKS) block pairs | SC_COUNTER_BYTES: arrival counters (sc_arrive)
// (must be 0 before a launch; the kernels re-arm it) | 8 words: the round's pair when it is not sent to a host mailbox
constexpr unsigned SC_MAX_BLOCKS = 2048;
constexpr size_t SC_SUMS_OFF = 8 * (size_t)SC_MAX_BLOCKS + SC_COUNTER_BYTES / 8;
constexpr size_t SC_MISC_BYTES = (SC_SUMS_OFF + 8) * 8;
// Grid of a 256-thread pair-sum kernel (raf / bit-split / range sums, dot products): one element per thread up to `cap` workgroups
// (default two per CU), grid-stride beyond. ZG_SC_MAX_BLOCKS overrides the cap of every sumcheck-family grid (tests: 1 = single block).
static unsigned sc_block_cap(unsigned dflt) {
    static const unsigned env = [] {
        const char *e = getenv("ZG_SC_MAX_BLOCKS");
        unsigned v = e && *e ? (unsigned)atoi(e) : 0u;
        return v > SC_MAX_BLOCKS ? SC_MAX_BLOCKS : v;
    }();
    return env ? env : dflt;
}
static unsigned sc_blocks(size_t half) {
    const unsigned cap = sc_block_cap(512u), b = div_up(half ? half : 1, 256);
    return b > cap ? cap : b;
}

// workgroups of a sums / fold launch over `half` pairs (the device-resident protocol's next launch is told: ScRunArg.nb_prev)
static unsigned sums_threads() {
    static const unsigned threads = env_uint("ZG_SC_SUMS_THREADS", 1024, 64, 1024) & ~63u;
    return threads;
}
static unsigned fold_threads() {
    static const unsigned threads = env_uint("ZG_SC_FOLD_THREADS", 512, 64, 512) & ~63u;
    return threads;
}
static unsigned sums_grid(size_t half, bool run) {
    const unsigned cap = sc_block_cap(256u);
    unsigned nb = div_up(half ? half : 1, (size_t)sums_threads() * SC_SUMS_U);
    if (nb > cap) nb = cap;
    return run && nb > SC_RUN_PAIRS ? SC_RUN_PAIRS : nb;
}
static unsigned fold_grid(size_t half, bool run) {
    const unsigned cap = sc_block_cap(512u);
    unsigned nb = div_up(half ? half : 1, fold_threads());
    if (nb > cap) nb = cap;
    return run && nb > SC_RUN_PAIRS ? SC_RUN_PAIRS : nb;
}

static int launch_sums(int layout, const uint64_t *t, size_t len, uint64_t *partials, uint64_t *sums, hipStream_t st,
                       uint64_t *flag = nullptr, uint64_t seq = 0, ScRunArg run = ScRunArg{nullptr, 0, 0, 0}) {
    size_t half = len / 2;
    // 1024-thread workgroups of SC_SUMS_U pairs per thread, at most one per CU (measured: tools/exp/fold_ab.hip)
    const unsigned threads = sums_threads(), nb = sums_grid(half, run.res != nullptr);
    uint32_t *counter = reinterpret_cast<uint32_t *>(partials + 8 * (size_t)SC_MAX_BLOCKS);  // zeroed at session creation
    prof_begin(ZG_PROF_SC_SUMS, st);
    if (layout == ZG_SC_HIGH_HALF)
        hipLaunchKernelGGL(sc_sums_kernel<ZG_SC_HIGH_HALF>, dim3(nb), dim3(threads), 0, st, t, half, partials, sums, counter, flag, seq, run);
    else
        hipLaunchKernelGGL(sc_sums_kernel<ZG_SC_LOW_PAIR>, dim3(nb), dim3(threads), 0, st, t, half, partials, sums, counter, flag, seq, run);
    prof_end(ZG_PROF_SC_SUMS, st);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

static int launch_fold(int layout, const uint64_t *t, size_t len, const uint64_t r[4], uint64_t *out, uint64_t *partials,
                       uint64_t *sums, hipStream_t st, uint64_t *flag = nullptr, uint64_t seq = 0,
                       ScRunArg run = ScRunArg{nullptr, 0, 0, 0}) {
    size_t half = len / 2;
    FrArg ra;
    for (int i = 0; i < 4; i++) {
        ra.l[2 * i] = r ? (uint32_t)r[i] : 0;
        ra.l[2 * i + 1] = r ? (uint32_t)(r[i] >> 32) : 0;
    }
    uint32_t *counter = reinterpret_cast<uint32_t *>(partials + 8 * (size_t)SC_MAX_BLOCKS);
    // 512-thread workgroups, at most two per CU (four waves per SIMD): measured at 2^16 .. 2^24 entries, tools/exp/fold_ab.hip
    const unsigned threads = fold_threads(), nb = fold_grid(half, run.res != nullptr);
    prof_begin(ZG_PROF_SC_FOLD, st);
    if (layout == ZG_SC_HIGH_HALF)
        hipLaunchKernelGGL(sc_fold_kernel<ZG_SC_HIGH_HALF>, dim3(nb), dim3(threads), 0, st, t, half, ra, out, partials, sums, counter, flag, seq, run);
    else
        hipLaunchKernelGGL(sc_fold_kernel<ZG_SC_LOW_PAIR>, dim3(nb), dim3(threads), 0, st, t, half, ra, out, partials, sums, counter, flag, seq, run);
    prof_end(ZG_PROF_SC_FOLD, st);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

static int eq_args_fill(EqArgs &a, const uint64_t *r_host, size_t v, const uint64_t *scale_host) {
    if (v > 34) {
        set_error("zg_fr_eq_table: v too large");
        return ZG_ERR_INVALID;
    }
    a.v = (int)v;
    a.has_scale = scale_host ? 1 : 0;
    for (size_t j = 0; j < v; j++)
        for (int i = 0; i < 4; i++) {
            a.r[j][2 * i] = (uint32_t)r_host[4 * j + i];
            a.r[j][2 * i + 1] = (uint32_t)(r_host[4 * j + i] >> 32);
        }
    for (int i = 0; i < 4; i++) {
        a.scale[2 * i] = scale_host ? (uint32_t)scale_host[i] : 0;
        a.scale[2 * i + 1] = scale_host ? (uint32_t)(scale_host[i] >> 32) : 0;
    }
    return ZG_OK;
}

// rows of 2^v_lo entries per workgroup: two workgroups per CU when the table is long enough (the factor prologue is per
// workgroup), never more than EQ_MAX_ROWS
static uint32_t eq_rows_per_block(uint32_t n_hi, uint32_t blocks_cap) {
    uint32_t nb = n_hi < blocks_cap ? n_hi : blocks_cap;
    uint32_t hpb = div_up(n_hi, nb);
    return hpb > (uint32_t)EQ_MAX_ROWS ? (uint32_t)EQ_MAX_ROWS : hpb;
}

// ASYNCHRONOUS: one launch on `st`, nothing of the caller's is read after the call returns (r and scale travel as kernel arguments)
static int eq_table_enqueue(const uint64_t *r_host, size_t v, const uint64_t *scale_host, uint64_t *d_out, hipStream_t st) {
    static EqArgs zero_args;  // (value-initialised once; the fill below overwrites what the kernel reads)
    EqArgs a = zero_args;
    ZG_TRY(eq_args_fill(a, r_host, v, scale_host));
    int v_lo = v < 8 ? (int)v : 8, v_hi = (int)v - v_lo;
    uint32_t n_hi = 1u << v_hi;
    // one block per CU up to 2^20 entries (the factor prologue is per block), two above; 512 threads: measured, profiles/r3f_eq_ab.txt
    static const uint32_t nb_env = env_uint("ZG_EQ_BLOCKS", 0, 0, 65536);
    const uint32_t nb_cap = nb_env ? nb_env : (n_hi <= 4096 ? 256u : 512u);
    uint32_t hpb = eq_rows_per_block(n_hi, nb_cap);
    // tables of >= 2^20 entries whose three middle variables are 128-bit challenges take the expanding kernel. Measured, narrow challenges,
    // back-to-back launches (tools/exp/archive/run_eq_expand_ab.sh, profiles/r5e_eq_expand_ab.txt): 2^24 entries 165 -> 139 us, 2^20 18.6 -> 17.6 us;
    // 2^16 7.6 -> 10.7 and 2^14 7.3 -> 9.8 us — a short table is a latency chain, and the expansion adds three dependent products to it.
    // ZG_EQ_EXPAND = 0: never; k >= 14: from 2^k entries on (the tests use 14 to run the kernel at small sizes).
    static const uint32_t expand_min = env_uint("ZG_EQ_EXPAND", 20, 0, 34);
    if (expand_min >= 14 && v >= expand_min) {
        const int xh = (int)v - 8 - EQ_XL;
        bool narrow = true;
        for (int j = 0; j < EQ_XL; j++) narrow = narrow && r_host[4 * (xh + j)] == 0 && r_host[4 * (xh + j) + 1] == 0;
        if (narrow) {
            const uint32_t xn = 1u << xh, xcap = nb_env ? nb_env : (xn <= 512 ? 256u : 512u);
            const uint32_t xhpb = eq_rows_per_block(xn, xcap), xwg = xhpb >= 2 ? 2u : 1u;
            prof_begin(ZG_PROF_EQ_TABLE, st);
            if (xwg == 2) hipLaunchKernelGGL(eq_expand_kernel<2>, dim3(div_up(xn, xhpb)), dim3(512), 0, st, a, xh, xhpb, d_out);
            else hipLaunchKernelGGL(eq_expand_kernel<1>, dim3(div_up(xn, xhpb)), dim3(256), 0, st, a, xh, xhpb, d_out);
            prof_end(ZG_PROF_EQ_TABLE, st);
            ZG_HIP(hipGetLastError());
            return ZG_OK;
        }
    }
    prof_begin(ZG_PROF_EQ_TABLE, st);
    static const uint32_t wg_env = env_uint("ZG_EQ_WG", 0, 0, 4);
    const uint32_t wg = wg_env ? wg_env : (n_hi <= 4096 ? 2u : 1u);
    if (wg >= 4 && hpb >= 4)
        hipLaunchKernelGGL(eq_main_kernel<4>, dim3(div_up(n_hi, hpb)), dim3(1024), 0, st, a, v_lo, v_hi, hpb, d_out);
    else if (wg >= 2 && hpb >= 2)
        hipLaunchKernelGGL(eq_main_kernel<2>, dim3(div_up(n_hi, hpb)), dim3(512), 0, st, a, v_lo, v_hi, hpb, d_out);
    else
        hipLaunchKernelGGL(eq_main_kernel<1>, dim3(div_up(n_hi, hpb)), dim3(256), 0, st, a, v_lo, v_hi, hpb, d_out);
    prof_end(ZG_PROF_EQ_TABLE, st);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

static int eq_prefix_enqueue(const uint64_t *tau_host, size_t v, uint64_t *d_out, hipStream_t st) {
    if (v > 24) {
        set_error("zg_fr_eq_prefix_tables: v too large");
        return ZG_ERR_INVALID;
    }
    Scratch s_r((v + 1) * 32);
    if (!s_r.p) return ZG_ERR_NOMEM;
    SyncGuard sync(st);
    uint64_t *d_r = s_r.as<uint64_t>();
    if (v) ZG_HIP(hipMemcpyAsync(d_r, tau_host, v * 32, hipMemcpyHostToDevice, st));
    uint64_t total = (2ull << v) - 1;
    prof_begin(ZG_PROF_EQ_TABLE, st);
    hipLaunchKernelGGL(eq_prefix_kernel, dim3((uint32_t)div_up(total, (uint64_t)64)), dim3(256), 0, st, d_r, (int)v, d_out);
    prof_end(ZG_PROF_EQ_TABLE, st);
    ZG_HIP(hipGetLastError());
    ZG_HIP(hipStreamSynchronize(st));  // tau_host / the temporary are released on return
    return ZG_OK;
}

// fused table/sums kernel of zg_sumcheck_open_spartan_dev (asynchronous, like eq_table_enqueue)
static int eq_spartan_enqueue(const uint64_t *r_host, size_t v, const uint64_t *scale_host, const uint64_t *d_az, const uint64_t *d_bz,
                              const uint64_t *d_cz, int layout, uint64_t *d_out, uint64_t *partials, uint64_t *sums, uint64_t *flag,
                              uint64_t seq, hipStream_t st) {
    static EqArgs zero_args;
    EqArgs a = zero_args;
    ZG_TRY(eq_args_fill(a, r_host, v, scale_host));
    int v_lo = v < 8 ? (int)v : 8, v_hi = (int)v - v_lo;
    uint32_t n_hi = 1u << v_hi;
    static const uint32_t nb_cap = env_uint("ZG_SC_SPARTAN_BLOCKS", 256, 1, SC_MAX_BLOCKS);  // one 512-thread workgroup per CU: the kernel is bound by its 3 products per entry, and the factor prologue is per workgroup (2^20: 256 -> 51.5 us, 512 -> 55, 1024 -> 67)
    uint32_t hpb = div_up(n_hi, n_hi < nb_cap ? n_hi : nb_cap);
    if (hpb > (uint32_t)EQ_MAX_ROWS) hpb = EQ_MAX_ROWS;
    uint32_t nb = div_up(n_hi, hpb);
    if (nb > SC_MAX_BLOCKS) {
        set_error("zg_sumcheck_open_spartan_dev: table too long");
        return ZG_ERR_INVALID;
    }
    uint32_t *counter = reinterpret_cast<uint32_t *>(partials + 8 * (size_t)SC_MAX_BLOCKS);
    static const uint32_t wg_env = env_uint("ZG_SC_SPARTAN_WG", 2, 1, 2);
    const bool wide = wg_env >= 2 && hpb >= 2;
    prof_begin(ZG_PROF_COMBINE, st);
    if (layout == ZG_SC_HIGH_HALF) {
        if (wide)
            hipLaunchKernelGGL((eq_spartan_kernel<ZG_SC_HIGH_HALF, 2>), dim3(nb), dim3(512), 0, st, a, v_lo, v_hi, hpb, d_az, d_bz, d_cz, d_out,
                               partials, sums, counter, flag, seq);
        else
            hipLaunchKernelGGL((eq_spartan_kernel<ZG_SC_HIGH_HALF, 1>), dim3(nb), dim3(256), 0, st, a, v_lo, v_hi, hpb, d_az, d_bz, d_cz, d_out,
                               partials, sums, counter, flag, seq);
    } else {
        if (wide)
            hipLaunchKernelGGL((eq_spartan_kernel<ZG_SC_LOW_PAIR, 2>), dim3(nb), dim3(512), 0, st, a, v_lo, v_hi, hpb, d_az, d_bz, d_cz, d_out,
                               partials, sums, counter, flag, seq);
        else
            hipLaunchKernelGGL((eq_spartan_kernel<ZG_SC_LOW_PAIR, 1>), dim3(nb), dim3(256), 0, st, a, v_lo, v_hi, hpb, d_az, d_bz, d_cz, d_out,
                               partials, sums, counter, flag, seq);
    }
    prof_end(ZG_PROF_COMBINE, st);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// out[i] = table[idx[i]]: the rows / columns a sparse-entry prover touches of a dense device table (RamReadWriteChecking's inc pairs
// and val_init checkpoints, src/zkvm/ram/read_write_checking.zig:431-446,591-602) — one small gather + one copy instead of the table
__global__ void __launch_bounds__(256) fr_gather_kernel(const uint64_t *table, const uint64_t *idx, size_t n, uint64_t *out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) fe_store(out + 4 * i, fe_load<FrParams>(table + 4 * idx[i]));
}

int gather_to_host(const uint64_t *d_table, size_t len, const uint64_t *idx_host, size_t n, uint64_t *out_host, hipStream_t st) {
    if (n == 0) return ZG_OK;
    for (size_t i = 0; i < n; i++)
        if (idx_host[i] >= len) {
            set_error("gather: index beyond the table's current length");
            return ZG_ERR_INVALID;
        }
    Scratch s_idx(n * 8), s_out(n * 32);
    if (!s_idx.p || !s_out.p) return ZG_ERR_NOMEM;
    SyncGuard sync(st);
    ZG_HIP(hipMemcpyAsync(s_idx.p, idx_host, n * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(fr_gather_kernel, dim3(div_up(n, 256)), dim3(256), 0, st, d_table, s_idx.as<uint64_t>(), n, s_out.as<uint64_t>());
    ZG_HIP(hipGetLastError());
    ZG_HIP(hipMemcpyAsync(out_host, s_out.p, n * 32, hipMemcpyDeviceToHost, st));
    ZG_HIP(hipStreamSynchronize(st));
    sync.dismiss();
    return ZG_OK;
}

}  // namespace zg

struct zg_sc_s {
    int device = -1;  // the HIP device the tables live on
    int layout = 0;
    size_t len = 0;
    uint64_t *buf[2] = {nullptr, nullptr};  // ping-pong tables (a fold cannot run in place across threads)
    uint64_t *own[2] = {nullptr, nullptr};  // ... as allocated. buf[i] == own[i] except while a session opened with
                                            // zg_sumcheck_open_dev_borrowed still reads the caller's table (buf[0]) — until its first bind
    bool borrowing = false;
    int cur = 0;
    uint64_t *d_partials = nullptr;
    uint64_t *h_pin = nullptr;  // pinned, device-visible: the kernels write the round sums (8 limbs) straight to the host
    bool sums_valid = false;
    hipStream_t st = nullptr;
    hipStream_t own_st = nullptr;  // zg_sumcheck_open (host table) runs on it: created with the session and kept in the pool, so the
                                   // independent provers of a batched sumcheck overlap (psc.hip: psc_open_stream has the measurements)
    uint64_t seq = 0;  // number of (sums) publications requested so far; h_pin[12] holds the last one completed
    size_t cap = 0, cap1 = 0;  // elements own[0] / own[1] can hold (sessions are pooled: hipMalloc / hipHostMalloc cost more than a round)
    // address-phase state of a Lasso session (zg_sumcheck_bit_round / bit_bind)
    bool bit_valid = false;     // bit_sums = the split of the first bit_n entries by bit bit_cached (left by the last bit_bind)
    unsigned bit_cached = 0;
    size_t bit_n = 0;
    const uint64_t *bit_idx = nullptr;
    uint64_t bit_sums[8] = {0};
    bool pad_valid = false;     // pad = sum of the entries [pad_from, len) — they do not change during the address phase
    size_t pad_from = 0;
    uint64_t pad[4] = {0};
    std::mutex mu;
};

using namespace zg;

static void sc_free(zg_sc_s *s) {
    if (!s) return;
    if (s->own_st) (void)hipStreamSynchronize(s->own_st);  // the tables go back to the device pool: nothing of this session may still run
    if (s->st && s->st != s->own_st) (void)hipDeviceSynchronize();  // (a caller's stream may be gone by now: wait for the device instead)
    void *ptrs[] = {s->own[0], s->own[1], s->d_partials};
    for (void *p : ptrs) pool_free(p);
    if (s->h_pin) (void)hipHostFree(s->h_pin);
    stream_release(s->own_st, s->device);
    delete s;
}

static std::mutex g_pool_mu;
static std::vector<zg_sc_s *> g_pool;  // closed sessions kept for reuse (at most 4)

// borrowed: the session will read its first table from the caller (zg_sumcheck_open_dev_borrowed): its own buffers hold the folds only
// (len / 2 and len / 4 entries)
static int sc_create(size_t len, int layout, hipStream_t st, zg_sc_s **out, bool borrowed = false) {
    if (len == 0 || (len & (len - 1)) || (layout != ZG_SC_HIGH_HALF && layout != ZG_SC_LOW_PAIR)) {
        set_error("zg_sumcheck_open: len must be a power of two and layout valid");
        return ZG_ERR_INVALID;
    }
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (size_t i = 0; i < g_pool.size(); i++) {
            // an ordinary session folds len -> own[0] ... a borrowed one only len / 2 -> own[1], len / 4 -> own[0]
            const size_t need0 = borrowed ? len / 4 : len, need1 = len / 2;
            if (g_pool[i]->device == current_device() && g_pool[i]->cap >= need0 && g_pool[i]->cap1 >= need1 && g_pool[i]->cap <= 4 * len) {
                zg_sc_s *s = g_pool[i];
                g_pool.erase(g_pool.begin() + i);
                s->layout = layout; s->len = len; s->st = st; s->cur = 0; s->sums_valid = false;
                s->buf[0] = s->own[0]; s->buf[1] = s->own[1]; s->borrowing = false;
                s->seq = 0; s->h_pin[12] = 0;
                s->bit_valid = false; s->pad_valid = false;
                if (!st) s->st = s->own_st;
                *out = s;
                return ZG_OK;
            }
        }
    }
    zg_sc_s *s = new zg_sc_s();
    s->device = current_device();
    s->cap = borrowed ? (len / 4 ? len / 4 : 1) : len;
    s->cap1 = len / 2 ? len / 2 : 1;
    s->layout = layout;
    s->len = len;
    s->own_st = stream_acquire();  // kept with the pooled session; from the runtime's free list (creating one costs ~3 ms)
    s->st = st ? st : s->own_st;
    hipError_t e = s->own_st ? hipSuccess : hipErrorOutOfMemory;
    // tables from the device pool (runtime.hip): a session of a size the session pool does not hold still reuses freed blocks
    auto grab = [&](uint64_t *&ptr, size_t bytes) {
        if (e == hipSuccess && !(ptr = reinterpret_cast<uint64_t *>(pool_alloc(bytes)))) e = hipErrorOutOfMemory;
    };
    grab(s->own[0], (borrowed ? (len / 4 ? len / 4 : 1) : len) * 32);
    grab(s->own[1], (len / 2 ? len / 2 : 1) * 32);
    s->buf[0] = s->own[0];
    s->buf[1] = s->own[1];
    grab(s->d_partials, SC_MISC_BYTES);
    if (e == hipSuccess) e = hipMemset(s->d_partials, 0, SC_MISC_BYTES);
    if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_pin, 128, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) s->h_pin[12] = 0;
    if (e != hipSuccess) {
        set_error(std::string("zg_sumcheck_open: ") + hipGetErrorString(e));
        sc_free(s);
        return e == hipErrorOutOfMemory ? ZG_ERR_NOMEM : ZG_ERR_HIP;
    }
    *out = s;
    return ZG_OK;
}

// the host side of a round that ended inside its kernel: spin on the mailbox's sequence word (written by the GPU after the sums,
// system-scope release); fall back to a stream synchronisation if it does not arrive promptly. The caller holds the session's mutex.
static int sc_wait_mailbox(zg_sc_s *s, uint64_t out[8]) {
    ZG_HIP(hipGetLastError());
    volatile uint64_t *flag = s->h_pin + 12;
    bool got = false;
    for (uint64_t spin = 0; spin < (1ull << 22); spin++) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == s->seq) { got = true; break; }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    if (!got) ZG_HIP(hipStreamSynchronize(s->st));
    for (int i = 0; i < 8; i++) out[i] = s->h_pin[i];
    return ZG_OK;
}

extern "C" {

int zg_fr_eq_table_dev(const uint64_t *r_host, size_t v, const uint64_t *scale_host, uint64_t *d_out, void *stream) {
    ZG_INIT();
    if (!d_out || (v && !r_host)) {
        set_error("zg_fr_eq_table_dev: invalid argument");
        return ZG_ERR_INVALID;
    }
    return eq_table_enqueue(r_host, v, scale_host, d_out, pick_stream(stream));
}

int zg_fr_eq_table(const uint64_t *r, size_t v, const uint64_t *scale, uint64_t *out) {
    ZG_INIT();
    if (!out || (v && !r) || v > 30) {
        set_error("zg_fr_eq_table: invalid argument");
        return ZG_ERR_INVALID;
    }
    size_t bytes = ((size_t)1 << v) * 32;
    Scratch s_out(bytes);
    if (!s_out.p) return ZG_ERR_NOMEM;
    uint64_t *d_out = s_out.as<uint64_t>();
    hipStream_t st = lib_stream();
    int rc = eq_table_enqueue(r, v, scale, d_out, st);
    hipError_t e = rc == ZG_OK ? hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, st) : hipSuccess;
    hipError_t e2 = hipStreamSynchronize(st);  // also after a failure: the scratch buffer goes back to the cache on return
    if (e == hipSuccess) e = e2;
    if (rc == ZG_OK && e != hipSuccess) {
        set_error(hipGetErrorString(e));
        rc = ZG_ERR_HIP;
    }
    return rc;
}

// eq+1(r, j) = 1 iff j = x + 1: over the boolean cube the only non-zero term of EqPlusOnePolynomial.mle's sum (src/poly/mod.zig:407-435)
// is the one whose flip position is the number of trailing zeros of j, and its factors are exactly those of eq(r, j - 1) (the low
// bits 10..0 of j against 01..1 of j - 1) — the table is the eq table moved up by one entry, with a zero in front.
static int eq_plus_one_enqueue(const uint64_t *r_host, size_t v, uint64_t *d_out, hipStream_t st) {
    size_t n = (size_t)1 << v;
    Scratch s_eq(n * 32);
    if (!s_eq.p) return ZG_ERR_NOMEM;
    SyncGuard sync(st);
    ZG_TRY(eq_table_enqueue(r_host, v, nullptr, s_eq.as<uint64_t>(), st));
    ZG_HIP(hipMemsetAsync(d_out, 0, 32, st));
    if (n > 1) ZG_HIP(hipMemcpyAsync(d_out + 4, s_eq.p, (n - 1) * 32, hipMemcpyDeviceToDevice, st));
    ZG_HIP(hipStreamSynchronize(st));
    return ZG_OK;
}

int zg_fr_eq_plus_one_table_dev(const uint64_t *r_host, size_t v, uint64_t *d_out, void *stream) {
    ZG_INIT();
    if (!d_out || (v && !r_host) || v > 30) {
        set_error("zg_fr_eq_plus_one_table_dev: invalid argument");
        return ZG_ERR_INVALID;
    }
    return eq_plus_one_enqueue(r_host, v, d_out, pick_stream(stream));
}

int zg_fr_eq_plus_one_table(const uint64_t *r, size_t v, uint64_t *out) {
    ZG_INIT();
    if (!out || (v && !r) || v > 30) {
        set_error("zg_fr_eq_plus_one_table: invalid argument");
        return ZG_ERR_INVALID;
    }
    size_t bytes = ((size_t)1 << v) * 32;
    Scratch s_out(bytes);
    if (!s_out.p) return ZG_ERR_NOMEM;
    int rc = eq_plus_one_enqueue(r, v, s_out.as<uint64_t>(), lib_stream());
    if (rc == ZG_OK) {
        hipError_t e = hipMemcpy(out, s_out.p, bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            set_error(hipGetErrorString(e));
            rc = ZG_ERR_HIP;
        }
    }
    return rc;
}

// LtPolynomial.evaluateAtIndex over the whole cube (src/zkvm/ram/val_evaluation.zig:309-330): lt(j) = sum over the zero bits i of j of
// r[i] * prod_{k > i} (j_k ? r[k] : 1 - r[k]) — the index's bit i belongs to r[i]. One thread per index walks the bits from the top with
// the running suffix product: 2 v products.
struct LtArgs {
    FrArg r[30];
};
// variables [lo, hi) of the point, index bit (i - lo) <-> r[i]: lt_out[j] = sum over the zero bits i of j of r_i * prod_{k > i} eq(r_k, j_k),
// eq_out[j] (optional) = prod over all the range's bits of eq(r_k, j_k)
__global__ void __launch_bounds__(256) lt_table_kernel(LtArgs a, int lo, int hi, size_t n, uint64_t *lt_out, uint64_t *eq_out) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        Fr suffix = Fr::one(), result = Fr::zero();
        for (int i = hi - 1; i >= lo; i--) {
            Fr ri;
#pragma unroll
            for (int l = 0; l < 8; l++) ri.l[l] = a.r[i].l[l];
            if ((j >> (i - lo)) & 1) {
                suffix = fr_mul29v(suffix, ri);
            } else {
                result = fe_add(result, fr_mul29v(ri, suffix));
                suffix = fr_mul29v(suffix, fe_sub(Fr::one(), ri));
            }
        }
        fe_store(lt_out + 4 * j, result);
        if (eq_out) fe_store(eq_out + 4 * j, suffix);
    }
}
// lt(j) = lt_hi(j_hi) + eq_hi(j_hi) * lt_lo(j_lo): one product per entry over three 2^(v/2)-entry factor tables (the direct form above
// is ~2v products per entry: 0.42 ms for 2^20 entries in the trace of the standard Stage 4)
__global__ void __launch_bounds__(256) lt_combine_kernel(const uint64_t *lt_hi, const uint64_t *eq_hi, const uint64_t *lt_lo, int h, size_t n, uint64_t *out) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t mask = ((size_t)1 << h) - 1;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const size_t jh = j >> h, jl = j & mask;
        Fr t = fr_mul29v(fe_load<FrParams>(eq_hi + 4 * jh), fe_load<FrParams>(lt_lo + 4 * jl));
        fe_store(out + 4 * j, fe_add(fe_load<FrParams>(lt_hi + 4 * jh), t));
    }
}
constexpr size_t LT_FACTORED_MIN_VARS = 12;
static int lt_table_enqueue(const uint64_t *r_host, size_t v, uint64_t *d_out, hipStream_t st) {
    LtArgs a = {};
    for (size_t i = 0; i < v; i++)
        for (int l = 0; l < 4; l++) {
            a.r[i].l[2 * l] = (uint32_t)r_host[4 * i + l];
            a.r[i].l[2 * l + 1] = (uint32_t)(r_host[4 * i + l] >> 32);
        }
    const size_t n = (size_t)1 << v;
    unsigned nb = (unsigned)div_up(n, 256);
    if (nb > 8192) nb = 8192;
    if (v < LT_FACTORED_MIN_VARS) {
        hipLaunchKernelGGL(lt_table_kernel, dim3(nb), dim3(256), 0, st, a, 0, (int)v, n, d_out, (uint64_t *)nullptr);
        ZG_HIP(hipGetLastError());
        return ZG_OK;
    }
    const size_t h = v / 2, n_lo = (size_t)1 << h, n_hi = (size_t)1 << (v - h);
    Scratch s_f((2 * n_hi + n_lo) * 32);
    if (!s_f.p) return ZG_ERR_NOMEM;
    SyncGuard sync(st);
    uint64_t *lt_hi = s_f.as<uint64_t>(), *eq_hi = lt_hi + 4 * n_hi, *lt_lo = eq_hi + 4 * n_hi;
    hipLaunchKernelGGL(lt_table_kernel, dim3(div_up(n_hi, 256)), dim3(256), 0, st, a, (int)h, (int)v, n_hi, lt_hi, eq_hi);
    hipLaunchKernelGGL(lt_table_kernel, dim3(div_up(n_lo, 256)), dim3(256), 0, st, a, 0, (int)h, n_lo, lt_lo, (uint64_t *)nullptr);
    hipLaunchKernelGGL(lt_combine_kernel, dim3(nb), dim3(256), 0, st, lt_hi, eq_hi, lt_lo, (int)h, n, d_out);
    ZG_HIP(hipGetLastError());
    ZG_HIP(hipStreamSynchronize(st));  // the factor tables go back to the scratch cache with this call
    sync.dismiss();
    return ZG_OK;
}

int zg_fr_lt_table_dev(const uint64_t *r_host, size_t v, uint64_t *d_out, void *stream) {
    ZG_INIT();
    if (!d_out || (v && !r_host) || v > 30) {
        set_error("zg_fr_lt_table_dev: invalid argument (at most 30 variables)");
        return ZG_ERR_INVALID;
    }
    return lt_table_enqueue(r_host, v, d_out, pick_stream(stream));
}

int zg_fr_lt_table(const uint64_t *r, size_t v, uint64_t *out) {
    ZG_INIT();
    if (!out || (v && !r) || v > 30) {
        set_error("zg_fr_lt_table: invalid argument (at most 30 variables)");
        return ZG_ERR_INVALID;
    }
    const size_t bytes = ((size_t)1 << v) * 32;
    Scratch s_out(bytes);
    if (!s_out.p) return ZG_ERR_NOMEM;
    hipStream_t st = lib_stream();
    int rc = lt_table_enqueue(r, v, s_out.as<uint64_t>(), st);
    hipError_t e = rc == ZG_OK ? hipMemcpyAsync(out, s_out.p, bytes, hipMemcpyDeviceToHost, st) : hipSuccess;
    hipError_t e2 = hipStreamSynchronize(st);
    if (e == hipSuccess) e = e2;
    if (rc == ZG_OK && e != hipSuccess) {
        set_error(hipGetErrorString(e));
        rc = ZG_ERR_HIP;
    }
    return rc;
}

// ValEvaluation's inc and wa from the list of writes (src/zkvm/ram/val_evaluation.zig:298-345 as the prover's stage 4 builds them,
// prover.zig:760-800): inc[cycle] = F.fromU64(post) - F.fromU64(pre), wa[cycle] = eq(r_address, word), zero elsewhere
__global__ void __launch_bounds__(256) write_tables_kernel(const uint32_t *cycle, const uint32_t *word, const uint64_t *pre, const uint64_t *post, uint32_t m,
                                                           const uint64_t *eq, uint32_t eq_mask, uint64_t *inc, uint64_t *wa) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    F29 r2p;
#pragma unroll
    for (int k = 0; k < 9; k++) r2p.l[k] = Fr29::R2PRE[k];
    Fr a = Fr::zero(), b = Fr::zero();
    const uint64_t hi = post[i], lo = pre[i];
    a.l[0] = (uint32_t)hi;
    a.l[1] = (uint32_t)(hi >> 32);
    b.l[0] = (uint32_t)lo;
    b.l[1] = (uint32_t)(lo >> 32);
    if (hi) a = fr_mul29(a, r2p);  // F.fromU64
    if (lo) b = fr_mul29(b, r2p);
    const size_t c = cycle[i];
    fe_store(inc + 4 * c, fe_sub(a, b));
    fe_store(wa + 4 * c, fe_load<FrParams>(eq + 4 * (size_t)(word[i] & eq_mask)));
}

int zg_fr_write_tables_dev(size_t n, size_t m, const uint32_t *cycle, const uint32_t *word, const uint64_t *pre, const uint64_t *post, const uint64_t *r_eq,
                           size_t log_k, uint64_t *d_inc, uint64_t *d_wa, void *stream) {
    ZG_INIT();
    if (!d_inc || !d_wa || n == 0 || log_k > 26 || (log_k && !r_eq) || m > ((size_t)1 << 30) || (m && (!cycle || !word || !pre || !post))) {
        set_error("zg_fr_write_tables_dev: invalid argument");
        return ZG_ERR_INVALID;
    }
    {
        std::vector<bool> seen(n, false);  // two writes to one cycle would race in the scatter: the caller keeps the later one
        for (size_t i = 0; i < m; i++) {
            if (cycle[i] >= n || seen[cycle[i]]) {
                set_error(cycle[i] >= n ? "zg_fr_write_tables_dev: a write beyond the tables" : "zg_fr_write_tables_dev: two writes in one cycle");
                return ZG_ERR_INVALID;
            }
            if (word[i] >= ((size_t)1 << log_k)) {  // the reference skips such a write; the kernel would fold it onto word mod K: refuse, the caller keeps its loop
                set_error("zg_fr_write_tables_dev: a word index beyond 2^log_k");
                return ZG_ERR_INVALID;
            }
            seen[cycle[i]] = true;
        }
    }
    hipStream_t st = pick_stream(stream);
    const size_t m8 = (m + 7) & ~(size_t)7, K = (size_t)1 << log_k;
    Scratch s_eq(K * 32), s_w(m8 * 24 + 8);
    if (!s_eq.p || !s_w.p) return ZG_ERR_NOMEM;
    SyncGuard sync(st);
    ZG_TRY(eq_table_enqueue(r_eq, log_k, nullptr, s_eq.as<uint64_t>(), st));
    ZG_HIP(hipMemsetAsync(d_inc, 0, n * 32, st));
    ZG_HIP(hipMemsetAsync(d_wa, 0, n * 32, st));
    if (m) {
        uint64_t *d_pre = s_w.as<uint64_t>(), *d_post = d_pre + m8;
        uint32_t *d_cyc = reinterpret_cast<uint32_t *>(d_post + m8), *d_word = d_cyc + m8;
        ZG_HIP(hipMemcpyAsync(d_pre, pre, m * 8, hipMemcpyHostToDevice, st));
        ZG_HIP(hipMemcpyAsync(d_post, post, m * 8, hipMemcpyHostToDevice, st));
        ZG_HIP(hipMemcpyAsync(d_cyc, cycle, m * 4, hipMemcpyHostToDevice, st));
        ZG_HIP(hipMemcpyAsync(d_word, word, m * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(write_tables_kernel, dim3(div_up(m, 256)), dim3(256), 0, st, d_cyc, d_word, d_pre, d_post, (uint32_t)m, s_eq.as<uint64_t>(),
                           (uint32_t)(K - 1), d_inc, d_wa);
        ZG_HIP(hipGetLastError());
    }
    ZG_HIP(hipStreamSynchronize(st));  // the write list and the eq table are the caller's / scratch
    sync.dismiss();
    return ZG_OK;
}

int zg_fr_eq_prefix_tables_dev(const uint64_t *tau_host, size_t v, uint64_t *d_out, void *stream) {
    ZG_INIT();
    if (!d_out || (v && !tau_host)) {
        set_error("zg_fr_eq_prefix_tables_dev: invalid argument");
        return ZG_ERR_INVALID;
    }
    return eq_prefix_enqueue(tau_host, v, d_out, pick_stream(stream));
}

int zg_fr_eq_prefix_tables(const uint64_t *tau, size_t v, uint64_t *out) {
    ZG_INIT();
    if (!out || (v && !tau) || v > 24) {
        set_error("zg_fr_eq_prefix_tables: invalid argument");
        return ZG_ERR_INVALID;
    }
    size_t bytes = (((size_t)2 << v) - 1) * 32;
    Scratch s_out(bytes);
    if (!s_out.p) return ZG_ERR_NOMEM;
    uint64_t *d_out = s_out.as<uint64_t>();
    int rc = eq_prefix_enqueue(tau, v, d_out, lib_stream());
    if (rc == ZG_OK) {
        hipError_t e = hipMemcpy(out, d_out, bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            set_error(hipGetErrorString(e));
            rc = ZG_ERR_HIP;
        }
    }
    return rc;
}

static int bind_host(int layout, const uint64_t *table, size_t len, const uint64_t r[4], uint64_t *out) {
    if (!table || !r || !out || len < 2 || (len & (len - 1))) {
        set_error("zg_fr_bind_*: len must be a power of two >= 2");
        return ZG_ERR_INVALID;
    }
    hipStream_t st = lib_stream();
    Scratch s_t(len * 32), s_o(len / 2 * 32), s_misc(SC_MISC_BYTES);
    if (!s_t.p || !s_o.p || !s_misc.p) return ZG_ERR_NOMEM;
    uint64_t *d_t = s_t.as<uint64_t>(), *d_o = s_o.as<uint64_t>(), *d_misc = s_misc.as<uint64_t>();
    uint64_t *d_sums = d_misc + SC_SUMS_OFF;
    ZG_HIP(hipMemsetAsync(d_misc + 8 * (size_t)SC_MAX_BLOCKS, 0, SC_COUNTER_BYTES, st));
    ZG_HIP(hipMemcpyAsync(d_t, table, len * 32, hipMemcpyHostToDevice, st));
    int rc = launch_fold(layout, d_t, len, r, d_o, d_misc, d_sums, st);
    if (rc == ZG_OK) {
        hipError_t e = hipMemcpyAsync(out, d_o, len / 2 * 32, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            set_error(hipGetErrorString(e));
            rc = ZG_ERR_HIP;
        }
    }
    return rc;
}

int zg_fr_bind_low(uint64_t *table, size_t len, const uint64_t r[4]) {
    ZG_INIT();
    return bind_host(ZG_SC_LOW_PAIR, table, len, r, table);
}

int zg_fr_bind_high(const uint64_t *table, size_t len, const uint64_t r[4], uint64_t *out) {
    ZG_INIT();
    return bind_host(ZG_SC_HIGH_HALF, table, len, r, out);
}

int zg_fr_dense_evaluate(const uint64_t *evals, size_t num_vars, const uint64_t *point, uint64_t out[4]) {
    ZG_INIT();
    if (!evals || !out || (num_vars && !point) || num_vars > 30) {
        set_error("zg_fr_dense_evaluate: invalid argument");
        return ZG_ERR_INVALID;
    }
    size_t n = (size_t)1 << num_vars;
    hipStream_t st = lib_stream();
    // the eq table's index MSB pairs with r[0]; evaluate() pairs index bit j with point[j]: reverse the point
    std::vector<uint64_t> rev(4 * (num_vars ? num_vars : 1));
    for (size_t j = 0; j < num_vars; j++)
        for (int l = 0; l < 4; l++) rev[4 * j + l] = point[4 * (num_vars - 1 - j) + l];
    Scratch s_ev(n * 32), s_eq(n * 32), s_misc(SC_MISC_BYTES);
    if (!s_ev.p || !s_eq.p || !s_misc.p) return ZG_ERR_NOMEM;
    uint64_t *d_ev = s_ev.as<uint64_t>(), *d_eq = s_eq.as<uint64_t>(), *d_misc = s_misc.as<uint64_t>();
    ZG_HIP(hipMemcpyAsync(d_ev, evals, n * 32, hipMemcpyHostToDevice, st));
    int rc = eq_table_enqueue(rev.data(), num_vars, nullptr, d_eq, st);
    if (rc == ZG_OK) {
        unsigned nb = sc_blocks(n);
        hipLaunchKernelGGL(fr_dot_kernel, dim3(nb), dim3(256), 0, st, d_ev, d_eq, n, d_misc);
        hipLaunchKernelGGL(sc_finish_kernel, dim3(1), dim3(256), 0, st, d_misc, nb, d_misc + SC_SUMS_OFF, (uint64_t *)nullptr, (uint64_t)0);
        uint64_t h[4];
        hipError_t e = hipMemcpyAsync(h, d_misc + SC_SUMS_OFF, 32, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            set_error(hipGetErrorString(e));
            rc = ZG_ERR_HIP;
        } else {
            for (int l = 0; l < 4; l++) out[l] = h[l];
        }
    }
    if (rc != ZG_OK) (void)hipStreamSynchronize(st);
    return rc;
}

int zg_fr_rows_mle_dev(const uint64_t *d_rows, size_t n_rows, size_t k, const uint64_t *r_host, size_t v, void *stream, uint64_t *out) {
    ZG_INIT();
    if (!out || k == 0 || k > ROWS_MLE_MAX_K || v > 30 || (v && !r_host) || (n_rows && !d_rows)) {
        set_error("zg_fr_rows_mle: 1..64 columns, at most 30 variables");
        return ZG_ERR_INVALID;
    }
    const size_t full = (size_t)1 << v;
    if (n_rows > full) n_rows = full;  // rows past the hypercube have no eq value
    hipStream_t st = pick_stream(stream);
    // blocks: a multiple of k / gcd(k, 256) so that the thread count is a multiple of k; enough threads for one element each, at most ~1024 blocks
    unsigned unit = (unsigned)k;
    for (unsigned d = 2; d <= 256 && unit % 2 == 0; d *= 2) unit /= 2;  // k / gcd(k, 256)
    size_t want = (n_rows * k + 255) / 256;
    if (want > 1024) want = 1024;
    unsigned nb = (unsigned)((want + unit - 1) / unit) * unit;
    if (nb < unit) nb = unit;
    const size_t threads = (size_t)nb * 256;
    Scratch s_eq(full * 32), s_part(threads * 32), s_out(k * 32);
    if (!s_eq.p || !s_part.p || !s_out.p) return ZG_ERR_NOMEM;
    SyncGuard sync(st);
    ZG_TRY(eq_table_enqueue(r_host, v, nullptr, s_eq.as<uint64_t>(), st));
    hipLaunchKernelGGL(rows_mle_kernel, dim3(nb), dim3(256), 0, st, d_rows, n_rows, (uint32_t)k, s_eq.as<uint64_t>(), s_part.as<uint64_t>());
    hipLaunchKernelGGL(rows_mle_finish_kernel, dim3((unsigned)k), dim3(256), 0, st, s_part.as<uint64_t>(), threads, (uint32_t)k, s_out.as<uint64_t>());
    ZG_HIP(hipGetLastError());
    ZG_HIP(hipMemcpyAsync(out, s_out.p, k * 32, hipMemcpyDeviceToHost, st));
    ZG_HIP(hipStreamSynchronize(st));
    sync.dismiss();
    return ZG_OK;
}

int zg_fr_rows_mle(const uint64_t *rows, size_t n_rows, size_t k, const uint64_t *r, size_t v, uint64_t *out) {
    ZG_INIT();
    if (!out || k == 0 || k > ROWS_MLE_MAX_K || v > 30 || (v && !r) || (n_rows && !rows)) {
        set_error("zg_fr_rows_mle: 1..64 columns, at most 30 variables");
        return ZG_ERR_INVALID;
    }
    const size_t full = (size_t)1 << v;
    if (n_rows > full) n_rows = full;
    hipStream_t st = lib_stream();
    Scratch s_rows((n_rows ? n_rows : 1) * k * 32);
    if (!s_rows.p) return ZG_ERR_NOMEM;
    SyncGuard sync(st);
    if (n_rows) ZG_HIP(hipMemcpyAsync(s_rows.p, rows, n_rows * k * 32, hipMemcpyHostToDevice, st));
    return zg_fr_rows_mle_dev(s_rows.as<uint64_t>(), n_rows, k, r, v, st, out);
}

// nout <= 16 affine maps of the rows; output c goes to tab[c] + 4 * (i * g + c % g) for row i
static int rows_affine_launch(const uint64_t *d_rows, size_t n_rows, size_t k, size_t stride, const uint64_t *coeffs, size_t nout, uint64_t *const *tab_of_output,
                              size_t g, size_t n_pad, hipStream_t st) {
    {
        if (n_pad == 0) return ZG_OK;
    }
    // per output: the columns with a non-zero coefficient
    std::vector<uint8_t> cols(64 * nout, 0);
    RowsAffineArgs a{};
    for (size_t c = 0; c < nout; c++) {
        if (!tab_of_output[c]) {
            set_error("zg_fr_rows_affine: null table");
            return ZG_ERR_INVALID;
        }
        a.tab[c] = tab_of_output[c];
        unsigned nnz = 0;
        for (size_t col = 0; col < k; col++) {
            const uint64_t *e = coeffs + 4 * (c * (k + 1) + col);
            if (e[0] | e[1] | e[2] | e[3]) {
                if (nnz == ROWS_AFFINE_MAX_NNZ) {
                    set_error("zg_fr_rows_affine: at most 64 non-zero coefficients per map");
                    return ZG_ERR_INVALID;
                }
                cols[64 * c + nnz++] = (uint8_t)col;
            }
        }
        a.nnz[c] = (uint8_t)nnz;
    }
    const size_t n_coeff = nout * (k + 1);
    Scratch s_coeff(n_coeff * 32), s_pre(n_coeff * 36), s_cols(64 * nout);
    if (!s_coeff.p || !s_pre.p || !s_cols.p) return ZG_ERR_NOMEM;
    SyncGuard sync(st);
    ZG_HIP(hipMemcpyAsync(s_coeff.p, coeffs, n_coeff * 32, hipMemcpyHostToDevice, st));
    ZG_HIP(hipMemcpyAsync(s_cols.p, cols.data(), cols.size(), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(rows_affine_prep_kernel, dim3(div_up(n_coeff, 256)), dim3(256), 0, st, s_coeff.as<uint64_t>(), (uint32_t)n_coeff, s_pre.as<uint32_t>());
    hipLaunchKernelGGL(rows_affine_kernel, dim3(div_up(n_pad, 64)), dim3((unsigned)(64 * nout)), 0, st, d_rows, n_rows, (uint32_t)k, (uint32_t)stride, s_coeff.as<uint64_t>(),
                       s_pre.as<uint32_t>(), s_cols.as<uint8_t>(), a, (uint32_t)g, n_pad);
    ZG_HIP(hipGetLastError());
    ZG_HIP(hipStreamSynchronize(st));  // the coefficient buffers go back to the cache; `cols` is a local
    sync.dismiss();
    return ZG_OK;
}

int zg_fr_rows_affine_dev(const uint64_t *d_rows, size_t n_rows, size_t k, size_t stride, const uint64_t *coeffs, size_t ntab, size_t g, size_t n_pad,
                          uint64_t *const *d_tables, void *stream) {
    ZG_INIT();
    const size_t nout = ntab * g;
    if (stride == 0) stride = k;
    if (!coeffs || !d_tables || k == 0 || k > ROWS_AFFINE_MAX_K || stride > k || ntab == 0 || g == 0 || nout > ROWS_AFFINE_MAX_OUT || n_pad < n_rows ||
        (n_rows && !d_rows)) {
        set_error("zg_fr_rows_affine: 1..128 columns, stride <= columns, 1..16 outputs (tables x interleave), n_pad >= n_rows");
        return ZG_ERR_INVALID;
    }
    uint64_t *tabs[ROWS_AFFINE_MAX_OUT];
    for (size_t c = 0; c < nout; c++) tabs[c] = d_tables[c / g];
    return rows_affine_launch(d_rows, n_rows, k, stride, coeffs, nout, tabs, g, n_pad, pick_stream(stream));
}

// Records of `record` elements per row: out[i * record + first + c] = map_c(row_i) for c < nout <= 16 consecutive positions of the record.
// JoltR1CS.computeAz / computeBz (src/zkvm/r1cs/jolt_r1cs.zig:143-190) lay the 19 uniform constraints of a cycle out this way
// (constraint_idx = cycle * 19 + i): two calls per vector (constraints 0..15, then 16..18) write it in place, cycle by cycle.
int zg_fr_rows_affine_records_dev(const uint64_t *d_rows, size_t n_rows, size_t k, size_t stride, const uint64_t *coeffs, size_t nout, size_t record,
                                  size_t first, uint64_t *d_out, void *stream) {
    ZG_INIT();
    if (stride == 0) stride = k;
    if (!coeffs || !d_out || k == 0 || k > ROWS_AFFINE_MAX_K || stride > k || nout == 0 || nout > ROWS_AFFINE_MAX_OUT || first + nout > record || (n_rows && !d_rows)) {
        set_error("zg_fr_rows_affine_records: 1..128 columns, stride <= columns, 1..16 outputs inside the record");
        return ZG_ERR_INVALID;
    }
    uint64_t *tabs[ROWS_AFFINE_MAX_OUT];
    for (size_t c = 0; c < nout; c++) tabs[c] = d_out + 4 * first;  // the kernel adds i * record + c (c % record = c: nout <= record)
    return rows_affine_launch(d_rows, n_rows, k, stride, coeffs, nout, tabs, record, n_rows, pick_stream(stream));
}

int zg_fr_rows_affine_prodsum_dev(const uint64_t *d_rows, size_t n_rows, size_t k, size_t stride, const uint64_t *coeffs, size_t npairs,
                                  const uint64_t *d_weights, size_t g, uint64_t *out, void *stream) {
    ZG_INIT();
    if (stride == 0) stride = k;
    if (!coeffs || !out || k == 0 || k > ROWS_AFFINE_MAX_K || stride > k || npairs == 0 || npairs > ROWS_PS_MAX_PAIRS || g == 0 ||
        (n_rows && (!d_rows || !d_weights))) {
        set_error("zg_fr_rows_affine_prodsum: 1..128 columns, stride <= columns, 1..32 pairs, a weight interleave >= 1");
        return ZG_ERR_INVALID;
    }
    if (n_rows == 0) {
        for (size_t i = 0; i < 4 * npairs; i++) out[i] = 0;
        return ZG_OK;
    }
    hipStream_t st = pick_stream(stream);
    const size_t nout = 2 * npairs;
    std::vector<uint8_t> cols(64 * nout, 0);
    RowsProdSumArgs a{};
    for (size_t c = 0; c < nout; c++) {
        unsigned nnz = 0;
        for (size_t col = 0; col < k; col++) {
            const uint64_t *e = coeffs + 4 * (c * (k + 1) + col);
            if (e[0] | e[1] | e[2] | e[3]) {
                if (nnz == ROWS_AFFINE_MAX_NNZ) {
                    set_error("zg_fr_rows_affine_prodsum: at most 64 non-zero coefficients per map");
                    return ZG_ERR_INVALID;
                }
                cols[64 * c + nnz++] = (uint8_t)col;
            }
        }
        a.nnz[c] = (uint8_t)nnz;
    }
    const size_t n_coeff = nout * (k + 1);
    unsigned nb = div_up(n_rows, 64);
    if (nb > 1024) nb = 1024;
    Scratch s_coeff(n_coeff * 32), s_pre(n_coeff * 36), s_small(n_coeff * 4), s_cols(64 * nout), s_part((size_t)nb * npairs * 32), s_out(npairs * 32);
    if (!s_coeff.p || !s_pre.p || !s_small.p || !s_cols.p || !s_part.p || !s_out.p) return ZG_ERR_NOMEM;
    SyncGuard sync(st);
    ZG_HIP(hipMemcpyAsync(s_coeff.p, coeffs, n_coeff * 32, hipMemcpyHostToDevice, st));
    ZG_HIP(hipMemcpyAsync(s_cols.p, cols.data(), cols.size(), hipMemcpyHostToDevice, st));
    const bool small_ok = env_uint("ZG_ROWS_SMALL_COEFF", 1, 0, 1) != 0;  // 0: every term through the general product (A/B, tests); read per call
    hipLaunchKernelGGL(rows_affine_prep_kernel, dim3(div_up(n_coeff, 256)), dim3(256), 0, st, s_coeff.as<uint64_t>(), (uint32_t)n_coeff, s_pre.as<uint32_t>(),
                       s_small.as<uint32_t>());
    if (!small_ok) ZG_HIP(hipMemsetAsync(s_small.p, 0, n_coeff * 4, st));
    // the row tile in LDS when it fits (k <= 73 columns: the R1CS witness has 43); ZG_ROWS_STAGE=0 reads the rows from memory per term
    const size_t tile_bytes = (size_t)k * ROWS_TILE_COL * 16;
    if (tile_bytes <= 150 * 1024 && env_uint("ZG_ROWS_STAGE", 1, 0, 1)) {
        static PerDeviceOnce once;
        ZG_HIP(once.run([] {
            return hipFuncSetAttribute(reinterpret_cast<const void *>(rows_affine_prodsum_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        }));
        const unsigned per_wave = div_up(npairs, ROWS_PS_STAGED_WAVES), waves = div_up(npairs, per_wave);  // 18 pairs: nine waves of two
        hipLaunchKernelGGL(rows_affine_prodsum_kernel<true>, dim3(nb), dim3(64 * waves), tile_bytes, st, d_rows, n_rows,
                           (uint32_t)k, (uint32_t)stride, s_coeff.as<uint64_t>(), s_pre.as<uint32_t>(), s_small.as<uint32_t>(), s_cols.as<uint8_t>(), a, d_weights,
                           (uint32_t)g, (uint32_t)npairs, s_part.as<uint64_t>());
    } else {
        hipLaunchKernelGGL(rows_affine_prodsum_kernel<false>, dim3(nb, div_up(npairs, ROWS_PS_WAVES)), dim3(64 * ROWS_PS_WAVES), 0, st, d_rows, n_rows,
                           (uint32_t)k, (uint32_t)stride, s_coeff.as<uint64_t>(), s_pre.as<uint32_t>(), s_small.as<uint32_t>(), s_cols.as<uint8_t>(), a, d_weights,
                           (uint32_t)g, (uint32_t)npairs, s_part.as<uint64_t>());
    }
    hipLaunchKernelGGL(rows_prodsum_finish_kernel, dim3((unsigned)npairs), dim3(256), 0, st, s_part.as<uint64_t>(), nb, (uint32_t)npairs, s_out.as<uint64_t>());
    ZG_HIP(hipGetLastError());
    ZG_HIP(hipMemcpyAsync(out, s_out.p, npairs * 32, hipMemcpyDeviceToHost, st));
    ZG_HIP(hipStreamSynchronize(st));
    sync.dismiss();
    return ZG_OK;
}

int zg_fr_rows_affine(const uint64_t *rows, size_t n_rows, size_t k, size_t stride, const uint64_t *coeffs, size_t ntab, size_t g, size_t n_pad,
                      uint64_t *const *tables) {
    ZG_INIT();
    if (stride == 0) stride = k;
    if (!tables || ntab == 0 || ntab > ROWS_AFFINE_MAX_OUT || g == 0 || k == 0 || k > ROWS_AFFINE_MAX_K || stride > k || n_pad < n_rows || (n_rows && !rows)) {
        set_error("zg_fr_rows_affine: 1..128 columns, stride <= columns, 1..16 outputs (tables x interleave), n_pad >= n_rows");
        return ZG_ERR_INVALID;
    }
    if (n_pad == 0) return ZG_OK;
    hipStream_t st = lib_stream();
    // the matrix holds (n_rows - 1) * stride + k elements: the last row's window ends there
    const size_t n_elems = n_rows ? (n_rows - 1) * stride + k : 1;
    Scratch s_rows(n_elems * 32), s_out(ntab * n_pad * g * 32);
    if (!s_rows.p || !s_out.p) return ZG_ERR_NOMEM;
    SyncGuard sync(st);
    if (n_rows) ZG_HIP(hipMemcpyAsync(s_rows.p, rows, n_elems * 32, hipMemcpyHostToDevice, st));
    uint64_t *d_tab[ROWS_AFFINE_MAX_OUT];
    for (size_t t = 0; t < ntab; t++) d_tab[t] = s_out.as<uint64_t>() + 4 * t * n_pad * g;
    ZG_TRY(zg_fr_rows_affine_dev(s_rows.as<uint64_t>(), n_rows, k, stride, coeffs, ntab, g, n_pad, d_tab, st));
    for (size_t t = 0; t < ntab; t++) {
        if (!tables[t]) {
            set_error("zg_fr_rows_affine: null table");
            return ZG_ERR_INVALID;
        }
        ZG_HIP(hipMemcpyAsync(tables[t], d_tab[t], n_pad * g * 32, hipMemcpyDeviceToHost, st));
    }
    ZG_HIP(hipStreamSynchronize(st));
    sync.dismiss();
    return ZG_OK;
}

// ---- weighted column sums: out[m][c] = sum_r W[m][r] * T[r * cols + c]  (m <= 4 weight vectors share one pass over T).
// The prefix / suffix provers of Stage 3 build their Q tables this way (src/zkvm/spartan/stage3_prover.zig:1066-1112, :2232-2290):
// Q[x_lo] = sum over x_hi of witness(x_lo + x_hi * 2^prefix_vars) * suffix[x_hi]. Consecutive lanes take consecutive columns (every
// row is read coalesced); the rows are cut into gridDim.y slabs whose partial sums a second launch adds up.
static constexpr int COLSUM_MAX_W = 4;
__global__ void __launch_bounds__(256) weighted_colsum_kernel(const uint64_t *tab, size_t rows, size_t cols, const uint64_t *w, int m,
                                                              size_t rows_per_slab, uint64_t *partials /* [slab][m][cols] */) {
    const size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const size_t r0 = (size_t)blockIdx.y * rows_per_slab, r1 = r0 + rows_per_slab < rows ? r0 + rows_per_slab : rows;
    Fr acc[COLSUM_MAX_W];
#pragma unroll
    for (int k = 0; k < COLSUM_MAX_W; k++) acc[k] = Fr::zero();
    for (size_t r = r0; r < r1; r++) {
        Fr t = fe_load<FrParams>(tab + 4 * (r * cols + c));
#pragma unroll
        for (int k = 0; k < COLSUM_MAX_W; k++)
            if (k < m) acc[k] = fe_add(acc[k], fr_mul29v(t, fe_load<FrParams>(w + 4 * ((size_t)k * rows + r))));
    }
#pragma unroll
    for (int k = 0; k < COLSUM_MAX_W; k++)
        if (k < m) fe_store(partials + 4 * (((size_t)blockIdx.y * m + k) * cols + c), acc[k]);
}
// sixteen outputs per workgroup, sixteen lanes per output (each takes every sixteenth slab), a tree over the sixteen in LDS: one thread
// per output walking all <= 256 slabs was a 97 us chain of dependent loads for a 2048-entry result (rocprofv3, Stage 3's Q tables)
constexpr int COLSUM_FIN_OUT = 16;
__global__ void __launch_bounds__(256) colsum_finish_kernel(const uint64_t *partials, size_t slabs, size_t n /* m * cols */, uint64_t *out) {
    __shared__ uint32_t sh[256][8];
    const uint32_t li = threadIdx.x % COLSUM_FIN_OUT, q = threadIdx.x / COLSUM_FIN_OUT;
    const size_t i = (size_t)blockIdx.x * COLSUM_FIN_OUT + li;
    Fr acc = Fr::zero();
    if (i < n)
        for (size_t s = q; s < slabs; s += 256 / COLSUM_FIN_OUT) acc = fe_add(acc, fe_load<FrParams>(partials + 4 * (s * n + i)));
    for (uint32_t half = 256 / COLSUM_FIN_OUT / 2; half >= 1; half >>= 1) {
        if (q >= half && q < 2 * half)
#pragma unroll
            for (int w = 0; w < 8; w++) sh[threadIdx.x][w] = acc.l[w];
        __syncthreads();
        if (q < half) {
            Fr other;
#pragma unroll
            for (int w = 0; w < 8; w++) other.l[w] = sh[threadIdx.x + half * COLSUM_FIN_OUT][w];
            acc = fe_add(acc, other);
        }
        __syncthreads();
    }
    if (q == 0 && i < n) fe_store(out + 4 * i, acc);
}

int zg_fr_weighted_colsum_dev(const uint64_t *d_table, size_t rows, size_t cols, const uint64_t *d_weights, size_t m, uint64_t *d_out, void *stream) {
    ZG_INIT();
    if (m == 0 || m > (size_t)COLSUM_MAX_W || rows == 0 || cols == 0 || !d_table || !d_weights || !d_out) {
        set_error("zg_fr_weighted_colsum_dev: 1..4 weight vectors, a non-empty table");
        return ZG_ERR_INVALID;
    }
    hipStream_t st = pick_stream(stream);
    // enough workgroups to fill the chip: columns / 256 blocks times row slabs
    const size_t col_blocks = div_up(cols, 256);
    size_t slabs = col_blocks >= 1024 ? 1 : 1024 / col_blocks;
    if (slabs > rows) slabs = rows;
    if (slabs > 256) slabs = 256;
    const size_t per = div_up(rows, slabs);
    slabs = div_up(rows, per);
    if (slabs == 1) {
        hipLaunchKernelGGL(weighted_colsum_kernel, dim3((unsigned)col_blocks, 1), dim3(256), 0, st, d_table, rows, cols, d_weights, (int)m, per, d_out);
        ZG_HIP(hipGetLastError());
        return ZG_OK;
    }
    Scratch part(slabs * m * cols * 32);
    if (!part.p) return ZG_ERR_NOMEM;
    hipLaunchKernelGGL(weighted_colsum_kernel, dim3((unsigned)col_blocks, (unsigned)slabs), dim3(256), 0, st, d_table, rows, cols, d_weights, (int)m, per,
                       part.as<uint64_t>());
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)div_up(m * cols, COLSUM_FIN_OUT)), dim3(256), 0, st, part.as<uint64_t>(), slabs, m * cols, d_out);
    ZG_HIP(hipGetLastError());
    ZG_HIP(hipStreamSynchronize(st));  // the scratch partials go back to the cache with this call
    return ZG_OK;
}

int zg_fr_weighted_colsum(const uint64_t *table, size_t rows, size_t cols, const uint64_t *weights, size_t m, uint64_t *out) {
    ZG_INIT();
    if (m == 0 || m > (size_t)COLSUM_MAX_W || rows == 0 || cols == 0 || !table || !weights || !out) {
        set_error("zg_fr_weighted_colsum: 1..4 weight vectors, a non-empty table");
        return ZG_ERR_INVALID;
    }
    hipStream_t st = lib_stream();
    Scratch s_t(rows * cols * 32), s_w(m * rows * 32), s_o(m * cols * 32);
    if (!s_t.p || !s_w.p || !s_o.p) return ZG_ERR_NOMEM;
    SyncGuard sync(st);
    ZG_HIP(hipMemcpyAsync(s_t.p, table, rows * cols * 32, hipMemcpyHostToDevice, st));
    ZG_HIP(hipMemcpyAsync(s_w.p, weights, m * rows * 32, hipMemcpyHostToDevice, st));
    ZG_TRY(zg_fr_weighted_colsum_dev(s_t.as<uint64_t>(), rows, cols, s_w.as<uint64_t>(), m, s_o.as<uint64_t>(), st));
    ZG_HIP(hipMemcpyAsync(out, s_o.p, m * cols * 32, hipMemcpyDeviceToHost, st));
    ZG_HIP(hipStreamSynchronize(st));
    sync.dismiss();
    return ZG_OK;
}

int zg_fr_spartan_combine_dev(const uint64_t *d_eq, const uint64_t *d_az, const uint64_t *d_bz, const uint64_t *d_cz, size_t n,
                              uint64_t *d_out, void *stream) {
    ZG_INIT();
    if (n && (!d_eq || !d_az || !d_bz || !d_cz || !d_out)) {
        set_error("zg_fr_spartan_combine_dev: invalid argument");
        return ZG_ERR_INVALID;
    }
    if (n == 0) return ZG_OK;
    unsigned nb = div_up(n, 256);
    if (nb > 4096) nb = 4096;
    prof_begin(ZG_PROF_COMBINE, pick_stream(stream));
    hipLaunchKernelGGL(spartan_combine_kernel, dim3(nb), dim3(256), 0, pick_stream(stream), d_eq, d_az, d_bz, d_cz, n, d_out);
    prof_end(ZG_PROF_COMBINE, pick_stream(stream));
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

int zg_fr_spartan_combine(const uint64_t *eq, const uint64_t *az, const uint64_t *bz, const uint64_t *cz, size_t n, uint64_t *out) {
    ZG_INIT();
    if (n && (!eq || !az || !bz || !cz || !out)) {
        set_error("zg_fr_spartan_combine: invalid argument");
        return ZG_ERR_INVALID;
    }
    if (n == 0) return ZG_OK;
    hipStream_t st = lib_stream();
    size_t bytes = n * 32;
    Scratch s_d(bytes * 5);
    if (!s_d.p) return ZG_ERR_NOMEM;
    uint64_t *d = s_d.as<uint64_t>();
    const uint64_t *src[4] = {eq, az, bz, cz};
    for (int k = 0; k < 4; k++) ZG_HIP(hipMemcpyAsync(d + 4 * n * k, src[k], bytes, hipMemcpyHostToDevice, st));
    int rc = zg_fr_spartan_combine_dev(d, d + 4 * n, d + 8 * n, d + 12 * n, n, d + 16 * n, st);
    if (rc == ZG_OK) {
        hipError_t e = hipMemcpyAsync(out, d + 16 * n, bytes, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            set_error(hipGetErrorString(e));
            rc = ZG_ERR_HIP;
        }
    }
    return rc;
}

// The fold / quotient / commit loop of HyperKZG.open and batchOpen (src/poly/commitment/mod.zig:283-317, :684-712) on a table
// already resident at d_a (n_evals entries; d_a, d_b: ping-pong buffers of >= n_evals and n_evals/2 entries, d_q: n_evals/2).
// Quotient i is commit(cur[half..] - cur[..half]); levels the fold cannot reach (half == 0, :289 / :686) are reported as
// identity and not counted in *n_quot.
static int hk_open_device(zg_bases_t srs, uint64_t *d_a, uint64_t *d_b, uint64_t *d_q, size_t n_evals, const uint64_t *point,
                          size_t num_vars, hipStream_t st, uint64_t *q_xy, uint8_t *q_inf, uint64_t final_eval[4], size_t *n_quot) {
    size_t srs_len = zg_g1_bases_len(srs);
    std::vector<uint64_t> h_res(9 * num_vars + 4, 0);
    // Levels whose quotient has at most HK_SMALL entries are committed TOGETHER at the end: their quotients are written
    // into rows of one zero-padded matrix (a zero scalar contributes no digit, so commit(q) over bases[0..row length) is
    // unchanged) and handed to zg_msm_g1_batch_dev, which fuses short vectors into one launch set — a lone short MSM is
    // ~0.4 ms of launch latency, and there are log2(HK_SMALL) + 1 of them.
    const size_t HK_SMALL = 16384;  // = the narrow-window side table of a wide-window handle (msm.hip)
    size_t small_rows = 0, small_len = 0, long_rows = 0, long_len = 0, long_len2 = 0;
    {
        size_t len = n_evals;
        for (size_t i = 0; i < num_vars; i++) {
            size_t half = len / 2;
            if (half == 0) break;
            size_t nc = half < srs_len ? half : srs_len;
            if (nc <= HK_SMALL) {
                if (small_rows == 0) small_len = nc;
                small_rows++;
            } else {
                if (long_rows == 0) long_len = nc;
                if (long_rows == 1) long_len2 = nc;
                long_rows++;
            }
            len = half;
        }
        if (small_rows < 2) small_rows = 0;  // nothing to fuse
    }
    // The long levels go out the same way: rows of one zero-padded matrix of row length long_len (the first level's), committed
    // by ONE fused launch set on a helper stream (msm_batch_dev_wide) — one sort, one accumulation over all levels' digits (as many
    // as a single MSM of twice the first level's size has) and one reduction, instead of five or six launch sets that overlap
    // badly. ZG_HK_FUSE_LONG=0 keeps one launch set per long level.
    // Mode 1 (default): all long levels in one matrix. Mode 2: the first long level — half of all live scalars — keeps a launch set of
    // its own on another helper stream, and the matrix holds the remaining long levels at the SECOND level's row length (the padding
    // that the digit and sort kernels walk shrinks from 2.6 M to 1.5 M scalars at 2^20 evaluations). Mode 2 was the default until a
    // kernel trace of round 4 showed what "side by side" means on the device: the second set's sort kernels (whole-CU workgroups) crawl
    // while the first set's accumulation owns the register files (colscan 12 -> 360 us, fine_place 77 -> 290 us), so its accumulation
    // starts when the first one ends and two reduction chains are exposed instead of one — 2^20 evaluations from a host table:
    // mode 2 3.30-3.41 ms, mode 1 3.08-3.13 ms, mode 0 3.72 ms (tools/ab_open.py, profiles/r4h_open_modes.txt).
    const unsigned fuse_long_env = env_uint("ZG_HK_FUSE_LONG", 1, 0, 2);  // read per call: tests switch it
    // The long levels' commits are independent of the folds that follow them: each gets its own quotient buffer and its MSM is
    // issued on one of three helper streams in turn (forked / joined by events), never on the caller's stream, so the
    // latency-bound tail of one commit runs under the accumulation of the next.
    constexpr int NAUX = 3;
    // the helper streams belong to THIS call: a GROUP taken from the runtime's per-device free list (creating a stream costs ~3 ms) and
    // handed back on return, so concurrent opens from different caller threads do not serialise through a shared set (round-2 review).
    // A group's streams were created back to back and therefore sit on different hardware queues (HIP hands its four queues out in
    // creation order): three streams taken one by one from the common free list put the first long level and the fused long levels
    // on ONE queue in a trace of round 4 — the two launch sets that are meant to overlap ran one after the other (2.69 ms per open).
    struct AuxStreams {
        hipStream_t s[NAUX] = {nullptr, nullptr, nullptr};
        int dev;
        AuxStreams() : dev(current_device()) { (void)stream_group_acquire(s); }
        ~AuxStreams() {
            if (s[0]) stream_group_release(s, dev);
        }
    } aux_streams;
    hipStream_t *aux = aux_streams.s;
    const bool fork = aux[0] && aux[1] && aux[2];
    const bool fuse_long = fork && fuse_long_env && long_rows >= 2;
    const bool split_first = fuse_long && fuse_long_env == 2 && long_rows >= 3;
    const size_t fl_len = split_first ? long_len2 : long_len, fl_rows = split_first ? long_rows - 1 : long_rows,
                 fl_off = split_first ? long_len : 0;  // matrix geometry inside d_qall (the lone first level's quotient sits before it)
    std::vector<hipEvent_t> events;
    struct EventGuard {
        std::vector<hipEvent_t> &ev;
        ~EventGuard() { for (hipEvent_t e : ev) (void)hipEventDestroy(e); }
    } guard{events};
    Scratch s_res((9 * num_vars + 4) * 8), s_misc(SC_MISC_BYTES), s_small((small_rows * small_len + 1) * 32),
        s_qall(((fuse_long ? fl_off + fl_rows * fl_len : 0) + n_evals + 1) * 32);  // matrix, then one buffer per unfused level
    if (!s_res.p || !s_misc.p || !s_small.p || !s_qall.p) return ZG_ERR_NOMEM;
    uint64_t *d_qall = s_qall.as<uint64_t>();
    size_t q_used = 0, long_used = 0, first_long = num_vars;
    bool first_split_done = false;
    std::vector<size_t> long_row_len;  // live entries of the matrix rows: the fused commit's sort skips the padding behind them
    struct PendingCommit { size_t level, nc; const uint64_t *q; };
    std::vector<PendingCommit> pending;
    bool aux_used[NAUX] = {false, false, false};
    uint64_t *d_res = s_res.as<uint64_t>(), *d_misc = s_misc.as<uint64_t>(), *d_small = s_small.as<uint64_t>();
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = hipMemsetAsync(d_res, 0, (9 * num_vars + 4) * 8, st);
    if (e == hipSuccess) e = hipMemsetAsync(d_misc + 8 * (size_t)SC_MAX_BLOCKS, 0, SC_COUNTER_BYTES, st);
    if (e == hipSuccess && small_rows) e = hipMemsetAsync(d_small, 0, small_rows * small_len * 32, st);
    if (e == hipSuccess && fuse_long) e = hipMemsetAsync(d_qall + 4 * fl_off, 0, fl_rows * fl_len * 32, st);
    int rc = ZG_OK;
    size_t len = n_evals, computed = 0, first_small = num_vars, row = 0;
    uint64_t *cur = d_a, *nxt = d_b;
    for (size_t i = 0; i < num_vars && e == hipSuccess && rc == ZG_OK; i++) {
        size_t half = len / 2;
        if (half == 0) {  // the reference stops folding; remaining quotients stay unset -> identity here
            for (size_t r = i; r < num_vars; r++) h_res[9 * r + 8] = 0x100;  // marker: identity
            break;
        }
        unsigned nb = div_up(half, 256);
        if (nb > 4096) nb = 4096;
        size_t nc = half < srs_len ? half : srs_len;  // commit(): n = min(evals.len, srs.len), :246
        FrArg ra;
        for (int k = 0; k < 4; k++) {
            ra.l[2 * k] = (uint32_t)point[4 * i + k];
            ra.l[2 * k + 1] = (uint32_t)(point[4 * i + k] >> 32);
        }
        if (nb > 1024) nb = 1024;
        if (small_rows && nc <= HK_SMALL) {
            if (first_small == num_vars) first_small = i;
            // only the first nc entries are committed; the row keeps zeros beyond them
            hipLaunchKernelGGL(hk_quot_fold_kernel, dim3(nb), dim3(256), 0, st, cur, half, ra, d_small + 4 * small_len * row, nc, nxt);
            row++;
        } else {
            if (fuse_long && nc > HK_SMALL) {  // row long_used of the padded matrix; only the nc committed entries are kept
                if (split_first && !first_split_done) {
                    first_split_done = true;
                    hipLaunchKernelGGL(hk_quot_fold_kernel, dim3(nb), dim3(256), 0, st, cur, half, ra, d_qall, nc, nxt);
                    pending.push_back(PendingCommit{i, nc, d_qall});
                } else {
                    if (first_long == num_vars) first_long = i;
                    hipLaunchKernelGGL(hk_quot_fold_kernel, dim3(nb), dim3(256), 0, st, cur, half, ra, d_qall + 4 * (fl_off + fl_len * long_used), nc,
                                       nxt);
                    long_row_len.push_back(nc);
                    long_used++;
                }
                computed++;
                uint64_t *t2 = cur; cur = nxt; nxt = t2;
                len = half;
                continue;
            }
            uint64_t *qi = fork ? d_qall + 4 * ((fuse_long ? fl_off + fl_rows * fl_len : 0) + q_used) : d_q;
            q_used += half;
            hipLaunchKernelGGL(hk_quot_fold_kernel, dim3(nb), dim3(256), 0, st, cur, half, ra, qi, half, nxt);
            if (fork) {
                pending.push_back(PendingCommit{i, nc, qi});  // issued after the chain, see below
            } else {
                rc = zg_msm_g1_dev_async(srs, 0, nc, qi, st, d_res + 9 * i, reinterpret_cast<uint8_t *>(d_res + 9 * i + 8));
                if (rc != ZG_OK) break;
            }
        }
        computed++;
        uint64_t *t = cur; cur = nxt; nxt = t;
        len = half;
    }
    // The short levels' fused commit depends on the chain only: it runs beside the long commits, on the group's third stream (the
    // caller's stream may share a hardware queue with one of the helpers; it only carries the chain, the joins and the result copy),
    // and it is enqueued FIRST: its short kernels start right behind the chain and are done before the long levels' accumulation needs
    // the chip (enqueued last, they ran under that accumulation's sort and stretched it: 616 -> 531 us in the trace of 2^20 evaluations).
    if (e == hipSuccess && rc == ZG_OK && row) {  // rows are consecutive levels first_small, first_small + 1, ...
        hipStream_t ss = st;
        if (fork) {
            hipEvent_t ev = nullptr;
            e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e == hipSuccess) {
                events.push_back(ev);
                e = hipEventRecord(ev, st);
            }
            if (e == hipSuccess) e = hipStreamWaitEvent(aux[NAUX - 1], ev, 0);
            if (e == hipSuccess) {
                aux_used[NAUX - 1] = true;
                ss = aux[NAUX - 1];
            }
        }
        if (e == hipSuccess) rc = zg_msm_g1_batch_dev(srs, small_len, d_small, row, ss, d_res + 9 * first_small);
    }
    // The whole chain is enqueued (and, being a few short kernels, finished) before the first commit starts: kernels of different
    // streams share the dispatch pipes, and a commit's sort kernels (whole-CU workgroups that launch as accumulate workgroups
    // retire) held the chain's next link back by 0.5-1 ms per level when both were in flight. Commits then go out on the three
    // helper streams in turn, largest first.
    if (fork && e == hipSuccess && rc == ZG_OK && !pending.empty()) {
        hipEvent_t ev = nullptr;
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e == hipSuccess) {
            events.push_back(ev);
            e = hipEventRecord(ev, st);
        }
        for (size_t k = 0; k < pending.size() && e == hipSuccess && rc == ZG_OK; k++) {
            hipStream_t si = aux[k % NAUX];
            if (!aux_used[k % NAUX]) {
                e = hipStreamWaitEvent(si, ev, 0);
                if (e != hipSuccess) break;
                aux_used[k % NAUX] = true;
            }
            const PendingCommit &pc = pending[k];
            rc = zg_msm_g1_dev_async(srs, 0, pc.nc, pc.q, si, d_res + 9 * pc.level, reinterpret_cast<uint8_t *>(d_res + 9 * pc.level + 8));
        }
    }
    if (fuse_long && e == hipSuccess && rc == ZG_OK && long_used) {
        hipEvent_t ev = nullptr;
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e == hipSuccess) {
            events.push_back(ev);
            e = hipEventRecord(ev, st);
        }
        const int fa = split_first ? 1 : 0;  // the lone first level took helper 0
        if (e == hipSuccess && !aux_used[fa]) e = hipStreamWaitEvent(aux[fa], ev, 0);
        if (e == hipSuccess) {
            aux_used[fa] = true;
            rc = msm_batch_dev_wide(srs, fl_len, d_qall + 4 * fl_off, long_used, aux[fa], d_res + 9 * first_long, long_row_len.data());
        }
    }
    for (int a = 0; a < NAUX; a++)  // join the helper streams (also after an error, so that they never run ahead of later work)
        if (aux_used[a]) {
            hipEvent_t ev = nullptr;
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) {
                events.push_back(ev);
                if (hipEventRecord(ev, aux[a]) == hipSuccess) (void)hipStreamWaitEvent(st, ev, 0);
            } else {
                (void)hipStreamSynchronize(aux[a]);
            }
        }
    if (e == hipSuccess && rc == ZG_OK && len > 0)
        e = hipMemcpyAsync(d_res + 9 * num_vars, cur, 32, hipMemcpyDeviceToDevice, st);  // final = current[0], :317
    std::vector<uint64_t> dev_res(9 * num_vars + 4);
    if (e == hipSuccess && rc == ZG_OK) e = hipMemcpyAsync(dev_res.data(), d_res, (9 * num_vars + 4) * 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    else (void)hipStreamSynchronize(st);
    if (e != hipSuccess) {
        set_error(std::string("hyperkzg open: ") + hipGetErrorString(e));
        return ZG_ERR_HIP;
    }
    if (rc != ZG_OK) return rc;
    for (size_t i = 0; i < num_vars; i++) {
        bool skipped = h_res[9 * i + 8] == 0x100;
        for (int j = 0; j < 8; j++) q_xy[8 * i + j] = skipped ? 0 : dev_res[9 * i + j];
        if (q_inf) q_inf[i] = skipped ? 1 : (uint8_t)(dev_res[9 * i + 8] & 0xff);
    }
    for (int j = 0; j < 4; j++) final_eval[j] = len > 0 ? dev_res[9 * num_vars + j] : 0;
    if (n_quot) *n_quot = computed;
    return ZG_OK;
}

int zg_hyperkzg_open(zg_bases_t srs, const uint64_t *evals, size_t n_evals, const uint64_t *point, size_t num_vars,
                     const uint64_t value[4], uint64_t *q_xy, uint8_t *q_inf, uint64_t final_eval[4]) {
    ZG_INIT();
    if (!final_eval || (num_vars && (!point || !q_xy)) || (n_evals && !evals) || (num_vars == 0 && !value)) {
        set_error("zg_hyperkzg_open: invalid argument");
        return ZG_ERR_INVALID;
    }
    if (num_vars == 0) {  // :270-276
        for (int i = 0; i < 4; i++) final_eval[i] = value[i];
        return ZG_OK;
    }
    if (!srs) {
        set_error("zg_hyperkzg_open: invalid argument");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(bases_device(srs));
    hipStream_t st = lib_stream();
    size_t cap = n_evals ? n_evals : 1;
    Scratch s_a(cap * 32), s_b((cap / 2 + 1) * 32), s_q((cap / 2 + 1) * 32);
    if (!s_a.p || !s_b.p || !s_q.p) return ZG_ERR_NOMEM;
    SyncGuard sync(st);  // whatever happens below, the scratch tables are idle when they return to the cache
    if (n_evals) ZG_HIP(hipMemcpyAsync(s_a.p, evals, n_evals * 32, hipMemcpyHostToDevice, st));
    return hk_open_device(srs, s_a.as<uint64_t>(), s_b.as<uint64_t>(), s_q.as<uint64_t>(), n_evals, point, num_vars, st, q_xy, q_inf,
                          final_eval, nullptr);
}

// ---- HyperKZG.batchOpen (src/poly/commitment/mod.zig:607-732)
// gpow[i] = gamma^i for i < k, gpow[k] = gamma;  gamma = fromU64(0x9a8b7c6d) * prod_j (point[j] + fromU64(11)), 0 -> 1 (:633-640)
__global__ void hk_gamma_kernel(const uint64_t *point, uint32_t v, uint32_t k, uint64_t *gpow) {
    Fr g = Fr::zero(), e11 = Fr::zero();
    g.l[0] = 0x9a8b7c6du;
    e11.l[0] = 11u;
    g = fe_to_mont(g);
    e11 = fe_to_mont(e11);
    for (uint32_t j = 0; j < v; j++) g = fe_mul(g, fe_add(fe_load<FrParams>(point + 4 * j), e11));
    if (fr_eq(g, Fr::zero())) g = Fr::one();
    Fr pw = Fr::one();
    for (uint32_t i = 0; i < k; i++) {
        fe_store(gpow + 4 * (size_t)i, pw);
        pw = fe_mul(pw, g);
    }
    fe_store(gpow + 4 * (size_t)k, g);
}

// out[j] (+)= g * p[j] for j < n_p; entries of out beyond n_p keep their value (zero on the first pass)  (:646-654)
__global__ void __launch_bounds__(256) hk_axpy_kernel(uint64_t *out, size_t n_out, const uint64_t *p, size_t n_p, const uint64_t *g, int first) {
    F29 gp = fr29_prescale(fe_load<FrParams>(g));
    size_t stride = (size_t)gridDim.x * 256;
    for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < n_out; j += stride) {
        Fr acc = first ? Fr::zero() : fe_load<FrParams>(out + 4 * j);
        if (j < n_p) acc = fe_add(acc, fr_mul29(fe_load<FrParams>(p + 4 * j), gp));
        fe_store(out + 4 * j, acc);
    }
}

// evaluateMultilinear's monomial sum (:796-813): sum_idx a[idx] * eq[idx & mask] (eq indexed LSB-first via the reversed point)
__global__ void __launch_bounds__(256) hk_dot_mask_kernel(const uint64_t *a, size_t n, const uint64_t *eq, size_t mask, uint64_t *partials) {
    __shared__ uint4 sh[256 * 4];
    Fr g0 = Fr::zero(), g1 = Fr::zero();
    size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        g0 = fe_add(g0, fr_mul29v(fe_load<FrParams>(a + 4 * i), fe_load<FrParams>(eq + 4 * (i & mask))));
    block_sum_pair(g0, g1, sh);
    if (threadIdx.x == 0) {
        fe_store(partials + 8 * (size_t)blockIdx.x, g0);
        fe_store(partials + 8 * (size_t)blockIdx.x + 4, g1);
    }
}

// combined_eval = sum_i gamma^i * evaluations[i] (:657-662)
__global__ void hk_combined_eval_kernel(const uint64_t *evals, const uint64_t *gpow, uint32_t k, uint64_t *out) {
    Fr acc = Fr::zero();
    for (uint32_t i = 0; i < k; i++) acc = fe_add(acc, fe_mul(fe_load<FrParams>(gpow + 4 * (size_t)i), fe_load<FrParams>(evals + 4 * (size_t)i)));
    fe_store(out, acc);
}

int zg_hyperkzg_open_dev(zg_bases_t srs, const uint64_t *d_evals, size_t n_evals, const uint64_t *point, size_t num_vars,
                         const uint64_t value[4], void *stream, uint64_t *q_xy, uint8_t *q_inf, uint64_t final_eval[4]) {
    ZG_INIT();
    if (!srs || !final_eval || (num_vars && (!point || !q_xy)) || (n_evals && !d_evals) || (num_vars == 0 && !value)) {
        set_error("zg_hyperkzg_open_dev: invalid argument");
        return ZG_ERR_INVALID;
    }
    if (num_vars == 0) {  // :270-276
        for (int i = 0; i < 4; i++) final_eval[i] = value[i];
        return ZG_OK;
    }
    DeviceGuard dg(bases_device(srs));
    hipStream_t st = pick_stream(stream);
    size_t cap = n_evals ? n_evals : 1;
    Scratch s_a(cap * 32), s_b((cap / 2 + 1) * 32), s_q((cap / 2 + 1) * 32);
    if (!s_a.p || !s_b.p || !s_q.p) return ZG_ERR_NOMEM;
    SyncGuard sync(st);
    // the loop folds its table in place: work on a copy, the caller's polynomial stays intact (open() takes evals by const slice)
    if (n_evals) ZG_HIP(hipMemcpyAsync(s_a.p, d_evals, n_evals * 32, hipMemcpyDeviceToDevice, st));
    return hk_open_device(srs, s_a.as<uint64_t>(), s_b.as<uint64_t>(), s_q.as<uint64_t>(), n_evals, point, num_vars, st, q_xy, q_inf,
                          final_eval, nullptr);
}

int zg_fr_scale(const uint64_t *a, size_t n, const uint64_t sc[4], uint64_t *out) {
    ZG_INIT();
    if (n && (!a || !sc || !out)) {
        set_error("zg_fr_scale: invalid argument");
        return ZG_ERR_INVALID;
    }
    if (n == 0) return ZG_OK;
    hipStream_t st = lib_stream();
    Scratch s_a(n * 32), s_o(n * 32), s_s(32);
    if (!s_a.p || !s_o.p || !s_s.p) return ZG_ERR_NOMEM;
    SyncGuard sync(st);
    ZG_HIP(hipMemcpyAsync(s_a.p, a, n * 32, hipMemcpyHostToDevice, st));
    ZG_HIP(hipMemcpyAsync(s_s.p, sc, 32, hipMemcpyHostToDevice, st));
    unsigned nb = div_up(n, 256);
    if (nb > 4096) nb = 4096;
    // out = 0 + a * s: the axpy kernel of HyperKZG.batchOpen's random linear combination with an empty accumulator
    hipLaunchKernelGGL(hk_axpy_kernel, dim3(nb), dim3(256), 0, st, s_o.as<uint64_t>(), n, s_a.as<uint64_t>(), n, s_s.as<uint64_t>(), 1);
    ZG_HIP(hipGetLastError());
    ZG_HIP(hipMemcpyAsync(out, s_o.p, n * 32, hipMemcpyDeviceToHost, st));
    ZG_HIP(hipStreamSynchronize(st));
    sync.dismiss();
    return ZG_OK;
}

int zg_hyperkzg_batch_open(zg_bases_t srs, const uint64_t *const *polys, const size_t *lens, size_t k, const uint64_t *point,
                           size_t num_vars, uint64_t *q_xy, uint8_t *q_inf, size_t *n_quot, uint64_t *evaluations,
                           uint64_t final_eval[4], uint64_t gamma[4]) {
    ZG_INIT();
    if (!final_eval || !gamma || !n_quot || (k && (!polys || !lens || !evaluations)) || (num_vars && (!point || !q_xy)) || num_vars > 34) {
        set_error("zg_hyperkzg_batch_open: invalid argument");
        return ZG_ERR_INVALID;
    }
    *n_quot = 0;
    if (k == 0) {  // :613-621
        for (int i = 0; i < 4; i++) {
            final_eval[i] = 0;
            gamma[i] = (uint64_t)FrParams::ONE[2 * i] | ((uint64_t)FrParams::ONE[2 * i + 1] << 32);
        }
        return ZG_OK;
    }
    for (size_t i = 0; i < k; i++)
        if (lens[i] && !polys[i]) {
            set_error("zg_hyperkzg_batch_open: null polynomial");
            return ZG_ERR_INVALID;
        }
    if (!srs) {
        set_error("zg_hyperkzg_batch_open: invalid argument");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(bases_device(srs));
    hipStream_t st = lib_stream();
    size_t poly_size = lens[0], max_len = 1;
    for (size_t i = 0; i < k; i++) max_len = lens[i] > max_len ? lens[i] : max_len;
    size_t cap = poly_size ? poly_size : 1;
    Scratch s_pt((num_vars + 1) * 32), s_g((k + 1) * 32), s_p(max_len * 32), s_a(cap * 32), s_b((cap / 2 + 1) * 32), s_q((cap / 2 + 1) * 32),
        s_ev((k + 1) * 32), s_misc(SC_MISC_BYTES), s_eq(((size_t)1 << (num_vars <= 10 ? num_vars : 0)) * 32);
    if (!s_pt.p || !s_g.p || !s_p.p || !s_a.p || !s_b.p || !s_q.p || !s_ev.p || !s_misc.p || !s_eq.p) return ZG_ERR_NOMEM;
    SyncGuard sync(st);  // an early error return waits for the enqueued work before the scratch buffers are recycled
    uint64_t *d_pt = s_pt.as<uint64_t>(), *d_g = s_g.as<uint64_t>(), *d_p = s_p.as<uint64_t>(), *d_a = s_a.as<uint64_t>(),
             *d_ev = s_ev.as<uint64_t>(), *d_misc = s_misc.as<uint64_t>(), *d_eq = s_eq.as<uint64_t>();
    if (num_vars) ZG_HIP(hipMemcpyAsync(d_pt, point, num_vars * 32, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(hk_gamma_kernel, dim3(1), dim3(1), 0, st, d_pt, (uint32_t)num_vars, (uint32_t)k, d_g);
    // evaluateMultilinear (:788-817): the direct sum only for point.len <= 10 and len <= 1024, else evals[0]; empty -> 0
    bool any_small = false;
    for (size_t i = 0; i < k; i++) any_small = any_small || (lens[i] && num_vars && num_vars <= 10 && lens[i] <= 1024);
    if (any_small) {  // eq(point, .) with index bit j <-> point[j]: the big-endian table of the reversed point
        std::vector<uint64_t> rev(4 * num_vars);
        for (size_t j = 0; j < num_vars; j++)
            for (int l = 0; l < 4; l++) rev[4 * j + l] = point[4 * (num_vars - 1 - j) + l];
        ZG_TRY(eq_table_enqueue(rev.data(), num_vars, nullptr, d_eq, st));
    }
    std::vector<uint64_t> h_ev(4 * k, 0);
    std::vector<char> on_dev(k, 0);
    for (size_t i = 0; i < k; i++) {
        if (lens[i]) ZG_HIP(hipMemcpyAsync(d_p, polys[i], lens[i] * 32, hipMemcpyHostToDevice, st));
        unsigned nb = div_up(cap, 256);
        if (nb > 4096) nb = 4096;
        hipLaunchKernelGGL(hk_axpy_kernel, dim3(nb), dim3(256), 0, st, d_a, poly_size, d_p, lens[i] < poly_size ? lens[i] : poly_size,
                           d_g + 4 * i, i == 0 ? 1 : 0);
        if (lens[i] && num_vars && num_vars <= 10 && lens[i] <= 1024) {
            unsigned nd = sc_blocks(lens[i]);
            hipLaunchKernelGGL(hk_dot_mask_kernel, dim3(nd), dim3(256), 0, st, d_p, lens[i], d_eq, ((size_t)1 << num_vars) - 1, d_misc);
            hipLaunchKernelGGL(sc_finish_kernel, dim3(1), dim3(256), 0, st, d_misc, nd, d_misc + SC_SUMS_OFF, (uint64_t *)nullptr, (uint64_t)0);
            ZG_HIP(hipMemcpyAsync(d_ev + 4 * i, d_misc + SC_SUMS_OFF, 32, hipMemcpyDeviceToDevice, st));
            on_dev[i] = 1;
        } else if (lens[i]) {
            for (int l = 0; l < 4; l++) h_ev[4 * i + l] = polys[i][l];  // point.len == 0 or the large-polynomial fallback: evals[0]
        }
        ZG_HIP(hipGetLastError());  // d_p is reused by the next polynomial: its upload is ordered behind these kernels on the same stream
    }
    std::vector<uint64_t> d2h(4 * (k + 1));
    ZG_HIP(hipMemcpyAsync(d2h.data(), d_ev, 4 * 8 * k, hipMemcpyDeviceToHost, st));
    ZG_HIP(hipMemcpyAsync(gamma, d_g + 4 * k, 32, hipMemcpyDeviceToHost, st));
    ZG_HIP(hipStreamSynchronize(st));
    for (size_t i = 0; i < k; i++)
        for (int l = 0; l < 4; l++) evaluations[4 * i + l] = on_dev[i] ? d2h[4 * i + l] : h_ev[4 * i + l];
    if (num_vars == 0) {  // :665-673: no quotients, final_eval = combined_eval
        ZG_HIP(hipMemcpyAsync(d_ev, evaluations, 4 * 8 * k, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(hk_combined_eval_kernel, dim3(1), dim3(1), 0, st, d_ev, d_g, (uint32_t)k, d_ev + 4 * k);
        ZG_HIP(hipMemcpyAsync(final_eval, d_ev + 4 * k, 32, hipMemcpyDeviceToHost, st));
        ZG_HIP(hipStreamSynchronize(st));
        return ZG_OK;
    }
    return hk_open_device(srs, d_a, s_b.as<uint64_t>(), s_q.as<uint64_t>(), poly_size, point, num_vars, st, q_xy, q_inf, final_eval, n_quot);
}

// ---------------------------------------------------------------- runSumcheck, device-resident

static uint32_t ilog2_sz(size_t x) {
    uint32_t r = 0;
    while (((size_t)1 << (r + 1)) <= x) r++;
    return r;
}

// pinned result blocks of zg_run_sumcheck (one per call in flight; hipHostMalloc costs more than the protocol, so they are pooled)
constexpr size_t RUN_PIN_WORDS = 1024;  // result words (17 + 12 v <= 425 for v <= 34), the flag in the last word
static std::mutex g_runpin_mu;
static std::vector<uint64_t *> g_runpin_free;
static uint64_t *runpin_get() {
    {
        std::lock_guard<std::mutex> lk(g_runpin_mu);
        if (!g_runpin_free.empty()) {
            uint64_t *p = g_runpin_free.back();
            g_runpin_free.pop_back();
            return p;
        }
    }
    uint64_t *p = nullptr;
    if (hipHostMalloc((void **)&p, RUN_PIN_WORDS * 8, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return nullptr;
    return p;
}
static void runpin_put(uint64_t *p) {
    std::lock_guard<std::mutex> lk(g_runpin_mu);
    g_runpin_free.push_back(p);
}
struct RunPin {
    uint64_t *p;
    RunPin() : p(runpin_get()) {}
    ~RunPin() { if (p) runpin_put(p); }
};
// partials + arrival counters of zg_run_sumcheck, pooled per device: zeroed when created, and every kernel leaves its counters at zero
// (sc_arrive), so a call needs no memset launch in front of its first kernel (2.4 us + a 4 us gap of a 0.19 ms protocol). A block that
// may hold a half-finished arrival (an error return) is zeroed again before it goes back.
static std::mutex g_runmisc_mu;
static std::vector<uint64_t *> g_runmisc_free[ZG_MAX_DEVICES];
struct RunMisc {
    uint64_t *p = nullptr;
    int dev;
    bool clean = false;
    RunMisc() : dev(current_device()) {
        if (dev < 0 || dev >= ZG_MAX_DEVICES) return;
        {
            std::lock_guard<std::mutex> lk(g_runmisc_mu);
            if (!g_runmisc_free[dev].empty()) {
                p = g_runmisc_free[dev].back();
                g_runmisc_free[dev].pop_back();
                return;
            }
        }
        if (hipMalloc((void **)&p, SC_MISC_BYTES) != hipSuccess || hipMemset(p, 0, SC_MISC_BYTES) != hipSuccess) {
            if (p) (void)hipFree(p);
            p = nullptr;
        }
    }
    ~RunMisc() {
        if (!p) return;
        if (!clean && hipMemset(p, 0, SC_MISC_BYTES) != hipSuccess) {  // (synchronous: the stream has been drained by the SyncGuard)
            (void)hipFree(p);
            return;
        }
        std::lock_guard<std::mutex> lk(g_runmisc_mu);
        g_runmisc_free[dev].push_back(p);
    }
};

static int run_sumcheck_enqueue(const uint64_t *d_evals, size_t len, hipStream_t st, uint64_t claim[4], uint64_t *rounds,
                                uint64_t *challenges, uint64_t final_eval[4], uint8_t *result) {
    if (!d_evals || len == 0 || (len & (len - 1)) || !claim || !final_eval || !result) {
        set_error("zg_run_sumcheck: len must be a power of two >= 1 and outputs non-null");
        return ZG_ERR_INVALID;
    }
    uint32_t v = ilog2_sz(len);
    if (v && (!rounds || !challenges)) {
        set_error("zg_run_sumcheck: rounds / challenges buffers missing");
        return ZG_ERR_INVALID;
    }
    size_t res_words = 17 + 12 * (size_t)v;
    if (v == 0) {  // no rounds: claim = final_eval = the single evaluation
        uint64_t h[4];
        ZG_HIP(hipMemcpyAsync(h, d_evals, 32, hipMemcpyDeviceToHost, st));
        ZG_HIP(hipStreamSynchronize(st));
        for (int i = 0; i < 4; i++) claim[i] = final_eval[i] = h[i];
        *result = 1;
        return ZG_OK;
    }
    Scratch s_a(len / 2 * 32), s_b((len / 4 ? len / 4 : 1) * 32), s_res(res_words * 8);
    RunMisc s_misc;  // (declared before the SyncGuard: destroyed after it has drained the stream)
    RunPin pin;
    if (!s_a.p || !s_b.p || !s_res.p || !s_misc.p) return ZG_ERR_NOMEM;
    if (!pin.p || res_words >= RUN_PIN_WORDS) {
        set_error("zg_run_sumcheck: no pinned result block");
        return ZG_ERR_NOMEM;
    }
    SyncGuard sync(st);  // the scratch buffers and the pinned block go back to their pools only after the work on st has drained
    uint64_t *d_res = s_res.as<uint64_t>(), *d_misc = s_misc.p;
    uint64_t *buf[2] = {s_a.as<uint64_t>(), s_b.as<uint64_t>()};
    uint64_t *h = pin.p, *hflag = pin.p + RUN_PIN_WORDS - 1;
    __atomic_store_n(hflag, 0ull, __ATOMIC_RELEASE);
    // round 0's sums: also fixes the claim; every later round's sums come out of the fold that precedes it
    // The launch that produces round k's sums only leaves block pairs; the NEXT launch opens with round k's verifier step (ScRunArg):
    // launch 0 = the sums of round 0, then fold k (verifier step k, fold by its challenge, sums of round k + 1) for k = 0, 1, ...
    ZG_TRY(launch_sums(ZG_SC_HIGH_HALF, d_evals, len, d_misc, d_misc + SC_SUMS_OFF, st, nullptr, 0, ScRunArg{d_res, v, 0, 1, 0}));
    uint32_t nb_prev = sums_grid(len / 2, true);
    const uint64_t *cur = d_evals;
    size_t cl = len;
    const uint32_t tail_max = (uint32_t)env_uint("ZG_SC_TAIL_MAX", SC_TAIL_MAX, 1, SC_TAIL_MAX);  // 1: every round as its own launch
    uint32_t k = 0;
    for (; k < v && cl > tail_max; k++) {
        uint64_t *nxt = buf[k & 1];
        ZG_TRY(launch_fold(ZG_SC_HIGH_HALF, cur, cl, nullptr, nxt, d_misc, d_misc + SC_SUMS_OFF, st, nullptr, 0,
                           ScRunArg{d_res, v, k, k == 0 ? 1u : 0u, nb_prev}));
        nb_prev = fold_grid(cl / 2, true);
        cur = nxt;
        cl /= 2;
    }
    if (cl > 1) {  // the remaining rounds in one launch, table in LDS; it also hands the result block to the host
        static PerDeviceOnce once;
        ZG_HIP(once.run([] {
            return hipFuncSetAttribute(reinterpret_cast<const void *>(sc_tail_run_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(SC_TAIL_MAX * 32 + SC_TAIL_LDS_EXTRA));
        }));
        hipLaunchKernelGGL(sc_tail_run_kernel, dim3(1), dim3(SC_TAIL_THREADS), cl * 32 + SC_TAIL_LDS_EXTRA, st, cur, (uint32_t)cl,
                           ScRunArg{d_res, v, k, k == 0 ? 1u : 0u, nb_prev}, d_misc, (uint32_t)res_words, h, hflag);
    } else {
        hipLaunchKernelGGL(sc_run_publish_kernel, dim3(1), dim3(256), 0, st, d_res, (uint32_t)res_words, h, hflag);
    }
    ZG_HIP(hipGetLastError());
    {  // spin on the flag; a stream synchronisation if it does not arrive promptly (or an error stopped the stream)
        bool got = false;
        for (uint64_t spin = 0; spin < (1ull << 22); spin++) {
            if (__atomic_load_n(hflag, __ATOMIC_ACQUIRE) == 1) { got = true; break; }
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
        if (!got) {
            ZG_HIP(hipStreamSynchronize(st));
            if (__atomic_load_n(hflag, __ATOMIC_ACQUIRE) != 1) {
                set_error("zg_run_sumcheck: the result block was not published");
                return ZG_ERR_HIP;
            }
        }
        sync.dismiss();  // the flag is the last thing the last kernel writes: nothing enqueued here touches the scratch buffers any more
        s_misc.clean = true;  // every launch ran to its end: all counters are back at zero
    }
    for (int i = 0; i < 4; i++) {
        claim[i] = h[i];
        final_eval[i] = h[4 + 12 * (size_t)v + i];
    }
    for (size_t i = 0; i < 8 * (size_t)v; i++) rounds[i] = h[4 + i];
    for (size_t i = 0; i < 4 * (size_t)v; i++) challenges[i] = h[4 + 8 * (size_t)v + i];
    uint64_t status = h[8 + 12 * (size_t)v];
    *result = (uint8_t)(status & 1);
    if (status >> 8) {
        set_error("zg_run_sumcheck: SumcheckVerificationFailed in round " + std::to_string((status >> 8) - 1));
        return ZG_ERR_VERIFY;
    }
    return ZG_OK;
}

int zg_selftest_handoff(unsigned blocks, unsigned threads, unsigned iters, int busy, uint64_t *mismatches, uint64_t *completed) {
    ZG_INIT();
    if (!mismatches || !completed || blocks < 2 || blocks > SC_MAX_BLOCKS || threads < 64 || threads > 1024 || (threads & 63u)) {
        set_error("zg_selftest_handoff: blocks in [2, 2048], threads a multiple of 64 in [64, 1024]");
        return ZG_ERR_INVALID;
    }
    hipStream_t st = lib_stream();
    Scratch s_misc(SC_MISC_BYTES), s_res(64), s_busy(busy ? (size_t)256 << 20 : 64);
    if (!s_misc.p || !s_res.p || !s_busy.p) return ZG_ERR_NOMEM;
    SyncGuard sync(st);
    uint64_t *d_misc = s_misc.as<uint64_t>(), *d_res = s_res.as<uint64_t>();
    ZG_HIP(hipMemsetAsync(d_misc, 0, SC_MISC_BYTES, st));
    ZG_HIP(hipMemsetAsync(d_res, 0, 64, st));
    hipStream_t st2 = nullptr;
    if (busy) ZG_HIP(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
    uint32_t *counter = reinterpret_cast<uint32_t *>(d_misc + 8 * (size_t)SC_MAX_BLOCKS);
    for (unsigned it = 0; it < iters; it++) {
        if (busy && (it % 4) == 0) hipLaunchKernelGGL(handoff_busy_kernel, dim3(1024), dim3(256), 0, st2, s_busy.as<uint4>(), ((size_t)256 << 20) / 16);
        hipLaunchKernelGGL(handoff_stress_kernel, dim3(blocks), dim3(threads), 0, st, d_misc, counter, it + 1, d_res);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (st2) {
        (void)hipStreamSynchronize(st2);
        (void)hipStreamDestroy(st2);
    }
    ZG_HIP(e);
    uint64_t h[2];
    ZG_HIP(hipMemcpy(h, d_res, 16, hipMemcpyDeviceToHost));
    *mismatches = h[0];
    *completed = h[1];
    return ZG_OK;
}

int zg_run_sumcheck_dev(const uint64_t *d_evals, size_t len, void *stream, uint64_t claim[4], uint64_t *rounds, uint64_t *challenges,
                        uint64_t final_eval[4], uint8_t *result) {
    ZG_INIT();
    return run_sumcheck_enqueue(d_evals, len, pick_stream(stream), claim, rounds, challenges, final_eval, result);
}

int zg_run_sumcheck(const uint64_t *evals, size_t len, uint64_t claim[4], uint64_t *rounds, uint64_t *challenges, uint64_t final_eval[4],
                    uint8_t *result) {
    ZG_INIT();
    if (!evals || len == 0 || (len & (len - 1))) {
        set_error("zg_run_sumcheck: len must be a power of two >= 1");
        return ZG_ERR_INVALID;
    }
    hipStream_t st = lib_stream();
    Scratch s_t(len * 32);
    if (!s_t.p) return ZG_ERR_NOMEM;
    ZG_HIP(hipMemcpyAsync(s_t.p, evals, len * 32, hipMemcpyHostToDevice, st));
    return run_sumcheck_enqueue(s_t.as<uint64_t>(), len, st, claim, rounds, challenges, final_eval, result);
}

// ---------------------------------------------------------------- sumcheck session
int zg_sumcheck_open(const uint64_t *evals, size_t len, int layout, zg_sc_t *out) {
    ZG_INIT();
    if (!evals || !out) {
        set_error("zg_sumcheck_open: invalid argument");
        return ZG_ERR_INVALID;
    }
    zg_sc_s *s = nullptr;
    ZG_TRY(sc_create(len, layout, nullptr, &s));  // a stream of its own
    hipError_t e = hipMemcpyAsync(s->buf[0], evals, len * 32, hipMemcpyHostToDevice, s->st);
    if (e == hipSuccess) e = hipStreamSynchronize(s->st);
    if (e != hipSuccess) {
        set_error(hipGetErrorString(e));
        sc_free(s);
        return ZG_ERR_HIP;
    }
    *out = s;
    return ZG_OK;
}

int zg_sumcheck_open_dev(const uint64_t *d_evals, size_t len, int layout, void *stream, zg_sc_t *out) {
    ZG_INIT();
    if (!d_evals || !out) {
        set_error("zg_sumcheck_open_dev: invalid argument");
        return ZG_ERR_INVALID;
    }
    zg_sc_s *s = nullptr;
    ZG_TRY(sc_create(len, layout, pick_stream(stream), &s));
    hipError_t e = hipMemcpyAsync(s->buf[0], d_evals, len * 32, hipMemcpyDeviceToDevice, s->st);
    if (e != hipSuccess) {
        set_error(hipGetErrorString(e));
        sc_free(s);
        return ZG_ERR_HIP;
    }
    *out = s;
    return ZG_OK;
}

int zg_sumcheck_open_dev_borrowed(const uint64_t *d_evals, size_t len, int layout, void *stream, zg_sc_t *out) {
    ZG_INIT();
    if (!d_evals || !out) {
        set_error("zg_sumcheck_open_dev_borrowed: invalid argument");
        return ZG_ERR_INVALID;
    }
    zg_sc_s *s = nullptr;
    ZG_TRY(sc_create(len, layout, pick_stream(stream), &s, true));
    s->buf[0] = const_cast<uint64_t *>(d_evals);  // read only: every kernel that takes buf[cur] while `borrowing` has a const table argument
    s->borrowing = true;
    *out = s;
    return ZG_OK;
}

int zg_sumcheck_open_column(const zg_col_t *col, size_t n_rows, size_t len, int layout, zg_sc_t *out) {
    ZG_INIT();
    if (!col || !out || n_rows > len) {
        set_error("zg_sumcheck_open_column: invalid argument (n_rows <= len)");
        return ZG_ERR_INVALID;
    }
    zg_sc_s *s = nullptr;
    ZG_TRY(sc_create(len, layout, nullptr, &s));  // a stream of its own
    int rc = ZG_OK;
    if (len > n_rows) {
        hipError_t e = hipMemsetAsync(s->buf[0] + 4 * n_rows, 0, (len - n_rows) * 32, s->st);
        if (e != hipSuccess) {
            set_error(hipGetErrorString(e));
            rc = ZG_ERR_HIP;
        }
    }
    if (rc == ZG_OK) rc = rows_from_host_columns(col, 1, n_rows, s->buf[0], s->st);  // synchronises the session's stream
    if (rc == ZG_OK && n_rows == 0 && hipStreamSynchronize(s->st) != hipSuccess) rc = ZG_ERR_HIP;
    if (rc != ZG_OK) {
        std::string keep = zg_last_error();
        sc_free(s);
        set_error(keep);
        return rc;
    }
    *out = s;
    return ZG_OK;
}

int zg_sumcheck_open_spartan_dev(const uint64_t *r, size_t v, const uint64_t *scale, const uint64_t *d_az, const uint64_t *d_bz,
                                 const uint64_t *d_cz, int layout, void *stream, zg_sc_t *out) {
    ZG_INIT();
    if (!out || (v && !r) || !d_az || !d_bz || v > 30) {  // d_cz may be NULL: Cz = 0
        set_error("zg_sumcheck_open_spartan_dev: invalid argument");
        return ZG_ERR_INVALID;
    }
    zg_sc_s *s = nullptr;
    ZG_TRY(sc_create((size_t)1 << v, layout, pick_stream(stream), &s));
    s->seq = 1;  // the fused kernel publishes round 0's pair
    int rc = eq_spartan_enqueue(r, v, scale, d_az, d_bz, d_cz, layout, s->buf[0], s->d_partials, s->h_pin, s->h_pin + 12, s->seq, s->st);
    if (rc != ZG_OK) {
        sc_free(s);
        return rc;
    }
    s->sums_valid = s->len >= 2;
    *out = s;
    return ZG_OK;
}

int zg_sumcheck_round_sums(zg_sc_t s, uint64_t g0[4], uint64_t g1[4]) {
    ZG_INIT();
    if (!s || !g0 || !g1 || s->len < 2) {
        set_error("zg_sumcheck_round_sums: invalid session or protocol already complete");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    if (!s->sums_valid) {
        s->seq++;
        ZG_TRY(launch_sums(s->layout, s->buf[s->cur], s->len, s->d_partials, s->h_pin, s->st, s->h_pin + 12, s->seq));
        s->sums_valid = true;
    }
    uint64_t h[8];
    ZG_TRY(sc_wait_mailbox(s, h));  // the only host<->device rendezvous of a round
    for (int i = 0; i < 4; i++) {
        g0[i] = h[i];
        g1[i] = h[4 + i];
    }
    return ZG_OK;
}

}  // extern "C"

namespace zg {
// sharded.hip: start the pass that produces the session's round sums (if the last fold did not leave them behind) without
// waiting for the mailbox; the following zg_sumcheck_round_sums only collects them
int sc_round_sums_start(zg_sc_t s) {
    if (!s || s->len < 2) {
        set_error("zg_sumcheck_round_sums: invalid session or protocol already complete");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    if (!s->sums_valid) {
        s->seq++;
        ZG_TRY(launch_sums(s->layout, s->buf[s->cur], s->len, s->d_partials, s->h_pin, s->st, s->h_pin + 12, s->seq));
        s->sums_valid = true;
    }
    return ZG_OK;
}
}  // namespace zg

extern "C" {

int zg_sumcheck_bind(zg_sc_t s, const uint64_t r[4]) {
    ZG_INIT();
    if (!s || !r || s->len < 2) {
        set_error("zg_sumcheck_bind: invalid session or protocol already complete");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    // buf[1] holds len/2 elements at most; after the first fold both buffers are large enough
    int nxt = s->cur ^ 1;
    s->seq++;
    ZG_TRY(launch_fold(s->layout, s->buf[s->cur], s->len, r, s->buf[nxt], s->d_partials, s->h_pin, s->st, s->h_pin + 12,
                       s->seq));  // asynchronous
    s->cur = nxt;
    s->len /= 2;
    s->sums_valid = s->len >= 2;
    s->bit_valid = s->pad_valid = false;
    if (s->borrowing) {  // the fold just enqueued was the last reader of the caller's table (nxt == 1: it wrote own[1])
        s->buf[0] = s->own[0];
        s->borrowing = false;
    }
    return ZG_OK;
}

size_t zg_sumcheck_len(zg_sc_t s) { return s ? s->len : 0; }

int zg_sumcheck_final(zg_sc_t s, uint64_t out[4]) {
    ZG_INIT();
    if (!s || !out || s->len != 1) {
        set_error("zg_sumcheck_final: protocol not complete");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    ZG_HIP(hipMemcpyAsync(s->h_pin + 8, s->buf[s->cur], 32, hipMemcpyDeviceToHost, s->st));
    ZG_HIP(hipStreamSynchronize(s->st));
    for (int i = 0; i < 4; i++) out[i] = s->h_pin[8 + i];
    return ZG_OK;
}

int zg_sumcheck_gather(zg_sc_t s, const uint64_t *idx, size_t n, uint64_t *out) {
    ZG_INIT();
    if (!s || (n && (!idx || !out))) {
        set_error("zg_sumcheck_gather: invalid argument");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    return gather_to_host(s->buf[s->cur], s->len, idx, n, out, s->st);
}

int zg_sumcheck_read(zg_sc_t s, uint64_t *out_table) {
    ZG_INIT();
    if (!s || !out_table) {
        set_error("zg_sumcheck_read: invalid argument");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    ZG_HIP(hipMemcpyAsync(out_table, s->buf[s->cur], s->len * 32, hipMemcpyDeviceToHost, s->st));
    ZG_HIP(hipStreamSynchronize(s->st));
    return ZG_OK;
}

// Stream-ordered variants for a table sharded over several GPUs: the local pair of round sums /
// the current table land in DEVICE memory on the session's stream, ready for an RCCL all-gather enqueued on
// the same stream; nothing is synchronised here.
int zg_sumcheck_round_sums_dev(zg_sc_t s, uint64_t *d_out) {
    ZG_INIT();
    if (!s || !d_out || s->len < 2) {
        set_error("zg_sumcheck_round_sums_dev: invalid session or protocol already complete");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    if (!s->sums_valid) {
        s->seq++;
        ZG_TRY(launch_sums(s->layout, s->buf[s->cur], s->len, s->d_partials, s->h_pin, s->st, s->h_pin + 12, s->seq));
        s->sums_valid = true;
    }
    ZG_HIP(hipMemcpyAsync(d_out, s->h_pin, 64, hipMemcpyHostToDevice, s->st));  // ordered before the next fold's write
    return ZG_OK;
}

int zg_sumcheck_read_dev(zg_sc_t s, uint64_t *d_out_table) {
    ZG_INIT();
    if (!s || !d_out_table) {
        set_error("zg_sumcheck_read_dev: invalid argument");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    ZG_HIP(hipMemcpyAsync(d_out_table, s->buf[s->cur], s->len * 32, hipMemcpyDeviceToDevice, s->st));
    return ZG_OK;
}

int zg_sumcheck_raf_round(zg_sc_t s, const uint64_t base[4], uint64_t current_power, uint64_t s0[4], uint64_t s2[4]) {
    ZG_INIT();
    if (!s || !base || !s0 || !s2 || s->layout != ZG_SC_LOW_PAIR || s->len < 2) {
        set_error("zg_sumcheck_raf_round: needs a LOW_PAIR session with at least two entries");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    const size_t half = s->len / 2;
    // rem(i) = 2 * current_power * i must fit 64 bits for every i < half (the reference's u64 `remaining_power *= 2` would
    // overflow otherwise, :377-386)
    unsigned __int128 top = (unsigned __int128)current_power * 2 * (half ? half : 1);
    if (current_power == 0 || (top >> 64) != 0) {
        set_error("zg_sumcheck_raf_round: current_power * table length overflows 64 bits");
        return ZG_ERR_INVALID;
    }
    const uint64_t step = current_power * 2;
    FrArg ba;
    for (int i = 0; i < 4; i++) {
        ba.l[2 * i] = (uint32_t)base[i];
        ba.l[2 * i + 1] = (uint32_t)(base[i] >> 32);
    }
    unsigned nb = sc_blocks(half);
    uint32_t *counter = reinterpret_cast<uint32_t *>(s->d_partials + 8 * (size_t)SC_MAX_BLOCKS);
    s->sums_valid = false;  // the mailbox now carries s(0), s(2)
    s->bit_valid = false;
    s->seq++;
    hipLaunchKernelGGL(raf_round_kernel, dim3(nb), dim3(256), 0, s->st, s->buf[s->cur], half, ba, step, s->d_partials, s->h_pin, counter,
                       s->h_pin + 12, s->seq);
    uint64_t h[8];
    ZG_TRY(sc_wait_mailbox(s, h));
    for (int i = 0; i < 4; i++) {
        s0[i] = h[i];
        s2[i] = h[4 + i];
    }
    return ZG_OK;
}

int zg_sumcheck_raf_claim(zg_sc_t s, uint64_t base, uint64_t step, uint64_t claim[4]) {
    ZG_INIT();
    if (!s || !claim) {
        set_error("zg_sumcheck_raf_claim: invalid argument");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    // base + step * k is a u64 in the reference (UnmapPolynomial.evaluateAtIndex, :207-209): it must not wrap for any k < len
    const unsigned __int128 top = (unsigned __int128)base + (unsigned __int128)step * (s->len ? s->len - 1 : 0);
    if ((top >> 64) != 0) {
        set_error("zg_sumcheck_raf_claim: base + step * (len - 1) overflows 64 bits");
        return ZG_ERR_INVALID;
    }
    unsigned nb = sc_blocks(s->len);
    uint32_t *counter = reinterpret_cast<uint32_t *>(s->d_partials + 8 * (size_t)SC_MAX_BLOCKS);
    s->sums_valid = false;  // the mailbox now carries (claim, 0)
    s->bit_valid = false;
    s->seq++;
    hipLaunchKernelGGL(raf_claim_kernel, dim3(nb), dim3(256), 0, s->st, s->buf[s->cur], s->len, base, step, s->d_partials, s->h_pin, counter, s->h_pin + 12,
                       s->seq);
    uint64_t h[8];
    ZG_TRY(sc_wait_mailbox(s, h));
    for (int i = 0; i < 4; i++) claim[i] = h[i];
    return ZG_OK;
}

int zg_sumcheck_bit_round(zg_sc_t s, const uint64_t *d_idx128, size_t n_idx, unsigned bit, uint64_t sum0[4], uint64_t sum1[4]) {
    ZG_INIT();
    if (!s || !sum0 || !sum1 || bit > 127 || n_idx > s->len || (n_idx && !d_idx128)) {
        set_error("zg_sumcheck_bit_round: invalid argument");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    if (!(s->bit_valid && s->bit_cached == bit && s->bit_n == n_idx && s->bit_idx == d_idx128)) {
        unsigned nb = sc_blocks(n_idx ? n_idx : 1);
        uint32_t *counter = reinterpret_cast<uint32_t *>(s->d_partials + 8 * (size_t)SC_MAX_BLOCKS);
        s->sums_valid = false;  // the mailbox now holds this pair
        s->seq++;
        hipLaunchKernelGGL(bit_split_sums_kernel, dim3(nb), dim3(256), 0, s->st, s->buf[s->cur], d_idx128, n_idx, (uint32_t)bit, s->d_partials,
                           s->h_pin, counter, s->h_pin + 12, s->seq);
        ZG_TRY(sc_wait_mailbox(s, s->bit_sums));
        s->bit_valid = true;
        s->bit_cached = bit;
        s->bit_n = n_idx;
        s->bit_idx = d_idx128;
    }
    for (int i = 0; i < 4; i++) {
        sum0[i] = s->bit_sums[i];
        sum1[i] = s->bit_sums[4 + i];
    }
    return ZG_OK;
}

int zg_sumcheck_bit_bind(zg_sc_t s, const uint64_t *d_idx128, size_t n_idx, unsigned bit, const uint64_t r[4], uint64_t claim[4]) {
    ZG_INIT();
    if (!s || !r || !claim || bit > 127 || n_idx > s->len || (n_idx && !d_idx128)) {
        set_error("zg_sumcheck_bit_bind: invalid argument");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    uint32_t *counter = reinterpret_cast<uint32_t *>(s->d_partials + 8 * (size_t)SC_MAX_BLOCKS);
    s->sums_valid = false;  // the table changes and the mailbox is reused: the HIGH_HALF / LOW_PAIR sums are stale
    if (!(s->pad_valid && s->pad_from == n_idx)) {  // once per session: the entries past the lookups stay as they are
        for (int i = 0; i < 4; i++) s->pad[i] = 0;
        if (n_idx < s->len) {
            unsigned nb = sc_blocks(s->len - n_idx);
            uint64_t h[8];
            s->seq++;
            hipLaunchKernelGGL(range_sum_kernel, dim3(nb), dim3(256), 0, s->st, s->buf[s->cur], n_idx, s->len, s->d_partials, s->h_pin, counter,
                               s->h_pin + 12, s->seq);
            ZG_TRY(sc_wait_mailbox(s, h));
            for (int i = 0; i < 4; i++) s->pad[i] = h[i];
        }
        s->pad_valid = true;
        s->pad_from = n_idx;
    }
    FrArg ra;
    for (int i = 0; i < 4; i++) {
        ra.l[2 * i] = (uint32_t)r[i];
        ra.l[2 * i + 1] = (uint32_t)(r[i] >> 32);
    }
    const unsigned next_bit = bit < 127 ? bit + 1 : bit;
    unsigned nb = sc_blocks(n_idx ? n_idx : 1);
    if (s->borrowing) {
        set_error("zg_sumcheck_bit_bind: binds in place — not on a session that still reads a borrowed table (zg_sumcheck_open_dev_borrowed)");
        return ZG_ERR_INVALID;
    }
    s->bit_valid = false;
    s->seq++;
    hipLaunchKernelGGL(bit_bind_kernel, dim3(nb), dim3(256), 0, s->st, s->buf[s->cur], d_idx128, n_idx, (uint32_t)bit, (uint32_t)next_bit, ra,
                       s->d_partials, s->h_pin, counter, s->h_pin + 12, s->seq);
    ZG_TRY(sc_wait_mailbox(s, s->bit_sums));
    if (next_bit != bit) {
        s->bit_valid = true;
        s->bit_cached = next_bit;
        s->bit_n = n_idx;
        s->bit_idx = d_idx128;
    }
    uint64_t t[4];
    fr_add_host(t, s->bit_sums, s->bit_sums + 4);
    fr_add_host(claim, t, s->pad);
    return ZG_OK;
}

int zg_fr_bit_split_sums_dev(const uint64_t *d_vals, const uint64_t *d_idx128, size_t n, unsigned bit, void *stream, uint64_t sum0[4],
                             uint64_t sum1[4]) {
    ZG_INIT();
    if (!sum0 || !sum1 || bit > 127 || (n && (!d_vals || !d_idx128))) {
        set_error("zg_fr_bit_split_sums_dev: invalid argument");
        return ZG_ERR_INVALID;
    }
    hipStream_t st = pick_stream(stream);
    Scratch s_misc(SC_MISC_BYTES);
    if (!s_misc.p) return ZG_ERR_NOMEM;
    SyncGuard sync(st);
    uint64_t *d_misc = s_misc.as<uint64_t>();
    unsigned nb = sc_blocks(n ? n : 1);
    hipLaunchKernelGGL(bit_split_sums_kernel, dim3(nb), dim3(256), 0, st, d_vals, d_idx128, n, (uint32_t)bit, d_misc, (uint64_t *)nullptr,
                       (uint32_t *)nullptr, (uint64_t *)nullptr, (uint64_t)0);
    hipLaunchKernelGGL(sc_finish_kernel, dim3(1), dim3(256), 0, st, d_misc, nb, d_misc + SC_SUMS_OFF, (uint64_t *)nullptr, (uint64_t)0);
    ZG_HIP(hipGetLastError());
    uint64_t h[8];
    ZG_HIP(hipMemcpyAsync(h, d_misc + SC_SUMS_OFF, 64, hipMemcpyDeviceToHost, st));
    ZG_HIP(hipStreamSynchronize(st));
    sync.dismiss();
    for (int i = 0; i < 4; i++) {
        sum0[i] = h[i];
        sum1[i] = h[4 + i];
    }
    return ZG_OK;
}

int zg_fr_bit_split_sums(const uint64_t *vals, const uint64_t *idx128, size_t n, unsigned bit, uint64_t sum0[4], uint64_t sum1[4]) {
    ZG_INIT();
    if (!sum0 || !sum1 || bit > 127 || (n && (!vals || !idx128))) {
        set_error("zg_fr_bit_split_sums: invalid argument");
        return ZG_ERR_INVALID;
    }
    hipStream_t st = lib_stream();
    Scratch s_v((n ? n : 1) * 32), s_i((n ? n : 1) * 16);
    if (!s_v.p || !s_i.p) return ZG_ERR_NOMEM;
    SyncGuard sync(st);
    if (n) {
        ZG_HIP(hipMemcpyAsync(s_v.p, vals, n * 32, hipMemcpyHostToDevice, st));
        ZG_HIP(hipMemcpyAsync(s_i.p, idx128, n * 16, hipMemcpyHostToDevice, st));
    }
    return zg_fr_bit_split_sums_dev(s_v.as<uint64_t>(), s_i.as<uint64_t>(), n, bit, st, sum0, sum1);
}

int zg_sumcheck_close(zg_sc_t s) {
    if (!s) return ZG_OK;
    ZG_INIT();
    DeviceGuard dg(s->device);
    (void)hipStreamSynchronize(s->st);
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        // (twelve: a proof opens sessions of very different lengths — three of 2^25 entries in Stage 1, then 2^16 ... 2^20 — and a pooled
        // session only serves lengths within a factor of four of its own; with four slots the large ones held the pool and every other
        // session of a proof was created and destroyed again, a pinned mailbox allocation each time)
        if (g_pool.size() < 12) {
            g_pool.push_back(s);
            return ZG_OK;
        }
    }
    sc_free(s);
    return ZG_OK;
}

}  // extern "C"

namespace zg {
void sc_shutdown() {  // zg_shutdown: drop the pooled sessions (buffers, pinned mailbox, stream)
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (zg_sc_s *s : g_pool) {
        DeviceGuard dg(s->device);
        sc_free(s);
    }
    g_pool.clear();
    std::lock_guard<std::mutex> lk2(g_runpin_mu);
    for (uint64_t *p : g_runpin_free) (void)hipHostFree(p);
    g_runpin_free.clear();
    std::lock_guard<std::mutex> lk3(g_runmisc_mu);
    for (int d = 0; d < ZG_MAX_DEVICES; d++) {
        if (g_runmisc_free[d].empty()) continue;
        DeviceGuard dg(d);
        for (uint64_t *p : g_runmisc_free[d]) (void)hipFree(p);
        g_runmisc_free[d].clear();
    }
}
}  // namespace zg
