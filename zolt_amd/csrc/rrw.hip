// rrw.hip — RegistersReadWriteChecking (Stage 4) on gfx950: the dense K x T tables of Stage4GruenProver
// (/root/reference/src/zkvm/spartan/stage4_gruen_prover.zig:65-1240) resident in HBM for the whole sumcheck.
//
// The reference keeps five tables indexed [k * T + j] (k < K = 128 registers, j < T cycles) — val, rd_wa, ra = gamma rs1_ra +
// gamma^2 rs2_ra, rs1_ra, rs2_ra — plus inc[T] and an eq structure over the cycles, and runs LOG_K + log T rounds in three phases:
//   phase 1  cycle variables, Gruen form:  (q0, qX2) = sum_i E_out[x_out(i)] E_in[x_in(i)] sum_k C(k, i)        (:561-741)
//   phase 2  register variables:           (e0, e2)  = sum_j eq[j] sum_i C(2i, 2i+1; j) at t = 0, 2            (:764-852)
//   phase 3  remaining cycle variables:    (e0, e2, e3) with the dense merged eq table                          (:854-953)
// with C = ra val + wa (val + inc), and folds every table by the challenge after each round (:1047-1163). At T = 2^20 that is
// 4 GiB per table and 128 M products per round for a CPU loop; here a round is one pass over ra / wa (and val where they are not
// zero) + a fold pass, all five tables moved between two device buffers (compact rows of the live length).
//
// Sparsity. ra and wa start one-hot per cycle (at most two reads, one write among 128 registers) and a fold at most doubles the
// share of non-zero entries, so in the large early rounds almost every (k, i) pair contributes nothing: the round kernels test
// ra / wa first and skip val and the products where all four are zero, the fold kernels skip the product where hi == lo. The sums
// and the folded tables are the reference's values exactly (zero terms, zero differences).
#include <cstdlib>
#include <mutex>
#include <vector>

#include "common.hip.h"
#include "field.hip.h"
#include "fp29.hip.h"
#include "sc_common.hip.h"

namespace zg {

static constexpr int RRW_K = 128, RRW_TABLES = 5;  // val, wa, ra, rs1_ra, rs2_ra
// Register indices are 5-bit instruction fields and the register file has 32 entries (:186-192, 199-246), so rows 32..127 of all five
// tables are zero — and stay zero under every fold of the cycle variables. Only the ACTIVE rows are stored and visited (32 until the
// register rounds, halved — rounded up — by each of them); a row at or past the active count reads as zero. A quarter of the memory
// and of the traffic of the full 128 x T tables, the same values.
static constexpr int RRW_ACTIVE = 32;
enum { RT_VAL = 0, RT_WA = 1, RT_RA = 2, RT_RS1 = 3, RT_RS2 = 4 };
static constexpr unsigned RRW_MAX_BLOCKS = 65536;

struct RrwTabs {
    const uint64_t *t[RRW_TABLES];
};
struct RrwTabsOut {
    uint64_t *t[RRW_TABLES];
};

ZG_DEV Fr fr_from_arg(const FrArg &a) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = a.l[i];
    return r;
}

// tables from the per-cycle trace columns: val[k][j] = F.fromU64(register k before cycle j) (k < 32, else 0), wa / rs1_ra / rs2_ra one-hot
// in the register index, ra = gamma rs1_ra + gamma^2 rs2_ra (:183-246)
__global__ void __launch_bounds__(256) rrw_build_kernel(const uint8_t *rs1, const uint8_t *rs2, const uint8_t *rd, const uint64_t *reg_vals,
                                                        size_t T, FrArg gamma, RrwTabsOut out) {
    size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)RRW_ACTIVE * T) return;
    size_t k = idx / T, j = idx - k * T;
    F29 r2p;
#pragma unroll
    for (int i = 0; i < 9; i++) r2p.l[i] = Fr29::R2PRE[i];
    Fr v = Fr::zero();
    if (k < 32) {
        uint64_t u = reg_vals[k * T + j];
        if (u) {
            v.l[0] = (uint32_t)u;
            v.l[1] = (uint32_t)(u >> 32);
            v = fr_mul29(v, r2p);  // F.fromU64
        }
    }
    const Fr one = Fr::one(), zero = Fr::zero();
    const bool a1 = rs1[j] == k, a2 = rs2[j] == k, w = rd[j] == k;
    Fr ra = zero;
    if (a1) ra = fr_from_arg(gamma);
    if (a2) {
        Fr g = fr_from_arg(gamma);
        ra = fe_add(ra, fr_mul29v(g, g));
    }
    fe_store(out.t[RT_VAL] + 4 * idx, v);
    fe_store(out.t[RT_WA] + 4 * idx, w ? one : zero);
    fe_store(out.t[RT_RA] + 4 * idx, ra);
    fe_store(out.t[RT_RS1] + 4 * idx, a1 ? one : zero);
    fe_store(out.t[RT_RS2] + 4 * idx, a2 ? one : zero);
}

// The register file before every cycle, rebuilt on the device from the write column alone (zg_rrw_open_trace): the value of register k
// before cycle j is rd_value of the last cycle < j that wrote k (:186-192, 249-258 walk the trace with a running register file). Three
// launches: the last write per register inside every 64-cycle chunk (one wave = one chunk: a ballot per register), an exclusive running
// maximum over the chunks per register, and the per-cycle lookup (ballot again, the lanes below mine that wrote k) which also forms inc.
__global__ void __launch_bounds__(256) rrw_last_write_kernel(const uint8_t *rd, size_t T, int32_t *last, size_t nch) {
    size_t j = (size_t)blockIdx.x * 256 + threadIdx.x, chunk = j >> 6;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t r = j < T ? rd[j] : 0xFFu;
    int32_t mine = -1;
    for (uint32_t k = 0; k < 32; k++) {
        uint64_t m = __ballot(r == k);
        if (lane == k && m) mine = (int32_t)((chunk << 6) + 63 - __clzll((long long)m));
    }
    if (lane < 32 && chunk < nch) last[lane * nch + chunk] = mine;
}
// one workgroup per register: last[k][c] := the last write to k in the chunks before c (-1: none yet)
__global__ void __launch_bounds__(256) rrw_carry_scan_kernel(int32_t *last, size_t nch) {
    __shared__ int32_t top[256];
    int32_t *row = last + (size_t)blockIdx.x * nch;
    const size_t per = (nch + 255) / 256, c0 = threadIdx.x * per, c1 = c0 + per < nch ? c0 + per : nch;
    int32_t m = -1;
    for (size_t c = c0; c < c1; c++) m = row[c] > m ? row[c] : m;
    top[threadIdx.x] = m;
    __syncthreads();
    int32_t run = -1;
    for (uint32_t t = 0; t < threadIdx.x; t++) run = top[t] > run ? top[t] : run;
    for (size_t c = c0; c < c1; c++) {
        int32_t t = row[c];
        row[c] = run;
        run = t > run ? t : run;
    }
}
__global__ void __launch_bounds__(256) rrw_regfile_kernel(const uint8_t *rd, const uint64_t *rd_value, size_t T, const int32_t *carry, size_t nch,
                                                          uint64_t *reg_vals, uint64_t *inc) {
    size_t j = (size_t)blockIdx.x * 256 + threadIdx.x, chunk = j >> 6;
    const uint32_t lane = threadIdx.x & 63;
    const bool live = j < T;
    const uint32_t r = live ? rd[j] : 0xFFu;
    const int32_t c_in = lane < 32 && chunk < nch ? carry[lane * nch + chunk] : -1;
    uint64_t old = 0;
    for (uint32_t k = 0; k < 32; k++) {
        const uint64_t below = __ballot(r == k) & (((uint64_t)1 << lane) - 1);
        const int32_t carried = __shfl(c_in, (int)k);  // by every lane: a shuffle under the branch below would read idle lanes
        int32_t idx = below ? (int32_t)((chunk << 6) + 63 - __clzll((long long)below)) : carried;
        uint64_t v = idx >= 0 ? rd_value[idx] : 0;
        if (live) reg_vals[k * T + j] = v;
        if (r == k) old = v;
    }
    if (!live) return;
    Fr d = Fr::zero();
    if (r < 32) {  // inc = F.fromU64(post) - F.fromU64(pre) (:240-243)
        F29 r2p;
#pragma unroll
        for (int i = 0; i < 9; i++) r2p.l[i] = Fr29::R2PRE[i];
        const uint64_t post = rd_value[j];
        Fr a = Fr::zero(), b = Fr::zero();
        a.l[0] = (uint32_t)post;
        a.l[1] = (uint32_t)(post >> 32);
        b.l[0] = (uint32_t)old;
        b.l[1] = (uint32_t)(old >> 32);
        d = fe_sub(fr_mul29(a, r2p), fr_mul29(b, r2p));
    }
    fe_store(inc + 4 * j, d);
}

// Which rows of a cycle can be non-zero in ra / rd_wa / rs1_ra / rs2_ra: bit k of mask[j]. The four tables start one-hot (at most three
// registers per cycle), and a cycle fold can only produce a non-zero where one of its two inputs had one: mask'[i] = mask[2i] | mask[2i+1].
// While the cycle variables are being folded, the round kernels visit and the fold kernel WRITES only the masked entries of the four
// tables — at round 0 three rows of 32, doubling per round at worst — instead of streaming a table of zeros; entries outside the mask
// are not written at all, rrw_materialize_kernel zeroes them once before the first reader that ignores the mask (the register rounds,
// when the tables are 2^-phase1 of their size). val is dense (a register holds its value whether or not a cycle touches it) and is
// folded densely.
__global__ void __launch_bounds__(256) rrw_mask_build_kernel(const uint8_t *rs1, const uint8_t *rs2, const uint8_t *rd, size_t T, uint32_t *mask) {
    size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= T) return;
    uint32_t m = 0;
    if (rs1[j] < RRW_ACTIVE) m |= 1u << rs1[j];
    if (rs2[j] < RRW_ACTIVE) m |= 1u << rs2[j];
    if (rd[j] < RRW_ACTIVE) m |= 1u << rd[j];
    mask[j] = m;
}
__global__ void __launch_bounds__(256) rrw_materialize_kernel(RrwTabsOut tb, size_t stride, const uint32_t *mask, size_t cur_T, uint32_t act_K) {
    size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= cur_T * act_K) return;
    size_t k = t / cur_T, j = t - k * cur_T;
    if ((mask[j] >> k) & 1u) return;
    const Fr z = Fr::zero();
    fe_store(tb.t[RT_WA] + 4 * (k * stride + j), z);
    fe_store(tb.t[RT_RA] + 4 * (k * stride + j), z);
    fe_store(tb.t[RT_RS1] + 4 * (k * stride + j), z);
    fe_store(tb.t[RT_RS2] + 4 * (k * stride + j), z);
}

// block partial sums of NV accumulators -> partials[block][NV] (finished by rrw_finish_kernel)
template <int NV>
__device__ __forceinline__ void rrw_block_out(Fr (&acc)[NV], uint4 *sh, uint64_t *partials) {
    Fr z = Fr::zero();
    block_sum_pair(acc[0], acc[1], sh);
    if constexpr (NV == 3) {
        __syncthreads();
        block_sum_pair(acc[2], z, sh);
    }
    if constexpr (NV == 4) {
        __syncthreads();
        block_sum_pair(acc[2], acc[3], sh);
    }
    if (threadIdx.x == 0)
#pragma unroll
        for (int a = 0; a < NV; a++) fe_store(partials + 4 * ((size_t)blockIdx.x * NV + a), acc[a]);
}

__global__ void __launch_bounds__(256) rrw_finish_kernel(const uint64_t *partials, uint32_t nblocks, int nv, uint64_t *out) {
    __shared__ uint4 sh[256 * 4];
    for (int a = 0; a < nv; a++) {
        Fr g0 = Fr::zero(), g1 = Fr::zero();
        for (uint32_t b = threadIdx.x; b < nblocks; b += 256) g0 = fe_add(g0, fe_load<FrParams>(partials + 4 * ((size_t)b * nv + a)));
        block_sum_pair(g0, g1, sh);
        if (threadIdx.x == 0) fe_store(out + 4 * a, g0);
        __syncthreads();
    }
}

ZG_DEV bool fr_is_zero(const Fr &a) { return a.is_zero(); }

// phase 1 (:561-741): thread t -> cycle pair i = t % half_T, register chunk t / half_T; (q0, qX2) += E(i) * sum_k C_0 / C_X2
__global__ void __launch_bounds__(256) rrw_cycle_gruen_kernel(RrwTabs tb, size_t stride, const uint64_t *inc, const uint64_t *e_out, uint32_t n_out,
                                                              const uint64_t *e_in, uint32_t n_in, uint32_t in_bits, size_t half_T, uint32_t cur_K,
                                                              uint32_t kc_n, const uint32_t *mask, uint64_t *partials) {
    __shared__ uint4 sh[256 * 4];
    Fr acc[2] = {Fr::zero(), Fr::zero()};
    size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t < half_T * kc_n) {
        size_t i = t % half_T;
        uint32_t kc = (uint32_t)(t / half_T);
        Fr inc0 = fe_load<FrParams>(inc + 8 * i), incs = fe_sub(fe_load<FrParams>(inc + 8 * i + 4), inc0);
        Fr c0 = Fr::zero(), cx = Fr::zero();
        bool any = false;
        // rows of this thread: k = kc (mod kc_n), k < cur_K; with masks each LANE walks its own set bits (a wave-uniform row loop would
        // run all 32 rows for every wave: some lane always has the bit), without them every row
        uint32_t rows = 0;
        for (uint32_t k = kc; k < cur_K; k += kc_n) rows |= 1u << k;
        const uint32_t me = mask ? mask[2 * i] : ~0u, mo = mask ? mask[2 * i + 1] : ~0u;
        uint32_t m = (me | mo) & rows;
        while (m) {
            const uint32_t k = (uint32_t)__builtin_ctz(m);
            m &= m - 1;
            const bool he = (me >> k) & 1u, ho = (mo >> k) & 1u;
            size_t o = 4 * ((size_t)k * stride + 2 * i);
            const Fr z = Fr::zero();
            Fr rae = he ? fe_load<FrParams>(tb.t[RT_RA] + o) : z, rao = ho ? fe_load<FrParams>(tb.t[RT_RA] + o + 4) : z;
            Fr wae = he ? fe_load<FrParams>(tb.t[RT_WA] + o) : z, wao = ho ? fe_load<FrParams>(tb.t[RT_WA] + o + 4) : z;
            if (fr_is_zero(rae) && fr_is_zero(rao) && fr_is_zero(wae) && fr_is_zero(wao)) continue;  // C_0 = C_X2 = 0
            any = true;
            Fr vae = fe_load<FrParams>(tb.t[RT_VAL] + o), vao = fe_load<FrParams>(tb.t[RT_VAL] + o + 4);
            Fr ras = fe_sub(rao, rae), was = fe_sub(wao, wae), vas = fe_sub(vao, vae);
            c0 = fe_add(c0, fe_add(fr_mul29v(rae, vae), fr_mul29v(wae, fe_add(vae, inc0))));
            cx = fe_add(cx, fe_add(fr_mul29v(ras, vas), fr_mul29v(was, fe_add(vas, incs))));
        }
        if (any) {
            size_t x_in = i & (((size_t)1 << in_bits) - 1), x_out = i >> in_bits;
            Fr eo = x_out < n_out ? fe_load<FrParams>(e_out + 4 * x_out) : Fr::one();
            Fr ei = x_in < n_in ? fe_load<FrParams>(e_in + 4 * x_in) : Fr::one();
            F29 ep = fr29_prescale(fr_mul29v(eo, ei));
            acc[0] = fr_mul29(c0, ep);
            acc[1] = fr_mul29(cx, ep);
        }
    }
    rrw_block_out<2>(acc, sh, partials);
}

// phase 2 (:764-852) and the register rounds once no cycle is left (:955-1013): thread t -> cycle j = t % cur_T, pair chunk t / cur_T;
// (e0, e2) += eq[j] * sum_i C at t = 0 / 2 of the row pair (2i, 2i + 1)
// E1: also the value at t = 1 (the odd rows), which Stage4Prover evaluates directly (stage4_prover.zig:666-706) instead of taking it from the claim
template <bool E1>
__global__ void __launch_bounds__(256) rrw_address_kernel(RrwTabs tb, size_t stride, const uint64_t *inc, const uint64_t *eq, size_t cur_T,
                                                          uint32_t half_K, uint32_t act_K, uint32_t ic_n, uint64_t *partials) {
    __shared__ uint4 sh[256 * 4];
    constexpr int NV = E1 ? 3 : 2;
    Fr acc[NV];
#pragma unroll
    for (int a = 0; a < NV; a++) acc[a] = Fr::zero();
    size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t < cur_T * ic_n) {
        size_t j = t % cur_T;
        uint32_t ic = (uint32_t)(t / cur_T);
        Fr incj = fe_load<FrParams>(inc + 4 * j);
        Fr c0 = Fr::zero(), c2 = Fr::zero(), c1 = Fr::zero();
        bool any = false;
        for (uint32_t i = ic; i < half_K; i += ic_n) {
            size_t oe = 4 * ((size_t)(2 * i) * stride + j), oo = oe + 4 * stride;
            const bool odd_row = 2 * i + 1 < act_K;  // past the active rows: zero
            Fr rae = fe_load<FrParams>(tb.t[RT_RA] + oe), rao = odd_row ? fe_load<FrParams>(tb.t[RT_RA] + oo) : Fr::zero();
            Fr wae = fe_load<FrParams>(tb.t[RT_WA] + oe), wao = odd_row ? fe_load<FrParams>(tb.t[RT_WA] + oo) : Fr::zero();
            if (fr_is_zero(rae) && fr_is_zero(rao) && fr_is_zero(wae) && fr_is_zero(wao)) continue;
            any = true;
            Fr vae = fe_load<FrParams>(tb.t[RT_VAL] + oe), vao = odd_row ? fe_load<FrParams>(tb.t[RT_VAL] + oo) : Fr::zero();
            c0 = fe_add(c0, fe_add(fr_mul29v(rae, vae), fr_mul29v(wae, fe_add(vae, incj))));
            Fr ra2 = fe_sub(fe_add(rao, rao), rae), wa2 = fe_sub(fe_add(wao, wao), wae), va2 = fe_sub(fe_add(vao, vao), vae);  // f(0) + 2 (f(1) - f(0))
            c2 = fe_add(c2, fe_add(fr_mul29v(ra2, va2), fr_mul29v(wa2, fe_add(va2, incj))));
            if constexpr (E1) c1 = fe_add(c1, fe_add(fr_mul29v(rao, vao), fr_mul29v(wao, fe_add(vao, incj))));
        }
        if (any) {
            F29 ep = fr29_prescale(fe_load<FrParams>(eq + 4 * j));
            acc[0] = fr_mul29(c0, ep);
            acc[1] = fr_mul29(c2, ep);
            if constexpr (E1) acc[2] = fr_mul29(c1, ep);
        }
    }
    rrw_block_out<NV>(acc, sh, partials);
}

// phase 3 with cycles left (:854-953): thread t -> cycle pair i, register chunk; (e0, e2, e3) += eq(t) * sum_k C(t), t = 0, 2, 3
template <bool E1>
__global__ void __launch_bounds__(256) rrw_cycle_dense_kernel(RrwTabs tb, size_t stride, const uint64_t *inc, const uint64_t *eq, size_t half_T,
                                                              uint32_t cur_K, uint32_t kc_n, const uint32_t *mask, uint64_t *partials) {
    __shared__ uint4 sh[256 * 4];
    constexpr int NV = E1 ? 4 : 3;
    Fr acc[NV];
#pragma unroll
    for (int a = 0; a < NV; a++) acc[a] = Fr::zero();
    size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t < half_T * kc_n) {
        size_t i = t % half_T;
        uint32_t kc = (uint32_t)(t / half_T);
        Fr inc0 = fe_load<FrParams>(inc + 8 * i), incs = fe_sub(fe_load<FrParams>(inc + 8 * i + 4), inc0);
        Fr inc2 = fe_add(fe_add(inc0, incs), incs), inc3 = fe_add(inc2, incs);
        Fr c0 = Fr::zero(), c2 = Fr::zero(), c3 = Fr::zero(), c1 = Fr::zero();
        bool any = false;
        uint32_t rows = 0;
        for (uint32_t k = kc; k < cur_K; k += kc_n) rows |= 1u << k;
        const uint32_t me = mask ? mask[2 * i] : ~0u, mo = mask ? mask[2 * i + 1] : ~0u;
        uint32_t m = (me | mo) & rows;
        while (m) {
            const uint32_t k = (uint32_t)__builtin_ctz(m);
            m &= m - 1;
            const bool he = (me >> k) & 1u, ho = (mo >> k) & 1u;
            size_t o = 4 * ((size_t)k * stride + 2 * i);
            const Fr z = Fr::zero();
            Fr rae = he ? fe_load<FrParams>(tb.t[RT_RA] + o) : z, rao = ho ? fe_load<FrParams>(tb.t[RT_RA] + o + 4) : z;
            Fr wae = he ? fe_load<FrParams>(tb.t[RT_WA] + o) : z, wao = ho ? fe_load<FrParams>(tb.t[RT_WA] + o + 4) : z;
            if (fr_is_zero(rae) && fr_is_zero(rao) && fr_is_zero(wae) && fr_is_zero(wao)) continue;
            any = true;
            Fr vae = fe_load<FrParams>(tb.t[RT_VAL] + o), vao = fe_load<FrParams>(tb.t[RT_VAL] + o + 4);
            Fr ras = fe_sub(rao, rae), was = fe_sub(wao, wae), vas = fe_sub(vao, vae);
            c0 = fe_add(c0, fe_add(fr_mul29v(rae, vae), fr_mul29v(wae, fe_add(vae, inc0))));
            if constexpr (E1) c1 = fe_add(c1, fe_add(fr_mul29v(rao, vao), fr_mul29v(wao, fe_add(vao, fe_add(inc0, incs)))));
            Fr ra2 = fe_add(rao, ras), wa2 = fe_add(wao, was), va2 = fe_add(vao, vas);
            c2 = fe_add(c2, fe_add(fr_mul29v(ra2, va2), fr_mul29v(wa2, fe_add(va2, inc2))));
            Fr ra3 = fe_add(ra2, ras), wa3 = fe_add(wa2, was), va3 = fe_add(va2, vas);
            c3 = fe_add(c3, fe_add(fr_mul29v(ra3, va3), fr_mul29v(wa3, fe_add(va3, inc3))));
        }
        if (any) {
            Fr eqe = fe_load<FrParams>(eq + 8 * i), eqs = fe_sub(fe_load<FrParams>(eq + 8 * i + 4), eqe);
            Fr eq2 = fe_add(fe_add(eqe, eqs), eqs), eq3 = fe_add(eq2, eqs);
            acc[0] = fr_mul29v(c0, eqe);
            acc[1] = fr_mul29v(c2, eq2);
            acc[2] = fr_mul29v(c3, eq3);
            if constexpr (E1) acc[3] = fr_mul29v(c1, fe_add(eqe, eqs));
        }
    }
    rrw_block_out<NV>(acc, sh, partials);
}

ZG_DEV Fr rrw_fold1(const Fr &lo, const Fr &hi, const FrMul &rm) {
    Fr d = fe_sub(hi, lo);
    if (fr_is_zero(d)) return lo;  // (also every pair of zeros of the one-hot tables)
    return fe_add(lo, rm.narrow ? frmul_apply(d, rm) : fr_mul29(d, rm.p));
}

// the same fold of the four one-hot-born tables under the row masks: thread i folds the masked rows of cycle pair i and writes mask'[i]
__global__ void __launch_bounds__(256) rrw_fold_cycle_masked_kernel(RrwTabs in, size_t stride, RrwTabsOut out, size_t half_T, const uint32_t *mask,
                                                                    uint32_t *mask_out, FrArg r) {
    const FrMul rm = frmul_prepare(fr_from_arg(r));
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= 4 * half_T) return;
    const size_t i = t % half_T;
    const uint32_t g = (uint32_t)(t / half_T);  // row group: rows 8 g .. 8 g + 7
    const uint32_t me = mask[2 * i], mo = mask[2 * i + 1];
    if (g == 0) mask_out[i] = me | mo;
    uint32_t m = (me | mo) & (0xFFu << (8 * g));
    const Fr z = Fr::zero();
    while (m) {
        const uint32_t k = (uint32_t)__builtin_ctz(m);
        m &= m - 1;
        const bool he = (me >> k) & 1u, ho = (mo >> k) & 1u;
        const size_t o = 4 * ((size_t)k * stride + 2 * i), d = 4 * ((size_t)k * half_T + i);
#pragma unroll
        for (int tt = RT_WA; tt <= RT_RS2; tt++) {
            Fr lo = he ? fe_load<FrParams>(in.t[tt] + o) : z, hi = ho ? fe_load<FrParams>(in.t[tt] + o + 4) : z;
            fe_store(out.t[tt] + d, rrw_fold1(lo, hi, rm));
        }
    }
}

// cycle fold (:1053-1090, 1124-1162): out[k][i] = lo (1 - c) + hi c; rows compacted to half_T. blockIdx.y + t0 = table.
__global__ void __launch_bounds__(256) rrw_fold_cycle_kernel(RrwTabs in, size_t stride, RrwTabsOut out, size_t half_T, uint32_t cur_K, FrArg r, int t0) {
    const FrMul rm = frmul_prepare(fr_from_arg(r));
    const uint64_t *src = in.t[blockIdx.y + t0];
    uint64_t *dst = out.t[blockIdx.y + t0];
    size_t n = (size_t)cur_K * half_T, step = (size_t)gridDim.x * 256;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n; t += step) {
        size_t k = t / half_T, i = t - k * half_T;
        const uint64_t *p = src + 4 * (k * stride + 2 * i);
        fe_store(dst + 4 * (k * half_T + i), rrw_fold1(fe_load<FrParams>(p), fe_load<FrParams>(p + 4), rm));
    }
}

// the vectors over the cycles (inc, and the merged eq table in phase 3), same fold
__global__ void __launch_bounds__(256) rrw_fold_vec_kernel(const uint64_t *a, uint64_t *a_out, const uint64_t *b, uint64_t *b_out, size_t half, FrArg r) {
    const FrMul rm = frmul_prepare(fr_from_arg(r));
    size_t step = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < half; i += step) {
        fe_store(a_out + 4 * i, rrw_fold1(fe_load<FrParams>(a + 8 * i), fe_load<FrParams>(a + 8 * i + 4), rm));
        if (b) fe_store(b_out + 4 * i, rrw_fold1(fe_load<FrParams>(b + 8 * i), fe_load<FrParams>(b + 8 * i + 4), rm));
    }
}

// register fold (:1092-1122): out[i][j] = row 2i (1 - c) + row 2i+1 c
__global__ void __launch_bounds__(256) rrw_fold_address_kernel(RrwTabs in, size_t stride, RrwTabsOut out, size_t cur_T, uint32_t half_K, uint32_t act_K,
                                                               FrArg r) {
    const FrMul rm = frmul_prepare(fr_from_arg(r));
    const uint64_t *src = in.t[blockIdx.y];
    uint64_t *dst = out.t[blockIdx.y];
    size_t n = (size_t)half_K * cur_T, step = (size_t)gridDim.x * 256;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n; t += step) {
        size_t i = t / cur_T, j = t - i * cur_T;
        const uint64_t *p = src + 4 * ((2 * i) * stride + j);
        fe_store(dst + 4 * (i * cur_T + j), rrw_fold1(fe_load<FrParams>(p), 2 * i + 1 < act_K ? fe_load<FrParams>(p + 4 * stride) : Fr::zero(), rm));
    }
}

}  // namespace zg

struct zg_rrw_s {
    int device = -1;
    size_t T = 0, cur_T = 0, stride = 0;
    uint32_t cur_K = zg::RRW_K, act_K = zg::RRW_ACTIVE;  // current_K of the reference / rows that can be non-zero (stored)
    uint64_t *tab[zg::RRW_TABLES][2] = {};  // [table][buffer]: buffer 0 holds K x T elements, buffer 1 K x T / 2
    uint64_t *inc[2] = {nullptr, nullptr}, *eq[2] = {nullptr, nullptr};
    int cur = 0, vcur = 0;  // live buffer of the tables / of the two cycle vectors
    bool have_eq = false;
    uint32_t *mask[2] = {nullptr, nullptr};  // row masks of the live cycles (see rrw_mask_build_kernel); mcur = live buffer
    int mcur = 0;
    bool mask_ok = true, garbage = false;    // masks describe the four tables / entries outside them have not been written
    int masked_folds = 0;
    uint64_t *d_part = nullptr, *d_out = nullptr, *h_out = nullptr;
    hipStream_t st = nullptr;
    std::mutex mu;
};

using namespace zg;

// session buffers come from the library's device pool and the pinned pool (runtime.hip): the ten tables of a 2^18-cycle session are 2 GB,
// and hipMalloc + hipFree of those cost 2-40 ms per open depending on the box (BENCH_r04: 45 ms of set-up against 4.8 ms)
static void rrw_free(zg_rrw_s *s) {
    if (!s) return;
    for (int t = 0; t < RRW_TABLES; t++)
        for (int b = 0; b < 2; b++) pool_free(s->tab[t][b]);
    for (int b = 0; b < 2; b++) {
        pool_free(s->inc[b]);
        pool_free(s->eq[b]);
    }
    for (int b = 0; b < 2; b++) pool_free(s->mask[b]);
    pool_free(s->d_part);
    pool_free(s->d_out);
    pinned_put(s->h_out);
    if (s->st) stream_release(s->st, s->device);
    delete s;
}

static FrArg fr_arg(const uint64_t r[4]) {
    FrArg a;
    for (int i = 0; i < 4; i++) {
        a.l[2 * i] = (uint32_t)r[i];
        a.l[2 * i + 1] = (uint32_t)(r[i] >> 32);
    }
    return a;
}

static RrwTabs rrw_tabs(const zg_rrw_s *s) {
    RrwTabs t;
    for (int i = 0; i < RRW_TABLES; i++) t.t[i] = s->tab[i][s->cur];
    return t;
}
static RrwTabsOut rrw_tabs_out(const zg_rrw_s *s, int buf) {
    RrwTabsOut t;
    for (int i = 0; i < RRW_TABLES; i++) t.t[i] = s->tab[i][buf];
    return t;
}

// threads per round launch: one per (pair / cycle, chunk); chunks spread the register loop when the cycle dimension alone is short
static uint32_t rrw_chunks(size_t inner, uint32_t outer) {
    static const size_t want = [] {
        const char *e = getenv("ZG_RRW_THREADS");
        long v = e ? atol(e) : 0;
        return (size_t)(v >= 1024 && v <= (1l << 26) ? v : 65536);
    }();
    uint32_t c = 1;
    while (c < outer && inner * c < want) c <<= 1;
    return c > outer ? outer : c;
}

// sum the block partials and bring nv values to the host
static int rrw_collect(zg_rrw_s *s, uint32_t nblocks, int nv, uint64_t *out) {
    hipLaunchKernelGGL(rrw_finish_kernel, dim3(1), dim3(256), 0, s->st, s->d_part, nblocks, nv, s->d_out);
    ZG_HIP(hipGetLastError());
    ZG_HIP(hipMemcpyAsync(s->h_out, s->d_out, (size_t)nv * 32, hipMemcpyDeviceToHost, s->st));
    ZG_HIP(hipStreamSynchronize(s->st));
    for (int i = 0; i < 4 * nv; i++) out[i] = s->h_out[i];
    return ZG_OK;
}

static int masked_folds_max() {
    static const int v = [] {
        const char *e = getenv("ZG_RRW_MASKED_FOLDS");
        int x = e ? atoi(e) : 3;
        return x < 0 ? 0 : (x > 30 ? 30 : x);
    }();
    return v;
}
// before the first reader that ignores the row masks: zero what the masked folds did not write
static int rrw_materialize(zg_rrw_s *s) {
    if (!s->garbage) return ZG_OK;
    size_t n = (size_t)s->act_K * s->cur_T;
    hipLaunchKernelGGL(rrw_materialize_kernel, dim3(div_up(n, 256)), dim3(256), 0, s->st, rrw_tabs_out(s, s->cur), s->stride, s->mask[s->mcur], s->cur_T, s->act_K);
    ZG_HIP(hipGetLastError());
    s->garbage = false;
    return ZG_OK;
}

extern "C" {

// reg_vals + inc from the host (zg_rrw_open), or rd_value alone and both rebuilt on the device (zg_rrw_open_trace)
static int rrw_open_impl(size_t log_t, const uint8_t *rs1, const uint8_t *rs2, const uint8_t *rd, const uint64_t *reg_vals, const uint64_t *inc,
                         const uint64_t *rd_value, const uint64_t gamma[4], zg_rrw_t *out) {
    ZG_INIT();
    if (!out || !rs1 || !rs2 || !rd || (rd_value ? false : !reg_vals || !inc) || !gamma || log_t < 1 || log_t > 24) {
        set_error("zg_rrw_open: invalid argument (1 <= log_t <= 24: five 32 x 2^log_t tables of active rows and their half-size partners)");
        return ZG_ERR_INVALID;
    }
    const size_t T = (size_t)1 << log_t;
    zg_rrw_s *s = new zg_rrw_s();
    s->device = current_device();
    s->T = s->cur_T = s->stride = T;
    s->st = stream_acquire();
    const bool split = setup_times_enabled();
    const double ts0 = split ? now_ms() : 0;
    bool ok = s->st != nullptr;
    auto grab = [&](auto *&ptr, size_t bytes) {
        if (ok) ok = (ptr = reinterpret_cast<std::remove_reference_t<decltype(ptr)>>(pool_alloc(bytes))) != nullptr;
    };
    for (int t = 0; t < RRW_TABLES; t++) {
        grab(s->tab[t][0], (size_t)RRW_ACTIVE * T * 32);
        grab(s->tab[t][1], (size_t)RRW_ACTIVE * (T / 2) * 32);
    }
    for (int b = 0; b < 2; b++) {
        grab(s->inc[b], (T >> b) * 32);
        grab(s->eq[b], (T >> b) * 32);
    }
    for (int b = 0; b < 2; b++) grab(s->mask[b], (T >> b) * 4);
    grab(s->d_part, (size_t)RRW_MAX_BLOCKS * 4 * 32);
    grab(s->d_out, 8 * 32);
    if (ok) ok = (s->h_out = reinterpret_cast<uint64_t *>(pinned_get(8 * 32))) != nullptr;
    if (!ok) {
        if (!s->st) set_error("zg_rrw_open: no stream");
        std::string keep = zg_last_error();
        rrw_free(s);
        set_error("zg_rrw_open: " + keep);
        return ZG_ERR_NOMEM;
    }
    // the trace columns travel in one scratch buffer: rs1 | rs2 | rd (T bytes each, padded) then the 32 x T register values
    const size_t pad = (T + 255) & ~(size_t)255;
    const size_t nch = (T + 63) / 64;
    Scratch s_cols(3 * pad), s_vals(32 * T * 8), s_trace(rd_value ? T * 8 + 32 * nch * 4 : 8);
    if (!s_cols.p || !s_vals.p || !s_trace.p) {
        rrw_free(s);
        return ZG_ERR_NOMEM;
    }
    const double ts1 = split ? now_ms() : 0;
    double ts2 = ts1;
    int rc = [&]() -> int {
        SyncGuard sync(s->st);
        uint8_t *d_cols = s_cols.as<uint8_t>();
        ZG_HIP(hipMemcpyAsync(d_cols, rs1, T, hipMemcpyHostToDevice, s->st));
        ZG_HIP(hipMemcpyAsync(d_cols + pad, rs2, T, hipMemcpyHostToDevice, s->st));
        ZG_HIP(hipMemcpyAsync(d_cols + 2 * pad, rd, T, hipMemcpyHostToDevice, s->st));
        if (rd_value) {
            uint64_t *d_rdv = s_trace.as<uint64_t>();
            int32_t *d_last = reinterpret_cast<int32_t *>(d_rdv + T);
            ZG_HIP(hipMemcpyAsync(d_rdv, rd_value, T * 8, hipMemcpyHostToDevice, s->st));
            if (split) {
                ZG_HIP(hipStreamSynchronize(s->st));
                ts2 = now_ms();
            }
            hipLaunchKernelGGL(rrw_last_write_kernel, dim3(div_up(nch * 64, 256)), dim3(256), 0, s->st, d_cols + 2 * pad, T, d_last, nch);
            hipLaunchKernelGGL(rrw_carry_scan_kernel, dim3(32), dim3(256), 0, s->st, d_last, nch);
            hipLaunchKernelGGL(rrw_regfile_kernel, dim3(div_up(nch * 64, 256)), dim3(256), 0, s->st, d_cols + 2 * pad, d_rdv, T, d_last, nch,
                               s_vals.as<uint64_t>(), s->inc[0]);
        } else {
            ZG_HIP(hipMemcpyAsync(s_vals.p, reg_vals, 32 * T * 8, hipMemcpyHostToDevice, s->st));
            ZG_HIP(hipMemcpyAsync(s->inc[0], inc, T * 32, hipMemcpyHostToDevice, s->st));
        }
        size_t n = (size_t)RRW_ACTIVE * T;
        hipLaunchKernelGGL(rrw_build_kernel, dim3(div_up(n, 256)), dim3(256), 0, s->st, d_cols, d_cols + pad, d_cols + 2 * pad, s_vals.as<uint64_t>(), T,
                           fr_arg(gamma), rrw_tabs_out(s, 0));
        hipLaunchKernelGGL(rrw_mask_build_kernel, dim3(div_up(T, 256)), dim3(256), 0, s->st, d_cols, d_cols + pad, d_cols + 2 * pad, T, s->mask[0]);
        ZG_HIP(hipGetLastError());
        ZG_HIP(hipStreamSynchronize(s->st));
        sync.dismiss();
        return ZG_OK;
    }();
    if (rc != ZG_OK) {
        std::string keep = zg_last_error();
        (void)hipStreamSynchronize(s->st);
        rrw_free(s);
        set_error(keep);
        return rc;
    }
    if (split) {
        SetupTimes &tm = setup_times();
        tm = SetupTimes{};
        tm.alloc_ms = ts1 - ts0;
        tm.h2d_ms = ts2 - ts1;
        tm.kernel_ms = now_ms() - ts2;
    }
    *out = s;
    return ZG_OK;
}
int zg_rrw_open(size_t log_t, const uint8_t *rs1, const uint8_t *rs2, const uint8_t *rd, const uint64_t *reg_vals, const uint64_t *inc,
                const uint64_t gamma[4], zg_rrw_t *out) {
    return rrw_open_impl(log_t, rs1, rs2, rd, reg_vals, inc, nullptr, gamma, out);
}
int zg_rrw_open_trace(size_t log_t, const uint8_t *rs1, const uint8_t *rs2, const uint8_t *rd, const uint64_t *rd_value, const uint64_t gamma[4],
                      zg_rrw_t *out) {
    if (!rd_value) {
        set_error("zg_rrw_open_trace: invalid argument");
        return ZG_ERR_INVALID;
    }
    return rrw_open_impl(log_t, rs1, rs2, rd, nullptr, nullptr, rd_value, gamma, out);
}

size_t zg_rrw_cycles(zg_rrw_t s) { return s ? s->cur_T : 0; }
size_t zg_rrw_registers(zg_rrw_t s) { return s ? s->cur_K : 0; }

int zg_rrw_round_cycle_gruen(zg_rrw_t s, const uint64_t *d_e_out, size_t n_out, const uint64_t *d_e_in, size_t n_in, uint64_t q0[4], uint64_t qx2[4]) {
    ZG_INIT();
    if (!s || !q0 || !qx2 || s->cur_T < 2 || !d_e_out || !d_e_in || n_in == 0 || (n_in & (n_in - 1))) {
        set_error("zg_rrw_round_cycle_gruen: invalid argument (|E_in| a power of two, at least two cycles left)");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    const size_t half = s->cur_T / 2;
    uint32_t in_bits = 0;
    while (((size_t)1 << in_bits) < n_in) in_bits++;
    uint32_t kc = rrw_chunks(half, s->act_K);
    uint32_t nb = div_up(half * kc, 256);
    if (nb > RRW_MAX_BLOCKS) {
        set_error("zg_rrw_round_cycle_gruen: table too long");
        return ZG_ERR_INVALID;
    }
    hipLaunchKernelGGL(rrw_cycle_gruen_kernel, dim3(nb), dim3(256), 0, s->st, rrw_tabs(s), s->stride, s->inc[s->vcur], d_e_out, (uint32_t)n_out, d_e_in,
                       (uint32_t)n_in, in_bits, half, s->act_K, kc, s->mask_ok ? s->mask[s->mcur] : (const uint32_t *)nullptr, s->d_part);
    ZG_HIP(hipGetLastError());
    uint64_t o[8];
    ZG_TRY(rrw_collect(s, nb, 2, o));
    for (int i = 0; i < 4; i++) {
        q0[i] = o[i];
        qx2[i] = o[4 + i];
    }
    return ZG_OK;
}

int zg_rrw_set_eq(zg_rrw_t s, const uint64_t *eq, size_t n) {
    ZG_INIT();
    if (!s || !eq || n != s->cur_T) {
        set_error("zg_rrw_set_eq: the merged eq table must have one entry per live cycle");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    ZG_HIP(hipMemcpyAsync(s->eq[s->vcur], eq, n * 32, hipMemcpyHostToDevice, s->st));
    ZG_HIP(hipStreamSynchronize(s->st));
    s->have_eq = true;
    return ZG_OK;
}

int zg_rrw_round_address(zg_rrw_t s, uint64_t e0[4], uint64_t *e1, uint64_t e2[4]) {
    ZG_INIT();
    if (!s || !e0 || !e2 || s->cur_K < 2 || !s->have_eq) {
        set_error("zg_rrw_round_address: needs the eq table (zg_rrw_set_eq) and at least two registers left");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    ZG_TRY(rrw_materialize(s));
    const uint32_t half_K = (s->act_K + 1) / 2;  // row pairs with an active member; the pairs past them are zero
    uint32_t ic = rrw_chunks(s->cur_T, half_K);
    uint32_t nb = div_up(s->cur_T * ic, 256);
    if (nb > RRW_MAX_BLOCKS) {
        set_error("zg_rrw_round_address: table too long");
        return ZG_ERR_INVALID;
    }
    if (e1)
        hipLaunchKernelGGL(rrw_address_kernel<true>, dim3(nb), dim3(256), 0, s->st, rrw_tabs(s), s->stride, s->inc[s->vcur], s->eq[s->vcur], s->cur_T, half_K,
                           s->act_K, ic, s->d_part);
    else
        hipLaunchKernelGGL(rrw_address_kernel<false>, dim3(nb), dim3(256), 0, s->st, rrw_tabs(s), s->stride, s->inc[s->vcur], s->eq[s->vcur], s->cur_T, half_K,
                           s->act_K, ic, s->d_part);
    ZG_HIP(hipGetLastError());
    uint64_t o[12];
    ZG_TRY(rrw_collect(s, nb, e1 ? 3 : 2, o));
    for (int i = 0; i < 4; i++) {
        e0[i] = o[i];
        e2[i] = o[4 + i];
        if (e1) e1[i] = o[8 + i];
    }
    return ZG_OK;
}

int zg_rrw_round_cycle(zg_rrw_t s, uint64_t e0[4], uint64_t *e1, uint64_t e2[4], uint64_t e3[4]) {
    ZG_INIT();
    if (!s || !e0 || !e2 || !e3 || s->cur_T < 2 || !s->have_eq) {
        set_error("zg_rrw_round_cycle: needs the eq table (zg_rrw_set_eq) and at least two cycles left");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    const size_t half = s->cur_T / 2;
    uint32_t kc = rrw_chunks(half, s->act_K);
    uint32_t nb = div_up(half * kc, 256);
    if (nb > RRW_MAX_BLOCKS) {
        set_error("zg_rrw_round_cycle: table too long");
        return ZG_ERR_INVALID;
    }
    if (e1)
        hipLaunchKernelGGL(rrw_cycle_dense_kernel<true>, dim3(nb), dim3(256), 0, s->st, rrw_tabs(s), s->stride, s->inc[s->vcur], s->eq[s->vcur], half, s->act_K, kc,
                           s->mask_ok ? s->mask[s->mcur] : (const uint32_t *)nullptr, s->d_part);
    else
        hipLaunchKernelGGL(rrw_cycle_dense_kernel<false>, dim3(nb), dim3(256), 0, s->st, rrw_tabs(s), s->stride, s->inc[s->vcur], s->eq[s->vcur], half, s->act_K, kc,
                           s->mask_ok ? s->mask[s->mcur] : (const uint32_t *)nullptr, s->d_part);
    ZG_HIP(hipGetLastError());
    uint64_t o[16];
    ZG_TRY(rrw_collect(s, nb, e1 ? 4 : 3, o));
    for (int i = 0; i < 4; i++) {
        e0[i] = o[i];
        e2[i] = o[4 + i];
        e3[i] = o[8 + i];
        if (e1) e1[i] = o[12 + i];
    }
    return ZG_OK;
}

int zg_rrw_bind_cycle(zg_rrw_t s, const uint64_t r[4]) {
    ZG_INIT();
    if (!s || !r || s->cur_T < 2) {
        set_error("zg_rrw_bind_cycle: no cycle variable left");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    const size_t half = s->cur_T / 2;
    const int nxt = s->cur ^ 1, vn = s->vcur ^ 1;
    size_t n = (size_t)s->act_K * half;
    unsigned nb = div_up(n, 256);
    if (nb > 16384) nb = 16384;
    // the masks pay while they are sparse: <= 3 rows of 32 per cycle at the start, doubling per fold at worst. After RRW_MASKED_FOLDS folds
    // (or when the table is short) the unwritten entries are zeroed once and everything is dense again.
    if (s->mask_ok && (s->masked_folds >= masked_folds_max() || half < 4096)) {
        ZG_TRY(rrw_materialize(s));
        s->mask_ok = false;
    }
    if (s->mask_ok) {  // val densely, the four one-hot-born tables under the row masks
        s->masked_folds++;
        hipLaunchKernelGGL(rrw_fold_cycle_kernel, dim3(nb, 1), dim3(256), 0, s->st, rrw_tabs(s), s->stride, rrw_tabs_out(s, nxt), half, s->act_K, fr_arg(r), RT_VAL);
        hipLaunchKernelGGL(rrw_fold_cycle_masked_kernel, dim3(div_up(4 * half, 256)), dim3(256), 0, s->st, rrw_tabs(s), s->stride, rrw_tabs_out(s, nxt), half,
                           s->mask[s->mcur], s->mask[s->mcur ^ 1], fr_arg(r));
        s->mcur ^= 1;
        s->garbage = true;
    } else {
        hipLaunchKernelGGL(rrw_fold_cycle_kernel, dim3(nb, RRW_TABLES), dim3(256), 0, s->st, rrw_tabs(s), s->stride, rrw_tabs_out(s, nxt), half, s->act_K, fr_arg(r), 0);
    }
    unsigned nv = div_up(half, 256);
    if (nv > 4096) nv = 4096;
    hipLaunchKernelGGL(rrw_fold_vec_kernel, dim3(nv), dim3(256), 0, s->st, s->inc[s->vcur], s->inc[vn], s->have_eq ? s->eq[s->vcur] : (const uint64_t *)nullptr,
                       s->eq[vn], half, fr_arg(r));
    ZG_HIP(hipGetLastError());
    s->cur = nxt;
    s->vcur = vn;
    s->cur_T = half;
    s->stride = half;
    return ZG_OK;
}

int zg_rrw_bind_address(zg_rrw_t s, const uint64_t r[4]) {
    ZG_INIT();
    if (!s || !r || s->cur_K < 2) {
        set_error("zg_rrw_bind_address: no register variable left");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    ZG_TRY(rrw_materialize(s));
    s->mask_ok = false;  // rows merge: the cycle masks no longer describe the tables (which are small by now)
    const uint32_t half_act = (s->act_K + 1) / 2;
    const int nxt = s->cur ^ 1;
    size_t n = (size_t)half_act * s->cur_T;
    unsigned nb = div_up(n, 256);
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(rrw_fold_address_kernel, dim3(nb, RRW_TABLES), dim3(256), 0, s->st, rrw_tabs(s), s->stride, rrw_tabs_out(s, nxt), s->cur_T, half_act,
                       s->act_K, fr_arg(r));
    ZG_HIP(hipGetLastError());
    s->cur = nxt;
    s->cur_K /= 2;
    s->act_K = half_act;
    s->stride = s->cur_T;
    return ZG_OK;
}

int zg_rrw_final(zg_rrw_t s, uint64_t *out) {
    ZG_INIT();
    if (!s || !out) {
        set_error("zg_rrw_final: invalid argument");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    // entry [0][0] of val, wa, ra, rs1_ra, rs2_ra, then inc[0] and eq[0] (zero when no eq table was set)
    for (int t = 0; t < RRW_TABLES; t++) ZG_HIP(hipMemcpyAsync(out + 4 * t, s->tab[t][s->cur], 32, hipMemcpyDeviceToHost, s->st));
    uint32_t m0 = ~0u;  // while the row masks are in force an entry outside them has not been written: it is zero
    if (s->garbage) ZG_HIP(hipMemcpyAsync(&m0, s->mask[s->mcur], 4, hipMemcpyDeviceToHost, s->st));
    ZG_HIP(hipMemcpyAsync(out + 20, s->inc[s->vcur], 32, hipMemcpyDeviceToHost, s->st));
    if (s->have_eq) ZG_HIP(hipMemcpyAsync(out + 24, s->eq[s->vcur], 32, hipMemcpyDeviceToHost, s->st));
    else for (int i = 0; i < 4; i++) out[24 + i] = 0;
    ZG_HIP(hipStreamSynchronize(s->st));
    if (!(m0 & 1u))
        for (int i = 4; i < 20; i++) out[i] = 0;
    return ZG_OK;
}

int zg_rrw_close(zg_rrw_t s) {
    if (!s) return ZG_OK;
    ZG_INIT();
    DeviceGuard dg(s->device);
    (void)hipStreamSynchronize(s->st);
    rrw_free(s);
    return ZG_OK;
}

}  // extern "C"
