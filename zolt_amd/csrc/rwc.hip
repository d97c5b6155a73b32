// rwc.hip — RamReadWriteCheckingProver (/root/reference/src/zkvm/ram/read_write_checking.zig:160-1323) with the field arithmetic of its
// SPARSE half on the device.
//
// The prover keeps a list of access entries (cycle, address, ra_coeff, val_coeff, prev_val, next_val: CycleMajorEntry, :91-157) next to
// three dense tables (eq_evals and inc over the cycles, val_init over the addresses) and runs log_t + log_k rounds in three phases. The
// reference walks the list with sequential loops: a pair merge per round in the cycle phases (:410-536, 1139-1185), a two-pointer walk
// over the even and the odd column of every column pair with carried checkpoints in the address phase (:538-769, 973-1137). Those walks
// decide WHO pairs with whom from the integer fields alone — cycle, address, and (for the checkpoints) which entry was consumed last —
// and then spend a dozen field products per step. The address walk cannot be turned into rank queries: the reference groups by
// (address >> addr_round) although it already halves every address at each bind, so from the second address round on a "column" holds
// several addresses and is not cycle-sorted; what its two-pointer loop does on such lists is defined only by running it. So the split is:
//   the integer skeleton of the list (cycle, address, prev_val, next_val) and, once per round, the walk over it — a PLAN of steps
//         {entry a, entry b or none, kind, key, implicit value or checkpoint column}, O(entries) integer work. The CYCLE-phase walk is a local
//         rule (an entry is the odd member of a pair exactly when its predecessor is the even entry of the same cycle pair and address; an
//         even entry always opens a step), so it runs on the device: head flags, a scan, one kernel that writes the plan and the bound
//         skeleton (round 4: the host walk plus the upload of a 24-byte step per entry cost ~2.5 ms per round at 2^18 entries). The
//         ADDRESS-phase walk stays on the host inside this library (see above); the skeleton moves between the two sides at the phase switches;
//   device: ra_coeff / val_coeff of every entry and the three dense tables; per round one kernel turns the plan into the two sums
//         (thread per step), one kernel into the bound entries (step k writes entry k: the plan order IS the reference's output order),
//         one folds the dense tables.
// Every value is the reference's, including two of its literal behaviours: the double address shift above, and val_init being folded in
// place BEFORE the address bind reads its checkpoints with the old size (:953-962, 974-996 — columns below size / 2 hold the new level,
// the others still the old one; both levels stay resident here).
#include <algorithm>
#include <mutex>
#include <vector>

#include "common.hip.h"
#include "field.hip.h"
#include "fp29.hip.h"
#include "sc_common.hip.h"

namespace zg {

enum : uint32_t { RWC_PAIR = 0, RWC_EVEN_ALONE = 1, RWC_ODD_ALONE = 2, RWC_KIND_MASK = 3, RWC_IMP_IS_COLUMN = 4 };
struct RwcStep {     // one step of the reference's walk
    uint32_t a, b;   // entry a (the even member of a pair / the lone entry), entry b (the odd member) or ~0u
    uint32_t key;    // cycle phases: the cycle pair index (cycle >> 1); address phase: the consumed entry's address (eqAddr, :571-584)
    uint32_t kind;   // RWC_PAIR / RWC_EVEN_ALONE / RWC_ODD_ALONE, | RWC_IMP_IS_COLUMN
    uint64_t imp;    // the implicit member's value as u64 (next_val / prev_val of an entry), or a val_init column (RWC_IMP_IS_COLUMN)
};
static constexpr unsigned RWC_MAX_BLOCKS = 65536;

ZG_DEV Fr rwc_arg(const FrArg &a) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = a.l[i];
    return r;
}
ZG_DEV Fr fr_from_u64_dev(uint64_t u) { return fr_from_u64_29(u); }  // F.fromU64 (short product, fp29.hip.h)

// ra_coeff = 1, val_coeff = F.fromU64(.) of the initial entries (:300-330)
__global__ void __launch_bounds__(256) rwc_init_kernel(const uint64_t *val_u64, uint32_t n, uint64_t *ra, uint64_t *val) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    fe_store(ra + 4 * (size_t)i, Fr::one());
    fe_store(val + 4 * (size_t)i, fr_from_u64_dev(val_u64[i]));
}

// inc of the written cycles from the entries themselves (zg_rwc_open_writes): F.fromU64(next) - F.fromU64(prev) where an entry is a write
// (:283-291 forms the same difference from the larger of the two), the rest of the table zeroed beforehand
__global__ void __launch_bounds__(256) rwc_inc_scatter_kernel(const uint32_t *cycle, const uint64_t *prev, const uint64_t *next, const uint8_t *is_write,
                                                              uint32_t n, uint64_t *inc) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !is_write[i]) return;
    fe_store(inc + 4 * (size_t)cycle[i], fe_sub(fr_from_u64_dev(next[i]), fr_from_u64_dev(prev[i])));
}

__device__ __forceinline__ void rwc_block_out(Fr (&acc)[2], uint4 *sh, uint64_t *partials) {
    block_sum_pair(acc[0], acc[1], sh);
    if (threadIdx.x == 0) {
        fe_store(partials + 8 * (size_t)blockIdx.x, acc[0]);
        fe_store(partials + 8 * (size_t)blockIdx.x + 4, acc[1]);
    }
}
__global__ void __launch_bounds__(256) rwc_finish_kernel(const uint64_t *partials, uint32_t nblocks, uint64_t *out) {
    __shared__ uint4 sh[256 * 4];
    Fr g0 = Fr::zero(), g1 = Fr::zero();
    for (uint32_t b = threadIdx.x; b < nblocks; b += 256) {
        g0 = fe_add(g0, fe_load<FrParams>(partials + 8 * (size_t)b));
        g1 = fe_add(g1, fe_load<FrParams>(partials + 8 * (size_t)b + 4));
    }
    block_sum_pair(g0, g1, sh);
    if (threadIdx.x == 0) {
        fe_store(out, g0);
        fe_store(out + 4, g1);
    }
}

// ---- the cycle-phase walk (computePhase1Polynomial's pairing :431-470, bindEntries :1146-1160) as a local rule. The reference scans the
// list once: an even-cycle entry pairs with its SUCCESSOR when that one holds the odd cycle of the same pair at the same address, every
// other entry stands alone. An odd entry can only be consumed by its predecessor and an even entry never is, so "entry i opens a step" needs
// entries i - 1 and i only.
static constexpr uint32_t RWC_WALK_BLOCK = 1024;
ZG_DEV bool rwc_is_odd_member(const uint32_t *cycle, const uint32_t *addr, uint32_t i) {
    if (i == 0) return false;
    const uint32_t c = cycle[i], cp = cycle[i - 1];
    return (c & 1u) && !(cp & 1u) && (cp >> 1) == (c >> 1) && addr[i - 1] == addr[i];
}
// steps opened inside each block of RWC_WALK_BLOCK entries
__global__ void __launch_bounds__(RWC_WALK_BLOCK) rwc_walk_count_kernel(const uint32_t *cycle, const uint32_t *addr, uint32_t n, uint32_t *blk) {
    const uint32_t i = blockIdx.x * RWC_WALK_BLOCK + threadIdx.x;
    const int heads = __syncthreads_count(i < n && !rwc_is_odd_member(cycle, addr, i));
    if (threadIdx.x == 0) blk[blockIdx.x] = (uint32_t)heads;
}
// exclusive scan of the block counts in place (one workgroup; <= 2^14 blocks for 2^24 entries), the total behind them
__global__ void __launch_bounds__(1024) rwc_walk_scan_kernel(uint32_t *blk, uint32_t nblk) {
    __shared__ uint32_t sh[1024];
    const uint32_t t = threadIdx.x, per = (nblk + 1023) / 1024;
    const uint32_t a = t * per < nblk ? t * per : nblk, b = a + per < nblk ? a + per : nblk;
    uint32_t sum = 0;
    for (uint32_t j = a; j < b; j++) sum += blk[j];
    sh[t] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint32_t v = t >= d ? sh[t - d] : 0u;
        __syncthreads();
        sh[t] += v;
        __syncthreads();
    }
    uint32_t run = sh[t] - sum;
    for (uint32_t j = a; j < b; j++) {
        const uint32_t v = blk[j];
        blk[j] = run;
        run += v;
    }
    if (t == 1023) blk[nblk] = sh[1023];
}
// step k of the plan and entry k of the bound skeleton (cycle halved, the address kept, prev_val of the first member, next_val of the last:
// CycleMajorEntry.bindEntries :110-156)
__global__ void __launch_bounds__(RWC_WALK_BLOCK) rwc_walk_plan_kernel(const uint32_t *cycle, const uint32_t *addr, const uint64_t *prev, const uint64_t *next,
                                                                       uint32_t n, const uint32_t *blk, RwcStep *plan, uint32_t *cycle2, uint32_t *addr2,
                                                                       uint64_t *prev2, uint64_t *next2) {
    __shared__ uint32_t wsum[RWC_WALK_BLOCK / 64];
    const uint32_t i = blockIdx.x * RWC_WALK_BLOCK + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool head = i < n && !rwc_is_odd_member(cycle, addr, i);
    const uint64_t bal = __ballot(head);
    uint32_t pre = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    for (uint32_t w = 0; w < wave; w++) pre += wsum[w];
    if (!head) return;
    const uint32_t k = blk[blockIdx.x] + pre, c = cycle[i];
    RwcStep st{i, ~0u, c >> 1, RWC_ODD_ALONE, 0};
    uint64_t nxt = next[i];
    if (!(c & 1u)) {
        if (i + 1 < n && rwc_is_odd_member(cycle, addr, i + 1)) {
            st.b = i + 1;
            st.kind = RWC_PAIR;
            nxt = next[i + 1];
        } else {
            st.kind = RWC_EVEN_ALONE;
            st.imp = nxt;  // the odd member is implicit: the value after this access
        }
    } else {
        st.imp = prev[i];  // the even member is implicit: the value before it
    }
    plan[k] = st;
    cycle2[k] = c >> 1;
    addr2[k] = addr[i];
    prev2[k] = prev[i];
    next2[k] = nxt;
}

// the members of a cycle-phase step at t = 0 and their slopes (:431-470): an even entry alone meets the value AFTER its access, an odd
// one the value BEFORE it
ZG_DEV void rwc_cycle_members(const RwcStep &s, const uint64_t *ra, const uint64_t *val, Fr &ra_e, Fr &ra_o, Fr &val_e, Fr &val_o) {
    Fr ra_a = fe_load<FrParams>(ra + 4 * (size_t)s.a), val_a = fe_load<FrParams>(val + 4 * (size_t)s.a);
    const uint32_t kind = s.kind & RWC_KIND_MASK;
    if (kind == RWC_PAIR) {
        ra_e = ra_a; val_e = val_a;
        ra_o = fe_load<FrParams>(ra + 4 * (size_t)s.b);
        val_o = fe_load<FrParams>(val + 4 * (size_t)s.b);
    } else if (kind == RWC_EVEN_ALONE) {
        ra_e = ra_a; val_e = val_a;
        ra_o = Fr::zero();
        val_o = fr_from_u64_dev(s.imp);
    } else {
        ra_e = Fr::zero();
        val_e = fr_from_u64_dev(s.imp);
        ra_o = ra_a; val_o = val_a;
    }
}

// computePhase1Polynomial (:410-536): (q_constant, q_quadratic) = sum over the steps of E(pair) ra (val + gamma (inc + val)) at t = 0 and
// at infinity (the slopes)
__global__ void __launch_bounds__(256) rwc_cycle_round_kernel(const RwcStep *plan, const uint32_t *n_steps, const uint64_t *ra, const uint64_t *val, const uint64_t *inc,
                                                              uint32_t live, const uint64_t *e_out, uint32_t n_out, const uint64_t *e_in, uint32_t n_in,
                                                              uint32_t in_bits, FrArg gamma_a, uint64_t *partials) {
    __shared__ uint4 sh[256 * 4];
    Fr acc[2] = {Fr::zero(), Fr::zero()};
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k < *n_steps) {  // the grid covers the entries; the plan kernel's step count is not on the host yet
        const RwcStep s = plan[k];
        Fr ra_0, ra_1, val_0, val_1;
        rwc_cycle_members(s, ra, val, ra_0, ra_1, val_0, val_1);
        const Fr ra_inf = fe_sub(ra_1, ra_0), val_inf = fe_sub(val_1, val_0);
        const uint32_t pair = s.key;
        Fr inc_0 = 2 * pair < live ? fe_load<FrParams>(inc + 8 * (size_t)pair) : Fr::zero();
        Fr inc_1 = 2 * pair + 1 < live ? fe_load<FrParams>(inc + 8 * (size_t)pair + 4) : Fr::zero();
        const Fr inc_inf = fe_sub(inc_1, inc_0), gamma = rwc_arg(gamma_a);
        const uint32_t x_out = pair >> in_bits, x_in = pair & ((1u << in_bits) - 1u);
        Fr eo = x_out < n_out ? fe_load<FrParams>(e_out + 4 * (size_t)x_out) : Fr::one();
        Fr ei = x_in < n_in ? fe_load<FrParams>(e_in + 4 * (size_t)x_in) : Fr::one();
        const Fr ep = fr_mul29v(eo, ei);
        if (!ra_0.is_zero()) acc[0] = fr_mul29v(fr_mul29v(ep, ra_0), fe_add(val_0, fr_mul29v(gamma, fe_add(inc_0, val_0))));
        if (!ra_inf.is_zero()) acc[1] = fr_mul29v(fr_mul29v(ep, ra_inf), fe_add(val_inf, fr_mul29v(gamma, fe_add(inc_inf, val_inf))));
    }
    rwc_block_out(acc, sh, partials);
}

// bindEntries (:1139-1185, CycleMajorEntry.bindEntries :110-156): step k -> entry k
__global__ void __launch_bounds__(256) rwc_cycle_bind_kernel(const RwcStep *plan, uint32_t n_steps, const uint64_t *ra, const uint64_t *val, FrArg r_a,
                                                             uint64_t *ra_out, uint64_t *val_out) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_steps) return;
    const FrMul rm = frmul_prepare(rwc_arg(r_a));
    Fr ra_0, ra_1, val_0, val_1;
    rwc_cycle_members(plan[k], ra, val, ra_0, ra_1, val_0, val_1);
    fe_store(ra_out + 4 * (size_t)k, fe_add(ra_0, frmul_apply(fe_sub(ra_1, ra_0), rm)));
    fe_store(val_out + 4 * (size_t)k, fe_add(val_0, frmul_apply(fe_sub(val_1, val_0), rm)));
}

ZG_DEV Fr rwc_fold1(const Fr &lo, const Fr &hi, const FrMul &rm) {
    Fr d = fe_sub(hi, lo);
    if (d.is_zero()) return lo;
    return fe_add(lo, frmul_apply(d, rm));
}
// LowToHigh fold of one or two tables (eq_evals and inc, :919-938; val_init, :953-959)
__global__ void __launch_bounds__(256) rwc_fold_kernel(const uint64_t *a, uint64_t *a_out, const uint64_t *b, uint64_t *b_out, size_t half, FrArg r_a) {
    const FrMul rm = frmul_prepare(rwc_arg(r_a));
    size_t step = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < half; i += step) {
        fe_store(a_out + 4 * i, rwc_fold1(fe_load<FrParams>(a + 8 * i), fe_load<FrParams>(a + 8 * i + 4), rm));
        if (b) fe_store(b_out + 4 * i, rwc_fold1(fe_load<FrParams>(b + 8 * i), fe_load<FrParams>(b + 8 * i + 4), rm));
    }
}

// the two coefficient columns by a permutation (the address-major order at the phase switch)
__global__ void __launch_bounds__(256) rwc_gather_kernel(const uint64_t *ra, const uint64_t *val, const uint32_t *perm, uint32_t n, uint64_t *ra_out,
                                                         uint64_t *val_out) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t s = perm[i];
    fe_store(ra_out + 4 * (size_t)i, fe_load<FrParams>(ra + 4 * (size_t)s));
    fe_store(val_out + 4 * (size_t)i, fe_load<FrParams>(val + 4 * (size_t)s));
}

struct RwcChal {  // the address challenges bound so far
    uint32_t r[24][8];
    int n;
};
ZG_DEV Fr rwc_eq_addr(const RwcChal &ch, uint32_t address) {  // eqAddr (:571-584): bit i of the address <-> challenge i
    Fr acc = Fr::one();
    bool used = false;
    for (int j = 0; j < ch.n; j++) {
        Fr rj;
#pragma unroll
        for (int w = 0; w < 8; w++) rj.l[w] = ch.r[j][w];
        Fr f = ((address >> j) & 1u) ? rj : fe_sub(Fr::one(), rj);
        acc = used ? fr_mul29v(acc, f) : f;
        used = true;
    }
    return acc;
}
// the members of an address-phase step: a lone entry meets the other column's checkpoint — the value after the last entry consumed there,
// or, before any, val_init's column as the reference's in-place array holds it (below `split`: the current table; from there to `size`:
// the previous level)
ZG_DEV void rwc_address_members(const RwcStep &s, const uint64_t *ra, const uint64_t *val, const uint64_t *val_cur, const uint64_t *val_prev, uint32_t split,
                                uint32_t size, Fr &ra_e, Fr &ra_o, Fr &val_e, Fr &val_o) {
    Fr ra_a = fe_load<FrParams>(ra + 4 * (size_t)s.a), val_a = fe_load<FrParams>(val + 4 * (size_t)s.a);
    const uint32_t kind = s.kind & RWC_KIND_MASK;
    if (kind == RWC_PAIR) {
        ra_e = ra_a; val_e = val_a;
        ra_o = fe_load<FrParams>(ra + 4 * (size_t)s.b);
        val_o = fe_load<FrParams>(val + 4 * (size_t)s.b);
        return;
    }
    Fr other;
    if (s.kind & RWC_IMP_IS_COLUMN) {
        const uint32_t c = (uint32_t)s.imp;
        other = c >= size ? Fr::zero() : (c < split ? fe_load<FrParams>(val_cur + 4 * (size_t)c) : fe_load<FrParams>(val_prev + 4 * (size_t)c));
    } else {
        other = fr_from_u64_dev(s.imp);
    }
    if (kind == RWC_EVEN_ALONE) {
        ra_e = ra_a; val_e = val_a;
        ra_o = Fr::zero(); val_o = other;
    } else {
        ra_e = Fr::zero(); val_e = other;
        ra_o = ra_a; val_o = val_a;
    }
}

// computePhase2Polynomial (:538-769): s(0), s(2) = sum eq_cycle eqAddr(address) ra_t (val_t (1 + gamma) + gamma inc), t = 0, 2
__global__ void __launch_bounds__(256) rwc_address_round_kernel(const RwcStep *plan, uint32_t n_steps, const uint64_t *ra, const uint64_t *val, RwcChal ch,
                                                                const uint64_t *val_cur, uint32_t size, FrArg eq_cycle_a, FrArg gamma_a, FrArg inc_a,
                                                                uint64_t *partials) {
    __shared__ uint4 sh[256 * 4];
    Fr acc[2] = {Fr::zero(), Fr::zero()};
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k < n_steps) {
        const RwcStep s = plan[k];
        Fr ra_0, ra_1, val_0, val_1;
        rwc_address_members(s, ra, val, val_cur, val_cur, size, size, ra_0, ra_1, val_0, val_1);
        const Fr ra_2 = fe_sub(fe_add(ra_1, ra_1), ra_0), val_2 = fe_sub(fe_add(val_1, val_1), val_0);
        const Fr gamma = rwc_arg(gamma_a), opg = fe_add(Fr::one(), gamma), ginc = fr_mul29v(gamma, rwc_arg(inc_a));
        const Fr eqp = fr_mul29v(rwc_arg(eq_cycle_a), rwc_eq_addr(ch, s.key));
        if (!ra_0.is_zero()) acc[0] = fr_mul29v(fr_mul29v(eqp, ra_0), fe_add(fr_mul29v(val_0, opg), ginc));
        if (!ra_2.is_zero()) acc[1] = fr_mul29v(fr_mul29v(eqp, ra_2), fe_add(fr_mul29v(val_2, opg), ginc));
    }
    rwc_block_out(acc, sh, partials);
}

// bindEntriesAddressMajor (:973-1137): step k -> entry k; the checkpoints are read from val_init AFTER this round's fold
__global__ void __launch_bounds__(256) rwc_address_bind_kernel(const RwcStep *plan, uint32_t n_steps, const uint64_t *ra, const uint64_t *val,
                                                               const uint64_t *val_new, const uint64_t *val_old, uint32_t size, FrArg r_a, uint64_t *ra_out,
                                                               uint64_t *val_out) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_steps) return;
    const FrMul rm = frmul_prepare(rwc_arg(r_a));
    Fr ra_0, ra_1, val_0, val_1;
    rwc_address_members(plan[k], ra, val, val_new, val_old, size / 2, size, ra_0, ra_1, val_0, val_1);
    fe_store(ra_out + 4 * (size_t)k, fe_add(ra_0, frmul_apply(fe_sub(ra_1, ra_0), rm)));
    fe_store(val_out + 4 * (size_t)k, fe_add(val_0, frmul_apply(fe_sub(val_1, val_0), rm)));
}

// getOpeningClaims (:1210-1322): sum_i eq(r_address, address_i) eq(r_cycle, cycle_i) * {ra_i, val_i - v0}
struct RwcPoint {
    uint32_t r[48][8];
    int n_addr, n_cyc;  // r[0 .. n_addr): r_address (r[0] <-> MSB), then r_cycle likewise
};
__global__ void __launch_bounds__(256) rwc_opening_kernel(const uint32_t *cycle, const uint32_t *addr, const uint64_t *ra, const uint64_t *val, uint32_t n,
                                                          RwcPoint pt, FrArg v0_a, uint64_t *partials) {
    __shared__ uint4 sh[256 * 4];
    Fr acc[2] = {Fr::zero(), Fr::zero()};
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const uint32_t a = addr[i], c = cycle[i];
        Fr w = Fr::one();
        for (int j = 0; j < pt.n_addr + pt.n_cyc; j++) {
            Fr rj;
#pragma unroll
            for (int k = 0; k < 8; k++) rj.l[k] = pt.r[j][k];
            const bool bit = j < pt.n_addr ? (a >> (pt.n_addr - 1 - j)) & 1u : (c >> (pt.n_cyc - 1 - (j - pt.n_addr))) & 1u;
            w = fr_mul29v(w, bit ? rj : fe_sub(Fr::one(), rj));
        }
        acc[0] = fr_mul29v(w, fe_load<FrParams>(ra + 4 * (size_t)i));
        acc[1] = fr_mul29v(w, fe_sub(fe_load<FrParams>(val + 4 * (size_t)i), rwc_arg(v0_a)));
    }
    rwc_block_out(acc, sh, partials);
}

}  // namespace zg

struct zg_rwc_s {
    int device = -1;
    size_t log_k = 0, log_t = 0;
    // the integer skeleton of the entry list (host) and the plan of the current round
    std::vector<uint32_t> cycle, addr;
    std::vector<uint64_t> prev, next;
    // the plan is written into PINNED memory (two buffers in turn: the upload of one round's plan is asynchronous, the next round's walk
    // must not overwrite it before the copy engine has read it) — a pageable 5 MB plan per round was staged synchronously
    struct Plan {
        zg::RwcStep *p = nullptr;
        size_t n = 0;
        void clear() { n = 0; }
        void reserve(size_t) {}
        void push_back(const zg::RwcStep &s) { p[n++] = s; }
        size_t size() const { return n; }
        bool empty() const { return n == 0; }
        const zg::RwcStep &operator[](size_t i) const { return p[i]; }
        const zg::RwcStep *data() const { return p; }
    } plan;
    zg::RwcStep *h_plan[2] = {nullptr, nullptr};
    hipEvent_t plan_uploaded[2] = {nullptr, nullptr};
    int plan_buf = 0;
    std::vector<uint32_t> c2, a2;  // rwc_apply_plan's output buffers, kept between rounds
    std::vector<uint64_t> p2, n2;
    bool plan_valid = false, plan_is_address = false;
    bool plan_m_known = false;  // a cycle-phase plan's step count has been read back (plan_m means nothing before that)
    size_t plan_addr_round = 0;
    size_t plan_m = 0;  // steps of the current plan (a cycle-phase plan is written on the device: its length arrives with the round's sums)
    // the skeleton on the device, authoritative in the cycle phases: cycle | address (u32) and prev_val | next_val (u64), cap words each, two
    // sets in turn (the walk writes the bound skeleton into the other set); host_skel / dev_skel say which side holds the current list
    uint32_t *d_sk32[2] = {nullptr, nullptr};
    uint64_t *d_sk64[2] = {nullptr, nullptr};
    int sk = 0;
    size_t n_entries = 0;
    bool dev_skel = false, host_skel = true;
    uint32_t *d_blk = nullptr;  // the walk's per-block step counts, then its step count
    // device: coefficient columns (double-buffered), the plan, the dense tables
    uint32_t cap = 0;
    uint64_t *ra[2] = {nullptr, nullptr}, *val_c[2] = {nullptr, nullptr};
    int cur = 0;
    zg::RwcStep *d_plan = nullptr;
    uint32_t *d_idx = nullptr;  // permutation / skeleton columns for the kernels that need them
    uint64_t *eq[2] = {nullptr, nullptr}, *inc[2] = {nullptr, nullptr}, *val[2] = {nullptr, nullptr};
    int vcur = 0, kcur = 0;  // live buffer of the cycle tables / of val_init
    size_t eq_size = 0, k_size = 0;
    bool address_major = false;
    uint64_t *d_part = nullptr, *d_out = nullptr, *h_out = nullptr;
    uint64_t eq_cycle[4] = {}, inc_scalar[4] = {};  // eq_evals[0], inc[0] at the phase switch (:543-552)
    hipStream_t st = nullptr;
    std::mutex mu;
};

using namespace zg;

// pinned host buffers (the address phase's plans, the result words) are pooled per process: hipHostMalloc of a 6 MB plan buffer costs about
// as much as a whole cycle phase. A buffer is reused for requests of at least a quarter of its size; rwc_shutdown (zg_shutdown) frees the pool.
static std::mutex g_rwc_pin_mu;
struct RwcPin { void *p; size_t bytes; };
static std::vector<RwcPin> g_rwc_pins;
static void *rwc_pin_get(size_t bytes) {
    {
        std::lock_guard<std::mutex> lk(g_rwc_pin_mu);
        for (size_t i = 0; i < g_rwc_pins.size(); i++)
            if (g_rwc_pins[i].bytes >= bytes && g_rwc_pins[i].bytes <= 4 * bytes + 4096) {
                void *p = g_rwc_pins[i].p;
                g_rwc_pins.erase(g_rwc_pins.begin() + i);
                return p;
            }
    }
    void *p = nullptr;
    return hipHostMalloc(&p, bytes ? bytes : 16) == hipSuccess ? p : nullptr;
}
static void rwc_pin_put(void *p, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_rwc_pin_mu);
    if (g_rwc_pins.size() >= 16) {
        (void)hipHostFree(p);
        return;
    }
    g_rwc_pins.push_back(RwcPin{p, bytes ? bytes : 16});
}
namespace zg {
void rwc_shutdown() {
    std::lock_guard<std::mutex> lk(g_rwc_pin_mu);
    for (auto &b : g_rwc_pins) (void)hipHostFree(b.p);
    g_rwc_pins.clear();
}
}  // namespace zg

static void rwc_free(zg_rwc_s *s) {
    if (!s) return;
    for (int b = 0; b < 2; b++)
        for (void *p : {(void *)s->ra[b], (void *)s->val_c[b], (void *)s->eq[b], (void *)s->inc[b], (void *)s->val[b]})
            if (p) scratch_put(p);
    for (void *p : {(void *)s->d_plan, (void *)s->d_idx, (void *)s->d_part, (void *)s->d_out, (void *)s->d_blk, (void *)s->d_sk32[0], (void *)s->d_sk32[1],
                    (void *)s->d_sk64[0], (void *)s->d_sk64[1]})
        if (p) scratch_put(p);
    if (s->h_out) rwc_pin_put(s->h_out, 16 * 32);
    for (int b = 0; b < 2; b++) {
        if (s->h_plan[b]) rwc_pin_put(s->h_plan[b], (size_t)s->cap * sizeof(RwcStep));
        if (s->plan_uploaded[b]) (void)hipEventDestroy(s->plan_uploaded[b]);
    }
    if (s->st) stream_release(s->st, s->device);
    delete s;
}
static FrArg rwc_fr_arg(const uint64_t r[4]) {
    FrArg a;
    for (int i = 0; i < 4; i++) {
        a.l[2 * i] = (uint32_t)r[i];
        a.l[2 * i + 1] = (uint32_t)(r[i] >> 32);
    }
    return a;
}
// the two sums of a round; with d_m, the step count of the device walk comes back under the same synchronisation
static int rwc_collect(zg_rwc_s *s, uint32_t nblocks, uint64_t *a, uint64_t *b, const uint32_t *d_m = nullptr) {
    hipLaunchKernelGGL(rwc_finish_kernel, dim3(1), dim3(256), 0, s->st, s->d_part, nblocks, s->d_out);
    ZG_HIP(hipGetLastError());
    ZG_HIP(hipMemcpyAsync(s->h_out, s->d_out, 64, hipMemcpyDeviceToHost, s->st));
    if (d_m) ZG_HIP(hipMemcpyAsync(s->h_out + 8, d_m, 4, hipMemcpyDeviceToHost, s->st));
    ZG_HIP(hipStreamSynchronize(s->st));
    if (d_m) {
        s->plan_m = (uint32_t)s->h_out[8];
        s->plan_m_known = true;
    }
    for (int i = 0; i < 4; i++) {
        a[i] = s->h_out[i];
        b[i] = s->h_out[4 + i];
    }
    return ZG_OK;
}

// the other pinned plan buffer, once its last upload has left the host
static int rwc_next_plan_buffer(zg_rwc_s *s) {
    s->plan_buf ^= 1;
    if (!s->h_plan[s->plan_buf]) {
        s->h_plan[s->plan_buf] = reinterpret_cast<RwcStep *>(rwc_pin_get((size_t)s->cap * sizeof(RwcStep)));
        if (!s->h_plan[s->plan_buf]) {
            set_error("zg_rwc: pinned plan buffer");
            return ZG_ERR_NOMEM;
        }
    }
    (void)hipEventSynchronize(s->plan_uploaded[s->plan_buf]);
    s->plan.p = s->h_plan[s->plan_buf];
    s->plan.n = 0;
    return ZG_OK;
}
// ---- where the skeleton lives. The cycle phases keep it on the device (rwc_walk_*), the address phase on the host.
static int rwc_host_skeleton(zg_rwc_s *s) {
    if (s->host_skel) return ZG_OK;
    const size_t n = s->n_entries, cap = s->cap;
    s->cycle.resize(n); s->addr.resize(n); s->prev.resize(n); s->next.resize(n);
    if (n) {
        ZG_HIP(hipMemcpyAsync(s->cycle.data(), s->d_sk32[s->sk], n * 4, hipMemcpyDeviceToHost, s->st));
        ZG_HIP(hipMemcpyAsync(s->addr.data(), s->d_sk32[s->sk] + cap, n * 4, hipMemcpyDeviceToHost, s->st));
        ZG_HIP(hipMemcpyAsync(s->prev.data(), s->d_sk64[s->sk], n * 8, hipMemcpyDeviceToHost, s->st));
        ZG_HIP(hipMemcpyAsync(s->next.data(), s->d_sk64[s->sk] + cap, n * 8, hipMemcpyDeviceToHost, s->st));
        ZG_HIP(hipStreamSynchronize(s->st));
    }
    s->host_skel = true;
    return ZG_OK;
}
static int rwc_dev_skeleton(zg_rwc_s *s) {
    if (s->dev_skel) return ZG_OK;
    const size_t n = s->n_entries, cap = s->cap;  // host_skel holds: one side always does
    if (n) {
        ZG_HIP(hipMemcpyAsync(s->d_sk32[s->sk], s->cycle.data(), n * 4, hipMemcpyHostToDevice, s->st));
        ZG_HIP(hipMemcpyAsync(s->d_sk32[s->sk] + cap, s->addr.data(), n * 4, hipMemcpyHostToDevice, s->st));
        ZG_HIP(hipMemcpyAsync(s->d_sk64[s->sk], s->prev.data(), n * 8, hipMemcpyHostToDevice, s->st));
        ZG_HIP(hipMemcpyAsync(s->d_sk64[s->sk] + cap, s->next.data(), n * 8, hipMemcpyHostToDevice, s->st));
        ZG_HIP(hipStreamSynchronize(s->st));  // pageable sources
    }
    s->dev_skel = true;
    return ZG_OK;
}
// ---- the walks
// cycle phases, on the device (rwc_walk_*): the plan into d_plan, the bound skeleton into the other set, the step count behind the block counts
static uint32_t *rwc_walk_total(zg_rwc_s *s) { return s->d_blk + div_up(s->n_entries, RWC_WALK_BLOCK); }
static int rwc_plan_cycle(zg_rwc_s *s) {
    ZG_TRY(rwc_dev_skeleton(s));
    const uint32_t n = (uint32_t)s->n_entries, nblk = (uint32_t)div_up(s->n_entries, RWC_WALK_BLOCK), cap = s->cap;
    if (n) {
        const uint32_t *cyc = s->d_sk32[s->sk], *adr = cyc + cap;
        const uint64_t *prv = s->d_sk64[s->sk], *nxt = prv + cap;
        hipLaunchKernelGGL(rwc_walk_count_kernel, dim3(nblk), dim3(RWC_WALK_BLOCK), 0, s->st, cyc, adr, n, s->d_blk);
        hipLaunchKernelGGL(rwc_walk_scan_kernel, dim3(1), dim3(1024), 0, s->st, s->d_blk, nblk);
        hipLaunchKernelGGL(rwc_walk_plan_kernel, dim3(nblk), dim3(RWC_WALK_BLOCK), 0, s->st, cyc, adr, prv, nxt, n, s->d_blk, s->d_plan, s->d_sk32[s->sk ^ 1],
                           s->d_sk32[s->sk ^ 1] + cap, s->d_sk64[s->sk ^ 1], s->d_sk64[s->sk ^ 1] + cap);
        ZG_HIP(hipGetLastError());
    }
    s->plan_m = 0;  // known once rwc_collect / zg_rwc_bind_cycle has read it
    s->plan_m_known = s->n_entries == 0;
    s->plan_valid = true;
    s->plan_is_address = false;
    return ZG_OK;
}
// address phase: column pairs by (address >> addr_round) / 2 and the two-pointer walk with carried checkpoints (:585-700, 1000-1075)
static int rwc_plan_address(zg_rwc_s *s, size_t addr_round) {
    const size_t n = s->cycle.size();
    ZG_TRY(rwc_next_plan_buffer(s));
    const uint32_t sh = (uint32_t)addr_round;
    s->plan.clear();
    s->plan.reserve(n);
    for (size_t i = 0; i < n;) {
        const uint32_t cp = (s->addr[i] >> sh) >> 1;
        size_t j = i;
        while (j < n && ((s->addr[j] >> sh) >> 1) == cp) j++;
        size_t m = i;
        while (m < j && !((s->addr[m] >> sh) & 1u)) m++;
        // checkpoints: a val_init column until an entry of that column has been consumed, then that entry's next_val
        bool ec_col = true, oc_col = true;
        uint64_t ec = 2 * (uint64_t)cp, oc = 2 * (uint64_t)cp + 1;
        size_t a = i, o = m;
        auto even_alone = [&](size_t e) {
            s->plan.push_back(RwcStep{(uint32_t)e, ~0u, s->addr[e], RWC_EVEN_ALONE | (oc_col ? (uint32_t)RWC_IMP_IS_COLUMN : 0u), oc});
            ec = s->next[e];
            ec_col = false;
        };
        auto odd_alone = [&](size_t e) {
            s->plan.push_back(RwcStep{(uint32_t)e, ~0u, s->addr[e], RWC_ODD_ALONE | (ec_col ? (uint32_t)RWC_IMP_IS_COLUMN : 0u), ec});
            oc = s->next[e];
            oc_col = false;
        };
        while (a < m && o < j) {
            if (s->cycle[a] == s->cycle[o]) {
                s->plan.push_back(RwcStep{(uint32_t)a, (uint32_t)o, s->addr[a], RWC_PAIR, 0});
                ec = s->next[a];
                oc = s->next[o];
                ec_col = oc_col = false;
                a++;
                o++;
            } else if (s->cycle[a] < s->cycle[o]) {
                even_alone(a++);
            } else {
                odd_alone(o++);
            }
        }
        while (a < m) even_alone(a++);
        while (o < j) odd_alone(o++);
        i = j;
    }
    s->plan_m = s->plan.size();
    s->plan_m_known = true;
    s->plan_valid = true;
    s->plan_is_address = true;
    s->plan_addr_round = addr_round;
    return ZG_OK;
}
// the skeleton of the bound list, address phase: step k -> entry k (bindAddressMajor* :1077-1137)
static void rwc_apply_plan(zg_rwc_s *s) {
    const size_t m = s->plan.size();
    std::vector<uint32_t> &c2 = s->c2, &a2 = s->a2;
    std::vector<uint64_t> &p2 = s->p2, &n2 = s->n2;
    c2.resize(m); a2.resize(m); p2.resize(m); n2.resize(m);
    for (size_t k = 0; k < m; k++) {
        const RwcStep &st = s->plan[k];
        c2[k] = s->cycle[st.a];
        a2[k] = s->addr[st.a] >> 1;  // the bound address variable leaves the column index
        p2[k] = s->prev[st.a];
        n2[k] = (st.kind & RWC_KIND_MASK) == RWC_PAIR ? s->next[st.b] : s->next[st.a];
    }
    s->cycle.swap(c2);
    s->addr.swap(a2);
    s->prev.swap(p2);
    s->next.swap(n2);
    s->n_entries = m;
    s->dev_skel = false;
    s->plan_valid = false;
}
static int rwc_upload_plan(zg_rwc_s *s) {
    if (!s->plan.empty()) ZG_HIP(hipMemcpyAsync(s->d_plan, s->plan.data(), s->plan.size() * sizeof(RwcStep), hipMemcpyHostToDevice, s->st));
    ZG_HIP(hipEventRecord(s->plan_uploaded[s->plan_buf], s->st));
    return ZG_OK;
}

extern "C" {

// inc from the host (zg_rwc_open) or, with is_write, scattered on the device from the entries (zg_rwc_open_writes)
static int rwc_open_impl(size_t log_k, size_t log_t, size_t n, const uint32_t *cycle, const uint32_t *address, const uint64_t *val_coeff, const uint64_t *prev_val,
                         const uint64_t *next_val, const uint64_t *inc, const uint8_t *is_write, const uint64_t *val_init, const uint64_t *r_cycle, zg_rwc_t *out) {
    ZG_INIT();
    if (!out || log_k > 24 || log_t > 26 || n > ((size_t)1 << 24) || (!inc && !is_write && n) || !val_init || (log_t && !r_cycle) ||
        (n && (!cycle || !address || !val_coeff || !prev_val || !next_val))) {
        set_error("zg_rwc_open: invalid argument (log_k <= 24, log_t <= 26, at most 2^24 entries)");
        return ZG_ERR_INVALID;
    }
    const size_t T = (size_t)1 << log_t, K = (size_t)1 << log_k;
    for (size_t i = 0; i < n; i++)
        if (cycle[i] >= T || address[i] >= K || (i && (cycle[i] < cycle[i - 1] || (cycle[i] == cycle[i - 1] && address[i] < address[i - 1])))) {
            set_error("zg_rwc_open: entries must lie inside the tables and be sorted by (cycle, address)");
            return ZG_ERR_INVALID;
        }
    if (!inc) {  // the reference keeps the LAST write of a cycle in access order, which the sorted list no longer knows
        uint32_t writes_in_cycle = 0;
        for (size_t i = 0; i < n; i++) {
            writes_in_cycle = (i && cycle[i] == cycle[i - 1] ? writes_in_cycle : 0) + (is_write[i] ? 1 : 0);
            if (writes_in_cycle > 1) {
                set_error("zg_rwc_open_writes: two writes in one cycle (pass inc with zg_rwc_open)");
                return ZG_ERR_INVALID;
            }
        }
    }
    zg_rwc_s *s = new zg_rwc_s();
    s->device = current_device();
    s->log_k = log_k;
    s->log_t = log_t;
    s->cap = (uint32_t)(n ? n : 1);
    s->eq_size = T;
    s->k_size = K;
    s->cycle.assign(cycle, cycle + n);
    s->addr.assign(address, address + n);
    s->prev.assign(prev_val, prev_val + n);
    s->next.assign(next_val, next_val + n);
    s->st = stream_acquire();
    hipError_t e = s->st ? hipSuccess : hipErrorOutOfMemory;
    // session buffers come from the library's scratch cache and a pinned pool: nineteen hipMalloc + three hipHostMalloc calls per prover
    // were most of a 6 ms set-up. The pinned plan buffers of the address phase are taken when that phase starts (rwc_next_plan_buffer).
    auto dev = [&](auto **pp, size_t bytes) {
        if (e != hipSuccess) return;
        *pp = reinterpret_cast<std::remove_reference_t<decltype(*pp)>>(scratch_get(bytes));
        if (!*pp) e = hipErrorOutOfMemory;
    };
    for (int b = 0; b < 2; b++) {
        dev(&s->ra[b], (size_t)s->cap * 32);
        dev(&s->val_c[b], (size_t)s->cap * 32);
        dev(&s->eq[b], (b ? (T / 2 ? T / 2 : 1) : T) * 32);
        dev(&s->inc[b], (b ? (T / 2 ? T / 2 : 1) : T) * 32);
        dev(&s->val[b], K * 32);  // both full size: the previous level stays readable
        dev(&s->d_sk32[b], (size_t)s->cap * 2 * 4);
        dev(&s->d_sk64[b], (size_t)s->cap * 2 * 8);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s->plan_uploaded[b], hipEventDisableTiming);
    }
    dev(&s->d_plan, (size_t)s->cap * sizeof(RwcStep));
    dev(&s->d_idx, (size_t)s->cap * 4);
    dev(&s->d_blk, (div_up(s->cap, RWC_WALK_BLOCK) + 2) * 4);
    s->n_entries = n;
    dev(&s->d_part, (size_t)RWC_MAX_BLOCKS * 2 * 32);
    dev(&s->d_out, 8 * 32);
    if (e == hipSuccess && !(s->h_out = reinterpret_cast<uint64_t *>(rwc_pin_get(16 * 32)))) e = hipErrorOutOfMemory;
    if (e != hipSuccess) {
        set_error(std::string("zg_rwc_open: ") + hipGetErrorString(e));
        rwc_free(s);
        return e == hipErrorOutOfMemory ? ZG_ERR_NOMEM : ZG_ERR_HIP;
    }
    Scratch s_val((size_t)s->cap * 8), s_wr(inc ? 8 : n + 8);  // val_coeff as u64; is_write of the entries, for the scatter
    if (!s_val.p || !s_wr.p) {
        rwc_free(s);
        return ZG_ERR_NOMEM;
    }
    int rc = [&]() -> int {
        SyncGuard sync(s->st);
        if (n) {  // the skeleton, set 0
            ZG_HIP(hipMemcpyAsync(s->d_sk32[0], cycle, n * 4, hipMemcpyHostToDevice, s->st));
            ZG_HIP(hipMemcpyAsync(s->d_sk32[0] + s->cap, address, n * 4, hipMemcpyHostToDevice, s->st));
            ZG_HIP(hipMemcpyAsync(s->d_sk64[0], prev_val, n * 8, hipMemcpyHostToDevice, s->st));
            ZG_HIP(hipMemcpyAsync(s->d_sk64[0] + s->cap, next_val, n * 8, hipMemcpyHostToDevice, s->st));
            ZG_HIP(hipMemcpyAsync(s_val.p, val_coeff, n * 8, hipMemcpyHostToDevice, s->st));
            hipLaunchKernelGGL(rwc_init_kernel, dim3(div_up(n, 256)), dim3(256), 0, s->st, s_val.as<uint64_t>(), (uint32_t)n, s->ra[0], s->val_c[0]);
            ZG_HIP(hipGetLastError());
        }
        if (inc) {
            ZG_HIP(hipMemcpyAsync(s->inc[0], inc, T * 32, hipMemcpyHostToDevice, s->st));
        } else {
            ZG_HIP(hipMemsetAsync(s->inc[0], 0, T * 32, s->st));
            if (n) {
                uint8_t *d_w = s_wr.as<uint8_t>();
                ZG_HIP(hipMemcpyAsync(d_w, is_write, n, hipMemcpyHostToDevice, s->st));
                hipLaunchKernelGGL(rwc_inc_scatter_kernel, dim3(div_up(n, 256)), dim3(256), 0, s->st, s->d_sk32[0], s->d_sk64[0], s->d_sk64[0] + s->cap, d_w,
                                   (uint32_t)n, s->inc[0]);
                ZG_HIP(hipGetLastError());
            }
        }
        ZG_HIP(hipMemcpyAsync(s->val[0], val_init, K * 32, hipMemcpyHostToDevice, s->st));
        ZG_TRY(zg_fr_eq_table_dev(r_cycle, log_t, nullptr, s->eq[0], s->st));  // computeEqBigEndian (:345-348)
        ZG_HIP(hipStreamSynchronize(s->st));
        sync.dismiss();
        s->dev_skel = true;
        return ZG_OK;
    }();
    if (rc != ZG_OK) {
        std::string keep = zg_last_error();
        (void)hipStreamSynchronize(s->st);
        rwc_free(s);
        set_error(keep);
        return rc;
    }
    *out = s;
    return ZG_OK;
}
int zg_rwc_open(size_t log_k, size_t log_t, size_t n, const uint32_t *cycle, const uint32_t *address, const uint64_t *val_coeff, const uint64_t *prev_val,
                const uint64_t *next_val, const uint64_t *inc, const uint64_t *val_init, const uint64_t *r_cycle, zg_rwc_t *out) {
    if (!inc) {
        set_error("zg_rwc_open: invalid argument (inc)");
        return ZG_ERR_INVALID;
    }
    return rwc_open_impl(log_k, log_t, n, cycle, address, val_coeff, prev_val, next_val, inc, nullptr, val_init, r_cycle, out);
}
int zg_rwc_open_writes(size_t log_k, size_t log_t, size_t n, const uint32_t *cycle, const uint32_t *address, const uint64_t *val_coeff, const uint64_t *prev_val,
                       const uint64_t *next_val, const uint8_t *is_write, const uint64_t *val_init, const uint64_t *r_cycle, zg_rwc_t *out) {
    if (n && !is_write) {
        set_error("zg_rwc_open_writes: invalid argument (is_write)");
        return ZG_ERR_INVALID;
    }
    return rwc_open_impl(log_k, log_t, n, cycle, address, val_coeff, prev_val, next_val, nullptr, is_write, val_init, r_cycle, out);
}

size_t zg_rwc_entries(zg_rwc_t s) { return s ? s->n_entries : 0; }
size_t zg_rwc_cycles(zg_rwc_t s) { return s ? s->eq_size : 0; }

int zg_rwc_round_cycle(zg_rwc_t s, const uint64_t *d_e_out, size_t n_out, const uint64_t *d_e_in, size_t n_in, const uint64_t gamma[4], uint64_t q_constant[4],
                       uint64_t q_quadratic[4]) {
    ZG_INIT();
    if (!s || !q_constant || !q_quadratic || !gamma || !d_e_out || !d_e_in || n_in == 0 || (n_in & (n_in - 1))) {
        set_error("zg_rwc_round_cycle: invalid argument (|E_in| a power of two)");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    ZG_TRY(rwc_plan_cycle(s));
    if (s->n_entries == 0) {
        for (int i = 0; i < 4; i++) q_constant[i] = q_quadratic[i] = 0;
        return ZG_OK;
    }
    uint32_t in_bits = 0;
    while (((size_t)1 << in_bits) < n_in) in_bits++;
    const uint32_t nb = div_up(s->n_entries, 256);  // one thread per ENTRY: the steps are at most as many, their count is still on the device
    hipLaunchKernelGGL(rwc_cycle_round_kernel, dim3(nb), dim3(256), 0, s->st, s->d_plan, rwc_walk_total(s), s->ra[s->cur], s->val_c[s->cur], s->inc[s->vcur],
                       (uint32_t)s->eq_size, d_e_out, (uint32_t)n_out, d_e_in, (uint32_t)n_in, in_bits, rwc_fr_arg(gamma), s->d_part);
    ZG_HIP(hipGetLastError());
    return rwc_collect(s, nb, q_constant, q_quadratic, rwc_walk_total(s));
}

int zg_rwc_bind_cycle(zg_rwc_t s, const uint64_t r[4]) {
    ZG_INIT();
    if (!s || !r || s->eq_size < 2) {
        set_error("zg_rwc_bind_cycle: no cycle variable left");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    const size_t half = s->eq_size / 2;
    const int vn = s->vcur ^ 1;
    unsigned nf = div_up(half, 256);
    if (nf > 4096) nf = 4096;
    hipLaunchKernelGGL(rwc_fold_kernel, dim3(nf), dim3(256), 0, s->st, s->eq[s->vcur], s->eq[vn], s->inc[s->vcur], s->inc[vn], half, rwc_fr_arg(r));
    ZG_HIP(hipGetLastError());
    s->vcur = vn;
    s->eq_size = half;
    if (!s->plan_valid || s->plan_is_address) ZG_TRY(rwc_plan_cycle(s));  // a bind without the round call before it: walk now
    if (!s->plan_m_known) {  // also after a round call whose collect failed between the walk and the read: never trust plan_m = 0
        ZG_HIP(hipMemcpyAsync(s->h_out + 8, rwc_walk_total(s), 4, hipMemcpyDeviceToHost, s->st));
        ZG_HIP(hipStreamSynchronize(s->st));
        s->plan_m = (uint32_t)s->h_out[8];
        s->plan_m_known = true;
    }
    const uint32_t m = (uint32_t)s->plan_m;
    if (m) {
        hipLaunchKernelGGL(rwc_cycle_bind_kernel, dim3(div_up(m, 256)), dim3(256), 0, s->st, s->d_plan, m, s->ra[s->cur], s->val_c[s->cur], rwc_fr_arg(r),
                           s->ra[s->cur ^ 1], s->val_c[s->cur ^ 1]);
        ZG_HIP(hipGetLastError());
        s->cur ^= 1;
    }
    // the bound skeleton is what the walk wrote into the other set
    s->sk ^= 1;
    s->n_entries = m;
    s->host_skel = false;
    s->plan_valid = false;
    return ZG_OK;
}

// the stable sort by (address, cycle) at the phase switch (:553-558), and eq_evals[0] / inc[0] as they stand then (:543-552)
static int rwc_to_address_major(zg_rwc_s *s) {
    if (s->address_major) return ZG_OK;
    ZG_TRY(rwc_host_skeleton(s));
    ZG_HIP(hipMemcpyAsync(s->h_out, s->eq[s->vcur], 32, hipMemcpyDeviceToHost, s->st));
    ZG_HIP(hipMemcpyAsync(s->h_out + 4, s->inc[s->vcur], 32, hipMemcpyDeviceToHost, s->st));
    ZG_HIP(hipStreamSynchronize(s->st));
    for (int i = 0; i < 4; i++) {
        s->eq_cycle[i] = s->h_out[i];
        s->inc_scalar[i] = s->h_out[4 + i];
    }
    const size_t n = s->cycle.size();
    if (n > 1) {
        std::vector<uint32_t> perm(n);
        // The list the cycle phase leaves is still in cycle order (a bind halves every cycle and keeps the steps in list order), so a STABLE
        // counting sort by address alone is the stable sort by (address, cycle): O(n + addresses) instead of 11 ms of comparisons at 2^18
        // entries. A list that is not in cycle order (entry points called out of the prover's order) takes the comparison sort.
        bool by_cycle = true;
        uint32_t amax = 0;
        for (size_t i = 0; i < n; i++) {
            by_cycle = by_cycle && (i == 0 || s->cycle[i - 1] <= s->cycle[i]);
            amax = s->addr[i] > amax ? s->addr[i] : amax;
        }
        if (by_cycle && (size_t)amax <= 8 * n + 65536) {
            std::vector<uint32_t> first((size_t)amax + 2, 0u);
            for (size_t i = 0; i < n; i++) first[(size_t)s->addr[i] + 1]++;
            for (size_t a = 1; a < first.size(); a++) first[a] += first[a - 1];
            for (size_t i = 0; i < n; i++) perm[first[s->addr[i]]++] = (uint32_t)i;
        } else {
            for (size_t i = 0; i < n; i++) perm[i] = (uint32_t)i;
            std::stable_sort(perm.begin(), perm.end(),
                             [&](uint32_t x, uint32_t y) { return s->addr[x] != s->addr[y] ? s->addr[x] < s->addr[y] : s->cycle[x] < s->cycle[y]; });
        }
        std::vector<uint32_t> c2(n), a2(n);
        std::vector<uint64_t> p2(n), n2(n);
        for (size_t i = 0; i < n; i++) {
            c2[i] = s->cycle[perm[i]];
            a2[i] = s->addr[perm[i]];
            p2[i] = s->prev[perm[i]];
            n2[i] = s->next[perm[i]];
        }
        s->cycle.swap(c2);
        s->addr.swap(a2);
        s->prev.swap(p2);
        s->next.swap(n2);
        ZG_HIP(hipMemcpyAsync(s->d_idx, perm.data(), n * 4, hipMemcpyHostToDevice, s->st));
        hipLaunchKernelGGL(rwc_gather_kernel, dim3(div_up(n, 256)), dim3(256), 0, s->st, s->ra[s->cur], s->val_c[s->cur], s->d_idx, (uint32_t)n, s->ra[s->cur ^ 1],
                           s->val_c[s->cur ^ 1]);
        ZG_HIP(hipGetLastError());
        ZG_HIP(hipStreamSynchronize(s->st));  // perm is a local
        s->cur ^= 1;
        s->dev_skel = false;
    }
    s->address_major = true;
    s->plan_valid = false;
    return ZG_OK;
}

static int rwc_chal(RwcChal &ch, const uint64_t *challenges, size_t addr_round) {
    if (addr_round > 24 || (addr_round && !challenges)) {
        set_error("zg_rwc: at most 24 address rounds");
        return ZG_ERR_INVALID;
    }
    ch.n = (int)addr_round;
    for (size_t j = 0; j < addr_round; j++)
        for (int w = 0; w < 4; w++) {
            ch.r[j][2 * w] = (uint32_t)challenges[4 * j + w];
            ch.r[j][2 * w + 1] = (uint32_t)(challenges[4 * j + w] >> 32);
        }
    return ZG_OK;
}

int zg_rwc_round_address(zg_rwc_t s, size_t addr_round, const uint64_t *challenges, const uint64_t gamma[4], uint64_t s0[4], uint64_t s2[4]) {
    ZG_INIT();
    if (!s || !s0 || !s2 || !gamma || addr_round >= s->log_k) {
        set_error("zg_rwc_round_address: invalid argument");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    if (((size_t)1 << s->log_k) >> addr_round != s->k_size) {
        set_error("zg_rwc_round_address: addr_round does not match the address binds so far");
        return ZG_ERR_INVALID;
    }
    ZG_TRY(rwc_host_skeleton(s));
    ZG_TRY(rwc_to_address_major(s));
    ZG_TRY(rwc_plan_address(s, addr_round));
    const uint32_t m = (uint32_t)s->plan.size();
    if (m == 0) {
        for (int i = 0; i < 4; i++) s0[i] = s2[i] = 0;
        return ZG_OK;
    }
    ZG_TRY(rwc_upload_plan(s));
    static RwcChal zero_ch;
    RwcChal ch = zero_ch;
    ZG_TRY(rwc_chal(ch, challenges, addr_round));
    const uint32_t nb = div_up(m, 256);
    hipLaunchKernelGGL(rwc_address_round_kernel, dim3(nb), dim3(256), 0, s->st, s->d_plan, m, s->ra[s->cur], s->val_c[s->cur], ch, s->val[s->kcur],
                       (uint32_t)s->k_size, rwc_fr_arg(s->eq_cycle), rwc_fr_arg(gamma), rwc_fr_arg(s->inc_scalar), s->d_part);
    ZG_HIP(hipGetLastError());
    return rwc_collect(s, nb, s0, s2);
}

int zg_rwc_bind_address(zg_rwc_t s, size_t addr_round, const uint64_t r[4]) {
    ZG_INIT();
    if (!s || !r || addr_round >= s->log_k || ((size_t)1 << s->log_k) >> addr_round != s->k_size) {
        set_error("zg_rwc_bind_address: invalid argument");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    ZG_TRY(rwc_host_skeleton(s));
    ZG_TRY(rwc_to_address_major(s));
    const size_t size = s->k_size;  // >= 2 here
    const int kn = s->kcur ^ 1;
    unsigned nf = div_up(size / 2, 256);
    if (nf > 4096) nf = 4096;
    // val_init LowToHigh (:953-959), out of place: the previous level stays readable for the checkpoints of this bind
    hipLaunchKernelGGL(rwc_fold_kernel, dim3(nf), dim3(256), 0, s->st, s->val[s->kcur], s->val[kn], (const uint64_t *)nullptr, (uint64_t *)nullptr, size / 2,
                       rwc_fr_arg(r));
    ZG_HIP(hipGetLastError());
    if (!s->plan_valid || !s->plan_is_address || s->plan_addr_round != addr_round) {
        ZG_TRY(rwc_plan_address(s, addr_round));
        ZG_TRY(rwc_upload_plan(s));
    }
    const uint32_t m = (uint32_t)s->plan.size();
    if (m) {
        hipLaunchKernelGGL(rwc_address_bind_kernel, dim3(div_up(m, 256)), dim3(256), 0, s->st, s->d_plan, m, s->ra[s->cur], s->val_c[s->cur], s->val[kn],
                           s->val[s->kcur], (uint32_t)size, rwc_fr_arg(r), s->ra[s->cur ^ 1], s->val_c[s->cur ^ 1]);
        ZG_HIP(hipGetLastError());
        s->cur ^= 1;
    }
    rwc_apply_plan(s);
    s->kcur = kn;
    s->k_size = size / 2;
    return ZG_OK;
}

int zg_rwc_opening(zg_rwc_t s, const uint64_t *r_address, const uint64_t *r_cycle, uint64_t out[12]) {
    ZG_INIT();
    if (!s || !out || (s->log_k && !r_address) || (s->log_t && !r_cycle) || s->log_k + s->log_t > 48) {
        set_error("zg_rwc_opening: invalid argument");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    // v0 = val_init[0] (every entry's column once the address variables are bound), inc_claim = inc[0]
    ZG_HIP(hipMemcpyAsync(s->h_out, s->val[s->kcur], 32, hipMemcpyDeviceToHost, s->st));
    ZG_HIP(hipMemcpyAsync(s->h_out + 4, s->inc[s->vcur], 32, hipMemcpyDeviceToHost, s->st));
    ZG_HIP(hipStreamSynchronize(s->st));
    uint64_t v0[4], inc0[4];
    for (int i = 0; i < 4; i++) {
        v0[i] = s->h_out[i];
        inc0[i] = s->h_out[4 + i];
    }
    uint64_t ra[4] = {0, 0, 0, 0}, dv[4] = {0, 0, 0, 0};
    const size_t n = s->n_entries;
    if (n) {
        static RwcPoint zero_pt;
        RwcPoint pt = zero_pt;
        pt.n_addr = (int)s->log_k;
        pt.n_cyc = (int)s->log_t;
        for (size_t j = 0; j < s->log_k + s->log_t; j++) {
            const uint64_t *src = j < s->log_k ? r_address + 4 * j : r_cycle + 4 * (j - s->log_k);
            for (int w = 0; w < 4; w++) {
                pt.r[j][2 * w] = (uint32_t)src[w];
                pt.r[j][2 * w + 1] = (uint32_t)(src[w] >> 32);
            }
        }
        ZG_TRY(rwc_dev_skeleton(s));
        const uint32_t nb = div_up(n, 256);
        hipLaunchKernelGGL(rwc_opening_kernel, dim3(nb), dim3(256), 0, s->st, s->d_sk32[s->sk], s->d_sk32[s->sk] + s->cap, s->ra[s->cur], s->val_c[s->cur],
                           (uint32_t)n, pt, rwc_fr_arg(v0), s->d_part);
        ZG_HIP(hipGetLastError());
        ZG_TRY(rwc_collect(s, nb, ra, dv));
    }
    uint64_t val[4];
    fr_add_host(val, v0, dv);
    for (int i = 0; i < 4; i++) {
        out[i] = ra[i];
        out[4 + i] = val[i];
        out[8 + i] = inc0[i];
    }
    return ZG_OK;
}

int zg_rwc_cycle_scalars(zg_rwc_t s, uint64_t eq0[4], uint64_t inc0[4]) {
    ZG_INIT();
    if (!s || !eq0 || !inc0) {
        set_error("zg_rwc_cycle_scalars: invalid argument");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    ZG_HIP(hipMemcpyAsync(s->h_out, s->eq[s->vcur], 32, hipMemcpyDeviceToHost, s->st));
    ZG_HIP(hipMemcpyAsync(s->h_out + 4, s->inc[s->vcur], 32, hipMemcpyDeviceToHost, s->st));
    ZG_HIP(hipStreamSynchronize(s->st));
    for (int i = 0; i < 4; i++) {
        eq0[i] = s->h_out[i];
        inc0[i] = s->h_out[4 + i];
    }
    return ZG_OK;
}

int zg_rwc_read_entries(zg_rwc_t s, uint32_t *cycle, uint32_t *address, uint64_t *ra_coeff, uint64_t *val_coeff, uint64_t *prev_val, uint64_t *next_val) {
    ZG_INIT();
    if (!s) {
        set_error("zg_rwc_read_entries: invalid argument");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(s->device);
    std::lock_guard<std::mutex> lk(s->mu);
    ZG_TRY(rwc_host_skeleton(s));
    const size_t n = s->n_entries;
    for (size_t i = 0; i < n; i++) {
        if (cycle) cycle[i] = s->cycle[i];
        if (address) address[i] = s->addr[i];
        if (prev_val) prev_val[i] = s->prev[i];
        if (next_val) next_val[i] = s->next[i];
    }
    if (n && ra_coeff) ZG_HIP(hipMemcpyAsync(ra_coeff, s->ra[s->cur], n * 32, hipMemcpyDeviceToHost, s->st));
    if (n && val_coeff) ZG_HIP(hipMemcpyAsync(val_coeff, s->val_c[s->cur], n * 32, hipMemcpyDeviceToHost, s->st));
    ZG_HIP(hipStreamSynchronize(s->st));
    return ZG_OK;
}

int zg_rwc_close(zg_rwc_t s) {
    if (!s) return ZG_OK;
    ZG_INIT();
    DeviceGuard dg(s->device);
    (void)hipStreamSynchronize(s->st);
    rwc_free(s);
    return ZG_OK;
}

}  // extern "C"
