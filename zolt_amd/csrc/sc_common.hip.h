// sc_common.hip.h — device helpers shared by the sumcheck kernels of poly.hip and psc.hip: wave / block reductions of Fr pairs,
// the pinned-mailbox publication and the by-value challenge argument.
#pragma once
#include "field.hip.h"
#include "fp29.hip.h"

namespace zg {

typedef __attribute__((address_space(1))) uint64_t sc_gu64;
typedef __attribute__((address_space(1))) uint32_t sc_gu32;

ZG_DEV Fr fr_shfl_down(const Fr &v, int d) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = __shfl_down(v.l[i], d, 64);
    return r;
}

// ---- lazy sums. A sum of canonical scalars is kept as a plain 288-bit integer (nine 32-bit limbs, no modular reduction: 9
// add-with-carry instructions per term instead of ~45 for fe_add) and reduced to the canonical element ONCE, where the value leaves
// the kernel. Integer addition is exact, so the canonical result is the reference's whatever the order and grouping. Bound: at most
// 2^30 terms (< 2^284) between two reductions — every table this library accepts is shorter.
struct Acc9 {
    u32 l[9];
};
ZG_DEV Acc9 acc9_zero() {
    Acc9 a;
#pragma unroll
    for (int i = 0; i < 9; i++) a.l[i] = 0;
    return a;
}
ZG_DEV void acc9_add(Acc9 &a, const Fr &v) {
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) a.l[i] = __builtin_addc(a.l[i], v.l[i], c, &c);
    a.l[8] += c;
}
ZG_DEV void acc9_add_if(Acc9 &a, const Fr &v, bool take) {  // a += take ? v : 0
    const u32 m = take ? 0xffffffffu : 0u;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) a.l[i] = __builtin_addc(a.l[i], v.l[i] & m, c, &c);
    a.l[8] += c;
}
ZG_DEV void acc9_add_acc(Acc9 &a, const Acc9 &b) {
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) a.l[i] = __builtin_addc(a.l[i], b.l[i], c, &c);
    a.l[8] += b.l[8] + c;
}
// S mod r for S < 2^284. Quotient estimate from the top 64 bits: q = floor(floor(S / 2^224) * floor(2^64 / d) / 2^64) with
// d = ceil(r / 2^224) never exceeds floor(S / r) and falls short of it by less than 3.8 for S / r < 2^30.4 (relative error of d:
// 1.24e-9), by at most 2 for S / r < 2^20 — two estimate-and-subtract passes leave a value < 3 r, two conditional subtractions
// finish. ~80 instructions, executed by the one or two lanes that hold a total.
ZG_DEV Fr acc9_reduce(const Acc9 &s) {
    constexpr u64 MAGIC = 0x54a474622ull;  // floor(2^64 / 811880051), 811880051 = ceil(r / 2^224)
    u32 x[9];
#pragma unroll
    for (int i = 0; i < 9; i++) x[i] = s.l[i];
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        const u64 hi = ((u64)x[8] << 32) | x[7];
        const u32 q = (u32)__umul64hi(hi, MAGIC);  // < 2^31 for S < 2^284
        u64 carry = 0;
        u32 borrow = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            carry += (u64)q * FrParams::MOD[i];
            x[i] = __builtin_subc(x[i], (u32)carry, borrow, &borrow);
            carry >>= 32;
        }
        x[8] = x[8] - (u32)carry - borrow;
    }
    Fr v;  // < 3 r < 2^256: x[8] == 0
#pragma unroll
    for (int i = 0; i < 8; i++) v.l[i] = x[i];
    return fe_reduce_once(fe_reduce_once(v));
}

template <int CTRL, int ROW_MASK>
ZG_DEV u32 dpp_take(u32 v) {  // lanes without a source (or outside ROW_MASK) read 0
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true);
}
// x += (x moved across lanes by a DPP control), nine limbs with the carry chain, the lane movement folded INTO the additions
// (v_add_co_u32_dpp / v_addc_co_u32_dpp): nine instructions. Left to the compiler the same step was a v_mov_b32_dpp, a hazard s_nop
// and a v_addc per limb (27). Lanes without a source add 0 (bound_ctrl), lanes of rows outside the row mask are not written.
// Hazards: a DPP read needs two wait states after a VALU write of the same register — inside the sequence every source limb was
// written nine instructions earlier; the leading s_nop 1 covers whatever the compiler scheduled in front.
#define ZG_ACC9_DPP_ADD(NAME, CTRL_TEXT)                                                                                              \
    ZG_DEV void NAME(Acc9 &x) {                                                                                                       \
        asm volatile("s_nop 1\n\t"                                                                                                    \
                     "v_add_co_u32_dpp %0, vcc, %0, %0 " CTRL_TEXT "\n\t"                                                             \
                     "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc " CTRL_TEXT "\n\t"                                                       \
                     "v_addc_co_u32_dpp %2, vcc, %2, %2, vcc " CTRL_TEXT "\n\t"                                                       \
                     "v_addc_co_u32_dpp %3, vcc, %3, %3, vcc " CTRL_TEXT "\n\t"                                                       \
                     "v_addc_co_u32_dpp %4, vcc, %4, %4, vcc " CTRL_TEXT "\n\t"                                                       \
                     "v_addc_co_u32_dpp %5, vcc, %5, %5, vcc " CTRL_TEXT "\n\t"                                                       \
                     "v_addc_co_u32_dpp %6, vcc, %6, %6, vcc " CTRL_TEXT "\n\t"                                                       \
                     "v_addc_co_u32_dpp %7, vcc, %7, %7, vcc " CTRL_TEXT "\n\t"                                                       \
                     "v_addc_co_u32_dpp %8, vcc, %8, %8, vcc " CTRL_TEXT "\n\t"                                                       \
                     : "+v"(x.l[0]), "+v"(x.l[1]), "+v"(x.l[2]), "+v"(x.l[3]), "+v"(x.l[4]), "+v"(x.l[5]), "+v"(x.l[6]), "+v"(x.l[7]),  \
                       "+v"(x.l[8])                                                                                                   \
                     :                                                                                                                \
                     : "vcc");                                                                                                        \
    }
ZG_ACC9_DPP_ADD(acc9_row_shr1_add, "row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
ZG_ACC9_DPP_ADD(acc9_row_shr2_add, "row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1")
ZG_ACC9_DPP_ADD(acc9_row_shr4_add, "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1")
ZG_ACC9_DPP_ADD(acc9_row_shr8_add, "row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1")
ZG_ACC9_DPP_ADD(acc9_row_bcast15_add, "row_bcast:15 row_mask:0xa bank_mask:0xf")  // rows 1 and 3 += lane 15 of the row before
#undef ZG_ACC9_DPP_ADD
// x += x shifted right by N lanes inside each row of 16 (lanes without a source add 0)
template <int N>
ZG_DEV void acc9_row_shr_add(Acc9 &x) {
    if (N == 1) acc9_row_shr1_add(x);
    else if (N == 2) acc9_row_shr2_add(x);
    else if (N == 4) acc9_row_shr4_add(x);
    else acc9_row_shr8_add(x);
}

// Sum of (g0, g1) over a wavefront without LDS traffic: one v_permlane32_swap per limb puts the g0 halves on lanes 0-31 and the g1
// halves on lanes 32-63 (one addition then serves both sums), four row_shr steps inside the rows of 16 and one row_bcast:15 step
// across the row pair. On return lane 31 holds the wave's g0 total and lane 63 its g1 total (in `g0`; g1 is scratch).
ZG_DEV void wave_sum_pair9(Acc9 &g0, Acc9 &g1) {
#pragma unroll
    for (int i = 0; i < 9; i++) {
        auto sw = __builtin_amdgcn_permlane32_swap(g0.l[i], g1.l[i], false, false);
        g0.l[i] = sw[0];
        g1.l[i] = sw[1];
    }
    acc9_add_acc(g0, g1);
    acc9_row_shr_add<1>(g0);
    acc9_row_shr_add<2>(g0);
    acc9_row_shr_add<4>(g0);
    acc9_row_shr_add<8>(g0);
    acc9_row_bcast15_add(g0);  // rows 1 and 3 += lane 15 of rows 0 and 2
}

// Block-wide sum of (g0, g1) (a multiple of 64 threads, at most 1024). `sh`: SC_RED_WORDS u32 of LDS. On return, in wave 0, lane
// SC_LANE_G0 holds the canonical g0 total and lane SC_LANE_G1 the canonical g1 total (both in the returned value; other lanes hold
// nothing useful). One barrier; the caller must put a barrier between two uses of the same `sh`.
constexpr u32 SC_RED_WORDS = 16 * 2 * 9, SC_LANE_G0 = 15, SC_LANE_G1 = 47;
ZG_DEV Fr block_sum_pair9(Acc9 g0, Acc9 g1, u32 *sh) {
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, nwaves = blockDim.x >> 6;
    wave_sum_pair9(g0, g1);
    if (nwaves > 1) {
        if ((lane & 31u) == 31u) {
            u32 *dst = sh + (wave * 2 + (lane >> 5)) * 9;
#pragma unroll
            for (int i = 0; i < 9; i++) dst[i] = g0.l[i];
        }
        __syncthreads();
        if (wave != 0) return Fr::zero();
        // lanes 0..nwaves-1 (row 0) take the waves' g0 totals, lanes 32..32+nwaves-1 (row 2) their g1 totals
        const u32 w = lane & 31u;
        g0 = acc9_zero();
        if (w < nwaves) {
            const u32 *src = sh + (w * 2 + (lane >> 5)) * 9;
#pragma unroll
            for (int i = 0; i < 9; i++) g0.l[i] = src[i];
        }
        acc9_row_shr_add<1>(g0);
        acc9_row_shr_add<2>(g0);
        if (nwaves > 4) {
            acc9_row_shr_add<4>(g0);
            acc9_row_shr_add<8>(g0);
        } else {  // totals sit on lanes 3 / 35: move them to lanes 15 / 47 (row_shr:12)
#pragma unroll
            for (int i = 0; i < 9; i++) g0.l[i] = dpp_take<0x110 + 12, 0xf>(g0.l[i]);
        }
    } else {  // one wave: totals on lanes 31 / 63 -> lanes 15 / 47 (row_shr... crosses rows: read them through a shuffle)
#pragma unroll
        for (int i = 0; i < 9; i++) g0.l[i] = __shfl(g0.l[i], (int)((lane & 32u) | 31u), 64);
    }
    return acc9_reduce(g0);
}
// wave 0 only: lane SC_LANE_G0 also receives lane SC_LANE_G1's value (the pair in one lane for a verifier step / mailbox store)
ZG_DEV Fr pair_second_to_first(const Fr &v) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = __shfl(v.l[i], (int)SC_LANE_G1, 64);
    return r;
}

// A kernel that produces the final values of a round publishes them to the pinned host mailbox (hipHostMalloc, mapped + coherent:
// uncached on the device): values first as write-through system-scope stores, the wave's s_waitcnt vmcnt(0), then the round's sequence
// number, so the host can spin on the mailbox instead of paying a stream synchronisation per round. No release fence: round 3's
// __threadfence_system() here was a buffer_wbl2 — a write-back of the XCD's L2 right behind the fold that had just dirtied it — and the
// mailbox never sits in that L2. flag == nullptr: the values go to device memory with plain stores (read by a later launch or copy).
ZG_DEV void mailbox_store_fr(uint64_t *dst, const Fr &v, const uint64_t *flag) {
    if (flag) {
        sc_gu64 *d = (sc_gu64 *)dst;
#pragma unroll
        for (int i = 0; i < 4; i++)
            __hip_atomic_store(d + i, (uint64_t)v.l[2 * i] | ((uint64_t)v.l[2 * i + 1] << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
        fe_store(dst, v);
    }
}
ZG_DEV void publish_seq(uint64_t *flag, uint64_t seq) {  // by a lane of the wave that stored the values
    if (flag) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store((sc_gu64 *)flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Round-3 interface kept for the kernels that still sum canonical pairs: same result (thread 0 receives both canonical totals), now
// through the lazy-sum / DPP reduction above (the old shuffle tree of modular additions was ~3 us of a small round).
// `sh`: at least SC_RED_WORDS u32; a barrier is needed between two calls that share it.
__device__ __forceinline__ void block_sum_pair(Fr &g0, Fr &g1, void *sh) {
    Acc9 a0 = acc9_zero(), a1 = acc9_zero();
    acc9_add(a0, g0);
    acc9_add(a1, g1);
    Fr tot = block_sum_pair9(a0, a1, reinterpret_cast<u32 *>(sh));
    if (threadIdx.x < 64) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            g0.l[i] = __shfl(tot.l[i], (int)SC_LANE_G0, 64);
            g1.l[i] = __shfl(tot.l[i], (int)SC_LANE_G1, 64);
        }
    }
}

// ---- hand-off of a workgroup's partial sums to the workgroup that finishes the round (MI355X_MICROARCH.md, inter-workgroup
// visibility, "hand-offs measured with sc1 loads", first row): the partials are written with write-through (sc1) stores by ONE wave,
// that wave drains them (s_waitcnt vmcnt(0)), then ONE of its lanes adds to the arrival counter with a RELAXED agent-scope atomic;
// the workgroup whose add returns the last ticket reads the partials with sc1 loads (they bypass its L1; no line of the partials
// buffer is ever loaded any other way inside a launch). No release fence and no per-workgroup acquire: round 3's ACQ_REL arrival wrote
// back and invalidated the XCD's L2 behind a freshly written table once per workgroup. The LAST arriver alone runs one agent-scope
// acquire per launch (sc_arrive), on both arrival paths.
ZG_DEV void sc1_store_fr(uint64_t *dst, const Fr &v) {
    sc_gu64 *d = (sc_gu64 *)dst;
#pragma unroll
    for (int i = 0; i < 4; i++)
        __hip_atomic_store(d + i, (uint64_t)v.l[2 * i] | ((uint64_t)v.l[2 * i + 1] << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
ZG_DEV Fr sc1_load_fr(const uint64_t *src) {
    const sc_gu64 *p = (const sc_gu64 *)src;
    Fr r;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint64_t w = __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        r.l[2 * i] = (uint32_t)w;
        r.l[2 * i + 1] = (uint32_t)(w >> 32);
    }
    return r;
}
ZG_DEV void sc_drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Arrival (called by ONE lane, after the storing wave's sc_drain_stores()); true in the workgroup that arrives last. Up to
// SC_ARRIVE_FLAT workgroups add to one counter (the table row above, literally: the last adder is told by the value its add returned;
// 255 -> 1 arrivals cost ~3 us when they all come at once). Larger grids arrive in two levels — 16 counters on lines of their own
// (blockIdx mod 16), the last arrival of each line moves on to the top counter — which is outside that row, so the final arriver
// runs ONE agent-scope acquire (then s_waitcnt vmcnt(0); the caller's barrier follows) before anyone loads: the guide's "consumer,
// always" form, with the sc1 stores standing in for the release. Every counter is left at zero for the next launch (stream order).
constexpr uint32_t SC_ARRIVE_FLAT = 256, SC_ARRIVE_LINES = 16, SC_ARRIVE_STRIDE = 32;  // uint32 words between counters (128 bytes)
constexpr size_t SC_COUNTER_BYTES = 128 * (1 + SC_ARRIVE_LINES);
ZG_DEV bool sc_arrive(uint32_t *counter, uint32_t nb) {
    sc_gu32 *c = (sc_gu32 *)counter;
    if (nb <= SC_ARRIVE_FLAT) {
        if (__hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != nb - 1) return false;
        __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ONE agent-scope acquire per launch, in the last arriver only (round-4 advisor finding: the flat path relied on the measured
        // table row alone, the two-level path fenced). The fence is `s_waitcnt vmcnt(0); buffer_inv sc1`: the invalidate is queued in
        // front of every later vector-memory instruction of this wave, the other waves load after the caller's barrier.
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        return true;
    }
    const uint32_t line = blockIdx.x % SC_ARRIVE_LINES;
    const uint32_t members = (nb - line + SC_ARRIVE_LINES - 1) / SC_ARRIVE_LINES;  // workgroups b < nb with b mod 16 == line
    sc_gu32 *lc = c + SC_ARRIVE_STRIDE * (1 + line);
    if (__hip_atomic_fetch_add(lc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != members - 1) return false;
    __hip_atomic_store(lc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != SC_ARRIVE_LINES - 1) return false;
    __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return true;
}

struct FrArg {  // a challenge travels as a kernel argument: no H2D copy, no staging buffer to recycle
    uint32_t l[8];
};

}  // namespace zg
