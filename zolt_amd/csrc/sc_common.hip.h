// sc_common.hip.h — device helpers shared by the sumcheck kernels of poly.hip and psc.hip: wave / block reductions of Fr pairs,
// the pinned-mailbox publication and the by-value challenge argument.
#pragma once
#include "field.hip.h"
#include "fp29.hip.h"

namespace zg {

ZG_DEV Fr fr_shfl_down(const Fr &v, int d) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = __shfl_down(v.l[i], d, 64);
    return r;
}

// block-wide sum of (g0, g1) pairs (a multiple of 64 threads, at most 1024); result valid in thread 0. Wave-level shuffle tree first
// (no barriers, no LDS round trips), then one LDS hop across the four waves: the latency of this reduction is
// what a small sumcheck round mostly consists of.
__device__ __forceinline__ void block_sum_pair(Fr &g0, Fr &g1, uint4 *sh) {
    uint32_t tid = threadIdx.x;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        g0 = fe_add(g0, fr_shfl_down(g0, d));
        g1 = fe_add(g1, fr_shfl_down(g1, d));
    }
    if ((tid & 63) == 0) {
        fe_store(&sh[(tid >> 6) * 4], g0);
        fe_store(&sh[(tid >> 6) * 4 + 2], g1);
    }
    __syncthreads();
    if (tid == 0) {
        const uint32_t nwaves = blockDim.x >> 6;  // 4 for the 256-thread launches, up to 16 (sh holds 4 * nwaves entries)
        for (uint32_t w = 1; w < nwaves; w++) {
            g0 = fe_add(g0, fe_load<FrParams>(&sh[w * 4]));
            g1 = fe_add(g1, fe_load<FrParams>(&sh[w * 4 + 2]));
        }
    }
}

// A kernel that produces the final values of a round publishes them to the pinned host mailbox: values first, then
// (after a system-scope fence) the round's sequence number, so the host can spin on the mailbox instead of
// paying a stream synchronisation per round. flag == nullptr: nothing is published.
ZG_DEV void publish_seq(uint64_t *flag, uint64_t seq) {
    if (flag) {
        __threadfence_system();
        __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Arrival of a workgroup at the end of a round; returns true in the one that arrives last (after every other group's partials are
// visible to it). Same-address atomics serialise at ~30 ns each: 2048 workgroups on one counter cost 60 us (measured: a 2^20-entry
// fold 37 -> 100 us), which pinned the grids to one workgroup per CU and the long folds to 3-4 TB/s. Above SC_ARRIVE_FLAT groups the
// arrival is two-level: 16 counters on separate 128-byte lines (blockIdx mod 16), the last arrival of each line moves on to the top
// counter — at most nb / 16 + 16 serialised atomics. Every counter is left at zero for the next launch (stream order).
constexpr uint32_t SC_ARRIVE_FLAT = 64, SC_ARRIVE_LINES = 16, SC_ARRIVE_STRIDE = 32;  // uint32 words between counters (128 bytes)
constexpr size_t SC_COUNTER_BYTES = 128 * (1 + SC_ARRIVE_LINES);
ZG_DEV bool sc_arrive(uint32_t *counter, uint32_t nb) {
    if (nb <= SC_ARRIVE_FLAT) {
        uint32_t arrived = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived != nb - 1) return false;
        __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
    }
    const uint32_t line = blockIdx.x % SC_ARRIVE_LINES;
    const uint32_t members = (nb - line + SC_ARRIVE_LINES - 1) / SC_ARRIVE_LINES;  // workgroups b < nb with b mod 16 == line
    uint32_t *lc = counter + SC_ARRIVE_STRIDE * (1 + line);
    if (__hip_atomic_fetch_add(lc, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) != members - 1) return false;
    __hip_atomic_store(lc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) != SC_ARRIVE_LINES - 1) return false;
    __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

struct FrArg {  // a challenge travels as a kernel argument: no H2D copy, no staging buffer to recycle
    uint32_t l[8];
};

}  // namespace zg
