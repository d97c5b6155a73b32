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

struct FrArg {  // a challenge travels as a kernel argument: no H2D copy, no staging buffer to recycle
    uint32_t l[8];
};

}  // namespace zg
