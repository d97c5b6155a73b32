// fold_ab.hip — round 4 A/B of the sumcheck fold / sums kernels at config 3's sizes: loop shapes (grid-stride with one
// prefetch, U pairs requested up front), workgroup sizes, grid sizes and round-ending arrivals. Every variant's outputs and sums are
// compared with variant 0's. Run under rocprofv3 --kernel-trace --stats for per-kernel durations (the template arguments are in
// the kernel names); the program itself prints HIP-event times of back-to-back launches (boundaries included).
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp/fold_ab.hip -o tools/exp/fold_ab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../zolt_amd/csrc/sc_common.hip.h"

using namespace zg;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(1))) uint64_t gu64;
typedef __attribute__((address_space(1))) uint32_t gu32;

enum { ARR_ACQREL = 0, ARR_SC1_FLAT = 1, ARR_SC1_2LVL_ACQ = 2, ARR_NONE = 3, ARR_SC1_2LVL_NOACQ = 4 };

constexpr uint32_t LINES = 16, STRIDE = 32;

ZG_DEV void store_pair_sc1(uint64_t *dst, const Fr &g0, const Fr &g1) {
    gu64 *d = (gu64 *)dst;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        __hip_atomic_store(d + i, (uint64_t)g0.l[2 * i] | ((uint64_t)g0.l[2 * i + 1] << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(d + 4 + i, (uint64_t)g1.l[2 * i] | ((uint64_t)g1.l[2 * i + 1] << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
ZG_DEV Fr load_fr_sc1(const uint64_t *src) {
    const gu64 *s = (const gu64 *)src;
    Fr r;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint64_t w = __hip_atomic_load(s + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        r.l[2 * i] = (uint32_t)w;
        r.l[2 * i + 1] = (uint32_t)(w >> 32);
    }
    return r;
}

template <int ARR>
ZG_DEV bool arrive(uint32_t *counter, uint32_t nb) {
    gu32 *c = (gu32 *)counter;
    if (ARR == ARR_ACQREL) {
        if (nb <= 64) {
            uint32_t a = __hip_atomic_fetch_add(c, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (a != nb - 1) return false;
            __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return true;
        }
        const uint32_t line = blockIdx.x % LINES, members = (nb - line + LINES - 1) / LINES;
        gu32 *lc = c + STRIDE * (1 + line);
        if (__hip_atomic_fetch_add(lc, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) != members - 1) return false;
        __hip_atomic_store(lc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(c, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) != LINES - 1) return false;
        __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
    }
    if (ARR == ARR_SC1_FLAT) {
        uint32_t a = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a != nb - 1) return false;
        __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
    }
    // two-level, relaxed
    if (nb <= 64) {
        uint32_t a = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a != nb - 1) return false;
        __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
    }
    const uint32_t line = blockIdx.x % LINES, members = (nb - line + LINES - 1) / LINES;
    gu32 *lc = c + STRIDE * (1 + line);
    if (__hip_atomic_fetch_add(lc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != members - 1) return false;
    __hip_atomic_store(lc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != LINES - 1) return false;
    __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

// g0/g1 valid in thread 0
template <int ARR>
ZG_DEV void finish(Fr &g0, Fr &g1, uint4 *sh, uint64_t *partials, uint64_t *sums, uint32_t *counter) {
    const uint32_t tid = threadIdx.x, nb = gridDim.x;
    __shared__ uint32_t last;
    if (ARR == ARR_NONE) {
        if (tid == 0) {
            fe_store(partials + 8 * (size_t)blockIdx.x, g0);
            fe_store(partials + 8 * (size_t)blockIdx.x + 4, g1);
        }
        return;
    }
    if (tid == 0) {
        uint64_t *dst = partials + 8 * (size_t)blockIdx.x;
        if (ARR == ARR_ACQREL) {
            fe_store(dst, g0);
            fe_store(dst + 4, g1);
        } else {
            store_pair_sc1(dst, g0, g1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        bool l = arrive<ARR>(counter, nb);
        if (l && ARR == ARR_SC1_2LVL_ACQ) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        last = l ? 1u : 0u;
    }
    __syncthreads();
    if (!last) return;
    Fr a0 = Fr::zero(), a1 = Fr::zero();
    for (uint32_t k = tid; k < nb; k += blockDim.x) {
        const uint64_t *src = partials + 8 * (size_t)k;
        if (ARR == ARR_ACQREL) {
            a0 = fe_add(a0, fe_load<FrParams>(src));
            a1 = fe_add(a1, fe_load<FrParams>(src + 4));
        } else {
            a0 = fe_add(a0, load_fr_sc1(src));
            a1 = fe_add(a1, load_fr_sc1(src + 4));
        }
    }
    __syncthreads();
    block_sum_pair(a0, a1, sh);
    if (tid == 0) {
        fe_store(sums, a0);
        fe_store(sums + 4, a1);
    }
}

// ---- fold, HIGH layout. GS: grid-stride loop with one prefetched pair (the round-3 kernel).
template <int ARR, int THREADS>
__global__ void __launch_bounds__(THREADS) fold_gs(const uint64_t *t, size_t half, FrArg r, uint64_t *out, uint64_t *partials, uint64_t *sums,
                                                   uint32_t *counter) {
    __shared__ uint4 sh[256 * 4];
    Fr rv;
#pragma unroll
    for (int i = 0; i < 8; i++) rv.l[i] = r.l[i];
    FrMul rp = frmul_prepare(rv);
    Fr g0 = Fr::zero(), g1 = Fr::zero();
    size_t stride = (size_t)gridDim.x * blockDim.x, quarter = half / 2;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fr lo = Fr::zero(), hi = Fr::zero();
    if (i < half) {
        lo = fe_load<FrParams>(t + 4 * i);
        hi = fe_load<FrParams>(t + 4 * (i + half));
    }
    while (i < half) {
        size_t ni = i + stride;
        Fr nlo = lo, nhi = hi;
        if (ni < half) {
            nlo = fe_load<FrParams>(t + 4 * ni);
            nhi = fe_load<FrParams>(t + 4 * (ni + half));
        }
        Fr v = fe_add(lo, frmul_apply(fe_sub(hi, lo), rp));
        fe_store(out + 4 * i, v);
        if (i >= quarter) g1 = fe_add(g1, v);
        else g0 = fe_add(g0, v);
        lo = nlo;
        hi = nhi;
        i = ni;
    }
    block_sum_pair(g0, g1, sh);
    finish<ARR>(g0, g1, sh, partials, sums, counter);
}

// UP: the block owns a contiguous tile of THREADS * U pairs; every thread requests its U pairs before the first product.
template <int ARR, int THREADS, int U>
__global__ void __launch_bounds__(THREADS) fold_up(const uint64_t *t, size_t half, FrArg r, uint64_t *out, uint64_t *partials, uint64_t *sums,
                                                   uint32_t *counter) {
    __shared__ uint4 sh[256 * 4];
    Fr rv;
#pragma unroll
    for (int i = 0; i < 8; i++) rv.l[i] = r.l[i];
    const size_t base = (size_t)blockIdx.x * (THREADS * U) + threadIdx.x, quarter = half / 2;
    Fr lo[U], hi[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
        size_t i = base + (size_t)k * THREADS;
        if (i < half) {
            lo[k] = fe_load<FrParams>(t + 4 * i);
            hi[k] = fe_load<FrParams>(t + 4 * (i + half));
        }
    }
    FrMul rp = frmul_prepare(rv);
    Fr g0 = Fr::zero(), g1 = Fr::zero();
#pragma unroll
    for (int k = 0; k < U; k++) {
        size_t i = base + (size_t)k * THREADS;
        if (i < half) {
            Fr v = fe_add(lo[k], frmul_apply(fe_sub(hi[k], lo[k]), rp));
            fe_store(out + 4 * i, v);
            if (i >= quarter) g1 = fe_add(g1, v);
            else g0 = fe_add(g0, v);
        }
    }
    block_sum_pair(g0, g1, sh);
    finish<ARR>(g0, g1, sh, partials, sums, counter);
}

// ---- sums only (round 0 of a session): GS as in round 3, UP with U pairs up front
template <int ARR, int THREADS>
__global__ void __launch_bounds__(THREADS) sums_gs(const uint64_t *t, size_t half, uint64_t *partials, uint64_t *sums, uint32_t *counter) {
    __shared__ uint4 sh[256 * 4];
    Fr g0 = Fr::zero(), g1 = Fr::zero();
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += stride) {
        g0 = fe_add(g0, fe_load<FrParams>(t + 4 * i));
        g1 = fe_add(g1, fe_load<FrParams>(t + 4 * (i + half)));
    }
    block_sum_pair(g0, g1, sh);
    finish<ARR>(g0, g1, sh, partials, sums, counter);
}
template <int ARR, int THREADS, int U>
__global__ void __launch_bounds__(THREADS) sums_up(const uint64_t *t, size_t half, uint64_t *partials, uint64_t *sums, uint32_t *counter) {
    __shared__ uint4 sh[256 * 4];
    Fr g0 = Fr::zero(), g1 = Fr::zero();
    size_t stride = (size_t)gridDim.x * THREADS * U;
    for (size_t b = (size_t)blockIdx.x * (THREADS * U) + threadIdx.x; b < half; b += stride) {
        Fr lo[U], hi[U];
#pragma unroll
        for (int k = 0; k < U; k++) {
            size_t i = b + (size_t)k * THREADS;
            lo[k] = Fr::zero();
            hi[k] = Fr::zero();
            if (i < half) {
                lo[k] = fe_load<FrParams>(t + 4 * i);
                hi[k] = fe_load<FrParams>(t + 4 * (i + half));
            }
        }
#pragma unroll
        for (int k = 0; k < U; k++) {
            g0 = fe_add(g0, lo[k]);
            g1 = fe_add(g1, hi[k]);
        }
    }
    block_sum_pair(g0, g1, sh);
    finish<ARR>(g0, g1, sh, partials, sums, counter);
}

// ================= v2: lazy sums (Acc9), DPP wave reduction, sc1 hand-off from the two lanes that hold the block's totals
template <int ARR>
ZG_DEV void finish2(const Acc9 &g0, const Acc9 &g1, u32 *sh, uint64_t *partials, uint64_t *sums, uint32_t *counter) {
    const uint32_t tid = threadIdx.x, nb = gridDim.x, lane = tid & 63u;
    Fr tot = block_sum_pair9(g0, g1, sh);
    const bool holder = tid < 64 && (lane == SC_LANE_G0 || lane == SC_LANE_G1);
    __shared__ uint32_t last;
    if (ARR == ARR_NONE) {
        if (holder) fe_store(partials + 8 * (size_t)blockIdx.x + (lane == SC_LANE_G1 ? 4 : 0), tot);
        return;
    }
    if (tid < 64) {
        if (holder) {
            gu64 *d = (gu64 *)(partials + 8 * (size_t)blockIdx.x + (lane == SC_LANE_G1 ? 4 : 0));
#pragma unroll
            for (int i = 0; i < 4; i++)
                __hip_atomic_store(d + i, (uint64_t)tot.l[2 * i] | ((uint64_t)tot.l[2 * i + 1] << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == SC_LANE_G0) {
            bool l = arrive<ARR>(counter, nb);
            if (l && ARR == ARR_SC1_2LVL_ACQ) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            last = l ? 1u : 0u;
        }
    }
    __syncthreads();
    if (!last) return;
    Acc9 a0 = acc9_zero(), a1 = acc9_zero();
    for (uint32_t k = tid; k < nb; k += blockDim.x) {
        const uint64_t *src = partials + 8 * (size_t)k;
        acc9_add(a0, load_fr_sc1(src));
        acc9_add(a1, load_fr_sc1(src + 4));
    }
    tot = block_sum_pair9(a0, a1, sh);
    if (tid < 64) {
        Fr second = pair_second_to_first(tot);
        if (lane == SC_LANE_G0) {
            fe_store(sums, tot);
            fe_store(sums + 4, second);
        }
    }
}

template <int ARR, int THREADS, int PF>
__global__ void __launch_bounds__(THREADS) fold2_gs(const uint64_t *t, size_t half, FrArg r, uint64_t *out, uint64_t *partials, uint64_t *sums,
                                                    uint32_t *counter) {
    __shared__ u32 sh[SC_RED_WORDS];
    Fr rv;
#pragma unroll
    for (int i = 0; i < 8; i++) rv.l[i] = r.l[i];
    const size_t stride = (size_t)gridDim.x * THREADS, quarter = half / 2;
    size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x;
    Fr lo[PF], hi[PF];
#pragma unroll
    for (int k = 0; k < PF; k++) {
        size_t j = i + k * stride;
        lo[k] = Fr::zero();
        hi[k] = Fr::zero();
        if (j < half) {
            lo[k] = fe_load<FrParams>(t + 4 * j);
            hi[k] = fe_load<FrParams>(t + 4 * (j + half));
        }
    }
    FrMul rp = frmul_prepare(rv);
    Acc9 g0 = acc9_zero(), g1 = acc9_zero();
    while (i < half) {
        Fr clo = lo[0], chi = hi[0];
#pragma unroll
        for (int k = 0; k + 1 < PF; k++) {
            lo[k] = lo[k + 1];
            hi[k] = hi[k + 1];
        }
        size_t ni = i + PF * stride;
        if (ni < half) {
            lo[PF - 1] = fe_load<FrParams>(t + 4 * ni);
            hi[PF - 1] = fe_load<FrParams>(t + 4 * (ni + half));
        }
        Fr v = fe_add(clo, frmul_apply(fe_sub(chi, clo), rp));
        fe_store(out + 4 * i, v);
        const bool second = i >= quarter;
        acc9_add_if(g0, v, !second);
        acc9_add_if(g1, v, second);
        i += stride;
    }
    finish2<ARR>(g0, g1, sh, partials, sums, counter);
}

template <int ARR, int THREADS, int U>
__global__ void __launch_bounds__(THREADS) fold2_up(const uint64_t *t, size_t half, FrArg r, uint64_t *out, uint64_t *partials, uint64_t *sums,
                                                    uint32_t *counter) {
    __shared__ u32 sh[SC_RED_WORDS];
    Fr rv;
#pragma unroll
    for (int i = 0; i < 8; i++) rv.l[i] = r.l[i];
    const size_t base = (size_t)blockIdx.x * (THREADS * U) + threadIdx.x, quarter = half / 2;
    Fr lo[U], hi[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
        size_t i = base + (size_t)k * THREADS;
        if (i < half) {
            lo[k] = fe_load<FrParams>(t + 4 * i);
            hi[k] = fe_load<FrParams>(t + 4 * (i + half));
        }
    }
    FrMul rp = frmul_prepare(rv);
    Acc9 g0 = acc9_zero(), g1 = acc9_zero();
#pragma unroll
    for (int k = 0; k < U; k++) {
        size_t i = base + (size_t)k * THREADS;
        if (i < half) {
            Fr v = fe_add(lo[k], frmul_apply(fe_sub(hi[k], lo[k]), rp));
            fe_store(out + 4 * i, v);
            const bool second = i >= quarter;
            acc9_add_if(g0, v, !second);
            acc9_add_if(g1, v, second);
        }
    }
    finish2<ARR>(g0, g1, sh, partials, sums, counter);
}

template <int ARR, int THREADS, int U>
__global__ void __launch_bounds__(THREADS) sums2_up(const uint64_t *t, size_t half, uint64_t *partials, uint64_t *sums, uint32_t *counter) {
    __shared__ u32 sh[SC_RED_WORDS];
    Acc9 g0 = acc9_zero(), g1 = acc9_zero();
    size_t stride = (size_t)gridDim.x * THREADS * U;
    for (size_t b = (size_t)blockIdx.x * (THREADS * U) + threadIdx.x; b < half; b += stride) {
        Fr lo[U], hi[U];
#pragma unroll
        for (int k = 0; k < U; k++) {
            size_t i = b + (size_t)k * THREADS;
            lo[k] = Fr::zero();
            hi[k] = Fr::zero();
            if (i < half) {
                lo[k] = fe_load<FrParams>(t + 4 * i);
                hi[k] = fe_load<FrParams>(t + 4 * (i + half));
            }
        }
#pragma unroll
        for (int k = 0; k < U; k++) {
            acc9_add(g0, lo[k]);
            acc9_add(g1, hi[k]);
        }
    }
    finish2<ARR>(g0, g1, sh, partials, sums, counter);
}

// diagnostic variants of fold2_gs: MODE 1 = no stores, 2 = nontemporal stores, 3 = no product (v = lo + d), 4 = stores only (no loads: lo = hi = i)
template <int ARR, int THREADS, int MODE>
__global__ void __launch_bounds__(THREADS) fold3_gs(const uint64_t *t, size_t half, FrArg r, uint64_t *out, uint64_t *partials, uint64_t *sums,
                                                    uint32_t *counter) {
    __shared__ u32 sh[SC_RED_WORDS];
    Fr rv;
#pragma unroll
    for (int i = 0; i < 8; i++) rv.l[i] = r.l[i];
    const size_t stride = (size_t)gridDim.x * THREADS, quarter = half / 2;
    size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x;
    Fr lo = Fr::zero(), hi = Fr::zero();
    if (i < half && MODE != 4) {
        lo = fe_load<FrParams>(t + 4 * i);
        hi = fe_load<FrParams>(t + 4 * (i + half));
    }
    FrMul rp = frmul_prepare(rv);
    Acc9 g0 = acc9_zero(), g1 = acc9_zero();
    while (i < half) {
        size_t ni = i + stride;
        Fr nlo = lo, nhi = hi;
        if (ni < half && MODE != 4) {
            nlo = fe_load<FrParams>(t + 4 * ni);
            nhi = fe_load<FrParams>(t + 4 * (ni + half));
        }
        if (MODE == 4) { lo.l[0] = (u32)i; hi.l[1] = (u32)i; }
        Fr d = fe_sub(hi, lo);
        Fr v = fe_add(lo, MODE == 3 ? d : frmul_apply(d, rp));
        if (MODE == 2) {
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            u32x4 *q = reinterpret_cast<u32x4 *>(out + 4 * i);
            u32x4 w0 = {v.l[0], v.l[1], v.l[2], v.l[3]}, w1 = {v.l[4], v.l[5], v.l[6], v.l[7]};
            __builtin_nontemporal_store(w0, q);
            __builtin_nontemporal_store(w1, q + 1);
        } else if (MODE != 1) {
            fe_store(out + 4 * i, v);
        }
        const bool second = i >= quarter;
        acc9_add_if(g0, v, !second);
        acc9_add_if(g1, v, second);
        lo = nlo;
        hi = nhi;
        i = ni;
    }
    finish2<ARR>(g0, g1, sh, partials, sums, counter);
}

__global__ void finish_only(const uint64_t *partials, uint32_t nb, uint64_t *sums) {
    __shared__ uint4 sh[256 * 4];
    Fr a0 = Fr::zero(), a1 = Fr::zero();
    for (uint32_t k = threadIdx.x; k < nb; k += blockDim.x) {
        a0 = fe_add(a0, fe_load<FrParams>(partials + 8 * (size_t)k));
        a1 = fe_add(a1, fe_load<FrParams>(partials + 8 * (size_t)k + 4));
    }
    block_sum_pair(a0, a1, sh);
    if (threadIdx.x == 0) {
        fe_store(sums, a0);
        fe_store(sums + 4, a1);
    }
}

struct Ctx {
    uint64_t *t, *out, *ref_out, *partials, *sums;
    uint32_t *counter;
    size_t n;
    FrArg r;
    std::vector<uint64_t> ref_sums;
    bool have_ref = false;
    hipStream_t st;
};

template <class F>
static void run_variant(Ctx &c, const char *name, unsigned nb, bool two_kernel, F launch, int reps = 40) {
    const size_t half = c.n / 2;
    CHK(hipMemsetAsync(c.out, 0, half * 32, c.st));
    CHK(hipMemsetAsync(c.sums, 0, 64, c.st));
    for (int w = 0; w < 3; w++) {
        launch();
        if (two_kernel) hipLaunchKernelGGL(finish_only, dim3(1), dim3(256), 0, c.st, c.partials, nb, c.sums);
    }
    CHK(hipStreamSynchronize(c.st));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0, c.st));
    for (int k = 0; k < reps; k++) {
        launch();
        if (two_kernel) hipLaunchKernelGGL(finish_only, dim3(1), dim3(256), 0, c.st, c.partials, nb, c.sums);
    }
    CHK(hipEventRecord(e1, c.st));
    CHK(hipEventSynchronize(e1));
    CHK(hipGetLastError());
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    // check
    std::vector<uint64_t> sums(8);
    CHK(hipMemcpy(sums.data(), c.sums, 64, hipMemcpyDeviceToHost));
    bool ok = true;
    if (!c.have_ref) {
        c.ref_sums = sums;
        if (c.out != c.ref_out) CHK(hipMemcpy(c.ref_out, c.out, half * 32, hipMemcpyDeviceToDevice));
        c.have_ref = true;
    } else {
        ok = sums == c.ref_sums;
    }
    printf("  %-44s nb=%5u  %8.2f us/launch  %s\n", name, nb, 1000.0 * ms / reps, ok ? "sums ok" : "SUMS MISMATCH");
    fflush(stdout);
}

int main(int argc, char **argv) {
    int lo_log = argc > 1 ? atoi(argv[1]) : 20, hi_log = argc > 2 ? atoi(argv[2]) : lo_log;
    CHK(hipSetDevice(0));
    Ctx c;
    CHK(hipStreamCreate(&c.st));
    for (int lg = lo_log; lg <= hi_log; lg += 2) {
        c.n = (size_t)1 << lg;
        const size_t half = c.n / 2;
        std::vector<uint64_t> h(c.n * 4);
        uint64_t s = 0x9e3779b97f4a7c15ull;
        for (size_t i = 0; i < c.n * 4; i++) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            h[i] = s ^ (s >> 29);
            if ((i & 3) == 3) h[i] &= 0x0fffffffffffffffull;  // < 2^252 < r
        }
        CHK(hipMalloc(&c.t, c.n * 32));
        CHK(hipMalloc(&c.out, half * 32));
        CHK(hipMalloc(&c.partials, 8 * 8 * 4096 + 128 * 17 + 64));
        c.counter = (uint32_t *)(c.partials + 8 * 4096);
        c.sums = c.partials + 8 * 4096 + 16 * 17;
        CHK(hipMemcpy(c.t, h.data(), c.n * 32, hipMemcpyHostToDevice));
        CHK(hipMemset(c.partials, 0, 8 * 8 * 4096 + 128 * 17 + 64));
        c.ref_out = c.out;
        c.have_ref = false;
        for (int i = 0; i < 8; i++) c.r.l[i] = (uint32_t)(h[8 + i / 2] >> (32 * (i & 1)));
        c.r.l[7] &= 0x0fffffffu;
        printf("== fold, 2^%d entries (half = %zu pairs), %.1f MB algorithmic\n", lg, half, (c.n * 32 + half * 32) / 1e6);
#define FOLD_GS(ARR, TH, NB)                                                                                                     \
    {                                                                                                                            \
        unsigned nb = (unsigned)((half + (TH)-1) / (TH));                                                                        \
        if (nb > (NB)) nb = (NB);                                                                                                \
        run_variant(c, "fold_gs<" #ARR "," #TH "> cap " #NB, nb, ARR == ARR_NONE,                                                \
                    [&] { hipLaunchKernelGGL((fold_gs<ARR, TH>), dim3(nb), dim3(TH), 0, c.st, c.t, half, c.r, c.out, c.partials, c.sums, c.counter); }); \
    }
#define FOLD_UP(ARR, TH, U)                                                                                                      \
    {                                                                                                                            \
        unsigned nb = (unsigned)((half + (TH) * (U)-1) / ((TH) * (U)));                                                          \
        if (nb <= 4096)                                                                                                          \
            run_variant(c, "fold_up<" #ARR "," #TH "," #U ">", nb, ARR == ARR_NONE,                                              \
                        [&] { hipLaunchKernelGGL((fold_up<ARR, TH, U>), dim3(nb), dim3(TH), 0, c.st, c.t, half, c.r, c.out, c.partials, c.sums, c.counter); }); \
    }
        FOLD_GS(ARR_ACQREL, 256, 256)
        FOLD_GS(ARR_SC1_FLAT, 256, 256)
        FOLD_GS(ARR_NONE, 256, 256)
        FOLD_GS(ARR_ACQREL, 1024, 256)
        FOLD_GS(ARR_SC1_FLAT, 1024, 256)
        FOLD_GS(ARR_NONE, 1024, 256)
        FOLD_GS(ARR_SC1_FLAT, 512, 256)
        FOLD_GS(ARR_SC1_FLAT, 256, 512)
        FOLD_GS(ARR_SC1_2LVL_ACQ, 256, 512)
        FOLD_GS(ARR_SC1_2LVL_ACQ, 256, 1024)
        FOLD_GS(ARR_SC1_2LVL_ACQ, 256, 2048)
        FOLD_GS(ARR_SC1_2LVL_NOACQ, 256, 2048)
        FOLD_GS(ARR_NONE, 256, 2048)
        FOLD_UP(ARR_SC1_2LVL_ACQ, 256, 1)
        FOLD_UP(ARR_SC1_2LVL_ACQ, 256, 2)
        FOLD_UP(ARR_SC1_2LVL_ACQ, 256, 4)
        FOLD_UP(ARR_SC1_2LVL_NOACQ, 256, 2)
        FOLD_UP(ARR_NONE, 256, 1)
        FOLD_UP(ARR_NONE, 256, 2)
        FOLD_UP(ARR_NONE, 256, 4)
        FOLD_UP(ARR_SC1_FLAT, 1024, 2)
        FOLD_UP(ARR_SC1_FLAT, 1024, 1)
        FOLD_UP(ARR_SC1_FLAT, 512, 2)
        FOLD_UP(ARR_SC1_2LVL_ACQ, 512, 2)
        FOLD_UP(ARR_SC1_2LVL_ACQ, 512, 1)
        FOLD_UP(ARR_ACQREL, 256, 2)

#define FOLD2_GS(ARR, TH, PF, NB)                                                                                                \
    {                                                                                                                            \
        unsigned nb = (unsigned)((half + (TH)-1) / (TH));                                                                        \
        if (nb > (NB)) nb = (NB);                                                                                                \
        run_variant(c, "fold2_gs<" #ARR "," #TH "," #PF "> cap " #NB, nb, ARR == ARR_NONE,                                       \
                    [&] { hipLaunchKernelGGL((fold2_gs<ARR, TH, PF>), dim3(nb), dim3(TH), 0, c.st, c.t, half, c.r, c.out, c.partials, c.sums, c.counter); }); \
    }
#define FOLD2_UP(ARR, TH, U)                                                                                                     \
    {                                                                                                                            \
        unsigned nb = (unsigned)((half + (TH) * (U)-1) / ((TH) * (U)));                                                          \
        if (nb <= 4096)                                                                                                          \
            run_variant(c, "fold2_up<" #ARR "," #TH "," #U ">", nb, ARR == ARR_NONE,                                             \
                        [&] { hipLaunchKernelGGL((fold2_up<ARR, TH, U>), dim3(nb), dim3(TH), 0, c.st, c.t, half, c.r, c.out, c.partials, c.sums, c.counter); }); \
    }
        FOLD2_GS(ARR_SC1_FLAT, 256, 1, 256)
        FOLD2_GS(ARR_NONE, 256, 1, 256)
        FOLD2_GS(ARR_SC1_FLAT, 256, 2, 256)
        FOLD2_GS(ARR_SC1_FLAT, 512, 1, 256)
        FOLD2_GS(ARR_SC1_FLAT, 512, 2, 256)
        FOLD2_GS(ARR_SC1_FLAT, 1024, 1, 256)
        FOLD2_GS(ARR_NONE, 1024, 1, 256)
        FOLD2_GS(ARR_SC1_FLAT, 1024, 2, 256)
        FOLD2_GS(ARR_SC1_2LVL_ACQ, 256, 1, 512)
        FOLD2_GS(ARR_SC1_2LVL_ACQ, 256, 1, 1024)
        FOLD2_GS(ARR_SC1_2LVL_ACQ, 256, 2, 1024)
        FOLD2_GS(ARR_SC1_2LVL_ACQ, 512, 1, 512)
        FOLD2_GS(ARR_SC1_2LVL_ACQ, 512, 2, 512)
        FOLD2_GS(ARR_SC1_FLAT, 256, 1, 1024)
        FOLD2_GS(ARR_NONE, 256, 1, 1024)
        FOLD2_GS(ARR_SC1_2LVL_ACQ, 256, 1, 2048)
        FOLD2_UP(ARR_SC1_2LVL_ACQ, 256, 1)
        FOLD2_UP(ARR_SC1_2LVL_ACQ, 256, 2)
        FOLD2_UP(ARR_SC1_2LVL_ACQ, 256, 4)
        FOLD2_UP(ARR_NONE, 256, 2)
        FOLD2_UP(ARR_NONE, 256, 4)
        FOLD2_UP(ARR_SC1_2LVL_ACQ, 512, 2)
        FOLD2_UP(ARR_SC1_2LVL_ACQ, 1024, 1)
        FOLD2_UP(ARR_SC1_FLAT, 1024, 2)
        FOLD2_UP(ARR_SC1_FLAT, 1024, 4)

#define FOLD3_GS(ARR, TH, MODE, NB)                                                                                              \
    {                                                                                                                            \
        unsigned nb = (unsigned)((half + (TH)-1) / (TH));                                                                        \
        if (nb > (NB)) nb = (NB);                                                                                                \
        bool keep = c.have_ref; std::vector<uint64_t> keep_sums = c.ref_sums; if (MODE == 3 || MODE == 4) c.have_ref = false;     \
        run_variant(c, "fold3_gs<" #ARR "," #TH ",mode " #MODE "> cap " #NB, nb, ARR == ARR_NONE,                                \
                    [&] { hipLaunchKernelGGL((fold3_gs<ARR, TH, MODE>), dim3(nb), dim3(TH), 0, c.st, c.t, half, c.r, c.out, c.partials, c.sums, c.counter); }); \
        c.have_ref = keep; c.ref_sums = keep_sums;                                                                               \
    }
        FOLD3_GS(ARR_SC1_FLAT, 512, 0, 256)
        FOLD3_GS(ARR_SC1_FLAT, 512, 1, 256)
        FOLD3_GS(ARR_SC1_FLAT, 512, 2, 256)
        FOLD3_GS(ARR_SC1_FLAT, 512, 3, 256)
        FOLD3_GS(ARR_SC1_FLAT, 512, 4, 256)
        FOLD3_GS(ARR_NONE, 512, 0, 256)
        FOLD3_GS(ARR_NONE, 512, 1, 256)
        FOLD3_GS(ARR_NONE, 512, 2, 256)
        FOLD3_GS(ARR_NONE, 512, 3, 256)
        FOLD3_GS(ARR_NONE, 512, 4, 256)
        printf("== sums, 2^%d entries, %.1f MB\n", lg, c.n * 32 / 1e6);
        c.have_ref = false;
#define SUMS_GS(ARR, TH, NB)                                                                                                     \
    {                                                                                                                            \
        unsigned nb = (unsigned)((half + (TH)-1) / (TH));                                                                        \
        if (nb > (NB)) nb = (NB);                                                                                                \
        run_variant(c, "sums_gs<" #ARR "," #TH "> cap " #NB, nb, ARR == ARR_NONE,                                                \
                    [&] { hipLaunchKernelGGL((sums_gs<ARR, TH>), dim3(nb), dim3(TH), 0, c.st, c.t, half, c.partials, c.sums, c.counter); }); \
    }
#define SUMS_UP(ARR, TH, U, NB)                                                                                                  \
    {                                                                                                                            \
        unsigned nb = (unsigned)((half + (TH) * (U)-1) / ((TH) * (U)));                                                          \
        if (nb > (NB)) nb = (NB);                                                                                                \
        run_variant(c, "sums_up<" #ARR "," #TH "," #U "> cap " #NB, nb, ARR == ARR_NONE,                                         \
                    [&] { hipLaunchKernelGGL((sums_up<ARR, TH, U>), dim3(nb), dim3(TH), 0, c.st, c.t, half, c.partials, c.sums, c.counter); }); \
    }
        SUMS_GS(ARR_ACQREL, 256, 256)
        SUMS_GS(ARR_SC1_FLAT, 256, 256)
        SUMS_GS(ARR_SC1_FLAT, 1024, 256)
        SUMS_UP(ARR_SC1_FLAT, 256, 4, 256)
        SUMS_UP(ARR_SC1_FLAT, 256, 8, 256)
        SUMS_UP(ARR_SC1_FLAT, 1024, 2, 256)
        SUMS_UP(ARR_SC1_FLAT, 1024, 4, 256)
        SUMS_UP(ARR_SC1_2LVL_ACQ, 256, 4, 512)
        SUMS_UP(ARR_SC1_2LVL_ACQ, 256, 2, 1024)
        SUMS_UP(ARR_SC1_2LVL_ACQ, 256, 4, 1024)
        SUMS_UP(ARR_SC1_2LVL_ACQ, 512, 4, 512)
        SUMS_UP(ARR_NONE, 256, 4, 1024)

#define SUMS2_UP(ARR, TH, U, NB)                                                                                                 \
    {                                                                                                                            \
        unsigned nb = (unsigned)((half + (TH) * (U)-1) / ((TH) * (U)));                                                          \
        if (nb > (NB)) nb = (NB);                                                                                                \
        run_variant(c, "sums2_up<" #ARR "," #TH "," #U "> cap " #NB, nb, ARR == ARR_NONE,                                        \
                    [&] { hipLaunchKernelGGL((sums2_up<ARR, TH, U>), dim3(nb), dim3(TH), 0, c.st, c.t, half, c.partials, c.sums, c.counter); }); \
    }
        SUMS2_UP(ARR_SC1_FLAT, 256, 1, 256)
        SUMS2_UP(ARR_SC1_FLAT, 256, 2, 256)
        SUMS2_UP(ARR_SC1_FLAT, 256, 4, 256)
        SUMS2_UP(ARR_SC1_FLAT, 512, 2, 256)
        SUMS2_UP(ARR_SC1_FLAT, 1024, 1, 256)
        SUMS2_UP(ARR_SC1_FLAT, 1024, 2, 256)
        SUMS2_UP(ARR_SC1_FLAT, 1024, 4, 256)
        SUMS2_UP(ARR_NONE, 1024, 2, 256)
        SUMS2_UP(ARR_SC1_2LVL_ACQ, 256, 4, 512)
        SUMS2_UP(ARR_SC1_2LVL_ACQ, 256, 2, 1024)
        SUMS2_UP(ARR_SC1_2LVL_ACQ, 256, 4, 1024)
        SUMS2_UP(ARR_SC1_2LVL_ACQ, 512, 2, 512)
        SUMS2_UP(ARR_SC1_FLAT, 256, 4, 1024)
        SUMS2_UP(ARR_NONE, 256, 4, 1024)
        CHK(hipFree(c.t));
        CHK(hipFree(c.out));
        CHK(hipFree(c.partials));
    }
    return 0;
}
