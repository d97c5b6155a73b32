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

namespace zg {

// ------------------------------------------------------------------ eq table
// hi[h] = scale * prod_{j < v_hi} (bit_j(h) ? r[j] : 1 - r[j]),  bit_j = bit (v_hi-1-j) of h  (r[0] <-> MSB)
__global__ void __launch_bounds__(256) eq_hi_kernel(const uint64_t *r, int v_hi, const uint64_t *scale, uint64_t *hi) {
    uint32_t h = blockIdx.x * 256 + threadIdx.x;
    if (h >= (1u << v_hi)) return;
    Fr prod = scale ? fe_load<FrParams>(scale) : Fr::one();
    Fr one = Fr::one();
    for (int j = 0; j < v_hi; j++) {
        Fr rj = fe_load<FrParams>(r + 4 * j);
        Fr f = ((h >> (v_hi - 1 - j)) & 1u) ? rj : fe_sub(one, rj);
        prod = fe_mul(prod, f);
    }
    fe_store(hi + 4 * (size_t)h, prod);
}

// out[(h << v_lo) | lo] = hi[h] * prod_{j < v_lo} (bit_j(lo) ? r_lo[j] : 1 - r_lo[j]); one 32-byte
// store per thread, 8 KiB contiguous per (block, h): the kernel is a pure HBM write stream.
__global__ void __launch_bounds__(256) eq_main_kernel(const uint64_t *r_lo, int v_lo, const uint64_t *hi, uint32_t n_hi,
                                                      uint32_t hi_per_block, uint64_t *out) {
    uint32_t lo = threadIdx.x;
    if (lo >= (1u << v_lo)) return;
    Fr t = Fr::one(), one = Fr::one();
    for (int j = 0; j < v_lo; j++) {
        Fr rj = fe_load<FrParams>(r_lo + 4 * j);
        Fr f = ((lo >> (v_lo - 1 - j)) & 1u) ? rj : fe_sub(one, rj);
        t = fe_mul(t, f);
    }
    F29 tp = fr29_prescale(t);  // shared factor of this thread's products: pre-scaled once
    uint32_t h0 = blockIdx.x * hi_per_block;
    for (uint32_t k = 0; k < hi_per_block; k++) {
        uint32_t h = h0 + k;
        if (h >= n_hi) break;
        Fr hv = fe_load<FrParams>(hi + 4 * (size_t)h);
        fe_store(out + 4 * (((size_t)h << v_lo) | lo), fr_mul29(hv, tp));
    }
}

// q[j] = t[j + half] - t[j]   (HyperKZG.open's quotient, src/poly/commitment/mod.zig:296-299)
__global__ void __launch_bounds__(256) fr_sub_halves_kernel(const uint64_t *t, size_t half, uint64_t *q) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < half; j += stride)
        fe_store(q + 4 * j, fe_sub(fe_load<FrParams>(t + 4 * (j + half)), fe_load<FrParams>(t + 4 * j)));
}

// f[i] = eq[i] * (Az[i]*Bz[i] - Cz[i])
__global__ void __launch_bounds__(256) spartan_combine_kernel(const uint64_t *eq, const uint64_t *az, const uint64_t *bz,
                                                              const uint64_t *cz, size_t n, uint64_t *out) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        Fr a = fe_load<FrParams>(az + 4 * i), b = fe_load<FrParams>(bz + 4 * i), c = fe_load<FrParams>(cz + 4 * i);
        Fr e = fe_load<FrParams>(eq + 4 * i);
        fe_store(out + 4 * i, fe_mul(e, fe_sub(fe_mul(a, b), c)));
    }
}

// dot product sum_i a[i]*b[i] over Fr: per-block partials, finished by sc_finish_kernel's first slot
__global__ void __launch_bounds__(256) fr_dot_kernel(const uint64_t *a, const uint64_t *b, size_t n, uint64_t *partials);

// ------------------------------------------------------------------ sums / folds
// block-wide sum of (g0, g1) pairs (256 threads); result valid in thread 0. Wave-level shuffle tree first
// (no barriers, no LDS round trips), then one LDS hop across the four waves: the latency of this reduction is
// what a small sumcheck round mostly consists of.
ZG_DEV Fr fr_shfl_down(const Fr &v, int d) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = __shfl_down(v.l[i], d, 64);
    return r;
}
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
        for (uint32_t w = 1; w < 4; w++) {
            g0 = fe_add(g0, fe_load<FrParams>(&sh[w * 4]));
            g1 = fe_add(g1, fe_load<FrParams>(&sh[w * 4 + 2]));
        }
    }
}

__global__ void __launch_bounds__(256) fr_dot_kernel(const uint64_t *a, const uint64_t *b, size_t n, uint64_t *partials) {
    __shared__ uint4 sh[256 * 4];
    Fr g0 = Fr::zero(), g1 = Fr::zero();
    size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        g0 = fe_add(g0, fe_mul(fe_load<FrParams>(a + 4 * i), fe_load<FrParams>(b + 4 * i)));
    block_sum_pair(g0, g1, sh);
    if (threadIdx.x == 0) {
        fe_store(partials + 8 * (size_t)blockIdx.x, g0);
        fe_store(partials + 8 * (size_t)blockIdx.x + 4, g1);
    }
}

// round sums of a table: HIGH: g0 = sum t[0..h), g1 = sum t[h..2h);  LOW: g0 = sum t[2i], g1 = sum t[2i+1]
// A kernel that produces the final pair of a round publishes it to the pinned host mailbox: sums first, then
// (after a system-scope fence) the round's sequence number, so the host can spin on the mailbox instead of
// paying a stream synchronisation per round. flag == nullptr: this launch only writes block partials.
ZG_DEV void publish_seq(uint64_t *flag, uint64_t seq) {
    if (flag) {
        __threadfence_system();
        __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <int LAYOUT>
__global__ void __launch_bounds__(256) sc_sums_kernel(const uint64_t *t, size_t half, uint64_t *partials, uint64_t *flag, uint64_t seq) {
    __shared__ uint4 sh[256 * 4];
    Fr g0 = Fr::zero(), g1 = Fr::zero();
    size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < half; i += stride) {
        size_t i0 = LAYOUT == ZG_SC_HIGH_HALF ? i : 2 * i, i1 = LAYOUT == ZG_SC_HIGH_HALF ? i + half : 2 * i + 1;
        g0 = fe_add(g0, fe_load<FrParams>(t + 4 * i0));
        g1 = fe_add(g1, fe_load<FrParams>(t + 4 * i1));
    }
    block_sum_pair(g0, g1, sh);
    if (threadIdx.x == 0) {
        fe_store(partials + 8 * (size_t)blockIdx.x, g0);
        fe_store(partials + 8 * (size_t)blockIdx.x + 4, g1);
        publish_seq(flag, seq);
    }
}

// fold by r and produce the NEXT round's two sums from the values just written:
//   HIGH: out[i] = (1-r)*t[i] + r*t[i+half]     next g0 over i < half/2, g1 over i >= half/2
//   LOW : out[i] = t[2i] + r*(t[2i+1]-t[2i])    next g0 over even i,     g1 over odd i
struct FrArg {  // a challenge travels as a kernel argument: no H2D copy, no staging buffer to recycle
    uint32_t l[8];
};

template <int LAYOUT>
__global__ void __launch_bounds__(256) sc_fold_kernel(const uint64_t *t, size_t half, FrArg r, uint64_t *out,
                                                      uint64_t *partials, uint64_t *flag, uint64_t seq) {
    __shared__ uint4 sh[256 * 4];
    Fr rv;
#pragma unroll
    for (int i = 0; i < 8; i++) rv.l[i] = r.l[i];
    F29 rp = fr29_prescale(rv);  // the challenge is the shared factor of every product of this launch
    Fr g0 = Fr::zero(), g1 = Fr::zero();
    size_t stride = (size_t)gridDim.x * 256;
    size_t quarter = half / 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < half; i += stride) {
        Fr v;
        if (LAYOUT == ZG_SC_HIGH_HALF) {
            Fr lo = fe_load<FrParams>(t + 4 * i), hi = fe_load<FrParams>(t + 4 * (i + half));
            v = fe_add(lo, fr_mul29(fe_sub(hi, lo), rp));  // (1-r)*lo + r*hi = lo + r*(hi - lo): one product, same value
        } else {
            Fr lo = fe_load<FrParams>(t + 8 * i), hi = fe_load<FrParams>(t + 8 * i + 4);
            v = fe_add(lo, fr_mul29(fe_sub(hi, lo), rp));
        }
        fe_store(out + 4 * i, v);
        bool second = LAYOUT == ZG_SC_HIGH_HALF ? (i >= quarter) : (i & 1);
        if (second) g1 = fe_add(g1, v);
        else g0 = fe_add(g0, v);
    }
    block_sum_pair(g0, g1, sh);
    if (threadIdx.x == 0) {
        fe_store(partials + 8 * (size_t)blockIdx.x, g0);
        fe_store(partials + 8 * (size_t)blockIdx.x + 4, g1);
        publish_seq(flag, seq);
    }
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
        fe_store(sums, g0);
        fe_store(sums + 4, g1);
        publish_seq(flag, seq);
    }
}

static unsigned sc_blocks(size_t half) {
    unsigned b = div_up(half ? half : 1, 256);
    return b > 2048 ? 2048 : b;  // grid-stride beyond 8 blocks per CU
}

static int launch_sums(int layout, const uint64_t *t, size_t len, uint64_t *partials, uint64_t *sums, hipStream_t st,
                       uint64_t *flag = nullptr, uint64_t seq = 0) {
    size_t half = len / 2;
    unsigned nb = sc_blocks(half);
    prof_begin(ZG_PROF_SC_SUMS, st);
    uint64_t *dst = nb == 1 ? sums : partials;  // one block: its pair IS the result
    if (layout == ZG_SC_HIGH_HALF)
        hipLaunchKernelGGL(sc_sums_kernel<ZG_SC_HIGH_HALF>, dim3(nb), dim3(256), 0, st, t, half, dst, nb == 1 ? flag : nullptr, seq);
    else
        hipLaunchKernelGGL(sc_sums_kernel<ZG_SC_LOW_PAIR>, dim3(nb), dim3(256), 0, st, t, half, dst, nb == 1 ? flag : nullptr, seq);
    if (nb > 1) hipLaunchKernelGGL(sc_finish_kernel, dim3(1), dim3(256), 0, st, partials, nb, sums, flag, seq);
    prof_end(ZG_PROF_SC_SUMS, st);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

static int launch_fold(int layout, const uint64_t *t, size_t len, const uint64_t r[4], uint64_t *out, uint64_t *partials,
                       uint64_t *sums, hipStream_t st, uint64_t *flag = nullptr, uint64_t seq = 0) {
    size_t half = len / 2;
    unsigned nb = sc_blocks(half);
    FrArg ra;
    for (int i = 0; i < 4; i++) {
        ra.l[2 * i] = (uint32_t)r[i];
        ra.l[2 * i + 1] = (uint32_t)(r[i] >> 32);
    }
    uint64_t *dst = nb == 1 ? sums : partials;
    prof_begin(ZG_PROF_SC_FOLD, st);
    if (layout == ZG_SC_HIGH_HALF)
        hipLaunchKernelGGL(sc_fold_kernel<ZG_SC_HIGH_HALF>, dim3(nb), dim3(256), 0, st, t, half, ra, out, dst, nb == 1 ? flag : nullptr, seq);
    else
        hipLaunchKernelGGL(sc_fold_kernel<ZG_SC_LOW_PAIR>, dim3(nb), dim3(256), 0, st, t, half, ra, out, dst, nb == 1 ? flag : nullptr, seq);
    if (nb > 1) hipLaunchKernelGGL(sc_finish_kernel, dim3(1), dim3(256), 0, st, partials, nb, sums, flag, seq);
    prof_end(ZG_PROF_SC_FOLD, st);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

static int eq_table_enqueue(const uint64_t *r_host, size_t v, const uint64_t *scale_host, uint64_t *d_out, hipStream_t st) {
    if (v > 34) {
        set_error("zg_fr_eq_table: v too large");
        return ZG_ERR_INVALID;
    }
    int v_lo = v < 8 ? (int)v : 8, v_hi = (int)v - v_lo;
    uint32_t n_hi = 1u << v_hi;
    Scratch s_r((v + 1) * 32 + 32), s_hi((size_t)n_hi * 32);
    if (!s_r.p || !s_hi.p) return ZG_ERR_NOMEM;
    uint64_t *d_r = s_r.as<uint64_t>(), *d_hi = s_hi.as<uint64_t>();
    if (v) ZG_HIP(hipMemcpyAsync(d_r + 4, r_host, v * 32, hipMemcpyHostToDevice, st));
    if (scale_host) ZG_HIP(hipMemcpyAsync(d_r, scale_host, 32, hipMemcpyHostToDevice, st));
    prof_begin(ZG_PROF_EQ_TABLE, st);
    hipLaunchKernelGGL(eq_hi_kernel, dim3(div_up(n_hi, 256)), dim3(256), 0, st, d_r + 4, v_hi, scale_host ? d_r : nullptr, d_hi);
    uint32_t hpb = n_hi / 512 ? n_hi / 512 : 1;  // rows per block: amortises the 8-mul low-table product of each thread
    hipLaunchKernelGGL(eq_main_kernel, dim3(div_up(n_hi, hpb)), dim3(256), 0, st, d_r + 4 + 4 * (size_t)v_hi, v_lo, d_hi, n_hi, hpb,
                       d_out);
    prof_end(ZG_PROF_EQ_TABLE, st);
    ZG_HIP(hipGetLastError());
    ZG_HIP(hipStreamSynchronize(st));  // r_host / temporaries are released on return
    return ZG_OK;
}

}  // namespace zg

struct zg_sc_s {
    int layout = 0;
    size_t len = 0;
    uint64_t *buf[2] = {nullptr, nullptr};  // ping-pong tables (a fold cannot run in place across threads)
    int cur = 0;
    uint64_t *d_partials = nullptr;
    uint64_t *h_pin = nullptr;  // pinned, device-visible: the kernels write the round sums (8 limbs) straight to the host
    bool sums_valid = false;
    hipStream_t st = nullptr;
    uint64_t seq = 0;  // number of (sums) publications requested so far; h_pin[12] holds the last one completed
    size_t cap = 0;  // elements buf[0] can hold (sessions are pooled: hipMalloc/hipFree cost more than a round)
    std::mutex mu;
};

using namespace zg;

static void sc_free(zg_sc_s *s) {
    if (!s) return;
    void *ptrs[] = {s->buf[0], s->buf[1], s->d_partials};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (s->h_pin) (void)hipHostFree(s->h_pin);
    delete s;
}

static std::mutex g_pool_mu;
static std::vector<zg_sc_s *> g_pool;  // closed sessions kept for reuse (at most 4)

static int sc_create(size_t len, int layout, hipStream_t st, zg_sc_s **out) {
    if (len == 0 || (len & (len - 1)) || (layout != ZG_SC_HIGH_HALF && layout != ZG_SC_LOW_PAIR)) {
        set_error("zg_sumcheck_open: len must be a power of two and layout valid");
        return ZG_ERR_INVALID;
    }
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (size_t i = 0; i < g_pool.size(); i++) {
            if (g_pool[i]->cap >= len && g_pool[i]->cap <= 4 * len) {
                zg_sc_s *s = g_pool[i];
                g_pool.erase(g_pool.begin() + i);
                s->layout = layout; s->len = len; s->st = st; s->cur = 0; s->sums_valid = false;
                s->seq = 0; s->h_pin[12] = 0;
                *out = s;
                return ZG_OK;
            }
        }
    }
    zg_sc_s *s = new zg_sc_s();
    s->cap = len;
    s->layout = layout;
    s->len = len;
    s->st = st;
    hipError_t e = hipMalloc((void **)&s->buf[0], len * 32);
    if (e == hipSuccess) e = hipMalloc((void **)&s->buf[1], (len / 2 ? len / 2 : 1) * 32);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_partials, 2048 * 64);
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
    int rc = eq_table_enqueue(r, v, scale, d_out, lib_stream());
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
    Scratch s_t(len * 32), s_o(len / 2 * 32), s_misc(2048 * 64 + 64);
    if (!s_t.p || !s_o.p || !s_misc.p) return ZG_ERR_NOMEM;
    uint64_t *d_t = s_t.as<uint64_t>(), *d_o = s_o.as<uint64_t>(), *d_misc = s_misc.as<uint64_t>();
    uint64_t *d_sums = d_misc + 2048 * 8;
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
    Scratch s_ev(n * 32), s_eq(n * 32), s_misc(2048 * 64 + 64);
    if (!s_ev.p || !s_eq.p || !s_misc.p) return ZG_ERR_NOMEM;
    uint64_t *d_ev = s_ev.as<uint64_t>(), *d_eq = s_eq.as<uint64_t>(), *d_misc = s_misc.as<uint64_t>();
    ZG_HIP(hipMemcpyAsync(d_ev, evals, n * 32, hipMemcpyHostToDevice, st));
    int rc = eq_table_enqueue(rev.data(), num_vars, nullptr, d_eq, st);
    if (rc == ZG_OK) {
        unsigned nb = sc_blocks(n);
        hipLaunchKernelGGL(fr_dot_kernel, dim3(nb), dim3(256), 0, st, d_ev, d_eq, n, d_misc);
        hipLaunchKernelGGL(sc_finish_kernel, dim3(1), dim3(256), 0, st, d_misc, nb, d_misc + 2048 * 8, (uint64_t *)nullptr, (uint64_t)0);
        uint64_t h[4];
        hipError_t e = hipMemcpyAsync(h, d_misc + 2048 * 8, 32, hipMemcpyDeviceToHost, st);
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
    hipStream_t st = lib_stream();
    size_t srs_len = zg_g1_bases_len(srs);
    size_t cap = n_evals ? n_evals : 1;
    std::vector<uint64_t> h_res(9 * num_vars + 4, 0);
    Scratch s_a(cap * 32), s_b((cap / 2 + 1) * 32), s_q((cap / 2 + 1) * 32), s_res((9 * num_vars + 4) * 8), s_misc(2048 * 64 + 64);
    if (!s_a.p || !s_b.p || !s_q.p || !s_res.p || !s_misc.p) return ZG_ERR_NOMEM;
    uint64_t *d_a = s_a.as<uint64_t>(), *d_b = s_b.as<uint64_t>(), *d_q = s_q.as<uint64_t>(), *d_res = s_res.as<uint64_t>(),
             *d_misc = s_misc.as<uint64_t>();
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = hipMemsetAsync(d_res, 0, (9 * num_vars + 4) * 8, st);
    if (e == hipSuccess && n_evals) e = hipMemcpyAsync(d_a, evals, n_evals * 32, hipMemcpyHostToDevice, st);
    int rc = ZG_OK;
    size_t len = n_evals;
    uint64_t *cur = d_a, *nxt = d_b;
    for (size_t i = 0; i < num_vars && e == hipSuccess && rc == ZG_OK; i++) {
        size_t half = len / 2;
        if (half == 0) {  // :289: the reference stops folding; remaining quotients stay unset -> identity here
            for (size_t r = i; r < num_vars; r++) h_res[9 * r + 8] = 0x100;  // marker: identity
            break;
        }
        unsigned nb = div_up(half, 256);
        if (nb > 4096) nb = 4096;
        hipLaunchKernelGGL(fr_sub_halves_kernel, dim3(nb), dim3(256), 0, st, cur, half, d_q);
        size_t nc = half < srs_len ? half : srs_len;  // commit(): n = min(evals.len, srs.len), :246
        rc = zg_msm_g1_dev_async(srs, 0, nc, d_q, st, d_res + 9 * i, reinterpret_cast<uint8_t *>(d_res + 9 * i + 8));
        if (rc != ZG_OK) break;
        rc = launch_fold(ZG_SC_HIGH_HALF, cur, 2 * half, point + 4 * i, nxt, d_misc, d_misc + 2048 * 8, st);
        uint64_t *t = cur; cur = nxt; nxt = t;
        len = half;
    }
    if (e == hipSuccess && rc == ZG_OK && len > 0)
        e = hipMemcpyAsync(d_res + 9 * num_vars, cur, 32, hipMemcpyDeviceToDevice, st);  // final = current[0], :317
    std::vector<uint64_t> dev_res(9 * num_vars + 4);
    if (e == hipSuccess && rc == ZG_OK) e = hipMemcpyAsync(dev_res.data(), d_res, (9 * num_vars + 4) * 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    else (void)hipStreamSynchronize(st);
    if (e != hipSuccess) {
        set_error(std::string("zg_hyperkzg_open: ") + hipGetErrorString(e));
        return ZG_ERR_HIP;
    }
    if (rc != ZG_OK) return rc;
    for (size_t i = 0; i < num_vars; i++) {
        bool skipped = h_res[9 * i + 8] == 0x100;
        for (int j = 0; j < 8; j++) q_xy[8 * i + j] = skipped ? 0 : dev_res[9 * i + j];
        if (q_inf) q_inf[i] = skipped ? 1 : (uint8_t)(dev_res[9 * i + 8] & 0xff);
    }
    for (int j = 0; j < 4; j++) final_eval[j] = len > 0 ? dev_res[9 * num_vars + j] : 0;
    return ZG_OK;
}

// ---------------------------------------------------------------- sumcheck session
int zg_sumcheck_open(const uint64_t *evals, size_t len, int layout, zg_sc_t *out) {
    ZG_INIT();
    if (!evals || !out) {
        set_error("zg_sumcheck_open: invalid argument");
        return ZG_ERR_INVALID;
    }
    zg_sc_s *s = nullptr;
    ZG_TRY(sc_create(len, layout, lib_stream(), &s));
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

int zg_sumcheck_round_sums(zg_sc_t s, uint64_t g0[4], uint64_t g1[4]) {
    ZG_INIT();
    if (!s || !g0 || !g1 || s->len < 2) {
        set_error("zg_sumcheck_round_sums: invalid session or protocol already complete");
        return ZG_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lk(s->mu);
    if (!s->sums_valid) {
        s->seq++;
        ZG_TRY(launch_sums(s->layout, s->buf[s->cur], s->len, s->d_partials, s->h_pin, s->st, s->h_pin + 12, s->seq));
        s->sums_valid = true;
    }
    // the only host<->device rendezvous of a round: spin on the mailbox's sequence word (written by the GPU after
    // the sums, system-scope release); fall back to a stream synchronisation if it does not arrive promptly
    {
        volatile uint64_t *flag = s->h_pin + 12;
        bool got = false;
        for (uint64_t spin = 0; spin < (1ull << 22); spin++) {
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == s->seq) { got = true; break; }
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
        if (!got) ZG_HIP(hipStreamSynchronize(s->st));
    }
    for (int i = 0; i < 4; i++) {
        g0[i] = s->h_pin[i];
        g1[i] = s->h_pin[4 + i];
    }
    return ZG_OK;
}

int zg_sumcheck_bind(zg_sc_t s, const uint64_t r[4]) {
    ZG_INIT();
    if (!s || !r || s->len < 2) {
        set_error("zg_sumcheck_bind: invalid session or protocol already complete");
        return ZG_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lk(s->mu);
    // buf[1] holds len/2 elements at most; after the first fold both buffers are large enough
    int nxt = s->cur ^ 1;
    s->seq++;
    ZG_TRY(launch_fold(s->layout, s->buf[s->cur], s->len, r, s->buf[nxt], s->d_partials, s->h_pin, s->st, s->h_pin + 12,
                       s->seq));  // asynchronous
    s->cur = nxt;
    s->len /= 2;
    s->sums_valid = s->len >= 2;
    return ZG_OK;
}

size_t zg_sumcheck_len(zg_sc_t s) { return s ? s->len : 0; }

int zg_sumcheck_final(zg_sc_t s, uint64_t out[4]) {
    ZG_INIT();
    if (!s || !out || s->len != 1) {
        set_error("zg_sumcheck_final: protocol not complete");
        return ZG_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lk(s->mu);
    ZG_HIP(hipMemcpyAsync(s->h_pin + 8, s->buf[s->cur], 32, hipMemcpyDeviceToHost, s->st));
    ZG_HIP(hipStreamSynchronize(s->st));
    for (int i = 0; i < 4; i++) out[i] = s->h_pin[8 + i];
    return ZG_OK;
}

int zg_sumcheck_read(zg_sc_t s, uint64_t *out_table) {
    ZG_INIT();
    if (!s || !out_table) {
        set_error("zg_sumcheck_read: invalid argument");
        return ZG_ERR_INVALID;
    }
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
    std::lock_guard<std::mutex> lk(s->mu);
    ZG_HIP(hipMemcpyAsync(d_out_table, s->buf[s->cur], s->len * 32, hipMemcpyDeviceToDevice, s->st));
    return ZG_OK;
}

int zg_sumcheck_close(zg_sc_t s) {
    if (!s) return ZG_OK;
    ZG_INIT();
    (void)hipStreamSynchronize(s->st);
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (g_pool.size() < 4) {
            g_pool.push_back(s);
            return ZG_OK;
        }
    }
    sc_free(s);
    return ZG_OK;
}

}  // extern "C"
