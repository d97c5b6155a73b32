// microbench.hip — gfx950 integer-ALU issue-rate probes for the MSM design
// (SURVEY §7.1 "micro-benchmarks to run first"): cycles per wave-instruction for the
// candidate multiply/add instructions at 1, 2, 4 and 8 waves per SIMD, and the
// throughput of the Montgomery multiply as compiled.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench.hip -o tools/microbench
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../zolt_amd/csrc/field.hip.h"
#include "../zolt_amd/csrc/g1.hip.h"
#include "../zolt_amd/csrc/g1_29.hip.h"

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

// each kernel: ITER iterations of 64 instructions spread over 8 independent register chains
#define DEFK(name, ASM8, DECL, OUTEXPR)                                                           \
    __global__ void name(unsigned long long *cyc, unsigned *sink, int iters, unsigned seed) {      \
        DECL;                                                                                      \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                      \
        for (int it = 0; it < iters; it++) { REP8(ASM8) }                                          \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                      \
        if (threadIdx.x % 64 == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;   \
        sink[blockIdx.x * blockDim.x + threadIdx.x] = OUTEXPR;                                     \
    }

#define DECL_U32                                                                                   \
    unsigned a0 = seed + threadIdx.x, a1 = a0 * 3u + 1, a2 = a0 * 5u + 2, a3 = a0 * 7u + 3, a4 = a0 * 11u, a5 = a0 * 13u, \
             a6 = a0 * 17u, a7 = a0 * 19u, b = a0 | 1u, c = a0 ^ 0x9e3779b9u
#define OUT_U32 (a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)
#define DECL_U64                                                                                   \
    unsigned long long a0 = seed + threadIdx.x, a1 = a0 * 3u + 1, a2 = a0 * 5u + 2, a3 = a0 * 7u + 3, a4 = a0 * 11u,       \
                       a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;                                \
    unsigned b = (unsigned)a0 | 1u, c = (unsigned)a0 ^ 0x9e3779b9u
#define OUT_U64 ((unsigned)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) ^ (unsigned)((a0 ^ a1 ^ a2 ^ a3) >> 32))

#define A1(op, r) asm volatile(op " %0, %0, %1" : "+v"(r) : "v"(b));
#define ASM8_2OP(op) A1(op, a0) A1(op, a1) A1(op, a2) A1(op, a3) A1(op, a4) A1(op, a5) A1(op, a6) A1(op, a7)
#define A3(op, r) asm volatile(op " %0, %1, %2, %0" : "+v"(r) : "v"(b), "v"(c));
#define ASM8_3OP(op) A3(op, a0) A3(op, a1) A3(op, a2) A3(op, a3) A3(op, a4) A3(op, a5) A3(op, a6) A3(op, a7)
#define AM(r) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r) : "v"(b), "v"(c) : "vcc");
#define ASM8_MAD64 AM(a0) AM(a1) AM(a2) AM(a3) AM(a4) AM(a5) AM(a6) AM(a7)
#define AL(r) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(r) : "v"(a7));
#define ASM8_LSHLADD64 AL(a0) AL(a1) AL(a2) AL(a3) AL(a4) AL(a5) AL(a6) AL(a0)
#define AC(r) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(r) : "v"(b) : "vcc");
#define ASM8_ADDC AC(a0) AC(a1) AC(a2) AC(a3)   /* 8 instructions: 4 x (add_co + addc_co) */
#define AMV(r) asm volatile("v_mov_b32 %0, %1" : "+v"(r) : "v"(b));
#define ASM8_MOV AMV(a0) AMV(a1) AMV(a2) AMV(a3) AMV(a4) AMV(a5) AMV(a6) AMV(a7)

DEFK(k_add_u32, ASM8_2OP("v_add_u32"), DECL_U32, OUT_U32)
DEFK(k_mov_b32, ASM8_MOV, DECL_U32, OUT_U32)
DEFK(k_mul_lo_u32, ASM8_2OP("v_mul_lo_u32"), DECL_U32, OUT_U32)
DEFK(k_mul_hi_u32, ASM8_2OP("v_mul_hi_u32"), DECL_U32, OUT_U32)
DEFK(k_mul_u32_u24, ASM8_2OP("v_mul_u32_u24"), DECL_U32, OUT_U32)
DEFK(k_mad_u32_u24, ASM8_3OP("v_mad_u32_u24"), DECL_U32, OUT_U32)
DEFK(k_add3_u32, ASM8_3OP("v_add3_u32"), DECL_U32, OUT_U32)
DEFK(k_addc_pair, ASM8_ADDC, DECL_U32, OUT_U32)
DEFK(k_mad_u64_u32, ASM8_MAD64, DECL_U64, OUT_U64)
DEFK(k_lshl_add_u64, ASM8_LSHLADD64, DECL_U64, OUT_U64)

__global__ void k_fma_f64(unsigned long long *cyc, unsigned *sink, int iters, unsigned seed) {
    double a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    double b = 1.0000001, c = 1e-9;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#define AF(r) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c));
        REP8(AF(a0) AF(a1) AF(a2) AF(a3) AF(a4) AF(a5) AF(a6) AF(a7))
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x % 64 == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}

// Montgomery multiply as compiled from field.hip.h: a dependent chain of fe_mul
__global__ void k_fe_mul(unsigned long long *cyc, unsigned *sink, int iters, unsigned seed) {
    zg::Fp x, y;
    for (int i = 0; i < 8; i++) { x.l[i] = (seed + threadIdx.x) * (i + 3); y.l[i] = (seed ^ threadIdx.x) * (i + 7); }
    x.l[7] &= 0x0fffffff; y.l[7] &= 0x0fffffff;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) { x = zg::fe_mul(x, y); y = zg::fe_mul(y, x); }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x % 64 == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = x.l[0] ^ y.l[3];
}
// XYZZ mixed add chain
__global__ void k_madd(unsigned long long *cyc, unsigned *sink, int iters, unsigned seed) {
    zg::XYZZ acc; zg::Affine p;
    for (int i = 0; i < 8; i++) {
        acc.x.l[i] = (seed + threadIdx.x) * (i + 3); acc.y.l[i] = (seed ^ threadIdx.x) * (i + 7);
        acc.zz.l[i] = (seed + 2 * threadIdx.x) * (i + 5); acc.zzz.l[i] = (seed + 3 * threadIdx.x) * (i + 9);
        p.x.l[i] = (seed + 5 * threadIdx.x) * (i + 1); p.y.l[i] = (seed + 7 * threadIdx.x) * (i + 2);
    }
    acc.x.l[7] &= 0x0fffffff; acc.y.l[7] &= 0x0fffffff; acc.zz.l[7] &= 0x0fffffff; acc.zzz.l[7] &= 0x0fffffff;
    p.x.l[7] &= 0x0fffffff; p.y.l[7] &= 0x0fffffff;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) { acc = zg::xyzz_madd(acc, p); p.x.l[0] ^= acc.x.l[1]; }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x % 64 == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc.x.l[0] ^ acc.zzz.l[3];
}

// ---- the numbers behind DESIGN.md's batched-affine analysis: what a field inversion costs a wave (all 64 lanes invert their
// own value: SIMD executes 64 inversions for the price of one), the lazy-limb XYZZ mixed addition the accumulate kernel runs,
// and the arithmetic of one batched-affine addition (2M + 1S for the group law + 3M for Montgomery's trick) WITHOUT its share of
// the inversion
__global__ void k_inv_safegcd(unsigned long long *cyc, unsigned *sink, int iters, unsigned seed) {
    zg::Fp x;
    for (int i = 0; i < 8; i++) x.l[i] = (seed + threadIdx.x + blockIdx.x * 977u) * (i + 3) + 12345u;
    x.l[7] &= 0x0fffffff;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) { x = zg::fe_inv_safegcd(x); x.l[0] ^= (unsigned)it * 2654435761u; x.l[7] &= 0x0fffffff; }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x % 64 == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = x.l[0] ^ x.l[5];
}
__global__ void k_madd29(unsigned long long *cyc, unsigned *sink, int iters, unsigned seed) {
    zg::XYZZ29 acc; zg::F29 px, py;
    for (int i = 0; i < 9; i++) {
        unsigned b = (seed + threadIdx.x) * (i + 3);
        acc.x.l[i] = b & 0x1fffffff; acc.y.l[i] = (b * 7) & 0x1fffffff; acc.zz.l[i] = (b * 11) & 0x1fffffff; acc.zzz.l[i] = (b * 13) & 0x1fffffff;
        px.l[i] = (b * 17) & 0x1fffffff; py.l[i] = (b * 19) & 0x1fffffff;
    }
    acc.x.l[8] &= 0x3fffff; acc.y.l[8] &= 0x3fffff; acc.zz.l[8] &= 0x3fffff; acc.zzz.l[8] &= 0x3fffff; px.l[8] &= 0x3fffff; py.l[8] &= 0x3fffff;
    bool acc_inf = false;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) { zg::xyzz29_madd(acc, acc_inf, px, py); px.l[0] ^= acc.x.l[1] & 0xff; }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x % 64 == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc.x.l[0] ^ acc.zzz.l[3];
}
__global__ void k_affine_add_body29(unsigned long long *cyc, unsigned *sink, int iters, unsigned seed) {
    zg::F29 x1, y1, x2, y2, pre, inv;
    for (int i = 0; i < 9; i++) {
        unsigned b = (seed + threadIdx.x) * (i + 3);
        x1.l[i] = b & 0x1fffffff; y1.l[i] = (b * 7) & 0x1fffffff; x2.l[i] = (b * 11) & 0x1fffffff; y2.l[i] = (b * 13) & 0x1fffffff;
        pre.l[i] = (b * 17) & 0x1fffffff; inv.l[i] = (b * 19) & 0x1fffffff;
    }
    x1.l[8] &= 0x3fffff; y1.l[8] &= 0x3fffff; x2.l[8] &= 0x3fffff; y2.l[8] &= 0x3fffff; pre.l[8] &= 0x3fffff; inv.l[8] &= 0x3fffff;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        // Montgomery's trick around the pair: prefix product forward (1M), then backward inv_d = inv * prefix (1M), inv *= d (1M)
        zg::F29 d = zg::f29_sub2(x2, x1);
        zg::F29 prefix = zg::f29_mul(pre, d);
        zg::F29 inv_d = zg::f29_mul(inv, pre);
        inv = zg::f29_mul(inv, d);
        // the affine group law: lambda = (y2 - y1) / (x2 - x1), x3 = lambda^2 - x1 - x2, y3 = lambda (x1 - x3) - y1  (2M + 1S)
        zg::F29 lam = zg::f29_mul(zg::f29_sub2(y2, y1), inv_d);
        zg::F29 x3 = zg::f29_sub4(zg::f29_sub2(zg::f29_sqr(lam), x1), x2);
        zg::F29 y3 = zg::f29_sub2(zg::f29_mul(lam, zg::f29_sub4(x1, x3)), y1);
        x1 = zg::f29_carry(x3); y1 = zg::f29_carry(y3); pre = prefix;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x % 64 == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = x1.l[0] ^ y1.l[3] ^ inv.l[2];
}

typedef void (*kern_t)(unsigned long long *, unsigned *, int, unsigned);

static void run(const char *name, kern_t k, int instr_per_iter, int iters, int ncu) {
    unsigned long long *cyc; unsigned *sink;
    int maxwaves = ncu * 32;
    CHK(hipMalloc(&cyc, maxwaves * sizeof(unsigned long long)));
    CHK(hipMalloc(&sink, (size_t)maxwaves * 64 * sizeof(unsigned)));
    printf("%-16s", name);
    for (int wps = 1; wps <= 8; wps *= 2) {  // waves per SIMD
        int threads = 256;                    // 4 waves = 1 per SIMD
        int blocks = ncu * wps;
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, cyc, sink, 10, 1u);  // warm
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, cyc, sink, iters, 1u);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(blocks * 4);
        CHK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
        double avg = 0; for (auto v : h) avg += (double)v; avg /= h.size();
        double per_wave_instr = avg / ((double)iters * instr_per_iter);     // memtime ticks per wave-instruction (latency view)
        double total_instr = (double)blocks * 4 * iters * instr_per_iter;    // wave-instructions
        double ns_per_simd_instr = ms * 1e6 / (total_instr / (ncu * 4.0));   // wall ns per instruction per SIMD
        printf(" | w/simd=%d: %7.2f tick/winstr %7.3f ns/instr/SIMD", wps, per_wave_instr, ns_per_simd_instr);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    printf("\n");
    (void)hipFree(cyc); (void)hipFree(sink);
}

// ---- HBM access-pattern calibration for the PMC figures (MI355X_MICROARCH.md: "calibrate on a known byte count in your own access
// pattern"): `microbench gather` reads N random 64-byte rows of a 2 GiB table with the accumulate kernel's load shape (one row per
// lane: four 16-byte loads of consecutive addresses), `microbench stream` reads the same number of bytes sequentially. Run each
// under rocprofv3 --pmc FETCH_SIZE: FETCH_SIZE / known bytes is the correction factor of that pattern; the printed rows/s also
// bounds the request width (rows/s x 64 B vs x 128 B against the 8 TB/s peak).
__global__ void k_gather64(const uint4 *table, size_t n_rows_table, size_t rows_per_thread, unsigned *sink) {
    size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long h = tid * 0x9E3779B97F4A7C15ull + 0x1234567ull;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t k = 0; k < rows_per_thread; k++) {
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        size_t row = (size_t)(h % n_rows_table);
        const uint4 *p = table + 4 * row;
        uint4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc.x ^= a.x ^ b.y ^ c.z ^ d.w; acc.y += a.y + d.x;
    }
    sink[tid] = acc.x ^ acc.y;
}
__global__ void k_stream64(const uint4 *table, size_t n_rows_table, size_t rows_per_thread, unsigned *sink) {
    size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (size_t)gridDim.x * blockDim.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t k = 0; k < 4 * rows_per_thread; k++) {  // 16 B per lane per load, consecutive lanes on consecutive addresses
        size_t i = (k * nthreads + tid) % (4 * n_rows_table);
        uint4 a = table[i];
        acc.x ^= a.x; acc.y += a.w;
    }
    sink[tid] = acc.x ^ acc.y;
}
// ---- XCD-aware block mapping for a pure stream: workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8), each XCD has its own
// L2. `microbench xcd` streams the same 2 GiB three ways: (a) grid-stride (every wave-load of consecutive blocks is adjacent, all XCDs
// touch every region), (b) one contiguous chunk per block, (c) chunks grouped so that the blocks of one XCD own one contiguous eighth.
__global__ void k_stream_map(const uint4 *table, size_t n16, int mapping, unsigned *sink) {
    const size_t nb = gridDim.x, per_block = n16 / nb;
    size_t b = blockIdx.x;
    if (mapping == 2) b = (b % 8) * (nb / 8) + b / 8;  // XCD x = blockIdx % 8 owns blocks' chunks [x * nb/8, (x+1) * nb/8)
    uint4 acc = make_uint4(0, 0, 0, 0);
    if (mapping == 0) {
        const size_t nthreads = nb * blockDim.x, tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
        for (size_t i = tid; i < n16; i += nthreads) {
            uint4 a = table[i];
            acc.x ^= a.x; acc.y += a.w;
        }
    } else if (mapping == 3 || mapping == 4) {  // each lane owns 64 (or 32) contiguous bytes, neighbours 64 (32) bytes apart: the fold kernels' shape
        const size_t per = mapping == 3 ? 4 : 2;
        const size_t nthreads = nb * blockDim.x, tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
        for (size_t i = tid; i < n16 / per; i += nthreads) {
            for (size_t q = 0; q < per; q++) {
                uint4 a = table[per * i + q];
                acc.x ^= a.x; acc.y += a.w;
            }
        }
    } else {
        const uint4 *p = table + b * per_block;
        for (size_t i = threadIdx.x; i < per_block; i += blockDim.x) {
            uint4 a = p[i];
            acc.x ^= a.x; acc.y += a.w;
        }
    }
    sink[blockIdx.x * (size_t)blockDim.x + threadIdx.x] = acc.x ^ acc.y;
}
// the LowToHigh fold's traffic without its arithmetic: 64 contiguous bytes read and 32 written per lane, grid-strided
__global__ void k_fold_traffic(const uint4 *in, uint4 *out, size_t n_pairs) {
    const size_t nthreads = (size_t)gridDim.x * blockDim.x, tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (size_t i = tid; i < n_pairs; i += nthreads) {
        uint4 a = in[4 * i], b = in[4 * i + 1], c = in[4 * i + 2], d = in[4 * i + 3];
        out[2 * i] = make_uint4(a.x ^ c.x, a.y + c.y, a.z ^ c.z, a.w + c.w);
        out[2 * i + 1] = make_uint4(b.x ^ d.x, b.y + d.y, b.z ^ d.z, b.w + d.w);
    }
}
// ---- what does the END of a workgroup cost when thousands of them finish a round? `microbench arrive`: 2048 workgroups of 256
// threads, each does a little work and then (0) nothing, (1) one plain store by thread 0, (2) + a relaxed agent-scope fetch_add,
// (3) + an acq_rel agent-scope fetch_add, (4) acq_rel on one of 16 counters (128-byte lines apart), (5) a release fence + relaxed add
__global__ void k_arrive(unsigned *cnt, unsigned *out, int mode) {
    __shared__ unsigned sh[256];
    unsigned v = threadIdx.x * 2654435761u + blockIdx.x;
    for (int i = 0; i < 64; i++) v = v * 1664525u + 1013904223u;
    sh[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned acc = 0;
        for (int w = 0; w < 4; w++) acc += sh[64 * w];
        if (mode >= 1) out[16 * blockIdx.x] = acc;
        if (mode == 2) (void)__hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (mode == 3) (void)__hip_atomic_fetch_add(cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (mode == 4) (void)__hip_atomic_fetch_add(cnt + 32 * (1 + blockIdx.x % 16), 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (mode == 5) {
            __atomic_thread_fence(__ATOMIC_RELEASE);
            (void)__hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}
static int run_arrive() {
    unsigned *cnt, *out;
    CHK(hipMalloc(&cnt, 4096));
    CHK(hipMemset(cnt, 0, 4096));
    CHK(hipMalloc(&out, 4096 * 64));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const char *names[6] = {"work only", "+ plain store", "+ relaxed agent fetch_add", "+ acq_rel agent fetch_add", "+ acq_rel, 16 counters", "+ release fence, wg-scope add"};
    for (int nb : {256, 2048}) {
        for (int mode = 0; mode < 6; mode++) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                CHK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_arrive, dim3(nb), dim3(256), 0, 0, cnt, out, mode);
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            printf("%4d workgroups, %-32s: %.1f us\n", nb, names[mode], best * 1e3);
        }
    }
    return 0;
}

static int run_xcd() {
    const size_t table_bytes = (size_t)2 << 30, n16 = table_bytes / 16;
    const int blocks = 256 * 8, threads = 256;
    uint4 *table; unsigned *sink;
    CHK(hipMalloc(&table, table_bytes));
    CHK(hipMemset(table, 1, table_bytes));
    CHK(hipMalloc(&sink, (size_t)blocks * threads * 4));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const char *names[5] = {"grid-stride", "contiguous chunk per block", "contiguous eighth per XCD", "64 B per lane, grid-stride", "32 B per lane, grid-stride"};
    for (int mapping = 0; mapping < 5; mapping++) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; rep++) {
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_stream_map, dim3(blocks), dim3(threads), 0, 0, table, n16, mapping, sink);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        printf("stream 2 GiB, %-28s: %.3f ms, %.2f TB/s\n", names[mapping], best, table_bytes / best / 1e9);
    }
    uint4 *outb;
    CHK(hipMalloc(&outb, table_bytes / 2));
    for (int nb : {256, 512, 1024, 2048, 4096}) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; rep++) {
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_fold_traffic, dim3(nb), dim3(threads), 0, 0, table, outb, n16 / 4);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        printf("fold traffic (2 GiB read + 1 GiB written), %4d workgroups: %.3f ms, %.2f TB/s\n", nb, best, 1.5 * table_bytes / best / 1e9);
    }
    return 0;
}

static int run_mem(const char *mode) {
    const size_t table_bytes = (size_t)2 << 30, n_rows = table_bytes / 64;
    const int blocks = 256 * 8, threads = 256;
    const size_t rows_per_thread = 64;  // 2048 * 256 * 64 rows = 33.5 M rows = 2.1 GB
    uint4 *table; unsigned *sink;
    CHK(hipMalloc(&table, table_bytes));
    CHK(hipMemset(table, 1, table_bytes));
    CHK(hipMalloc(&sink, (size_t)blocks * threads * 4));
    bool gather = mode[0] == 'g';
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
        CHK(hipEventRecord(e0));
        if (gather) hipLaunchKernelGGL(k_gather64, dim3(blocks), dim3(threads), 0, 0, table, n_rows, rows_per_thread, sink);
        else hipLaunchKernelGGL(k_stream64, dim3(blocks), dim3(threads), 0, 0, table, n_rows, rows_per_thread, sink);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        double rows = (double)blocks * threads * rows_per_thread;
        printf("%s: %.0f rows of 64 B = %.1f MB per launch, %.3f ms, %.2f G rows/s, %.2f TB/s of row bytes\n", gather ? "gather64" : "stream64", rows,
               rows * 64 / 1e6, ms, rows / ms / 1e6, rows * 64 / ms / 1e9);
    }
    return 0;
}

// ---- launch chains: nine dependent short kernels (the shape of zg_run_sumcheck's 8 folds + tail) launched one by one against the same
// chain replayed as a hipGraph. usage: microbench launch
__global__ void k_chain_step(unsigned *p, unsigned rounds) {
    unsigned v = p[threadIdx.x];
    for (unsigned i = 0; i < rounds; i++) v = v * 1664525u + 1013904223u;
    p[threadIdx.x] = v;
}
static int run_launch() {
    unsigned *d;
    CHK(hipMalloc(&d, 256 * 4));
    CHK(hipMemset(d, 1, 256 * 4));
    hipStream_t st;
    CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int chain = 9, reps = 200;
    for (unsigned work : {1u, 2000u}) {  // ~2 us and ~10 us kernels
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; r++) {
            for (int k = 0; k < chain; k++) hipLaunchKernelGGL(k_chain_step, dim3(64), dim3(256), 0, st, d, work);
            CHK(hipStreamSynchronize(st));
        }
        double direct = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
        hipGraph_t graph;
        hipGraphExec_t exec;
        CHK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int k = 0; k < chain; k++) hipLaunchKernelGGL(k_chain_step, dim3(64), dim3(256), 0, st, d, work);
        CHK(hipStreamEndCapture(st, &graph));
        auto tc = std::chrono::steady_clock::now();
        CHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        double inst = std::chrono::duration<double>(std::chrono::steady_clock::now() - tc).count();
        CHK(hipGraphLaunch(exec, st));
        CHK(hipStreamSynchronize(st));
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; r++) {
            CHK(hipGraphLaunch(exec, st));
            CHK(hipStreamSynchronize(st));
        }
        double replay = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
        printf("chain of %d dependent kernels (%u iterations each): direct launches %.1f us, hipGraph replay %.1f us, instantiate %.1f us\n", chain, work,
               direct * 1e6, replay * 1e6, inst * 1e6);
        CHK(hipGraphExecDestroy(exec));
        CHK(hipGraphDestroy(graph));
    }
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 1 && argv[1][0] == 'l') return run_launch();
    if (argc > 1 && argv[1][0] == 'x') return run_xcd();
    if (argc > 1 && argv[1][0] == 'a') return run_arrive();
    if (argc > 1 && (argv[1][0] == 'g' || argv[1][0] == 's')) return run_mem(argv[1]);
    bool only_ec = argc > 1 && argv[1][0] == 'e';
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    int ncu = prop.multiProcessorCount;
    printf("device %s, CUs %d, clock %d kHz, memtime ticks are at 100MHz*? (compare with ns column)\n", prop.name, ncu, prop.clockRate);
    if (only_ec) {
        run("xyzz29_madd", k_madd29, 1, 500, ncu);
        run("affine_add_body", k_affine_add_body29, 1, 500, ncu);
        run("fe_inv_safegcd", k_inv_safegcd, 1, 40, ncu);
        return 0;
    }
    run("v_add_u32", k_add_u32, 64, 4000, ncu);
    run("v_mov_b32", k_mov_b32, 64, 4000, ncu);
    run("v_add3_u32", k_add3_u32, 64, 4000, ncu);
    run("add_co+addc_co", k_addc_pair, 64, 4000, ncu);
    run("v_lshl_add_u64", k_lshl_add_u64, 64, 4000, ncu);
    run("v_mul_u32_u24", k_mul_u32_u24, 64, 4000, ncu);
    run("v_mad_u32_u24", k_mad_u32_u24, 64, 4000, ncu);
    run("v_mul_lo_u32", k_mul_lo_u32, 64, 2000, ncu);
    run("v_mul_hi_u32", k_mul_hi_u32, 64, 2000, ncu);
    run("v_mad_u64_u32", k_mad_u64_u32, 64, 2000, ncu);
    run("v_fma_f64", k_fma_f64, 64, 2000, ncu);
    run("fe_mul(x2)", k_fe_mul, 2, 2000, ncu);
    run("xyzz_madd", k_madd, 1, 500, ncu);
    run("xyzz29_madd", k_madd29, 1, 500, ncu);
    run("affine_add_body", k_affine_add_body29, 1, 500, ncu);
    run("fe_inv_safegcd", k_inv_safegcd, 1, 40, ncu);
    return 0;
}
