// microbench_tlb.hip — does the cost of a random 64-byte row gather depend on the SIZE OF THE ALLOCATION it falls in, or only on the
// footprint touched? (The 2^22-point MSM's accumulate kernel pays +10 % per addition over the 2^20 one; folding its gathers into
// the first 64 MB of the same 4 GiB table removes all of it, folding them into the first 1 GiB removes almost nothing, while the
// 2^20 MSM's own 1 GiB table costs nothing: tools/exp/archive/exp_mask.sh.)
//
//   hipcc -O3 --offload-arch=gfx950 tools/microbench_tlb.hip -o tools/microbench_tlb && tools/microbench_tlb
//
// Every lane walks a DEPENDENT chain of gathers (the next row index comes out of the loaded row, as far as the compiler can tell),
// so the figure is latency-shaped like the accumulate kernel's prefetch, not a bandwidth figure; `waves` sets how many chains
// are in flight per SIMD.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x)                                                                          \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)

struct Regions {
    const uint4 *base[16];
    unsigned long long rows_per_region;  // power of two
    unsigned n_regions;                  // power of two
};

__global__ void __launch_bounds__(256) k_chain(Regions rg, unsigned steps, unsigned *sink) {
    size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long h = tid * 0x9E3779B97F4A7C15ull + 0x1234567ull;
    unsigned acc = 0;
    for (unsigned k = 0; k < steps; k++) {
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        unsigned long long row = h & (rg.rows_per_region * rg.n_regions - 1);
        const uint4 *p = rg.base[row / rg.rows_per_region] + 4 * (row & (rg.rows_per_region - 1));
        uint4 a = p[0], b = p[1], c = p[2], d = p[3];
        unsigned v = a.x ^ b.y ^ c.z ^ d.w;  // the table holds zeros: v == 0, but the next index waits for the load
        acc += v;
        h += v;
    }
    sink[tid] = acc;
}

static double run(const Regions &rg, int blocks, unsigned steps, unsigned *sink) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(256), 0, 0, rg, steps, sink);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
    return best;
}

int main() {
    const size_t GiB = (size_t)1 << 30;
    unsigned *sink;
    CHK(hipMalloc(&sink, (size_t)8192 * 256 * 4));
    uint4 *big, *one, *four[4], *sixteen[16];
    CHK(hipMalloc(&big, 4 * GiB)); CHK(hipMemset(big, 0, 4 * GiB));
    CHK(hipMalloc(&one, GiB)); CHK(hipMemset(one, 0, GiB));
    for (auto &p : four) { CHK(hipMalloc(&p, GiB)); CHK(hipMemset(p, 0, GiB)); }
    for (auto &p : sixteen) { CHK(hipMalloc(&p, GiB / 4)); CHK(hipMemset(p, 0, GiB / 4)); }
    CHK(hipDeviceSynchronize());
    struct Case { const char *name; Regions rg; };
    std::vector<Case> cases;
    auto single = [](const uint4 *p, size_t bytes) { Regions r{}; r.base[0] = p; r.rows_per_region = bytes / 64; r.n_regions = 1; return r; };
    cases.push_back({"4 GiB allocation, first 64 MB", single(big, GiB / 16)});
    cases.push_back({"4 GiB allocation, first 256 MB", single(big, GiB / 4)});
    cases.push_back({"4 GiB allocation, first 1 GiB", single(big, GiB)});
    cases.push_back({"4 GiB allocation, all of it", single(big, 4 * GiB)});
    cases.push_back({"1 GiB allocation, all of it", single(one, GiB)});
    cases.push_back({"1 GiB allocation, first 64 MB", single(one, GiB / 16)});
    { Regions r{}; for (int i = 0; i < 4; i++) r.base[i] = four[i]; r.rows_per_region = GiB / 64; r.n_regions = 4; cases.push_back({"4 x 1 GiB allocations", r}); }
    { Regions r{}; for (int i = 0; i < 16; i++) r.base[i] = sixteen[i]; r.rows_per_region = GiB / 4 / 64; r.n_regions = 16; cases.push_back({"16 x 256 MB allocations", r}); }
    { Regions r{}; for (int i = 0; i < 4; i++) r.base[i] = big + i * (GiB / 16); r.rows_per_region = GiB / 64; r.n_regions = 4; cases.push_back({"4 GiB allocation as 4 regions", r}); }
    for (int waves : {1, 4, 8}) {  // chains in flight per SIMD (256 CUs x 4 SIMDs)
        int blocks = 256 * waves;
        unsigned steps = 256;
        printf("-- %d wave(s) per SIMD, %u dependent gathers per lane\n", waves, steps);
        for (auto &c : cases) {
            double ms = run(c.rg, blocks, steps, sink);
            printf("%-34s %8.3f ms  %7.1f ns per step  %6.2f G rows/s\n", c.name, ms, ms * 1e6 / steps, (double)blocks * 256 * steps / ms / 1e6);
        }
    }
    return 0;
}
