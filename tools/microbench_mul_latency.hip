// Latency of a dependent chain of field products (fp29.hip.h), one wave up to four waves per SIMD. Build twice:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lat_col tools/microbench_mul_latency.hip                      (column-serial multiplier)
//   hipcc --offload-arch=gfx950 -O3 -DZG_F29_ROWWISE -o /tmp/lat_row tools/microbench_mul_latency.hip     (row-wise: 17 independent columns)
// Result (profiles/r3f_microbench_mul_latency.txt): 0.48 us per product for ONE wave in both forms, i.e. ~240 instructions at the
// 4-cycle issue rate — a lone wave already issues dependent multiply-adds back to back, so the reduction tail of an MSM (chains of
// point operations by one or two waves) is bound by its instruction count, not by multiplier latency; a row-wise multiplier buys nothing.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../zolt_amd/csrc/common.hip.h"
#include "../zolt_amd/csrc/field.hip.h"
#include "../zolt_amd/csrc/fp29.hip.h"
using namespace zg;
__global__ void chain(uint32_t *out, int n) {
    F29 a, b;
    for (int i = 0; i < 9; i++) { a.l[i] = threadIdx.x * 77 + i * 13 + 5; b.l[i] = threadIdx.x * 31 + i * 7 + 3; }
    for (int k = 0; k < n; k++) a = f29_mul(a, b);
    uint32_t x = 0;
    for (int i = 0; i < 9; i++) x ^= a.l[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
int main() {
    uint32_t *d;
    if (hipMalloc(&d, 8 << 20) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {1, 1024, 4096})
        for (int threads : {64, 256}) {
            float best = 1e9;
            for (int rep = 0; rep < 5; rep++) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(chain, dim3(blocks), dim3(threads), 0, 0, d, 2000);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            uint32_t h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
            printf("blocks %4d threads %3d: %8.3f us per product (chain of 2000) chk %08x\n", blocks, threads, best * 1e3 / 2000, h);
        }
    return 0;
}
