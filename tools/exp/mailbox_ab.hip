// round 4 experiment: how a kernel's 64-byte result reaches the host. (a) system-scope stores + drain + sequence flag (the product's
// mailbox), (b) sixteen 8-byte {data, tag} granules, no drain. Host-observed latency from the launch call to the moment the host has
// the values, and the kernel's own duration (HIP events).   hipcc --offload-arch=gfx950 -O3 -o mailbox_ab mailbox_ab.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(1))) uint64_t gu64;
__global__ void pub_flag(uint64_t *vals, uint64_t *flag, uint64_t seq, uint64_t *sink) {
    if (threadIdx.x == 0) {
        for (int i = 0; i < 8; i++) __hip_atomic_store((gu64 *)vals + i, seq * 1000 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store((gu64 *)flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void pub_tag(uint64_t *gran, uint64_t seq) {
    if (threadIdx.x < 16) __hip_atomic_store((gu64 *)gran + threadIdx.x, ((seq * 1000 + threadIdx.x) << 32) | (seq & 0xffffffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void pub_none(uint64_t *dev, uint64_t seq) {
    if (threadIdx.x == 0) dev[0] = seq;
}
using clk = std::chrono::steady_clock;
int main() {
    uint64_t *h = nullptr, *d = nullptr;
    hipHostMalloc((void **)&h, 4096, hipHostMallocMapped | hipHostMallocCoherent);
    hipMalloc((void **)&d, 4096);
    for (int i = 0; i < 512; i++) h[i] = 0;
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int N = 2000;
    volatile uint64_t *vh = h;
    for (int mode = 0; mode < 3; mode++) {
        std::vector<double> lat;
        double evt = 0;
        for (int it = 1; it <= N; it++) {
            uint64_t seq = (uint64_t)mode * 100000 + it;
            auto t0 = clk::now();
            if (mode == 0) { hipLaunchKernelGGL(pub_flag, dim3(1), dim3(64), 0, st, h, h + 64, seq, d); while (vh[64] != seq) {} }
            else if (mode == 1) {
                hipLaunchKernelGGL(pub_tag, dim3(1), dim3(64), 0, st, h + 128, seq);
                for (;;) { bool ok = true; for (int i = 0; i < 16; i++) ok = ok && (uint32_t)vh[128 + i] == (uint32_t)seq; if (ok) break; }
            } else { hipLaunchKernelGGL(pub_none, dim3(1), dim3(64), 0, st, d, seq); hipStreamSynchronize(st); }
            lat.push_back(std::chrono::duration<double>(clk::now() - t0).count() * 1e6);
            hipStreamSynchronize(st);
        }
        for (int it = 0; it < 200; it++) {  // kernel duration by events
            uint64_t seq = (uint64_t)mode * 100000 + 50000 + it;
            hipEventRecord(e0, st);
            for (int k = 0; k < 10; k++) {
                if (mode == 0) hipLaunchKernelGGL(pub_flag, dim3(1), dim3(64), 0, st, h, h + 64, seq, d);
                else if (mode == 1) hipLaunchKernelGGL(pub_tag, dim3(1), dim3(64), 0, st, h + 128, seq);
                else hipLaunchKernelGGL(pub_none, dim3(1), dim3(64), 0, st, d, seq);
            }
            hipEventRecord(e1, st);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            evt += ms * 1e3 / 10;
        }
        std::sort(lat.begin(), lat.end());
        std::printf("%s: launch -> host has the values: median %.2f us, p10 %.2f, p90 %.2f; ten launches back to back: %.2f us each\n",
                    mode == 0 ? "stores + drain + flag" : mode == 1 ? "16 tagged granules    " : "device store + stream sync", lat[N / 2], lat[N / 10], lat[9 * N / 10], evt / 200);
    }
    return 0;
}
