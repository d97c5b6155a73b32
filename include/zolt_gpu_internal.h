/*
 * zolt_gpu_internal.h — test and bench scaffolding exported by libzolt_gpu.so that is NOT part of the drop-in boundary
 * (include/zolt_gpu.h). tests/ and bench.py bind these through zolt_amd/lib.py; a Zig host never needs them.
 */
#ifndef ZOLT_GPU_INTERNAL_H
#define ZOLT_GPU_INTERNAL_H

#include "zolt_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* zg_field_op self-test hooks for the device arithmetic (results always come back canonical, Montgomery-2^256) */
#define ZG_OP_MUL29 9        /* Fp only: a*b through the MSM's 9x29-bit lazy representation (csrc/fp29.hip.h) */
#define ZG_OP_SQR29 10       /* Fp only: a^2 through the lazy representation (b ignored) */
#define ZG_OP_X3_29 11       /* Fp only: a chain touching every biased lazy subtraction and the zero test (see runtime.hip) */
#define ZG_OP_INV_XGCD 12    /* same value as ZG_OP_INV via plain binary extended Euclid (cross-check) */
#define ZG_OP_INV_SAFEGCD 13 /* same value via batched Bernstein-Yang division steps (the device's toAffine path) */

/* ------------------------------------------------------------------ profiling */
/* Per-kernel timing with HIP events recorded on the stream each kernel is launched on
 * (bench.py's roofline figure). zg_profile_begin enables recording of up to max_records
 * kernel intervals; zg_profile_end synchronises, adds the elapsed times up per kernel id
 * (milliseconds, launch counts) and disables recording. Not thread-safe; bench use only. */
#define ZG_PROF_MSM_DIGITS 0
#define ZG_PROF_MSM_SORT 1       /* scan + scatter */
#define ZG_PROF_MSM_ACCUMULATE 2 /* bucket accumulation: the dominant MSM kernel */
#define ZG_PROF_MSM_REDUCE 3     /* bucket reduction levels + final */
#define ZG_PROF_EQ_TABLE 4
#define ZG_PROF_SC_FOLD 5        /* fold + fused next-round sums */
#define ZG_PROF_SC_SUMS 6
#define ZG_PROF_COMBINE 7        /* Spartan combine */
#define ZG_PROF_NKERNELS 8
ZG_API int zg_profile_begin(int max_records);
ZG_API int zg_profile_end(double ms_out[ZG_PROF_NKERNELS], uint64_t count_out[ZG_PROF_NKERNELS]);

/* Set-up phase split (tools/bench_sumcheck: "time set-up phases separately"). With ZG_SETUP_TIMES=1 in the environment the two entry points
 * that build large device state from host inputs — zg_fr_rows_from_columns and zg_rrw_open / zg_rrw_open_trace — synchronise between
 * their phases and record, for the calling thread's last call: out[0] = device allocation (pool), out[1] = host-to-device copies,
 * out[2] = kernels, out[3] = everything else inside the call, in milliseconds. Without the variable the calls run unsplit and this
 * returns zeros. */
ZG_API int zg_last_setup_times(double out[4]);

/* RCCL communicator sets (ncclCommInitAll) created so far by the one-process / several-GPU path: a set is shared by the sharded
 * handles made while the same number of devices is bound; tests check that re-binding creates a new set instead of failing. */
ZG_API int zg_sharded_comm_sets_created(void);

/* ZG_POOL_DEBUG (csrc/runtime.hip): the device pool's "a freed block is idle" contract, checked. With ZG_POOL_DEBUG=1 in the environment
 * freed (and fresh) blocks are filled with 0xDBDBDBDB and verified before they are handed out again; a block written after its free is
 * refused and counted (=2: the process aborts). zg_pool_debug_stats: out[0] = hits, out[1] = frees made while the device's library stream
 * was busy (information only), out[2] = blocks verified, out[3] = bytes poisoned; returns the mode (0 = off).
 * zg_pool_debug_selftest breaks the contract on purpose: 1 = the mode caught it, 0 = mode off (nothing checked), < 0 = error. */
ZG_API int zg_pool_debug_stats(uint64_t out[4]);
ZG_API int zg_pool_debug_selftest(void);

#ifdef __cplusplus
}
#endif
#endif /* ZOLT_GPU_INTERNAL_H */
