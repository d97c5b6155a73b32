/*
 * zolt_gpu.h — C ABI of libzolt_gpu.so, the MI355X (gfx950) backend for Zolt's
 * data-parallel prover inner loops.
 *
 * The reference (MatteoMer/zolt) has no FFI boundary; the seam is a set of Zig
 * generic functions. Each entry point below names the reference
 * function it replaces (paths relative to the reference root). A Zig shim keeps the
 * `src/msm`, `src/poly`, `src/subprotocols` module APIs and forwards here — see
 * INTEGRATION.md for the `extern fn` declarations.
 *
 * Conventions
 *   - field element  : uint64_t[4], little-endian limbs, Montgomery form (R = 2^256),
 *                      canonical (< modulus) — src/field/mod.zig:131,583-584.
 *                      "fr" = BN254 scalar field (:16-41), "fp" = base field (:51-75).
 *   - affine G1 point: uint64_t[8] = x limbs then y limbs (Fp), plus an out-of-band
 *                      uint8_t infinity flag (the Zig struct layout {x,y,infinity} is not
 *                      ABI-stable, src/msm/mod.zig:19-21). Identity is written as
 *                      x = y = 0, inf = 1 like AffinePoint.identity() (:24-30).
 *   - Jacobian record: uint64_t[12] = X,Y,Z (Fp); identity = (1,1,0) (:154-160).
 *   - every function returns 0 on success or a ZG_ERR_* code, never throws, and is
 *     re-entrant (MSM.compute is called from std.Thread workers, src/msm/mod.zig:637,732).
 *     zg_last_error() gives the calling thread's last message.
 *   - "_dev" variants take DEVICE pointers (hipMalloc'ed / torch tensors' data_ptr) and a
 *     hipStream_t passed as void* (NULL = the library's own non-blocking stream, which is NOT ordered
 *     with the legacy default stream: pass your stream explicitly when other work consumes the
 *     result); work is enqueued asynchronously unless the function returns a host value.
 *   - No CPU fallback exists: without a usable gfx950 device every compute entry point
 *     fails with ZG_ERR_NO_DEVICE.
 */
#ifndef ZOLT_GPU_H
#define ZOLT_GPU_H

#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__)
#define ZG_API __attribute__((visibility("default")))
#else
#define ZG_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define ZG_OK 0
#define ZG_ERR_INVALID 1   /* bad argument */
#define ZG_ERR_HIP 2       /* HIP runtime error (see zg_last_error) */
#define ZG_ERR_NOMEM 3
#define ZG_ERR_NO_DEVICE 4
#define ZG_ERR_VERIFY 5    /* error.SumcheckVerificationFailed (src/subprotocols/mod.zig:175-178) */

#define ZG_FIELD_FR 0
#define ZG_FIELD_FP 1

/* elementwise ops for zg_field_op */
#define ZG_OP_MUL 0        /* montgomeryMul, src/field/mod.zig:269-308 / :735-779 */
#define ZG_OP_ADD 1        /* :402-417 / :782-798 */
#define ZG_OP_SUB 2        /* :420-435 / :801-816 */
#define ZG_OP_NEG 3        /* :494-497 / :944-947 (b ignored) */
#define ZG_OP_SQR 4        /* :443-445 / :866-941 (b ignored) */
#define ZG_OP_INV 5        /* :500-518 / :955-983; inverse(0) -> 0 (b ignored) */
#define ZG_OP_FROM_MONT 6  /* :187-189 / :642-645 (b ignored) */
#define ZG_OP_TO_MONT 7    /* fromBytes' reduction :171-184 / :625-639: raw 256-bit LE -> Montgomery (b ignored) */
/* op codes 9..13 are self-test hooks for the device arithmetic: include/zolt_gpu_internal.h */

/* ------------------------------------------------------------------ ABI version and sections
 * MAJOR changes when an existing signature or layout changes, MINOR when entry points are added. A host binds against the header it was
 * built with and checks zg_abi_version() >> 16 == ZG_ABI_MAJOR (and the minor it needs) after loading the library.
 *
 * Sections of this header:
 *   CORE      lifecycle, device memory, field vectors, G1 bases / MSM / HyperKZG, poly tables, sumcheck sessions, product-form
 *             sessions, several GPUs in one process — the path BASELINE.json's north_star names.
 *   OPTIONAL  protocol-specific device sessions for two of the reference's stage provers (zg_rrw_*: RegistersReadWriteChecking,
 *             zg_rwc_*: RamReadWriteChecking). A host that only re-points MSM / poly / sumcheck call sites never touches them:
 *             compile with -DZG_NO_PROTOCOL_SESSIONS to leave them out of the binding; zg_abi_features() reports whether the loaded
 *             library carries them.
 * 1.10 (round 6): NO entry point added — the boundary is frozen; zg_hyperkzg_setup accepts the reference's largest key (2^24 + 256 powers,
 * formerly at most 2^24) and keeps the identity flags of tau = 0. */
#define ZG_ABI_MAJOR 1
#define ZG_ABI_MINOR 10
#define ZG_FEATURE_PROTOCOL_SESSIONS 1u /* zg_rrw_* and zg_rwc_* are exported */
#define ZG_FEATURE_RCCL 2u              /* the several-GPU entry points can exchange partials over RCCL */
#define ZG_FEATURE_COLUMN_INGEST 4u     /* zg_fr_rows_from_columns[_dev] */
ZG_API uint32_t zg_abi_version(void);  /* (ZG_ABI_MAJOR << 16) | ZG_ABI_MINOR of the library that was loaded */
ZG_API uint32_t zg_abi_features(void); /* ZG_FEATURE_* bits */

/* ------------------------------------------------------------------ lifecycle */
/* Binds the calling process to one GPU (device < 0: keep the current HIP device) and
 * creates the library stream. Idempotent. */
ZG_API int zg_init(int device);
/* ONE process driving several GPUs — the reference's own process model: `zolt prove` is a single process (src/main.zig:271-696)
 * and ParallelMSM.compute runs threads inside it (src/msm/mod.zig:588-653). Binds devices 0..n_devices-1 (n_devices <= 0: every
 * visible device); device 0 stays the primary for the single-device entry points. May follow zg_init(0) / zg_init(-1) on
 * device 0. The sharded entry points below ("several GPUs in one process") then spread their work over the bound devices. */
ZG_API int zg_init_devices(int n_devices);
ZG_API int zg_n_devices(void); /* devices bound: 1 after zg_init, n after zg_init_devices(n), 0 before either */
ZG_API void zg_shutdown(void);
ZG_API const char *zg_last_error(void);
ZG_API const char *zg_version(void);
ZG_API int zg_device_count(void);

/* raw device memory, for hosts without their own HIP binding (the Zig shim). The two copies run on the library's stream and return when
 * they are done: ordered after every earlier call that was given stream = NULL (e.g. the asynchronous zg_fr_eq_table_dev).
 * zg_dev_free returns when the device is idle (hipFree's own guarantee) and keeps the block in the library's device pool for the next
 * allocation of its size class — by zg_dev_alloc or by the library itself (scratch buffers, session tables): a prover that allocates
 * its tables per proof does not pay hipMalloc / hipFree each time, whatever their size. The pool keeps up to a quarter of the device's
 * memory (ZG_DEV_ALLOC_CACHE_MB overrides; 0 = no caching); an allocation that fails returns the idle blocks to the driver and tries
 * again; zg_dev_trim does so on request, zg_shutdown at the end. */
ZG_API int zg_dev_alloc(size_t bytes, void **dptr);
ZG_API int zg_dev_free(void *dptr);
ZG_API int zg_dev_trim(void);
/* pinned host memory (hipHostMalloc) for callers that fill large inputs in place — trace columns, scalar vectors: copies from it run by
 * DMA at link rate whatever the page state of the process. Not pooled: the caller owns the lifetime. */
ZG_API int zg_host_alloc(size_t bytes, void **ptr);
ZG_API int zg_host_free(void *ptr);
ZG_API int zg_dev_memset(void *dst_dev, int byte_value, size_t bytes); /* on the library's stream, returns when done (as the copies do) */
ZG_API int zg_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes);
ZG_API int zg_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes);
ZG_API int zg_sync(void);

/* ------------------------------------------------------------------ field vectors */
/* out[i] = op(a[i], b[i]) over n elements; host pointers. Replaces the scalar loops of
 * field.BatchOps (src/field/mod.zig:1164-1280) and backs the device-arithmetic unit tests. */
ZG_API int zg_field_op(int field, int op, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);

/* DensePolynomial.scale (src/poly/mod.zig:112-126): out[i] = a[i] * s. (DensePolynomial.add, :94-110, is zg_field_op ZG_OP_ADD.) */
ZG_API int zg_fr_scale(const uint64_t *a, size_t n, const uint64_t s[4], uint64_t *out);

/* ------------------------------------------------------------------ G1 bases (SRS) */
typedef struct zg_bases_s *zg_bases_t;

/* MSM tuning. window_bits 0 = auto from n; precompute_levels: 1 = none, 0 = auto,
 * k>1 = store 2^(c*G*l)*P for l<k at upload so k windows share one bucket set
 * (HBM cost k*64 B per base). expected_uses: how many MSMs the handle is expected to serve — 0 = many (an SRS that lives for
 * the whole run: the full table, 64*W bytes per base and a one-time build — 1 GB and ~13 ms at 2^20 bases, 4 GB and ~47 ms at 2^22 —
 * which pays back after ~12 MSMs issued one at a time, ~35 when several are kept in flight: bench.py reports table_build_ms,
 * table_bytes and breakeven_msms in its config); 1..15 = an ad-hoc bases slice (MSM.compute on a temporary): auto picks
 * precompute_levels = 1, no table build. Results do not depend on these. */
typedef struct {
    int window_bits;
    int precompute_levels;
    int expected_uses;
} zg_msm_config;

/* Upload n affine bases once and keep them resident in HBM for any number of MSMs —
 * the device-side image of HyperKZG SetupParams.powers_of_tau_g1
 * (src/poly/commitment/mod.zig:122-140), which lives for the whole prover run.
 * inf may be NULL (no infinity bases). cfg may be NULL (auto). Bases must be on the
 * curve y^2 = x^3 + 3 (src/msm/mod.zig:106-115). */
ZG_API int zg_g1_bases_upload(const uint64_t *xy, const uint8_t *inf, size_t n, const zg_msm_config *cfg, zg_bases_t *out);
ZG_API int zg_g1_bases_upload_dev(const uint64_t *d_xy, const uint8_t *d_inf, size_t n, const zg_msm_config *cfg, void *stream,
                           zg_bases_t *out);
ZG_API int zg_g1_bases_free(zg_bases_t b);
ZG_API size_t zg_g1_bases_len(zg_bases_t b);
/* the plan the handle was built with: window bits c (optimalWindowSize's role, src/msm/mod.zig:475-484, chosen for the GPU),
 * windows per scalar ceil(255 / c), table levels stored per base. Any pointer may be NULL. */
ZG_API int zg_g1_bases_plan(zg_bases_t b, int *window_bits, int *windows, int *precompute_levels);
/* bytes of HBM the handle holds for its bases: the table of precomputed multiples (levels * 64 B per base) or the plain bases */
ZG_API size_t zg_g1_bases_table_bytes(zg_bases_t b);

/* ------------------------------------------------------------------ MSM */
/* MSM(F,G).compute(bases[off..off+n], scalars) -> Affine   (src/msm/mod.zig:355-438)
 * = HyperKZG.commit when bases is the SRS (src/poly/commitment/mod.zig:239-255).
 * n = 0 -> identity (:361-363). Scalars: n x 4 Montgomery Fr limbs on the host. */
ZG_API int zg_msm_g1(zg_bases_t b, size_t off, size_t n, const uint64_t *scalars_mont, uint64_t out_xy[8], uint8_t *out_inf);
/* The same MSM for scalars that are F.fromU64 of machine words, handed over as the words: what `zolt prove` commits to — program bytes,
 * memory values, rd_value per cycle (commitBytecode / commitMemory / commitRegisters, src/zkvm/mod.zig:1518-1617) — is such a vector.
 * 8 instead of 32 bytes per scalar cross PCIe, the conversion runs on the device, and a 64-bit scalar has digits in 4 of the 15 windows
 * only. out = MSM.compute(bases[off..off+n], values.map(F.fromU64)), bit for bit. */
ZG_API int zg_msm_g1_u64(zg_bases_t b, size_t off, size_t n, const uint64_t *values_u64, uint64_t out_xy[8], uint8_t *out_inf);
/* same, scalars already resident in HBM */
ZG_API int zg_msm_g1_dev(zg_bases_t b, size_t off, size_t n, const uint64_t *d_scalars_mont, void *stream, uint64_t out_xy[8],
                  uint8_t *out_inf);
/* asynchronous form: the affine result (8 limbs) and the flag are written to DEVICE memory */
ZG_API int zg_msm_g1_dev_async(zg_bases_t b, size_t off, size_t n, const uint64_t *d_scalars_mont, void *stream,
                        uint64_t *d_out_xy, uint8_t *d_out_inf);
/* BatchMSM / ParallelBatchMSM.compute (src/msm/mod.zig:545-565,683-748) = HyperKZG.batchCommit
 * (src/poly/commitment/mod.zig:558-570): k scalar vectors of length n over bases[0..n]. */
ZG_API int zg_msm_g1_batch(zg_bases_t b, size_t n, const uint64_t *const *scalar_batches, size_t k, uint64_t *out_xy /* k*8 */,
                    uint8_t *out_inf /* k */);
/* same with the k vectors resident in HBM back to back (k*n*4 limbs); asynchronous: record i = d_out9[9*i .. 9*i+9) =
 * affine xy[8] followed by a flag word (low byte 1 = identity). Short vectors (Dory row commitments,
 * src/poly/commitment/dory.zig:646-670; commitments of <= 2^14-entry polynomials) are FUSED: the k MSMs run as one
 * sort / accumulate / reduce pass with k times the bucket groups, instead of k latency-bound launch sets; long vectors
 * rotate over three streams so their tails overlap. Results are identical to k separate zg_msm_g1 calls. */
ZG_API int zg_msm_g1_batch_dev(zg_bases_t b, size_t n, const uint64_t *d_scalars_mont, size_t k, void *stream, uint64_t *d_out9);
/* ParallelMSM's per-chunk result (src/msm/mod.zig:656-665): this GPU's partial sum as the
 * reference's Jacobian record fromAffine(SingleMSM.compute(chunk)) = (x,y,1) or (1,1,0),
 * written to DEVICE memory so it can be all-gathered over RCCL without a host round trip. */
ZG_API int zg_msm_g1_partial_dev(zg_bases_t b, size_t off, size_t n, const uint64_t *d_scalars_mont, void *stream,
                          uint64_t *d_out_jac /* 12 */);
/* Same partial without the normalisation: some Jacobian representative (X, Y, Z) of the shard's sum (identity:
 * Z = 0). It exists only to be fed to zg_g1_combine_partials_dev*, whose result is identical; skipping the
 * per-GPU inversion shortens every rank's step in a sharded MSM. */
ZG_API int zg_msm_g1_partial_fast_dev(zg_bases_t b, size_t off, size_t n, const uint64_t *d_scalars_mont, void *stream,
                               uint64_t *d_out_jac /* 12 */);
/* ParallelMSM's serial combine + toAffine (src/msm/mod.zig:647-652) over k gathered
 * Jacobian partials resident on the device. */
ZG_API int zg_g1_combine_partials_dev(const uint64_t *d_partials_jac /* k*12 */, size_t k, void *stream, uint64_t out_xy[8],
                               uint8_t *out_inf);
/* asynchronous form: result (8 limbs) and flag land in DEVICE memory, ordered on `stream` */
ZG_API int zg_g1_combine_partials_dev_async(const uint64_t *d_partials_jac /* k*12 */, size_t k, void *stream, uint64_t *d_out_xy,
                                     uint8_t *d_out_inf);
/* The same combine for a BATCH of m MSMs behind ONE exchange (ParallelBatchMSM's per-thread partial lists, src/msm/mod.zig:683-748): the
 * gathered buffer holds, for every rank r, that rank's m records back to back at d_partials_jac + r * rank_stride (u64 words,
 * rank_stride >= 12 * m — the layout of one all-gather of m * 96 bytes per rank); result j (xy[8] + flag word, 9 words) goes to
 * d_out9 + 9 * j. One launch of m workgroups, stream-ordered. */
ZG_API int zg_g1_combine_partials_batch_dev_async(const uint64_t *d_partials_jac, size_t ranks, size_t rank_stride, size_t m, void *stream,
                                           uint64_t *d_out9);
/* AffinePoint.isOnCurve (src/msm/mod.zig:106-115) for n points: out[i] = 1 iff infinity or y^2 == x^3 + 3
 * (what parseG1Uncompressed checks per SRS point, src/poly/commitment/srs.zig:93-96). */
ZG_API int zg_g1_is_on_curve_batch(const uint64_t *xy, const uint8_t *inf, size_t n, uint8_t *out);
/* AffinePoint.add for n independent pairs (src/msm/mod.zig:74-103): the lambda formulas with one field inversion per pair,
 * x1 == x2 resolved as the reference does (:79-88: y1 == -y2 -> identity, y1 == y2 -> double, :118-138); identity operands pass
 * the other point through. AffinePoint.double(p) = add(p, p). Flags may be NULL (no identity inputs / flags not wanted). */
ZG_API int zg_g1_affine_add_batch(const uint64_t *a_xy, const uint8_t *a_inf, const uint64_t *b_xy, const uint8_t *b_inf, size_t n,
                           uint64_t *out_xy, uint8_t *out_inf);
/* MSM(F,G).scalarMul(base, scalar).toAffine() for n independent (base, scalar) pairs
 * (src/msm/mod.zig:503-540) — the primitive of HyperKZG.setup (commitment/mod.zig:194-199). */
ZG_API int zg_g1_scalar_mul_batch(const uint64_t *xy, const uint8_t *inf, const uint64_t *scalars_mont, size_t n,
                           uint64_t *out_xy, uint8_t *out_inf);

/* The same for n scalars over ONE base — HyperKZG.setup's loop powers[i] = MSM.scalarMul(g1, tau^i).toAffine()
 * (src/poly/commitment/mod.zig:194-199; generateMockSRS, srs.zig:326-355): a shared table of the 255 multiples of 2^(8w) * base
 * per 8-bit window, so an output is at most 32 mixed additions and one toAffine instead of 254 doublings + ~127 additions. */
ZG_API int zg_g1_fixed_base_mul_batch(const uint64_t base_xy[8], uint8_t base_inf, const uint64_t *scalars_mont, size_t n, uint64_t *out_xy,
                               uint8_t *out_inf);

/* HyperKZG.setup's G1 side with nothing leaving the device (generateMockSRS, src/poly/commitment/mod.zig:174-213: powers[i] =
 * scalarMul(g1, tau^i), i < n < 2^27 — the reference's largest key is 2^24 + 256 powers, src/host/mod.zig:384-387): powers of tau,
 * fixed-base batch and the handle (with its table of multiples) are built in HBM.
 * out_xy / out_inf (n*8 words / n bytes, host) may be NULL when the caller does not need the points themselves. The handle equals
 * zg_g1_bases_upload of zg_g1_fixed_base_mul_batch's output for the scalars tau^i, infinity flags included (tau = 0: every power
 * but the first is the identity). */
ZG_API int zg_hyperkzg_setup(const uint64_t base_xy[8], const uint64_t tau[4], size_t n, const zg_msm_config *cfg, uint64_t *out_xy, uint8_t *out_inf,
                             zg_bases_t *out);
/* HyperKZG.open (src/poly/commitment/mod.zig:261-324), resident on the device: per variable i the quotient
 * q[j] = cur[j+half] - cur[j] is committed (MSM over srs[0..min(half, srs_len))), then cur is folded by point[i]
 * (high half). Writes num_vars quotient commitments (rounds that the reference skips when the table runs out are
 * reported as the identity) and the final evaluation; num_vars == 0 returns `value` like the reference. */
ZG_API int zg_hyperkzg_open(zg_bases_t srs, const uint64_t *evals, size_t n_evals, const uint64_t *point, size_t num_vars,
                     const uint64_t value[4], uint64_t *q_xy /* num_vars*8 */, uint8_t *q_inf /* num_vars */,
                     uint64_t final_eval[4]);
/* same with the evaluation table already resident in HBM (a prover that has just committed the polynomial from device memory):
 * no 32*n-byte upload; the table is copied on the device (the loop folds in place, the caller's polynomial stays intact) */
ZG_API int zg_hyperkzg_open_dev(zg_bases_t srs, const uint64_t *d_evals, size_t n_evals, const uint64_t *point, size_t num_vars,
                         const uint64_t value[4], void *stream, uint64_t *q_xy /* num_vars*8 */, uint8_t *q_inf /* num_vars */,
                         uint64_t final_eval[4]);
/* HyperKZG.batchOpen (src/poly/commitment/mod.zig:607-732): k polynomials opened at one point through the random linear
 * combination P = sum_i gamma^i p_i, gamma = fromU64(0x9a8b7c6d) * prod_j (point[j] + 11) (the reference's deterministic
 * stand-in for a transcript challenge, :633-640). Outputs: evaluations[k] = evaluateMultilinear(p_i, point) EXACTLY as the
 * reference computes it (:788-817 — the direct sum for point.len <= 10 and len <= 1024, evals[0] otherwise); the quotient
 * commitments of P (*n_quot of them; fewer than num_vars when the fold runs out of elements, :686,714-720; the unused
 * slots are reported as identity); final_eval (combined_eval when num_vars == 0, :665-673) and gamma. Polynomials shorter
 * than polys[0] are zero-extended, longer ones truncated (:649-651). */
ZG_API int zg_hyperkzg_batch_open(zg_bases_t srs, const uint64_t *const *polys, const size_t *lens, size_t k, const uint64_t *point,
                           size_t num_vars, uint64_t *q_xy /* num_vars*8 */, uint8_t *q_inf /* num_vars */, size_t *n_quot,
                           uint64_t *evaluations /* k*4 */, uint64_t final_eval[4], uint64_t gamma[4]);

/* ------------------------------------------------------------------ poly tables */
/* EqPolynomial.evals / evalsSliceWithScaling (src/poly/mod.zig:240-290): out[2^v], index MSB <-> r[0];
 * scale may be NULL (= one). Identical values to GruenSplitEqPolynomial's tables
 * (src/poly/split_eq.zig:122-171). */
ZG_API int zg_fr_eq_table(const uint64_t *r, size_t v, const uint64_t *scale, uint64_t *out);
/* device output; ASYNCHRONOUS: one launch on `stream`, r and scale travel as kernel arguments (nothing of the caller's is read
 * after the call returns, no upload precedes the kernel). v <= 34. */
ZG_API int zg_fr_eq_table_dev(const uint64_t *r_host, size_t v, const uint64_t *scale_host, uint64_t *d_out, void *stream);
/* The eq+1 evaluation table of EqPlusOnePrefixSuffixPoly (src/poly/mod.zig:462-560, computeEqPlusOneEvals :530-548; the same helper in
 * src/zkvm/spartan/stage3_prover.zig:1878-1894): out[j] = EqPlusOnePolynomial.mle(r, bits(j)), r[0] <-> MSB, 2^v entries. Over the
 * boolean cube eq+1(r, j) = eq(r, j - 1) and out[0] = 0, so this is the eq table moved up by one entry (the reference evaluates the
 * general formula, v^2 products per entry). */
ZG_API int zg_fr_eq_plus_one_table(const uint64_t *r, size_t v, uint64_t *out);
ZG_API int zg_fr_eq_plus_one_table_dev(const uint64_t *r_host, size_t v, uint64_t *d_out, void *stream);
/* GruenSplitEqPolynomial.initWithScaling's prefix-table set in one launch (src/poly/split_eq.zig:122-171: E_out_vec / E_in_vec,
 * "append the new variable as LSB" and KEEP every level): out holds the v+1 tables eq(tau[0..k), .), k = 0..v, back to back —
 * table k has 2^k entries and starts at element 2^k - 1 (so out has 2^(v+1) - 1 elements; table 0 is [1]); within a table
 * tau[0] <-> MSB of the index. The caller passes tau[0..m) for E_out_vec and tau[m..n-1) for E_in_vec (:91-93). v <= 24. */
ZG_API int zg_fr_eq_prefix_tables(const uint64_t *tau, size_t v, uint64_t *out /* (2^(v+1)-1)*4 */);
ZG_API int zg_fr_eq_prefix_tables_dev(const uint64_t *tau_host, size_t v, uint64_t *d_out, void *stream);
/* DensePolynomial.evaluate (src/poly/mod.zig:73-92): sum_i evals[i] * prod_j (bit_j(i) ? point[j] : 1 - point[j]),
 * index bit j <-> point[j] (LSB first). The reference expands every term (O(n*v) multiplications); here the
 * weights are one eq table (point reversed) and the sum is a device dot product — same field value. */
ZG_API int zg_fr_dense_evaluate(const uint64_t *evals, size_t num_vars, const uint64_t *point, uint64_t out[4]);
/* R1CSInputEvaluator.computeClaimedInputs (src/zkvm/r1cs/evaluation.zig:55-122) and the factor claims of
 * ProductVirtualRemainderProver.computeOpeningClaims (src/zkvm/spartan/product_remainder.zig:396-425): the MLE evaluation of k <= 64
 * column polynomials at one point from a CYCLE-MAJOR matrix (rows[t*k + i] = column i at cycle t, the layout of R1CSCycleInputs.values):
 * out[i] = sum_{t < min(n_rows, 2^v)} eq(r, t) * rows[t*k + i], eq = EqPolynomial(r).evals (r[0] <-> MSB of t). */
ZG_API int zg_fr_rows_mle(const uint64_t *rows, size_t n_rows, size_t k, const uint64_t *r, size_t v, uint64_t *out /* k*4 */);
ZG_API int zg_fr_rows_mle_dev(const uint64_t *d_rows, size_t n_rows, size_t k, const uint64_t *r_host, size_t v, void *stream,
                       uint64_t *out /* host, k*4 */);
/* StreamingOuterProver.materializeLinearPhasePolynomials (src/zkvm/spartan/streaming_outer.zig:258-372): per cycle, Az and Bz of the two
 * constraint groups are Lagrange-weighted sums of the 19 uniform constraints' linear combinations (src/zkvm/r1cs/constraints.zig:248-531)
 * — AFFINE maps of the cycle's R1CS inputs. Generic form, over the same cycle-major matrix zg_fr_rows_mle reads (k <= 64 columns):
 *     table t, element i * g + j  =  C[t*g + j][k] + sum_{col < k} C[t*g + j][col] * rows[i*stride + col] i < n_rows
 *                                 =  0                                                                     n_rows <= i < n_pad
 * coeffs: (ntab * g) rows of k + 1 elements (the last one the constant), host memory; ntab * g <= 16. For the outer prover ntab = 2
 * (Az, Bz), g = 2 (the group selector is the lowest variable), coefficient rows (az0, az1, bz0, bz1), n_pad = the padded trace length.
 * stride: elements between the starts of consecutive rows, 0 = k. With stride < k a map is a SLIDING WINDOW over the matrix — row i's
 * columns stride .. k-1 are the first columns of the rows after it, the matrix must hold (n_rows - 1) * stride + k elements — which is how
 * ProductVirtualRemainderProver's fused right factor reads the NEXT cycle's IsNoop flag (src/zkvm/spartan/product_remainder.zig:436-476). */
ZG_API int zg_fr_rows_affine(const uint64_t *rows, size_t n_rows, size_t k, size_t stride, const uint64_t *coeffs, size_t ntab, size_t g, size_t n_pad,
                      uint64_t *const *tables /* ntab host pointers, n_pad * g elements each */);
ZG_API int zg_fr_rows_affine_dev(const uint64_t *d_rows, size_t n_rows, size_t k, size_t stride, const uint64_t *coeffs_host, size_t ntab, size_t g,
                          size_t n_pad, uint64_t *const *d_tables /* host array of ntab DEVICE pointers */, void *stream);
/* The same affine maps written as RECORDS: d_out[(i * record + first + c) * 4 ..] = map_c(row_i) for c < nout <= 16 consecutive positions of a
 * record of `record` elements per row. JoltR1CS.computeAz / computeBz (src/zkvm/r1cs/jolt_r1cs.zig:143-190) lay the 19 uniform constraints
 * of a cycle out this way (constraint_idx = cycle * 19 + i): two calls per vector (constraints 0..15, 16..18) fill it in place; the zero
 * padding up to the power of two is the caller's (a cleared buffer). coeffs_host: nout x (k + 1) elements, the constant last. */
ZG_API int zg_fr_rows_affine_records_dev(const uint64_t *d_rows, size_t n_rows, size_t k, size_t stride, const uint64_t *coeffs_host, size_t nout,
                                         size_t record, size_t first, uint64_t *d_out, void *stream);
/* LtPolynomial over the cube (src/zkvm/ram/val_evaluation.zig:289-330), the third factor of ValEvaluationProver's inc * wa * lt:
 * out[j] = sum over the ZERO bits i of j of r[i] * prod_{k > i} (bit_k(j) ? r[k] : 1 - r[k]); index bit i <-> r[i]. v <= 30. */
/* The witness matrix of a stage built ON THE DEVICE from integer columns. The reference derives every R1CS input of a cycle from machine
 * integers — F.fromU64 of a register / memory / pc value, signedI64ToField of an immediate, a 0/1 flag, the field product of two inputs
 * (R1CSCycleInputs.fromTraceStep, src/zkvm/r1cs/constraints.zig:929-1223, signedI64ToField :868-876) — and the provers then read the
 * 32-byte elements (R1CSInputEvaluator.computeClaimedInputs, src/zkvm/r1cs/evaluation.zig:55-122; StreamingOuterProver,
 * src/zkvm/spartan/streaming_outer.zig:258-372). A host hands the columns over as they are (~180 instead of 1376 bytes per cycle
 * across PCIe) and d_rows[row * n_cols + c] = the column's value at `row` as a canonical Montgomery Fr element, bit for bit what the
 * reference's conversions produce. One array per column, n_rows values each:
 *   ZG_COL_ZERO  no data: the column is zero              ZG_COL_U8 / U32 / U64  unsigned integers -> F.fromU64
 *   ZG_COL_I64   int64_t: v < 0 -> r - |v|                ZG_COL_I128 / U128     16 bytes per row (lo, hi), two's complement / unsigned
 *   ZG_COL_FR    a Montgomery element per row, copied     ZG_COL_BIT             bit `a` of a packed flag word of `b` = 1, 4 or 8 bytes
 *   ZG_COL_MUL   the product of columns a and b of the same row, plus — when data is given — a           per row -> 0 or F.one()
 *                128-bit two's-complement addend per row. A factor may itself be a ZG_COL_MUL column whose factors are plain columns
 *                (RightLookupOperand = Product * FlagMultiplyOperands + the sum / difference of the other rows), no deeper.
 *   ZG_COL_LUT   data = an index per row (a = 1, 2 or 4 bytes each), aux = a table of b Montgomery elements: the value is table[index]
 *                (an index >= b reads as zero) — eq_evals[j] = table[rd(j)] of proveStage5, src/zkvm/prover.zig:880-900
 * Several ZG_COL_BIT columns may name the same word array (it crosses once). n_cols <= 64. */
#define ZG_COL_ZERO 0
#define ZG_COL_U8 1
#define ZG_COL_U32 2
#define ZG_COL_U64 3
#define ZG_COL_I64 4
#define ZG_COL_I128 5
#define ZG_COL_U128 6
#define ZG_COL_FR 7
#define ZG_COL_BIT 8
#define ZG_COL_MUL 9
#define ZG_COL_LUT 10
typedef struct {
    uint32_t kind; /* ZG_COL_* */
    uint32_t a, b; /* ZG_COL_BIT: bit index, bytes per word; ZG_COL_MUL: the two factor columns (data: optional addend); ZG_COL_LUT: bytes per index, table length */
    const void *data;
    const void *aux; /* ZG_COL_LUT: the table (b elements); NULL otherwise */
} zg_col_t;
/* host columns (pageable, or pinned from zg_host_alloc); returns when the matrix is complete */
ZG_API int zg_fr_rows_from_columns(const zg_col_t *cols, size_t n_cols, size_t n_rows, uint64_t *d_rows /* device, n_rows*n_cols*4 */);
/* columns already in HBM (data = device addresses); one asynchronous launch */
ZG_API int zg_fr_rows_from_columns_dev(const zg_col_t *cols, size_t n_cols, size_t n_rows, uint64_t *d_rows, void *stream);

ZG_API int zg_fr_lt_table(const uint64_t *r, size_t v, uint64_t *out /* 2^v * 4 */);
ZG_API int zg_fr_lt_table_dev(const uint64_t *r_host, size_t v, uint64_t *d_out, void *stream);
/* ValEvaluation's other two tables from the LIST OF WRITES (the prover's stage 4, src/zkvm/prover.zig:760-800 over
 * ram/val_evaluation.zig:298-345): d_inc[cycle[i]] = F.fromU64(post[i]) - F.fromU64(pre[i]) and d_wa[cycle[i]] = eq(r_eq, word[i] mod 2^log_k)
 * for the m writes, zero elsewhere (n entries each, DEVICE). r_eq is the point in EqPolynomial order (the reference passes r_address
 * reversed). The cycles must be distinct — a repeated or out-of-range cycle is refused (ZG_ERR_INVALID); the caller keeps the last write of a
 * cycle, as the reference's loop does by overwriting.
 * Returns after the work has completed: 24 bytes per write cross the boundary instead of two n-element tables. */
ZG_API int zg_fr_write_tables_dev(size_t n, size_t m, const uint32_t *cycle, const uint32_t *word, const uint64_t *pre, const uint64_t *post,
                           const uint64_t *r_eq, size_t log_k, uint64_t *d_inc, uint64_t *d_wa, void *stream);
/* The Q tables of Stage 3's prefix / suffix provers (ShiftPrefixSuffixProver.init, src/zkvm/spartan/stage3_prover.zig:1066-1112;
 * RegistersPrefixSuffixProver.init, :2232-2290): Q[x_lo] = sum over x_hi of witness(x_lo + x_hi * 2^prefix_vars) * suffix[x_hi] — column
 * sums of the cycle-length table read as a (rows = 2^suffix_vars) x (cols = 2^prefix_vars) matrix, under up to four weight vectors at
 * once (eq and eq+1 suffixes share the pass):   out[k * cols + c] = sum_r weights[k * rows + r] * table[r * cols + c],  k < m <= 4.
 * The _dev form is enqueued on `stream`; when the rows are cut into slabs (more than one) it also waits for the stream, because the
 * slab partials live in the library's scratch cache. */
ZG_API int zg_fr_weighted_colsum(const uint64_t *table, size_t rows, size_t cols, const uint64_t *weights, size_t m, uint64_t *out);
ZG_API int zg_fr_weighted_colsum_dev(const uint64_t *d_table, size_t rows, size_t cols, const uint64_t *d_weights, size_t m, uint64_t *d_out,
                              void *stream);
/* StreamingOuterProver.computeFirstRoundPoly's extended evaluations (src/zkvm/spartan/streaming_outer.zig:523-597, evaluateAzBzAtTargetY
 * :599-671): for each UniSkip target Y_j and constraint group, Az(x, Y_j) and Bz(x, Y_j) are Lagrange extrapolations (COEFFS_PER_J,
 * src/zkvm/r1cs/univariate_skip.zig:469-476) of the group's constraint values — affine maps of the cycle's inputs — and
 * t1(Y_j) = sum_{cycle, group} eq(tau_low, (cycle, group)) * Az * Bz. Generic form over the cycle-major matrix:
 *     out[p] = sum_{i < n_rows} W[i * g + p % g] * A_p(rows[i]) * B_p(rows[i]),     p < npairs <= 32,
 * A_p / B_p = coefficient rows 2p / 2p + 1 (k + 1 elements each, the constant last; host memory); W: DEVICE table of n_rows * g weights
 * (for the outer prover g = 2 and W = zg_fr_eq_table_dev(tau_low): index = cycle * 2 + group; pairs ordered (target, group)). */
ZG_API int zg_fr_rows_affine_prodsum_dev(const uint64_t *d_rows, size_t n_rows, size_t k, size_t stride, const uint64_t *coeffs_host, size_t npairs,
                                  const uint64_t *d_weights, size_t g, uint64_t *out /* host, npairs * 4 */, void *stream);
/* DensePolynomial.bindLow, in place: t[i] = t[2i] + r*(t[2i+1]-t[2i]), len -> len/2 (src/poly/mod.zig:160-175) */
ZG_API int zg_fr_bind_low(uint64_t *table, size_t len, const uint64_t r[4]);
/* DensePolynomial.bindFirst: out[i] = (1-r)*t[i] + r*t[i+len/2] (src/poly/mod.zig:128-149) */
ZG_API int zg_fr_bind_high(const uint64_t *table, size_t len, const uint64_t r[4], uint64_t *out);
/* Spartan combine f[i] = eq[i]*(Az[i]*Bz[i] - Cz[i]) (src/zkvm/spartan/mod.zig:191-199) */
ZG_API int zg_fr_spartan_combine(const uint64_t *eq, const uint64_t *az, const uint64_t *bz, const uint64_t *cz, size_t n,
                          uint64_t *out);
ZG_API int zg_fr_spartan_combine_dev(const uint64_t *d_eq, const uint64_t *d_az, const uint64_t *d_bz, const uint64_t *d_cz,
                              size_t n, uint64_t *d_out, void *stream);

/* ------------------------------------------------------------------ sumcheck session */
/* A device-resident Sumcheck(F).Prover (src/subprotocols/mod.zig:50-134): the table stays in
 * HBM across rounds; per round only two field elements come back and one goes in.
 * layout ZG_SC_HIGH_HALF = bindFirst order (the generic prover, :112-122);
 *        ZG_SC_LOW_PAIR  = bindLow order (src/poly/mod.zig:160-175; zkvm/r1cs/jolt_r1cs.zig:470-477). */
#define ZG_SC_HIGH_HALF 0
#define ZG_SC_LOW_PAIR 1
typedef struct zg_sc_s *zg_sc_t;
ZG_API int zg_sumcheck_open(const uint64_t *evals, size_t len, int layout, zg_sc_t *s);
ZG_API int zg_sumcheck_open_dev(const uint64_t *d_evals, size_t len, int layout, void *stream, zg_sc_t *s); /* copies */
/* The same session WITHOUT the copy: the session reads the caller's table (never writes it) until its first bind has run — the caller
 * keeps d_evals alive and unchanged until a later call on the session has returned round sums or a final value (or zg_sync). Its own
 * buffers hold the folds only (len/2 + len/4 entries instead of len + len/2): Az and Bz of an outer sumcheck, 1 GB each at 2^20 cycles,
 * are bound alongside without a second copy in HBM. zg_sumcheck_bit_bind (an in-place bind) is refused before the first zg_sumcheck_bind. */
ZG_API int zg_sumcheck_open_dev_borrowed(const uint64_t *d_evals, size_t len, int layout, void *stream, zg_sc_t *out);
/* A session whose table is ONE integer column widened on the device (zg_fr_rows_from_columns' kinds; host data): entries [0, n_rows) from
 * the column, [n_rows, len) zero. The tables of proveStage5 (eq_evals[j] = table[rd of cycle j]: a ZG_COL_LUT over one byte per cycle) and
 * proveStage6 (all zero: ZG_COL_ZERO) are built this way without a 32-byte-per-cycle upload (src/zkvm/prover.zig:880-944, 1024-1097). */
ZG_API int zg_sumcheck_open_column(const zg_col_t *col, size_t n_rows, size_t len, int layout, zg_sc_t *s);
/* Spartan's first sumcheck instance in ONE pass (src/zkvm/spartan/mod.zig:182-206, then Sumcheck.Prover.init): opens a session
 * whose table is f[i] = eq(r, i) * (Az[i]*Bz[i] - Cz[i]) (eq as zg_fr_eq_table: r[0] <-> MSB, optional scale; Az, Bz, Cz: 2^v
 * device-resident elements; d_cz = NULL: Cz is identically zero, as JoltR1CS.computeCz leaves it, src/zkvm/r1cs/jolt_r1cs.zig:193-200)
 * with round 0's sums already computed; the eq table is never materialised, nothing is copied. */
ZG_API int zg_sumcheck_open_spartan_dev(const uint64_t *r /* host, v*4 */, size_t v, const uint64_t *scale /* host 4 or NULL */,
                                 const uint64_t *d_az, const uint64_t *d_bz, const uint64_t *d_cz, int layout, void *stream,
                                 zg_sc_t *s);
/* Prover.nextRound's two sums (:79-93): HIGH_HALF g0 = sum first half, g1 = sum second half;
 * LOW_PAIR g0 = sum even, g1 = sum odd (jolt_r1cs.zig:436-444). */
ZG_API int zg_sumcheck_round_sums(zg_sc_t s, uint64_t g0[4], uint64_t g1[4]);
/* Prover.receiveChallenge (:112-122): fold by r. The next round's sums are produced by the
 * same kernel and returned by the following zg_sumcheck_round_sums without another pass. */
ZG_API int zg_sumcheck_bind(zg_sc_t s, const uint64_t r[4]);
ZG_API size_t zg_sumcheck_len(zg_sc_t s);
ZG_API int zg_sumcheck_final(zg_sc_t s, uint64_t out[4]); /* getFinalEval (:130-133), needs len == 1 */
ZG_API int zg_sumcheck_read(zg_sc_t s, uint64_t *out_table); /* copy the current table to the host (tests) */
/* out[i] = table[idx[i]], i < n (every idx < the current length): what a SPARSE-entry prover reads of a dense table —
 * RamReadWriteCheckingProver's val_init checkpoints of the column pairs its entries touch (src/zkvm/ram/read_write_checking.zig:591-602,
 * 985-996) — instead of copying the table back every round. */
ZG_API int zg_sumcheck_gather(zg_sc_t s, const uint64_t *idx, size_t n, uint64_t *out /* n*4 */);
/* Sharded tables: the LOCAL pair g0||g1 (8 limbs) / the current local table, written to device
 * memory in stream order on the session's stream (the one given to zg_sumcheck_open_dev) with no host
 * synchronisation, so an RCCL all-gather enqueued on that stream can follow directly. */
ZG_API int zg_sumcheck_round_sums_dev(zg_sc_t s, uint64_t *d_out8);
ZG_API int zg_sumcheck_read_dev(zg_sc_t s, uint64_t *d_out_table);
ZG_API int zg_sumcheck_close(zg_sc_t s);
/* Prover fold sites beyond the generic Sumcheck.Prover. On a LOW_PAIR session holding RaPolynomial's table:
 * RafEvaluationProver.computeRoundPolynomialCubic's two sums (src/zkvm/ram/raf_checking.zig:335-410)
 *   s(0) = sum_i ra[2i] * u0(i),  s(2) = sum_i (2 ra[2i+1] - ra[2i]) * (u0(i) + 2 * current_power),
 *   u0(i) = base + 2 * current_power * i   (as field elements; `base` = start_address + 8 * sum_j bound_j 2^j, Montgomery),
 * in one pass over the table; s(1) = claim - s(0) and s(3) = s(0) - 3 s(1) + 3 s(2) are host scalar code, the fold of the round
 * (RaPolynomial.bind, :162-174) is zg_sumcheck_bind. */
ZG_API int zg_sumcheck_raf_round(zg_sc_t s, const uint64_t base[4], uint64_t current_power, uint64_t s0[4], uint64_t s2[4]);
/* RafEvaluationProver.computeInitialClaim (src/zkvm/ram/raf_checking.zig:312-321): claim = sum_k t[k] * F.fromU64(base + step * k) over the
 * session's current table (base = start_address, step = 8 for UnmapPolynomial :207-209); base + step * (len - 1) must fit 64 bits. */
ZG_API int zg_sumcheck_raf_claim(zg_sc_t s, uint64_t base, uint64_t step, uint64_t claim[4]);
/* LassoProver's eq_evals path on ONE session (src/zkvm/lasso/prover.zig:262-453): open the session with layout ZG_SC_HIGH_HALF over
 * the padded eq_evals (:153-171); the log_K address rounds use the two calls below, the log_T cycle rounds are the session's
 * ordinary round_sums / bind (:313-340,411-441). d_idx128: the u128 lookup indices in DEVICE memory (two little-endian u64 words
 * each; zg_dev_alloc + zg_memcpy_h2d), n_idx <= the session's length.
 *   bit_round = computeAddressRoundPoly's sum_0 / sum_1 (:283-293) over the first n_idx entries;
 *   bit_bind  = receiveChallenge's address branch (:375-399): entry j *= (bit of idx[j]) ? r : 1 - r, claim = sum of ALL entries.
 * bit_bind also leaves the next round's sums (bit + 1) behind, so the following bit_round costs no pass over the table (they are
 * keyed by d_idx128, n_idx and the bit: the index buffer's CONTENTS must stay unchanged between the two calls, as a prover's do). */
ZG_API int zg_sumcheck_bit_round(zg_sc_t s, const uint64_t *d_idx128, size_t n_idx, unsigned bit, uint64_t sum0[4], uint64_t sum1[4]);
ZG_API int zg_sumcheck_bit_bind(zg_sc_t s, const uint64_t *d_idx128, size_t n_idx, unsigned bit, const uint64_t r[4], uint64_t claim[4]);
/* LassoProver.computeAddressRoundPoly's two sums (src/zkvm/lasso/prover.zig:283-293): sum0 / sum1 = the sum of vals[j] over the
 * entries whose u128 lookup index (idx128: n x 2 little-endian u64 words) has bit `bit` clear / set. The _dev form keeps eq_evals
 * and the indices resident across the LOG_K address rounds. */
ZG_API int zg_fr_bit_split_sums(const uint64_t *vals, const uint64_t *idx128, size_t n, unsigned bit, uint64_t sum0[4], uint64_t sum1[4]);
ZG_API int zg_fr_bit_split_sums_dev(const uint64_t *d_vals, const uint64_t *d_idx128, size_t n, unsigned bit, void *stream, uint64_t sum0[4],
                             uint64_t sum1[4]);

/* Self-test of the round-ending hand-off every sumcheck-family kernel uses (block partials -> the workgroup that arrives last; the
 * reference's serial sums of src/subprotocols/mod.zig:79-93 have no counterpart): `iters` launches of `blocks` workgroups of `threads`
 * threads on the library's stream. Each workgroup first reads every partial line with PLAIN loads (its L1 then holds the previous
 * launch's bytes), waits an uneven, launch-dependent time, and hands in a pattern unique to (launch, workgroup); the last arriver
 * checks EVERY word it reads against the pattern. With busy != 0 a second stream streams through a 256 MiB buffer meanwhile.
 * Outputs: words that differed (must be 0) and launches that elected exactly one last arriver (must be iters). */
ZG_API int zg_selftest_handoff(unsigned blocks, unsigned threads, unsigned iters, int busy, uint64_t *mismatches, uint64_t *completed);

/* runSumcheck (src/subprotocols/mod.zig:302-354) with the WHOLE protocol on the device: the prover's sums and folds
 * (bindFirst order) and the reference's toy verifier (verifyRound / deriveChallenge, :165-243: the deterministic
 * 64-bit mixer, F.fromU64, claim <- p(challenge)) — the verifier step runs at the end of the kernel that produced the
 * round's sums, so no round crosses PCIe. len = 2^v (v >= 0). Outputs (host): claim[4]; rounds: v x (c0 || c1) =
 * Round.poly.coeffs [g(0), g(1) - g(0)]; challenges: v x 4 (= Proof.final_point); final_eval[4]; *result =
 * verifier.claim == final_eval. Returns ZG_ERR_VERIFY where the reference returns error.SumcheckVerificationFailed.
 * The input table is not modified. Provers driven by a host transcript (Keccak/Blake2b) use the session API above. */
ZG_API int zg_run_sumcheck_dev(const uint64_t *d_evals, size_t len, void *stream, uint64_t claim[4], uint64_t *rounds,
                               uint64_t *challenges, uint64_t final_eval[4], uint8_t *result);
ZG_API int zg_run_sumcheck(const uint64_t *evals, size_t len, uint64_t claim[4], uint64_t *rounds, uint64_t *challenges,
                           uint64_t final_eval[4], uint8_t *result);

/* ------------------------------------------------------------------ product-form sumcheck sessions */
/* The zkVM provers whose round polynomial is a sum over adjacent pairs of a PRODUCT of multilinear tables, all folded LowToHigh by
 * the same challenge — ValEvaluationProver (src/zkvm/ram/val_evaluation.zig:554-628: inc * wa * lt), ValFinalProver
 * (ram/val_final.zig:149-200: inc * wa), OutputSumcheckProver (ram/output_check.zig:375-470: eq * io * (vf - vio)),
 * InstructionLookupsClaimReduction (claim_reductions/instruction_lookups.zig:146-240: eq * (out + gamma left + gamma^2 right)) and the
 * Gruen-form ProductVirtualRemainderProver (spartan/product_remainder.zig:269-420: left * right under split-eq weights). A session keeps
 * k <= 12 tables of `len` entries (a power of two) resident in HBM; the transcript, the claim update and the Gruen scalar algebra stay
 * on the host. */
typedef struct zg_psc_s *zg_psc_t;
ZG_API int zg_psc_open(const uint64_t *const *tables /* k host pointers, len*4 words each */, size_t k, size_t len, zg_psc_t *s);
ZG_API int zg_psc_open_dev(const uint64_t *const *d_tables /* host array of k device pointers */, size_t k, size_t len, void *stream,
                    zg_psc_t *s); /* copies. stream = NULL: the session runs on its own stream; the copies follow the work already on the
                                   * library stream (the *_dev builders' default) and are complete on return. Otherwise the session
                                   * lives on `stream` and the copies are only enqueued there. */
ZG_API size_t zg_psc_len(zg_psc_t s);
ZG_API size_t zg_psc_tables(zg_psc_t s);
/* out[t] (t = 0..3, 4 words each) = sum over the pairs g of  prod_{j<p} T[prod_idx[j]](t) * L(t), where T(t) = T[2g] + t (T[2g+1] - T[2g])
 * and L(t) = sum_{m<q} lin_coeff[m] * T[lin_idx[m]](t) (q = 0: L = 1). p, q <= 4, p + q >= 1. These are the reference's
 * [p(0), p(1), p(2), p(3)] (val_evaluation.zig:554-603); provers that send fewer values drop the rest. */
ZG_API int zg_psc_round_evals(zg_psc_t s, const int *prod_idx, size_t p, const int *lin_idx, const uint64_t *lin_coeff /* q*4 */, size_t q,
                       uint64_t out[16]);
/* A SUM of up to four such product terms in one pass — the round polynomials of stage3_prover.zig: ShiftSumcheck phase 1 (four P*Q
 * pairs, :1351-1392) and phase 2 (:1399-1455), InstructionInput ((eq_out + g^2 eq_prod) * (is_rs2*rs2 + is_imm*imm + g (is_rs1*rs1 +
 * is_pc*pc)), :2029-2100: four terms of two plain factors and a two-table combination). out as zg_psc_round_evals; provers that
 * derive p(1) from the claim drop it. */
#define ZG_PSC_PAIR_SUM 256 /* n_prod = 4 | ZG_PSC_PAIR_SUM: the term is (T[prod[0]]*T[prod[1]] + T[prod[2]]*T[prod[3]]) * L (L = 1 when
                             * n_lin = 0) — two terms with the same weight as one (InstructionInput: is_rs2*rs2 + is_imm*imm under one
                             * weight; ShiftSumcheck phase 1: P_0*Q_0 + P_1*Q_1): three field products per evaluation point, not four */
typedef struct {
    int n_prod;
    int prod[4];
    int n_lin;
    int lin[4];
    uint64_t lin_coeff[16]; /* n_lin x 4 words, Montgomery */
} zg_psc_term;
ZG_API int zg_psc_round_expr(zg_psc_t s, const zg_psc_term *terms, size_t n_terms /* 1..4 */, uint64_t out[16]);
/* The evaluation points the following zg_psc_round_evals / zg_psc_round_expr calls (and the evaluations fused into zg_psc_bind)
 * compute: bit t of `points` = p(t) is wanted; the other slots of out[] come back as zero and their field products are not spent.
 * Several reference provers read p(0) and p(2) only and derive p(1) = claim - p(0), p(3) = p(0) - 3 p(1) + 3 p(2)
 * (claim_reductions/instruction_lookups.zig:146-200; stage3_prover.zig:1399-1455, 2334-2389) or all but p(1) (:2029-2100).
 * Default 0xF (all four); a session handed out again by zg_psc_open starts at 0xF. */
ZG_API int zg_psc_set_points(zg_psc_t s, unsigned points /* 1..15 */);
/* Gruen's pair (product_remainder.zig:281-330): t0 = sum_g w(g) prod_j T_j[2g], t_inf = sum_g w(g) prod_j (T_j[2g+1] - T_j[2g]),
 * w(g) = E_out[g >> log2|E_in|] * E_in[g & (|E_in| - 1)], pairs with g >> log2|E_in| >= |E_out| skipped. d_e_out / d_e_in: DEVICE
 * pointers (e.g. into the buffer zg_fr_eq_prefix_tables_dev filled: table k starts at element 2^k - 1). */
ZG_API int zg_psc_round_gruen(zg_psc_t s, const int *prod_idx, size_t p, const uint64_t *d_e_out, size_t n_out, const uint64_t *d_e_in,
                       size_t n_in, uint64_t t0[4], uint64_t t_inf[4]);
/* every table folded: T'[i] = (1 - r) T[2i] + r T[2i+1]; len -> len / 2 (val_evaluation.zig:609-628) */
ZG_API int zg_psc_bind(zg_psc_t s, const uint64_t r[4]);
ZG_API int zg_psc_read(zg_psc_t s, size_t table, uint64_t *out /* len*4 */);
/* the table where it lies in HBM (zg_psc_len(s) * 4 words), after the session's pending folds have completed: valid until the next
 * bind / close of this session. For handing folded tables to another session or kernel without a host round trip (Stage 3's phase
 * transitions, src/zkvm/spartan/stage3_prover.zig:1506-1700, 2427-2466, re-open their provers over the folded witness columns). */
ZG_API int zg_psc_table_dev(zg_psc_t s, size_t table, const uint64_t **d_ptr);
/* out[i] = T[table][idx[i]], i < n: the inc values of the row pairs a sparse prover's entries touch (read_write_checking.zig:431-446) */
ZG_API int zg_psc_gather(zg_psc_t s, size_t table, const uint64_t *idx, size_t n, uint64_t *out /* n*4 */);
ZG_API int zg_psc_final(zg_psc_t s, uint64_t *out /* k*4: each table's single remaining entry */);
ZG_API int zg_psc_close(zg_psc_t s);

/* ================================================================== OPTIONAL section: protocol-specific sessions */
#ifndef ZG_NO_PROTOCOL_SESSIONS
/* ------------------------------------------------------------------ registers read/write checking (Stage 4) */
/* Stage4GruenProver (src/zkvm/spartan/stage4_gruen_prover.zig:65-1240) and the original Stage4Prover (stage4_prover.zig:74-865: the same
 * tables, every cycle variable first under a dense eq table set at the start, all four evaluations computed directly), the
 * RegistersReadWriteChecking sumcheck: five DENSE tables of
 * K = 128 registers x T = 2^log_t cycles — val, rd_wa, ra = gamma rs1_ra + gamma^2 rs2_ra, rs1_ra, rs2_ra, indexed [k * T + j] — plus
 * inc[T], LOG_K + log_t rounds in three phases (cycle variables in Gruen form, the seven register variables, the remaining cycle
 * variables under the merged dense eq table). The session builds the tables on the device from the per-cycle trace columns and keeps
 * them in HBM — the 32 rows that can be non-zero (register indices are 5-bit fields; rows 32..127 are zero in the reference's tables
 * and are neither stored nor visited here); every round is one pass (sums) + one pass (folds). The eq structure, Gruen's cubic, the claim and the transcript stay on
 * the host. Values are the reference's, exactly.
 *   rs1 / rs2 / rd [T]: the register a cycle reads / writes, 0xFF = none — decided as initWithPhaseConfig does (:199-246: by opcode; rd
 *                        only when it is used and non-zero); padding cycles are 0xFF;
 *   reg_vals [32 * T] : value of register k before cycle j at [k * T + j] (:186-192, 249-258);  inc [T * 4]: F.fromU64(post) - F.fromU64(pre)
 *                        of the written register, zero elsewhere (:240-243);  gamma: the batching challenge. */
typedef struct zg_rrw_s *zg_rrw_t;
ZG_API int zg_rrw_open(size_t log_t, const uint8_t *rs1, const uint8_t *rs2, const uint8_t *rd, const uint64_t *reg_vals, const uint64_t *inc,
                const uint64_t gamma[4], zg_rrw_t *s);
/* the same session from the write column alone: rd_value [T] = what cycle j writes to register rd[j] (ignored where rd[j] = 0xFF). The
 * register file before every cycle (the last write to each register at an earlier cycle, zero before the first) and inc are rebuilt on
 * the device — 11 bytes per cycle cross the boundary instead of 291, and the host keeps no 32 x T table (stage4_gruen_prover.zig:183-258). */
ZG_API int zg_rrw_open_trace(size_t log_t, const uint8_t *rs1, const uint8_t *rs2, const uint8_t *rd, const uint64_t *rd_value,
                      const uint64_t gamma[4], zg_rrw_t *s);
ZG_API size_t zg_rrw_cycles(zg_rrw_t s);    /* current_T */
ZG_API size_t zg_rrw_registers(zg_rrw_t s); /* current_K */
/* phase1ComputeMessage's pair (:561-741): q0 = sum_i E_out[i >> log2|E_in|] E_in[i & (|E_in| - 1)] sum_k C_0(k, i), qX2 likewise with the
 * slopes; C = ra val + wa (val + inc) over the cycle pair (2i, 2i + 1). d_e_out / d_e_in: DEVICE tables (zg_fr_eq_prefix_tables_dev). */
ZG_API int zg_rrw_round_cycle_gruen(zg_rrw_t s, const uint64_t *d_e_out, size_t n_out, const uint64_t *d_e_in, size_t n_in, uint64_t q0[4], uint64_t qx2[4]);
/* the merged eq table of gruen_eq.merge (gruen_eq.zig:119-146) after the last phase-1 bind: n = zg_rrw_cycles(s) entries (host) */
ZG_API int zg_rrw_set_eq(zg_rrw_t s, const uint64_t *eq, size_t n);
/* phase2ComputeMessage's (eval_0, eval_2) (:764-852; also the register rounds once a single cycle is left, :955-1013). e1 != NULL: the value
 * at t = 1 as well, computed from the odd rows — Stage4Prover.computeRoundEvalsInternal's register rounds evaluate it directly
 * (src/zkvm/spartan/stage4_prover.zig:666-706) instead of taking it from the claim. */
ZG_API int zg_rrw_round_address(zg_rrw_t s, uint64_t e0[4], uint64_t *e1 /* 4 words, or NULL */, uint64_t e2[4]);
/* phase3ComputeMessage's (eval_0, eval_2, eval_3) (:854-953); e1 != NULL: p(1) too, as Stage4Prover's cycle rounds compute all four
 * (stage4_prover.zig:617-664) */
ZG_API int zg_rrw_round_cycle(zg_rrw_t s, uint64_t e0[4], uint64_t *e1 /* 4 words, or NULL */, uint64_t e2[4], uint64_t e3[4]);
/* bindPolynomials (:1047-1163): fold the cycle variable (five tables, inc, and the merged eq table once it is set) / the register variable */
ZG_API int zg_rrw_bind_cycle(zg_rrw_t s, const uint64_t r[4]);
ZG_API int zg_rrw_bind_address(zg_rrw_t s, const uint64_t r[4]);
/* entry [0][0] of val, rd_wa, ra, rs1_ra, rs2_ra, then inc[0] and merged_eq[0] (7 x 4 words): getFinalClaims (:1219-1236) */
ZG_API int zg_rrw_final(zg_rrw_t s, uint64_t *out /* 28 */);
ZG_API int zg_rrw_close(zg_rrw_t s);

/* ------------------------------------------------------------------ RAM read/write checking (Stage 2, the sparse instance) */
/* RamReadWriteCheckingProver (src/zkvm/ram/read_write_checking.zig:160-1323): a list of access entries {cycle, address, ra_coeff, val_coeff,
 * prev_val, next_val} (:91-157) beside three dense tables — eq_evals = eq(r_cycle, .) and inc over 2^log_t cycles, val_init over 2^log_k
 * words — and log_t + log_k rounds in three phases: phase1_num_rounds cycle variables, the address variables, the remaining cycle
 * variables. The session owns everything: the reference's walks over the integer fields of the list (who pairs with whom, which
 * checkpoint a lone entry meets) run once per round inside the library — the cycle phases' neighbour rule on the device, the address
 * phase's sequential column walk on the host; ra_coeff, val_coeff and the dense tables live in HBM, where one kernel turns a round's
 * walk into its two sums and one into the bound list. The caller keeps what the
 * reference's struct keeps besides: the GruenSplitEqPolynomial (its prefix tables as device buffers), the cubic, the claim.
 *   entries: sorted by (cycle, address) as init leaves them (:333-340); val_coeff as u64 (F.fromU64 of prev_val for a write, of the value
 *   for a read, :300-330), ra_coeff = 1; inc / val_init: field elements (:253-330); r_cycle: the eq point, r_cycle[0] <-> MSB (:345-348). */
typedef struct zg_rwc_s *zg_rwc_t;
ZG_API int zg_rwc_open(size_t log_k, size_t log_t, size_t n, const uint32_t *cycle, const uint32_t *address, const uint64_t *val_coeff,
                const uint64_t *prev_val, const uint64_t *next_val, const uint64_t *inc /* 2^log_t * 4 */, const uint64_t *val_init /* 2^log_k * 4 */,
                const uint64_t *r_cycle /* log_t * 4 */, zg_rwc_t *s);
/* the same session without the dense inc table crossing the boundary: is_write [n] marks the entries that are writes, and the device
 * forms inc[cycle] = F.fromU64(next_val) - F.fromU64(prev_val) for them (zero elsewhere) — what init's loop assigns (:283-291). A list
 * with two writes in one cycle is refused (ZG_ERR_INVALID): the reference keeps the later one in ACCESS order; use zg_rwc_open then. */
ZG_API int zg_rwc_open_writes(size_t log_k, size_t log_t, size_t n, const uint32_t *cycle, const uint32_t *address, const uint64_t *val_coeff,
                       const uint64_t *prev_val, const uint64_t *next_val, const uint8_t *is_write, const uint64_t *val_init /* 2^log_k * 4 */,
                       const uint64_t *r_cycle /* log_t * 4 */, zg_rwc_t *s);
ZG_API size_t zg_rwc_entries(zg_rwc_t s); /* current length of the list */
ZG_API size_t zg_rwc_cycles(zg_rwc_t s);  /* current length of eq_evals / inc */
/* computePhase1Polynomial's (q_constant, q_quadratic) (:410-536) under the split-eq weights E_out x E_in (DEVICE tables, as
 * zg_psc_round_gruen takes them); cycle phases 1 and 3 */
ZG_API int zg_rwc_round_cycle(zg_rwc_t s, const uint64_t *d_e_out, size_t n_out, const uint64_t *d_e_in, size_t n_in, const uint64_t gamma[4],
                       uint64_t q_constant[4], uint64_t q_quadratic[4]);
/* bindChallenge in a cycle phase (:919-938, 1139-1185): eq_evals and inc LowToHigh, the entry list pairwise. Needs zg_rwc_cycles(s) >= 2
 * (the reference skips the whole bind otherwise, :915). */
ZG_API int zg_rwc_bind_cycle(zg_rwc_t s, const uint64_t r[4]);
/* computePhase2Polynomial's (s(0), s(2)) (:538-769) in address round addr_round = round - phase1_num_rounds; challenges: the address
 * challenges bound so far (addr_round elements, host). The first call sorts the list address-major (:553-558). */
ZG_API int zg_rwc_round_address(zg_rwc_t s, size_t addr_round, const uint64_t *challenges, const uint64_t gamma[4], uint64_t s0[4], uint64_t s2[4]);
/* bindChallenge in the address phase (:953-970, 973-1137): val_init LowToHigh, then the entry list by column pairs */
ZG_API int zg_rwc_bind_address(zg_rwc_t s, size_t addr_round, const uint64_t r[4]);
/* getOpeningClaims (:1210-1322): ra_claim, val_claim, inc_claim (3 x 4 words) at r_address (log_k, [0] <-> MSB) and r_cycle (log_t) */
ZG_API int zg_rwc_opening(zg_rwc_t s, const uint64_t *r_address, const uint64_t *r_cycle, uint64_t out[12]);
/* eq_evals[0] and inc[0] as they stand (the two scalars the address phase works with, :543-552) */
ZG_API int zg_rwc_cycle_scalars(zg_rwc_t s, uint64_t eq0[4], uint64_t inc0[4]);
/* the current list (any pointer may be NULL); the coefficient columns come from the device */
ZG_API int zg_rwc_read_entries(zg_rwc_t s, uint32_t *cycle, uint32_t *address, uint64_t *ra_coeff, uint64_t *val_coeff, uint64_t *prev_val, uint64_t *next_val);
ZG_API int zg_rwc_close(zg_rwc_t s);

#endif /* ZG_NO_PROTOCOL_SESSIONS */
/* ================================================================== end of the OPTIONAL section */

/* ------------------------------------------------------------------ several GPUs in one process */
/* The bases (SRS) sharded over the bound devices in ParallelMSM's contiguous chunks of ceil(n / S) points
 * (src/msm/mod.zig:609,619-639), one resident table per device. */
typedef struct zg_sbases_s *zg_sbases_t;
/* the partition itself — chunk `shard` of `shards` over n points: start = min(shard * ceil(n / shards), n), end = min(start +
 * ceil(n / shards), n) (src/msm/mod.zig:609,619-621). Pure host arithmetic, needs no device. */
ZG_API int zg_shard_bounds(size_t n, int shards, int shard, size_t *start, size_t *len);
ZG_API int zg_g1_bases_upload_sharded(const uint64_t *xy, const uint8_t *inf, size_t n, const zg_msm_config *cfg, zg_sbases_t *out);
ZG_API int zg_g1_sbases_free(zg_sbases_t sb);
ZG_API size_t zg_g1_sbases_len(zg_sbases_t sb);
ZG_API int zg_g1_sbases_shards(zg_sbases_t sb);   /* number of shards S */
ZG_API int zg_g1_sbases_exchange(zg_sbases_t sb); /* how partials meet: 0 = single shard, 1 = RCCL all-gather, 2 = peer copies */
ZG_API int zg_g1_sbases_shard(zg_sbases_t sb, int shard, int *device, size_t *start, size_t *len);
/* ParallelMSM.compute(bases[0..n), scalars) (src/msm/mod.zig:588-653) over the devices: every device runs the Pippenger pipeline
 * on its chunk (one host worker thread per device issues its launches), the S Jacobian partials (96 B each, un-normalised) are
 * exchanged with ONE ncclAllGather over xGMI (RCCL has no user-defined reduction for group elements) and device 0 performs the
 * serial combine + the single toAffine (:647-652). Result bytes equal zg_msm_g1's. Scalars: n x 4 limbs on the host. */
ZG_API int zg_msm_g1_sharded(zg_sbases_t sb, size_t n, const uint64_t *scalars_mont, uint64_t out_xy[8], uint8_t *out_inf);
/* same with the scalars resident: d_scalars_per_shard[i] points into device memory of shard i's device and holds the scalars of
 * that shard's chunk (zg_g1_sbases_shard tells which). The buffers are read on the handle's own streams, which are not ordered
 * with the streams of the caller: their contents must be complete (synchronised) when the call is made. */
ZG_API int zg_msm_g1_sharded_dev(zg_sbases_t sb, size_t n, const uint64_t *const *d_scalars_per_shard, uint64_t out_xy[8],
                          uint8_t *out_inf);
/* ParallelBatchMSM.compute / HyperKZG.batchCommit (src/msm/mod.zig:683-748, src/poly/commitment/mod.zig:558-570) sharded: k scalar
 * vectors over bases[0..n) -> k partials per device, ONE all-gather of k x 96 B per device, k combines. */
ZG_API int zg_msm_g1_batch_sharded(zg_sbases_t sb, size_t n, const uint64_t *const *scalar_batches, size_t k, uint64_t *out_xy /* k*8 */,
                            uint8_t *out_inf /* k */);
/* Pipelined forms: the call returns a ticket as soon as every device's launch set, the exchange and the combine are ENQUEUED;
 * zg_sharded_wait(ticket) blocks until that call's result records have reached the host and hands them out (out_xy: k*8 words,
 * out_inf: k bytes or NULL; k = 1 for zg_msm_g1_sharded_dev_async). A handle keeps zg_g1_sbases_inflight() calls in flight (3 by
 * default; ZG_SHARDED_INFLIGHT = 1..8 at upload), each on its own stream per device, so the latency-bound tail and the exchange of
 * one MSM run under the accumulation of the next — HyperKZG.batchCommit over long vectors (src/poly/commitment/mod.zig:558-570)
 * and `zolt prove`'s three commitments (src/zkvm/mod.zig:1538,1572,1607) overlap this way. One more call than that without a wait
 * returns ZG_ERR_INVALID. Tickets may be waited for in any order, each exactly once. The synchronous entry points above are
 * these + zg_sharded_wait.
 * Scalars: the host vectors of zg_msm_g1_batch_sharded_async must stay valid until the ticket is waited for. The device buffers
 * of zg_msm_g1_sharded_dev[_async] are read on the handle's own streams: they must be COMPLETE when the call is made, unless
 * ready_streams (S stream handles, shard order; NULL or NULL entries = complete) names for every shard the stream whose enqueued
 * work fills that shard's buffer — the shard's launch set is then ordered behind it with an event. */
ZG_API int zg_msm_g1_sharded_dev_async(zg_sbases_t sb, size_t n, const uint64_t *const *d_scalars_per_shard, void *const *ready_streams,
                                uint64_t *ticket);
ZG_API int zg_msm_g1_batch_sharded_async(zg_sbases_t sb, size_t n, const uint64_t *const *scalar_batches, size_t k /* >= 1 */,
                                  uint64_t *ticket);
ZG_API int zg_sharded_wait(zg_sbases_t sb, uint64_t ticket, uint64_t *out_xy, uint8_t *out_inf);
ZG_API int zg_g1_sbases_inflight(zg_sbases_t sb); /* calls the handle keeps in flight */

/* Sumcheck(F).Prover over a table sharded across the bound devices (S = the largest power of two <= devices and <= len):
 * LOW_PAIR tables by contiguous chunks, HIGH_HALF tables by residue class i mod S, so that every fold of the first
 * log2(len / S) rounds is local to a device; a round's sums are the modular sum of the per-device pairs (64 B per device, read
 * from each session's pinned mailbox); the last log2(S) rounds run on the S gathered residuals on device 0. Same messages as
 * the single-device session, bit for bit. */
typedef struct zg_ssc_s *zg_ssc_t;
ZG_API int zg_sumcheck_open_sharded(const uint64_t *evals, size_t len, int layout, zg_ssc_t *s);
ZG_API int zg_sumcheck_shards(zg_ssc_t s);
ZG_API size_t zg_sumcheck_len_sharded(zg_ssc_t s);
ZG_API int zg_sumcheck_round_sums_sharded(zg_ssc_t s, uint64_t g0[4], uint64_t g1[4]);
ZG_API int zg_sumcheck_bind_sharded(zg_ssc_t s, const uint64_t r[4]);
ZG_API int zg_sumcheck_final_sharded(zg_ssc_t s, uint64_t out[4]);
ZG_API int zg_sumcheck_close_sharded(zg_ssc_t s);

#ifdef __cplusplus
}
#endif
#endif /* ZOLT_GPU_H */
