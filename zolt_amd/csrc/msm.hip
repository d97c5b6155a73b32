// msm.hip — BN254 G1 multi-scalar multiplication on gfx950.
//
// Replaces the reference's single-threaded Pippenger (/root/reference/src/msm/mod.zig:375-438:
// c = 8, 32 windows, 255 Jacobian buckets on the stack, per-window fromMontgomery) with a
// pipeline shaped for 256 CUs and 288 GB of HBM:
//
//   upload (once per SRS)   bases -> table[l][i] = 2^(c*G*l) * P_i as 64-byte rows (l < L "precompute
//                           levels", stored in the accumulate kernel's lazy 29-bit-limb format); with L = W
//                           every window of a scalar lands in ONE shared bucket set and no window-combining
//                           doublings remain at MSM time.
//   msm_digits_lds          one fromMontgomery per scalar (the reference does 32), then W signed c-bit
//                           digits -> keys (bucket group g, |digit|-1); per-block histogram in LDS
//   msm_colscan / scan_a/b / scatter_lds   counting sort of the n*W (key, point-ref) pairs: per-block
//                           offsets, two-pass coalesced scan, scatter with LDS cursors (no global atomics)
//   msm_partition / fine_count / fine_offsets / fine_place   the same sort in two LDS-staged passes (coarse bins,
//                           then slices of a bin by fine key) for >= 8192 buckets: all HBM traffic coalesced
//   msm_accumulate_chunk    the sorted list is cut into equal chunks, one per thread, whatever the bucket
//                           sizes (skew-robust); complete XYZZ mixed adds on lazy limbs; one partial per
//                           bucket run
//   msm_bucket_combine / heavy_wave / heavy_a/b   per-bucket sums of the partials: GS lanes, a wave, or
//                           block trees depending on how many partials a bucket has
//   msm_bitsum              sum_k k*B_k = sum_b 2^b * T_b,  T_b = sum of the buckets whose index has
//                           bit b set: c plain tree sums (log depth) instead of the reference's
//                           serial running sum (a lone GPU lane needs ~10 us per point add)
//   msm_rowcol / bits2d     the same T_b for c >= 11 from row and column sums of the bucket matrix (every bucket
//                           added twice instead of (c-1)/2 times)
//   msm_final               b doublings of T_b in parallel quads, tree sum, window Horner if G > 1,
//                           one inversion (safegcd) -> affine
// Every kernel but the accumulation raises its wave priority: they are short dependent chains that would otherwise
// get a third of the issue slots beside another stream's accumulate waves.
//
// Bucket sums are order-independent group sums and the final affine coordinates are
// canonical field values, so the 64-byte result is bit-identical to the reference's
// MSM(F,G).compute whatever c, L and the scheduling are (tests/test_gpu_msm.py).
#include <string.h>

#include <mutex>
#include <vector>

#include "common.hip.h"
#include "g1.hip.h"
#include "g1_29.hip.h"
#include "g1_29x4.hip.h"

// Every MSM kernel except the bucket accumulation is short and mostly a chain of dependent operations; when it shares a SIMD with
// two accumulate waves of another stream the arbiter gives it a third of the issue slots and its latency triples, which
// stretches the per-stream chain digits -> sort -> accumulate -> reduce. Raised wave priority lets these kernels issue first;
// the accumulation fills the remaining slots (its total work is unchanged).
#ifndef ZG_TAIL_PRIO
#define ZG_TAIL_PRIO 3
#endif
#define ZG_HIPRIO() __builtin_amdgcn_s_setprio(ZG_TAIL_PRIO)

namespace zg {

static constexpr int MAX_GROUPS = 64;

struct MsmPlan {
    int c;         // window bits
    int W;         // windows = ceil(255 / c)
    int L;         // precompute levels stored in the table
    int G;         // bucket groups = ceil(W / L); window w -> group w % G, level w / G
    int S;         // accumulate threads (slices) per bucket
    uint32_t NB;   // buckets per group = 2^(c-1)
    uint32_t NK;   // total buckets = K * G * NB
    int K;         // MSMs sharing one launch set (scalar vectors over the same bases); bucket group = batch * G + w % G
    int PB;        // bit-sum partial blocks per (group, bit)
    int lb, hb;    // two-dimensional bucket reduction: low / high bits of a digit magnitude (0 = one-dimensional bit sums)
    uint32_t NT;   // chunk-scheduled accumulate: threads (0 = per-bucket scheduling)
    int GS;        // lanes per bucket in the combine pass
    int fb;        // two-pass sort: low key bits resolved by the second pass (0 = single-pass LDS / atomic sort)
    int rb;        // two-pass sort: bits of a table-row reference inside an intermediate entry (= 31 - fb)
    uint32_t NCB;  // two-pass sort: coarse bins = ceil(NK / 2^fb)
};

}  // namespace zg

struct zg_bases_s {
    size_t n = 0;
    int device = -1;  // the HIP device the table and the workspaces live on
    zg::MsmPlan plan;
    char *d_table = nullptr;     // L * n * 64 B
    uint8_t *d_inf = nullptr;    // n B or null
    uint64_t *d_scal = nullptr;  // staging for host scalars, n * 32 B
    // One MSM in flight needs one workspace ("lane"). Several lanes let independent MSMs issued on different
    // streams overlap: the latency-bound tail of one runs under the ALU-bound accumulation of the next.
    struct Lane {
        uint32_t *d_dig = nullptr, *d_sorted = nullptr, *d_hist = nullptr, *d_starts = nullptr;
        uint32_t *d_blockhist = nullptr;  // LDS sort path: nblk * NK per-block histograms / offsets (two-pass: nblk2 * NCB)
        uint32_t *d_tmp = nullptr;        // two-pass sort: entries partitioned by coarse bin, W * n
        uint32_t *d_cstarts = nullptr;    // two-pass sort: cstarts | totals | tstarts | istarts, NCB + 1 each
        uint32_t *d_fine = nullptr;       // two-pass sort: slicecnt[max items][2^fb] then fbase[NCB][2^fb]
        size_t fine_words = 0;            // ... its size (a point slice may sort under its own plan: slice_sort_plan)
        size_t blockhist_words = 0, tmp_words = 0, cstarts_words = 0;  // sizes of the two-pass buffers (0 = not there)
        char *d_partial = nullptr;        // NK * 144 B: bucket sums (lazy 29-bit-limb XYZZ records)
        char *d_slice_buckets = nullptr;  // point slices (msm_enqueue_sliced): the bucket sums of slices 1 .. S-1, built on first use
        size_t slice_buckets = 0;         // ... how many sets it holds
        uint32_t *d_slice_meta = nullptr; // ... and per slice: starts | nzrank | nzlist | MsmState (every slice is sorted before the first is accumulated)
        char *d_bits = nullptr;           // G * c * PB * 144 B: per-bit partial sums
        char *d_rg = nullptr;             // G * 128 B: per-group results
        uint32_t *d_nzrank = nullptr;     // NK + 1: non-empty buckets before k
        uint32_t *d_nzlist = nullptr;     // NK: the non-empty buckets, compacted
        uint32_t *d_scan_tmp = nullptr;   // 2*NK + 2*tiles: tile-local scans and tile totals
        char *d_part = nullptr;           // (NT + NK) * 144 B: per-(chunk, bucket-run) partial sums
        char *d_part2 = nullptr;          // heavy-bucket stage-A partials
        uint32_t *d_heavy = nullptr;      // 2*NK: heavy bucket list, then huge bucket list
        void *d_state = nullptr;          // MsmState
        hipEvent_t done = nullptr;        // recorded after the lane's last MSM; the next user waits on it
        hipStream_t last_st = nullptr;    // ... and the stream it was recorded on
        bool used = false;
#ifdef ZG_EXP_SKIP_SORT
        bool exp_sorted_once = false;
#endif
    };
    std::vector<Lane> lanes;
    size_t next_lane = 0;
    uint32_t nblk = 0;
    // zg_msm_g1_batch: k short scalar vectors over the same bases run as ONE launch set (k times the bucket groups);
    // its workspace is built on first use and kept while (k, n) fit
    Lane batch_lane;
    zg::MsmPlan batch_plan;
    size_t batch_n = 0;
    uint32_t batch_nblk = 0;
    // long vectors are not fused: they rotate over the caller's stream and three forked helper streams instead
    static constexpr int NAUX = 3;  // four streams with the caller's: fewer can land two of them on one of HIP's four hardware queues
    hipStream_t aux[NAUX] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[NAUX] = {nullptr, nullptr, nullptr};
    // Wide-window handles (2^15 buckets) also keep a narrow-window table of their first SIDE_TABLE_POINTS bases: MSMs over a
    // short prefix (HyperKZG.commit of a short polynomial on a long SRS, HyperKZG.open's last levels) then sort into 2^7
    // buckets instead of 2^15, and batches of them can be fused into one launch set. Same results, by construction.
    zg_bases_s *small = nullptr;
    uint64_t *d_out = nullptr;  // 16 x u64: result record + flag
    uint64_t *h_out = nullptr;  // pinned mirror
    uint64_t *d_slice_parts = nullptr;  // zg_msm_g1 (host scalars, sliced): one Jacobian partial per slice
    // rows of a batch on a handle without a table (msm_rows_shared_tail): the bucket sums of every row side by side, the reduction's
    // bit sums / rows / columns and group results for all of them, sized for rows_cap rows; rows_done: the last set that used them
    char *d_rows_buckets = nullptr, *d_rows_bits = nullptr, *d_rows_rg = nullptr;
    size_t rows_cap = 0;
    hipEvent_t rows_done = nullptr;
    std::mutex mu;
};

namespace zg {

// ------------------------------------------------------------------ kernels

// Table build: level 0 = the bases; level l = 2^(c*G) * level (l-1), as affine.
// Levels per inversion: the chain keeps its Jacobian point across levels; every level leaves (X, Y, Z, product of the group's earlier Zs) in a
// 144-byte record of `scratch` (n * min(levels - 1, PRE_GROUP) records, index j * n + i), one safegcd inversion per group of PRE_GROUP levels
// then yields every 1 / Z_j on the way back (Montgomery's trick: two products per level) and the rows are written as affine points —
// 7 products per level instead of an inversion (~80 products' worth), and Jacobian doublings (945 multiply-adds) instead of XYZZ ones
// (1269): 2^20 bases x 15 levels 20.2 -> 13 ms. Rows are lazy representatives (< 1.1p) of the same affine coordinates as before.
constexpr int PRE_GROUP = 7;
constexpr size_t PRE_CHUNK = (size_t)1 << 21;  // bases per launch of the table kernel (bounds its records)
// A launch carries up to TWO tables: job `side` (the handle's side table: 2^14 bases, 32 narrow levels — a 2.4 ms latency chain on 64
// workgroups) takes the first side.blocks workgroups, the main table's the rest. Workgroups are dispatched in index order, so the side
// table's chain starts first and runs UNDER the main table's thousands of workgroups: no second stream (a stream created for it would
// move the caller's later streams onto other hardware queues, an idle one may share the launch stream's queue and run behind it).
struct PreJob {
    const uint64_t *xy;
    const uint8_t *inf;
    size_t n;          // bases of the handle (row stride of its table)
    int levels, dbl_per_level;
    char *table, *scratch;
    size_t first, count;  // this launch: bases [first, first + count); the records of base i sit at local index i - first (stride count)
    unsigned blocks;      // workgroups of this job in the launch
};
__global__ void __launch_bounds__(256) msm_precompute_kernel(PreJob side, PreJob main_job) {
    const bool is_side = blockIdx.x < side.blocks;
    const PreJob &jb = is_side ? side : main_job;
    const uint64_t *xy = jb.xy;
    const uint8_t *inf = jb.inf;
    const size_t n = jb.n, first = jb.first, count = jb.count;
    const int levels = jb.levels, dbl_per_level = jb.dbl_per_level;
    char *table = jb.table, *scratch = jb.scratch;
    const size_t li = (size_t)(blockIdx.x - (is_side ? 0u : side.blocks)) * blockDim.x + threadIdx.x, i = first + li;
    if (li >= count) return;
    Affine p = affine_load(xy + 8 * i);
    Jac29 a;
    a.x = f29_from_fp(p.x);
    a.y = f29_from_fp(p.y);
    f29_store_packed(table + 64 * i, a.x);
    f29_store_packed(table + 64 * i + 32, a.y);
    if (inf && inf[i]) return;  // infinity bases are never referenced (digits are suppressed)
    F29 one29;
#pragma unroll
    for (int k = 0; k < 9; k++) one29.l[k] = Fp29::ONE[k];
    a.z = one29;
    for (int l0 = 1; l0 < levels; l0 += PRE_GROUP) {
        const int g = levels - l0 < PRE_GROUP ? levels - l0 : PRE_GROUP;
        F29 pref = one29;
        for (int j = 0; j < g; j++) {
            for (int k = 0; k < dbl_per_level; k++) a = jac29_dbl(a);
            XYZZ29 rec;
            rec.x = a.x; rec.y = a.y; rec.zz = a.z; rec.zzz = pref;
            xyzz29_store(scratch + 144 * ((size_t)j * count + li), rec);
            pref = f29_mul(pref, a.z);
        }
        F29 t = f29_from_fp(fe_inv_safegcd(f29_to_fp(pref)));  // 1 / (Z_0 ... Z_(g-1))
        for (int j = g - 1; j >= 0; j--) {
            const XYZZ29 rec = xyzz29_load(scratch + 144 * ((size_t)j * count + li));  // this thread's own record
            const F29 iz = f29_mul(t, rec.zzz);  // 1 / Z_j
            t = f29_mul(t, rec.zz);              // ... and Z_j leaves the running inverse
            const F29 iz2 = f29_sqr(iz);
            char *row = table + 64 * ((size_t)(l0 + j) * n + i);
            f29_store_packed(row, f29_mul(rec.x, iz2));                     // x = X / Z^2
            f29_store_packed(row + 32, f29_mul(rec.y, f29_mul(iz2, iz)));   // y = Y / Z^3
        }
    }
}

// The same table with an inversion per level and no scratch (rounds 1-4): kept for the case that the scratch does not fit, and as the
// A/B reference (ZG_MSM_PRECOMPUTE_V1=1)
__global__ void __launch_bounds__(256) msm_precompute_v1_kernel(const uint64_t *xy, const uint8_t *inf, size_t n, int levels,
                                                                int dbl_per_level, char *table) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine p = affine_load(xy + 8 * i);
    // rows are stored in the accumulate kernel's format: x, y as packed Montgomery-2^261 values (fp29.hip.h)
    F29 px = f29_from_fp(p.x), py = f29_from_fp(p.y);
    f29_store_packed(table + 64 * i, px);
    f29_store_packed(table + 64 * i + 32, py);
    if (inf && inf[i]) return;  // infinity bases are never referenced (digits are suppressed)
    F29 one29;
#pragma unroll
    for (int k = 0; k < 9; k++) one29.l[k] = Fp29::ONE[k];
    for (int l = 1; l < levels; l++) {
        XYZZ29 a;
        a.x = px; a.y = py; a.zz = one29; a.zzz = one29;
        for (int k = 0; k < dbl_per_level; k++) a = xyzz29_dbl(a);  // never the identity: the group has odd prime order
        // back to affine in the lazy domain: 1/Z = ZZ/ZZZ, x = X/Z^2, y = Y/ZZZ (one canonical-form inversion)
        F29 izzz = f29_from_fp(fe_inv_safegcd(f29_to_fp(a.zzz)));
        F29 iz = f29_mul(izzz, a.zz);
        px = f29_mul(a.x, f29_sqr(iz));
        py = f29_mul(a.y, izzz);
        char *row = table + 64 * ((size_t)l * n + i);
        f29_store_packed(row, px);
        f29_store_packed(row + 32, py);
    }
}

// Signed-digit decomposition. The reference extracts unsigned 8-bit windows of the canonical
// integer (getWindow, msm/mod.zig:441-471) and converts from Montgomery inside every call;
// here: one conversion, digits in [-2^(c-1), 2^(c-1)], so a group needs 2^(c-1) buckets.
template <int C>
__global__ void __launch_bounds__(256) msm_digits_kernel(const uint64_t *scalars, const uint8_t *inf, uint32_t n, uint32_t n_pts, int G,
                                                         uint32_t *dig, uint32_t *hist) {
    constexpr int W = (255 + C - 1) / C;
    constexpr uint32_t NB = 1u << (C - 1);
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Fr s = fr_from_mont29(fe_load<FrParams>(scalars + 4 * (size_t)i));
    uint32_t batch = i / n_pts, pt = i - batch * n_pts;  // scalar vector `batch`, base `pt` (n == n_pts: one MSM)
    bool skip = inf && inf[pt];  // msm/mod.zig:407: infinity bases contribute nothing
    uint32_t carry = 0;
#pragma unroll
    for (int w = 0; w < W; w++) {
        const int bit = w * C, limb = bit / 32, sh = bit % 32;
        uint32_t v = s.l[limb] >> sh;
        if (sh + C > 32 && limb + 1 < 8) v |= s.l[limb + 1] << (32 - sh);
        v = (v & ((1u << C) - 1u)) + carry;
        uint32_t neg = v > NB ? 1u : 0u;
        uint32_t d = neg ? (1u << C) - v : v;
        carry = neg;
        uint32_t e = 0xFFFFFFFFu;
        if (d != 0 && !skip) {
            uint32_t key = (batch * (uint32_t)G + (uint32_t)(w % G)) * NB + (d - 1);
            e = key | (neg << 31);
            atomicAdd(&hist[key], 1u);
        }
        dig[(size_t)w * n + i] = e;
    }
}

// exclusive scan of the bucket histogram (NK <= 2^21 entries), one block
// Exclusive scan of the bucket histogram in two coalesced passes (a one-block scan with strided slices was
// L2-latency-bound: 83 us for 32768 buckets). Pass A: each 1024-bucket tile is scanned in LDS (bucket sizes and
// non-empty flags) and its totals are published. Pass B: every tile adds the totals of the tiles before it and
// writes starts / nzrank / the compacted list of non-empty buckets.
__global__ void __launch_bounds__(1024) msm_scan_a_kernel(const uint32_t *__restrict__ hist, uint32_t NK, uint32_t *__restrict__ loc,
                                                          uint32_t *__restrict__ locz, uint32_t *__restrict__ tile_tot) {
    ZG_HIPRIO();
    __shared__ uint32_t sh[1024], shz[1024];
    uint32_t tid = threadIdx.x, k = blockIdx.x * 1024 + tid;
    uint32_t h = k < NK ? hist[k] : 0u, z = h ? 1u : 0u;
    sh[tid] = h;
    shz[tid] = z;
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {
        uint32_t v = tid >= o ? sh[tid - o] : 0, vz = tid >= o ? shz[tid - o] : 0;
        __syncthreads();
        sh[tid] += v;
        shz[tid] += vz;
        __syncthreads();
    }
    if (k < NK) {
        loc[k] = sh[tid] - h;    // exclusive within the tile
        locz[k] = shz[tid] - z;
    }
    if (tid == 1023) {
        tile_tot[2 * blockIdx.x] = sh[1023];
        tile_tot[2 * blockIdx.x + 1] = shz[1023];
    }
}

__global__ void __launch_bounds__(1024) msm_scan_b_kernel(const uint32_t *__restrict__ hist, uint32_t NK, const uint32_t *__restrict__ loc,
                                                          const uint32_t *__restrict__ locz, const uint32_t *__restrict__ tile_tot,
                                                          uint32_t *__restrict__ starts, uint32_t *__restrict__ nzrank,
                                                          uint32_t *__restrict__ nzlist, uint32_t *__restrict__ state, uint32_t nstate) {
    ZG_HIPRIO();
    __shared__ uint32_t pre[2];
    // the launch set's MsmState (heavy / huge bucket counters of the reduction and the arrival counters of the huge buckets) is
    // cleared here: one launch less than a memset
    if (state && blockIdx.x == 0)
        for (uint32_t w = threadIdx.x; w < nstate; w += 1024) state[w] = 0;
    uint32_t tid = threadIdx.x, k = blockIdx.x * 1024 + tid;
    if (tid < 64) {  // one wave sums the totals of the preceding tiles (at most 2048 tiles)
        uint32_t a = 0, az = 0;
        for (uint32_t t = tid; t < blockIdx.x; t += 64) {
            a += tile_tot[2 * t];
            az += tile_tot[2 * t + 1];
        }
        for (int d = 32; d > 0; d >>= 1) {
            a += __shfl_down(a, d, 64);
            az += __shfl_down(az, d, 64);
        }
        if (tid == 0) {
            pre[0] = a;
            pre[1] = az;
        }
    }
    __syncthreads();
    if (k < NK) {
        uint32_t s = pre[0] + loc[k], r = pre[1] + locz[k];
        starts[k] = s;
        nzrank[k] = r;  // number of non-empty buckets before k
        uint32_t h = hist[k];
        if (h) nzlist[r] = k;  // compacted list of the non-empty buckets
        if (k == NK - 1) {
            starts[NK] = s + h;
            nzrank[NK] = r + (h ? 1u : 0u);
        }
    }
}

// counting-sort scatter; order inside a bucket is irrelevant (group sums commute).
// fill[] is the histogram array re-zeroed by the caller.
__global__ void __launch_bounds__(256) msm_scatter_kernel(const uint32_t *dig, uint32_t n, uint32_t n_pts, int G, size_t table_n,
                                                          uint32_t off, const uint32_t *starts, uint32_t *fill, uint32_t *sorted) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    uint32_t w = blockIdx.y;
    if (i >= n) return;
    uint32_t e = dig[(size_t)w * n + i];
    if (e == 0xFFFFFFFFu) return;
    uint32_t key = e & 0x7FFFFFFFu;
    uint32_t pos = starts[key] + atomicAdd(&fill[key], 1u);
    uint32_t ref = (uint32_t)((size_t)(w / G) * table_n + off + i % n_pts);
    sorted[pos] = (e & 0x80000000u) | ref;
}

// Rows of a zero-padded matrix of scalar vectors by their LIVE lengths (HyperKZG.open's long levels: 2^19, 2^18, ... entries in rows of
// 2^19): the digit and sort kernels walk the compact index space [0, off[k]) — row j holds [off[j], off[j + 1]), its scalars sit at
// j * n_pts + (i - off[j]) — instead of k * n_pts scalars of which four in five are padding. k = 0: uniform rows of n_pts scalars.
struct RowOffs {
    uint32_t k;
    uint32_t off[33];
};
ZG_DEV void row_split(const RowOffs &r, uint32_t n_pts, uint32_t i, uint32_t &batch, uint32_t &pt) {
    if (r.k == 0) {
        batch = i / n_pts;
        pt = i - batch * n_pts;
        return;
    }
    uint32_t b = 0;
    while (b + 1 < r.k && i >= r.off[b + 1]) b++;
    batch = b;
    pt = i - r.off[b];
}

// ---- LDS-staged counting sort (used when the whole bucket histogram fits in LDS: NK*4 <= 128 KiB).
// One 1024-thread block per CU owns a contiguous slice of the scalars; its histogram and, later, its
// scatter cursors live in LDS (ds_add_rtn_u32), so the sort issues no global atomics at all. The global
// path above costs one device-scope atomic per digit, twice (histogram + scatter).
template <int C>
__global__ void __launch_bounds__(1024) msm_digits_lds_kernel(const uint64_t *scalars, const uint8_t *inf, uint32_t n, uint32_t n_pts,
                                                              int G, uint32_t per_block, uint32_t NK, int shift, uint32_t *dig,
                                                              uint32_t *blockhist, RowOffs rows) {
    ZG_HIPRIO();
    extern __shared__ uint32_t lds_hist[];
    constexpr int W = (255 + C - 1) / C;
    constexpr uint32_t NB = 1u << (C - 1);
    for (uint32_t k = threadIdx.x; k < NK; k += blockDim.x) lds_hist[k] = 0;
    __syncthreads();
    uint32_t i0 = blockIdx.x * per_block, i1 = i0 + per_block < n ? i0 + per_block : n;
    for (uint32_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        uint32_t batch, pt;
        row_split(rows, n_pts, i, batch, pt);
        Fr s = fr_from_mont29(fe_load<FrParams>(scalars + 4 * (rows.k ? (size_t)batch * n_pts + pt : (size_t)i)));
        bool skip = inf && inf[pt];
        uint32_t carry = 0;
#pragma unroll
        for (int w = 0; w < W; w++) {
            const int bit = w * C, limb = bit / 32, sh = bit % 32;
            uint32_t v = s.l[limb] >> sh;
            if (sh + C > 32 && limb + 1 < 8) v |= s.l[limb + 1] << (32 - sh);
            v = (v & ((1u << C) - 1u)) + carry;
            uint32_t neg = v > NB ? 1u : 0u;
            uint32_t d = neg ? (1u << C) - v : v;
            carry = neg;
            uint32_t e = 0xFFFFFFFFu;
            if (d != 0 && !skip) {
                uint32_t key = (batch * (uint32_t)G + (uint32_t)(w % G)) * NB + (d - 1);
                e = key | (neg << 31);
                atomicAdd(&lds_hist[key >> shift], 1u);  // shift > 0: coarse bins of the two-pass sort (NK = their number)
            }
            dig[(size_t)w * n + i] = e;
        }
    }
    __syncthreads();
    uint32_t *row = blockhist + (size_t)blockIdx.x * NK;
    for (uint32_t k = threadIdx.x; k < NK; k += blockDim.x) row[k] = lds_hist[k];
}

// per key: exclusive prefix over the blocks (in place) and the key's total
__global__ void __launch_bounds__(256) msm_colscan_kernel(uint32_t *blockhist, uint32_t nblk, uint32_t NK, uint32_t *total) {
    ZG_HIPRIO();
    uint32_t key = blockIdx.x * 256 + threadIdx.x;
    if (key >= NK) return;
    uint32_t run = 0;
    for (uint32_t b0 = 0; b0 < nblk; b0 += 16) {  // 16 independent loads in flight, then the in-place prefix
        uint32_t v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = (b0 + u < nblk) ? blockhist[(size_t)(b0 + u) * NK + key] : 0u;
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (b0 + u < nblk) blockhist[(size_t)(b0 + u) * NK + key] = run;
            run += v[u];
        }
    }
    total[key] = run;
}

__global__ void __launch_bounds__(1024) msm_scatter_lds_kernel(const uint32_t *dig, uint32_t n, uint32_t n_pts, int W, int G,
                                                               size_t table_n, uint32_t off, uint32_t per_block, uint32_t NK,
                                                               const uint32_t *starts, const uint32_t *blockhist, uint32_t *sorted, RowOffs rows) {
    ZG_HIPRIO();
    extern __shared__ uint32_t lds_cur[];
    const uint32_t *row = blockhist + (size_t)blockIdx.x * NK;
    for (uint32_t k = threadIdx.x; k < NK; k += blockDim.x) lds_cur[k] = starts[k] + row[k];
    __syncthreads();
    uint32_t i0 = blockIdx.x * per_block, i1 = i0 + per_block < n ? i0 + per_block : n;
    if (i1 - i0 < blockDim.x) {
        // fewer scalars than threads (a short MSM): the (window, scalar) pairs are spread over the threads — with the window as the outer
        // loop a 16-point MSM of 51 windows walked 51 dependent load -> LDS atomic -> store rounds with 16 lanes busy (17 us of a 130 us call)
        const uint32_t cnt = i1 - i0, total = cnt * (uint32_t)W;
        for (uint32_t t = threadIdx.x; t < total; t += blockDim.x) {
            const uint32_t w = t / cnt, i = i0 + (t - w * cnt);
            uint32_t e = dig[(size_t)w * n + i];
            if (e == 0xFFFFFFFFu) continue;
            uint32_t pos = atomicAdd(&lds_cur[e & 0x7FFFFFFFu], 1u);
            uint32_t batch, pt;
            row_split(rows, n_pts, i, batch, pt);
            sorted[pos] = (e & 0x80000000u) | (uint32_t)((size_t)(w / (uint32_t)G) * table_n + off + pt);
        }
        return;
    }
    // four windows per trip, their digit loads issued together: a trip used to be one dependent load -> LDS atomic -> store round per
    // window (~0.9 us each: 33 us of a 1024-point MSM's 37 windows)
    for (uint32_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        uint32_t batch, pt;
        row_split(rows, n_pts, i, batch, pt);
        for (int w0 = 0; w0 < W; w0 += 4) {
            uint32_t e[4];
#pragma unroll
            for (int u = 0; u < 4; u++) e[u] = w0 + u < W ? dig[(size_t)(w0 + u) * n + i] : 0xFFFFFFFFu;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (e[u] == 0xFFFFFFFFu) continue;
                uint32_t pos = atomicAdd(&lds_cur[e[u] & 0x7FFFFFFFu], 1u);
                sorted[pos] = (e[u] & 0x80000000u) | (uint32_t)((size_t)((uint32_t)(w0 + u) / (uint32_t)G) * table_n + off + pt);
            }
        }
    }
}

// ---- two-pass counting sort for many buckets (NK >= 8192). The single-pass scatter above writes runs of only
// n*W / (blocks * NK) entries per (block, bucket) — 2 entries = 8 bytes at 2^20 points — and rocprofv3 shows the price:
// WRITE_SIZE 514 MB for 64 MB of sorted references (every 64-byte line leaves L2 eight times, partially filled).
// Pass 1 (msm_scatter_lds_kernel with shift = fb) partitions by the HIGH key bits only: <= 8192 coarse bins, so a block
// writes runs of hundreds of bytes. Pass 2 (below) gives each coarse bin to one block: its entries are contiguous
// (~256 KiB, L2-resident), it counts the 2^fb fine keys in LDS, publishes the per-bucket histogram the accumulate
// scheduler needs, and places the entries — scattered 4-byte stores again, but confined to a region that is complete
// before it leaves L2.
// cstarts[b] = entries before coarse bin b (positions in the sorted list); tstarts[b] = the same with every bin rounded up to
// a multiple of 4 entries (positions in the intermediate buffer, so that pass 2 can use 16-byte loads); istarts[b] = work items
// before bin b, an item being a slice of at most FINE_SLICE entries of one bin (a skewed witness column can put half of all
// entries into one bin: it is then spread over many blocks instead of one).
static constexpr uint32_t FINE_SLICE = 32768;
__global__ void __launch_bounds__(1024) msm_coarse_base_kernel(const uint32_t *totals, uint32_t NCB, uint32_t *cstarts, uint32_t *tstarts,
                                                               uint32_t *istarts) {
    ZG_HIPRIO();
    __shared__ uint32_t sh[1024], sh4[1024], shi[1024];
    uint32_t tid = threadIdx.x, per = (NCB + 1023) / 1024, a = tid * per, b = a + per < NCB ? a + per : NCB;
    uint32_t s = 0, s4 = 0, si = 0;
    for (uint32_t k = a; k < b; k++) {
        s += totals[k];
        s4 += (totals[k] + 3u) & ~3u;
        si += (totals[k] + FINE_SLICE - 1) / FINE_SLICE;
    }
    sh[tid] = s;
    sh4[tid] = s4;
    shi[tid] = si;
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {
        uint32_t v = tid >= o ? sh[tid - o] : 0, v4 = tid >= o ? sh4[tid - o] : 0, vi = tid >= o ? shi[tid - o] : 0;
        __syncthreads();
        sh[tid] += v;
        sh4[tid] += v4;
        shi[tid] += vi;
        __syncthreads();
    }
    uint32_t run = sh[tid] - s, run4 = sh4[tid] - s4, runi = shi[tid] - si;  // exclusive
    for (uint32_t k = a; k < b; k++) {
        cstarts[k] = run;
        tstarts[k] = run4;
        istarts[k] = runi;
        run += totals[k];
        run4 += (totals[k] + 3u) & ~3u;
        runi += (totals[k] + FINE_SLICE - 1) / FINE_SLICE;
    }
    if (tid == 1023) {
        cstarts[NCB] = sh[1023];
        tstarts[NCB] = sh4[1023];
        istarts[NCB] = shi[1023];
    }
}

// pass 1: partition the digit entries by coarse bin (key >> fb). An intermediate entry keeps the fine key bits above the rb bits
// of the table-row reference: sign | fine << rb | ref. A block owns <= STAGE_ENTRIES entries (its scalars x W windows), sorts
// them by coarse bin INSIDE LDS (count, scan, place — LDS handles divergent addresses at full rate, the global store path
// handles about one divergent lane per two cycles per CU), and then copies every bin's run to its place in `tmp` with
// consecutive lanes on consecutive addresses: all HBM traffic of the pass is coalesced.
static constexpr uint32_t STAGE_ENTRIES = 32768;  // 128 KiB of LDS
// PLAIN: one bucket group and one scalar vector (the full-precompute single MSM) — level = window, point = scalar index; the
// general form divides by G and reduces modulo n_pts per entry (a select would evaluate both: 50 instructions per entry)
template <bool PLAIN>
__global__ void __launch_bounds__(1024) msm_partition_kernel(const uint32_t *dig, uint32_t n, uint32_t n_pts, int W, int G, size_t table_n,
                                                             uint32_t off, uint32_t per_block, uint32_t NCB, int fb, int rb,
                                                             const uint32_t *tstarts, const uint32_t *blockoff, uint32_t *tmp, int local_shift,
                                                             RowOffs rows) {
    ZG_HIPRIO();
    extern __shared__ uint32_t lds[];
    uint32_t *buf = lds, *cnt = lds + STAGE_ENTRIES, *lbase = cnt + NCB, *sums = lbase + NCB + 1;  // sums: 1024 scan partials
    const uint32_t tid = threadIdx.x, T = 1024, fmask = (1u << fb) - 1u;
    uint32_t i0 = blockIdx.x * per_block, cntl = i0 < n ? (n - i0 < per_block ? n - i0 : per_block) : 0;
    for (uint32_t k = tid; k < NCB; k += T) cnt[k] = 0;
    __syncthreads();
    // the block's entries stay in registers between the counting and the placing pass (<= 32 per thread). Register r holds
    // window w = r / JW, scalar (r % JW) * T + tid of the block, JW = ceil(cntl / T) in {1, 2} (two_pass_span keeps JW * W <= 32):
    // w and the row base are wave-uniform, so no per-entry division is needed to find them.
    const uint32_t jw2 = cntl > T ? 1u : 0u;
    uint32_t e[STAGE_ENTRIES / 1024];
#pragma unroll
    for (int r = 0; r < (int)(STAGE_ENTRIES / 1024); r++) {
        uint32_t w = (uint32_t)r >> jw2, j = ((uint32_t)r & jw2) * T + tid;
        e[r] = 0xFFFFFFFFu;
        if (w < (uint32_t)W && j < cntl) {
            uint32_t d = dig[(size_t)w * n + i0 + j];
            e[r] = d;  // key | sign, or 0xFFFFFFFF for a dropped digit
            if (d != 0xFFFFFFFFu) atomicAdd(&cnt[(d & 0x7FFFFFFFu) >> fb], 1u);
        }
    }
    __syncthreads();
    // exclusive scan of cnt[0..NCB) -> lbase (each thread owns a contiguous run of bins)
    {
        uint32_t per = (NCB + T - 1) / T, a = tid * per, b = a + per < NCB ? a + per : NCB, sacc = 0;
        for (uint32_t k = a; k < b && k < NCB; k++) sacc += cnt[k];
        sums[tid] = sacc;
        __syncthreads();
        for (uint32_t o = 1; o < T; o <<= 1) {
            uint32_t v = tid >= o ? sums[tid - o] : 0;
            __syncthreads();
            sums[tid] += v;
            __syncthreads();
        }
        uint32_t run = sums[tid] - sacc;
        for (uint32_t k = a; k < b && k < NCB; k++) {
            lbase[k] = run;
            run += cnt[k];
            cnt[k] = 0;  // becomes the placing cursor
        }
        if (tid == T - 1) lbase[NCB] = sums[T - 1];
    }
    __syncthreads();
    // destinations of this block's bins (at most 3 per thread: NCB <= 3000), loaded now so that their latency hides under the placing pass
    const uint32_t *row = blockoff + (size_t)blockIdx.x * NCB;
    uint32_t dst_pre[3];
#pragma unroll
    for (int u = 0; u < 3; u++) {
        uint32_t k = (uint32_t)u * T + tid;
        dst_pre[u] = k < NCB ? tstarts[k] + row[k] : 0u;
    }
    // a thread's entries belong to at most two scalars (rows 0 and 1 of every window): their point indices are found once
    // (several scalar vectors lie back to back over the same bases: index modulo n_pts)
    uint32_t pt_row0 = i0 + tid, pt_row1 = i0 + T + tid, row_unused;
    if (!PLAIN) {
        row_split(rows, n_pts, i0 + tid, row_unused, pt_row0);
        row_split(rows, n_pts, i0 + T + tid, row_unused, pt_row1);
    }
#pragma unroll
    for (int r = 0; r < (int)(STAGE_ENTRIES / 1024); r++) {
        if (e[r] != 0xFFFFFFFFu) {
            uint32_t w = (uint32_t)r >> jw2;
            uint32_t key = e[r] & 0x7FFFFFFFu, bin = key >> fb, lvl = (PLAIN || G == 1) ? w : w / (uint32_t)G;
            uint32_t pt = ((uint32_t)r & jw2) ? pt_row1 : pt_row0;
            // a point slice keeps slice-local references level << local_shift | point in the INTERMEDIATE entries (fewer bits than a table
            // row index: more fine bits, fewer coarse bins); msm_fine_place_kernel writes table rows into the final list
            const uint32_t ref = local_shift ? (lvl << local_shift) | pt : (uint32_t)((size_t)lvl * table_n + off + pt);
            uint32_t packed = (e[r] & 0x80000000u) | ((key & fmask) << rb) | ref;
            buf[lbase[bin] + atomicAdd(&cnt[bin], 1u)] = packed;
        }
    }
    __syncthreads();
    // every bin's destination goes to LDS first (the placing cursors are done with): read inside the copy loop, the two global
    // loads per bin were a dependent ~2 us each, i.e. most of this kernel's time (NCB / 16 iterations per wave)
#pragma unroll
    for (int u = 0; u < 3; u++) {
        uint32_t k = (uint32_t)u * T + tid;
        if (k < NCB) cnt[k] = dst_pre[u];
    }
    __syncthreads();
    // copy-out: a group of sg lanes (64, 32 or 16, by the average run length) walks one bin's run; wave v takes the bins
    // v * (64 / sg) + group, stepping by 16 waves' worth
    const uint32_t avg = STAGE_ENTRIES / NCB, sg = avg >= 48 ? 64u : (avg >= 24 ? 32u : 16u), per = 64u / sg;
    const uint32_t lane = tid & 63, sub = lane / sg, sl = lane % sg;
    for (uint32_t bin = (tid >> 6) * per + sub; bin < NCB; bin += (T >> 6) * per) {
        uint32_t a = lbase[bin], b = lbase[bin + 1];
        uint32_t dst = cnt[bin];
        for (uint32_t j = a + sl; j < b; j += sg) tmp[dst + (j - a)] = buf[j];
    }
}

// pass 2, three launches over the work items (slices of coarse bins; the grid is an upper bound, surplus blocks exit):
//   count : fine-key histogram of the slice (LDS) -> slicecnt[item][f]
//   offsets (one block per bin): per fine key, exclusive prefix over the bin's slices (in place) and the bucket's size ->
//           hist[key]; exclusive prefix over the fine keys -> fbase[bin][f]
//   place : cursor[f] = cstarts[bin] + fbase[bin][f] + slicecnt[item][f]; scattered 4-byte stores confined to the bin's region
// Entries are read with 16-byte loads (bins are 16-byte aligned in tmp and FINE_SLICE is a multiple of 4).
ZG_DEV bool fine_item(uint32_t item, const uint32_t *istarts, uint32_t NCB, uint32_t &bin, uint32_t &q0) {
    if (item >= istarts[NCB]) return false;
    uint32_t lo = 0, hi = NCB;  // istarts[lo] <= item < istarts[hi]
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (istarts[mid] <= item) lo = mid; else hi = mid;
    }
    // bins without entries have no items: skip back over equal istarts values is not needed (istarts[lo] <= item < istarts[lo+1])
    bin = lo;
    q0 = (item - istarts[lo]) * FINE_SLICE;
    return true;
}

__global__ void __launch_bounds__(1024) msm_fine_count_kernel(const uint32_t *tmp, const uint32_t *cstarts, const uint32_t *tstarts,
                                                              const uint32_t *istarts, uint32_t NCB, int fb, int rb, uint32_t *slicecnt) {
    ZG_HIPRIO();
    __shared__ uint32_t cnt[128];
    __shared__ uint32_t sh_bin, sh_q0, sh_ok;
    uint32_t tid = threadIdx.x, nf = 1u << fb, fmask = nf - 1u;
    if (tid == 0) {
        uint32_t bb = 0, qq = 0;
        sh_ok = fine_item(blockIdx.x, istarts, NCB, bb, qq) ? 1u : 0u;
        sh_bin = bb;
        sh_q0 = qq;
    }
    if (tid < nf) cnt[tid] = 0;
    __syncthreads();
    if (!sh_ok) return;
    uint32_t bin = sh_bin, q0 = sh_q0;
    uint32_t total = cstarts[bin + 1] - cstarts[bin];
    uint32_t count = total - q0 < FINE_SLICE ? total - q0 : FINE_SLICE;
    const uint4 *src = reinterpret_cast<const uint4 *>(tmp + tstarts[bin] + q0);
    const uint32_t T = blockDim.x, quads = (count + 3) / 4;
    for (uint32_t qb = 0; qb < quads; qb += 2 * T) {
        uint4 v[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            uint32_t q = qb + u * T + tid;
            v[u] = q < quads ? src[q] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            uint32_t q = qb + u * T + tid, e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (4 * q + j < count) atomicAdd(&cnt[(e[j] >> rb) & fmask], 1u);
        }
    }
    __syncthreads();
    if (tid < nf) slicecnt[(size_t)blockIdx.x * nf + tid] = cnt[tid];
}

// place: the slice (<= FINE_SLICE = STAGE_ENTRIES entries, held in registers) is sorted by fine key inside LDS, then every
// fine key's run is copied to  cstarts[bin] + fbase[bin][f] + (prefix over the earlier slices)  with coalesced stores.
__global__ void __launch_bounds__(1024) msm_fine_place_kernel(const uint32_t *tmp, const uint32_t *cstarts, const uint32_t *tstarts,
                                                              const uint32_t *istarts, uint32_t NCB, int fb, int rb, const uint32_t *slicecnt,
                                                              const uint32_t *fbase, uint32_t *sorted, int local_shift, uint32_t table_n,
                                                              uint32_t off) {
    ZG_HIPRIO();
    extern __shared__ uint32_t lds[];
    uint32_t *buf = lds, *cnt = lds + STAGE_ENTRIES, *lbase = cnt + 128;  // lbase: 129 entries
    __shared__ uint32_t sh_bin, sh_q0, sh_ok;
    uint32_t tid = threadIdx.x, nf = 1u << fb, fmask = nf - 1u, rmask = (1u << rb) - 1u;
    if (tid == 0) {
        uint32_t bb = 0, qq = 0;
        sh_ok = fine_item(blockIdx.x, istarts, NCB, bb, qq) ? 1u : 0u;
        sh_bin = bb;
        sh_q0 = qq;
    }
    if (tid < nf) cnt[tid] = 0;
    __syncthreads();
    if (!sh_ok) return;
    uint32_t bin = sh_bin, q0 = sh_q0;
    uint32_t total = cstarts[bin + 1] - cstarts[bin];
    uint32_t count = total - q0 < FINE_SLICE ? total - q0 : FINE_SLICE;
    const uint4 *src = reinterpret_cast<const uint4 *>(tmp + tstarts[bin] + q0);
    const uint32_t T = 1024, quads = (count + 3) / 4;
    // where every fine key's run goes: loaded now (latency under the counting pass), kept in LDS for the copy loop, which used
    // to wait for two dependent global loads per key
    uint32_t dst_pre = tid < nf ? cstarts[bin] + fbase[(size_t)bin * nf + tid] + slicecnt[(size_t)blockIdx.x * nf + tid] : 0u;
    uint4 v[FINE_SLICE / 4 / 1024];
#pragma unroll
    for (int u = 0; u < (int)(FINE_SLICE / 4 / 1024); u++) {
        uint32_t q = (uint32_t)u * T + tid;
        v[u] = q < quads ? src[q] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < (int)(FINE_SLICE / 4 / 1024); u++) {
        uint32_t q = (uint32_t)u * T + tid, e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (4 * q + j < count) atomicAdd(&cnt[(e[j] >> rb) & fmask], 1u);
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (uint32_t f = 0; f < nf; f++) {
            lbase[f] = run;
            run += cnt[f];
            cnt[f] = 0;
        }
        lbase[nf] = run;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < (int)(FINE_SLICE / 4 / 1024); u++) {
        uint32_t q = (uint32_t)u * T + tid, e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (4 * q + j < count) {
                uint32_t f = (e[j] >> rb) & fmask;
                uint32_t ref = e[j] & rmask;
                // a point slice sorted on slice-local references (level << shift | point): the final list holds table rows again,
                // so the accumulate kernel is the same for every launch (decoding there cost it 3 %)
                if (local_shift) ref = (ref >> local_shift) * table_n + off + (ref & ((1u << local_shift) - 1u));
                buf[lbase[f] + atomicAdd(&cnt[f], 1u)] = (e[j] & 0x80000000u) | ref;
            }
    }
    __syncthreads();
    if (tid < nf) cnt[tid] = dst_pre;  // the cursors are done with
    __syncthreads();
    for (uint32_t f = tid >> 6; f < nf; f += T >> 6) {
        uint32_t a = lbase[f], b = lbase[f + 1];
        uint32_t dst = cnt[f];
        for (uint32_t j = a + (tid & 63); j < b; j += 64) sorted[dst + (j - a)] = buf[j];
    }
}

__global__ void __launch_bounds__(128) msm_fine_offsets_kernel(const uint32_t *istarts, int fb, uint32_t NK, uint32_t *slicecnt, uint32_t *fbase,
                                                               uint32_t *hist) {
    ZG_HIPRIO();
    __shared__ uint32_t sh[128];
    uint32_t bin = blockIdx.x, f = threadIdx.x, nf = 1u << fb;
    uint32_t i0 = istarts[bin], i1 = istarts[bin + 1], run = 0;
    if (f < nf)
        for (uint32_t it = i0; it < i1; it++) {
            uint32_t v = slicecnt[(size_t)it * nf + f];
            slicecnt[(size_t)it * nf + f] = run;
            run += v;
        }
    sh[f] = f < nf ? run : 0;
    __syncthreads();
    for (uint32_t o = 1; o < 128; o <<= 1) {
        uint32_t v = f >= o ? sh[f - o] : 0;
        __syncthreads();
        sh[f] += v;
        __syncthreads();
    }
    if (f < nf) {
        fbase[(size_t)bin * nf + f] = sh[f] - run;
        uint32_t key = (bin << fb) | f;
        if (key < NK) hist[key] = run;
    }
}

// per coarse bin: exclusive prefix over the blocks' counts, in place, and the bin's total. The layout is chist[blk][bin]: a
// 1024-thread workgroup takes 16 CONSECUTIVE bins (16 lanes along the bin index: every access is a 64-byte row segment) and
// splits the blocks into 64 contiguous slices; the slices' sums meet in LDS. (One workgroup per bin walking a column with an
// 8 KiB stride cost 100 us at 2^22 points — 4 M uncoalesced 4-byte accesses; 64 bins per workgroup left only 32 workgroups.)
__global__ void __launch_bounds__(1024) msm_colscan_bins_kernel(uint32_t *chist, uint32_t nblk, uint32_t NCB, uint32_t *total) {
    ZG_HIPRIO();
    __shared__ uint32_t sh[64][16];
    const uint32_t lane = threadIdx.x & 15, slice = threadIdx.x >> 4, bin = blockIdx.x * 16 + lane;
    const uint32_t per = (nblk + 63) / 64, a = slice * per < nblk ? slice * per : nblk, b = a + per < nblk ? a + per : nblk;
    uint32_t s = 0;
    if (bin < NCB)
        for (uint32_t k = a; k < b; k++) s += chist[(size_t)k * NCB + bin];
    sh[slice][lane] = s;
    __syncthreads();
    uint32_t run = 0;
    for (uint32_t q = 0; q < slice; q++) run += sh[q][lane];
    if (bin < NCB) {
        for (uint32_t k = a; k < b; k++) {
            uint32_t v = chist[(size_t)k * NCB + bin];
            chist[(size_t)k * NCB + bin] = run;
            run += v;
        }
        if (slice == 63) total[bin] = run;
    }
}

ZG_DEV XYZZ xyzz_shfl_down(const XYZZ &v, int delta) {
    XYZZ r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        r.x.l[i] = __shfl_down(v.x.l[i], delta, 64);
        r.y.l[i] = __shfl_down(v.y.l[i], delta, 64);
        r.zz.l[i] = __shfl_down(v.zz.l[i], delta, 64);
        r.zzz.l[i] = __shfl_down(v.zzz.l[i], delta, 64);
    }
    return r;
}

// Bucket accumulation: the reference's inner loop buckets[idx] = buckets[idx].addAffine(base)
// (msm/mod.zig:406-418). S adjacent lanes share one bucket's list (S | 64); their partial sums
// are combined by a segmented shuffle tree and lane 0 of the segment stores the bucket.
__global__ void __launch_bounds__(256) msm_accumulate_kernel(const uint32_t *sorted, const uint32_t *starts, const char *table,
                                                             uint32_t NK, int S, char *buckets) {
    uint32_t t = blockIdx.x * 256 + threadIdx.x;
    uint32_t key = t / (uint32_t)S, s = t % (uint32_t)S;
    uint32_t a = 0, b = 0;
    if (key < NK) {
        uint32_t b0 = starts[key], len = starts[key + 1] - b0;
        a = b0 + (uint32_t)(((uint64_t)len * s) / (uint32_t)S);
        b = b0 + (uint32_t)(((uint64_t)len * (s + 1)) / (uint32_t)S);
    }
    XYZZ29 acc29;
    bool acc_inf = true;
    if (a < b) {
        uint32_t e = sorted[a];
        Affine cur = affine_load(table + 64 * (size_t)(e & 0x7FFFFFFFu));  // packed lazy-form row
        uint32_t cneg = e >> 31;
        for (uint32_t p = a; p < b; p++) {
            Affine nxt = cur;
            uint32_t nneg = 0;
            if (p + 1 < b) {  // prefetch the next row under the current add
                uint32_t e2 = sorted[p + 1];
                nxt = affine_load(table + 64 * (size_t)(e2 & 0x7FFFFFFFu));
                nneg = e2 >> 31;
            }
            F29 px = f29_unpack(cur.x.l), py = f29_unpack(cur.y.l);
            if (cneg) py = f29_neg2(py);
            xyzz29_madd(acc29, acc_inf, px, py);
            cur = nxt;
            cneg = nneg;
        }
    }
    XYZZ29 acc = acc_inf ? xyzz29_identity() : acc29;
    for (int d = 1; d < S; d <<= 1) {
        XYZZ29 o = xyzz29_shfl_down(acc, d);
        if ((s & (uint32_t)(2 * d - 1)) == 0) acc = xyzz29_add(acc, o);
    }
    if (key < NK && s == 0) xyzz29_store(buckets + 144 * (size_t)key, acc);
}

// ---------------------------------------------------------------------------------------------
// Skew-robust scheduling of the bucket accumulation. The kernel above gives every bucket S lanes, which
// is perfectly balanced for uniform scalars and pathological for real witness columns (a 0/1 column puts
// half of all points into ONE bucket). Here the sorted reference list is cut into NT equal chunks
// instead, one per thread, whatever the bucket sizes; a thread emits one partial sum per bucket run it
// crosses, at slot  chunk + (number of non-empty buckets before the run's bucket)  — unique and, for one
// bucket, contiguous. Buckets with few partials are finished by one thread each; "heavy" buckets
// (more than 8 partials) go through two block-level tree stages.
struct MsmState {
    uint32_t nheavy;  // buckets with more than 8*GS and at most HUGE_PARTIALS partials (queued by msm_bucket_combine)
    uint32_t nhuge;   // buckets with more than HUGE_PARTIALS partials (queued by msm_bucket_combine)
    uint32_t pad[2];
    // followed by one arrival counter per huge bucket (index = position in the huge list): see msm_heavy_kernel
};
static constexpr uint32_t HUGE_PARTIALS = 2048;
static constexpr uint32_t HEAVY_BLOCK_ITEMS = 2048;
// words of a launch set's state: the header and one arrival counter per possible huge bucket (each owns > 2048 partial slots)
static uint32_t state_words(uint32_t NT, uint32_t NK) { return 4 + (NT + NK) / HUGE_PARTIALS + 1; }

ZG_DEV uint32_t chunk_len(uint32_t total, uint32_t NT) {
    uint32_t C = (total + NT - 1) / NT;
    return C ? C : 1;
}

// QUAD = true: one chunk per quad of lanes, every mixed add by xyzz29_madd4 — for launch sets too short to fill the GPU, where
// the kernel is a latency chain of a few adds per chunk rather than a throughput problem (4 * NT lanes are launched).
#ifdef ZG_EXP_TABLE_MASK  // timing experiment only (tools/build_variant.sh): folds the gathers into a smaller table footprint — wrong sums
#define ZG_ROW_MASK (0x7FFFFFFFu & (uint32_t)(ZG_EXP_TABLE_MASK))
#else
#define ZG_ROW_MASK 0x7FFFFFFFu
#endif
#ifdef ZG_EXP_ACC_WAVES  // experiment: force the register budget of N waves per SIMD (tools/build_variant.sh)
#define ZG_ACC_ATTR __attribute__((amdgpu_waves_per_eu(ZG_EXP_ACC_WAVES, ZG_EXP_ACC_WAVES)))
#else
#define ZG_ACC_ATTR
#endif
template <bool QUAD>
__global__ void __launch_bounds__(256) ZG_ACC_ATTR msm_accumulate_chunk_kernel(const uint32_t *sorted, const uint32_t *starts, const uint32_t *nzrank,
                                                                   const uint32_t *nzlist, const char *table, uint32_t NK, uint32_t NT,
                                                                   char *part) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, q = 0;
    if (QUAD) {
        q = i & 3;
        i >>= 2;
    }
    uint32_t total = starts[NK];
    uint32_t C = chunk_len(total, NT);
    uint64_t a64 = (uint64_t)i * C;
    if (i >= NT || a64 >= total) return;
    uint32_t a = (uint32_t)a64, b = a + C < total ? a + C : total;
    // bucket of entry a: starts[k] <= a < starts[k+1]
    uint32_t lo = 0, hi = NK;  // invariant: starts[lo] <= a < starts[hi]
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (starts[mid] <= a) lo = mid; else hi = mid;
    }
    uint32_t r = nzrank[lo], kend = starts[lo + 1];  // r-th non-empty bucket; nzlist[r] == lo
    XYZZ29 acc;
    bool acc_inf = true;
    // Two loads feed an addition: the reference sorted[p] and the table row it names. Both run ahead of the arithmetic — the row
    // of entry p+1 is in flight during addition p (one row of 16 registers), and the reference of entry p+2 is loaded then too, so
    // the row gather of the next iteration never waits for its index (measured: accumulate 1.18 -> 1.13-1.14 ms at 2^20 points,
    // 784 -> 791-794 MSM/s; the sorted list is read at a stride of one chunk per lane, i.e. every lane pulls its own cache line).
    // TWO rows in flight (214 registers instead of 200) were measured slower: 759 MSM/s.
    uint32_t e = sorted[a];
    Affine cur = affine_load(table + 64 * (size_t)(e & ZG_ROW_MASK));  // packed lazy-form row
    uint32_t cneg = e >> 31;
    uint32_t e1 = a + 1 < b ? sorted[a + 1] : 0u;
    for (uint32_t p = a; p < b; p++) {
        Affine nxt = cur;
        uint32_t nneg = 0;
        if (p + 1 < b) {  // prefetch the next row under the current add
            uint32_t e2 = e1;
            if (p + 2 < b) e1 = sorted[p + 2];
            nxt = affine_load(table + 64 * (size_t)(e2 & ZG_ROW_MASK));
            nneg = e2 >> 31;
        }
        if (p == kend) {  // the run of bucket k ended inside this chunk: emit its partial, move on
            if (q == 0) xyzz29_store(part + 144 * (size_t)(i + r), acc_inf ? xyzz29_identity() : acc);
            acc_inf = true;
            r++;  // next non-empty bucket (p < total, so it exists); empty buckets are never walked
            kend = starts[nzlist[r] + 1];  // (loading the NEXT run's end one run early measured slower: 768 against 787 MSM/s)
        }
        F29 px = f29_unpack(cur.x.l), py = f29_unpack(cur.y.l);
        if (cneg) py = f29_neg2(py);
        if (QUAD) xyzz29_madd4(acc, acc_inf, px, py, q);
        else xyzz29_madd(acc, acc_inf, px, py);
        cur = nxt;
        cneg = nneg;
    }
    if (q == 0) xyzz29_store(part + 144 * (size_t)(i + r), acc_inf ? xyzz29_identity() : acc);
}

// GS adjacent QUADS of lanes per bucket (4*GS | 64, GS chosen from the expected partials per bucket) sum its partials: strided
// serial adds, then a segmented shuffle tree, every addition by a quad (g1_29x4.hip.h); buckets with more than 8*GS partials
// are queued as heavy
__global__ void __launch_bounds__(64) msm_bucket_combine_kernel(const char *part, const uint32_t *starts, const uint32_t *nzrank, uint32_t NK,
                                                                uint32_t NT, int GS, char *buckets, uint32_t *heavy_list, MsmState *st) {
    ZG_HIPRIO();
    uint32_t t = (blockIdx.x * 64 + threadIdx.x) >> 2, q = threadIdx.x & 3;
    uint32_t k = t / (uint32_t)GS, g = t % (uint32_t)GS;
    XYZZ29 acc = xyzz29_identity();
    bool light = false;
    if (k < NK) {
        uint32_t C = chunk_len(starts[NK], NT);
        uint32_t s0 = starts[k], s1 = starts[k + 1];
        if (s1 == s0) {
            light = true;  // empty bucket: identity
        } else {
            uint32_t q0 = s0 / C, q1 = (s1 - 1) / C, cnt = q1 - q0 + 1, base = q0 + nzrank[k];
            if (cnt > 8u * (uint32_t)GS) {  // heavy: a wave each; huge (a 0/1 witness column's bucket): block trees — msm_heavy_kernel
                if (g == 0 && q == 0) {
                    if (cnt > HUGE_PARTIALS) heavy_list[NK + atomicAdd(&st->nhuge, 1u)] = k;
                    else heavy_list[atomicAdd(&st->nheavy, 1u)] = k;
                }
            } else {
                light = true;
                if (g < cnt) acc = xyzz29_load(part + 144 * (size_t)(base + g));  // the first partial is taken as it is
                // the next partial is loaded under the current addition (the chain used to pay a memory latency per partial)
                uint32_t j = g + (uint32_t)GS;
                XYZZ29 nxt = j < cnt ? xyzz29_load(part + 144 * (size_t)(base + j)) : xyzz29_identity();
                while (j < cnt) {
                    XYZZ29 cur = nxt;
                    j += (uint32_t)GS;
                    if (j < cnt) nxt = xyzz29_load(part + 144 * (size_t)(base + j));
                    acc = xyzz29_add4(acc, cur, q);
                }
            }
        }
    }
    for (int d = 1; d < GS; d <<= 1) {
        XYZZ29 o = xyzz29_shfl_down(acc, 4 * d);  // the same lane of the quad d quads further
        if ((g & (uint32_t)(2 * d - 1)) == 0) acc = xyzz29_add4(acc, o, q);
    }
    if (light && g == 0 && q == 0) xyzz29_store(buckets + 144 * (size_t)k, acc);
}

// block-wide sum of lazy XYZZ points through LDS (256 threads x 144 B); result returned to every thread. After the first level
// (128 additions, one lane each) every addition of a level is done by a quad of lanes (g1_29x4.hip.h): the chain is one
// single-lane addition plus seven quad additions instead of eight single-lane ones.
__device__ __forceinline__ XYZZ29 block_sum_xyzz29(const XYZZ29 &acc, uint4 *sh) {
    uint32_t tid = threadIdx.x;
    xyzz29_store(&sh[tid * 9], acc);
    __syncthreads();
    if (tid < 128) {
        XYZZ29 x = xyzz29_load(&sh[tid * 9]), y = xyzz29_load(&sh[(tid + 128) * 9]);
        xyzz29_store(&sh[tid * 9], xyzz29_add(x, y));
    }
    __syncthreads();
    uint32_t g = tid >> 2, q = tid & 3;
    for (uint32_t o = 64; o > 0; o >>= 1) {
        if (g < o) {  // slot g is read and written by quad g only, slot g + o read by quad g only: no hazard inside a level
            XYZZ29 x = xyzz29_load(&sh[g * 9]), y = xyzz29_load(&sh[(g + o) * 9]);
            XYZZ29 r = xyzz29_add4(x, y, q);
            if (q == 0) xyzz29_store(&sh[g * 9], r);
        }
        __syncthreads();
    }
    return xyzz29_load(&sh[0]);
}

// The same sum over ONE wave (64 partial sums, 9 KiB of LDS): one single-lane addition, then five levels of quad additions. For launches
// with many rows (several bucket sets: a batch, a handle without a table) — see msm_rowcol_wave_kernel.
__device__ __forceinline__ XYZZ29 wave_sum_xyzz29(const XYZZ29 &acc, uint4 *sh) {
    const uint32_t tid = threadIdx.x;  // 64 threads
    xyzz29_store(&sh[tid * 9], acc);
    __syncthreads();
    if (tid < 32) {
        XYZZ29 x = xyzz29_load(&sh[tid * 9]), y = xyzz29_load(&sh[(tid + 32) * 9]);
        xyzz29_store(&sh[tid * 9], xyzz29_add(x, y));
    }
    __syncthreads();
    const uint32_t g = tid >> 2, q = tid & 3;
    for (uint32_t o = 16; o > 0; o >>= 1) {
        if (g < o) {
            XYZZ29 x = xyzz29_load(&sh[g * 9]), y = xyzz29_load(&sh[(g + o) * 9]);
            XYZZ29 r = xyzz29_add4(x, y, q);
            if (q == 0) xyzz29_store(&sh[g * 9], r);
        }
        __syncthreads();
    }
    return xyzz29_load(&sh[0]);
}

// Heavy buckets in ONE launch (uniform scalars have none: the kernel then returns at once — it used to be three launches of ~5 us
// each). Part 1, one wave per heavy bucket (more than 8*GS, at most HUGE_PARTIALS partials; typical source: the top window of a
// c-bit decomposition covers only a few bits, so its digits pile into a small set of buckets): strided serial sums (<= 32 per lane)
// and a shuffle tree. Part 2, huge buckets (a 0/1 column: one bucket holds everything): block b owns the partial slots
// [2048 b, 2048 (b+1)) and tree-sums every run of a huge bucket inside into part2[b + nzrank[k]] (stage A); the block that
// arrives LAST at the bucket's counter (device-scope acq_rel, as the sumcheck rounds end) sums the bucket's stage-A partials
// (stage B). The counters are cleared by msm_scan_b_kernel.
__global__ void __launch_bounds__(256) msm_heavy_kernel(const char *part, const uint32_t *starts, const uint32_t *nzrank, const uint32_t *nzlist,
                                                        uint32_t NK, uint32_t NT, const uint32_t *heavy_list, MsmState *st, char *part2,
                                                        char *buckets) {
    ZG_HIPRIO();
    __shared__ uint4 sh[256 * 9];
    __shared__ uint32_t sh_last;
    const uint32_t nheavy = st->nheavy, nhuge = st->nhuge;
    if (nheavy == 0 && nhuge == 0) return;
    const uint32_t C = chunk_len(starts[NK], NT), tid = threadIdx.x;
    {
        const uint32_t lane = tid & 63;
        for (uint32_t h = blockIdx.x * 4 + (tid >> 6); h < nheavy; h += gridDim.x * 4) {
            uint32_t k = heavy_list[h];
            uint32_t s0 = starts[k], s1 = starts[k + 1];
            uint32_t q0 = s0 / C, q1 = (s1 - 1) / C, cnt = q1 - q0 + 1, base = q0 + nzrank[k];
            XYZZ29 acc = xyzz29_identity();
            for (uint32_t j = lane; j < cnt; j += 64) acc = xyzz29_add(acc, xyzz29_load(part + 144 * (size_t)(base + j)));
            for (int d = 1; d < 64; d <<= 1) {
                XYZZ29 o = xyzz29_shfl_down(acc, d);
                if ((lane & (uint32_t)(2 * d - 1)) == 0) acc = xyzz29_add(acc, o);
            }
            if (lane == 0) xyzz29_store(buckets + 144 * (size_t)k, acc);
        }
    }
    if (nhuge == 0) return;
    const uint32_t *huge_list = heavy_list + NK;
    uint32_t *arrive = reinterpret_cast<uint32_t *>(st) + 4;
    uint32_t NZ = nzrank[NK];  // non-empty buckets; the r-th one, k = nzlist[r], owns slots [starts[k]/C + r, ...)
    uint32_t lo_slot = blockIdx.x * HEAVY_BLOCK_ITEMS, hi_slot = lo_slot + HEAVY_BLOCK_ITEMS;
    if (lo_slot >= NT + NK) return;
    uint32_t lo = 0, hi = NZ;  // last r whose first slot is <= lo_slot (first slots increase strictly with r)
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (starts[nzlist[mid]] / C + mid <= lo_slot) lo = mid; else hi = mid;
    }
    for (uint32_t r = lo; r < NZ; r++) {
        uint32_t k = nzlist[r];
        uint32_t s0 = starts[k], s1 = starts[k + 1];
        uint32_t q0 = s0 / C, q1 = (s1 - 1) / C, cnt = q1 - q0 + 1, base = q0 + r;
        if (base >= hi_slot) break;
        if (cnt <= HUGE_PARTIALS || base + cnt <= lo_slot) continue;
        uint32_t r0 = base > lo_slot ? base : lo_slot, r1 = base + cnt < hi_slot ? base + cnt : hi_slot;
        XYZZ29 acc = xyzz29_identity();
        for (uint32_t j = r0 + tid; j < r1; j += 256) acc = xyzz29_add(acc, xyzz29_load(part + 144 * (size_t)j));
        XYZZ29 res = block_sum_xyzz29(acc, sh);
        const uint32_t b0 = base / HEAVY_BLOCK_ITEMS, b1 = (base + cnt - 1) / HEAVY_BLOCK_ITEMS;
        if (tid == 0) {
            xyzz29_store(part2 + 144 * (size_t)(blockIdx.x + r), res);  // plain store: the arrival's release publishes it
            uint32_t h = 0;
            while (h < nhuge && huge_list[h] != k) h++;
            uint32_t arrived = __hip_atomic_fetch_add(&arrive[h], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the acquire's invalidate has completed before the barrier releases the other waves' loads
            sh_last = arrived == b1 - b0 ? 1u : 0u;
        }
        __syncthreads();
        if (sh_last) {  // stage B: this block saw every other block of the bucket arrive
            XYZZ29 a2 = xyzz29_identity();
            for (uint32_t bb = b0 + tid; bb <= b1; bb += 256) a2 = xyzz29_add(a2, xyzz29_load(part2 + 144 * (size_t)(bb + r)));
            __syncthreads();  // sh is reused
            XYZZ29 tot = block_sum_xyzz29(a2, sh);
            if (tid == 0) xyzz29_store(buckets + 144 * (size_t)k, tot);
        }
        __syncthreads();
    }
}

// Bucket reduction, step 1. The reference computes sum_k k*B_k with a serial running sum
// (msm/mod.zig:423-432). Here  sum_k k*B_k = sum_b 2^b * T_b  with  T_b = sum_{k: bit b of k set} B_k;
// block (x, b, g) tree-sums a slice of T_b of group g. Bucket index idx holds digit magnitude k = idx+1.
__global__ void __launch_bounds__(256) msm_bitsum_kernel(const char *buckets, uint32_t NB, int c, char *out) {
    ZG_HIPRIO();
    __shared__ uint4 sh[256 * 9];
    uint32_t b = blockIdx.y, g = blockIdx.z, PB = gridDim.x;
    uint32_t total = (b == (uint32_t)(c - 1)) ? 1u : NB / 2;
    XYZZ29 acc = xyzz29_identity();
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < total; j += PB * 256) {
        uint32_t k = (b == (uint32_t)(c - 1)) ? NB : ((((j >> b) << 1) | 1u) << b) | (j & ((1u << b) - 1u));
        acc = xyzz29_add(acc, xyzz29_load(buckets + 144 * ((size_t)g * NB + (k - 1))));
    }
    XYZZ29 r = block_sum_xyzz29(acc, sh);
    if (threadIdx.x == 0) xyzz29_store(out + 144 * (((size_t)g * c + b) * PB + blockIdx.x), r);
}

// Bucket reduction, step 1, two-dimensional form for wide windows (c >= 11). The bit sums above touch every bucket (c-1)/2 times
// on average: 2^(c-2) * (c-1) additions per group — for c = 16 a quarter of a million, 16 M VALU wave-instructions, a fixed cost
// that does not shrink with the number of points. Write a digit magnitude k < NB = 2^(c-1) as k = h * 2^lb + l: then
//     sum_k k * B_k = 2^lb * sum_h h * R_h + sum_l l * C_l,    R_h = sum_l B_(h,l),  C_l = sum_h B_(h,l),
// i.e. every bucket is added twice (one row sum, one column sum: 2^c additions), and the weighted sums run over 2^hb rows and
// 2^lb columns only. msm_rowcol_kernel: block x < 2^hb sums row x, block 2^hb + l sums column l (<= 256 elements each, one
// block tree). msm_bits2d_kernel then forms the same T_b the one-dimensional kernel produces — for b < lb from the columns whose
// index has bit b, for lb <= b < c-1 from the rows whose index has bit b - lb, T_(c-1) = B_NB — so msm_final_kernel is unchanged.
__global__ void __launch_bounds__(256) msm_rowcol_kernel(const char *buckets, uint32_t NB, int lb, int hb, char *rc) {
    ZG_HIPRIO();
    __shared__ uint4 sh[256 * 9];
    const uint32_t x = blockIdx.x, g = blockIdx.y, nrow = 1u << hb, ncol = 1u << lb, t = threadIdx.x;
    XYZZ29 acc = xyzz29_identity();
    // (rows / columns of more than 256 elements — windows of 18 bits and more — take several elements per thread)
    const uint32_t len = x < nrow ? ncol : nrow;
    for (uint32_t e = t; e < len; e += 256) {
        const uint32_t k = x < nrow ? ((x << lb) | e) : ((e << lb) | (x - nrow));  // digit magnitude of the element; 0 = none
        if (k != 0) {
            XYZZ29 v = xyzz29_load(buckets + 144 * ((size_t)g * NB + (k - 1)));
            acc = e < 256 ? v : xyzz29_add(acc, v);
        }
    }
    XYZZ29 r = block_sum_xyzz29(acc, sh);
    if (t == 0) xyzz29_store(rc + 144 * ((size_t)g * (nrow + ncol) + x), r);
}
// The same sums by ONE wave per row / column: a lane adds up to four of the 256 elements itself, then the wave's tree. A 256-thread
// workgroup per row keeps three quarters of its lanes idle from the second tree level on and needs 36 KiB of LDS: with several bucket
// sets in one launch (K rows of a batch, the G window groups of a handle without a table: 2560 ... 8704 workgroups) the launch ran in
// waves of workgroups — 230 us for the five long levels of HyperKZG.open, 240 us for ONE table-less MSM. One wave per row is a few
// additions longer alone (57 -> ~70 us) and several times shorter there, so it is taken from three bucket sets on.
__global__ void __launch_bounds__(64) msm_rowcol_wave_kernel(const char *buckets, uint32_t NB, int lb, int hb, char *rc) {
    ZG_HIPRIO();
    __shared__ uint4 sh[64 * 9];
    const uint32_t x = blockIdx.x, g = blockIdx.y, nrow = 1u << hb, ncol = 1u << lb, t = threadIdx.x;
    XYZZ29 acc = xyzz29_identity();
    const uint32_t len = x < nrow ? ncol : nrow;
    bool first = true;
    for (uint32_t e = t; e < len; e += 64) {
        const uint32_t k = x < nrow ? ((x << lb) | e) : ((e << lb) | (x - nrow));  // digit magnitude of the element; 0 = none
        if (k != 0) {
            XYZZ29 v = xyzz29_load(buckets + 144 * ((size_t)g * NB + (k - 1)));
            acc = first ? v : xyzz29_add(acc, v);
            first = false;
        }
    }
    XYZZ29 r = wave_sum_xyzz29(acc, sh);
    if (t == 0) xyzz29_store(rc + 144 * ((size_t)g * (nrow + ncol) + x), r);
}
static int env_int(const char *name, int dflt);
static void launch_rowcol(hipStream_t st, const char *buckets, uint32_t NB, int lb, int hb, int sets, char *rc) {
    const int wave_from = env_int("ZG_MSM_ROWCOL_WAVE_FROM", 3);  // bucket sets from which one wave sums a row (0: never)
    if (wave_from > 0 && sets >= wave_from)
        hipLaunchKernelGGL(msm_rowcol_wave_kernel, dim3((1u << hb) + (1u << lb), sets), dim3(64), 0, st, buckets, NB, lb, hb, rc);
    else
        hipLaunchKernelGGL(msm_rowcol_kernel, dim3((1u << hb) + (1u << lb), sets), dim3(256), 0, st, buckets, NB, lb, hb, rc);
}

__global__ void __launch_bounds__(256) msm_bits2d_kernel(const char *buckets, const char *rc, uint32_t NB, int c, int lb, int hb, char *out) {
    ZG_HIPRIO();
    __shared__ uint4 sh[256 * 9];
    const uint32_t b = blockIdx.x, g = blockIdx.y, nrow = 1u << hb, ncol = 1u << lb, t = threadIdx.x;
    const char *rcg = rc + 144 * (size_t)g * (nrow + ncol);
    XYZZ29 acc = xyzz29_identity();
    if (b == (uint32_t)(c - 1)) {
        if (t == 0) acc = xyzz29_load(buckets + 144 * ((size_t)g * NB + (NB - 1)));
    } else if (b < (uint32_t)lb) {
        for (uint32_t e = t; e < ncol; e += 256)
            if ((e >> b) & 1u) acc = xyzz29_add(acc, xyzz29_load(rcg + 144 * (size_t)(nrow + e)));
    } else {
        for (uint32_t e = t; e < nrow; e += 256)
            if ((e >> (b - (uint32_t)lb)) & 1u) acc = xyzz29_add(acc, xyzz29_load(rcg + 144 * (size_t)e));
    }
    XYZZ29 r = block_sum_xyzz29(acc, sh);
    if (t == 0) xyzz29_store(out + 144 * ((size_t)g * c + b), r);
}

ZG_DEV void write_result(const XYZZ &acc, int mode, uint64_t *out_rec, uint8_t *out_inf) {
    Affine r;
    bool inf = xyzz_to_affine(acc, r);
    if (mode == 0) {  // affine xy[8] + inf flag (AffinePoint, msm/mod.zig:15-30)
        affine_store(out_rec, r);
        *out_inf = inf ? 1 : 0;
    } else {  // ParallelMSM's threadWorker record fromAffine(result) = (x,y,1) / (1,1,0) (msm/mod.zig:663-664)
        Fp one = Fp::one();
        if (inf) {
            fe_store(out_rec, one); fe_store(out_rec + 4, one); fe_store(out_rec + 8, Fp::zero());
        } else {
            fe_store(out_rec, r.x); fe_store(out_rec + 4, r.y); fe_store(out_rec + 8, one);
        }
    }
}

// mode 2: a per-GPU partial that is only ever fed to the combine step. Any Jacobian representative of the shard's
// sum works there, so the inversion (the longest single piece of a small MSM's tail) is skipped:
// (X*ZZ, Y*ZZZ, ZZ) has the same affine image as the XYZZ point; identity = (1,1,0).
ZG_DEV void write_partial_unnormalised(const XYZZ &acc, uint64_t *out_rec) {
    Fp X, Y, Z;
    xyzz_to_jacobian(acc, X, Y, Z);
    fe_store(out_rec, X); fe_store(out_rec + 4, Y); fe_store(out_rec + 8, Z);
}

// Bucket reduction, step 2 (block g, 512 threads = 128 quads): the c x PB partial sums of group g go to LDS; a tree over the
// partials of each bit, then quad b doubles T_b b times, then a tree over the bits  ->  R_g = sum_k k*B_k of group g.
// Every point operation is done by a quad of lanes (g1_29x4.hip.h). With a single group (full precompute) thread 0 goes
// straight on to toAffine (msm/mod.zig:178-189).
__global__ void __launch_bounds__(512) msm_final_kernel(const char *bits, int c, int PB, int G, char *rg, int mode, uint64_t *out_rec,
                                                        uint8_t *out_inf, uint32_t rec_stride, uint32_t inf_stride) {
    ZG_HIPRIO();
    __shared__ uint4 pts[256 * 9];  // slot b * S + j, S = PB rounded up to a power of two: 256 / S bit rows (c <= 16 for S = 16, c <= 32 below)
    uint32_t tid = threadIdx.x, g = blockIdx.x, quad = tid >> 2, q = tid & 3;
    uint32_t S = 1;
    while (S < (uint32_t)PB) S <<= 1;
    if (tid < 256) {
        uint32_t b = tid / S, j = tid % S;
        XYZZ29 v = xyzz29_identity();
        if (b < (uint32_t)c && j < (uint32_t)PB) v = xyzz29_load(bits + 144 * (((size_t)g * c + b) * PB + j));
        xyzz29_store(&pts[tid * 9], v);
    }
    __syncthreads();
    for (uint32_t d = 1; d < S; d <<= 1) {   // per bit: partial j += partial j + d, for j a multiple of 2d
        uint32_t per_bit = S / (2 * d);      // additions per bit at this level; (256 / S) * per_bit <= 128 quads
        if (quad < (256 / S) * per_bit) {
            uint32_t b = quad / per_bit, j = (quad % per_bit) * 2 * d;
            XYZZ29 x = xyzz29_load(&pts[(b * S + j) * 9]), y = xyzz29_load(&pts[(b * S + j + d) * 9]);
            XYZZ29 r = xyzz29_add4(x, y, q);
            if (q == 0) xyzz29_store(&pts[(b * S + j) * 9], r);
        }
        __syncthreads();
    }
    if (quad < (uint32_t)c) {  // 2^b * T_b
        XYZZ29 v = xyzz29_load(&pts[(quad * S) * 9]);
#if !(defined(ZG_EXP_SKIP) && (ZG_EXP_SKIP & 1))
        for (uint32_t i = 0; i < quad; i++) v = xyzz29_dbl4(v, q);
#endif
        if (q == 0) xyzz29_store(&pts[(quad * S) * 9], v);
    }
    __syncthreads();
    uint32_t top = 1;  // bit rows c .. top-1 hold the identity (top <= 256 / S)
    while (top < (uint32_t)c) top <<= 1;
    for (uint32_t o = top >> 1; o > 0; o >>= 1) {
        if (quad < o) {
            XYZZ29 x = xyzz29_load(&pts[(quad * S) * 9]), y = xyzz29_load(&pts[((quad + o) * S) * 9]);
            XYZZ29 r = xyzz29_add4(x, y, q);
            if (q == 0) xyzz29_store(&pts[(quad * S) * 9], r);
        }
        __syncthreads();
    }
    if (tid != 0) return;
#if defined(ZG_EXP_SKIP) && (ZG_EXP_SKIP & 4)
    if (G == 1) { xyzz29_store(rg, xyzz29_load(&pts[0])); return; }
#endif
    XYZZ r = xyzz29_to_std_val(xyzz29_load(&pts[0]));  // canonical Montgomery-2^256 from here on
    if (G == 1) {  // block g is scalar vector g of a batched launch (g == 0 for a single MSM)
        out_rec += (size_t)rec_stride * g;
        out_inf += (size_t)inf_stride * g;
        if (mode == 2) write_partial_unnormalised(r, out_rec);
        else write_result(r, mode, out_rec, out_inf);
    } else {
        xyzz_store(rg + 128 * (size_t)g, r);
    }
}

// window combine for G > 1 (msm/mod.zig:393-398,434): Horner from the top group with c doublings per step
__global__ void msm_groups_kernel(const char *rg, int G, int c, int mode, uint64_t *out_rec, uint8_t *out_inf, uint32_t rec_stride,
                                  uint32_t inf_stride) {
    ZG_HIPRIO();
    rg += 128 * (size_t)G * blockIdx.x;  // one single-thread block per scalar vector
    out_rec += (size_t)rec_stride * blockIdx.x;
    out_inf += (size_t)inf_stride * blockIdx.x;
    // one quad of lanes per scalar vector (g1_29x4.hip.h): the (G-1)*c doublings are a serial chain
    uint32_t q = threadIdx.x & 3;
    XYZZ29 acc29 = xyzz29_from_std_val(xyzz_load(rg + 128 * (size_t)(G - 1)));
    for (int g = G - 2; g >= 0; g--) {
        for (int k = 0; k < c; k++) acc29 = xyzz29_dbl4(acc29, q);
        acc29 = xyzz29_add4(acc29, xyzz29_from_std_val(xyzz_load(rg + 128 * (size_t)g)), q);
    }
    if (threadIdx.x != 0) return;
    XYZZ acc = xyzz29_to_std_val(acc29);
    if (mode == 2) write_partial_unnormalised(acc, out_rec);
    else write_result(acc, mode, out_rec, out_inf);
}

__global__ void msm_identity_kernel(int mode, uint64_t *out_rec, uint8_t *out_inf) {
    if (mode == 0) {
        for (int i = 0; i < 8; i++) out_rec[i] = 0;
        *out_inf = 1;
    } else {
        Fp one = Fp::one();
        fe_store(out_rec, one); fe_store(out_rec + 4, one); fe_store(out_rec + 8, Fp::zero());
    }
}

// ParallelMSM combine (msm/mod.zig:647-652): serial add of k Jacobian partials + toAffine. Block j combines scalar vector j of
// a sharded batch: rank i's record for it sits at partials + i * rank_stride + 12 * j (u64 units); its result record goes to
// out_xy + j * rec_stride / out_inf + j * inf_stride.
__global__ void __launch_bounds__(64) msm_combine_kernel(const uint64_t *partials, uint32_t k, uint32_t rank_stride, uint64_t *out_xy,
                                                         uint8_t *out_inf, uint32_t rec_stride, uint32_t inf_stride, int mode = 0) {
    // one wave = 16 quads of lanes: quad i folds partials i, i+16, ... (Jacobian -> lazy XYZZ), then a shuffle tree over the
    // quads, every addition by a quad (g1_29x4.hip.h); the group sum does not depend on the association order, and the
    // affine result is canonical
    partials += 12 * (size_t)blockIdx.x;
    out_xy += (size_t)rec_stride * blockIdx.x;
    out_inf += (size_t)inf_stride * blockIdx.x;
    uint32_t lane = threadIdx.x, quad = lane >> 2, q = lane & 3;
    XYZZ29 acc = xyzz29_identity();
    for (uint32_t i = quad; i < k; i += 16) {
        const uint64_t *rec = partials + (size_t)rank_stride * i;
        Fp X = fe_load<FpParams>(rec), Y = fe_load<FpParams>(rec + 4), Z = fe_load<FpParams>(rec + 8);
        if (!Z.is_zero()) {
            XYZZ29 p;
            F29 z = f29_from_fp(Z);
            p.x = f29_from_fp(X); p.y = f29_from_fp(Y);
            p.zz = f29_sqr(z);
            p.zzz = f29_mul(p.zz, z);
            acc = xyzz29_add4(acc, p, q);
        }
    }
    for (int d = 1; d < 16; d <<= 1) {
        XYZZ29 o = xyzz29_shfl_down(acc, 4 * d);
        if ((quad & (uint32_t)(2 * d - 1)) == 0 && (uint32_t)d < k) acc = xyzz29_add4(acc, o, q);
    }
    if (lane != 0) return;
    XYZZ tot = xyzz29_to_std_val(acc);
    if (mode == 2) write_partial_unnormalised(tot, out_xy);  // the record modes of msm_final_kernel (a sliced MSM ends here instead)
    else write_result(tot, mode, out_xy, out_inf);
}

// MSM.scalarMul(base, scalar).toAffine() (msm/mod.zig:503-540), one pair per thread
__global__ void __launch_bounds__(256) g1_scalar_mul_kernel(const uint64_t *xy, const uint8_t *inf, const uint64_t *scalars, size_t n,
                                                            uint64_t *out_xy, uint8_t *out_inf) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine p = affine_load(xy + 8 * i);
    Fr s = fe_from_mont(fe_load<FrParams>(scalars + 4 * i));
    XYZZ acc = XYZZ::identity();
    if (!(inf && inf[i])) {
        for (int limb = 7; limb >= 0; limb--) {
            uint32_t wv = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) wv = (k == limb) ? s.l[k] : wv;
            for (int bit = 31; bit >= 0; bit--) {
                acc = xyzz_dbl(acc);
                if ((wv >> bit) & 1u) acc = xyzz_madd(acc, p);
            }
        }
    }
    Affine r;
    bool isinf = xyzz_to_affine(acc, r);
    affine_store(out_xy + 8 * i, r);
    out_inf[i] = isinf ? 1 : 0;
}

// ---- fixed-base batch scalar multiplication: HyperKZG.setup's loop powers[i] = scalarMul(g1, tau^i).toAffine()
// (src/poly/commitment/mod.zig:194-199; generateMockSRS, srs.zig:326-355) — every product has the SAME base. The reference's
// scalarMul is double-and-add (254 doublings + ~127 additions per output, msm/mod.zig:503-540); with one shared table
// T[w][d-1] = d * 2^(8w) * G (32 windows x 255 multiples, 510 KiB, L2-resident) an output is at most 32 mixed additions
// and one toAffine, no doubling at all.
// Window width by batch size: 8 bits (32 windows, 510 KiB of table, 8160 rows to build) up to 2^15 outputs, 10 bits (26 windows, 1.7 MB)
// up to 2^18, 11 bits (24 windows, 3.1 MB — still inside one XCD's L2) beyond: an output then costs 24 additions instead of 32.
// HyperKZG.setup of 2^20 powers, key without a table: 5.25 ms at 8 bits, 4.87 / 4.69 / 4.74 / 4.83 at 10 / 11 / 12 / 13.
// ZG_FB_WINDOW_BITS overrides.
struct FbPlan {
    int c, W;
    uint32_t rows;  // 2^c - 1 multiples per window
};
static constexpr int FB_W_MAX = 64;  // c >= 4

// step 1 (one block, quad w of lanes per window): B_w = 2^(8w) * G by 8w doublings, every doubling by a quad (g1_29x4.hip.h) —
// the only serial chain of the build (248 doublings for the top window)
__global__ void __launch_bounds__(4 * FB_W_MAX) fb_window_bases_kernel(const uint64_t *base_xy, int FB_C, char *bw /* W * 144 */) {
    uint32_t w = threadIdx.x >> 2, q = threadIdx.x & 3;
    Affine g = affine_load(base_xy);
    F29 one29;
#pragma unroll
    for (int k = 0; k < 9; k++) one29.l[k] = Fp29::ONE[k];
    XYZZ29 step;
    step.x = f29_from_fp(g.x); step.y = f29_from_fp(g.y); step.zz = one29; step.zzz = one29;
    for (uint32_t k = 0; k < (uint32_t)FB_C * w; k++) step = xyzz29_dbl4(step, q);  // odd prime order: never the identity
    if (q == 0) xyzz29_store(bw + 144 * (size_t)w, step);
}

// step 2 (one thread per table row): d * B_w by double-and-add over the 8 bits of d, then to affine — the packed 64-byte row
// format of the accumulate kernel (x, y as lazy Montgomery-2^261 values)
__global__ void __launch_bounds__(256) fb_table_rows_kernel(const char *bw, uint32_t n_rows, int FB_C, uint32_t FB_ROWS, char *table) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_rows) return;
    uint32_t w = i / FB_ROWS, d = i % FB_ROWS + 1;
    XYZZ29 b = xyzz29_load(bw + 144 * (size_t)w), a = xyzz29_identity();
    for (int bit = FB_C - 1; bit >= 0; bit--) {
        a = xyzz29_dbl(a);
        if ((d >> bit) & 1u) a = xyzz29_add(a, b);
    }
    F29 izzz = f29_from_fp(fe_inv_safegcd(f29_to_fp(a.zzz)));
    F29 iz = f29_mul(izzz, a.zz);  // 1/Z = ZZ / ZZZ
    f29_store_packed(table + 64 * (size_t)i, f29_mul(a.x, f29_sqr(iz)));
    f29_store_packed(table + 64 * (size_t)i + 32, f29_mul(a.y, izzz));
}

__global__ void __launch_bounds__(256) fb_mul_kernel(const char *table, const uint64_t *scalars, size_t n, int FB_C, int FB_W, uint32_t FB_ROWS,
                                                     uint64_t *out_xy, uint8_t *out_inf) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr s = fr_from_mont29(fe_load<FrParams>(scalars + 4 * i));  // canonical integer, 8 x 32-bit words
    XYZZ29 acc;
    bool acc_inf = true;
    const uint32_t mask = (1u << FB_C) - 1u;
#pragma unroll 1
    for (int w = 0; w < FB_W; w++) {
        const uint32_t d = s.l[0] & mask;  // the next FB_C bits; the scalar is shifted down behind it (static register indices)
#pragma unroll
        for (int k = 0; k < 8; k++) s.l[k] = (s.l[k] >> FB_C) | (k < 7 ? s.l[k + 1] << (32 - FB_C) : 0u);
        if (d == 0) continue;
        Affine row = affine_load(table + 64 * ((size_t)w * FB_ROWS + (d - 1)));
        xyzz29_madd(acc, acc_inf, f29_unpack(row.x.l), f29_unpack(row.y.l));
    }
    Affine r;
    bool isinf = acc_inf;
    if (!isinf) isinf = xyzz_to_affine(xyzz29_to_std_val(acc), r);
    if (isinf) {
        r.x = Fp::zero();
        r.y = Fp::zero();
    }
    affine_store(out_xy + 8 * i, r);
    out_inf[i] = isinf ? 1 : 0;
}

// tau^i for i < n as canonical Montgomery elements, for HyperKZG.setup on the device (zg_hyperkzg_setup): three 256-entry tables of
// powers — tau^a, (tau^256)^b, (tau^65536)^c, one thread each walks its chain of 255 products — then every i = a + 256 b + 65536 c is two
// products. Exact products: the bytes are those of the reference's running product tau_power = tau_power * tau
// (src/poly/commitment/mod.zig:190-199).
struct TauArg { uint32_t l[8]; };  // a field element as a kernel argument
__global__ void __launch_bounds__(64) tau_tables_kernel(TauArg tau, uint64_t *tabs /* 4 x 256 x 4 */) {
    if (threadIdx.x >= 4) return;
    Fr t;
#pragma unroll
    for (int i = 0; i < 8; i++) t.l[i] = tau.l[i];
    for (uint32_t k = 0; k < threadIdx.x; k++)  // t = tau^(256^table)
        for (int j = 0; j < 8; j++) t = fr_mul29v(t, t);
    Fr acc = Fr::one();
    uint64_t *out = tabs + (size_t)threadIdx.x * 256 * 4;
    for (int j = 0; j < 256; j++) {
        fe_store(out + 4 * j, acc);
        acc = fr_mul29v(acc, t);
    }
}
__global__ void __launch_bounds__(256) tau_powers_kernel(const uint64_t *tabs, size_t n, uint64_t *out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Fr v = fe_load<FrParams>(tabs + 4 * (i & 255));
    if (i >> 8) v = fr_mul29v(v, fe_load<FrParams>(tabs + 4 * (256 + ((i >> 8) & 255))));
    if (i >> 16) v = fr_mul29v(v, fe_load<FrParams>(tabs + 4 * (512 + ((i >> 16) & 255))));
    if (i >> 24) v = fr_mul29v(v, fe_load<FrParams>(tabs + 4 * (768 + (i >> 24))));  // the reference's largest key is 2^24 + 256 powers (src/host/mod.zig:384-387)
    fe_store(out + 4 * i, v);
}

// AffinePoint.add (msm/mod.zig:74-103) and, through add(p, p), AffinePoint.double (:118-138): the lambda formulas on canonical
// Montgomery values, one inversion per pair (safegcd, the value of the reference's Fermat inverse)
__global__ void __launch_bounds__(256) g1_affine_add_kernel(const uint64_t *a_xy, const uint8_t *a_inf, const uint64_t *b_xy,
                                                            const uint8_t *b_inf, size_t n, uint64_t *out_xy, uint8_t *out_inf) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine a = affine_load(a_xy + 8 * i), b = affine_load(b_xy + 8 * i), r;
    bool ai = a_inf && a_inf[i], bi = b_inf && b_inf[i], ri = false;
    if (ai) {  // :75-76
        r = b;
        ri = bi;
    } else if (bi) {
        r = a;
    } else {
        Fp num, den;
        bool dbl = false;
        if (a.x.eq(b.x)) {  // :79-88
            if (a.y.eq(fe_neg(b.y))) ri = true;
            else if (a.y.eq(b.y)) dbl = true;
        }
        if (!ri) {
            if (dbl) {  // :118-138: lambda = 3 x^2 / 2 y; y == 0 -> identity
                Fp xx = fe_sqr(a.x);
                num = fe_add(fe_add(xx, xx), xx);
                den = fe_add(a.y, a.y);
                if (a.y.is_zero()) ri = true;
            } else {  // :90-93: lambda = (y2 - y1) / (x2 - x1)
                num = fe_sub(b.y, a.y);
                den = fe_sub(b.x, a.x);
            }
        }
        if (!ri && den.is_zero()) ri = true;  // dx.inverse() orelse return identity() (:93,:129)
        if (!ri) {
            Fp lam = fe_mul(num, fe_inv_safegcd(den));
            Fp x2 = dbl ? a.x : b.x;
            r.x = fe_sub(fe_sub(fe_sqr(lam), a.x), x2);
            r.y = fe_sub(fe_mul(lam, fe_sub(a.x, r.x)), a.y);
        }
    }
    if (ri) {  // AffinePoint.identity(): x = y = 0, infinity = true (:24-30)
        r.x = Fp::zero();
        r.y = Fp::zero();
    }
    affine_store(out_xy + 8 * i, r);
    if (out_inf) out_inf[i] = ri ? 1 : 0;
}

// AffinePoint.isOnCurve (msm/mod.zig:106-115)
__global__ void __launch_bounds__(256) g1_on_curve_kernel(const uint64_t *xy, const uint8_t *inf, size_t n, uint8_t *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (inf && inf[i]) {
        out[i] = 1;
        return;
    }
    Affine p = affine_load(xy + 8 * i);
    Fp three = fe_add(fe_dbl(Fp::one()), Fp::one());
    Fp rhs = fe_add(fe_mul(fe_sqr(p.x), p.x), three);
    out[i] = fe_sqr(p.y).eq(rhs) ? 1 : 0;
}

// ------------------------------------------------------------------ host side

static int ilog2(uint32_t v) {
    int r = 0;
    while ((1u << (r + 1)) <= v) r++;
    return r;
}

static int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

// scalars per block of the single-pass LDS sort: with few buckets the per-block histogram is cheap, and a short input spread over
// more blocks is less of a dependent load -> LDS atomic -> store chain per thread (1024 points: scatter 30 -> 10 us)
static uint32_t sort_span(uint32_t NK) {
    int v = env_int("ZG_MSM_SORT_SPAN", 0);
    if (v > 0) return (uint32_t)v;
    return NK <= 4096 ? 256u : 2048u;
}

// chunks (threads, or quads of lanes) of the chunk-scheduled accumulate for a launch set of `digits` entries
// alone: no other MSM of the handle is in flight — nothing needs the spare registers, the kernel takes every slot (2^20 points:
// 1.26 -> 1.18 ms)
static uint32_t chunk_threads(uint64_t digits, bool alone = false) {
    static const uint64_t per_chunk = [] {
        int v = env_int("ZG_MSM_CHUNK_ENTRIES", 16);  // sorted entries per chunk a launch aims for (each chunk also emits >= 1 partial)
        return (uint64_t)(v < 1 ? 1 : v);
    }();
    uint64_t want = digits / per_chunk;
    uint32_t nt = 1024;
    while (nt < want && nt < 131072u) nt <<= 1;
    // full size = 2 waves per SIMD on 256 CUs (512 workgroups). With other MSMs in flight a launch takes 7/8 of it (448 workgroups: a
    // quarter of the CUs hold one workgroup instead of two): the spare registers let another stream's latency-bound kernels (bit sums,
    // final) run under this kernel, and the NEXT accumulation's first workgroups start at once on the half-filled CUs, so that the
    // equal-length chunks of consecutive launches stop draining and refilling the chip in step. Round 2 measured 15/16 against the
    // full grid (+4-6 % MSM/s); round 4 swept the count (tools/exp/archive/run_nt_sweep.sh, profiles/r4h_accumulate_slots_sweep.txt, three
    // streams at 2^20): 512 / 496 / 480 / 464 / 448 / 440 / 432 / 416 / 384 workgroups = 793 / 790 / 800 / 790 / 816 / 810 / 800 /
    // 809 / 792 MSM/s — 448 held +2 % over 480 in three separate runs.
    static const uint32_t inflight = [] {
        int v = env_int("ZG_MSM_INFLIGHT_CHUNKS", 114688);
        return (uint32_t)(v < 1024 ? 1024 : (v > 131072 ? 131072 : v));
    }();
    return nt == 131072u && !alone ? inflight : nt;
}

static int make_plan(size_t n, const zg_msm_config *cfg, MsmPlan &p, size_t batch = 1) {
    p.fb = 0;  // sort mode is decided afterwards (plan_two_pass)
    p.rb = 0;
    p.NCB = 0;
    int c = cfg ? cfg->window_bits : 0;
    if (c == 0) c = env_int("ZG_MSM_WINDOW_BITS", 0);
    int L = cfg ? cfg->precompute_levels : 0;
    if (L == 0) L = env_int("ZG_MSM_PRECOMPUTE", 0);
    // a handle that will serve only a few MSMs (MSM.compute on a temporary slice) skips the table: its build costs about as
    // much as twenty MSMs save (2^20 points: 41 ms against 2 ms per MSM)
    if (L == 0 && cfg && cfg->expected_uses > 0 && cfg->expected_uses < 16) L = 1;
    if (c == 0 && L == 1) {
        // table-less plan (one bucket set per window): the windows cost buckets, not table rows, so the choice differs from the table
        // plan's. Measured (tools/exp/archive/run_noprecomp_sweep.sh, profiles/r4_noprecomp_sweep.txt): at 2^20 points c = 15 runs 559 MSM/s
        // pipelined / 3.0 ms alone, c = 13 553 / 3.3, and the table plan's c = 16 347 / 4.4 (2^19 buckets overflow the LDS sort:
        // digits 0.03 -> 0.64 ms, sort 0.26 -> 0.93); at 2^16 points c = 13 is 0.55 / 1.42 ms against 0.90 / 1.63 for c = 16.
        // What remains alone is the window combine: (W - 1) c = 240 dependent doublings (msm_groups_kernel, 0.86 ms) that a table
        // would have removed and nothing else can.
        c = n >= ((size_t)1 << 19) ? 15 : (n >= 8192 ? 13 : (n >= 2048 ? 8 : (n >= 64 ? 7 : 5)));
    }
    if (c == 0) {
        // measured on MI355X (tools/bench_window.py): window sizes whose last window covers only a couple of the 254
        // scalar bits (c = 9, 12, 14) waste a window and pile its digits into a handful of buckets; 16 wins from
        // 2^15 points up (fewest windows; the rest of the pipeline is latency), 8 / 7 below.
        c = n >= 32768 ? 16 : (n >= 8192 ? 10 : (n >= 2048 ? 8 : (n >= 64 ? 7 : 5)));  // 2^13 points: 0.40 ms with c = 8, 0.32 with 10
        // 17 bits = 15 windows instead of 16 (6 % fewer bucket additions) for twice the buckets: pays from about 2^20 points,
        // as long as the 15 n table rows leave the two-pass sort at least 5 fine key bits beside the 26-bit reference of an
        // intermediate entry (n <= 4.4 M: 2^22 points run 177 instead of 172 MSM/s, accumulate 5.98 -> 5.61 ms)
        if (batch == 1 && n >= (size_t)env_int("ZG_MSM_C17_MIN", 900000) && (uint64_t)n * 15 <= (1u << 26)) c = 17;
        // 18 / 19 bits exist (window_bits, ZG_MSM_WINDOW_BITS) and are NOT chosen: 19 bits = 14 windows take 8 % off the accumulate kernel
        // (1.18 -> 1.09 ms at 2^20) and put more than that back into the per-bucket work of 2^18 buckets (sort 0.13 -> 0.22 ms, combine +
        // row / column sums 0.27 -> 0.49 ms): 781 -> 735 MSM/s pipelined, 1.65 -> 1.95 ms alone (round 4, tools/exp/archive/run_c19.sh)
    }
    if (c < 2 || c > 19) {
        set_error("msm: window_bits must be in [2,19]");
        return ZG_ERR_INVALID;
    }
    p.c = c;
    p.W = (255 + c - 1) / c;
    if (L == 0) L = p.W;  // 288 GB of HBM: full precompute is 64*W bytes per base
    if (L < 1) L = 1;
    if (L > p.W) L = p.W;
    p.G = (p.W + L - 1) / L;
    p.L = (p.W + p.G - 1) / p.G;
    if (p.G > MAX_GROUPS) {
        set_error("msm: too many bucket groups for this window size");
        return ZG_ERR_INVALID;
    }
    p.NB = 1u << (c - 1);
    p.K = (int)batch;
    if ((uint64_t)p.NB * p.G * batch > (1u << 21)) {
        set_error("msm: too many buckets");
        return ZG_ERR_INVALID;
    }
    p.NK = p.NB * (uint32_t)p.G * (uint32_t)batch;
    // slices per bucket: aim for ~2^18-2^19 accumulate threads
    int S = 1;
    while ((uint64_t)p.NK * S * 2 <= (1u << 18) && S < 64) S *= 2;
    p.S = env_int("ZG_MSM_SLICES", S);
    if (p.S < 1 || p.S > 64 || (p.S & (p.S - 1))) {
        set_error("msm: slices per bucket must be a power of two <= 64");
        return ZG_ERR_INVALID;
    }
    // chunk-scheduled accumulate: enough threads to fill 2 waves per SIMD on 256 CUs, fewer for small inputs
    p.NT = 0;
    if (env_int("ZG_MSM_CHUNK_SCHED", 1)) p.NT = (uint32_t)env_int("ZG_MSM_CHUNK_THREADS", (int)chunk_threads((uint64_t)n * batch * p.W, true));  // the most a launch uses
    // combine lanes per bucket: a bucket expects about NT/NK + 1 partials; about 4 per quad (every tree level costs the whole
    // wave one more addition; ZG_MSM_COMBINE_PER_QUAD = 8 halves the quads, measured equal)
    p.GS = 1;
    const uint64_t nt_usual = p.NT ? (getenv("ZG_MSM_CHUNK_THREADS") ? p.NT : chunk_threads((uint64_t)n * batch * p.W)) : 0;  // with other MSMs in flight
    while (p.GS < 16 && (uint64_t)p.GS * (uint64_t)env_int("ZG_MSM_COMBINE_PER_QUAD", 4) < nt_usual / p.NK + 1) p.GS <<= 1;  // GS quads of lanes per bucket: 4 * GS <= 64
    // bit-sum partial blocks: ~4 buckets per thread, at most 16 (the final kernel reduces 16 lanes per bit)
    int pb = (int)(p.NB / 2 / (256 * 4));
    p.PB = pb < 1 ? 1 : (pb > 16 ? 16 : pb);
    if (c > 16 && p.PB > 8) p.PB = 8;  // msm_final_kernel holds 256 partial sums: 17 bit rows need a stride of at most 8
    // wide windows: row / column sums first (msm_rowcol_kernel); rows and columns of at most 256 buckets
    p.lb = p.hb = 0;
    if (c >= 11 && env_int("ZG_MSM_REDUCE_2D", 1)) {
        p.lb = c / 2;  // c - 1 = lb + hb, lb >= hb
        p.hb = c - 1 - p.lb;
    }
    return ZG_OK;
}

// Decide whether a launch set of n_total scalars under plan p sorts in two passes (see msm_finesort_kernel): worth it when
// the per-(block, bucket) runs of the single-pass scatter are a few bytes, i.e. many buckets. table_rows = L * (bases in the
// handle) bounds a row reference, which shares a 32-bit intermediate entry with the sign and the fine key bits.
static uint32_t two_pass_span(int W) {
    // a partition block stages per_block * W entries in LDS and keeps them in 32 registers per thread as ceil(per_block / 1024)
    // rows per window (msm_partition_kernel): <= 2048 scalars for W <= 16 windows, <= 1024 up to 32 windows
    uint32_t cap = W <= 16 ? 2048u : 1024u;
    uint32_t v = (uint32_t)env_int("ZG_MSM_TWO_PASS_SPAN", 2048);
    v = v < 256 ? 256 : v;
    return v > cap ? cap : v;
}
static void plan_two_pass(MsmPlan &p, size_t table_rows, size_t n_total) {
    p.fb = 0;
    p.rb = 0;
    p.NCB = 0;
    if (!env_int("ZG_MSM_TWO_PASS_SORT", 1) || p.NK < 8192 || p.W > 32 || (uint64_t)n_total * p.W < (1u << 17)) return;  // W: see two_pass_span
    int need = 1;
    while (((size_t)1 << need) < table_rows) need++;
    int fb = 31 - need, fb_max = env_int("ZG_MSM_FINE_BITS", 7);
    if (fb_max > 7) fb_max = 7;
    if (fb > fb_max) fb = fb_max;
    // fewer than 7 fine bits mean >= 512 coarse bins; down to 5 bits (2^22 points, 1024 bins) the two passes still beat the
    // single-pass sort there (0.67 vs 1.3 ms alone, +2-3 % pipelined); below that the single pass is used
    if (fb < env_int("ZG_MSM_FINE_BITS_MIN", 5)) return;
    uint32_t ncb = (p.NK + (1u << fb) - 1) >> fb;
    if (ncb > 3000) return;  // pass 1 keeps 2 * NCB counters + 1024 scan partials next to 128 KiB of staged entries in 156 KiB of LDS
    p.fb = fb;
    p.rb = 31 - fb;
    p.NCB = ncb;
}

// ---- point slices (see msm_enqueue_lane): how a launch set of n_pts points under plan p is cut, and the sort plan of one slice
static size_t table_span_points(int L) {
    const size_t span_mb = (size_t)env_int("ZG_MSM_TABLE_SPAN_MB", 1024);  // 0 = never slice
    if (!span_mb) return 0;
    const size_t pts = (span_mb << 20) / (64 * (size_t)L), least = (size_t)env_int("ZG_MSM_TABLE_SPAN_MIN_POINTS", 65536);  // tests lower it
    return pts < least ? least : pts;
}
static constexpr size_t DEV_SLICES_MAX = 128;  // 2^27 bases (the most a handle takes) / 2^20
static void slice_counts(const MsmPlan &p, size_t n_pts, size_t &S, size_t &per) {
    S = 1;
    per = n_pts;
    if (p.K != 1) return;
    const size_t sp = table_span_points(p.L);
    if (!sp || n_pts < 2 * sp) return;  // slices only pay when there are at least two full ones
    S = (n_pts + sp - 1) / sp;
    if (S > DEV_SLICES_MAX) S = DEV_SLICES_MAX;
    per = (n_pts + S - 1) / S;
    S = (n_pts + per - 1) / per;  // no empty slice
}
// A slice's sorted references need not be table rows (L * n of them): level << shift | point-of-the-slice takes fewer bits, which
// leaves more fine-key bits in a 32-bit intermediate entry and therefore fewer coarse bins — at 2^22 points the slices then sort under
// the 2^20 plan (7 fine bits, 512 bins: 130 us) instead of the handle's (5 bits, 2048 bins: 181 us). The second pass writes table rows
// into the final list (msm_fine_place_kernel). Returns false when the slice plan is no finer than the handle's.
static void plan_two_pass(MsmPlan &p, size_t table_rows, size_t n_total);
static bool slice_sort_plan(const MsmPlan &p, size_t per, MsmPlan &ps, int &shift) {
    shift = 0;
    ps = p;
    if (!p.NT || !env_int("ZG_MSM_SLICE_LOCAL_REFS", 1)) return false;  // (p.fb == 0: the handle's own launch sets sort in one pass)
    int k = 1;
    while (((size_t)1 << k) < per) k++;
    plan_two_pass(ps, (size_t)p.L << k, per);
    if (ps.fb <= p.fb) {
        ps = p;
        return false;
    }
    shift = k;
    return true;
}

static size_t fine_max_items(const MsmPlan &p, size_t n_total) { return (size_t)p.NCB + (size_t)p.W * n_total / FINE_SLICE + 1; }

static void lane_free(zg_bases_s::Lane &ln) {
    void *lp[] = {ln.d_dig, ln.d_sorted, ln.d_hist, ln.d_starts, ln.d_blockhist, ln.d_tmp, ln.d_cstarts, ln.d_fine, ln.d_partial, ln.d_slice_buckets, ln.d_slice_meta, ln.d_bits, ln.d_rg,
                  ln.d_nzrank, ln.d_nzlist, ln.d_scan_tmp, ln.d_part, ln.d_part2, ln.d_heavy, ln.d_state};
    for (void *p : lp) pool_free(p);
    if (ln.done) (void)hipEventDestroy(ln.done);
    ln = zg_bases_s::Lane();
}

static hipError_t lane_malloc(void **p, size_t bytes) {  // from the device pool, like the handle's table (free_bases)
    *p = pool_alloc(bytes ? bytes : 16);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
// workspace of one MSM launch set: n_total scalars (all scalar vectors of a batched launch together) under plan p
static hipError_t lane_alloc(zg_bases_s::Lane &ln, const MsmPlan &p, size_t n_total, uint32_t nblk_lds) {
    hipError_t e = hipSuccess;
    auto A = [&](auto &ptr, size_t bytes) {
        if (e == hipSuccess) e = lane_malloc((void **)&ptr, bytes);
    };
    A(ln.d_dig, (size_t)p.W * n_total * 4);
    A(ln.d_sorted, (size_t)p.W * n_total * 4);
    A(ln.d_hist, (size_t)p.NK * 4);
    A(ln.d_starts, ((size_t)p.NK + 1) * 4);
    {
        // two-pass sort buffers: for the handle's own plan (p.fb), for the plan of a point slice (slice_sort_plan: a handle whose own
        // launch sets sort in one pass — 2^23 points and up, where a table row index leaves fewer than 5 fine bits — still sorts its
        // slices in two), or both; every buffer takes the larger of the two needs
        size_t S, per;
        MsmPlan ps;
        int shift;
        slice_counts(p, n_total, S, per);
        const bool slice_two_pass = S > 1 && slice_sort_plan(p, per, ps, shift);
        size_t bh = 0, tmp = 0, cst = 0, fine = 0;
        if (p.fb) {
            bh = (size_t)nblk_lds * p.NCB;
            tmp = (size_t)p.W * n_total + 4 * (size_t)p.NCB + 4;
            cst = 4 * (size_t)p.NCB + 8;
            fine = (fine_max_items(p, n_total) + (size_t)p.NCB) * ((size_t)1 << p.fb);
        } else if (nblk_lds) {
            bh = (size_t)nblk_lds * p.NK;
        }
        if (slice_two_pass) {
            const size_t nblk_s = div_up(per, two_pass_span(p.W));
            bh = std::max(bh, nblk_s * ps.NCB);
            tmp = std::max(tmp, (size_t)p.W * per + 4 * (size_t)ps.NCB + 4);
            cst = std::max(cst, 4 * (size_t)ps.NCB + 8);
            fine = std::max(fine, (fine_max_items(ps, per) + (size_t)ps.NCB) * ((size_t)1 << ps.fb));
        }
        if (bh) A(ln.d_blockhist, bh * 4);
        if (tmp) A(ln.d_tmp, tmp * 4);
        if (cst) A(ln.d_cstarts, cst * 4);
        if (fine) A(ln.d_fine, fine * 4);
        ln.blockhist_words = bh; ln.tmp_words = tmp; ln.cstarts_words = cst; ln.fine_words = fine;
    }
    A(ln.d_partial, (size_t)p.NK * 144);
    {
        size_t per_group = (size_t)p.c * p.PB;  // one-dimensional bit sums; the two-dimensional form keeps rows + columns + c sums
        if (p.lb && ((size_t)1 << p.lb) + ((size_t)1 << p.hb) + p.c > per_group) per_group = ((size_t)1 << p.lb) + ((size_t)1 << p.hb) + p.c;
        A(ln.d_bits, (size_t)p.G * p.K * per_group * 144);
    }
    A(ln.d_rg, (size_t)p.G * p.K * 128);
    A(ln.d_nzrank, ((size_t)p.NK + 1) * 4);
    A(ln.d_nzlist, (size_t)p.NK * 4);
    A(ln.d_scan_tmp, (2 * (size_t)p.NK + 2 * (p.NK / 1024 + 1)) * 4);
    if (p.NT) {
        size_t slots = (size_t)p.NT + p.NK;
        A(ln.d_part, slots * 144);
        A(ln.d_part2, (slots / HEAVY_BLOCK_ITEMS + 1 + p.NK) * 144);
        A(ln.d_heavy, (size_t)p.NK * 8);
        A(ln.d_state, state_words(p.NT, p.NK) * 4);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ln.done, hipEventDisableTiming);
    return e;
}

// A handle's table and workspaces come from the device pool (runtime.hip) like every transient allocation: a prover that rebuilds its key
// per proof (or a test that uploads bases per case) reuses the blocks instead of paying hipMalloc / hipFree of gigabytes — 26 ms against
// 113 ms for HyperKZG.setup at 2^20 powers when another process holds a context on the same GPU (bench.py's child runs, round 5). The blocks
// go back only after the device is idle (hipFree used to imply that).
static hipError_t handle_malloc(void **p, size_t bytes) {
    *p = pool_alloc(bytes ? bytes : 16);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
static void free_bases(zg_bases_s *b) {
    if (!b) return;
    (void)hipDeviceSynchronize();
    free_bases(b->small);
    void *ptrs[] = {b->d_table, b->d_inf, b->d_scal, b->d_out, b->d_slice_parts, b->d_rows_buckets, b->d_rows_bits, b->d_rows_rg};
    for (void *p : ptrs) pool_free(p);
    if (b->rows_done) (void)hipEventDestroy(b->rows_done);
    for (auto &ln : b->lanes) lane_free(ln);
    lane_free(b->batch_lane);
    for (int i = 0; i < zg_bases_s::NAUX; i++) {
        if (b->aux[i]) (void)hipStreamDestroy(b->aux[i]);
        if (b->ev_join[i]) (void)hipEventDestroy(b->ev_join[i]);
    }
    if (b->ev_fork) (void)hipEventDestroy(b->ev_fork);
    if (b->h_out) (void)hipHostFree(b->h_out);
    delete b;
}

#define ZG_ALLOC(ptr, bytes)                                                         \
    do {                                                                             \
        hipError_t _e = handle_malloc((void **)&(ptr), (bytes));                      \
        if (_e != hipSuccess) {                                                      \
            set_error(std::string("hipMalloc(" #ptr "): ") + hipGetErrorString(_e)); \
            free_bases(b);                                                           \
            return _e == hipErrorOutOfMemory ? ZG_ERR_NOMEM : ZG_ERR_HIP;            \
        }                                                                            \
    } while (0)

static constexpr size_t SIDE_TABLE_POINTS = 16384;

// defer_table: everything but the table (plan, allocations, workspaces) — the caller builds it (a handle's side table is built by the
// launch that builds its main table: PreJob).
static int bases_create(const uint64_t *d_xy, const uint8_t *d_inf_in, size_t n, const zg_msm_config *cfg, hipStream_t st,
                        zg_bases_t *out, bool defer_table = false) {
    if (n >= (1ull << 27)) {
        set_error("msm: at most 2^27 bases per handle");
        return ZG_ERR_INVALID;
    }
    zg_bases_s *b = new zg_bases_s();
    b->n = n;
    b->device = current_device();
    int rc = make_plan(n ? n : 1, cfg, b->plan);
    if (rc != ZG_OK) {
        free_bases(b);
        return rc;
    }
    const MsmPlan &p = b->plan;
    if ((uint64_t)p.L * n >= (1ull << 31)) {
        set_error("msm: precompute table too large for 31-bit point references");
        free_bases(b);
        return ZG_ERR_INVALID;
    }
    ZG_ALLOC(b->d_table, (size_t)p.L * n * 64);
    if (d_inf_in) ZG_ALLOC(b->d_inf, n);
    int nlanes = env_int("ZG_MSM_LANES", 6);  // 2^17-point MSMs: 0.36 ms per MSM with 3 in flight, 0.25 with 6 (tools/bench_tail.py)
    if (nlanes < 1) nlanes = 1;
    if (nlanes > 8) nlanes = 8;
    // a one-shot handle (MSM.compute on a temporary slice) serves one MSM: one workspace — a workspace is ~18 allocations, and the
    // upload + free of such a handle is most of what its caller pays (tools/crossover.py: one_shot_msm)
    if (cfg && cfg->expected_uses > 0 && cfg->expected_uses < 16 && !getenv("ZG_MSM_LANES")) nlanes = 1;
    bool lds_sort = (size_t)p.NK * 4 <= 128 * 1024 && env_int("ZG_MSM_LDS_SORT", 1);
    plan_two_pass(b->plan, (size_t)p.L * n, n);
    if (p.fb) {
        uint32_t nblk = (uint32_t)div_up(n, two_pass_span(p.W));
        b->nblk = nblk < 1 ? 1 : nblk;
    } else if (lds_sort) {
        uint32_t nblk = (uint32_t)(n / (size_t)sort_span(p.NK));
        b->nblk = nblk < 1 ? 1 : (nblk > 256 ? 256 : nblk);
    }
    b->lanes.resize(nlanes);
    for (auto &ln : b->lanes) {
        hipError_t le = lane_alloc(ln, p, n, (p.fb || lds_sort) ? b->nblk : 0);
        if (le != hipSuccess) {
            set_error(std::string("msm workspace: ") + hipGetErrorString(le));
            free_bases(b);
            return le == hipErrorOutOfMemory ? ZG_ERR_NOMEM : ZG_ERR_HIP;
        }
    }
    ZG_ALLOC(b->d_out, 16 * 8);
    if (hipHostMalloc((void **)&b->h_out, 16 * 8) != hipSuccess) {
        set_error("hipHostMalloc failed");
        free_bases(b);
        return ZG_ERR_NOMEM;
    }
    // The side table (narrow windows over the first SIDE_TABLE_POINTS bases, for short MSMs) is a handle of its own; its table is built by
    // the first launch of THIS handle's table kernel (PreJob): created here without a table.
    const bool want_small = !defer_table && (size_t)p.NB * p.G > 4096 && n > SIDE_TABLE_POINTS && env_int("ZG_MSM_SIDE_TABLE", 1);
    if (want_small) {
        zg_msm_config small_cfg{8, 0, 0};
        int src = bases_create(d_xy, d_inf_in, SIDE_TABLE_POINTS, &small_cfg, st, &b->small, true);
        if (src != ZG_OK) {
            b->small = nullptr;
            free_bases(b);
            return src;
        }
    }
    if (defer_table) {  // (the inf flags are copied by the caller's launch sequence, on the same stream)
        *out = b;
        return ZG_OK;
    }
    // the table kernel's per-level records (see msm_precompute_kernel); without them (allocation refused, ZG_MSM_PRECOMPUTE_V1) the
    // round-4 kernel builds the same table with an inversion per level. Released after the synchronisation below.
    const size_t pre_levels = p.L > 1 ? (size_t)(p.L - 1 < PRE_GROUP ? p.L - 1 : PRE_GROUP) : 0;
    // ... for at most PRE_CHUNK bases at a time (2^21: 2.1 GB of records whatever the handle's size — a 2^24-base handle would ask for 17 GB
    // beside its 16 GiB table, and a fresh allocation of that size costs more than the kernel it serves)
    const size_t pre_chunk = n < PRE_CHUNK ? (n ? n : 1) : PRE_CHUNK;
    const bool v1 = env_int("ZG_MSM_PRECOMPUTE_V1", 0) != 0;
    Scratch pre_scratch, side_scratch;
    if (n && pre_levels && !v1 && !pre_scratch.alloc(pre_levels * pre_chunk * 144)) (void)hipGetLastError();
    const zg_bases_s *sm = b->small;
    if (sm && !v1 && !side_scratch.alloc((size_t)PRE_GROUP * sm->n * 144)) (void)hipGetLastError();
    {
        hipError_t e = hipSuccess;
        if (n) {
            if (d_inf_in) e = hipMemcpyAsync(b->d_inf, d_inf_in, n, hipMemcpyDeviceToDevice, st);
            if (e == hipSuccess && sm && d_inf_in) e = hipMemcpyAsync(sm->d_inf, d_inf_in, sm->n, hipMemcpyDeviceToDevice, st);
            if (e == hipSuccess) {
                PreJob side{};  // no workgroups unless the side table rides along
                bool side_pending = sm != nullptr;
                if (side_pending && side_scratch.p)
                    side = PreJob{d_xy, sm->d_inf, sm->n, sm->plan.L, sm->plan.c * sm->plan.G, sm->d_table, side_scratch.as<char>(), 0, sm->n, (unsigned)div_up(sm->n, 256)};
                if (pre_scratch.p || p.L == 1) {
                    // (a handle without a table, L == 1, needs no records: its job only converts the bases; the side table still rides along)
                    for (size_t first = 0; first < n; first += pre_chunk) {  // the launches reuse the records one after the other (stream order)
                        const size_t count = n - first < pre_chunk ? n - first : pre_chunk;
                        const PreJob mj{d_xy, b->d_inf, n, p.L, p.c * p.G, b->d_table, pre_scratch.as<char>(), first, count, (unsigned)div_up(count, 256)};
                        hipLaunchKernelGGL(msm_precompute_kernel, dim3(side.blocks + mj.blocks), dim3(256), 0, st, side, mj);
                        if (side.blocks) side_pending = false;
                        side = PreJob{};
                    }
                } else {
                    hipLaunchKernelGGL(msm_precompute_v1_kernel, dim3(div_up(n, 256)), dim3(256), 0, st, d_xy, b->d_inf, n, p.L, p.c * p.G,
                                       b->d_table);
                }
                if (side_pending)  // no records for it (allocation refused / ZG_MSM_PRECOMPUTE_V1): the round-4 kernel, on its own
                    hipLaunchKernelGGL(msm_precompute_v1_kernel, dim3(div_up(sm->n, 256)), dim3(256), 0, st, d_xy, sm->d_inf, sm->n, sm->plan.L,
                                       sm->plan.c * sm->plan.G, sm->d_table);
                e = hipGetLastError();
            }
        }
        hipError_t e2 = hipStreamSynchronize(st);  // also on failure: nothing of this handle may still be in flight when it is freed
        if (e == hipSuccess) e = e2;
        if (e != hipSuccess) {
            set_error(std::string("msm table build: ") + hipGetErrorString(e));
            free_bases(b);
            return ZG_ERR_HIP;
        }
    }
    *out = b;
    return ZG_OK;
}

template <int C>
static void launch_digits(hipStream_t st, const uint64_t *sc, const uint8_t *inf, uint32_t n, uint32_t n_pts, int G, uint32_t *dig,
                          uint32_t *hist) {
    hipLaunchKernelGGL(msm_digits_kernel<C>, dim3(div_up(n, 256)), dim3(256), 0, st, sc, inf, n, n_pts, G, dig, hist);
}

// threads per block of the two LDS sort kernels (one block per CU either way: each needs all bucket counters in LDS)
static unsigned sort_threads() {
    static const unsigned t = [] {
        int v = env_int("ZG_MSM_SORT_THREADS", 512);  // 8 waves: fit beside two accumulate workgroups of another stream (measured +3.7 % MSM/s)
        return (unsigned)(v >= 1024 ? 1024 : (v >= 512 ? 512 : 256));
    }();
    return t;
}

template <int C>
static int launch_digits_lds(hipStream_t st, const uint64_t *sc, const uint8_t *inf, uint32_t n, uint32_t n_pts, int G, uint32_t per_block,
                             uint32_t NK, uint32_t nblk, uint32_t *dig, uint32_t *blockhist, int shift, unsigned threads, const RowOffs &rows) {
    static PerDeviceOnce once;  // per instantiation and device; MSM entry points are re-entrant (std.Thread workers call MSM.compute)
    ZG_HIP(once.run([] {
        return hipFuncSetAttribute(reinterpret_cast<const void *>(msm_digits_lds_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    }));
    hipLaunchKernelGGL(msm_digits_lds_kernel<C>, dim3(nblk), dim3(threads), NK * 4, st, sc, inf, n, n_pts, G, per_block, NK, shift, dig, blockhist, rows);
    return ZG_OK;
}

static int launch_digits_lds_c(int c, hipStream_t st, const uint64_t *sc, const uint8_t *inf, uint32_t n, uint32_t n_pts, int G,
                               uint32_t per_block, uint32_t NK, uint32_t nblk, uint32_t *dig, uint32_t *blockhist, int shift = 0,
                               unsigned threads = 0, const RowOffs *rows_in = nullptr) {
    static const RowOffs uniform{};
    const RowOffs &rows = rows_in ? *rows_in : uniform;
    if (!threads) threads = sort_threads();
    switch (c) {
#define ZG_CASE(C) case C: return launch_digits_lds<C>(st, sc, inf, n, n_pts, G, per_block, NK, nblk, dig, blockhist, shift, threads, rows);
        ZG_CASE(2) ZG_CASE(3) ZG_CASE(4) ZG_CASE(5) ZG_CASE(6) ZG_CASE(7) ZG_CASE(8) ZG_CASE(9) ZG_CASE(10)
        ZG_CASE(11) ZG_CASE(12) ZG_CASE(13) ZG_CASE(14) ZG_CASE(15) ZG_CASE(16) ZG_CASE(17) ZG_CASE(18) ZG_CASE(19)
#undef ZG_CASE
        default: set_error("msm: unsupported window size"); return ZG_ERR_INVALID;
    }
}

static int launch_digits_c(int c, hipStream_t st, const uint64_t *sc, const uint8_t *inf, uint32_t n, uint32_t n_pts, int G, uint32_t *dig,
                           uint32_t *hist) {
    switch (c) {
#define ZG_CASE(C) case C: launch_digits<C>(st, sc, inf, n, n_pts, G, dig, hist); break;
        ZG_CASE(2) ZG_CASE(3) ZG_CASE(4) ZG_CASE(5) ZG_CASE(6) ZG_CASE(7) ZG_CASE(8) ZG_CASE(9) ZG_CASE(10)
        ZG_CASE(11) ZG_CASE(12) ZG_CASE(13) ZG_CASE(14) ZG_CASE(15) ZG_CASE(16) ZG_CASE(17) ZG_CASE(18) ZG_CASE(19)
#undef ZG_CASE
        default: set_error("msm: unsupported window size"); return ZG_ERR_INVALID;
    }
    return ZG_OK;
}

static int two_pass_attrs() {
    static PerDeviceOnce once;
    ZG_HIP(once.run([] {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(msm_partition_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
        if (err == hipSuccess)
            err = hipFuncSetAttribute(reinterpret_cast<const void *>(msm_partition_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
        if (err == hipSuccess)
            err = hipFuncSetAttribute(reinterpret_cast<const void *>(msm_fine_place_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
        return err;
    }));
    return ZG_OK;
}

static int msm_enqueue_lane(zg_bases_s *b, const MsmPlan &p, zg_bases_s::Lane &ln, uint32_t nblk_cap, size_t off, size_t n_pts,
                            const uint64_t *d_scalars, hipStream_t st, int mode, uint64_t *d_rec, uint8_t *d_inf_out,
                            uint32_t rec_stride, uint32_t inf_stride);

static int ensure_aux_streams(zg_bases_s *b) {
    if (b->aux[0]) return ZG_OK;
    hipError_t e = hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming);
    for (int i = 0; i < zg_bases_s::NAUX && e == hipSuccess; i++) {
        e = hipStreamCreateWithFlags(&b->aux[i], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&b->ev_join[i], hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        set_error(std::string("msm helper streams: ") + hipGetErrorString(e));
        return ZG_ERR_HIP;
    }
    return ZG_OK;
}

// ---- point slices. The accumulate kernel gathers one random 64-byte table row per addition; the rows of one launch set span
// L * n * 64 bytes, and random rows over more than ~1-2 GiB run at half the rate of rows inside 1 GiB on this part (translation
// reach: tools/microbench_tlb — 56 G rows/s inside 1 GiB, 25 G over 4 GiB, however the 4 GiB are allocated). At 2^22 points
// (4 GiB table) that cost the kernel +10 % per addition, at 2^24 (16 GiB) +45 %. A long MSM is therefore cut into slices of
// consecutive points whose L table segments together stay within ZG_MSM_TABLE_SPAN_MB (default 1024): slice j is sorted and
// accumulated by itself (same workspace, same stream, one after the other) into its own set of bucket sums, one launch adds
// the sets up bucket by bucket, and the reduction runs ONCE — the fixed tail is not multiplied (slices as separate MSMs with
// a Jacobian combine were measured first: 4 x 0.45 ms of sorts and tails ate the whole gain). The sums are the same group
// elements, so the result bytes are those of the unsliced launch set.
static size_t table_span_points(const zg_bases_s *b) { return table_span_points(b->plan.L); }
struct SliceView {  // what the sort of a point slice hands to its accumulation
    uint32_t *sorted, *starts, *nzrank, *nzlist;
    void *state;
};

// bucket set 0 += sets 1 .. S-1 (set j at sets + (j-1) * NK * 144), one quad of lanes per bucket
__global__ void __launch_bounds__(256) msm_bucket_fold_kernel(char *buckets, const char *sets, uint32_t NK, uint32_t S) {
    ZG_HIPRIO();
    uint32_t k = (blockIdx.x * 256 + threadIdx.x) >> 2, q = threadIdx.x & 3;
    if (k >= NK) return;
    XYZZ29 acc = xyzz29_load(buckets + 144 * (size_t)k);
    XYZZ29 nxt = xyzz29_load(sets + 144 * (size_t)k);
    for (uint32_t j = 1; j < S; j++) {
        XYZZ29 cur = nxt;
        if (j + 1 < S) nxt = xyzz29_load(sets + 144 * ((size_t)j * NK + k));
        acc = xyzz29_add4(acc, cur, q);
    }
    if (q == 0) xyzz29_store(buckets + 144 * (size_t)k, acc);
}

// Enqueue one MSM over bases[off, off+n) on `st`; result record lands in d_rec / d_inf_out.
static int msm_enqueue(zg_bases_s *b, size_t off, size_t n, const uint64_t *d_scalars, hipStream_t st, int mode, uint64_t *d_rec,
                       uint8_t *d_inf_out) {
    if (off + n > b->n) {
        set_error("msm: range exceeds uploaded bases");
        return ZG_ERR_INVALID;
    }
    if (n == 0) {  // msm/mod.zig:361-363
        hipLaunchKernelGGL(msm_identity_kernel, dim3(1), dim3(1), 0, st, mode, d_rec, d_inf_out);
        ZG_HIP(hipGetLastError());
        return ZG_OK;
    }
    if (b->small && off + n <= b->small->n) return msm_enqueue(b->small, off, n, d_scalars, st, mode, d_rec, d_inf_out);  // short prefix
    zg_bases_s::Lane &ln = b->lanes[b->next_lane];
    b->next_lane = (b->next_lane + 1) % b->lanes.size();
    return msm_enqueue_lane(b, b->plan, ln, b->nblk, off, n, d_scalars, st, mode, d_rec, d_inf_out, 0, 0);
}

// Live row lengths of the zero-padded matrix the next launch set sorts (msm_batch_dev_wide_rows sets it around its call; the sort reads
// it when the set is the whole matrix): see RowOffs.
static thread_local const RowOffs *t_row_offs = nullptr;
// Where the next launch set leaves its bucket sums INSTEAD of reducing them (msm_rows_shared_tail sets it around each row): the set ends
// after its accumulation, the caller reduces the rows' buckets together.
static thread_local char *t_bucket_sink = nullptr;

// One launch set on workspace `ln` under plan `p`: p.K scalar vectors of n_pts scalars each, stored back to back at
// d_scalars, all over bases[off, off+n_pts); vector i's record lands at d_rec + i*rec_stride / d_inf_out + i*inf_stride.
static int msm_enqueue_lane(zg_bases_s *b, const MsmPlan &p, zg_bases_s::Lane &ln, uint32_t nblk_cap, size_t off, size_t n_pts,
                            const uint64_t *d_scalars, hipStream_t st, int mode, uint64_t *d_rec, uint8_t *d_inf_out,
                            uint32_t rec_stride, uint32_t inf_stride) {
    if (ln.used) ZG_HIP(hipStreamWaitEvent(st, ln.done, 0));  // the lane's previous MSM may be on another stream
    ln.used = true;
    ln.last_st = st;
    // point slices (see table_span_points): S > 1 only for one scalar vector over a table wider than the span
    size_t S, per;
    slice_counts(p, n_pts, S, per);
    MsmPlan ps = p;  // the sort plan of a slice (slice_sort_plan)
    int local_shift = 0;
    if (S > 1 && slice_sort_plan(p, per, ps, local_shift)) {
        const size_t nblk_s = div_up(per, two_pass_span(p.W));
        if ((fine_max_items(ps, per) + (size_t)ps.NCB) * ((size_t)1 << ps.fb) > ln.fine_words || nblk_s * ps.NCB > ln.blockhist_words ||
            (size_t)p.W * per + 4 * (size_t)ps.NCB + 4 > ln.tmp_words || 4 * (size_t)ps.NCB + 8 > ln.cstarts_words) {
            ps = p;  // the workspace was sized under another slice setting: keep table-row references and the handle's plan
            local_shift = 0;
        }
    }
    // per slice: bucket starts, non-empty ranks, non-empty list, reduction state (what a sort hands to its accumulation), 16-byte aligned
    const size_t meta_stride = (3 * ((size_t)p.NK + 1) + state_words(p.NT, p.NK) + 3) & ~(size_t)3;
    if (S > 1 && ln.slice_buckets < S - 1) {
        if (ln.d_slice_buckets || ln.d_slice_meta) ZG_HIP(hipDeviceSynchronize());  // once per lane and slice count: the old buffers may be in use
        pool_free(ln.d_slice_buckets);
        ln.d_slice_buckets = nullptr;
        ln.slice_buckets = 0;
        ZG_HIP(lane_malloc((void **)&ln.d_slice_buckets, (S - 1) * (size_t)p.NK * 144));
        pool_free(ln.d_slice_meta);
        ln.d_slice_meta = nullptr;
        ZG_HIP(lane_malloc((void **)&ln.d_slice_meta, S * meta_stride * 4));
        ln.slice_buckets = S - 1;
    }
    // digits and sort of bases[off, off + n_pts) x p.K scalar vectors at d_scalars -> sv (sorted references, bucket starts, ...)
    auto sort_range = [&](size_t off, size_t n_pts, const uint64_t *d_scalars, const SliceView &sv) -> int {
#ifdef ZG_EXP_SKIP_SORT  // timing experiment only (tools/build_variant.sh): a workspace's second and later MSMs reuse its sorted list
    if (ln.exp_sorted_once) return ZG_OK;
    ln.exp_sorted_once = true;
#endif
    const MsmPlan &q = local_shift ? ps : p;  // fine bits / coarse bins of this launch
    // rows by their live lengths: the LDS-histogram sorts only (the global-atomic sort of very small sets walks the padded matrix)
    const RowOffs *rows = t_row_offs && S == 1 && t_row_offs->k == (uint32_t)p.K && (q.fb || (nblk_cap && ln.d_blockhist)) ? t_row_offs : nullptr;
    static const RowOffs uniform_rows{};
    const RowOffs &rowv = rows ? *rows : uniform_rows;
    const size_t n = rows ? rows->off[rows->k] : n_pts * (size_t)p.K;  // scalars in this launch set
    const uint8_t *infp = b->d_inf ? b->d_inf + off : nullptr;
    if (q.fb) {
        // two-pass sort: blocks of 256 threads over TWO_PASS_SPAN scalars each (coarse counters are a few KiB of LDS)
        uint32_t nblk = (uint32_t)div_up(n, two_pass_span(p.W));  // per_block * W <= STAGE_ENTRIES
        if ((size_t)nblk * q.NCB > ln.blockhist_words) {
            set_error("msm: two-pass workspace too small for this launch");
            return ZG_ERR_INVALID;
        }
        uint32_t per_block = (uint32_t)((n + nblk - 1) / nblk);
        prof_begin(ZG_PROF_MSM_DIGITS, st);
        ZG_TRY(launch_digits_lds_c(p.c, st, d_scalars, infp, (uint32_t)n, (uint32_t)n_pts, p.G, per_block, q.NCB, nblk, ln.d_dig,
                                   ln.d_blockhist, q.fb, 256, rows));
        prof_end(ZG_PROF_MSM_DIGITS, st);
        prof_begin(ZG_PROF_MSM_SORT, st);
        uint32_t *d_tot = ln.d_cstarts + q.NCB + 1, *d_tst = ln.d_cstarts + 2 * (size_t)q.NCB + 2, *d_ist = ln.d_cstarts + 3 * (size_t)q.NCB + 3;
        hipLaunchKernelGGL(msm_colscan_bins_kernel, dim3(div_up(q.NCB, 16)), dim3(1024), 0, st, ln.d_blockhist, nblk, q.NCB, d_tot);
        hipLaunchKernelGGL(msm_coarse_base_kernel, dim3(1), dim3(1024), 0, st, d_tot, q.NCB, ln.d_cstarts, d_tst, d_ist);
        ZG_TRY(two_pass_attrs());
        if (!rows && p.G == 1 && n == n_pts)  // PLAIN uses i as the point index: never with live row lengths (their total may equal n_pts)
            hipLaunchKernelGGL(msm_partition_kernel<true>, dim3(nblk), dim3(1024), (STAGE_ENTRIES + 2 * (size_t)q.NCB + 1 + 1024) * 4, st, ln.d_dig,
                               (uint32_t)n, (uint32_t)n_pts, p.W, p.G, b->n, (uint32_t)off, per_block, q.NCB, q.fb, q.rb, d_tst, ln.d_blockhist,
                               ln.d_tmp, local_shift, rowv);
        else
            hipLaunchKernelGGL(msm_partition_kernel<false>, dim3(nblk), dim3(1024), (STAGE_ENTRIES + 2 * (size_t)q.NCB + 1 + 1024) * 4, st, ln.d_dig,
                               (uint32_t)n, (uint32_t)n_pts, p.W, p.G, b->n, (uint32_t)off, per_block, q.NCB, q.fb, q.rb, d_tst, ln.d_blockhist,
                               ln.d_tmp, local_shift, rowv);
        {
            uint32_t items = (uint32_t)fine_max_items(q, n);
            uint32_t *d_fbase = ln.d_fine + (size_t)items * ((size_t)1 << q.fb);
            hipLaunchKernelGGL(msm_fine_count_kernel, dim3(items), dim3(1024), 0, st, ln.d_tmp, ln.d_cstarts, d_tst, d_ist, q.NCB, q.fb, q.rb,
                               ln.d_fine);
            hipLaunchKernelGGL(msm_fine_offsets_kernel, dim3(q.NCB), dim3(128), 0, st, d_ist, q.fb, p.NK, ln.d_fine, d_fbase, ln.d_hist);
            hipLaunchKernelGGL(msm_fine_place_kernel, dim3(items), dim3(1024), (STAGE_ENTRIES + 128 + 132) * 4, st, ln.d_tmp, ln.d_cstarts, d_tst,
                               d_ist, q.NCB, q.fb, q.rb, ln.d_fine, d_fbase, sv.sorted, local_shift, (uint32_t)b->n, (uint32_t)off);
        }
        uint32_t tiles = div_up(p.NK, 1024);
        hipLaunchKernelGGL(msm_scan_a_kernel, dim3(tiles), dim3(1024), 0, st, ln.d_hist, p.NK, ln.d_scan_tmp, ln.d_scan_tmp + p.NK,
                           ln.d_scan_tmp + 2 * (size_t)p.NK);
        hipLaunchKernelGGL(msm_scan_b_kernel, dim3(tiles), dim3(1024), 0, st, ln.d_hist, p.NK, ln.d_scan_tmp, ln.d_scan_tmp + p.NK,
                           ln.d_scan_tmp + 2 * (size_t)p.NK, sv.starts, sv.nzrank, sv.nzlist, reinterpret_cast<uint32_t *>(sv.state),
                           sv.state ? state_words(p.NT, p.NK) : 0u);
    } else if (nblk_cap && ln.d_blockhist) {  // the handle's own plan sorts in LDS (a non-null d_blockhist alone may belong to a point slice's two-pass plan)
        uint32_t nblk = nblk_cap;
        while (nblk > 1 && (size_t)(nblk - 1) * 1024 >= n) nblk--;  // no empty blocks for short sub-range MSMs
        uint32_t per_block = (uint32_t)((n + nblk - 1) / nblk);
        prof_begin(ZG_PROF_MSM_DIGITS, st);
        ZG_TRY(launch_digits_lds_c(p.c, st, d_scalars, infp, (uint32_t)n, (uint32_t)n_pts, p.G, per_block, p.NK, nblk, ln.d_dig,
                                   ln.d_blockhist, 0, 0, rows));
        prof_end(ZG_PROF_MSM_DIGITS, st);
        prof_begin(ZG_PROF_MSM_SORT, st);
        hipLaunchKernelGGL(msm_colscan_kernel, dim3(div_up(p.NK, 256)), dim3(256), 0, st, ln.d_blockhist, nblk, p.NK, ln.d_hist);
        {
            uint32_t tiles = div_up(p.NK, 1024);
            hipLaunchKernelGGL(msm_scan_a_kernel, dim3(tiles), dim3(1024), 0, st, ln.d_hist, p.NK, ln.d_scan_tmp, ln.d_scan_tmp + p.NK,
                               ln.d_scan_tmp + 2 * (size_t)p.NK);
            hipLaunchKernelGGL(msm_scan_b_kernel, dim3(tiles), dim3(1024), 0, st, ln.d_hist, p.NK, ln.d_scan_tmp, ln.d_scan_tmp + p.NK,
                               ln.d_scan_tmp + 2 * (size_t)p.NK, sv.starts, sv.nzrank, sv.nzlist, reinterpret_cast<uint32_t *>(sv.state),
                           sv.state ? state_words(p.NT, p.NK) : 0u);
        }
        static PerDeviceOnce scatter_once;
        ZG_HIP(scatter_once.run([] {
            return hipFuncSetAttribute(reinterpret_cast<const void *>(msm_scatter_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        }));
        hipLaunchKernelGGL(msm_scatter_lds_kernel, dim3(nblk), dim3(sort_threads()), p.NK * 4, st, ln.d_dig, (uint32_t)n, (uint32_t)n_pts, p.W, p.G,
                           b->n, (uint32_t)off, per_block, p.NK, sv.starts, ln.d_blockhist, sv.sorted, rowv);
    } else {
        prof_begin(ZG_PROF_MSM_DIGITS, st);
        ZG_HIP(hipMemsetAsync(ln.d_hist, 0, (size_t)p.NK * 4, st));
        ZG_TRY(launch_digits_c(p.c, st, d_scalars, infp, (uint32_t)n, (uint32_t)n_pts, p.G, ln.d_dig, ln.d_hist));
        prof_end(ZG_PROF_MSM_DIGITS, st);
        prof_begin(ZG_PROF_MSM_SORT, st);
        {
            uint32_t tiles = div_up(p.NK, 1024);
            hipLaunchKernelGGL(msm_scan_a_kernel, dim3(tiles), dim3(1024), 0, st, ln.d_hist, p.NK, ln.d_scan_tmp, ln.d_scan_tmp + p.NK,
                               ln.d_scan_tmp + 2 * (size_t)p.NK);
            hipLaunchKernelGGL(msm_scan_b_kernel, dim3(tiles), dim3(1024), 0, st, ln.d_hist, p.NK, ln.d_scan_tmp, ln.d_scan_tmp + p.NK,
                               ln.d_scan_tmp + 2 * (size_t)p.NK, sv.starts, sv.nzrank, sv.nzlist, reinterpret_cast<uint32_t *>(sv.state),
                           sv.state ? state_words(p.NT, p.NK) : 0u);
        }
        ZG_HIP(hipMemsetAsync(ln.d_hist, 0, (size_t)p.NK * 4, st));
        hipLaunchKernelGGL(msm_scatter_kernel, dim3(div_up(n, 256), p.W), dim3(256), 0, st, ln.d_dig, (uint32_t)n, (uint32_t)n_pts, p.G,
                           b->n, (uint32_t)off, sv.starts, ln.d_hist, sv.sorted);
    }
    prof_end(ZG_PROF_MSM_SORT, st);
    return ZG_OK;
    };
    // accumulation and bucket sums of a sorted range -> bucket_out
    auto accumulate_range = [&](size_t n_pts, const SliceView &sv, char *bucket_out) -> int {
    const size_t n = n_pts * (size_t)p.K;
    prof_begin(ZG_PROF_MSM_ACCUMULATE, st);
    if (p.NT) {
        // a launch over a sub-range of the handle (a short prefix, the last set of a batch) gets as many chunks as ITS digits
        // warrant, never more than the workspace was sized for
        bool alone = env_int("ZG_MSM_ALONE_FULL", 1) != 0;
        // (round 6) an MSM still in flight ON THIS STREAM is not company: its kernels finish before this launch starts (stream order), so
        // there is nothing the spare registers could run under. Only work on OTHER streams makes the 7/8 launch pay; a caller that
        // pipelines on one stream gets the full grid (serial bench at 2^20: accumulate 1.30 -> 1.18 ms per launch).
        for (auto &o : b->lanes)
            if (alone && &o != &ln && o.used && o.last_st != st && hipEventQuery(o.done) == hipErrorNotReady) alone = false;
        (void)hipGetLastError();  // hipErrorNotReady is an answer, not a failure
        // a table-less launch set (one bucket set per window: its reduction is a few hundred short workgroups, not a latency chain under
        // someone else's accumulation) takes every slot either way: 612 MSM/s against 587 / 596 / 603 at 15/16, 7/8, 13/16 of them
        if (p.G > 1) alone = true;
        uint32_t NT = chunk_threads((uint64_t)n * p.W, alone);
        if (NT > p.NT || getenv("ZG_MSM_CHUNK_THREADS")) NT = p.NT;
        // threads per accumulate workgroup (64 / 128 / 256). Smaller workgroups spread evenly over the CUs and lose what the 7/8 launch
        // gains (profiles/r4h_accumulate_slots_sweep.txt: 64 threads 771 / 745 / 696 MSM/s at 512 / 480 / 448 workgroups' worth of
        // chunks against 761 / 780 / 791 with 256; 512-thread workgroups 776 / 754 / 777)
        static const unsigned acc_block = [] {
            int v = env_int("ZG_MSM_ACC_BLOCK", 256);
            return (unsigned)(v == 64 || v == 128 ? v : 256);
        }();
        if (NT <= (uint32_t)env_int("ZG_MSM_QUAD_ACC_MAX_CHUNKS", 32768))
            hipLaunchKernelGGL(msm_accumulate_chunk_kernel<true>, dim3(div_up((size_t)NT * 4, 256)), dim3(256), 0, st, sv.sorted, sv.starts,
                               sv.nzrank, sv.nzlist, b->d_table, p.NK, NT, ln.d_part);
        else
            hipLaunchKernelGGL(msm_accumulate_chunk_kernel<false>, dim3(div_up(NT, acc_block)), dim3(acc_block), 0, st, sv.sorted, sv.starts,
                               sv.nzrank, sv.nzlist, b->d_table, p.NK, NT, ln.d_part);
        prof_end(ZG_PROF_MSM_ACCUMULATE, st);  // the dominant kernel alone; combine/heavy stages count as reduction
        prof_begin(ZG_PROF_MSM_REDUCE, st);
        hipLaunchKernelGGL(msm_bucket_combine_kernel, dim3(div_up((size_t)p.NK * p.GS * 4, 64)), dim3(64), 0, st, ln.d_part, sv.starts,
                           sv.nzrank, p.NK, NT, p.GS, bucket_out, ln.d_heavy, reinterpret_cast<MsmState *>(sv.state));
        uint32_t nblk_a = (NT + p.NK) / HEAVY_BLOCK_ITEMS + 1;  // stage-A blocks of the huge buckets; at least 256 blocks = 1024 waves for the heavy ones
        hipLaunchKernelGGL(msm_heavy_kernel, dim3(nblk_a < 256 ? 256 : nblk_a), dim3(256), 0, st, ln.d_part, sv.starts, sv.nzrank, sv.nzlist,
                           p.NK, NT, ln.d_heavy, reinterpret_cast<MsmState *>(sv.state), ln.d_part2, bucket_out);
    } else {
        hipLaunchKernelGGL(msm_accumulate_kernel, dim3(div_up((size_t)p.NK * p.S, 256)), dim3(256), 0, st, sv.sorted, sv.starts,
                           b->d_table, p.NK, p.S, bucket_out);
        prof_end(ZG_PROF_MSM_ACCUMULATE, st);
        prof_begin(ZG_PROF_MSM_REDUCE, st);
    }
    return ZG_OK;
    };
    // every slice is sorted before the first one is accumulated: with other MSMs in flight on other streams, the sorts then run
    // under THEIR accumulations (a long stretch with 1/16 of the CUs free) instead of between this MSM's own (measured at 2^22:
    // sort - accumulate - sort - ... left 0.7 ms per MSM outside the accumulations, this order ...)
    auto view = [&](size_t j) {
        SliceView sv{ln.d_sorted, ln.d_starts, ln.d_nzrank, ln.d_nzlist, ln.d_state};
        if (S > 1) {
            uint32_t *m = ln.d_slice_meta + j * meta_stride;
            sv = SliceView{ln.d_sorted + j * (size_t)p.W * per, m, m + ((size_t)p.NK + 1), m + 2 * ((size_t)p.NK + 1),
                           ln.d_state ? (void *)(m + 3 * ((size_t)p.NK + 1)) : nullptr};
        }
        return sv;
    };
    const bool sort_first = S > 1 && env_int("ZG_MSM_SLICE_SORT_FIRST", 0) != 0;
    for (size_t j = 0; j < S && sort_first; j++) {
        const size_t a = j * per, cnt = n_pts - a < per ? n_pts - a : per;
        ZG_TRY(sort_range(off + a, cnt, d_scalars + 4 * a, view(j)));
    }
    for (size_t j = 0; j < S; j++) {
        const size_t a = j * per, cnt = n_pts - a < per ? n_pts - a : per;
        if (j) prof_end(ZG_PROF_MSM_REDUCE, st);  // the bucket sums of a slice count as reduction
        if (!sort_first) ZG_TRY(sort_range(off + a, cnt, d_scalars + 4 * a, view(j)));
        ZG_TRY(accumulate_range(cnt, view(j), j == 0 ? (t_bucket_sink && S == 1 ? t_bucket_sink : ln.d_partial) : ln.d_slice_buckets + (j - 1) * (size_t)p.NK * 144));
    }
    if (t_bucket_sink && S == 1) {  // bucket sums only: the reduction belongs to the caller
        prof_end(ZG_PROF_MSM_REDUCE, st);
        ZG_HIP(hipGetLastError());
        ZG_HIP(hipEventRecord(ln.done, st));
        return ZG_OK;
    }
    if (S > 1)
        hipLaunchKernelGGL(msm_bucket_fold_kernel, dim3(div_up((size_t)p.NK * 4, 256)), dim3(256), 0, st, ln.d_partial, ln.d_slice_buckets, p.NK,
                           (uint32_t)S);
    if (p.lb) {
        char *d_rc = ln.d_bits + 144 * (size_t)p.G * p.K * p.c;  // rows and columns behind the c bit sums of every group
        launch_rowcol(st, ln.d_partial, p.NB, p.lb, p.hb, p.G * p.K, d_rc);
        hipLaunchKernelGGL(msm_bits2d_kernel, dim3(p.c, p.G * p.K), dim3(256), 0, st, ln.d_partial, d_rc, p.NB, p.c, p.lb, p.hb, ln.d_bits);
    } else {
        hipLaunchKernelGGL(msm_bitsum_kernel, dim3(p.PB, p.c, p.G * p.K), dim3(256), 0, st, ln.d_partial, p.NB, p.c, ln.d_bits);
    }
    hipLaunchKernelGGL(msm_final_kernel, dim3(p.G * p.K), dim3(512), 0, st, ln.d_bits, p.c, p.lb ? 1 : p.PB, p.G, ln.d_rg, mode, d_rec, d_inf_out,
                       rec_stride, inf_stride);
    if (p.G > 1)
        hipLaunchKernelGGL(msm_groups_kernel, dim3(p.K), dim3(4), 0, st, ln.d_rg, p.G, p.c, mode, d_rec, d_inf_out, rec_stride, inf_stride);
    prof_end(ZG_PROF_MSM_REDUCE, st);
    ZG_HIP(hipGetLastError());
    ZG_HIP(hipEventRecord(ln.done, st));
    return ZG_OK;
}

// ---- toAffine on the host for the entry points that hand the result to the host anyway. On the device the inversion is a
// single-lane safegcd of ~41 us at the very end of a latency chain (a third of msm_final_kernel); the host does it in a few
// microseconds. The device hands over the un-normalised Jacobian record of mode 2, (X*ZZ, Y*ZZZ, ZZ): x = X'/Z^2, y = Y'/Z^3 —
// canonical Montgomery residues, so the bytes are those of the device's own toAffine (msm/mod.zig:178-189).
namespace hostfp {
typedef unsigned __int128 u128;
static const uint64_t P[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static uint64_t n0inv() {  // -P^-1 mod 2^64
    uint64_t inv = 1;
    for (int i = 0; i < 6; i++) inv *= 2 - P[0] * inv;
    return 0 - inv;
}
static void mul(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {  // Montgomery product, CIOS, result < P
    static const uint64_t N0 = n0inv();
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a[j] * b[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * N0;
        c = ((u128)m * P[0] + t[0]) >> 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * P[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    uint64_t d[4];
    u128 br = 0;
    for (int j = 0; j < 4; j++) {
        u128 v = (u128)t[j] - P[j] - (uint64_t)br;
        d[j] = (uint64_t)v;
        br = (v >> 64) & 1;
    }
    const bool ge = t[4] != 0 || br == 0;
    for (int j = 0; j < 4; j++) r[j] = ge ? d[j] : t[j];
}
static void inv(uint64_t r[4], const uint64_t a[4]) {  // a^(P-2), Montgomery domain in and out; a != 0
    uint64_t e[4] = {P[0] - 2, P[1], P[2], P[3]}, acc[4], base[4];
    bool started = false;
    for (int j = 0; j < 4; j++) base[j] = a[j];
    for (int bit = 253; bit >= 0; bit--) {
        if (started) mul(acc, acc, acc);
        if ((e[bit >> 6] >> (bit & 63)) & 1) {
            if (started) mul(acc, acc, base);
            else {
                for (int j = 0; j < 4; j++) acc[j] = base[j];
                started = true;
            }
        }
    }
    for (int j = 0; j < 4; j++) r[j] = acc[j];
}
// (X, Y, Z) Jacobian record -> affine xy[8] + infinity flag
static void jacobian_to_affine(const uint64_t rec[12], uint64_t out_xy[8], uint8_t *out_inf) {
    const uint64_t *X = rec, *Y = rec + 4, *Z = rec + 8;
    if ((Z[0] | Z[1] | Z[2] | Z[3]) == 0) {
        for (int i = 0; i < 8; i++) out_xy[i] = 0;
        if (out_inf) *out_inf = 1;
        return;
    }
    uint64_t iz[4], iz2[4], iz3[4];
    inv(iz, Z);
    mul(iz2, iz, iz);
    mul(iz3, iz2, iz);
    mul(out_xy, X, iz2);
    mul(out_xy + 4, Y, iz3);
    if (out_inf) *out_inf = 0;
}
}  // namespace hostfp

static int msm_to_host(zg_bases_s *b, size_t off, size_t n, const uint64_t *d_scalars, hipStream_t st, uint64_t out_xy[8],
                       uint8_t *out_inf) {
    if (!env_int("ZG_MSM_HOST_AFFINE", 1)) {  // A/B: toAffine inside msm_final_kernel
        ZG_TRY(msm_enqueue(b, off, n, d_scalars, st, 0, b->d_out, reinterpret_cast<uint8_t *>(b->d_out + 8)));
        ZG_HIP(hipMemcpyAsync(b->h_out, b->d_out, 9 * 8, hipMemcpyDeviceToHost, st));
        ZG_HIP(hipStreamSynchronize(st));
        for (int i = 0; i < 8; i++) out_xy[i] = b->h_out[i];
        if (out_inf) *out_inf = (uint8_t)(b->h_out[8] & 0xff);
        return ZG_OK;
    }
    ZG_TRY(msm_enqueue(b, off, n, d_scalars, st, 2, b->d_out, nullptr));
    ZG_HIP(hipMemcpyAsync(b->h_out, b->d_out, 12 * 8, hipMemcpyDeviceToHost, st));
    ZG_HIP(hipStreamSynchronize(st));
    hostfp::jacobian_to_affine(b->h_out, out_xy, out_inf);
    return ZG_OK;
}

}  // namespace zg

using namespace zg;

extern "C" {

int zg_g1_bases_upload_dev(const uint64_t *d_xy, const uint8_t *d_inf, size_t n, const zg_msm_config *cfg, void *stream,
                           zg_bases_t *out) {
    ZG_INIT();
    if (!out || (n && !d_xy)) {
        set_error("zg_g1_bases_upload_dev: invalid argument");
        return ZG_ERR_INVALID;
    }
    return bases_create(d_xy, d_inf, n, cfg, pick_stream(stream), out);
}

int zg_g1_bases_upload(const uint64_t *xy, const uint8_t *inf, size_t n, const zg_msm_config *cfg, zg_bases_t *out) {
    ZG_INIT();
    if (!out || (n && !xy)) {
        set_error("zg_g1_bases_upload: invalid argument");
        return ZG_ERR_INVALID;
    }
    hipStream_t st = lib_stream();
    Scratch s_xy(n ? n * 64 : 16), s_inf(inf && n ? n : 16);
    if (!s_xy.p || !s_inf.p) return ZG_ERR_NOMEM;
    SyncGuard sync(st);  // the staging copies go back to the pool only after the table build that reads them has finished
    uint64_t *dxy = s_xy.as<uint64_t>();
    uint8_t *dinf = inf ? s_inf.as<uint8_t>() : nullptr;
    if (n) ZG_HIP(hipMemcpyAsync(dxy, xy, n * 64, hipMemcpyHostToDevice, st));
    if (inf && n) ZG_HIP(hipMemcpyAsync(dinf, inf, n, hipMemcpyHostToDevice, st));
    int rc = bases_create(dxy, dinf, n, cfg, st, out);
    hipError_t e = hipStreamSynchronize(st);
    sync.dismiss();
    if (rc != ZG_OK) return rc;
    ZG_HIP(e);
    return ZG_OK;
}

int zg_g1_bases_free(zg_bases_t b) {
    if (!b) return ZG_OK;
    ZG_INIT();
    DeviceGuard dg(b->device);
    (void)hipDeviceSynchronize();
    free_bases(b);
    return ZG_OK;
}

size_t zg_g1_bases_len(zg_bases_t b) { return b ? b->n : 0; }

int zg_g1_bases_plan(zg_bases_t b, int *window_bits, int *windows, int *precompute_levels) {
    if (!b) {
        set_error("zg_g1_bases_plan: invalid argument");
        return ZG_ERR_INVALID;
    }
    if (window_bits) *window_bits = b->plan.c;
    if (windows) *windows = b->plan.W;
    if (precompute_levels) *precompute_levels = b->plan.L;
    return ZG_OK;
}

size_t zg_g1_bases_table_bytes(zg_bases_t b) { return b ? (size_t)(b->plan.L > 1 ? b->plan.L : 1) * b->n * 64 : 0; }

int zg_msm_g1_dev(zg_bases_t b, size_t off, size_t n, const uint64_t *d_scalars, void *stream, uint64_t out_xy[8], uint8_t *out_inf) {
    ZG_INIT();
    if (!b || !out_xy || (n && !d_scalars)) {
        set_error("zg_msm_g1_dev: invalid argument");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(b->device);
    std::lock_guard<std::mutex> lk(b->mu);
    return msm_to_host(b, off, n, d_scalars, pick_stream(stream), out_xy, out_inf);
}

// The host-scalar entry point — what an unmodified MSM.compute / HyperKZG.commit call site reaches — pays a 32*n-byte H2D copy
// (0.7 ms at 2^20 over PCIe Gen5) before anything can run. Long vectors are therefore cut into slices: slice i's copy runs
// while slice i-1's launch set computes (three streams, the handle's workspaces rotate), every slice ends in an un-normalised
// Jacobian partial, and one combine launch adds them up (the group sum does not depend on how the points were grouped, so
// the bytes are those of the unsliced MSM).
static constexpr size_t HOST_SLICE_MIN_POINTS = (size_t)1 << 18;
static size_t host_slice_min() { return (size_t)env_int("ZG_MSM_HOST_SLICE_MIN", (int)HOST_SLICE_MIN_POINTS); }  // tests lower it
static constexpr int HOST_SLICES_MAX = 8;

static int msm_host_sliced(zg_bases_s *b, size_t off, size_t n, const uint64_t *scalars, int slices, uint64_t out_xy[8], uint8_t *out_inf) {
    hipStream_t st = lib_stream();
    ZG_TRY(ensure_aux_streams(b));
    if (!b->d_slice_parts) ZG_HIP(lane_malloc((void **)&b->d_slice_parts, HOST_SLICES_MAX * 12 * 8));
    hipStream_t ss[zg_bases_s::NAUX + 1] = {st, b->aux[0], b->aux[1], b->aux[2]};
    ZG_HIP(hipEventRecord(b->ev_fork, st));
    for (int i = 0; i < zg_bases_s::NAUX; i++) ZG_HIP(hipStreamWaitEvent(b->aux[i], b->ev_fork, 0));
    const size_t per = (n + (size_t)slices - 1) / (size_t)slices;
    int rc = ZG_OK;
    hipError_t e = hipSuccess;
    for (int i = 0; i < slices && rc == ZG_OK && e == hipSuccess; i++) {
        size_t a = (size_t)i * per, cnt = a >= n ? 0 : (n - a < per ? n - a : per);
        hipStream_t si = ss[i % (zg_bases_s::NAUX + 1)];
        if (cnt) e = hipMemcpyAsync(b->d_scal + 4 * a, scalars + 4 * a, cnt * 32, hipMemcpyHostToDevice, si);
        // (an empty trailing slice — more slices than scalars, a test setting — still writes its identity record; its range is [off, off))
        if (e == hipSuccess) rc = msm_enqueue(b, cnt ? off + a : off, cnt, b->d_scal + 4 * (cnt ? a : 0), si, 2, b->d_slice_parts + 12 * (size_t)i, nullptr);
    }
    for (int i = 0; i < zg_bases_s::NAUX; i++) {  // join even after an error so the helpers never run ahead of the caller's next work
        hipError_t e1 = hipEventRecord(b->ev_join[i], b->aux[i]);
        if (e1 == hipSuccess) e1 = hipStreamWaitEvent(st, b->ev_join[i], 0);
        if (e == hipSuccess) e = e1;
    }
    if (rc == ZG_OK && e == hipSuccess) {
        hipLaunchKernelGGL(msm_combine_kernel, dim3(1), dim3(64), 0, st, b->d_slice_parts, (uint32_t)slices, 12u, b->d_out,
                           reinterpret_cast<uint8_t *>(b->d_out + 12), 0u, 0u, 2);  // Jacobian record: toAffine on the host (msm_to_host)
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(b->h_out, b->d_out, 12 * 8, hipMemcpyDeviceToHost, st);
    }
    hipError_t e2 = hipStreamSynchronize(st);
    if (e == hipSuccess) e = e2;
    if (rc != ZG_OK) return rc;
    if (e != hipSuccess) {
        set_error(std::string("zg_msm_g1 (sliced): ") + hipGetErrorString(e));
        return ZG_ERR_HIP;
    }
    hostfp::jacobian_to_affine(b->h_out, out_xy, out_inf);
    return ZG_OK;
}

int zg_msm_g1(zg_bases_t b, size_t off, size_t n, const uint64_t *scalars, uint64_t out_xy[8], uint8_t *out_inf) {
    ZG_INIT();
    if (!b || !out_xy || (n && !scalars)) {
        set_error("zg_msm_g1: invalid argument");
        return ZG_ERR_INVALID;
    }
    if (off + n > b->n) {
        set_error("msm: range exceeds uploaded bases");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(b->device);
    std::lock_guard<std::mutex> lk(b->mu);
    hipStream_t st = lib_stream();
    if (n && !b->d_scal) ZG_HIP(lane_malloc((void **)&b->d_scal, b->n * 32));
    int slices = env_int("ZG_MSM_HOST_SLICES", 3);  // 2^20 scalars: 2.32 ms unsliced, 2.20 / 2.15 / 2.20 / 2.30 / 2.45 with 2 / 3 / 4 / 6 / 8 slices
    if (slices > HOST_SLICES_MAX) slices = HOST_SLICES_MAX;
    if (slices >= 2 && n >= host_slice_min() && b->lanes.size() >= 2) return msm_host_sliced(b, off, n, scalars, slices, out_xy, out_inf);
    if (n) ZG_HIP(hipMemcpyAsync(b->d_scal, scalars, n * 32, hipMemcpyHostToDevice, st));
    return msm_to_host(b, off, n, b->d_scal, st, out_xy, out_inf);
}

// Scalars that are F.fromU64 of machine words — every polynomial `zolt prove` commits to (commitBytecode / commitMemory / commitRegisters,
// src/zkvm/mod.zig:1518-1617: program bytes, memory values, rd_value per cycle): 8 bytes per scalar cross PCIe instead of 32, the
// Montgomery conversion runs on the device (the reference spends a field multiplication per evaluation on the CPU for it), and since such
// a scalar has no digits above bit 64 the accumulation walks 4 of the 15 windows. Same bytes out as zg_msm_g1 on the converted vector.
int zg_msm_g1_u64(zg_bases_t b, size_t off, size_t n, const uint64_t *values, uint64_t out_xy[8], uint8_t *out_inf) {
    ZG_INIT();
    if (!b || !out_xy || (n && !values)) {
        set_error("zg_msm_g1_u64: invalid argument");
        return ZG_ERR_INVALID;
    }
    if (off + n > b->n) {
        set_error("msm: range exceeds uploaded bases");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(b->device);
    std::lock_guard<std::mutex> lk(b->mu);
    hipStream_t st = lib_stream();
    if (n && !b->d_scal) ZG_HIP(lane_malloc((void **)&b->d_scal, b->n * 32));
    Scratch s_vals(n ? n * 8 : 16);
    if (!s_vals.p) return ZG_ERR_NOMEM;
    SyncGuard sync(st);
    if (n) {
        ZG_HIP(hipMemcpyAsync(s_vals.p, values, n * 8, hipMemcpyHostToDevice, st));
        ZG_TRY(ingest_u64_to_fr(s_vals.as<uint64_t>(), n, b->d_scal, st));
    }
    int rc = msm_to_host(b, off, n, b->d_scal, st, out_xy, out_inf);  // synchronises st before it returns
    if (rc == ZG_OK) sync.dismiss();
    return rc;
}

int zg_msm_g1_dev_async(zg_bases_t b, size_t off, size_t n, const uint64_t *d_scalars, void *stream, uint64_t *d_out_xy,
                        uint8_t *d_out_inf) {
    ZG_INIT();
    if (!b || !d_out_xy || !d_out_inf || (n && !d_scalars)) {
        set_error("zg_msm_g1_dev_async: invalid argument");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(b->device);
    std::lock_guard<std::mutex> lk(b->mu);
    return msm_enqueue(b, off, n, d_scalars, pick_stream(stream), 0, d_out_xy, d_out_inf);
}

int zg_msm_g1_partial_dev(zg_bases_t b, size_t off, size_t n, const uint64_t *d_scalars, void *stream, uint64_t *d_out_jac) {
    ZG_INIT();
    if (!b || !d_out_jac || (n && !d_scalars)) {
        set_error("zg_msm_g1_partial_dev: invalid argument");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(b->device);
    std::lock_guard<std::mutex> lk(b->mu);
    return msm_enqueue(b, off, n, d_scalars, pick_stream(stream), 1, d_out_jac, nullptr);
}

int zg_msm_g1_partial_fast_dev(zg_bases_t b, size_t off, size_t n, const uint64_t *d_scalars, void *stream, uint64_t *d_out_jac) {
    ZG_INIT();
    if (!b || !d_out_jac || (n && !d_scalars)) {
        set_error("zg_msm_g1_partial_fast_dev: invalid argument");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(b->device);
    std::lock_guard<std::mutex> lk(b->mu);
    return msm_enqueue(b, off, n, d_scalars, pick_stream(stream), 2, d_out_jac, nullptr);
}

// Largest number of scalar vectors one fused launch set can take for this handle and vector length (0: do not fuse).
// Fusing pays when the MSMs are short (a lone short MSM is pure launch/dependency latency, ~0.4 ms whatever its size):
// the k vectors become k times the bucket groups of ONE sort / accumulate / reduce pass. It needs the LDS counting sort
// (all bucket counters of the launch in 128 KiB), i.e. handles with a small window.
// How many scalar vectors of n scalars one launch set may hold. Narrow windows: all bucket counters of the set must fit the
// single-pass sort's LDS histogram. wide_ok (HyperKZG.open's long levels, zero-padded rows): wide-window handles too, as many
// vectors as the two-pass sort has coarse bins for (plan_two_pass: <= 3000 bins of 2^7 buckets when the table rows fit 24 bits).
static size_t batch_fuse_limit(const zg_bases_s *b, size_t n, bool wide_ok = false) {
    const MsmPlan &p = b->plan;
    if (n == 0 || !env_int("ZG_MSM_BATCH_FUSE", 1)) return 0;
    size_t by_lds = (128 * 1024 / 4) / ((size_t)p.NB * p.G);
    if (by_lds < 2 && wide_ok) {
        int need = 1;
        while (((size_t)1 << need) < (size_t)p.L * b->n) need++;
        int fb = 31 - need > 7 ? 7 : 31 - need;
        if (fb >= 5) by_lds = ((size_t)3000 << fb) / ((size_t)p.NB * p.G);
    }
    size_t by_size = ((size_t)1 << 22) / n;  // keep a launch set at or below 2^22 scalars
    size_t lim = by_lds < by_size ? by_lds : by_size;
    return lim >= 2 ? lim : 0;
}

// Rows of a batch on a handle WITHOUT a table (G > 1: one bucket set per window) cannot share a sort (K * G * NB buckets leave the coarse
// counters' LDS), and as launch sets of their own each ends in its own reduction: row / column sums (0.24 ms), bit sums, msm_final and the
// window combine msm_groups_kernel — (G - 1) * c dependent doublings on one quad of lanes, 0.85 ms — all latency chains. HyperKZG.open's five
// long levels paid that five times, one after the other (9.1 ms per opening of 2^20 evaluations, 4.3 ms of it in msm_groups_kernel). Here
// every row is sorted and accumulated as a set of its own, its bucket sums land side by side in ONE array, and the reduction kernels run
// once over K * G groups (their grids already take a batch): the chains of all rows run beside each other.
static constexpr size_t ROWS_SHARED_TAIL_MAX = 16;
static bool rows_shared_tail_ok(const zg_bases_s *b, size_t n) {
    const MsmPlan &p = b->plan;
    if (p.G <= 1 || !p.lb || !env_int("ZG_MSM_ROWS_SHARED_TAIL", 1)) return false;
    size_t S, per;
    slice_counts(p, n, S, per);
    return S == 1;
}
static int msm_rows_shared_tail(zg_bases_s *b, size_t n, const uint64_t *d_scalars, size_t k, hipStream_t st, uint64_t *d_out9) {
    const MsmPlan &p = b->plan;
    const size_t row_bytes = (size_t)p.NK * 144;  // p.K == 1: NK = G * NB
    size_t per_group = (size_t)p.c * p.PB;
    if (((size_t)1 << p.lb) + ((size_t)1 << p.hb) + p.c > per_group) per_group = ((size_t)1 << p.lb) + ((size_t)1 << p.hb) + p.c;
    if (!b->rows_done) ZG_HIP(hipEventCreateWithFlags(&b->rows_done, hipEventDisableTiming));
    if (b->rows_cap < k) {
        if (b->rows_cap) ZG_HIP(hipEventSynchronize(b->rows_done));
        for (char **q : {&b->d_rows_buckets, &b->d_rows_bits, &b->d_rows_rg}) {
            pool_free(*q);
            *q = nullptr;
        }
        b->rows_cap = 0;
        ZG_HIP(lane_malloc((void **)&b->d_rows_buckets, k * row_bytes));
        ZG_HIP(lane_malloc((void **)&b->d_rows_bits, (size_t)p.G * k * per_group * 144));
        ZG_HIP(lane_malloc((void **)&b->d_rows_rg, (size_t)p.G * k * 128));
        b->rows_cap = k;
    } else {
        ZG_HIP(hipStreamWaitEvent(st, b->rows_done, 0));  // the previous batch may have run on another stream
    }
    int rc = ZG_OK;
    for (size_t i = 0; i < k && rc == ZG_OK; i++) {
        t_bucket_sink = b->d_rows_buckets + i * row_bytes;
        rc = msm_enqueue(b, 0, n, d_scalars + 4 * n * i, st, 0, d_out9 + 9 * i, reinterpret_cast<uint8_t *>(d_out9 + 9 * i + 8));
        t_bucket_sink = nullptr;
    }
    if (rc != ZG_OK) return rc;
    const int GK = p.G * (int)k;
    char *d_rc = b->d_rows_bits + 144 * (size_t)GK * p.c;  // rows and columns behind the c bit sums of every group (as in msm_enqueue_lane)
    prof_begin(ZG_PROF_MSM_REDUCE, st);
    launch_rowcol(st, b->d_rows_buckets, p.NB, p.lb, p.hb, GK, d_rc);
    hipLaunchKernelGGL(msm_bits2d_kernel, dim3(p.c, GK), dim3(256), 0, st, b->d_rows_buckets, d_rc, p.NB, p.c, p.lb, p.hb, b->d_rows_bits);
    hipLaunchKernelGGL(msm_final_kernel, dim3(GK), dim3(512), 0, st, b->d_rows_bits, p.c, 1, p.G, b->d_rows_rg, 0, d_out9, reinterpret_cast<uint8_t *>(d_out9 + 8), 9u, 72u);
    hipLaunchKernelGGL(msm_groups_kernel, dim3((unsigned)k), dim3(4), 0, st, b->d_rows_rg, p.G, p.c, 0, d_out9, reinterpret_cast<uint8_t *>(d_out9 + 8), 9u, 72u);
    prof_end(ZG_PROF_MSM_REDUCE, st);
    ZG_HIP(hipGetLastError());
    ZG_HIP(hipEventRecord(b->rows_done, st));
    return ZG_OK;
}

// the handle's two helper streams (+ fork / join events): independent launch sets rotate over the caller's stream and these

// enqueue k scalar vectors (device, back to back) over bases[0, n) on st. mode 0: record i = d_out9[9*i .. 9*i+8] (xy[8], flag
// word); mode 1 / 2: record i = 12 limbs at d_out9 + 12*i (Jacobian partial, normalised / any representative — see write_result)
static int msm_batch_enqueue(zg_bases_s *b, size_t n, const uint64_t *d_scalars, size_t k, hipStream_t st, uint64_t *d_out9,
                             bool wide_ok = false, int mode = 0) {
    const uint32_t RS = mode == 0 ? 9u : 12u;  // record stride in u64
    auto inf_of = [&](size_t i) { return mode == 0 ? reinterpret_cast<uint8_t *>(d_out9 + RS * i + 8) : (uint8_t *)nullptr; };
    if (n > b->n) {
        set_error("msm: range exceeds uploaded bases");
        return ZG_ERR_INVALID;
    }
    if (b->small && n <= b->small->n) return msm_batch_enqueue(b->small, n, d_scalars, k, st, d_out9, false, mode);  // narrow-window side table
    size_t lim = batch_fuse_limit(b, n, wide_ok);
    if (lim >= 2 && k >= 2) {
        // batch_fuse_limit prices a wide set by the two-pass sort's coarse bins; whether that sort really applies is plan_two_pass's
        // decision (ZG_MSM_TWO_PASS_SORT, ZG_MSM_FINE_BITS*, the set's size). A set that would sort in ONE pass needs all its K * G * NB
        // counters in 128 KiB of LDS: if it has neither, it is not fused (round 6: with ZG_MSM_TWO_PASS_SORT=0 HyperKZG.open's long
        // levels ran the LDS scatter over 160 k counters — a memory fault; found by running the suite under the alternate switches)
        zg_msm_config cfg_t{b->plan.c, b->plan.L, 0};
        MsmPlan trial;
        const size_t kc_t = k < lim ? k : lim;
        if (make_plan(n, &cfg_t, trial, kc_t) != ZG_OK) {
            lim = 0;
        } else {
            plan_two_pass(trial, (size_t)b->plan.L * b->n, n * kc_t);
            if (!trial.fb && (size_t)trial.NK * 4 > 128 * 1024) lim = 0;
        }
    }
    if (lim == 0 && wide_ok && k >= 2 && k <= ROWS_SHARED_TAIL_MAX && mode == 0 && rows_shared_tail_ok(b, n)) return msm_rows_shared_tail(b, n, d_scalars, k, st, d_out9);
    if (lim == 0 || k < 2) {
        // one launch set per vector, rotating through the handle's workspaces AND through three streams (the caller's
        // plus two forked helpers), so the latency-bound tail of one MSM runs under the accumulation of the next
        bool fork = k >= 2 && n > 0 && b->lanes.size() >= 2;
        if (fork) ZG_TRY(ensure_aux_streams(b));
        if (fork) {
            ZG_HIP(hipEventRecord(b->ev_fork, st));
            for (int i = 0; i < zg_bases_s::NAUX; i++) ZG_HIP(hipStreamWaitEvent(b->aux[i], b->ev_fork, 0));
        }
        int rc = ZG_OK;
        for (size_t i = 0; i < k && rc == ZG_OK; i++) {
            hipStream_t si = !fork || i % (zg_bases_s::NAUX + 1) == 0 ? st : b->aux[i % (zg_bases_s::NAUX + 1) - 1];
            rc = msm_enqueue(b, 0, n, d_scalars + 4 * n * i, si, mode, d_out9 + RS * i, inf_of(i));
        }
        if (fork) {  // join even after an error so the helpers never run ahead of the caller's next work
            for (int i = 0; i < zg_bases_s::NAUX; i++) {
                ZG_HIP(hipEventRecord(b->ev_join[i], b->aux[i]));
                ZG_HIP(hipStreamWaitEvent(st, b->ev_join[i], 0));
            }
        }
        return rc;
    }
    zg_msm_config cfg{b->plan.c, b->plan.L, 0};
    size_t kc = k < lim ? k : lim;
    if (b->batch_n != n || (size_t)b->batch_plan.K < kc) {  // (re)build the fused workspace for kc vectors of n scalars
        if (b->batch_lane.done) {
            (void)hipEventSynchronize(b->batch_lane.done);
            lane_free(b->batch_lane);
        }
        b->batch_n = 0;
        ZG_TRY(make_plan(n, &cfg, b->batch_plan, kc));
        plan_two_pass(b->batch_plan, (size_t)b->plan.L * b->n, n * kc);
        if (b->batch_plan.fb) {
            b->batch_nblk = (uint32_t)div_up(n * kc, two_pass_span(b->batch_plan.W));
        } else {
            uint32_t nblk = (uint32_t)(n * kc / (size_t)sort_span(b->batch_plan.NK));
            b->batch_nblk = nblk < 1 ? 1 : (nblk > 256 ? 256 : nblk);
        }
        hipError_t e = lane_alloc(b->batch_lane, b->batch_plan, n * kc, b->batch_nblk);
        if (e != hipSuccess) {
            lane_free(b->batch_lane);
            set_error(std::string("msm batch workspace: ") + hipGetErrorString(e));
            return e == hipErrorOutOfMemory ? ZG_ERR_NOMEM : ZG_ERR_HIP;
        }
        b->batch_n = n;
    }
    for (size_t i0 = 0; i0 < k; i0 += kc) {
        size_t kk = k - i0 < kc ? k - i0 : kc;
        MsmPlan pl = b->batch_plan;
        if (kk != (size_t)pl.K) {  // a shorter last set fits the same workspace (and keeps its sort mode)
            ZG_TRY(make_plan(n, &cfg, pl, kk));
            pl.fb = b->batch_plan.fb;
            pl.rb = b->batch_plan.rb;
            pl.NCB = pl.fb ? (pl.NK + (1u << pl.fb) - 1) >> pl.fb : 0;
        }
        ZG_TRY(msm_enqueue_lane(b, pl, b->batch_lane, b->batch_nblk, 0, n, d_scalars + 4 * n * i0, st, mode, d_out9 + RS * i0, inf_of(i0), RS,
                                mode == 0 ? 72 : 0));
    }
    return ZG_OK;
}

}  // extern "C"

namespace zg {
// zg_msm_g1_batch_dev for zero-padded rows of different live lengths (HyperKZG.open's long levels): also fuses on wide-window
// handles. Not exported: a general batch of full-length vectors on such a handle is better served by the stream rotation.
int msm_batch_dev_wide(zg_bases_t b, size_t n, const uint64_t *d_scalars, size_t k, hipStream_t st, uint64_t *d_out9, const size_t *row_len) {
    DeviceGuard dg(b->device);
    std::lock_guard<std::mutex> lk(b->mu);
    RowOffs rows{};
    if (row_len && k >= 2 && k <= 32) {  // row j is zero beyond row_len[j] <= n: the sort walks the live entries only
        rows.k = (uint32_t)k;
        for (size_t j = 0; j < k; j++) rows.off[j + 1] = rows.off[j] + (uint32_t)(row_len[j] < n ? row_len[j] : n);
        t_row_offs = &rows;
    }
    int rc = msm_batch_enqueue(b, n, d_scalars, k, st, d_out9, true);
    t_row_offs = nullptr;
    return rc;
}

// sharded.hip: this device's k partial sums of a sharded batch (BatchMSM over one shard of the bases), as un-normalised
// Jacobian records back to back (12 limbs each) — the batch form of zg_msm_g1_partial_fast_dev
int msm_batch_partials_dev(zg_bases_t b, size_t n, const uint64_t *d_scalars, size_t k, hipStream_t st, uint64_t *d_out12) {
    DeviceGuard dg(b->device);
    std::lock_guard<std::mutex> lk(b->mu);
    return msm_batch_enqueue(b, n, d_scalars, k, st, d_out12, false, 2);
}

// sharded.hip: combine `ranks` gathered partial records per scalar vector (rank r's k records back to back at
// d_partials + r * rank_stride) into k affine result records of 9 words (xy[8], flag)
int msm_combine_batch_enqueue(const uint64_t *d_partials, size_t ranks, size_t rank_stride, size_t k, hipStream_t st, uint64_t *d_out9) {
    if (k == 0) return ZG_OK;
    hipLaunchKernelGGL(msm_combine_kernel, dim3((unsigned)k), dim3(64), 0, st, d_partials, (uint32_t)ranks, (uint32_t)rank_stride, d_out9,
                       reinterpret_cast<uint8_t *>(d_out9 + 8), 9u, 72u);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

int bases_device(zg_bases_t b) { return b ? b->device : -1; }
}  // namespace zg

extern "C" {

int zg_msm_g1_batch_dev(zg_bases_t b, size_t n, const uint64_t *d_scalars, size_t k, void *stream, uint64_t *d_out9) {
    ZG_INIT();
    if (!b || (k && (!d_out9 || (n && !d_scalars)))) {
        set_error("zg_msm_g1_batch_dev: invalid argument");
        return ZG_ERR_INVALID;
    }
    DeviceGuard dg(b->device);
    std::lock_guard<std::mutex> lk(b->mu);
    return msm_batch_enqueue(b, n, d_scalars, k, pick_stream(stream), d_out9);
}

int zg_msm_g1_batch(zg_bases_t b, size_t n, const uint64_t *const *batches, size_t k, uint64_t *out_xy, uint8_t *out_inf) {
    ZG_INIT();
    if (!b || (k && (!batches || !out_xy))) {
        set_error("zg_msm_g1_batch: invalid argument");
        return ZG_ERR_INVALID;
    }
    if (n > b->n) {
        set_error("msm: range exceeds uploaded bases");
        return ZG_ERR_INVALID;
    }
    if (k == 0) return ZG_OK;
    DeviceGuard dg(b->device);
    std::lock_guard<std::mutex> lk(b->mu);
    hipStream_t st = lib_stream();
    // the k vectors are staged back to back on the device; all results stay there until one final copy
    Scratch s_sc((n ? n : 1) * 32 * k), s_res(9 * 8 * k);
    if (!s_sc.p || !s_res.p) return ZG_ERR_NOMEM;
    uint64_t *d_sc = s_sc.as<uint64_t>(), *d_res = s_res.as<uint64_t>();
    ZG_HIP(hipMemsetAsync(d_res, 0, 9 * 8 * k, st));
    int rc = ZG_OK;
    const bool routed_small = b->small && n <= b->small->n;
    if (!routed_small && k >= 2 && n >= host_slice_min() && batch_fuse_limit(b, n) == 0 && b->lanes.size() >= 2) {
        // long vectors (HyperKZG.batchCommit of full-size polynomials): vector i's copy and launch set go on stream i mod 3, so the
        // 32n-byte copy of the next vector runs under the MSM of the previous one instead of all k copies preceding all k MSMs
        rc = ensure_aux_streams(b);
        if (rc == ZG_OK) {
            hipStream_t ss[zg_bases_s::NAUX + 1] = {st, b->aux[0], b->aux[1], b->aux[2]};
            hipError_t e = hipEventRecord(b->ev_fork, st);
            for (int i = 0; i < zg_bases_s::NAUX && e == hipSuccess; i++) e = hipStreamWaitEvent(b->aux[i], b->ev_fork, 0);
            for (size_t i = 0; i < k && rc == ZG_OK && e == hipSuccess; i++) {
                hipStream_t si = ss[i % (zg_bases_s::NAUX + 1)];
                e = hipMemcpyAsync(d_sc + 4 * n * i, batches[i], n * 32, hipMemcpyHostToDevice, si);
                if (e == hipSuccess) rc = msm_enqueue(b, 0, n, d_sc + 4 * n * i, si, 0, d_res + 9 * i, reinterpret_cast<uint8_t *>(d_res + 9 * i + 8));
            }
            for (int i = 0; i < zg_bases_s::NAUX; i++) {  // join even after an error
                hipError_t e1 = hipEventRecord(b->ev_join[i], b->aux[i]);
                if (e1 == hipSuccess) e1 = hipStreamWaitEvent(st, b->ev_join[i], 0);
                if (e == hipSuccess) e = e1;
            }
            if (e != hipSuccess && rc == ZG_OK) {
                set_error(std::string("zg_msm_g1_batch: ") + hipGetErrorString(e));
                rc = ZG_ERR_HIP;
            }
        }
    } else {
        hipError_t e = hipSuccess;
        for (size_t i = 0; i < k && n && e == hipSuccess; i++) e = hipMemcpyAsync(d_sc + 4 * n * i, batches[i], n * 32, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) {
            set_error(std::string("zg_msm_g1_batch: ") + hipGetErrorString(e));
            rc = ZG_ERR_HIP;
        } else {
            rc = msm_batch_enqueue(b, n, d_sc, k, st, d_res);
        }
    }
    std::vector<uint64_t> h_res(9 * k);
    hipError_t e = hipSuccess;
    if (rc == ZG_OK) e = hipMemcpyAsync(h_res.data(), d_res, 9 * 8 * k, hipMemcpyDeviceToHost, st);
    hipError_t e2 = hipStreamSynchronize(st);
    if (e == hipSuccess) e = e2;
    if (rc != ZG_OK) return rc;
    if (e != hipSuccess) {
        set_error(std::string("zg_msm_g1_batch: ") + hipGetErrorString(e));
        return ZG_ERR_HIP;
    }
    for (size_t i = 0; i < k; i++) {
        for (int j = 0; j < 8; j++) out_xy[8 * i + j] = h_res[9 * i + j];
        if (out_inf) out_inf[i] = (uint8_t)(h_res[9 * i + 8] & 0xff);
    }
    return ZG_OK;
}

int zg_g1_combine_partials_dev(const uint64_t *d_partials, size_t k, void *stream, uint64_t out_xy[8], uint8_t *out_inf) {
    ZG_INIT();
    if (!out_xy || (k && !d_partials)) {
        set_error("zg_g1_combine_partials_dev: invalid argument");
        return ZG_ERR_INVALID;
    }
    hipStream_t st = pick_stream(stream);
    Scratch s_out(16 * 8);
    if (!s_out.p) return ZG_ERR_NOMEM;
    SyncGuard sync(st);
    uint64_t *d_out = s_out.as<uint64_t>();
    hipLaunchKernelGGL(msm_combine_kernel, dim3(1), dim3(64), 0, st, d_partials, (uint32_t)k, 12u, d_out, reinterpret_cast<uint8_t *>(d_out + 8), 0u, 0u);
    uint64_t h[9];
    ZG_HIP(hipMemcpyAsync(h, d_out, 9 * 8, hipMemcpyDeviceToHost, st));
    ZG_HIP(hipStreamSynchronize(st));
    sync.dismiss();
    for (int i = 0; i < 8; i++) out_xy[i] = h[i];
    if (out_inf) *out_inf = (uint8_t)(h[8] & 0xff);
    return ZG_OK;
}

int zg_g1_combine_partials_dev_async(const uint64_t *d_partials, size_t k, void *stream, uint64_t *d_out_xy, uint8_t *d_out_inf) {
    ZG_INIT();
    if (!d_out_xy || !d_out_inf || (k && !d_partials)) {
        set_error("zg_g1_combine_partials_dev_async: invalid argument");
        return ZG_ERR_INVALID;
    }
    hipLaunchKernelGGL(msm_combine_kernel, dim3(1), dim3(64), 0, pick_stream(stream), d_partials, (uint32_t)k, 12u, d_out_xy, d_out_inf, 0u, 0u);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

int zg_g1_combine_partials_batch_dev_async(const uint64_t *d_partials, size_t ranks, size_t rank_stride, size_t m, void *stream, uint64_t *d_out9) {
    ZG_INIT();
    if ((m && (!d_partials || !d_out9)) || ranks == 0 || rank_stride < 12 * m || ranks > 0xffffffffu || rank_stride > 0xffffffffu) {
        set_error("zg_g1_combine_partials_batch_dev_async: invalid argument");
        return ZG_ERR_INVALID;
    }
    return msm_combine_batch_enqueue(d_partials, ranks, rank_stride, m, pick_stream(stream), d_out9);
}

int zg_g1_is_on_curve_batch(const uint64_t *xy, const uint8_t *inf, size_t n, uint8_t *out) {
    ZG_INIT();
    if (n && (!xy || !out)) {
        set_error("zg_g1_is_on_curve_batch: invalid argument");
        return ZG_ERR_INVALID;
    }
    if (n == 0) return ZG_OK;
    hipStream_t st = lib_stream();
    Scratch s_xy(n * 64), s_f(2 * n);
    if (!s_xy.p || !s_f.p) return ZG_ERR_NOMEM;
    SyncGuard sync(st);
    uint64_t *dxy = s_xy.as<uint64_t>();
    uint8_t *dout = s_f.as<uint8_t>(), *dinf = nullptr;
    ZG_HIP(hipMemcpyAsync(dxy, xy, n * 64, hipMemcpyHostToDevice, st));
    if (inf) {
        dinf = dout + n;
        ZG_HIP(hipMemcpyAsync(dinf, inf, n, hipMemcpyHostToDevice, st));
    }
    hipLaunchKernelGGL(g1_on_curve_kernel, dim3(div_up(n, 256)), dim3(256), 0, st, dxy, dinf, n, dout);
    ZG_HIP(hipGetLastError());
    ZG_HIP(hipMemcpyAsync(out, dout, n, hipMemcpyDeviceToHost, st));
    ZG_HIP(hipStreamSynchronize(st));
    sync.dismiss();
    return ZG_OK;
}

int zg_g1_affine_add_batch(const uint64_t *a_xy, const uint8_t *a_inf, const uint64_t *b_xy, const uint8_t *b_inf, size_t n,
                           uint64_t *out_xy, uint8_t *out_inf) {
    ZG_INIT();
    if (n && (!a_xy || !b_xy || !out_xy)) {
        set_error("zg_g1_affine_add_batch: invalid argument");
        return ZG_ERR_INVALID;
    }
    if (n == 0) return ZG_OK;
    hipStream_t st = lib_stream();
    Scratch s_a(n * 64), s_b(n * 64), s_o(n * 64), s_f(3 * n);
    if (!s_a.p || !s_b.p || !s_o.p || !s_f.p) return ZG_ERR_NOMEM;
    SyncGuard sync(st);
    uint8_t *d_ai = s_f.as<uint8_t>(), *d_bi = d_ai + n, *d_oi = d_bi + n;
    ZG_HIP(hipMemcpyAsync(s_a.p, a_xy, n * 64, hipMemcpyHostToDevice, st));
    ZG_HIP(hipMemcpyAsync(s_b.p, b_xy, n * 64, hipMemcpyHostToDevice, st));
    if (a_inf) ZG_HIP(hipMemcpyAsync(d_ai, a_inf, n, hipMemcpyHostToDevice, st));
    if (b_inf) ZG_HIP(hipMemcpyAsync(d_bi, b_inf, n, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(g1_affine_add_kernel, dim3(div_up(n, 256)), dim3(256), 0, st, s_a.as<uint64_t>(), a_inf ? d_ai : nullptr,
                       s_b.as<uint64_t>(), b_inf ? d_bi : nullptr, n, s_o.as<uint64_t>(), d_oi);
    ZG_HIP(hipGetLastError());
    ZG_HIP(hipMemcpyAsync(out_xy, s_o.p, n * 64, hipMemcpyDeviceToHost, st));
    if (out_inf) ZG_HIP(hipMemcpyAsync(out_inf, d_oi, n, hipMemcpyDeviceToHost, st));
    ZG_HIP(hipStreamSynchronize(st));
    sync.dismiss();
    return ZG_OK;
}

static FbPlan fb_plan(size_t n) {
    int c = n <= ((size_t)1 << 15) ? 8 : (n <= ((size_t)1 << 18) ? 10 : 11);
    int e = env_int("ZG_FB_WINDOW_BITS", 0);
    if (e >= 4 && e <= 14) c = e;
    return FbPlan{c, (254 + c - 1) / c, (1u << c) - 1u};
}

int zg_g1_fixed_base_mul_batch(const uint64_t base_xy[8], uint8_t base_inf, const uint64_t *scalars, size_t n, uint64_t *out_xy,
                               uint8_t *out_inf) {
    ZG_INIT();
    if (!base_xy || (n && (!scalars || !out_xy || !out_inf))) {
        set_error("zg_g1_fixed_base_mul_batch: invalid argument");
        return ZG_ERR_INVALID;
    }
    if (n == 0) return ZG_OK;
    if (base_inf) {  // k * infinity = infinity (src/msm/mod.zig:504-506)
        memset(out_xy, 0, n * 64);
        memset(out_inf, 1, n);
        return ZG_OK;
    }
    hipStream_t st = lib_stream();
    const FbPlan fb = fb_plan(n);
    const uint32_t n_rows = (uint32_t)fb.W * fb.rows;
    Scratch s_base(64), s_rows((size_t)fb.W * 144), s_tab((size_t)n_rows * 64), s_sc(n * 32), s_out(n * 64), s_inf(n);
    if (!s_base.p || !s_rows.p || !s_tab.p || !s_sc.p || !s_out.p || !s_inf.p) return ZG_ERR_NOMEM;
    SyncGuard sync(st);
    ZG_HIP(hipMemcpyAsync(s_base.p, base_xy, 64, hipMemcpyHostToDevice, st));
    ZG_HIP(hipMemcpyAsync(s_sc.p, scalars, n * 32, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(fb_window_bases_kernel, dim3(1), dim3(4 * fb.W), 0, st, s_base.as<uint64_t>(), fb.c, s_rows.as<char>());
    hipLaunchKernelGGL(fb_table_rows_kernel, dim3(div_up(n_rows, 256)), dim3(256), 0, st, s_rows.as<char>(), n_rows, fb.c, fb.rows, s_tab.as<char>());
    hipLaunchKernelGGL(fb_mul_kernel, dim3(div_up(n, 256)), dim3(256), 0, st, s_tab.as<char>(), s_sc.as<uint64_t>(), n, fb.c, fb.W, fb.rows,
                       s_out.as<uint64_t>(), s_inf.as<uint8_t>());
    ZG_HIP(hipGetLastError());
    ZG_HIP(hipMemcpyAsync(out_xy, s_out.p, n * 64, hipMemcpyDeviceToHost, st));
    ZG_HIP(hipMemcpyAsync(out_inf, s_inf.p, n, hipMemcpyDeviceToHost, st));
    ZG_HIP(hipStreamSynchronize(st));
    sync.dismiss();
    return ZG_OK;
}

// HyperKZG.setup's G1 side on the device (generateMockSRS, src/poly/commitment/mod.zig:174-213: powers[i] = scalarMul(g1, tau^i)): the
// powers of tau, the fixed-base batch and the handle's table of multiples are built in HBM and stay there — the points cross PCIe only if
// the caller asks for them (out_xy). The compiled mirror spent 131 ms of a 2^20-cycle proof here (131 of 210 ms, tools/bench_prove_path):
// 2^20 host field products for the powers, 64 MB down, struct conversions, 64 MB up again.
int zg_hyperkzg_setup(const uint64_t base_xy[8], const uint64_t tau[4], size_t n, const zg_msm_config *cfg, uint64_t *out_xy, uint8_t *out_inf,
                      zg_bases_t *out) {
    ZG_INIT();
    if (!base_xy || !tau || !out || n >= ((size_t)1 << 27)) {
        set_error("zg_hyperkzg_setup: invalid argument (fewer than 2^27 powers, the most a handle holds)");
        return ZG_ERR_INVALID;
    }
    hipStream_t st = lib_stream();
    const FbPlan fb = fb_plan(n);
    const uint32_t n_rows = (uint32_t)fb.W * fb.rows;
    const size_t nn = n ? n : 1;
    Scratch s_base(64), s_rows((size_t)fb.W * 144), s_tab((size_t)n_rows * 64), s_pw(4 * 256 * 32), s_sc(nn * 32), s_out(nn * 64), s_inf(nn);
    if (!s_base.p || !s_rows.p || !s_tab.p || !s_pw.p || !s_sc.p || !s_out.p || !s_inf.p) return ZG_ERR_NOMEM;
    SyncGuard sync(st);
    if (n) {
        TauArg ta;
        for (int i = 0; i < 4; i++) {
            ta.l[2 * i] = (uint32_t)tau[i];
            ta.l[2 * i + 1] = (uint32_t)(tau[i] >> 32);
        }
        ZG_HIP(hipMemcpyAsync(s_base.p, base_xy, 64, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(tau_tables_kernel, dim3(1), dim3(64), 0, st, ta, s_pw.as<uint64_t>());
        hipLaunchKernelGGL(tau_powers_kernel, dim3(div_up(n, 256)), dim3(256), 0, st, s_pw.as<uint64_t>(), n, s_sc.as<uint64_t>());
            hipLaunchKernelGGL(fb_window_bases_kernel, dim3(1), dim3(4 * fb.W), 0, st, s_base.as<uint64_t>(), fb.c, s_rows.as<char>());
        hipLaunchKernelGGL(fb_table_rows_kernel, dim3(div_up(n_rows, 256)), dim3(256), 0, st, s_rows.as<char>(), n_rows, fb.c, fb.rows, s_tab.as<char>());
        hipLaunchKernelGGL(fb_mul_kernel, dim3(div_up(n, 256)), dim3(256), 0, st, s_tab.as<char>(), s_sc.as<uint64_t>(), n, fb.c, fb.W, fb.rows,
                           s_out.as<uint64_t>(), s_inf.as<uint8_t>());
        ZG_HIP(hipGetLastError());
            if (out_xy) ZG_HIP(hipMemcpyAsync(out_xy, s_out.p, n * 64, hipMemcpyDeviceToHost, st));
        if (out_inf) ZG_HIP(hipMemcpyAsync(out_inf, s_inf.p, n, hipMemcpyDeviceToHost, st));
    }
    // tau != 0: tau^i is never 0 mod r and the base has prime order, no power is the identity and the handle carries no infinity flags.
    // tau == 0 (canonical zero; the ABI accepts any field element): powers[i] = scalarMul(g1, 0) = identity for every i >= 1, as in the
    // reference — the flags fb_mul_kernel wrote travel into the handle, or its MSMs would take (0, 0) for a point (round-5 advisor)
    const bool tau_zero = !(tau[0] | tau[1] | tau[2] | tau[3]);
    int rc = bases_create(s_out.as<uint64_t>(), tau_zero && n ? s_inf.as<uint8_t>() : nullptr, n, cfg, st, out);  // copies the points into the handle's table (and builds the side table beside it)
    hipError_t e = hipStreamSynchronize(st);
    sync.dismiss();
    if (rc != ZG_OK) return rc;
    ZG_HIP(e);
    return ZG_OK;
}

int zg_g1_scalar_mul_batch(const uint64_t *xy, const uint8_t *inf, const uint64_t *scalars, size_t n, uint64_t *out_xy,
                           uint8_t *out_inf) {
    ZG_INIT();
    if (n && (!xy || !scalars || !out_xy || !out_inf)) {
        set_error("zg_g1_scalar_mul_batch: invalid argument");
        return ZG_ERR_INVALID;
    }
    if (n == 0) return ZG_OK;
    hipStream_t st = lib_stream();
    Scratch s_xy(n * 64), s_sc(n * 32), s_o(n * 64), s_f(2 * n);
    if (!s_xy.p || !s_sc.p || !s_o.p || !s_f.p) return ZG_ERR_NOMEM;
    SyncGuard sync(st);
    uint64_t *dxy = s_xy.as<uint64_t>(), *dsc = s_sc.as<uint64_t>(), *dout = s_o.as<uint64_t>();
    uint8_t *doinf = s_f.as<uint8_t>(), *dinf = nullptr;
    ZG_HIP(hipMemcpyAsync(dxy, xy, n * 64, hipMemcpyHostToDevice, st));
    ZG_HIP(hipMemcpyAsync(dsc, scalars, n * 32, hipMemcpyHostToDevice, st));
    if (inf) {
        dinf = doinf + n;
        ZG_HIP(hipMemcpyAsync(dinf, inf, n, hipMemcpyHostToDevice, st));
    }
    hipLaunchKernelGGL(g1_scalar_mul_kernel, dim3(div_up(n, 256)), dim3(256), 0, st, dxy, dinf, dsc, n, dout, doinf);
    ZG_HIP(hipGetLastError());
    ZG_HIP(hipMemcpyAsync(out_xy, dout, n * 64, hipMemcpyDeviceToHost, st));
    ZG_HIP(hipMemcpyAsync(out_inf, doinf, n, hipMemcpyDeviceToHost, st));
    ZG_HIP(hipStreamSynchronize(st));
    sync.dismiss();
    return ZG_OK;
}

}  // extern "C"
