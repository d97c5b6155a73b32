// ingest.hip — integer columns in, Montgomery field elements out: the witness matrix of a prover stage is built ON THE DEVICE.
//
// The reference holds what a cycle contributes to the R1CS as machine integers and flags — register and memory values (u64), immediates
// (signed), sums and products of two of those, one-bit circuit flags (src/zkvm/r1cs/constraints.zig:929-1223: every input of
// R1CSCycleInputs.fromTraceStep is F.fromU64 / signedI64ToField / a flag / a product of two inputs) — and only then widens every one of
// the 43 inputs of a cycle to a 32-byte Montgomery element (src/zkvm/r1cs/evaluation.zig:55-122 reads that matrix). A host that uploaded
// the widened matrix moved 1376 bytes per cycle across PCIe (round 4: 1.44 GB, 26-55 ms at 2^20 cycles, against 4.6 ms of device work on
// it). Here the columns cross as they are (156 bytes per cycle) and one launch writes the matrix in HBM.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <vector>

#include "common.hip.h"
#include "field.hip.h"
#include "fp29.hip.h"

namespace zg {

static constexpr uint32_t ING_MAX_COLS = 64;   // 64 rows x 64 columns x 32 B = 128 KiB of LDS per workgroup (kernel arguments: 64 x 32 B)
static constexpr uint32_t ING_TILE_ROWS = 64;  // one lane per row of a tile: column reads are 64 consecutive values

struct IngCol {
    uint32_t kind, a, b, pad;
    const void *data;  // device address of the column's n_rows values (nullptr for derived kinds)
    const void *aux;   // ZG_COL_LUT: device address of the table
};
struct IngArgs {
    IngCol c[ING_MAX_COLS];
};

// |v| < 2^128 as two words -> v * R mod r: a 9 x 5-limb product with five reduction steps, (2^401 mod r) * v * 2^-145 = v * 2^256
ZG_DEV Fr fr_from_u128_29(u64 lo, u64 hi) {
    constexpr u32 K401[9] = {0x1d71d770u, 0x1c54f317u, 0x0cd7268du, 0x169f4852u, 0x10d339c1u, 0x1be53833u, 0x0512c7f4u, 0x061af979u, 0x001f007eu};
    F29 k, b;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        k.l[i] = K401[i];
        b.l[i] = 0;
    }
    b.l[0] = (u32)lo & Fr29::MASK;
    b.l[1] = (u32)(lo >> 29) & Fr29::MASK;
    b.l[2] = ((u32)(lo >> 58) | ((u32)hi << 6)) & Fr29::MASK;
    b.l[3] = (u32)(hi >> 23) & Fr29::MASK;
    b.l[4] = (u32)(hi >> 52);
    return fr29_out(f29t_mul_short<Fr29, 5>(k, b));
}

ZG_DEV void tile_store(uint4 *slot, const Fr &v) {
    slot[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    slot[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}
ZG_DEV Fr tile_load(const uint4 *slot) {
    const uint4 a = slot[0], b = slot[1];
    Fr v;
    v.l[0] = a.x; v.l[1] = a.y; v.l[2] = a.z; v.l[3] = a.w;
    v.l[4] = b.x; v.l[5] = b.y; v.l[6] = b.z; v.l[7] = b.w;
    return v;
}

// A workgroup owns 64 consecutive rows. A WAVE converts one column at a time (the kind is wave-uniform: no divergence; lane = row, so a
// column's 64 values are one coalesced read) into an LDS tile laid out like the output; derived columns (products) are formed from the
// tile after a barrier; then the tile leaves as one contiguous, fully coalesced block of 64 x n_cols x 32 bytes.
__global__ void __launch_bounds__(512) rows_from_columns_kernel(IngArgs args, uint32_t n_cols, size_t n_rows, uint64_t *out) {
    extern __shared__ uint4 tile[];  // [row][col][2]
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const size_t row0 = (size_t)blockIdx.x * ING_TILE_ROWS, row = row0 + lane;
    const bool live = row < n_rows;
    for (uint32_t c = wave; c < n_cols; c += nw) {
        const IngCol d = args.c[c];
        if (d.kind == ZG_COL_MUL) continue;
        Fr v = Fr::zero();
        if (live) {
            switch (d.kind) {
                case ZG_COL_U8: v = fr_from_u64_29(reinterpret_cast<const uint8_t *>(d.data)[row]); break;
                case ZG_COL_U32: v = fr_from_u64_29(reinterpret_cast<const uint32_t *>(d.data)[row]); break;
                case ZG_COL_U64: v = fr_from_u64_29(reinterpret_cast<const uint64_t *>(d.data)[row]); break;
                case ZG_COL_I64: {  // signedI64ToField (src/zkvm/r1cs/constraints.zig:868-876): negative -> F.zero().sub(F.fromU64(-val))
                    const int64_t s = reinterpret_cast<const int64_t *>(d.data)[row];
                    v = fr_from_u64_29(s < 0 ? (uint64_t)0 - (uint64_t)s : (uint64_t)s);
                    if (s < 0) v = fe_neg(v);
                    break;
                }
                case ZG_COL_I128:
                case ZG_COL_U128: {
                    const uint64_t *p = reinterpret_cast<const uint64_t *>(d.data) + 2 * row;
                    uint64_t lo = p[0], hi = p[1];
                    const bool neg = d.kind == ZG_COL_I128 && (hi >> 63);
                    if (neg) {  // two's complement magnitude
                        lo = ~lo + 1;
                        hi = ~hi + (lo == 0 ? 1 : 0);
                    }
                    v = fr_from_u128_29(lo, hi);
                    if (neg) v = fe_neg(v);
                    break;
                }
                case ZG_COL_FR: v = fe_load<FrParams>(reinterpret_cast<const uint64_t *>(d.data) + 4 * row); break;
                case ZG_COL_BIT: {  // bit a of a packed flag word of b bytes per row: 0 or F.one()
                    uint64_t w = d.b == 8 ? reinterpret_cast<const uint64_t *>(d.data)[row]
                               : d.b == 4 ? (uint64_t) reinterpret_cast<const uint32_t *>(d.data)[row]
                                          : (uint64_t) reinterpret_cast<const uint8_t *>(d.data)[row];
                    if ((w >> d.a) & 1) v = Fr::one();
                    break;
                }
                case ZG_COL_LUT: {  // table[index], an index past the table reads as zero
                    const uint32_t ix = d.a == 4 ? reinterpret_cast<const uint32_t *>(d.data)[row]
                                      : d.a == 2 ? (uint32_t) reinterpret_cast<const uint16_t *>(d.data)[row]
                                                 : (uint32_t) reinterpret_cast<const uint8_t *>(d.data)[row];
                    if (ix < d.b) v = fe_load<FrParams>(reinterpret_cast<const uint64_t *>(d.aux) + 4 * (size_t)ix);
                    break;
                }
                default: break;  // ZG_COL_ZERO
            }
        }
        tile_store(tile + ((size_t)lane * n_cols + c) * 2, v);
    }
    __syncthreads();
    // derived columns: col[a] * col[b] (+ a 128-bit two's-complement addend when the column has data). Depth 1 = both factors are plain
    // columns, depth 2 = a factor is itself a depth-1 product (IngCol.pad holds the depth): one pass and one barrier per depth.
    for (uint32_t depth = 1; depth <= 2; depth++) {
        for (uint32_t c = wave; c < n_cols; c += nw) {
            const IngCol d = args.c[c];
            if (d.kind != ZG_COL_MUL || d.pad != depth) continue;
            Fr v = Fr::zero();
            if (live) {
                v = fr_mul29v(tile_load(tile + ((size_t)lane * n_cols + d.a) * 2), tile_load(tile + ((size_t)lane * n_cols + d.b) * 2));
                if (d.data) {
                    const uint64_t *p = reinterpret_cast<const uint64_t *>(d.data) + 2 * row;
                    uint64_t lo = p[0], hi = p[1];
                    const bool neg = hi >> 63;
                    if (neg) {
                        lo = ~lo + 1;
                        hi = ~hi + (lo == 0 ? 1 : 0);
                    }
                    const Fr add = fr_from_u128_29(lo, hi);
                    v = neg ? fe_sub(v, add) : fe_add(v, add);
                }
            }
            tile_store(tile + ((size_t)lane * n_cols + c) * 2, v);
        }
        __syncthreads();
    }
    const size_t rows_here = n_rows - row0 < ING_TILE_ROWS ? n_rows - row0 : ING_TILE_ROWS;
    const size_t n16 = rows_here * n_cols * 2;
    uint4 *dst = reinterpret_cast<uint4 *>(out) + row0 * n_cols * 2;
    for (size_t i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = tile[i];
}

static size_t col_width(uint32_t kind, uint32_t b, uint32_t a_of_lut = 0) {
    switch (kind) {
        case ZG_COL_U8: return 1;
        case ZG_COL_U32: return 4;
        case ZG_COL_U64: case ZG_COL_I64: return 8;
        case ZG_COL_I128: case ZG_COL_U128: return 16;
        case ZG_COL_FR: return 32;
        case ZG_COL_BIT: return b;
        case ZG_COL_MUL: return 16;  // the optional addend (no data: nothing crosses)
        case ZG_COL_LUT: return a_of_lut;
        default: return 0;
    }
}
// 1: both factors are plain columns; 2: a factor is a depth-1 product; 0: anything else (out of range, itself, a deeper chain)
static uint32_t mul_depth(const zg_col_t *cols, size_t n_cols, size_t c) {
    uint32_t depth = 1;
    for (uint32_t f : {cols[c].a, cols[c].b}) {
        if (f >= n_cols || f == c) return 0;
        if (cols[f].kind != ZG_COL_MUL) continue;
        for (uint32_t g : {cols[f].a, cols[f].b})
            if (g >= n_cols || cols[g].kind == ZG_COL_MUL) return 0;
        depth = 2;
    }
    return depth;
}
static int validate_cols(const zg_col_t *cols, size_t n_cols, size_t n_rows, uint64_t *d_rows) {
    if (!cols || n_cols == 0 || n_cols > ING_MAX_COLS || (n_rows && !d_rows)) {
        set_error("zg_fr_rows_from_columns: 1..64 columns, an output matrix");
        return ZG_ERR_INVALID;
    }
    for (size_t c = 0; c < n_cols; c++) {
        const zg_col_t &d = cols[c];
        bool ok = d.kind <= ZG_COL_LUT;
        if (ok && d.kind == ZG_COL_LUT) ok = (d.a == 1 || d.a == 2 || d.a == 4) && (d.b == 0 || d.aux) && (d.data || n_rows == 0);
        else
        if (ok && d.kind == ZG_COL_MUL) ok = mul_depth(cols, n_cols, c) != 0;
        else if (ok && d.kind == ZG_COL_BIT) ok = (d.b == 1 || d.b == 4 || d.b == 8) && d.a < 8 * d.b && d.data;
        else if (ok && d.kind != ZG_COL_ZERO) ok = d.data != nullptr || n_rows == 0;
        if (!ok) {
            set_error("zg_fr_rows_from_columns: column " + std::to_string(c) + ": unknown kind, missing data or table, a bit outside its word, an index width other than 1 / 2 / 4, or a product nested deeper than two");
            return ZG_ERR_INVALID;
        }
    }
    return ZG_OK;
}
static PerDeviceOnce g_ing_attr;
static int launch_rows_from_columns(const IngArgs &args, size_t n_cols, size_t n_rows, uint64_t *d_rows, hipStream_t st) {
    if (n_rows == 0) return ZG_OK;
    const size_t lds = (size_t)ING_TILE_ROWS * n_cols * 32;
    ZG_HIP(g_ing_attr.run([] { return hipFuncSetAttribute((const void *)rows_from_columns_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ING_TILE_ROWS * ING_MAX_COLS * 32); }));
    hipLaunchKernelGGL(rows_from_columns_kernel, dim3(div_up(n_rows, ING_TILE_ROWS)), dim3(512), lds, st, args, (uint32_t)n_cols, n_rows, d_rows);
    ZG_HIP(hipGetLastError());
    return ZG_OK;
}

// host columns -> the matrix at d_rows, on st; returns after the stream is synchronised (the staging buffer goes back to the pool)
int rows_from_host_columns(const zg_col_t *cols, size_t n_cols, size_t n_rows, uint64_t *d_rows, hipStream_t st) {
    ZG_TRY(validate_cols(cols, n_cols, n_rows, d_rows));
    if (n_rows == 0) return ZG_OK;
    // several ZG_COL_BIT columns usually share one packed word array: upload each distinct (pointer, size) once
    struct Src { const void *host; size_t bytes, off; };
    std::vector<Src> srcs;
    size_t total = 0;
    auto source = [&](const void *host, size_t bytes) -> size_t {
        for (size_t s = 0; s < srcs.size(); s++)
            if (srcs[s].host == host && srcs[s].bytes == bytes) return s;
        srcs.push_back(Src{host, bytes, total});
        total += (bytes + 255) & ~(size_t)255;
        return srcs.size() - 1;
    };
    size_t src_of[ING_MAX_COLS], aux_of[ING_MAX_COLS];
    for (size_t c = 0; c < n_cols; c++) {
        src_of[c] = aux_of[c] = (size_t)-1;
        const size_t w = col_width(cols[c].kind, cols[c].b, cols[c].a);
        if (w && cols[c].data) src_of[c] = source(cols[c].data, w * n_rows);
        if (cols[c].kind == ZG_COL_LUT && cols[c].b) aux_of[c] = source(cols[c].aux, (size_t)cols[c].b * 32);
    }
    // Columns carved out of one host slab (zolt::CycleColumns; any caller that fills one allocation) cross as ONE copy: issuing a copy costs
    // the host ~15 us, a 2 MB column slice takes 40 us on the link — sixteen small copies per slice of a streamed build left the DMA engine
    // waiting for the host. When the sources span little more than their own bytes, the span is copied and every source keeps its offset.
    // (a gap shorter than a page lies in pages its two neighbours already occupy: reading it cannot fault, whoever allocated the sources)
    if (srcs.size() > 1) {
        std::vector<size_t> order(srcs.size());
        for (size_t k = 0; k < order.size(); k++) order[k] = k;
        std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return srcs[x].host < srcs[y].host; });
        const uintptr_t lo = reinterpret_cast<uintptr_t>(srcs[order[0]].host);
        uintptr_t hi = lo + srcs[order[0]].bytes;
        bool dense = true;
        for (size_t k = 1; k < order.size() && dense; k++) {
            const uintptr_t a = reinterpret_cast<uintptr_t>(srcs[order[k]].host);
            if (a > hi && a - hi >= 4096) dense = false;
            hi = std::max(hi, a + srcs[order[k]].bytes);
        }
        if (dense) {
            const size_t head = lo & 255;  // the span sits at the same offset from a 256-byte boundary on both sides: every source keeps its alignment
            for (Src &s : srcs) {
                s.off = head + (reinterpret_cast<uintptr_t>(s.host) - lo);
                s.bytes = 0;  // carries its offset only
            }
            total = head + (hi - lo);
            srcs.push_back(Src{reinterpret_cast<const void *>(lo), hi - lo, head});  // the one copy
        }
    }
    const bool split = setup_times_enabled();
    const double t0 = split ? now_ms() : 0;
    Scratch stage(total ? total : 16);
    if (!stage.p) return ZG_ERR_NOMEM;
    const double t1 = split ? now_ms() : 0;
    SyncGuard sync(st);
    for (const Src &s : srcs)
        if (s.bytes) ZG_HIP(hipMemcpyAsync(stage.as<char>() + s.off, s.host, s.bytes, hipMemcpyHostToDevice, st));
    if (split) ZG_HIP(hipStreamSynchronize(st));
    const double t2 = split ? now_ms() : 0;
    IngArgs args{};
    for (size_t c = 0; c < n_cols; c++)
        args.c[c] = IngCol{cols[c].kind, cols[c].a, cols[c].b, cols[c].kind == ZG_COL_MUL ? mul_depth(cols, n_cols, c) : 0u,
                           src_of[c] == (size_t)-1 ? nullptr : (const void *)(stage.as<char>() + srcs[src_of[c]].off),
                           aux_of[c] == (size_t)-1 ? nullptr : (const void *)(stage.as<char>() + srcs[aux_of[c]].off)};
    ZG_TRY(launch_rows_from_columns(args, n_cols, n_rows, d_rows, st));
    ZG_HIP(hipStreamSynchronize(st));
    sync.dismiss();
    if (split) {
        SetupTimes &tm = setup_times();
        tm = SetupTimes{};
        tm.alloc_ms = t1 - t0;
        tm.h2d_ms = t2 - t1;
        tm.kernel_ms = now_ms() - t2;
    }
    return ZG_OK;
}

int ingest_u64_to_fr(const uint64_t *d_vals, size_t n, uint64_t *d_out, hipStream_t st) {
    IngArgs args{};
    args.c[0] = IngCol{ZG_COL_U64, 0, 0, 0, d_vals, nullptr};
    return launch_rows_from_columns(args, 1, n, d_out, st);
}

}  // namespace zg

using namespace zg;

extern "C" {

int zg_fr_rows_from_columns_dev(const zg_col_t *cols, size_t n_cols, size_t n_rows, uint64_t *d_rows, void *stream) {
    ZG_INIT();
    ZG_TRY(validate_cols(cols, n_cols, n_rows, d_rows));
    IngArgs args{};
    for (size_t c = 0; c < n_cols; c++)
        args.c[c] = IngCol{cols[c].kind, cols[c].a, cols[c].b, cols[c].kind == ZG_COL_MUL ? mul_depth(cols, n_cols, c) : 0u, cols[c].data, cols[c].aux};
    return launch_rows_from_columns(args, n_cols, n_rows, d_rows, pick_stream(stream));
}

// Host columns. Every distinct column array crosses PCIe once, straight from the caller's memory (pinned memory from zg_host_alloc is copied
// by DMA at link rate; pageable memory goes through the HIP runtime's staging) into one pooled device buffer; the conversion launch follows
// on the same stream and the call returns when the matrix is complete.
int zg_fr_rows_from_columns(const zg_col_t *cols, size_t n_cols, size_t n_rows, uint64_t *d_rows) {
    ZG_INIT();
    return rows_from_host_columns(cols, n_cols, n_rows, d_rows, lib_stream());
}

// Pinned host memory for callers that fill large inputs in place (trace columns, scalar vectors): copies from it run by DMA at link rate
// and do not depend on the page state of the process (hipHostMalloc; freed by zg_host_free, not pooled: the caller owns the lifetime).
int zg_host_alloc(size_t bytes, void **ptr) {
    ZG_INIT();
    if (!ptr) return ZG_ERR_INVALID;
    hipError_t e = hipHostMalloc(ptr, bytes ? bytes : 16);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error(std::string("hipHostMalloc: ") + hipGetErrorString(e));
        return ZG_ERR_NOMEM;
    }
    return ZG_OK;
}
int zg_host_free(void *ptr) {  // no ZG_INIT(): a holder's destructor may run after zg_shutdown, and must not bring the library up again
    if (ptr) ZG_HIP(hipHostFree(ptr));
    return ZG_OK;
}

}  // extern "C"
