//! Glue between Zolt's module APIs and libzolt_gpu.so, for Zolt's src/gpu/backend.zig.
//! COMPILE-UNVERIFIED (no Zig toolchain in the build image); Zig >= 0.14. INTEGRATION.md shows where the reference's
//! `pub fn` bodies call into this file; the `pub` signatures of src/msm, src/poly and src/subprotocols do not change.
//!
//! Everything here is generic over the reference's own types, so this file imports nothing from Zolt: `F` is
//! BN254Scalar, `G` is BN254BaseField, `Affine` is msm.AffinePoint(G) — any struct with `.x.limbs`, `.y.limbs`,
//! `.infinity`, `identity()` and `fromCoords(x, y)` (src/msm/mod.zig:15-49).
//!
//! Three rules this file enforces (each was a hazard in the first version of the shim):
//!  1. TYPE GATE. The GPU kernels are BN254 G1 only. The reference also instantiates MSM(Fr, Fr) with off-curve "points"
//!     (src/msm/mod.zig:853-873,911-936; commitment/mod.zig:930,1000). `isBn254Pair` decides at comptime from the fields'
//!     Montgomery constants; every other instantiation keeps the original Zig body (`msmCompute*` return null).
//!  2. OWNERSHIP, NOT A CACHE. A device table belongs to an `SrsHandle` stored IN SetupParams: created by setup()/load(), freed by
//!     deinit(). Nothing is keyed by (ptr, len), so a re-used stack or heap address can never meet a stale table, and no
//!     other thread can free a handle that is in use (the owner outlives its commits).
//!  3. AD-HOC SLICES ARE ONE-SHOT. MSM.compute on a temporary slice (dory.zig row commitments, folded G vectors, tests,
//!     src/bench.zig:261-268) uploads with `expected_uses = 1` (no precompute table: the build would cost more than it
//!     saves), computes, and frees before returning; below `one_shot_min_points` the CPU body is faster and is kept.
const std = @import("std");
pub const ffi = @import("ffi.zig");

pub const Error = error{ GpuFailure, OutOfMemory, SumcheckVerificationFailed };

/// BN254 Montgomery R = 2^256 mod r / mod q (src/field/mod.zig:23-28, 60-65): what F.one() / G.one() hold.
const FR_ONE = [4]u64{ 0xac96341c4ffffffb, 0x36fc76959f60cd29, 0x666ea36f7879462e, 0x0e0a77c19a07df2f };
const FP_ONE = [4]u64{ 0xd35d438dc58f0d9d, 0x0a78eb28f5c70b3d, 0x666ea36f7879462c, 0x0e0a77c19a07df2f };

/// Is (F, G) = (BN254 scalar field, BN254 base field), both 4x64-bit Montgomery? Anything else must not reach the GPU. Every
/// input is a comptime type, so a call site writes `if (comptime gpu.isBn254Pair(F, G))` and the other branch is never analysed
/// (the usual `if (comptime cond) return ...;` idiom: statements behind a comptime-known return are dead code).
pub fn isBn254Pair(comptime F: type, comptime G: type) bool {
    if (@sizeOf(F) != 32 or @sizeOf(G) != 32) return false;
    if (!@hasDecl(F, "one") or !@hasDecl(G, "one")) return false;
    if (!@hasField(F, "limbs") or !@hasField(G, "limbs")) return false;
    const f1 = F.one();
    const g1 = G.one();
    return std.mem.eql(u64, &f1.limbs, &FR_ONE) and std.mem.eql(u64, &g1.limbs, &FP_ONE);
}

/// Is F the BN254 scalar field (the poly / sumcheck kernels are Fr only)?
pub fn isBn254Scalar(comptime F: type) bool {
    if (@sizeOf(F) != 32 or !@hasDecl(F, "one") or !@hasField(F, "limbs")) return false;
    const f1 = F.one();
    return std.mem.eql(u64, &f1.limbs, &FR_ONE);
}

var init_once = std.once(initDevice);
var available: bool = false;
var n_devices: c_int = 1;

fn initDevice() void {
    // ZOLT_GPU=0 keeps the original Zig bodies. ZOLT_GPU_DEVICES=n (or "all") lets this ONE process drive n GPUs through the
    // sharded entry points (ParallelMSM / batchCommit); otherwise ZOLT_GPU_DEVICE selects the single GPU of this process.
    if (std.posix.getenv("ZOLT_GPU")) |v| {
        if (v.len > 0 and v[0] == '0') return;
    }
    // a library built from another major version of the header is not this binding's library: keep the Zig bodies
    if ((ffi.zg_abi_version() >> 16) != ffi.ABI_MAJOR) return;
    if (std.posix.getenv("ZOLT_GPU_DEVICES")) |v| {
        const n: c_int = if (std.mem.eql(u8, v, "all")) 0 else (std.fmt.parseInt(c_int, v, 10) catch 1);
        available = ffi.zg_init_devices(n) == ffi.OK;
        if (available) n_devices = ffi.zg_n_devices();
        return;
    }
    var dev: c_int = 0;
    if (std.posix.getenv("ZOLT_GPU_DEVICE")) |v| dev = std.fmt.parseInt(c_int, v, 10) catch 0;
    available = ffi.zg_init(dev) == ffi.OK;
}

pub fn enabled() bool {
    init_once.call();
    return available;
}

pub fn deviceCount() usize {
    _ = enabled();
    return @intCast(n_devices);
}

pub fn lastError() []const u8 {
    return std.mem.span(ffi.zg_last_error());
}

fn limbsOf(comptime F: type, s: []const F) [*]const u64 {
    comptime std.debug.assert(@sizeOf(F) == 32); // struct { limbs: [4]u64 }
    return @ptrCast(s.ptr);
}

fn affineFrom(comptime Affine: type, xy: *const [8]u64, inf: u8) Affine {
    if (inf != 0) return Affine.identity();
    return Affine.fromCoords(.{ .limbs = xy[0..4].* }, .{ .limbs = xy[4..8].* });
}

/// AffinePoint is an auto-layout struct (field order not ABI-stable, src/msm/mod.zig:19-21): pack x‖y into n×8 u64 + n flags.
const Packed = struct {
    xy: []u64,
    inf: []u8,
    fn init(comptime Affine: type, a: std.mem.Allocator, bases: []const Affine) Error!Packed {
        const xy = a.alloc(u64, @max(bases.len, 1) * 8) catch return Error.OutOfMemory;
        errdefer a.free(xy);
        const inf = a.alloc(u8, @max(bases.len, 1)) catch return Error.OutOfMemory;
        for (bases, 0..) |p, i| {
            @memcpy(xy[8 * i .. 8 * i + 4], &p.x.limbs);
            @memcpy(xy[8 * i + 4 .. 8 * i + 8], &p.y.limbs);
            inf[i] = @intFromBool(p.infinity);
        }
        return .{ .xy = xy, .inf = inf };
    }
    fn deinit(self: Packed, a: std.mem.Allocator) void {
        a.free(self.xy);
        a.free(self.inf);
    }
};

// ---------------------------------------------------------------------------------------------------------------
// Size gates: the CPU / GPU crossover of every host-pointer wrapper, MEASURED on an MI355X box against the reference's CPU bodies
// restated in C (tools/crossover.py -> profiles/r4_crossover.json, re-measured in round 4; tests/test_abi_and_host.py holds these constants to that file).
// Below its gate a wrapper returns null and the caller keeps its Zig body: a host-pointer call pays upload + launch + download
// (>= 30-40 us), more than the CPU needs for the 2^8..2^13-entry tables of a small trace (logs/zolt.log: log_t = 8).
// ---------------------------------------------------------------------------------------------------------------
pub const srs_commit_min_points: usize = 16; // zg_msm_g1 on a resident handle: 0.18 ms against 0.66 ms at 16 points
pub const one_shot_min_points: usize = 64; // upload + table-less MSM + free: 1.36 ms against 1.85 ms at 64 points (16 points: 1.26 against 0.65); round 4's table-less plan did not move the gate
pub const eq_table_min_entries: usize = 4096; // 2^12 entries: 42 us against 69 us (2^10: 36 against 19)
pub const bind_low_min_entries: usize = 16384; // in place: the table crosses PCIe both ways (2^12: 71 us against 60)
pub const bind_high_min_entries: usize = 4096; // 2^12: 70 us against 83
pub const run_sumcheck_min_entries: usize = 4096; // whole protocol on the device: 0.086 ms against 0.21 ms (2^10: 0.067 against 0.057)
pub const open_min_entries: usize = 16; // HyperKZG.open: 0.25 ms against 1.2 ms at 16 evaluations
// round-3 additions (tools/crossover.py --only lt_table,weighted_colsum -> profiles/r3_crossover_stage3.json)
pub const lt_table_min_entries: usize = 256; // LtPolynomial over the cube (ValEvaluationProver.init): 44 us against 71 us at 2^8 (2^6: 41 against 13)
pub const weighted_colsum_min_entries: usize = 4096; // 86 us against 149 us at 2^12 (2^10: 70 against 41): the Q tables of Stage 3's prefix / suffix provers, Dory's vector-matrix product

// ---------------------------------------------------------------------------------------------------------------
// SrsHandle: the device image of SetupParams.powers_of_tau_g1 (src/poly/commitment/mod.zig:122-140). A FIELD of SetupParams
// (`gpu: gpu.SrsHandle = .{}`): HyperKZG.setup / SRS loaders call init(), SetupParams.deinit calls deinit(). One table on one
// GPU, or — when the process drives several (ZOLT_GPU_DEVICES) — one shard per GPU.
// ---------------------------------------------------------------------------------------------------------------
/// Machine words in pinned host memory (zg_host_alloc): where a prover keeps the word vectors it commits to (commitWords) and any other
/// large per-proof input. Copies from it run at link rate whatever the page state of the process; allocate once, refill per proof.
pub const PinnedWords = struct {
    ptr: ?[*]u64 = null,
    len: usize = 0,

    pub fn init(n: usize) PinnedWords {
        var p: ?*anyopaque = null;
        if (!enabled() or n == 0 or ffi.zg_host_alloc(n * 8, &p) != ffi.OK or p == null) return .{};
        return .{ .ptr = @ptrCast(@alignCast(p.?)), .len = n };
    }
    pub fn slice(self: PinnedWords) []u64 {
        return if (self.ptr) |p| p[0..self.len] else &[_]u64{};
    }
    pub fn deinit(self: *PinnedWords) void {
        if (self.ptr) |p| _ = ffi.zg_host_free(@ptrCast(p));
        self.* = .{};
    }
};

pub const SrsHandle = struct {
    bases: ffi.Bases = null,
    sharded: ffi.ShardedBases = null,
    len: usize = 0,

    /// Upload `powers` (kept resident until deinit). On any failure the handle stays empty and every caller keeps its CPU body.
    pub fn init(comptime Affine: type, powers: []const Affine) SrsHandle {
        var h: SrsHandle = .{};
        if (!enabled() or powers.len == 0) return h;
        const a = std.heap.page_allocator;
        const pk = Packed.init(Affine, a, powers) catch return h;
        defer pk.deinit(a);
        if (n_devices > 1) {
            if (ffi.zg_g1_bases_upload_sharded(pk.xy.ptr, pk.inf.ptr, powers.len, null, &h.sharded) != ffi.OK) h.sharded = null;
        } else {
            if (ffi.zg_g1_bases_upload(pk.xy.ptr, pk.inf.ptr, powers.len, null, &h.bases) != ffi.OK) h.bases = null;
        }
        if (h.ready()) h.len = powers.len;
        return h;
    }

    pub fn ready(self: *const SrsHandle) bool {
        return self.bases != null or self.sharded != null;
    }

    pub fn deinit(self: *SrsHandle) void {
        if (self.bases != null) _ = ffi.zg_g1_bases_free(self.bases);
        if (self.sharded != null) _ = ffi.zg_g1_sbases_free(self.sharded);
        self.* = .{};
    }

    /// HyperKZG.commit (src/poly/commitment/mod.zig:239-255): MSM over powers[0..n). null = not on the GPU, run the Zig body.
    pub fn commit(self: *const SrsHandle, comptime F: type, comptime Affine: type, evals: []const F) ?Affine {
        if (!self.ready()) return null;
        const n = @min(evals.len, self.len);
        if (n == 0) return Affine.identity();
        if (n < srs_commit_min_points) return null; // measured crossover: the CPU body is faster
        var xy: [8]u64 = undefined;
        var inf: u8 = 0;
        const rc = if (self.sharded != null)
            ffi.zg_msm_g1_sharded(self.sharded, n, limbsOf(F, evals), &xy, @ptrCast(&inf))
        else
            ffi.zg_msm_g1(self.bases, 0, n, limbsOf(F, evals), &xy, @ptrCast(&inf));
        if (rc != ffi.OK) return null; // the caller falls back to the CPU body; lastError() says why
        return affineFrom(Affine, &xy, inf);
    }

    /// `words` should live in pinned memory that is kept across proofs (`PinnedWords` below): a fresh pageable slice per call is pinned on
    /// the fly and unpinned again by the HIP runtime — usually cheap, 10-15 ms per proof when the allocator's address reuse is unlucky.
    /// commitBytecode / commitMemory / commitRegisters (src/zkvm/mod.zig:1518-1617) build `poly[i] = F.fromU64(word_i)` and commit to it:
    /// here the machine words cross as they are (8 bytes each, no host fromU64 per evaluation) and the conversion runs on the device.
    /// The result equals commit(evals) for evals[i] = F.fromU64(words[i]). null = run the Zig body.
    pub fn commitWords(self: *const SrsHandle, comptime Affine: type, words: []const u64) ?Affine {
        if (self.bases == null) return null; // (the sharded handle takes field elements: keep the Zig body there)
        const n = @min(words.len, self.len);
        if (n == 0) return Affine.identity();
        if (n < srs_commit_min_points) return null;
        var xy: [8]u64 = undefined;
        var inf: u8 = 0;
        if (ffi.zg_msm_g1_u64(self.bases, 0, n, words.ptr, &xy, @ptrCast(&inf)) != ffi.OK) return null;
        return affineFrom(Affine, &xy, inf);
    }

    /// HyperKZG.setup's G1 side with nothing leaving the device (src/poly/commitment/mod.zig:174-213): powers of tau, the fixed-base batch
    /// and the resident handle are built in HBM. `out` (may be empty) receives the points when the caller keeps them on the host
    /// (SetupParams.powers_of_tau_g1); a prover that only commits and opens passes an empty slice and n = the SRS size.
    /// `expected_uses`: 0 = an SRS that lives on (the table of multiples is built with it: 13 ms and 1 GB at 2^20 powers, repaid after ~12
    /// MSMs); 1..15 = a key that serves ONE proof, as `zolt prove` builds it in-process (src/main.zig:271-696) — three commits and an
    /// opening are fewer MSMs than the break-even, and the table-less plan proves the same bytes sooner (37 against 46 ms at 2^20 cycles).
    pub fn initFromTau(comptime F: type, comptime Affine: type, g1: Affine, tau: F, n: usize, out: []Affine, allocator: std.mem.Allocator, expected_uses: c_int) !SrsHandle {
        var h: SrsHandle = .{};
        if (!enabled() or n == 0 or n_devices > 1 or g1.infinity) return h;
        var g: [8]u64 = undefined;
        @memcpy(g[0..4], &g1.x.limbs);
        @memcpy(g[4..8], &g1.y.limbs);
        const cfg = ffi.MsmConfig{ .expected_uses = expected_uses };
        if (out.len == 0) {
            if (ffi.zg_hyperkzg_setup(&g, &tau.limbs, n, &cfg, null, null, &h.bases) != ffi.OK) h.bases = null;
        } else {
            const oxy = try allocator.alloc(u64, 8 * n);
            defer allocator.free(oxy);
            const oinf = try allocator.alloc(u8, n);
            defer allocator.free(oinf);
            if (ffi.zg_hyperkzg_setup(&g, &tau.limbs, n, &cfg, oxy.ptr, oinf.ptr, &h.bases) != ffi.OK) {
                h.bases = null;
            } else {
                for (out, 0..) |*r, i| r.* = affineFrom(Affine, oxy[8 * i ..][0..8], oinf[i]);
            }
        }
        if (h.ready()) h.len = n;
        return h;
    }

    /// HyperKZG.batchCommit (:558-570) / BatchMSM / ParallelBatchMSM (src/msm/mod.zig:545-565,683-748): k vectors of one length n
    /// over powers[0..n); short vectors run as ONE fused launch set, several GPUs exchange k partials in one all-gather.
    pub fn batchCommit(self: *const SrsHandle, comptime F: type, comptime Affine: type, polys: []const []const F, allocator: std.mem.Allocator) !?[]Affine {
        if (!self.ready() or polys.len == 0) return null;
        const n = @min(polys[0].len, self.len);
        for (polys) |p| if (@min(p.len, self.len) != n) return null; // ragged batch: keep the per-polynomial loop
        if (n < srs_commit_min_points) return null; // measured crossover (per vector; a fused batch only lowers it)
        const k = polys.len;
        const results = try allocator.alloc(Affine, k);
        errdefer allocator.free(results);
        const ptrs = try allocator.alloc(?[*]const u64, k);
        defer allocator.free(ptrs);
        for (polys, 0..) |p, i| ptrs[i] = limbsOf(F, p);
        const xy = try allocator.alloc(u64, 8 * k);
        defer allocator.free(xy);
        const inf = try allocator.alloc(u8, k);
        defer allocator.free(inf);
        const rc = if (self.sharded != null)
            ffi.zg_msm_g1_batch_sharded(self.sharded, n, ptrs.ptr, k, xy.ptr, inf.ptr)
        else
            ffi.zg_msm_g1_batch(self.bases, n, ptrs.ptr, k, xy.ptr, inf.ptr);
        if (rc != ffi.OK) {
            allocator.free(results);
            return null;
        }
        for (results, 0..) |*r, i| r.* = affineFrom(Affine, xy[8 * i ..][0..8], inf[i]);
        return results;
    }

    /// HyperKZG.open (:261-324) on the device: the quotient -> commit -> fold loop never leaves HBM. Returns the quotient
    /// commitments (one per variable the fold reaches) and the final evaluation; null = run the Zig body.
    pub fn open(self: *const SrsHandle, comptime F: type, comptime Affine: type, evals: []const F, point: []const F, value: F, allocator: std.mem.Allocator) !?struct { quotients: []Affine, final_eval: F } {
        if (self.bases == null) return null; // single-device handle only (the sharded form commits level by level: not offered)
        if (evals.len < open_min_entries) return null; // measured crossover
        const v = point.len;
        const q = try allocator.alloc(Affine, v);
        errdefer allocator.free(q);
        const xy = try allocator.alloc(u64, 8 * @max(v, 1));
        defer allocator.free(xy);
        const inf = try allocator.alloc(u8, @max(v, 1));
        defer allocator.free(inf);
        var fin: F = undefined;
        if (ffi.zg_hyperkzg_open(self.bases, limbsOf(F, evals), evals.len, limbsOf(F, point), v, &value.limbs, xy.ptr, inf.ptr, &fin.limbs) != ffi.OK) {
            allocator.free(q);
            return null;
        }
        for (q, 0..) |*r, i| r.* = affineFrom(Affine, xy[8 * i ..][0..8], inf[i]);
        return .{ .quotients = q, .final_eval = fin };
    }

    /// HyperKZG.batchOpen (:607-732): gamma, the combined polynomial, per-polynomial evaluations and the same loop, one call.
    pub fn batchOpen(self: *const SrsHandle, comptime F: type, comptime Affine: type, polys: []const []const F, point: []const F, allocator: std.mem.Allocator) !?struct { quotients: []Affine, evaluations: []F, final_eval: F, gamma: F } {
        if (self.bases == null) return null;
        if (polys.len == 0 or polys[0].len < open_min_entries) return null; // measured crossover (the combined polynomial has polys[0]'s length)
        const k = polys.len;
        const v = point.len;
        const ptrs = try allocator.alloc(?[*]const u64, @max(k, 1));
        defer allocator.free(ptrs);
        const lens = try allocator.alloc(usize, @max(k, 1));
        defer allocator.free(lens);
        for (polys, 0..) |p, i| {
            ptrs[i] = if (p.len > 0) limbsOf(F, p) else null;
            lens[i] = p.len;
        }
        const xy = try allocator.alloc(u64, 8 * @max(v, 1));
        defer allocator.free(xy);
        const inf = try allocator.alloc(u8, @max(v, 1));
        defer allocator.free(inf);
        const evaluations = try allocator.alloc(F, k);
        errdefer allocator.free(evaluations);
        var nq: usize = 0;
        var fin: F = undefined;
        var gamma: F = undefined;
        if (ffi.zg_hyperkzg_batch_open(self.bases, ptrs.ptr, lens.ptr, k, limbsOf(F, point), v, xy.ptr, inf.ptr, &nq, @ptrCast(evaluations.ptr), &fin.limbs, &gamma.limbs) != ffi.OK) {
            allocator.free(evaluations);
            return null;
        }
        const q = try allocator.alloc(Affine, nq);
        for (q, 0..) |*r, i| r.* = affineFrom(Affine, xy[8 * i ..][0..8], inf[i]);
        return .{ .quotients = q, .evaluations = evaluations, .final_eval = fin, .gamma = gamma };
    }
};

/// HyperKZG.setup's loop `powers[i] = MSM.scalarMul(g1, tau^i).toAffine()` (src/poly/commitment/mod.zig:194-199) as one batch:
/// `scalars[i]` = tau^i (Montgomery Fr), all over the same base point. Returns false -> run the Zig loop.
pub fn setupPowers(comptime F: type, comptime Affine: type, g1: Affine, scalars: []const F, out: []Affine, allocator: std.mem.Allocator) !bool {
    if (!enabled() or scalars.len == 0) return false;
    const n = scalars.len;
    var g: [8]u64 = undefined;
    @memcpy(g[0..4], &g1.x.limbs);
    @memcpy(g[4..8], &g1.y.limbs);
    const oxy = try allocator.alloc(u64, 8 * n);
    defer allocator.free(oxy);
    const oinf = try allocator.alloc(u8, n);
    defer allocator.free(oinf);
    // one shared base: the fixed-base kernel (a 255 x 32 table of multiples, at most 32 additions per output)
    if (ffi.zg_g1_fixed_base_mul_batch(&g, @intFromBool(g1.infinity), limbsOf(F, scalars), n, oxy.ptr, oinf.ptr) != ffi.OK) return false;
    for (out, 0..) |*r, i| r.* = affineFrom(Affine, oxy[8 * i ..][0..8], oinf[i]);
    return true;
}

// ---------------------------------------------------------------------------------------------------------------
// The R1CS witness matrix, widened on the device from integer trace columns (zg_fr_rows_from_columns). R1CSWitnessGenerator.generateWitness
// (src/zkvm/r1cs/constraints.zig:1469-1494) builds 43 field elements per cycle on the CPU; every one of them is F.fromU64 of a machine
// word, signedI64ToField of an immediate, a flag, or a sum / product of two inputs, so the shim fills COLUMNS of integers instead
// (zolt_amd/host/witness.hpp CycleColumns::fromTrace is the compiled reference for the column layout: 156 bytes per cycle) and the
// matrix the stage provers read appears in HBM. The returned table is released with ffi.zg_dev_free.
// ---------------------------------------------------------------------------------------------------------------
pub const WitnessMatrix = struct {
    ptr: ?[*]u64 = null, // device address: rows * cols elements of 4 words, cycle-major
    rows: usize = 0,
    cols: usize = 0,

    pub fn fromColumns(cols: []const ffi.Column, n_rows: usize) Error!WitnessMatrix {
        if (!enabled() or cols.len == 0 or cols.len > 64) return Error.GpuFailure;
        var raw: ?*anyopaque = null;
        if (ffi.zg_dev_alloc(@max(n_rows * cols.len * 32, 32), &raw) != ffi.OK) return Error.OutOfMemory;
        const d: [*]u64 = @ptrCast(@alignCast(raw.?));
        if (ffi.zg_fr_rows_from_columns(cols.ptr, cols.len, n_rows, d) != ffi.OK) {
            _ = ffi.zg_dev_free(raw);
            return Error.GpuFailure;
        }
        return .{ .ptr = d, .rows = n_rows, .cols = cols.len };
    }

    pub fn deinit(self: *WitnessMatrix) void {
        if (self.ptr) |p| _ = ffi.zg_dev_free(@ptrCast(p));
        self.* = .{};
    }
};

// ---------------------------------------------------------------------------------------------------------------
// MSM(F, G).compute on an arbitrary slice (src/msm/mod.zig:355-372): ONE-SHOT. No table outlives the call.
// ---------------------------------------------------------------------------------------------------------------
/// below this many points the upload + launch latency (~0.5 ms) loses to the CPU body

/// null = not handled here (wrong field types, GPU disabled, too small, or a device failure): run the original body.
pub fn msmComputeOneShot(comptime F: type, comptime G: type, comptime Affine: type, bases: []const Affine, scalars: []const F) ?Affine {
    if (comptime !isBn254Pair(F, G)) return null; // rule 1: MSM(Fr, Fr) and friends never reach the G1 kernels
    std.debug.assert(bases.len == scalars.len);
    if (bases.len < one_shot_min_points or !enabled()) return null;
    const a = std.heap.page_allocator;
    const pk = Packed.init(Affine, a, bases) catch return null;
    defer pk.deinit(a);
    const cfg = ffi.MsmConfig{ .expected_uses = 1 }; // no precompute table for a slice that is used once
    var h: ffi.Bases = null;
    if (ffi.zg_g1_bases_upload(pk.xy.ptr, pk.inf.ptr, bases.len, &cfg, &h) != ffi.OK) return null;
    defer _ = ffi.zg_g1_bases_free(h); // rule 2: the handle dies with the call — nothing to go stale
    var xy: [8]u64 = undefined;
    var inf: u8 = 0;
    if (ffi.zg_msm_g1(h, 0, bases.len, limbsOf(F, scalars), &xy, @ptrCast(&inf)) != ffi.OK) return null;
    return affineFrom(Affine, &xy, inf);
}

/// ParallelMSM.compute (src/msm/mod.zig:588-653) on an arbitrary slice with several GPUs bound: one-shot sharded upload, one
/// all-gather of 96-byte partials, device combine. null = run the Zig body.
pub fn parallelMsmOneShot(comptime F: type, comptime G: type, comptime Affine: type, bases: []const Affine, scalars: []const F) ?Affine {
    if (comptime !isBn254Pair(F, G)) return null;
    if (!enabled() or n_devices < 2) return msmComputeOneShot(F, G, Affine, bases, scalars);
    if (bases.len < one_shot_min_points * @as(usize, @intCast(n_devices))) return msmComputeOneShot(F, G, Affine, bases, scalars);
    const a = std.heap.page_allocator;
    const pk = Packed.init(Affine, a, bases) catch return null;
    defer pk.deinit(a);
    const cfg = ffi.MsmConfig{ .expected_uses = 1 };
    var h: ffi.ShardedBases = null;
    if (ffi.zg_g1_bases_upload_sharded(pk.xy.ptr, pk.inf.ptr, bases.len, &cfg, &h) != ffi.OK) return null;
    defer _ = ffi.zg_g1_sbases_free(h);
    var xy: [8]u64 = undefined;
    var inf: u8 = 0;
    if (ffi.zg_msm_g1_sharded(h, bases.len, limbsOf(F, scalars), &xy, @ptrCast(&inf)) != ffi.OK) return null;
    return affineFrom(Affine, &xy, inf);
}

// ---------------------------------------------------------------------------------------------------------------
// poly: EqPolynomial.evalsSliceWithScaling, DensePolynomial.bindLow / bindFirst / evaluate (src/poly/mod.zig). Fr only:
// callers gate with `comptime gpu.isBn254Scalar(F)`.
// ---------------------------------------------------------------------------------------------------------------
/// null = below the measured crossover (eq_table_min_entries): the caller runs its Zig loop
pub fn eqTable(comptime F: type, allocator: std.mem.Allocator, r: []const F, scaling_factor: ?F) !?[]F {
    if ((@as(usize, 1) << @intCast(r.len)) < eq_table_min_entries) return null;
    const result = try allocator.alloc(F, @as(usize, 1) << @intCast(r.len));
    errdefer allocator.free(result);
    const sc: ?[*]const u64 = if (scaling_factor) |*s| &s.limbs else null;
    if (ffi.zg_fr_eq_table(limbsOf(F, r), r.len, sc, @ptrCast(result.ptr)) != ffi.OK) return Error.GpuFailure;
    return result;
}

/// GruenSplitEqPolynomial.initWithScaling's E_vec for one half of tau (src/poly/split_eq.zig:122-171), all levels from one
/// launch: appends w.len + 1 tables (table k = eq(w[0..k), .), 2^k entries) to `vec`, each its own allocation so that
/// bind()'s pop + free (:230-246) and deinit (:185-197) stay as they are.
pub fn gruenPrefixTables(comptime F: type, allocator: std.mem.Allocator, w: []const F, vec: *std.ArrayListUnmanaged([]F)) !void {
    const total = (@as(usize, 2) << @intCast(w.len)) - 1;
    const flat = try allocator.alloc(F, total);
    defer allocator.free(flat);
    if (ffi.zg_fr_eq_prefix_tables(limbsOf(F, w), w.len, @ptrCast(flat.ptr)) != ffi.OK) return Error.GpuFailure;
    var k: usize = 0;
    while (k <= w.len) : (k += 1) {
        const size = @as(usize, 1) << @intCast(k);
        const table = try allocator.alloc(F, size);
        errdefer allocator.free(table);
        @memcpy(table, flat[size - 1 .. 2 * size - 1]);
        try vec.append(allocator, table);
    }
}

/// LtPolynomial.evaluateAtIndex for every index (src/zkvm/ram/val_evaluation.zig:309-330): what ValEvaluationProver.init loops over.
/// null = below the measured crossover (lt_table_min_entries): the caller keeps its loop
pub fn ltTable(comptime F: type, allocator: std.mem.Allocator, r_cycle: []const F) !?[]F {
    if (r_cycle.len > 30) return null; // beyond what the device entry point accepts: the caller keeps its Zig body
    if ((@as(usize, 1) << @intCast(r_cycle.len)) < lt_table_min_entries) return null;
    const out = try allocator.alloc(F, @as(usize, 1) << @intCast(r_cycle.len));
    errdefer allocator.free(out);
    if (ffi.zg_fr_lt_table(limbsOf(F, r_cycle), r_cycle.len, @ptrCast(out.ptr)) != ffi.OK) return Error.GpuFailure;
    return out;
}

/// out[k * cols + c] = sum_r weights[k * rows + r] * table[r * cols + c]: the x_hi / x_lo double loop of ShiftPrefixSuffixProver.init /
/// RegistersPrefixSuffixProver.init (src/zkvm/spartan/stage3_prover.zig:1066-1112, 2232-2290) and Dory's computeVectorMatrixProduct
/// (src/poly/commitment/dory.zig:622-642). null = below the measured crossover (weighted_colsum_min_entries)
pub fn weightedColsum(comptime F: type, allocator: std.mem.Allocator, table: []const F, rows: usize, cols: usize, weights: []const F) !?[]F {
    if (table.len < weighted_colsum_min_entries) return null;
    // shapes the device entry point does not take (or that would read past a slice): the caller keeps its Zig body
    if (rows == 0 or cols == 0 or table.len != rows * cols or weights.len % rows != 0) return null;
    const m = weights.len / rows;
    if (m == 0 or m > 4) return null;
    const out = try allocator.alloc(F, m * cols);
    errdefer allocator.free(out);
    if (ffi.zg_fr_weighted_colsum(limbsOf(F, table), rows, cols, limbsOf(F, weights), m, @ptrCast(out.ptr)) != ffi.OK) return Error.GpuFailure;
    return out;
}

/// in place; the caller then halves its live length / decrements num_vars as the original does (:160-175).
/// false = below the measured crossover (bind_low_min_entries): nothing was done, the caller runs its Zig loop
pub fn bindLow(comptime F: type, evaluations: []F, value: F) Error!bool {
    if (evaluations.len < bind_low_min_entries) return false;
    if (ffi.zg_fr_bind_low(@ptrCast(evaluations.ptr), evaluations.len, &value.limbs) != ffi.OK) return Error.GpuFailure;
    return true;
}

/// new allocation of len / 2 entries, like bindFirst (:128-149); null = below the measured crossover (bind_high_min_entries)
pub fn bindHigh(comptime F: type, allocator: std.mem.Allocator, evaluations: []const F, value: F) !?[]F {
    if (evaluations.len < bind_high_min_entries) return null;
    const out = try allocator.alloc(F, evaluations.len / 2);
    errdefer allocator.free(out);
    if (ffi.zg_fr_bind_high(limbsOf(F, evaluations), evaluations.len, &value.limbs, @ptrCast(out.ptr)) != ffi.OK) return Error.GpuFailure;
    return out;
}

pub fn denseEvaluate(comptime F: type, evaluations: []const F, point: []const F) Error!F {
    var out: F = undefined;
    if (ffi.zg_fr_dense_evaluate(limbsOf(F, evaluations), point.len, limbsOf(F, point), &out.limbs) != ffi.OK) return Error.GpuFailure;
    return out;
}

/// DensePolynomial.scale (:112-126)
pub fn scale(comptime F: type, allocator: std.mem.Allocator, evaluations: []const F, scalar: F) ![]F {
    const out = try allocator.alloc(F, evaluations.len);
    errdefer allocator.free(out);
    if (ffi.zg_fr_scale(limbsOf(F, evaluations), evaluations.len, &scalar.limbs, @ptrCast(out.ptr)) != ffi.OK) return Error.GpuFailure;
    return out;
}

// ---------------------------------------------------------------------------------------------------------------
// Sumcheck(F).Prover as a device-resident session (src/subprotocols/mod.zig:55-133). The table stays in HBM; per round two
// field elements come back and one challenge goes in, so any host transcript (src/transcripts) keeps working unchanged.
// With several GPUs bound the table is sharded over them (zg_sumcheck_*_sharded): same messages, bit for bit.
// ---------------------------------------------------------------------------------------------------------------
pub fn SumcheckSession(comptime F: type) type {
    return struct {
        const Self = @This();
        handle: ffi.Session = null,
        sharded: ffi.ShardedSession = null,

        pub fn open(evaluations: []const F, layout: c_int) Error!Self {
            var self: Self = .{};
            if (!enabled()) return Error.GpuFailure;
            if (n_devices > 1) {
                if (ffi.zg_sumcheck_open_sharded(limbsOf(F, evaluations), evaluations.len, layout, &self.sharded) != ffi.OK) return Error.GpuFailure;
            } else {
                if (ffi.zg_sumcheck_open(limbsOf(F, evaluations), evaluations.len, layout, &self.handle) != ffi.OK) return Error.GpuFailure;
            }
            return self;
        }
        /// A session over a table that already sits in device memory (`d_table`: `len` elements, e.g. Az / Bz materialised by
        /// zg_fr_rows_affine_records_dev). `borrow`: no copy — the session reads the caller's table in place until its first bind has run
        /// (zg_sumcheck_open_dev_borrowed: the caller keeps it alive and unchanged until a later call on the session has returned), its
        /// own buffers hold the folds only. One device only.
        pub fn openDevice(d_table: [*]const u64, len: usize, layout: c_int, borrow: bool) Error!Self {
            var self: Self = .{};
            if (!enabled() or n_devices > 1) return Error.GpuFailure;
            const rc = if (borrow)
                ffi.zg_sumcheck_open_dev_borrowed(d_table, len, layout, null, &self.handle)
            else
                ffi.zg_sumcheck_open_dev(d_table, len, layout, null, &self.handle);
            if (rc != ffi.OK) return Error.GpuFailure;
            return self;
        }
        /// nextRound (:69-109): coefficients [g(0), g(1) - g(0)]
        pub fn roundCoeffs(self: *Self) Error![2]F {
            var g0: F = undefined;
            var g1: F = undefined;
            const rc = if (self.sharded != null)
                ffi.zg_sumcheck_round_sums_sharded(self.sharded, &g0.limbs, &g1.limbs)
            else
                ffi.zg_sumcheck_round_sums(self.handle, &g0.limbs, &g1.limbs);
            if (rc != ffi.OK) return Error.GpuFailure;
            return .{ g0, g1.sub(g0) };
        }
        /// receiveChallenge (:112-122)
        pub fn bind(self: *Self, challenge: F) Error!void {
            const rc = if (self.sharded != null)
                ffi.zg_sumcheck_bind_sharded(self.sharded, &challenge.limbs)
            else
                ffi.zg_sumcheck_bind(self.handle, &challenge.limbs);
            if (rc != ffi.OK) return Error.GpuFailure;
        }
        pub fn len(self: *const Self) usize {
            return if (self.sharded != null) ffi.zg_sumcheck_len_sharded(self.sharded) else ffi.zg_sumcheck_len(self.handle);
        }
        /// getFinalEval (:130-133)
        pub fn finalEval(self: *Self) Error!F {
            var out: F = undefined;
            const rc = if (self.sharded != null)
                ffi.zg_sumcheck_final_sharded(self.sharded, &out.limbs)
            else
                ffi.zg_sumcheck_final(self.handle, &out.limbs);
            if (rc != ffi.OK) return Error.GpuFailure;
            return out;
        }
        /// RafEvaluationProver.computeRoundPolynomialCubic's two sums s(0), s(2) over this LOW_PAIR session's table
        /// (src/zkvm/ram/raf_checking.zig:335-410); `base` = start_address + 8 * sum_j bound_j 2^j, `current_power` = 8 * 2^round.
        /// s(1) = claim - s(0) and s(3) = s(0) - 3 s(1) + 3 s(2) stay host code; the bind of the round is `bind`.
        /// RafEvaluationProver.computeInitialClaim (src/zkvm/ram/raf_checking.zig:312-321): sum_k ra(k) * F.fromU64(start_address + 8 k)
        /// over this session's resident table, before the first bind — one pass on the device instead of 2^log_k host products.
        pub fn rafInitialClaim(self: *Self, start_address: u64) Error!F {
            var claim: F = undefined;
            if (self.handle == null or ffi.zg_sumcheck_raf_claim(self.handle, start_address, 8, &claim.limbs) != ffi.OK) return Error.GpuFailure;
            return claim;
        }
        pub fn rafRoundSums(self: *Self, base: F, current_power: u64) Error![2]F {
            var s0: F = undefined;
            var s2: F = undefined;
            if (self.handle == null or ffi.zg_sumcheck_raf_round(self.handle, &base.limbs, current_power, &s0.limbs, &s2.limbs) != ffi.OK) return Error.GpuFailure;
            return .{ s0, s2 };
        }
        /// LassoProver's address rounds on this session (opened SC_HIGH_HALF over the padded eq_evals, src/zkvm/lasso/prover.zig:153-171).
        /// `d_idx`: the u128 lookup indices uploaded once with `uploadLookupIndices`. computeAddressRoundPoly's sum_0 / sum_1 (:283-293):
        pub fn bitRoundSums(self: *Self, d_idx: DeviceIndices, round_bit: u32) Error![2]F {
            var s0: F = undefined;
            var s1: F = undefined;
            if (self.handle == null or ffi.zg_sumcheck_bit_round(self.handle, @ptrCast(d_idx.ptr), d_idx.len, round_bit, &s0.limbs, &s1.limbs) != ffi.OK) return Error.GpuFailure;
            return .{ s0, s1 };
        }
        /// receiveChallenge's address branch (:375-399): eq_evals[j] *= bit ? r : 1 - r in place; returns the new current_claim.
        /// The next round's bitRoundSums is then already computed (fused into this pass).
        pub fn bitBind(self: *Self, d_idx: DeviceIndices, round_bit: u32, challenge: F) Error!F {
            var claim: F = undefined;
            if (self.handle == null or ffi.zg_sumcheck_bit_bind(self.handle, @ptrCast(d_idx.ptr), d_idx.len, round_bit, &challenge.limbs, &claim.limbs) != ffi.OK) return Error.GpuFailure;
            return claim;
        }
        /// materialise the current table for callers that index prover.polynomial.evaluations directly (:79-92,126,132);
        /// single-device sessions only
        pub fn read(self: *Self, out: []F) Error!void {
            std.debug.assert(out.len == self.len());
            if (self.handle == null or ffi.zg_sumcheck_read(self.handle, @ptrCast(out.ptr)) != ffi.OK) return Error.GpuFailure;
        }
        pub fn close(self: *Self) void {
            if (self.handle != null) _ = ffi.zg_sumcheck_close(self.handle);
            if (self.sharded != null) _ = ffi.zg_sumcheck_close_sharded(self.sharded);
            self.* = .{};
        }
    };
}

/// k multilinear tables folded together (LowToHigh) with product-form round evaluations: the loops of ValEvaluationProver
/// (src/zkvm/ram/val_evaluation.zig:554-628), ValFinalProver (ram/val_final.zig:149-200), OutputSumcheckProver (ram/output_check.zig:375-470),
/// InstructionLookupsClaimReductionProver (claim_reductions/instruction_lookups.zig:146-240) and ProductVirtualRemainderProver
/// (spartan/product_remainder.zig:269-340). The prover struct keeps one of these instead of its `[]F` tables; the claim update, the
/// compressed-coefficient conversion and the transcript stay as they are.
pub fn ProductSession(comptime F: type) type {
    comptime std.debug.assert(isBn254Scalar(F));
    return struct {
        const Self = @This();
        handle: ffi.ProductSession = null,

        /// `tables`: up to 12 slices of the same power-of-two length (copied to the device)
        pub fn open(tables: []const []const F) Error!Self {
            std.debug.assert(tables.len >= 1 and tables.len <= 12);
            var ptrs: [12]?[*]const u64 = .{null} ** 12;
            for (tables, 0..) |t, j| ptrs[j] = limbsOf(F, t);
            var self: Self = .{};
            if (ffi.zg_psc_open(@ptrCast(&ptrs), tables.len, tables[0].len, &self.handle) != ffi.OK) return Error.GpuFailure;
            return self;
        }
        pub fn len(self: *const Self) usize {
            return ffi.zg_psc_len(self.handle);
        }
        /// [p(0), p(1), p(2), p(3)] of sum_g prod_j T[prod[j]](t) * sum_m coeff[m] * T[lin[m]](t)  (lin empty: the plain product)
        pub fn roundEvals(self: *Self, prod: []const c_int, lin: []const c_int, coeff: []const F) Error![4]F {
            std.debug.assert(lin.len == coeff.len);
            var out: [4]F = undefined;
            if (ffi.zg_psc_round_evals(self.handle, prod.ptr, prod.len, lin.ptr, limbsOf(F, coeff), lin.len, @ptrCast(&out)) != ffi.OK) return Error.GpuFailure;
            return out;
        }
        /// [p(0..3)] of a SUM of up to four product terms (ffi.PscTerm each) — ShiftSumcheckProver's two phases and InstructionInputProver
        /// (src/zkvm/spartan/stage3_prover.zig:1351-1455,2029-2100)
        /// A term with `n_prod = 4 | ffi.PSC_PAIR_SUM` is (T[prod[0]]*T[prod[1]] + T[prod[2]]*T[prod[3]]) * L: InstructionInput's two sides.
        pub fn roundExpr(self: *Self, terms: []const ffi.PscTerm) Error![4]F {
            var out: [4]F = undefined;
            if (ffi.zg_psc_round_expr(self.handle, terms.ptr, terms.len, @ptrCast(&out)) != ffi.OK) return Error.GpuFailure;
            return out;
        }
        /// bit t of `points`: roundEvals / roundExpr compute p(t) only; the other slots come back as zero. The provers that derive
        /// p(1) from the claim (stage3_prover.zig:1399-1455, 2029-2100, 2334-2389) save the products of the points they never read.
        pub fn setPoints(self: *Self, points: c_uint) Error!void {
            if (ffi.zg_psc_set_points(self.handle, points) != ffi.OK) return Error.GpuFailure;
        }
        /// Gruen's (t0, t_inf) under the split-eq weights; e_out / e_in: device tables (see `GruenDeviceTables`)
        pub fn roundGruen(self: *Self, prod: []const c_int, e_out: DeviceTable, e_in: DeviceTable) Error![2]F {
            var t0: F = undefined;
            var t_inf: F = undefined;
            if (ffi.zg_psc_round_gruen(self.handle, prod.ptr, prod.len, e_out.ptr, e_out.len, e_in.ptr, e_in.len, &t0.limbs, &t_inf.limbs) != ffi.OK) return Error.GpuFailure;
            return .{ t0, t_inf };
        }
        /// folds every table by the challenge; the next roundEvals with the same arguments is prepared by the same launch
        pub fn bind(self: *Self, challenge: F) Error!void {
            if (ffi.zg_psc_bind(self.handle, &challenge.limbs) != ffi.OK) return Error.GpuFailure;
        }
        /// each table's single remaining entry (getFinalClaims / getOpeningClaims)
        pub fn finalValues(self: *Self, out: []F) Error!void {
            if (ffi.zg_psc_final(self.handle, @ptrCast(out.ptr)) != ffi.OK) return Error.GpuFailure;
        }
        pub fn close(self: *Self) void {
            if (self.handle != null) _ = ffi.zg_psc_close(self.handle);
            self.* = .{};
        }
    };
}

/// a table in device memory (element pointer + entry count)
pub const DeviceTable = struct { ptr: ?[*]const u64 = null, len: usize = 0 };

/// GruenSplitEqPolynomial's two prefix-table sets resident in HBM for `ProductSession.roundGruen`: built once next to
/// initWithScaling; `window(k_out, k_in)` = the tables getWindowEqTables returns (E_out_vec[k_out], E_in_vec[k_in]).
pub const GruenDeviceTables = struct {
    d_out: ?*anyopaque = null,
    d_in: ?*anyopaque = null,

    pub fn init(comptime F: type, w_out: []const F, w_in: []const F) Error!GruenDeviceTables {
        var self: GruenDeviceTables = .{};
        errdefer self.deinit();
        if (ffi.zg_dev_alloc(((@as(usize, 2) << @intCast(w_out.len)) - 1) * 32, &self.d_out) != ffi.OK) return Error.GpuFailure;
        if (ffi.zg_dev_alloc(((@as(usize, 2) << @intCast(w_in.len)) - 1) * 32, &self.d_in) != ffi.OK) return Error.GpuFailure;
        if (ffi.zg_fr_eq_prefix_tables_dev(limbsOf(F, w_out), w_out.len, @ptrCast(@alignCast(self.d_out)), null) != ffi.OK) return Error.GpuFailure;
        if (ffi.zg_fr_eq_prefix_tables_dev(limbsOf(F, w_in), w_in.len, @ptrCast(@alignCast(self.d_in)), null) != ffi.OK) return Error.GpuFailure;
        return self;
    }
    pub fn window(self: *const GruenDeviceTables, k_out: usize, k_in: usize) struct { e_out: DeviceTable, e_in: DeviceTable } {
        const o: [*]const u64 = @ptrCast(@alignCast(self.d_out.?));
        const i: [*]const u64 = @ptrCast(@alignCast(self.d_in.?));
        const no = @as(usize, 1) << @intCast(k_out);
        const ni = @as(usize, 1) << @intCast(k_in);
        return .{ .e_out = .{ .ptr = o + 4 * (no - 1), .len = no }, .e_in = .{ .ptr = i + 4 * (ni - 1), .len = ni } };
    }
    pub fn deinit(self: *GruenDeviceTables) void {
        if (self.d_out != null) _ = ffi.zg_dev_free(self.d_out);
        if (self.d_in != null) _ = ffi.zg_dev_free(self.d_in);
        self.* = .{};
    }
};

/// runSumcheck (:302-354) with prover AND toy verifier on the device. rounds: v × [c0, c1]; challenges: v (= final_point).
pub fn runSumcheck(comptime F: type, allocator: std.mem.Allocator, evaluations: []const F) !?struct { claim: F, rounds: []F, challenges: []F, final_eval: F, result: bool } {
    if (evaluations.len < run_sumcheck_min_entries) return null; // measured crossover: the Zig loop is faster on short tables
    const v: usize = std.math.log2_int(usize, evaluations.len);
    const rounds = try allocator.alloc(F, 2 * v);
    errdefer allocator.free(rounds);
    const challenges = try allocator.alloc(F, v);
    errdefer allocator.free(challenges);
    var claim: F = undefined;
    var final_eval: F = undefined;
    var result: u8 = 0;
    const rc = ffi.zg_run_sumcheck(limbsOf(F, evaluations), evaluations.len, &claim.limbs, @ptrCast(rounds.ptr), @ptrCast(challenges.ptr), &final_eval.limbs, @ptrCast(&result));
    if (rc == ffi.ERR_VERIFY) return Error.SumcheckVerificationFailed;
    if (rc != ffi.OK) return Error.GpuFailure;
    return .{ .claim = claim, .rounds = rounds, .challenges = challenges, .final_eval = final_eval, .result = result != 0 };
}

/// The u128 lookup indices of a LassoProver in device memory (little-endian u128 = two u64 words, the layout zg_sumcheck_bit_* reads).
/// Owned by the prover struct: `uploadLookupIndices` in init, `free` in deinit.
pub const DeviceIndices = struct {
    ptr: ?*anyopaque = null,
    len: usize = 0,
    pub fn free(self: *DeviceIndices) void {
        if (self.ptr != null) _ = ffi.zg_dev_free(self.ptr);
        self.* = .{};
    }
};
pub fn uploadLookupIndices(lookup_indices: []const u128) Error!DeviceIndices {
    var d: DeviceIndices = .{ .len = lookup_indices.len };
    const bytes = @max(lookup_indices.len, 1) * @sizeOf(u128);
    if (ffi.zg_dev_alloc(bytes, &d.ptr) != ffi.OK) return Error.GpuFailure;
    if (lookup_indices.len != 0 and ffi.zg_memcpy_h2d(d.ptr, @ptrCast(lookup_indices.ptr), lookup_indices.len * @sizeOf(u128)) != ffi.OK) {
        d.free();
        return Error.GpuFailure;
    }
    return d;
}

/// LassoProver.computeAddressRoundPoly's two sums (src/zkvm/lasso/prover.zig:283-293): eq_evals split by bit `round_bit` of the
/// u128 lookup indices. Host slices in; provers that run all LOG_K address rounds keep both arrays resident and call
/// ffi.zg_fr_bit_split_sums_dev instead.
pub fn lassoAddressSums(comptime F: type, eq_evals: []const F, lookup_indices: []const u128, round_bit: u32) Error![2]F {
    std.debug.assert(eq_evals.len == lookup_indices.len);
    var s0: F = undefined;
    var s1: F = undefined;
    if (ffi.zg_fr_bit_split_sums(limbsOf(F, eq_evals), @ptrCast(lookup_indices.ptr), eq_evals.len, round_bit, &s0.limbs, &s1.limbs) != ffi.OK) return Error.GpuFailure;
    return .{ s0, s1 };
}
