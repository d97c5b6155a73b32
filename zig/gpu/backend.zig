//! Glue between Zolt's module APIs and libzolt_gpu.so, for Zolt's src/gpu/backend.zig.
//! COMPILE-UNVERIFIED (no Zig toolchain in the build image); Zig >= 0.14. INTEGRATION.md shows where the reference's
//! `pub fn` bodies call into this file; the `pub` signatures of src/msm, src/poly and src/subprotocols do not change.
//!
//! Everything here is generic over the reference's own types, so this file imports nothing from Zolt: `F` is
//! BN254Scalar, `G` is BN254BaseField, `Affine` is msm.AffinePoint(G) — any struct with `.x.limbs`, `.y.limbs`,
//! `.infinity`, `identity()` and `fromCoords(x, y)` (src/msm/mod.zig:15-49).
const std = @import("std");
pub const ffi = @import("ffi.zig");

pub const Error = error{ GpuFailure, OutOfMemory, SumcheckVerificationFailed };

var init_once = std.once(initDevice);
var available: bool = false;

fn initDevice() void {
    // ZOLT_GPU=0 keeps the original Zig bodies; ZOLT_GPU_DEVICE selects the GPU of this process (one process per GPU)
    if (std.posix.getenv("ZOLT_GPU")) |v| {
        if (v.len > 0 and v[0] == '0') return;
    }
    var dev: c_int = 0;
    if (std.posix.getenv("ZOLT_GPU_DEVICE")) |v| dev = std.fmt.parseInt(c_int, v, 10) catch 0;
    available = ffi.zg_init(dev) == ffi.OK;
}

pub fn enabled() bool {
    init_once.call();
    return available;
}

pub fn lastError() []const u8 {
    return std.mem.span(ffi.zg_last_error());
}

fn limbsOf(comptime F: type, s: []const F) [*]const u64 {
    comptime std.debug.assert(@sizeOf(F) == 32); // struct { limbs: [4]u64 }
    return @ptrCast(s.ptr);
}

// ---------------------------------------------------------------------------------------------------------------
// Bases: `[]const Affine` -> device handle, cached by (ptr, len). HyperKZG.commit is called three times per proof with the
// same SetupParams.powers_of_tau_g1 (src/zkvm/mod.zig:1538,1572,1607): the SRS is packed and uploaded once.
// MSM.compute is reached from std.Thread workers (src/msm/mod.zig:637,732), hence the mutex.
// ---------------------------------------------------------------------------------------------------------------
const CacheEntry = struct { ptr: usize, len: usize, handle: ffi.Bases };
var cache_mutex: std.Thread.Mutex = .{};
var cache: [8]CacheEntry = [_]CacheEntry{.{ .ptr = 0, .len = 0, .handle = null }} ** 8;
var cache_next: usize = 0;

/// AffinePoint is an auto-layout struct (field order not ABI-stable, src/msm/mod.zig:19-21): pack x‖y into n×8 u64 + n flags.
pub fn basesHandleFor(comptime Affine: type, bases: []const Affine) Error!ffi.Bases {
    const key_ptr = @intFromPtr(bases.ptr);
    cache_mutex.lock();
    defer cache_mutex.unlock();
    for (cache) |e| {
        if (e.handle != null and e.ptr == key_ptr and e.len == bases.len) return e.handle;
    }
    const a = std.heap.page_allocator;
    const xy = a.alloc(u64, bases.len * 8) catch return Error.OutOfMemory;
    defer a.free(xy);
    const inf = a.alloc(u8, bases.len) catch return Error.OutOfMemory;
    defer a.free(inf);
    for (bases, 0..) |p, i| {
        @memcpy(xy[8 * i .. 8 * i + 4], &p.x.limbs);
        @memcpy(xy[8 * i + 4 .. 8 * i + 8], &p.y.limbs);
        inf[i] = @intFromBool(p.infinity);
    }
    var h: ffi.Bases = null;
    if (ffi.zg_g1_bases_upload(xy.ptr, inf.ptr, bases.len, null, &h) != ffi.OK) return Error.GpuFailure;
    const slot = &cache[cache_next % cache.len];
    if (slot.handle != null) _ = ffi.zg_g1_bases_free(slot.handle); // evict the oldest
    slot.* = .{ .ptr = key_ptr, .len = bases.len, .handle = h };
    cache_next += 1;
    return h;
}

/// Call from SetupParams.deinit (src/poly/commitment/mod.zig:136-140) before the powers are freed.
pub fn forgetBases(ptr: *const anyopaque, len: usize) void {
    cache_mutex.lock();
    defer cache_mutex.unlock();
    for (&cache) |*e| {
        if (e.handle != null and e.ptr == @intFromPtr(ptr) and e.len == len) {
            _ = ffi.zg_g1_bases_free(e.handle);
            e.* = .{ .ptr = 0, .len = 0, .handle = null };
        }
    }
}

fn affineFrom(comptime Affine: type, xy: *const [8]u64, inf: u8) Affine {
    if (inf != 0) return Affine.identity();
    return Affine.fromCoords(.{ .limbs = xy[0..4].* }, .{ .limbs = xy[4..8].* });
}

/// Body of MSM(F, G).compute (src/msm/mod.zig:355-372) on the GPU. `compute` has no error channel: a device failure panics,
/// exactly like an assertion failure of the original would (the caller can still run with ZOLT_GPU=0).
pub fn msmCompute(comptime F: type, comptime Affine: type, bases: []const Affine, scalars: []const F) Affine {
    std.debug.assert(bases.len == scalars.len);
    if (bases.len == 0) return Affine.identity();
    const h = basesHandleFor(Affine, bases) catch @panic("zolt-gpu: bases upload failed");
    var xy: [8]u64 = undefined;
    var inf: u8 = 0;
    if (ffi.zg_msm_g1(h, 0, bases.len, limbsOf(F, scalars), &xy, &inf) != ffi.OK) @panic("zolt-gpu: zg_msm_g1 failed");
    return affineFrom(Affine, &xy, inf);
}

/// Body of BatchMSM.compute / ParallelBatchMSM.compute (src/msm/mod.zig:545-565, 683-748): k scalar vectors over the same
/// bases; short vectors run as ONE fused launch set. The result slice is owned by the caller's allocator, as before.
pub fn msmBatch(comptime F: type, comptime Affine: type, bases: []const Affine, scalar_batches: []const []const F, allocator: std.mem.Allocator) ![]Affine {
    const k = scalar_batches.len;
    const results = try allocator.alloc(Affine, k);
    errdefer allocator.free(results);
    if (k == 0) return results;
    const n = scalar_batches[0].len;
    for (scalar_batches) |b| std.debug.assert(b.len == n and n <= bases.len);
    const h = try basesHandleFor(Affine, bases);
    const ptrs = try allocator.alloc([*]const u64, k);
    defer allocator.free(ptrs);
    for (scalar_batches, 0..) |b, i| ptrs[i] = limbsOf(F, b);
    const xy = try allocator.alloc(u64, 8 * k);
    defer allocator.free(xy);
    const inf = try allocator.alloc(u8, k);
    defer allocator.free(inf);
    if (ffi.zg_msm_g1_batch(h, n, ptrs.ptr, k, xy.ptr, inf.ptr) != ffi.OK) return Error.GpuFailure;
    for (results, 0..) |*r, i| r.* = affineFrom(Affine, xy[8 * i ..][0..8], inf[i]);
    return results;
}

// ---------------------------------------------------------------------------------------------------------------
// poly: EqPolynomial.evalsSliceWithScaling, DensePolynomial.bindLow / bindFirst / evaluate (src/poly/mod.zig)
// ---------------------------------------------------------------------------------------------------------------
pub fn eqTable(comptime F: type, allocator: std.mem.Allocator, r: []const F, scaling_factor: ?F) ![]F {
    const result = try allocator.alloc(F, @as(usize, 1) << @intCast(r.len));
    errdefer allocator.free(result);
    const sc: ?*const [4]u64 = if (scaling_factor) |*s| &s.limbs else null;
    if (ffi.zg_fr_eq_table(limbsOf(F, r), r.len, sc, @ptrCast(result.ptr)) != ffi.OK) return Error.GpuFailure;
    return result;
}

/// in place; the caller then halves its live length / decrements num_vars as the original does (:160-175)
pub fn bindLow(comptime F: type, evaluations: []F, value: F) Error!void {
    if (ffi.zg_fr_bind_low(@ptrCast(evaluations.ptr), evaluations.len, &value.limbs) != ffi.OK) return Error.GpuFailure;
}

/// new allocation of len / 2 entries, like bindFirst (:128-149)
pub fn bindHigh(comptime F: type, allocator: std.mem.Allocator, evaluations: []const F, value: F) ![]F {
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

// ---------------------------------------------------------------------------------------------------------------
// Sumcheck(F).Prover as a device-resident session (src/subprotocols/mod.zig:55-133). The table stays in HBM; per round two
// field elements come back and one challenge goes in, so any host transcript (src/transcripts) keeps working unchanged.
// ---------------------------------------------------------------------------------------------------------------
pub fn SumcheckSession(comptime F: type) type {
    return struct {
        const Self = @This();
        handle: ffi.Session,

        pub fn open(evaluations: []const F, layout: c_int) Error!Self {
            var s: ffi.Session = null;
            if (ffi.zg_sumcheck_open(limbsOf(F, evaluations), evaluations.len, layout, &s) != ffi.OK) return Error.GpuFailure;
            return .{ .handle = s };
        }
        /// nextRound (:69-109): coefficients [g(0), g(1) - g(0)]
        pub fn roundCoeffs(self: *Self) Error![2]F {
            var g0: F = undefined;
            var g1: F = undefined;
            if (ffi.zg_sumcheck_round_sums(self.handle, &g0.limbs, &g1.limbs) != ffi.OK) return Error.GpuFailure;
            return .{ g0, g1.sub(g0) };
        }
        /// receiveChallenge (:112-122)
        pub fn bind(self: *Self, challenge: F) Error!void {
            if (ffi.zg_sumcheck_bind(self.handle, &challenge.limbs) != ffi.OK) return Error.GpuFailure;
        }
        pub fn len(self: *const Self) usize {
            return ffi.zg_sumcheck_len(self.handle);
        }
        /// getFinalEval (:130-133)
        pub fn finalEval(self: *Self) Error!F {
            var out: F = undefined;
            if (ffi.zg_sumcheck_final(self.handle, &out.limbs) != ffi.OK) return Error.GpuFailure;
            return out;
        }
        /// materialise the current table for callers that index prover.polynomial.evaluations directly (:79-92,126,132)
        pub fn read(self: *Self, out: []F) Error!void {
            std.debug.assert(out.len == self.len());
            if (ffi.zg_sumcheck_read(self.handle, @ptrCast(out.ptr)) != ffi.OK) return Error.GpuFailure;
        }
        pub fn close(self: *Self) void {
            _ = ffi.zg_sumcheck_close(self.handle);
            self.handle = null;
        }
    };
}

/// runSumcheck (:302-354) with prover AND toy verifier on the device. rounds: v × [c0, c1]; challenges: v (= final_point).
pub fn runSumcheck(comptime F: type, allocator: std.mem.Allocator, evaluations: []const F) !struct { claim: F, rounds: []F, challenges: []F, final_eval: F, result: bool } {
    const v: usize = std.math.log2_int(usize, evaluations.len);
    const rounds = try allocator.alloc(F, 2 * v);
    errdefer allocator.free(rounds);
    const challenges = try allocator.alloc(F, v);
    errdefer allocator.free(challenges);
    var claim: F = undefined;
    var final_eval: F = undefined;
    var result: u8 = 0;
    const rc = ffi.zg_run_sumcheck(limbsOf(F, evaluations), evaluations.len, &claim.limbs, @ptrCast(rounds.ptr), @ptrCast(challenges.ptr), &final_eval.limbs, &result);
    if (rc == ffi.ERR_VERIFY) return Error.SumcheckVerificationFailed;
    if (rc != ffi.OK) return Error.GpuFailure;
    return .{ .claim = claim, .rounds = rounds, .challenges = challenges, .final_eval = final_eval, .result = result != 0 };
}
