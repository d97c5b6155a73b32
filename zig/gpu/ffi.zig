//! extern declarations of libzolt_gpu.so (include/zolt_gpu.h), for Zolt's src/gpu/ffi.zig.
//! COMPILE-UNVERIFIED: the build image has no Zig toolchain (Zig >= 0.14 syntax, build.zig.zon:5 of the reference).
//! The same ABI is exercised for real by zolt_amd/host/zolt_host.hpp (C++) and zolt_amd/lib.py (ctypes).
//!
//! Field elements cross the boundary as they are: BN254Scalar / BN254BaseField are `struct { limbs: [4]u64 }`
//! (src/field/mod.zig:131,583-584), Montgomery form, so `[]const F` is passed as `[*]const u64` via @ptrCast.

pub const Bases = ?*opaque {};
pub const Session = ?*opaque {};

pub const MsmConfig = extern struct { window_bits: c_int = 0, precompute_levels: c_int = 0 };

pub const OK: c_int = 0;
pub const ERR_VERIFY: c_int = 5; // zg_run_sumcheck: the toy verifier rejected (error.SumcheckVerificationFailed)
pub const SC_HIGH_HALF: c_int = 0; // DensePolynomial.bindFirst layout (src/poly/mod.zig:128-149)
pub const SC_LOW_PAIR: c_int = 1; // DensePolynomial.bindLow layout (src/poly/mod.zig:160-175)

pub extern fn zg_init(device: c_int) c_int;
pub extern fn zg_shutdown() void;
pub extern fn zg_last_error() [*:0]const u8;
pub extern fn zg_device_count() c_int;

// bases (the SRS): uploaded once, resident for the run (SetupParams.powers_of_tau_g1, src/poly/commitment/mod.zig:122-140)
pub extern fn zg_g1_bases_upload(xy: [*]const u64, inf: ?[*]const u8, n: usize, cfg: ?*const MsmConfig, out: *Bases) c_int;
pub extern fn zg_g1_bases_free(b: Bases) c_int;
pub extern fn zg_g1_bases_len(b: Bases) usize;
pub extern fn zg_g1_bases_plan(b: Bases, window_bits: ?*c_int, windows: ?*c_int, precompute_levels: ?*c_int) c_int;

// MSM(F,G).compute / BatchMSM.compute / scalarMul loops / curve checks (src/msm/mod.zig:355-565)
pub extern fn zg_msm_g1(b: Bases, off: usize, n: usize, scalars: [*]const u64, out_xy: *[8]u64, out_inf: *u8) c_int;
pub extern fn zg_msm_g1_batch(b: Bases, n: usize, batches: [*]const [*]const u64, k: usize, out_xy: [*]u64, out_inf: [*]u8) c_int;
pub extern fn zg_g1_scalar_mul_batch(xy: [*]const u64, inf: ?[*]const u8, scalars: [*]const u64, n: usize, out_xy: [*]u64, out_inf: [*]u8) c_int;
pub extern fn zg_g1_is_on_curve_batch(xy: [*]const u64, inf: ?[*]const u8, n: usize, out: [*]u8) c_int;

// HyperKZG.open / batchOpen (src/poly/commitment/mod.zig:261-324, 607-732)
pub extern fn zg_hyperkzg_open(srs: Bases, evals: [*]const u64, n: usize, point: [*]const u64, num_vars: usize, value: *const [4]u64, q_xy: [*]u64, q_inf: [*]u8, final_eval: *[4]u64) c_int;

// poly (src/poly/mod.zig:73-92, 128-175, 252-290) and Spartan's combine (src/zkvm/spartan/mod.zig:191-199)
pub extern fn zg_fr_dense_evaluate(evals: [*]const u64, num_vars: usize, point: [*]const u64, out: *[4]u64) c_int;
pub extern fn zg_fr_eq_table(r: [*]const u64, v: usize, scale: ?*const [4]u64, out: [*]u64) c_int;
pub extern fn zg_fr_bind_low(table: [*]u64, len: usize, r: *const [4]u64) c_int;
pub extern fn zg_fr_bind_high(table: [*]const u64, len: usize, r: *const [4]u64, out: [*]u64) c_int;
pub extern fn zg_fr_spartan_combine(eq: [*]const u64, az: [*]const u64, bz: [*]const u64, cz: [*]const u64, n: usize, out: [*]u64) c_int;

// Sumcheck(F).Prover as a device-resident session; runSumcheck with the toy verifier (src/subprotocols/mod.zig:55-133, 302-354)
pub extern fn zg_sumcheck_open(evals: [*]const u64, len: usize, layout: c_int, s: *Session) c_int;
pub extern fn zg_sumcheck_round_sums(s: Session, g0: *[4]u64, g1: *[4]u64) c_int;
pub extern fn zg_sumcheck_bind(s: Session, r: *const [4]u64) c_int;
pub extern fn zg_sumcheck_len(s: Session) usize;
pub extern fn zg_sumcheck_final(s: Session, out: *[4]u64) c_int;
pub extern fn zg_sumcheck_read(s: Session, out_table: [*]u64) c_int;
pub extern fn zg_sumcheck_close(s: Session) c_int;
pub extern fn zg_run_sumcheck(evals: [*]const u64, len: usize, claim: *[4]u64, rounds: [*]u64, challenges: [*]u64, final_eval: *[4]u64, result: *u8) c_int;
