//! extern declarations of libzolt_gpu.so, for Zolt's src/gpu/ffi.zig.
//! GENERATED from include/zolt_gpu.h by tools/gen_zig_ffi.py — do not edit; tests/test_abi_and_host.py holds the two together.
//! COMPILE-UNVERIFIED: the build image has no Zig toolchain (Zig >= 0.14 syntax, build.zig.zon:5 of the reference).
//! The same ABI is exercised for real by zolt_amd/host/zolt_host.hpp (C++) and zolt_amd/lib.py (ctypes).
//!
//! Field elements cross the boundary as they are: BN254Scalar / BN254BaseField are `struct { limbs: [4]u64 }`
//! (src/field/mod.zig:131,583-584), Montgomery form, so `[]const F` is passed as `[*]const u64` via @ptrCast.

pub const Bases = ?*opaque {}; // zg_bases_t
pub const Session = ?*opaque {}; // zg_sc_t
pub const ShardedBases = ?*opaque {}; // zg_sbases_t
pub const ShardedSession = ?*opaque {}; // zg_ssc_t
pub const ProductSession = ?*opaque {}; // zg_psc_t
pub const RegistersSession = ?*opaque {}; // zg_rrw_t
pub const RamRwSession = ?*opaque {}; // zg_rwc_t

pub const MsmConfig = extern struct { window_bits: c_int = 0, precompute_levels: c_int = 0, expected_uses: c_int = 0 };
pub const Column = extern struct { kind: u32 = 0, a: u32 = 0, b: u32 = 0, data: ?*const anyopaque = null, aux: ?*const anyopaque = null }; // zg_col_t
pub const PscTerm = extern struct { n_prod: c_int = 0, prod: [4]c_int = .{ 0, 0, 0, 0 }, n_lin: c_int = 0, lin: [4]c_int = .{ 0, 0, 0, 0 }, lin_coeff: [16]u64 = .{0} ** 16 }; // zg_psc_term

pub const OK: c_int = 0;
pub const ERR_INVALID: c_int = 1;
pub const ERR_HIP: c_int = 2;
pub const ERR_NOMEM: c_int = 3;
pub const ERR_NO_DEVICE: c_int = 4;
pub const ERR_VERIFY: c_int = 5;
pub const FIELD_FR: c_int = 0;
pub const FIELD_FP: c_int = 1;
pub const OP_MUL: c_int = 0;
pub const OP_ADD: c_int = 1;
pub const OP_SUB: c_int = 2;
pub const OP_NEG: c_int = 3;
pub const OP_SQR: c_int = 4;
pub const OP_INV: c_int = 5;
pub const OP_FROM_MONT: c_int = 6;
pub const OP_TO_MONT: c_int = 7;
pub const ABI_MAJOR: u32 = 1;
pub const ABI_MINOR: u32 = 10;
pub const FEATURE_PROTOCOL_SESSIONS: u32 = 1;
pub const FEATURE_RCCL: u32 = 2;
pub const FEATURE_COLUMN_INGEST: u32 = 4;
pub const COL_ZERO: u32 = 0;
pub const COL_U8: u32 = 1;
pub const COL_U32: u32 = 2;
pub const COL_U64: u32 = 3;
pub const COL_I64: u32 = 4;
pub const COL_I128: u32 = 5;
pub const COL_U128: u32 = 6;
pub const COL_FR: u32 = 7;
pub const COL_BIT: u32 = 8;
pub const COL_MUL: u32 = 9;
pub const COL_LUT: u32 = 10;
pub const SC_HIGH_HALF: c_int = 0;
pub const SC_LOW_PAIR: c_int = 1;
pub const PSC_PAIR_SUM: c_int = 256;

pub extern fn zg_abi_version() u32;
pub extern fn zg_abi_features() u32;
pub extern fn zg_init(device: c_int) c_int;
pub extern fn zg_init_devices(n_devices: c_int) c_int;
pub extern fn zg_n_devices() c_int;
pub extern fn zg_shutdown() void;
pub extern fn zg_last_error() [*:0]const u8;
pub extern fn zg_version() [*:0]const u8;
pub extern fn zg_device_count() c_int;
pub extern fn zg_dev_alloc(bytes: usize, dptr: *?*anyopaque) c_int;
pub extern fn zg_dev_free(dptr: ?*anyopaque) c_int;
pub extern fn zg_dev_trim() c_int;
pub extern fn zg_host_alloc(bytes: usize, ptr: *?*anyopaque) c_int;
pub extern fn zg_host_free(ptr: ?*anyopaque) c_int;
pub extern fn zg_dev_memset(dst_dev: ?*anyopaque, byte_value: c_int, bytes: usize) c_int;
pub extern fn zg_memcpy_h2d(dst_dev: ?*anyopaque, src_host: ?*const anyopaque, bytes: usize) c_int;
pub extern fn zg_memcpy_d2h(dst_host: ?*anyopaque, src_dev: ?*const anyopaque, bytes: usize) c_int;
pub extern fn zg_sync() c_int;
pub extern fn zg_field_op(field: c_int, op: c_int, a: ?[*]const u64, b: ?[*]const u64, out: ?[*]u64, n: usize) c_int;
pub extern fn zg_fr_scale(a: ?[*]const u64, n: usize, s: *const [4]u64, out: ?[*]u64) c_int;
pub extern fn zg_g1_bases_upload(xy: ?[*]const u64, inf: ?[*]const u8, n: usize, cfg: ?*const MsmConfig, out: *Bases) c_int;
pub extern fn zg_g1_bases_upload_dev(d_xy: ?[*]const u64, d_inf: ?[*]const u8, n: usize, cfg: ?*const MsmConfig, stream: ?*anyopaque, out: *Bases) c_int;
pub extern fn zg_g1_bases_free(b: Bases) c_int;
pub extern fn zg_g1_bases_len(b: Bases) usize;
pub extern fn zg_g1_bases_plan(b: Bases, window_bits: ?*c_int, windows: ?*c_int, precompute_levels: ?*c_int) c_int;
pub extern fn zg_g1_bases_table_bytes(b: Bases) usize;
pub extern fn zg_msm_g1(b: Bases, off: usize, n: usize, scalars_mont: ?[*]const u64, out_xy: *[8]u64, out_inf: ?[*]u8) c_int;
pub extern fn zg_msm_g1_u64(b: Bases, off: usize, n: usize, values_u64: ?[*]const u64, out_xy: *[8]u64, out_inf: ?[*]u8) c_int;
pub extern fn zg_msm_g1_dev(b: Bases, off: usize, n: usize, d_scalars_mont: ?[*]const u64, stream: ?*anyopaque, out_xy: *[8]u64, out_inf: ?[*]u8) c_int;
pub extern fn zg_msm_g1_dev_async(b: Bases, off: usize, n: usize, d_scalars_mont: ?[*]const u64, stream: ?*anyopaque, d_out_xy: ?[*]u64, d_out_inf: ?[*]u8) c_int;
pub extern fn zg_msm_g1_batch(b: Bases, n: usize, scalar_batches: ?[*]const ?[*]const u64, k: usize, out_xy: ?[*]u64, out_inf: ?[*]u8) c_int;
pub extern fn zg_msm_g1_batch_dev(b: Bases, n: usize, d_scalars_mont: ?[*]const u64, k: usize, stream: ?*anyopaque, d_out9: ?[*]u64) c_int;
pub extern fn zg_msm_g1_partial_dev(b: Bases, off: usize, n: usize, d_scalars_mont: ?[*]const u64, stream: ?*anyopaque, d_out_jac: ?[*]u64) c_int;
pub extern fn zg_msm_g1_partial_fast_dev(b: Bases, off: usize, n: usize, d_scalars_mont: ?[*]const u64, stream: ?*anyopaque, d_out_jac: ?[*]u64) c_int;
pub extern fn zg_g1_combine_partials_dev(d_partials_jac: ?[*]const u64, k: usize, stream: ?*anyopaque, out_xy: *[8]u64, out_inf: ?[*]u8) c_int;
pub extern fn zg_g1_combine_partials_dev_async(d_partials_jac: ?[*]const u64, k: usize, stream: ?*anyopaque, d_out_xy: ?[*]u64, d_out_inf: ?[*]u8) c_int;
pub extern fn zg_g1_combine_partials_batch_dev_async(d_partials_jac: ?[*]const u64, ranks: usize, rank_stride: usize, m: usize, stream: ?*anyopaque, d_out9: ?[*]u64) c_int;
pub extern fn zg_g1_is_on_curve_batch(xy: ?[*]const u64, inf: ?[*]const u8, n: usize, out: ?[*]u8) c_int;
pub extern fn zg_g1_affine_add_batch(a_xy: ?[*]const u64, a_inf: ?[*]const u8, b_xy: ?[*]const u64, b_inf: ?[*]const u8, n: usize, out_xy: ?[*]u64, out_inf: ?[*]u8) c_int;
pub extern fn zg_g1_scalar_mul_batch(xy: ?[*]const u64, inf: ?[*]const u8, scalars_mont: ?[*]const u64, n: usize, out_xy: ?[*]u64, out_inf: ?[*]u8) c_int;
pub extern fn zg_g1_fixed_base_mul_batch(base_xy: *const [8]u64, base_inf: u8, scalars_mont: ?[*]const u64, n: usize, out_xy: ?[*]u64, out_inf: ?[*]u8) c_int;
pub extern fn zg_hyperkzg_setup(base_xy: *const [8]u64, tau: *const [4]u64, n: usize, cfg: ?*const MsmConfig, out_xy: ?[*]u64, out_inf: ?[*]u8, out: *Bases) c_int;
pub extern fn zg_hyperkzg_open(srs: Bases, evals: ?[*]const u64, n_evals: usize, point: ?[*]const u64, num_vars: usize, value: *const [4]u64, q_xy: ?[*]u64, q_inf: ?[*]u8, final_eval: *[4]u64) c_int;
pub extern fn zg_hyperkzg_open_dev(srs: Bases, d_evals: ?[*]const u64, n_evals: usize, point: ?[*]const u64, num_vars: usize, value: *const [4]u64, stream: ?*anyopaque, q_xy: ?[*]u64, q_inf: ?[*]u8, final_eval: *[4]u64) c_int;
pub extern fn zg_hyperkzg_batch_open(srs: Bases, polys: ?[*]const ?[*]const u64, lens: ?[*]const usize, k: usize, point: ?[*]const u64, num_vars: usize, q_xy: ?[*]u64, q_inf: ?[*]u8, n_quot: ?*usize, evaluations: ?[*]u64, final_eval: *[4]u64, gamma: *[4]u64) c_int;
pub extern fn zg_fr_eq_table(r: ?[*]const u64, v: usize, scale: ?[*]const u64, out: ?[*]u64) c_int;
pub extern fn zg_fr_eq_table_dev(r_host: ?[*]const u64, v: usize, scale_host: ?[*]const u64, d_out: ?[*]u64, stream: ?*anyopaque) c_int;
pub extern fn zg_fr_eq_plus_one_table(r: ?[*]const u64, v: usize, out: ?[*]u64) c_int;
pub extern fn zg_fr_eq_plus_one_table_dev(r_host: ?[*]const u64, v: usize, d_out: ?[*]u64, stream: ?*anyopaque) c_int;
pub extern fn zg_fr_eq_prefix_tables(tau: ?[*]const u64, v: usize, out: ?[*]u64) c_int;
pub extern fn zg_fr_eq_prefix_tables_dev(tau_host: ?[*]const u64, v: usize, d_out: ?[*]u64, stream: ?*anyopaque) c_int;
pub extern fn zg_fr_dense_evaluate(evals: ?[*]const u64, num_vars: usize, point: ?[*]const u64, out: *[4]u64) c_int;
pub extern fn zg_fr_rows_mle(rows: ?[*]const u64, n_rows: usize, k: usize, r: ?[*]const u64, v: usize, out: ?[*]u64) c_int;
pub extern fn zg_fr_rows_mle_dev(d_rows: ?[*]const u64, n_rows: usize, k: usize, r_host: ?[*]const u64, v: usize, stream: ?*anyopaque, out: ?[*]u64) c_int;
pub extern fn zg_fr_rows_affine(rows: ?[*]const u64, n_rows: usize, k: usize, stride: usize, coeffs: ?[*]const u64, ntab: usize, g: usize, n_pad: usize, tables: ?[*]const ?[*]u64) c_int;
pub extern fn zg_fr_rows_affine_dev(d_rows: ?[*]const u64, n_rows: usize, k: usize, stride: usize, coeffs_host: ?[*]const u64, ntab: usize, g: usize, n_pad: usize, d_tables: ?[*]const ?[*]u64, stream: ?*anyopaque) c_int;
pub extern fn zg_fr_rows_affine_records_dev(d_rows: ?[*]const u64, n_rows: usize, k: usize, stride: usize, coeffs_host: ?[*]const u64, nout: usize, record: usize, first: usize, d_out: ?[*]u64, stream: ?*anyopaque) c_int;
pub extern fn zg_fr_rows_from_columns(cols: ?[*]const Column, n_cols: usize, n_rows: usize, d_rows: ?[*]u64) c_int;
pub extern fn zg_fr_rows_from_columns_dev(cols: ?[*]const Column, n_cols: usize, n_rows: usize, d_rows: ?[*]u64, stream: ?*anyopaque) c_int;
pub extern fn zg_fr_lt_table(r: ?[*]const u64, v: usize, out: ?[*]u64) c_int;
pub extern fn zg_fr_lt_table_dev(r_host: ?[*]const u64, v: usize, d_out: ?[*]u64, stream: ?*anyopaque) c_int;
pub extern fn zg_fr_write_tables_dev(n: usize, m: usize, cycle: ?[*]const u32, word: ?[*]const u32, pre: ?[*]const u64, post: ?[*]const u64, r_eq: ?[*]const u64, log_k: usize, d_inc: ?[*]u64, d_wa: ?[*]u64, stream: ?*anyopaque) c_int;
pub extern fn zg_fr_weighted_colsum(table: ?[*]const u64, rows: usize, cols: usize, weights: ?[*]const u64, m: usize, out: ?[*]u64) c_int;
pub extern fn zg_fr_weighted_colsum_dev(d_table: ?[*]const u64, rows: usize, cols: usize, d_weights: ?[*]const u64, m: usize, d_out: ?[*]u64, stream: ?*anyopaque) c_int;
pub extern fn zg_fr_rows_affine_prodsum_dev(d_rows: ?[*]const u64, n_rows: usize, k: usize, stride: usize, coeffs_host: ?[*]const u64, npairs: usize, d_weights: ?[*]const u64, g: usize, out: ?[*]u64, stream: ?*anyopaque) c_int;
pub extern fn zg_fr_bind_low(table: ?[*]u64, len: usize, r: *const [4]u64) c_int;
pub extern fn zg_fr_bind_high(table: ?[*]const u64, len: usize, r: *const [4]u64, out: ?[*]u64) c_int;
pub extern fn zg_fr_spartan_combine(eq: ?[*]const u64, az: ?[*]const u64, bz: ?[*]const u64, cz: ?[*]const u64, n: usize, out: ?[*]u64) c_int;
pub extern fn zg_fr_spartan_combine_dev(d_eq: ?[*]const u64, d_az: ?[*]const u64, d_bz: ?[*]const u64, d_cz: ?[*]const u64, n: usize, d_out: ?[*]u64, stream: ?*anyopaque) c_int;
pub extern fn zg_sumcheck_open(evals: ?[*]const u64, len: usize, layout: c_int, s: *Session) c_int;
pub extern fn zg_sumcheck_open_dev(d_evals: ?[*]const u64, len: usize, layout: c_int, stream: ?*anyopaque, s: *Session) c_int;
pub extern fn zg_sumcheck_open_dev_borrowed(d_evals: ?[*]const u64, len: usize, layout: c_int, stream: ?*anyopaque, out: *Session) c_int;
pub extern fn zg_sumcheck_open_column(col: ?[*]const Column, n_rows: usize, len: usize, layout: c_int, s: *Session) c_int;
pub extern fn zg_sumcheck_open_spartan_dev(r: ?[*]const u64, v: usize, scale: ?[*]const u64, d_az: ?[*]const u64, d_bz: ?[*]const u64, d_cz: ?[*]const u64, layout: c_int, stream: ?*anyopaque, s: *Session) c_int;
pub extern fn zg_sumcheck_round_sums(s: Session, g0: *[4]u64, g1: *[4]u64) c_int;
pub extern fn zg_sumcheck_bind(s: Session, r: *const [4]u64) c_int;
pub extern fn zg_sumcheck_len(s: Session) usize;
pub extern fn zg_sumcheck_final(s: Session, out: *[4]u64) c_int;
pub extern fn zg_sumcheck_read(s: Session, out_table: ?[*]u64) c_int;
pub extern fn zg_sumcheck_gather(s: Session, idx: ?[*]const u64, n: usize, out: ?[*]u64) c_int;
pub extern fn zg_sumcheck_round_sums_dev(s: Session, d_out8: ?[*]u64) c_int;
pub extern fn zg_sumcheck_read_dev(s: Session, d_out_table: ?[*]u64) c_int;
pub extern fn zg_sumcheck_close(s: Session) c_int;
pub extern fn zg_sumcheck_raf_round(s: Session, base: *const [4]u64, current_power: u64, s0: *[4]u64, s2: *[4]u64) c_int;
pub extern fn zg_sumcheck_raf_claim(s: Session, base: u64, step: u64, claim: *[4]u64) c_int;
pub extern fn zg_sumcheck_bit_round(s: Session, d_idx128: ?[*]const u64, n_idx: usize, bit: c_uint, sum0: *[4]u64, sum1: *[4]u64) c_int;
pub extern fn zg_sumcheck_bit_bind(s: Session, d_idx128: ?[*]const u64, n_idx: usize, bit: c_uint, r: *const [4]u64, claim: *[4]u64) c_int;
pub extern fn zg_fr_bit_split_sums(vals: ?[*]const u64, idx128: ?[*]const u64, n: usize, bit: c_uint, sum0: *[4]u64, sum1: *[4]u64) c_int;
pub extern fn zg_fr_bit_split_sums_dev(d_vals: ?[*]const u64, d_idx128: ?[*]const u64, n: usize, bit: c_uint, stream: ?*anyopaque, sum0: *[4]u64, sum1: *[4]u64) c_int;
pub extern fn zg_selftest_handoff(blocks: c_uint, threads: c_uint, iters: c_uint, busy: c_int, mismatches: ?[*]u64, completed: ?[*]u64) c_int;
pub extern fn zg_run_sumcheck_dev(d_evals: ?[*]const u64, len: usize, stream: ?*anyopaque, claim: *[4]u64, rounds: ?[*]u64, challenges: ?[*]u64, final_eval: *[4]u64, result: ?[*]u8) c_int;
pub extern fn zg_run_sumcheck(evals: ?[*]const u64, len: usize, claim: *[4]u64, rounds: ?[*]u64, challenges: ?[*]u64, final_eval: *[4]u64, result: ?[*]u8) c_int;
pub extern fn zg_psc_open(tables: ?[*]const ?[*]const u64, k: usize, len: usize, s: *ProductSession) c_int;
pub extern fn zg_psc_open_dev(d_tables: ?[*]const ?[*]const u64, k: usize, len: usize, stream: ?*anyopaque, s: *ProductSession) c_int;
pub extern fn zg_psc_len(s: ProductSession) usize;
pub extern fn zg_psc_tables(s: ProductSession) usize;
pub extern fn zg_psc_round_evals(s: ProductSession, prod_idx: ?[*]const c_int, p: usize, lin_idx: ?[*]const c_int, lin_coeff: ?[*]const u64, q: usize, out: *[16]u64) c_int;
pub extern fn zg_psc_round_expr(s: ProductSession, terms: ?[*]const PscTerm, n_terms: usize, out: *[16]u64) c_int;
pub extern fn zg_psc_set_points(s: ProductSession, points: c_uint) c_int;
pub extern fn zg_psc_round_gruen(s: ProductSession, prod_idx: ?[*]const c_int, p: usize, d_e_out: ?[*]const u64, n_out: usize, d_e_in: ?[*]const u64, n_in: usize, t0: *[4]u64, t_inf: *[4]u64) c_int;
pub extern fn zg_psc_bind(s: ProductSession, r: *const [4]u64) c_int;
pub extern fn zg_psc_read(s: ProductSession, table: usize, out: ?[*]u64) c_int;
pub extern fn zg_psc_table_dev(s: ProductSession, table: usize, d_ptr: *?[*]const u64) c_int;
pub extern fn zg_psc_gather(s: ProductSession, table: usize, idx: ?[*]const u64, n: usize, out: ?[*]u64) c_int;
pub extern fn zg_psc_final(s: ProductSession, out: ?[*]u64) c_int;
pub extern fn zg_psc_close(s: ProductSession) c_int;
pub extern fn zg_rrw_open(log_t: usize, rs1: ?[*]const u8, rs2: ?[*]const u8, rd: ?[*]const u8, reg_vals: ?[*]const u64, inc: ?[*]const u64, gamma: *const [4]u64, s: *RegistersSession) c_int;
pub extern fn zg_rrw_open_trace(log_t: usize, rs1: ?[*]const u8, rs2: ?[*]const u8, rd: ?[*]const u8, rd_value: ?[*]const u64, gamma: *const [4]u64, s: *RegistersSession) c_int;
pub extern fn zg_rrw_cycles(s: RegistersSession) usize;
pub extern fn zg_rrw_registers(s: RegistersSession) usize;
pub extern fn zg_rrw_round_cycle_gruen(s: RegistersSession, d_e_out: ?[*]const u64, n_out: usize, d_e_in: ?[*]const u64, n_in: usize, q0: *[4]u64, qx2: *[4]u64) c_int;
pub extern fn zg_rrw_set_eq(s: RegistersSession, eq: ?[*]const u64, n: usize) c_int;
pub extern fn zg_rrw_round_address(s: RegistersSession, e0: *[4]u64, e1: ?[*]u64, e2: *[4]u64) c_int;
pub extern fn zg_rrw_round_cycle(s: RegistersSession, e0: *[4]u64, e1: ?[*]u64, e2: *[4]u64, e3: *[4]u64) c_int;
pub extern fn zg_rrw_bind_cycle(s: RegistersSession, r: *const [4]u64) c_int;
pub extern fn zg_rrw_bind_address(s: RegistersSession, r: *const [4]u64) c_int;
pub extern fn zg_rrw_final(s: RegistersSession, out: ?[*]u64) c_int;
pub extern fn zg_rrw_close(s: RegistersSession) c_int;
pub extern fn zg_rwc_open(log_k: usize, log_t: usize, n: usize, cycle: ?[*]const u32, address: ?[*]const u32, val_coeff: ?[*]const u64, prev_val: ?[*]const u64, next_val: ?[*]const u64, inc: ?[*]const u64, val_init: ?[*]const u64, r_cycle: ?[*]const u64, s: *RamRwSession) c_int;
pub extern fn zg_rwc_open_writes(log_k: usize, log_t: usize, n: usize, cycle: ?[*]const u32, address: ?[*]const u32, val_coeff: ?[*]const u64, prev_val: ?[*]const u64, next_val: ?[*]const u64, is_write: ?[*]const u8, val_init: ?[*]const u64, r_cycle: ?[*]const u64, s: *RamRwSession) c_int;
pub extern fn zg_rwc_entries(s: RamRwSession) usize;
pub extern fn zg_rwc_cycles(s: RamRwSession) usize;
pub extern fn zg_rwc_round_cycle(s: RamRwSession, d_e_out: ?[*]const u64, n_out: usize, d_e_in: ?[*]const u64, n_in: usize, gamma: *const [4]u64, q_constant: *[4]u64, q_quadratic: *[4]u64) c_int;
pub extern fn zg_rwc_bind_cycle(s: RamRwSession, r: *const [4]u64) c_int;
pub extern fn zg_rwc_round_address(s: RamRwSession, addr_round: usize, challenges: ?[*]const u64, gamma: *const [4]u64, s0: *[4]u64, s2: *[4]u64) c_int;
pub extern fn zg_rwc_bind_address(s: RamRwSession, addr_round: usize, r: *const [4]u64) c_int;
pub extern fn zg_rwc_opening(s: RamRwSession, r_address: ?[*]const u64, r_cycle: ?[*]const u64, out: *[12]u64) c_int;
pub extern fn zg_rwc_cycle_scalars(s: RamRwSession, eq0: *[4]u64, inc0: *[4]u64) c_int;
pub extern fn zg_rwc_read_entries(s: RamRwSession, cycle: ?[*]u32, address: ?[*]u32, ra_coeff: ?[*]u64, val_coeff: ?[*]u64, prev_val: ?[*]u64, next_val: ?[*]u64) c_int;
pub extern fn zg_rwc_close(s: RamRwSession) c_int;
pub extern fn zg_shard_bounds(n: usize, shards: c_int, shard: c_int, start: ?*usize, len: ?*usize) c_int;
pub extern fn zg_g1_bases_upload_sharded(xy: ?[*]const u64, inf: ?[*]const u8, n: usize, cfg: ?*const MsmConfig, out: *ShardedBases) c_int;
pub extern fn zg_g1_sbases_free(sb: ShardedBases) c_int;
pub extern fn zg_g1_sbases_len(sb: ShardedBases) usize;
pub extern fn zg_g1_sbases_shards(sb: ShardedBases) c_int;
pub extern fn zg_g1_sbases_exchange(sb: ShardedBases) c_int;
pub extern fn zg_g1_sbases_shard(sb: ShardedBases, shard: c_int, device: ?*c_int, start: ?*usize, len: ?*usize) c_int;
pub extern fn zg_msm_g1_sharded(sb: ShardedBases, n: usize, scalars_mont: ?[*]const u64, out_xy: *[8]u64, out_inf: ?[*]u8) c_int;
pub extern fn zg_msm_g1_sharded_dev(sb: ShardedBases, n: usize, d_scalars_per_shard: ?[*]const ?[*]const u64, out_xy: *[8]u64, out_inf: ?[*]u8) c_int;
pub extern fn zg_msm_g1_batch_sharded(sb: ShardedBases, n: usize, scalar_batches: ?[*]const ?[*]const u64, k: usize, out_xy: ?[*]u64, out_inf: ?[*]u8) c_int;
pub extern fn zg_msm_g1_sharded_dev_async(sb: ShardedBases, n: usize, d_scalars_per_shard: ?[*]const ?[*]const u64, ready_streams: ?[*]const ?*anyopaque, ticket: ?[*]u64) c_int;
pub extern fn zg_msm_g1_batch_sharded_async(sb: ShardedBases, n: usize, scalar_batches: ?[*]const ?[*]const u64, k: usize, ticket: ?[*]u64) c_int;
pub extern fn zg_sharded_wait(sb: ShardedBases, ticket: u64, out_xy: ?[*]u64, out_inf: ?[*]u8) c_int;
pub extern fn zg_g1_sbases_inflight(sb: ShardedBases) c_int;
pub extern fn zg_sumcheck_open_sharded(evals: ?[*]const u64, len: usize, layout: c_int, s: *ShardedSession) c_int;
pub extern fn zg_sumcheck_shards(s: ShardedSession) c_int;
pub extern fn zg_sumcheck_len_sharded(s: ShardedSession) usize;
pub extern fn zg_sumcheck_round_sums_sharded(s: ShardedSession, g0: *[4]u64, g1: *[4]u64) c_int;
pub extern fn zg_sumcheck_bind_sharded(s: ShardedSession, r: *const [4]u64) c_int;
pub extern fn zg_sumcheck_final_sharded(s: ShardedSession, out: *[4]u64) c_int;
pub extern fn zg_sumcheck_close_sharded(s: ShardedSession) c_int;
