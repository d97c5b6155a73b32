"""GENERATED from include/zolt_gpu.h and include/zolt_gpu_internal.h by tools/gen_bindings.py — do not edit.
tests/test_abi_and_host.py re-runs the generator and compares. zolt_amd/lib.py applies these signatures to the loaded library."""
from ctypes import c_char_p, c_double, c_int, c_size_t, c_uint, c_uint8, c_uint32, c_uint64, c_void_p  # noqa: F401

ZG_OK = 0
ZG_ERR_INVALID = 1
ZG_ERR_HIP = 2
ZG_ERR_NOMEM = 3
ZG_ERR_NO_DEVICE = 4
ZG_ERR_VERIFY = 5
ZG_FIELD_FR = 0
ZG_FIELD_FP = 1
ZG_OP_MUL = 0
ZG_OP_ADD = 1
ZG_OP_SUB = 2
ZG_OP_NEG = 3
ZG_OP_SQR = 4
ZG_OP_INV = 5
ZG_OP_FROM_MONT = 6
ZG_OP_TO_MONT = 7
ZG_ABI_MAJOR = 1
ZG_ABI_MINOR = 10
ZG_FEATURE_PROTOCOL_SESSIONS = 1
ZG_FEATURE_RCCL = 2
ZG_FEATURE_COLUMN_INGEST = 4
ZG_COL_ZERO = 0
ZG_COL_U8 = 1
ZG_COL_U32 = 2
ZG_COL_U64 = 3
ZG_COL_I64 = 4
ZG_COL_I128 = 5
ZG_COL_U128 = 6
ZG_COL_FR = 7
ZG_COL_BIT = 8
ZG_COL_MUL = 9
ZG_COL_LUT = 10
ZG_SC_HIGH_HALF = 0
ZG_SC_LOW_PAIR = 1
ZG_PSC_PAIR_SUM = 256
ZG_OP_MUL29 = 9
ZG_OP_SQR29 = 10
ZG_OP_X3_29 = 11
ZG_OP_INV_XGCD = 12
ZG_OP_INV_SAFEGCD = 13
ZG_PROF_MSM_DIGITS = 0
ZG_PROF_MSM_SORT = 1
ZG_PROF_MSM_ACCUMULATE = 2
ZG_PROF_MSM_REDUCE = 3
ZG_PROF_EQ_TABLE = 4
ZG_PROF_SC_FOLD = 5
ZG_PROF_SC_SUMS = 6
ZG_PROF_COMBINE = 7
ZG_PROF_NKERNELS = 8

PROTOS = {
    "zg_abi_version": (c_uint32, []),  # 
    "zg_abi_features": (c_uint32, []),  # 
    "zg_init": (c_int, [c_int]),  # device
    "zg_init_devices": (c_int, [c_int]),  # n_devices
    "zg_n_devices": (c_int, []),  # 
    "zg_shutdown": (None, []),  # 
    "zg_last_error": (c_char_p, []),  # 
    "zg_version": (c_char_p, []),  # 
    "zg_device_count": (c_int, []),  # 
    "zg_dev_alloc": (c_int, [c_size_t, c_void_p]),  # bytes, dptr
    "zg_dev_free": (c_int, [c_void_p]),  # dptr
    "zg_dev_trim": (c_int, []),  # 
    "zg_host_alloc": (c_int, [c_size_t, c_void_p]),  # bytes, ptr
    "zg_host_free": (c_int, [c_void_p]),  # ptr
    "zg_dev_memset": (c_int, [c_void_p, c_int, c_size_t]),  # dst_dev, byte_value, bytes
    "zg_memcpy_h2d": (c_int, [c_void_p, c_void_p, c_size_t]),  # dst_dev, src_host, bytes
    "zg_memcpy_d2h": (c_int, [c_void_p, c_void_p, c_size_t]),  # dst_host, src_dev, bytes
    "zg_sync": (c_int, []),  # 
    "zg_field_op": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t]),  # field, op, a, b, out, n
    "zg_fr_scale": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p]),  # a, n, s, out
    "zg_g1_bases_upload": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),  # xy, inf, n, cfg, out
    "zg_g1_bases_upload_dev": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),  # d_xy, d_inf, n, cfg, stream, out
    "zg_g1_bases_free": (c_int, [c_void_p]),  # b
    "zg_g1_bases_len": (c_size_t, [c_void_p]),  # b
    "zg_g1_bases_plan": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),  # b, window_bits, windows, precompute_levels
    "zg_g1_bases_table_bytes": (c_size_t, [c_void_p]),  # b
    "zg_msm_g1": (c_int, [c_void_p, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p]),  # b, off, n, scalars_mont, out_xy, out_inf
    "zg_msm_g1_u64": (c_int, [c_void_p, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p]),  # b, off, n, values_u64, out_xy, out_inf
    "zg_msm_g1_dev": (c_int, [c_void_p, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),  # b, off, n, d_scalars_mont, stream, out_xy, out_inf
    "zg_msm_g1_dev_async": (c_int, [c_void_p, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),  # b, off, n, d_scalars_mont, stream, d_out_xy, d_out_inf
    "zg_msm_g1_batch": (c_int, [c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p]),  # b, n, scalar_batches, k, out_xy, out_inf
    "zg_msm_g1_batch_dev": (c_int, [c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p]),  # b, n, d_scalars_mont, k, stream, d_out9
    "zg_msm_g1_partial_dev": (c_int, [c_void_p, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p]),  # b, off, n, d_scalars_mont, stream, d_out_jac
    "zg_msm_g1_partial_fast_dev": (c_int, [c_void_p, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p]),  # b, off, n, d_scalars_mont, stream, d_out_jac
    "zg_g1_combine_partials_dev": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),  # d_partials_jac, k, stream, out_xy, out_inf
    "zg_g1_combine_partials_dev_async": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),  # d_partials_jac, k, stream, d_out_xy, d_out_inf
    "zg_g1_combine_partials_batch_dev_async": (c_int, [c_void_p, c_size_t, c_size_t, c_size_t, c_void_p, c_void_p]),  # d_partials_jac, ranks, rank_stride, m, stream, d_out9
    "zg_g1_is_on_curve_batch": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),  # xy, inf, n, out
    "zg_g1_affine_add_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),  # a_xy, a_inf, b_xy, b_inf, n, out_xy, out_inf
    "zg_g1_scalar_mul_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),  # xy, inf, scalars_mont, n, out_xy, out_inf
    "zg_g1_fixed_base_mul_batch": (c_int, [c_void_p, c_uint8, c_void_p, c_size_t, c_void_p, c_void_p]),  # base_xy, base_inf, scalars_mont, n, out_xy, out_inf
    "zg_hyperkzg_setup": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),  # base_xy, tau, n, cfg, out_xy, out_inf, out
    "zg_hyperkzg_open": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),  # srs, evals, n_evals, point, num_vars, value, q_xy, q_inf, final_eval
    "zg_hyperkzg_open_dev": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),  # srs, d_evals, n_evals, point, num_vars, value, stream, q_xy, q_inf, final_eval
    "zg_hyperkzg_batch_open": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),  # srs, polys, lens, k, point, num_vars, q_xy, q_inf, n_quot, evaluations, final_eval, gamma
    "zg_fr_eq_table": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p]),  # r, v, scale, out
    "zg_fr_eq_table_dev": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),  # r_host, v, scale_host, d_out, stream
    "zg_fr_eq_plus_one_table": (c_int, [c_void_p, c_size_t, c_void_p]),  # r, v, out
    "zg_fr_eq_plus_one_table_dev": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p]),  # r_host, v, d_out, stream
    "zg_fr_eq_prefix_tables": (c_int, [c_void_p, c_size_t, c_void_p]),  # tau, v, out
    "zg_fr_eq_prefix_tables_dev": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p]),  # tau_host, v, d_out, stream
    "zg_fr_dense_evaluate": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p]),  # evals, num_vars, point, out
    "zg_fr_rows_mle": (c_int, [c_void_p, c_size_t, c_size_t, c_void_p, c_size_t, c_void_p]),  # rows, n_rows, k, r, v, out
    "zg_fr_rows_mle_dev": (c_int, [c_void_p, c_size_t, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p]),  # d_rows, n_rows, k, r_host, v, stream, out
    "zg_fr_rows_affine": (c_int, [c_void_p, c_size_t, c_size_t, c_size_t, c_void_p, c_size_t, c_size_t, c_size_t, c_void_p]),  # rows, n_rows, k, stride, coeffs, ntab, g, n_pad, tables
    "zg_fr_rows_affine_dev": (c_int, [c_void_p, c_size_t, c_size_t, c_size_t, c_void_p, c_size_t, c_size_t, c_size_t, c_void_p, c_void_p]),  # d_rows, n_rows, k, stride, coeffs_host, ntab, g, n_pad, d_tables, stream
    "zg_fr_rows_affine_records_dev": (c_int, [c_void_p, c_size_t, c_size_t, c_size_t, c_void_p, c_size_t, c_size_t, c_size_t, c_void_p, c_void_p]),  # d_rows, n_rows, k, stride, coeffs_host, nout, record, first, d_out, stream
    "zg_fr_rows_from_columns": (c_int, [c_void_p, c_size_t, c_size_t, c_void_p]),  # cols, n_cols, n_rows, d_rows
    "zg_fr_rows_from_columns_dev": (c_int, [c_void_p, c_size_t, c_size_t, c_void_p, c_void_p]),  # cols, n_cols, n_rows, d_rows, stream
    "zg_fr_lt_table": (c_int, [c_void_p, c_size_t, c_void_p]),  # r, v, out
    "zg_fr_lt_table_dev": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p]),  # r_host, v, d_out, stream
    "zg_fr_write_tables_dev": (c_int, [c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),  # n, m, cycle, word, pre, post, r_eq, log_k, d_inc, d_wa, stream
    "zg_fr_weighted_colsum": (c_int, [c_void_p, c_size_t, c_size_t, c_void_p, c_size_t, c_void_p]),  # table, rows, cols, weights, m, out
    "zg_fr_weighted_colsum_dev": (c_int, [c_void_p, c_size_t, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p]),  # d_table, rows, cols, d_weights, m, d_out, stream
    "zg_fr_rows_affine_prodsum_dev": (c_int, [c_void_p, c_size_t, c_size_t, c_size_t, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p]),  # d_rows, n_rows, k, stride, coeffs_host, npairs, d_weights, g, out, stream
    "zg_fr_bind_low": (c_int, [c_void_p, c_size_t, c_void_p]),  # table, len, r
    "zg_fr_bind_high": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p]),  # table, len, r, out
    "zg_fr_spartan_combine": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),  # eq, az, bz, cz, n, out
    "zg_fr_spartan_combine_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),  # d_eq, d_az, d_bz, d_cz, n, d_out, stream
    "zg_sumcheck_open": (c_int, [c_void_p, c_size_t, c_int, c_void_p]),  # evals, len, layout, s
    "zg_sumcheck_open_dev": (c_int, [c_void_p, c_size_t, c_int, c_void_p, c_void_p]),  # d_evals, len, layout, stream, s
    "zg_sumcheck_open_dev_borrowed": (c_int, [c_void_p, c_size_t, c_int, c_void_p, c_void_p]),  # d_evals, len, layout, stream, out
    "zg_sumcheck_open_column": (c_int, [c_void_p, c_size_t, c_size_t, c_int, c_void_p]),  # col, n_rows, len, layout, s
    "zg_sumcheck_open_spartan_dev": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),  # r, v, scale, d_az, d_bz, d_cz, layout, stream, s
    "zg_sumcheck_round_sums": (c_int, [c_void_p, c_void_p, c_void_p]),  # s, g0, g1
    "zg_sumcheck_bind": (c_int, [c_void_p, c_void_p]),  # s, r
    "zg_sumcheck_len": (c_size_t, [c_void_p]),  # s
    "zg_sumcheck_final": (c_int, [c_void_p, c_void_p]),  # s, out
    "zg_sumcheck_read": (c_int, [c_void_p, c_void_p]),  # s, out_table
    "zg_sumcheck_gather": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),  # s, idx, n, out
    "zg_sumcheck_round_sums_dev": (c_int, [c_void_p, c_void_p]),  # s, d_out8
    "zg_sumcheck_read_dev": (c_int, [c_void_p, c_void_p]),  # s, d_out_table
    "zg_sumcheck_close": (c_int, [c_void_p]),  # s
    "zg_sumcheck_raf_round": (c_int, [c_void_p, c_void_p, c_uint64, c_void_p, c_void_p]),  # s, base, current_power, s0, s2
    "zg_sumcheck_raf_claim": (c_int, [c_void_p, c_uint64, c_uint64, c_void_p]),  # s, base, step, claim
    "zg_sumcheck_bit_round": (c_int, [c_void_p, c_void_p, c_size_t, c_uint, c_void_p, c_void_p]),  # s, d_idx128, n_idx, bit, sum0, sum1
    "zg_sumcheck_bit_bind": (c_int, [c_void_p, c_void_p, c_size_t, c_uint, c_void_p, c_void_p]),  # s, d_idx128, n_idx, bit, r, claim
    "zg_fr_bit_split_sums": (c_int, [c_void_p, c_void_p, c_size_t, c_uint, c_void_p, c_void_p]),  # vals, idx128, n, bit, sum0, sum1
    "zg_fr_bit_split_sums_dev": (c_int, [c_void_p, c_void_p, c_size_t, c_uint, c_void_p, c_void_p, c_void_p]),  # d_vals, d_idx128, n, bit, stream, sum0, sum1
    "zg_selftest_handoff": (c_int, [c_uint, c_uint, c_uint, c_int, c_void_p, c_void_p]),  # blocks, threads, iters, busy, mismatches, completed
    "zg_run_sumcheck_dev": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),  # d_evals, len, stream, claim, rounds, challenges, final_eval, result
    "zg_run_sumcheck": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),  # evals, len, claim, rounds, challenges, final_eval, result
    "zg_psc_open": (c_int, [c_void_p, c_size_t, c_size_t, c_void_p]),  # tables, k, len, s
    "zg_psc_open_dev": (c_int, [c_void_p, c_size_t, c_size_t, c_void_p, c_void_p]),  # d_tables, k, len, stream, s
    "zg_psc_len": (c_size_t, [c_void_p]),  # s
    "zg_psc_tables": (c_size_t, [c_void_p]),  # s
    "zg_psc_round_evals": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_size_t, c_void_p]),  # s, prod_idx, p, lin_idx, lin_coeff, q, out
    "zg_psc_round_expr": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),  # s, terms, n_terms, out
    "zg_psc_set_points": (c_int, [c_void_p, c_uint]),  # s, points
    "zg_psc_round_gruen": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p]),  # s, prod_idx, p, d_e_out, n_out, d_e_in, n_in, t0, t_inf
    "zg_psc_bind": (c_int, [c_void_p, c_void_p]),  # s, r
    "zg_psc_read": (c_int, [c_void_p, c_size_t, c_void_p]),  # s, table, out
    "zg_psc_table_dev": (c_int, [c_void_p, c_size_t, c_void_p]),  # s, table, d_ptr
    "zg_psc_gather": (c_int, [c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]),  # s, table, idx, n, out
    "zg_psc_final": (c_int, [c_void_p, c_void_p]),  # s, out
    "zg_psc_close": (c_int, [c_void_p]),  # s
    "zg_rrw_open": (c_int, [c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),  # log_t, rs1, rs2, rd, reg_vals, inc, gamma, s
    "zg_rrw_open_trace": (c_int, [c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),  # log_t, rs1, rs2, rd, rd_value, gamma, s
    "zg_rrw_cycles": (c_size_t, [c_void_p]),  # s
    "zg_rrw_registers": (c_size_t, [c_void_p]),  # s
    "zg_rrw_round_cycle_gruen": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p]),  # s, d_e_out, n_out, d_e_in, n_in, q0, qx2
    "zg_rrw_set_eq": (c_int, [c_void_p, c_void_p, c_size_t]),  # s, eq, n
    "zg_rrw_round_address": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),  # s, e0, e1, e2
    "zg_rrw_round_cycle": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),  # s, e0, e1, e2, e3
    "zg_rrw_bind_cycle": (c_int, [c_void_p, c_void_p]),  # s, r
    "zg_rrw_bind_address": (c_int, [c_void_p, c_void_p]),  # s, r
    "zg_rrw_final": (c_int, [c_void_p, c_void_p]),  # s, out
    "zg_rrw_close": (c_int, [c_void_p]),  # s
    "zg_rwc_open": (c_int, [c_size_t, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),  # log_k, log_t, n, cycle, address, val_coeff, prev_val, next_val, inc, val_init, r_cycle, s
    "zg_rwc_open_writes": (c_int, [c_size_t, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),  # log_k, log_t, n, cycle, address, val_coeff, prev_val, next_val, is_write, val_init, r_cycle, s
    "zg_rwc_entries": (c_size_t, [c_void_p]),  # s
    "zg_rwc_cycles": (c_size_t, [c_void_p]),  # s
    "zg_rwc_round_cycle": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),  # s, d_e_out, n_out, d_e_in, n_in, gamma, q_constant, q_quadratic
    "zg_rwc_bind_cycle": (c_int, [c_void_p, c_void_p]),  # s, r
    "zg_rwc_round_address": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),  # s, addr_round, challenges, gamma, s0, s2
    "zg_rwc_bind_address": (c_int, [c_void_p, c_size_t, c_void_p]),  # s, addr_round, r
    "zg_rwc_opening": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),  # s, r_address, r_cycle, out
    "zg_rwc_cycle_scalars": (c_int, [c_void_p, c_void_p, c_void_p]),  # s, eq0, inc0
    "zg_rwc_read_entries": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),  # s, cycle, address, ra_coeff, val_coeff, prev_val, next_val
    "zg_rwc_close": (c_int, [c_void_p]),  # s
    "zg_shard_bounds": (c_int, [c_size_t, c_int, c_int, c_void_p, c_void_p]),  # n, shards, shard, start, len
    "zg_g1_bases_upload_sharded": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),  # xy, inf, n, cfg, out
    "zg_g1_sbases_free": (c_int, [c_void_p]),  # sb
    "zg_g1_sbases_len": (c_size_t, [c_void_p]),  # sb
    "zg_g1_sbases_shards": (c_int, [c_void_p]),  # sb
    "zg_g1_sbases_exchange": (c_int, [c_void_p]),  # sb
    "zg_g1_sbases_shard": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),  # sb, shard, device, start, len
    "zg_msm_g1_sharded": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),  # sb, n, scalars_mont, out_xy, out_inf
    "zg_msm_g1_sharded_dev": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),  # sb, n, d_scalars_per_shard, out_xy, out_inf
    "zg_msm_g1_batch_sharded": (c_int, [c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p]),  # sb, n, scalar_batches, k, out_xy, out_inf
    "zg_msm_g1_sharded_dev_async": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),  # sb, n, d_scalars_per_shard, ready_streams, ticket
    "zg_msm_g1_batch_sharded_async": (c_int, [c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]),  # sb, n, scalar_batches, k, ticket
    "zg_sharded_wait": (c_int, [c_void_p, c_uint64, c_void_p, c_void_p]),  # sb, ticket, out_xy, out_inf
    "zg_g1_sbases_inflight": (c_int, [c_void_p]),  # sb
    "zg_sumcheck_open_sharded": (c_int, [c_void_p, c_size_t, c_int, c_void_p]),  # evals, len, layout, s
    "zg_sumcheck_shards": (c_int, [c_void_p]),  # s
    "zg_sumcheck_len_sharded": (c_size_t, [c_void_p]),  # s
    "zg_sumcheck_round_sums_sharded": (c_int, [c_void_p, c_void_p, c_void_p]),  # s, g0, g1
    "zg_sumcheck_bind_sharded": (c_int, [c_void_p, c_void_p]),  # s, r
    "zg_sumcheck_final_sharded": (c_int, [c_void_p, c_void_p]),  # s, out
    "zg_sumcheck_close_sharded": (c_int, [c_void_p]),  # s
}

INTERNAL_PROTOS = {
    "zg_profile_begin": (c_int, [c_int]),  # max_records
    "zg_profile_end": (c_int, [c_void_p, c_void_p]),  # ms_out, count_out
    "zg_last_setup_times": (c_int, [c_void_p]),  # out
    "zg_sharded_comm_sets_created": (c_int, []),  # 
    "zg_pool_debug_stats": (c_int, [c_void_p]),  # out
    "zg_pool_debug_selftest": (c_int, []),  # 
}

SYMBOLS = list(PROTOS)
INTERNAL_SYMBOLS = list(INTERNAL_PROTOS)


def apply(lib):
    """restype / argtypes of every declared entry point; a symbol the header declares and the library lacks raises AttributeError"""
    for table in (PROTOS, INTERNAL_PROTOS):
        for name, (ret, args) in table.items():
            fn = getattr(lib, name)
            fn.restype = ret
            fn.argtypes = args
