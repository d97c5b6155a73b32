"""ctypes binding for the CPU oracle (TEST INFRASTRUCTURE).

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg import
this module. The product package (zolt_amd/) must never import it.

Arrays are numpy uint64: field elements are (..., 4) Montgomery limbs, affine
points (..., 8) (x limbs, y limbs) plus a uint8 infinity flag array.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
FR, FP = 0, 1


def _cpu_has(*flags):
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    have = set(line.split(":", 1)[1].split())
                    return all(fl in have for fl in flags)
    except OSError:
        pass
    return False


def build():
    subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)


def _load():
    v3 = os.path.join(_HERE, "libzolt_oracle_v3.so")
    v2 = os.path.join(_HERE, "libzolt_oracle.so")
    if not os.path.exists(v2):
        build()
    path = v3 if (os.path.exists(v3) and _cpu_has("bmi2", "adx", "avx2")) else v2
    return C.CDLL(path), path


lib, LIB_PATH = _load()

_u64p = C.POINTER(C.c_uint64)
_u8p = C.POINTER(C.c_uint8)


def _p(a):
    return None if a is None else a.ctypes.data_as(_u64p)


def _b(a):
    return None if a is None else a.ctypes.data_as(_u8p)


def _c(a, dtype=np.uint64):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


for _name in ("zo_optimal_window_size", "zo_get_window", "zo_f_inv"):
    getattr(lib, _name).restype = C.c_size_t
lib.zo_run_sumcheck.restype = C.c_int
lib.zo_g1_is_on_curve.restype = C.c_int


# ---- field
def _binop(name, f, a, b):
    a, b = _c(a), _c(b)
    o = np.empty_like(a)
    getattr(lib, name)(C.c_int(f), _p(a), _p(b), _p(o), C.c_size_t(a.size // 4))
    return o


def _unop(name, f, a):
    a = _c(a)
    o = np.empty_like(a)
    getattr(lib, name)(C.c_int(f), _p(a), _p(o), C.c_size_t(a.size // 4))
    return o


def f_mul(f, a, b): return _binop("zo_f_mul", f, a, b)
def f_add(f, a, b): return _binop("zo_f_add", f, a, b)
def f_sub(f, a, b): return _binop("zo_f_sub", f, a, b)
def f_neg(f, a): return _unop("zo_f_neg", f, a)
def f_sqr(f, a): return _unop("zo_f_sqr", f, a)
def f_inv(f, a): return _unop("zo_f_inv", f, a)
def f_from_mont(f, a): return _unop("zo_f_from_mont", f, a)
def f_to_mont(f, a): return _unop("zo_f_to_mont", f, a)


def f_from_u64(f, v):
    v = _c(v)
    o = np.empty(v.shape + (4,), dtype=np.uint64)
    lib.zo_f_from_u64(C.c_int(f), _p(v), _p(o), C.c_size_t(v.size))
    return o


def f_to_bytes_be(f, a):
    a = _c(a)
    out = np.empty(32, dtype=np.uint8)
    lib.zo_f_to_bytes_be(C.c_int(f), _p(a), _b(out))
    return out.tobytes()


# ---- G1 / MSM
def optimal_window_size(n): return int(lib.zo_optimal_window_size(C.c_size_t(n)))


def get_window(scalar, w, c):
    s = _c(scalar)
    return int(lib.zo_get_window(_p(s), C.c_size_t(w), C.c_size_t(c)))


def msm_g1(xy, inf, scalars):
    xy, inf, scalars = _c(xy), _c(inf, np.uint8), _c(scalars)
    n = scalars.size // 4
    out = np.empty(8, dtype=np.uint64)
    oinf = np.zeros(1, dtype=np.uint8)
    lib.zo_msm_g1(_p(xy), _b(inf), _p(scalars), C.c_size_t(n), _p(out), _b(oinf))
    return out, int(oinf[0])


def msm_g1_parallel(xy, inf, scalars, threads):
    xy, inf, scalars = _c(xy), _c(inf, np.uint8), _c(scalars)
    n = scalars.size // 4
    out = np.empty(8, dtype=np.uint64)
    oinf = np.zeros(1, dtype=np.uint8)
    lib.zo_msm_g1_parallel(_p(xy), _b(inf), _p(scalars), C.c_size_t(n), C.c_size_t(threads), _p(out), _b(oinf))
    return out, int(oinf[0])


def msm_g1_batch(xy, inf, batches):
    xy, inf = _c(xy), _c(inf, np.uint8)
    batches = [_c(b) for b in batches]
    n = batches[0].size // 4 if batches else 0
    k = len(batches)
    arr = (_u64p * k)(*[_p(b) for b in batches])
    out = np.empty((k, 8), dtype=np.uint64)
    oinf = np.zeros(k, dtype=np.uint8)
    lib.zo_msm_g1_batch(_p(xy), _b(inf), C.c_size_t(n), arr, C.c_size_t(k), _p(out), _b(oinf))
    return out, oinf


def g1_scalar_mul(xy, inf, scalar):
    xy, scalar = _c(xy), _c(scalar)
    out = np.empty(8, dtype=np.uint64)
    oinf = np.zeros(1, dtype=np.uint8)
    lib.zo_g1_scalar_mul(_p(xy), C.c_uint8(inf), _p(scalar), _p(out), _b(oinf))
    return out, int(oinf[0])


def g1_add_affine(a, ainf, b, binf):
    a, b = _c(a), _c(b)
    out = np.empty(8, dtype=np.uint64)
    oinf = np.zeros(1, dtype=np.uint8)
    lib.zo_g1_add_affine(_p(a), C.c_uint8(ainf), _p(b), C.c_uint8(binf), _p(out), _b(oinf))
    return out, int(oinf[0])


def g1_double_affine(a, ainf):
    a = _c(a)
    out = np.empty(8, dtype=np.uint64)
    oinf = np.zeros(1, dtype=np.uint8)
    lib.zo_g1_double_affine(_p(a), C.c_uint8(ainf), _p(out), _b(oinf))
    return out, int(oinf[0])


def g1_jac_add(a, b):
    a, b = _c(a), _c(b)
    o = np.empty(12, dtype=np.uint64)
    lib.zo_g1_jac_add(_p(a), _p(b), _p(o))
    return o


def g1_jac_double(a):
    a = _c(a)
    o = np.empty(12, dtype=np.uint64)
    lib.zo_g1_jac_double(_p(a), _p(o))
    return o


def g1_jac_add_affine(a, b, binf=0):
    a, b = _c(a), _c(b)
    o = np.empty(12, dtype=np.uint64)
    lib.zo_g1_jac_add_affine(_p(a), _p(b), C.c_uint8(binf), _p(o))
    return o


def g1_jac_to_affine(a):
    a = _c(a)
    out = np.empty(8, dtype=np.uint64)
    oinf = np.zeros(1, dtype=np.uint8)
    lib.zo_g1_jac_to_affine(_p(a), _p(out), _b(oinf))
    return out, int(oinf[0])


def g1_is_on_curve(xy, inf=0):
    xy = _c(xy)
    return bool(lib.zo_g1_is_on_curve(_p(xy), C.c_uint8(inf)))


def g1_gen_multiples(n):
    out = np.empty((n, 8), dtype=np.uint64)
    lib.zo_g1_gen_multiples(C.c_size_t(n), _p(out))
    return out


# ---- HyperKZG
def hyperkzg_setup(n):
    out = np.empty((n, 8), dtype=np.uint64)
    inf = np.zeros(n, dtype=np.uint8)
    lib.zo_hyperkzg_setup(C.c_size_t(n), _p(out), _b(inf))
    return out, inf


def hyperkzg_commit(srs_xy, srs_inf, evals):
    srs_xy, srs_inf, evals = _c(srs_xy), _c(srs_inf, np.uint8), _c(evals)
    out = np.empty(8, dtype=np.uint64)
    oinf = np.zeros(1, dtype=np.uint8)
    lib.zo_hyperkzg_commit(_p(srs_xy), _b(srs_inf), C.c_size_t(srs_xy.size // 8), _p(evals),
                           C.c_size_t(evals.size // 4), _p(out), _b(oinf))
    return out, int(oinf[0])


def hyperkzg_open(srs_xy, srs_inf, evals, point, value):
    srs_xy, srs_inf, evals, point, value = _c(srs_xy), _c(srs_inf, np.uint8), _c(evals), _c(point), _c(value)
    v = point.size // 4
    q = np.zeros((v, 8), dtype=np.uint64)
    qinf = np.zeros(v, dtype=np.uint8)
    fin = np.empty(4, dtype=np.uint64)
    lib.zo_hyperkzg_open(_p(srs_xy), _b(srs_inf), C.c_size_t(srs_xy.size // 8), _p(evals),
                         C.c_size_t(evals.size // 4), _p(point), C.c_size_t(v), _p(value), _p(q), _b(qinf), _p(fin))
    return q, qinf, fin


def hyperkzg_batch_open(srs_xy, srs_inf, polys, point):
    """-> (q_xy[nq,8], q_inf[nq], evaluations[k,4], final_eval, gamma)"""
    srs_xy, srs_inf, point = _c(srs_xy), _c(srs_inf, np.uint8), _c(point)
    polys = [_c(p) for p in polys]
    k, v = len(polys), point.size // 4
    ptrs = (C.c_void_p * max(k, 1))(*[p.ctypes.data for p in polys])
    lens = (C.c_size_t * max(k, 1))(*[p.size // 4 for p in polys])
    q = np.zeros((max(v, 1), 8), dtype=np.uint64)
    qi = np.zeros(max(v, 1), dtype=np.uint8)
    nq = C.c_size_t(0)
    ev = np.zeros((max(k, 1), 4), dtype=np.uint64)
    fin = np.zeros(4, dtype=np.uint64)
    gam = np.zeros(4, dtype=np.uint64)
    lib.zo_hyperkzg_batch_open(_p(srs_xy), _b(srs_inf), C.c_size_t(srs_xy.size // 8), ptrs, lens, C.c_size_t(k), _p(point),
                               C.c_size_t(v), _p(q), _b(qi), C.byref(nq), _p(ev), _p(fin), _p(gam))
    return q[:nq.value], qi[:nq.value], ev[:k], fin, gam


def commitment_to_bytes(xy):
    xy = _c(xy)
    out = np.empty(64, dtype=np.uint8)
    lib.zo_commitment_to_bytes(_p(xy), _b(out))
    return out.tobytes()


# ---- poly / sumcheck
def fr_eq_table(r, scale=None):
    r = _c(r)
    v = r.size // 4
    out = np.empty((1 << v, 4), dtype=np.uint64)
    lib.zo_fr_eq_table(_p(r), C.c_size_t(v), _p(_c(scale)), _p(out))
    return out


def fr_eq_mle(r, x):
    """EqPolynomial.mle / evaluate (src/poly/mod.zig:214-227,311-321)"""
    r, x = _c(r), _c(x)
    out = np.empty(4, dtype=np.uint64)
    lib.zo_fr_eq_mle(_p(r), _p(x), C.c_size_t(r.size // 4), _p(out))
    return out


def fr_poly_scale(a, s):
    """DensePolynomial.scale (src/poly/mod.zig:112-126)"""
    a = _c(a)
    out = np.empty_like(a)
    lib.zo_fr_poly_scale(_p(a), C.c_size_t(a.size // 4), _p(_c(s)), _p(out))
    return out


def fr_eq_table_append_lsb(tau):
    tau = _c(tau)
    v = tau.size // 4
    out = np.empty((1 << v, 4), dtype=np.uint64)
    lib.zo_fr_eq_table_append_lsb(_p(tau), C.c_size_t(v), _p(out))
    return out


def fr_eq_prefix_tables(tau):
    """list of the v+1 prefix tables eq(tau[0..k), .) — GruenSplitEqPolynomial E_out_vec / E_in_vec (split_eq.zig:122-171)"""
    tau = _c(np.asarray(tau, dtype=np.uint64).reshape(-1, 4))
    v = tau.shape[0]
    out = np.empty(((2 << v) - 1, 4), dtype=np.uint64)
    lib.zo_fr_eq_prefix_tables(_p(tau), C.c_size_t(v), _p(out))
    return [out[(1 << k) - 1:(2 << k) - 1].copy() for k in range(v + 1)]


class GruenSplitEq:
    """CPU restatement of GruenSplitEqPolynomial's state machine (src/poly/split_eq.zig:63-514); field arithmetic in the C oracle."""

    def __init__(self, tau, scaling_factor=None):
        self.tau = _c(np.asarray(tau, dtype=np.uint64).reshape(-1, 4)).copy()
        n = self.tau.shape[0]
        self.current_index = n
        self.current_scalar = f_from_u64(FR, np.array([1], dtype=np.uint64))[0] if scaling_factor is None else _c(scaling_factor).copy()
        m = n // 2
        self.num_x_out = m if n else 0
        self.num_x_in = (n - 1 - m) if n > 1 else 0
        if n == 0:  # :75-86: no tables at all
            self.E_out_vec, self.E_in_vec = [], []
        else:
            self.E_out_vec = fr_eq_prefix_tables(self.tau[:m])
            self.E_in_vec = fr_eq_prefix_tables(self.tau[m:m + self.num_x_in])

    def bind(self, r):  # :213-248
        if self.current_index == 0:
            return
        out = np.empty(4, dtype=np.uint64)
        lib.zo_gruen_bind_scalar(_p(_c(self.current_scalar)), _p(_c(self.tau[self.current_index - 1])), _p(_c(r)), _p(out))
        self.current_scalar = out
        self.current_index -= 1
        m = self.tau.shape[0] // 2
        if m < self.current_index:
            if len(self.E_in_vec) > 1:
                self.E_in_vec.pop()
        elif self.current_index > 0:
            if len(self.E_out_vec) > 1:
                self.E_out_vec.pop()

    def getFullEqTable(self):  # :254-285 (same values as the append-LSB build scaled by current_scalar)
        t = fr_eq_table_append_lsb(self.tau[:self.current_index])
        return fr_poly_scale(t, self.current_scalar)

    def getTauHigh(self):  # :291-294
        return self.tau[-1] if self.tau.shape[0] else np.zeros(4, dtype=np.uint64)

    def getWindowEqTables(self, window_size):  # :312-343
        num_unbound = self.current_index
        head_len = max(num_unbound - min(window_size, num_unbound), 0)
        m = self.tau.shape[0] // 2
        head_out_bits = min(head_len, m)
        head_in_bits = max(head_len - head_out_bits, 0)
        e_out = self.E_out_vec[head_out_bits] if head_out_bits < len(self.E_out_vec) else self.E_out_vec[-1]
        e_in = self.E_in_vec[head_in_bits] if head_in_bits < len(self.E_in_vec) else self.E_in_vec[-1]
        return e_out, e_in, head_in_bits

    def computeCubicRoundPoly(self, q_constant, q_quadratic_coeff, previous_claim):  # :353-434
        if self.current_index == 0:
            out = np.zeros((4, 4), dtype=np.uint64)
            out[0] = previous_claim
            return out
        out = np.empty((4, 4), dtype=np.uint64)
        lib.zo_gruen_cubic_round_poly(_p(_c(self.current_scalar)), _p(_c(self.tau[self.current_index - 1])), _p(_c(q_constant)),
                                      _p(_c(q_quadratic_coeff)), _p(_c(previous_claim)), _p(out))
        return out

    def getEActiveForWindow(self, window_size):  # :466-514
        one = f_from_u64(FR, np.array([1], dtype=np.uint64))
        if window_size <= 1 or window_size > self.current_index:
            return one
        ws = self.current_index - window_size
        return fr_eq_table_append_lsb(self.tau[ws:ws + window_size - 1])


def fr_bind_low(table, r):
    t = np.array(table, dtype=np.uint64, copy=True)
    n = t.size // 4
    lib.zo_fr_bind_low(_p(t), C.c_size_t(n), _p(_c(r)))
    return t.reshape(-1, 4)[: n // 2].copy()


def fr_bind_high(table, r):
    t = _c(table)
    n = t.size // 4
    out = np.empty((n // 2, 4), dtype=np.uint64)
    lib.zo_fr_bind_high(_p(t), C.c_size_t(n), _p(_c(r)), _p(out))
    return out


def fr_bind_low_2mul(table, r):
    t = _c(table)
    n = t.size // 4
    out = np.empty((n // 2, 4), dtype=np.uint64)
    lib.zo_fr_bind_low_2mul(_p(t), C.c_size_t(n), _p(_c(r)), _p(out))
    return out


def fr_dense_evaluate(evals, point):
    e, p = _c(evals), _c(point)
    out = np.empty(4, dtype=np.uint64)
    lib.zo_fr_dense_evaluate(_p(e), C.c_size_t(p.size // 4), _p(p), _p(out))
    return out


def fr_sum_halves(table):
    t = _c(table)
    g0 = np.empty(4, dtype=np.uint64)
    g1 = np.empty(4, dtype=np.uint64)
    lib.zo_fr_sum_halves(_p(t), C.c_size_t(t.size // 4), _p(g0), _p(g1))
    return g0, g1


def fr_sum_even_odd(table):
    t = _c(table)
    g0 = np.empty(4, dtype=np.uint64)
    g1 = np.empty(4, dtype=np.uint64)
    lib.zo_fr_sum_even_odd(_p(t), C.c_size_t(t.size // 4), _p(g0), _p(g1))
    return g0, g1


def fr_spartan_combine(eq, az, bz, cz):
    eq, az, bz, cz = _c(eq), _c(az), _c(bz), _c(cz)
    out = np.empty_like(eq)
    lib.zo_fr_spartan_combine(_p(eq), _p(az), _p(bz), _p(cz), C.c_size_t(eq.size // 4), _p(out))
    return out


def sumcheck_derive_challenge(rnd, claim, coeffs):
    claim, coeffs = _c(claim), _c(coeffs)
    out = np.empty(4, dtype=np.uint64)
    lib.zo_sumcheck_derive_challenge(C.c_size_t(rnd), _p(claim), _p(coeffs), C.c_size_t(coeffs.size // 4), _p(out))
    return out


def run_sumcheck(evals):
    e = _c(evals)
    n = e.size // 4
    v = n.bit_length() - 1
    claim = np.empty(4, dtype=np.uint64)
    rounds = np.empty((v, 2, 4), dtype=np.uint64)
    chals = np.empty((v, 4), dtype=np.uint64)
    fin = np.empty(4, dtype=np.uint64)
    ok = lib.zo_run_sumcheck(_p(e), C.c_size_t(v), _p(claim), _p(rounds), _p(chals), _p(fin))
    return claim, rounds, chals, fin, int(ok)


# ---- prover fold sites + host transcript
def keccak_f1600(lanes):
    st = np.array(lanes, dtype=np.uint64).copy()
    lib.zo_keccak_f1600(_p(st))
    return st


class Transcript:
    """Keccak Transcript(F) of the reference (src/transcripts/mod.zig:49-161), state held in a 208-byte buffer."""

    def __init__(self, domain=b"Jolt"):
        self._buf = C.create_string_buffer(208)
        lib.zo_transcript_init(self._buf, C.c_char_p(bytes(domain)), C.c_size_t(len(domain)))

    def append_bytes(self, data):
        data = bytes(data)
        lib.zo_transcript_append_bytes(self._buf, C.c_char_p(data), C.c_size_t(len(data)))

    def append_scalar(self, label, scalar):
        label = bytes(label)
        lib.zo_transcript_append_scalar(self._buf, C.c_char_p(label), C.c_size_t(len(label)), _p(_c(scalar)))

    def challenge_scalar(self, label):
        label = bytes(label)
        out = np.empty(4, dtype=np.uint64)
        lib.zo_transcript_challenge_scalar(self._buf, C.c_char_p(label), C.c_size_t(len(label)), _p(out))
        return out

    def state_bytes(self):
        return bytes(self._buf.raw[:200]), int.from_bytes(self._buf.raw[200:208], "little")


def stage1_prove(combined_poly, num_rounds, transcript):
    """prover.zig:397-432 -> (round_polys (rounds,3,4), challenges (rounds,4), final_eval)"""
    poly = np.array(combined_poly, dtype=np.uint64, copy=True).reshape(-1, 4)
    rp = np.zeros((num_rounds, 3, 4), dtype=np.uint64)
    ch = np.zeros((num_rounds, 4), dtype=np.uint64)
    fin = np.zeros(4, dtype=np.uint64)
    lib.zo_stage1_prove(_p(poly), C.c_size_t(poly.shape[0]), C.c_size_t(num_rounds), transcript._buf, _p(rp), _p(ch), _p(fin))
    return rp, ch, fin


def raf_round_cubic(ra, start_address, bound, unmap_num_vars, current_claim):
    """RafEvaluationProver.computeRoundPolynomialCubic (raf_checking.zig:335-410) -> (4,4): s(0..3)"""
    ra = _c(ra).reshape(-1, 4)
    bound = _c(bound).reshape(-1, 4) if bound is not None and len(bound) else np.zeros((0, 4), dtype=np.uint64)
    nv = ra.shape[0].bit_length() - 1
    out = np.zeros((4, 4), dtype=np.uint64)
    lib.zo_raf_round_cubic(_p(ra), C.c_size_t(nv), C.c_uint64(start_address), _p(bound), C.c_size_t(bound.shape[0]), C.c_size_t(unmap_num_vars),
                           _p(_c(current_claim)), _p(out))
    return out


def raf_update_claim(evals, challenge):
    out = np.zeros(4, dtype=np.uint64)
    lib.zo_raf_update_claim(_p(_c(evals)), _p(_c(challenge)), _p(out))
    return out


def lasso_address_sums(eq_evals, idx128, round_bit):
    """LassoProver.computeAddressRoundPoly's sums (lasso/prover.zig:283-293); idx128: (n,2) uint64 little-endian halves"""
    eq_evals, idx128 = _c(eq_evals), _c(idx128)
    s0, s1 = np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
    lib.zo_lasso_address_sums(_p(eq_evals), _p(idx128), C.c_size_t(eq_evals.size // 4), C.c_uint(round_bit), _p(s0), _p(s1))
    return s0, s1


class LassoProver:
    """CPU restatement of LassoProver's eq_evals path (src/zkvm/lasso/prover.zig:113-453): init, computeRoundPolynomial and
    receiveChallenge in both phases. The prefix-suffix structures the reference also binds do not enter the round polynomials."""

    def __init__(self, lookup_indices_u128, log_T, log_K, r_reduction):
        self.idx = _c(np.asarray(lookup_indices_u128, dtype=np.uint64).reshape(-1, 2))
        self.log_T, self.log_K = log_T, log_K
        w = _c(np.asarray(r_reduction, dtype=np.uint64).reshape(-1, 4))
        assert w.shape[0] == log_T
        outer = log_T // 2
        self.eq_evals = np.zeros((1 << log_T, 4), dtype=np.uint64)
        self.current_claim = np.zeros(4, dtype=np.uint64)
        lib.zo_lasso_init_eq_evals(_p(w), C.c_size_t(outer), C.c_size_t(log_T - outer), C.c_size_t(self.idx.shape[0]),
                                   _p(self.eq_evals), _p(self.current_claim))
        self.eq_evals_len = 1 << log_T
        self.round = 0
        self.challenges = []

    def isAddressPhase(self):
        return self.round < self.log_K

    def isComplete(self):
        return self.round >= self.log_K + self.log_T

    def computeRoundPolynomial(self):
        """-> coeffs [c0, c1, c2] (:262-345)"""
        zero = np.zeros(4, dtype=np.uint64)
        if self.isAddressPhase():
            s0, s1 = lasso_address_sums(self.eq_evals[:self.idx.shape[0]], self.idx, self.round)
        else:
            n = self.eq_evals_len
            if n <= 1:
                return np.stack([self.eq_evals[0] if n else zero, zero, zero])
            s0, s1 = fr_sum_halves(self.eq_evals[:n])
        return np.stack([s0, f_sub(FR, s1[None, :], s0[None, :])[0], zero])

    def receiveChallenge(self, challenge):
        """:352-453"""
        challenge = _c(challenge)
        self.challenges.append(challenge.copy())
        if self.isAddressPhase():
            claim = np.zeros(4, dtype=np.uint64)
            lib.zo_lasso_receive_address(_p(self.eq_evals), C.c_size_t(self.eq_evals.shape[0]), _p(self.idx), C.c_size_t(self.idx.shape[0]),
                                         C.c_uint(self.round), _p(challenge), _p(claim))
            self.current_claim = claim
        else:
            n = self.eq_evals_len
            if n > 1:
                folded = fr_bind_high(self.eq_evals[:n], challenge)
                self.eq_evals[:n // 2] = folded
                self.eq_evals_len = n // 2
                if len(folded) >= 2:  # :431-435: the sum of the folded array (a modular sum: any grouping gives the same value)
                    h0, h1 = fr_sum_halves(folded)
                    self.current_claim = f_add(FR, h0[None, :], h1[None, :])[0]
                else:
                    self.current_claim = folded[0].copy()
        self.round += 1

    def getFinalEval(self):
        """:458-462: expanding_v.get(0) after log_K binds = prod (1 - r_i) (expanding_table.zig:83-99: new[0] = old[0] * (1 - r))"""
        acc = f_from_u64(FR, np.array([1], dtype=np.uint64))
        one = acc.copy()
        for c in self.challenges[:self.log_K]:
            acc = f_mul(FR, acc, f_sub(FR, one, c[None, :]))
        return acc[0]


def lasso_derive_challenge(coeffs, round_index):
    coeffs = _c(coeffs)
    out = np.empty(4, dtype=np.uint64)
    lib.zo_lasso_derive_challenge(_p(coeffs), C.c_size_t(coeffs.size // 4), C.c_uint64(round_index), _p(out))
    return out


def run_lasso_prover(lookup_indices_u128, log_T, log_K, r_reduction):
    """runLassoProver (lasso/prover.zig:495-530)"""
    p = LassoProver(lookup_indices_u128, log_T, log_K, r_reduction)
    polys, rnd = [], 0
    while not p.isComplete():
        polys.append(p.computeRoundPolynomial())
        p.receiveChallenge(lasso_derive_challenge(polys[-1], rnd))
        rnd += 1
    return {"round_polys": np.stack(polys) if polys else np.zeros((0, 3, 4), dtype=np.uint64), "final_eval": p.getFinalEval(),
            "challenges": np.stack(p.challenges) if p.challenges else np.zeros((0, 4), dtype=np.uint64)}


# ---- product-form zkVM provers (sumcheck loops only: tables in, round polynomials / claims / final values out)
def interpolate_degree3(evals):
    out = np.empty((4, 4), dtype=np.uint64)
    lib.zo_interpolate_degree3(_p(_c(evals)), _p(out))
    return out


def evals_to_compressed(evals):
    """UniPoly.evalsToCompressed (poly/mod.zig:682-685): [c0, c2, c3]"""
    c = interpolate_degree3(evals)
    return np.stack([c[0], c[2], c[3]])


class ValEvaluationProver:
    """ValEvaluationProver (src/zkvm/ram/val_evaluation.zig:545-660); lt=None: ValFinalProver (src/zkvm/ram/val_final.zig:144-230)"""

    def __init__(self, inc, wa, lt, claim):
        self.t = [_c(x).reshape(-1, 4).copy() for x in ((inc, wa) if lt is None else (inc, wa, lt))]
        self.n = self.t[0].shape[0]
        self.current_claim = _c(claim).copy()
        self.round = 0

    def computeRoundPolynomial(self):
        out = np.empty((4, 4), dtype=np.uint64)
        lt = self.t[2] if len(self.t) == 3 else None
        lib.zo_val_evaluation_round(_p(self.t[0]), _p(self.t[1]), _p(lt), C.c_size_t(self.n), _p(out))
        return out

    def bindChallengeWithPoly(self, r, round_poly):
        if self.n // 2 == 0:
            self.round += 1
            return
        self.t = [fr_bind_low_2mul(x[:self.n], r) for x in self.t]  # (1-r)*lo + r*hi (:609-620)
        self.n //= 2
        self.current_claim = raf_update_claim(round_poly, r)  # the same cubic Lagrange step (:630-660)
        self.round += 1

    def getFinalClaims(self):
        return [x[0].copy() for x in self.t]


class OutputSumcheckProver:
    """OutputSumcheckProver's loop (src/zkvm/ram/output_check.zig:375-499)"""

    def __init__(self, eq_r_address, io_mask, val_final, val_io, val_init, claim):
        self.t = [_c(x).reshape(-1, 4).copy() for x in (eq_r_address, io_mask, val_final, val_io, val_init)]
        self.current_size = self.t[0].shape[0]
        self.current_claim = _c(claim).copy()

    def roundEvals(self):
        out = np.empty((4, 4), dtype=np.uint64)
        lib.zo_output_check_round(_p(self.t[0]), _p(self.t[1]), _p(self.t[2]), _p(self.t[3]), C.c_size_t(self.current_size), _p(out))
        return out

    def computeRoundPolynomial(self):
        return evals_to_compressed(self.roundEvals())

    def bindChallenge(self, r):
        self.t = [fr_bind_low(x[:self.current_size], r) for x in self.t]
        self.current_size //= 2

    def updateClaim(self, evals, challenge):
        out = np.empty(4, dtype=np.uint64)
        lib.zo_output_check_update_claim(_p(_c(evals)), _p(_c(challenge)), _p(out))
        self.current_claim = out

    def getFinalClaims(self):
        return {"val_final": self.t[2][0], "val_init": self.t[4][0], "val_io": self.t[3][0], "eq_r_address": self.t[0][0], "io_mask": self.t[1][0]}


class InstructionLookupsClaimReduction:
    """InstructionLookupsClaimReductionProver's loop (src/zkvm/claim_reductions/instruction_lookups.zig:146-284)"""

    def __init__(self, eq_evals, lookup_outputs, left_operands, right_operands, gamma, claim):
        self.t = [_c(x).reshape(-1, 4).copy() for x in (eq_evals, lookup_outputs, left_operands, right_operands)]
        self.gamma = _c(gamma).copy()
        self.current_claim = _c(claim).copy()
        self.round = 0

    def computeRoundPolynomialCubic(self):
        out = np.empty((4, 4), dtype=np.uint64)
        lib.zo_instruction_lookups_round(_p(self.t[0]), _p(self.t[1]), _p(self.t[2]), _p(self.t[3]), C.c_size_t(self.t[0].shape[0]),
                                         _p(self.gamma), _p(self.current_claim), _p(out))
        return out

    def bindChallenge(self, challenge):
        self.t = [fr_bind_low(x, challenge) for x in self.t]
        self.round += 1

    def updateClaim(self, evals, challenge):
        self.current_claim = raf_update_claim(evals, challenge)

    def getOpeningClaims(self):
        return {"lookup_output": self.t[1][0], "left_operand": self.t[2][0], "right_operand": self.t[3][0]}


class ProductRemainderProver:
    """ProductVirtualRemainderProver's loop (src/zkvm/spartan/product_remainder.zig:269-394) over given fused left / right tables"""

    def __init__(self, left_evals, right_evals, tau_low, lagrange_kernel, uni_skip_claim):
        self.left = _c(left_evals).reshape(-1, 4).copy()
        self.right = _c(right_evals).reshape(-1, 4).copy()
        self.split_eq = GruenSplitEq(tau_low, lagrange_kernel)
        self.current_claim = _c(uni_skip_claim).copy()
        self.current_round = 0

    def roundEvals(self):
        n = self.left.shape[0]
        if n // 2 == 0:
            return None
        e_out, e_in, _ = self.split_eq.getWindowEqTables(1)
        t0, ti = np.empty(4, dtype=np.uint64), np.empty(4, dtype=np.uint64)
        lib.zo_product_remainder_sums(_p(self.left), _p(self.right), C.c_size_t(n), _p(_c(e_out)), C.c_size_t(len(e_out)), _p(_c(e_in)),
                                      C.c_size_t(len(e_in)), _p(t0), _p(ti))
        return self.split_eq.computeCubicRoundPoly(t0, ti, self.current_claim)

    def computeRoundPolynomial(self):
        ev = self.roundEvals()
        if ev is None:
            z = np.zeros(4, dtype=np.uint64)
            return np.stack([self.current_claim, z, z])
        return evals_to_compressed(ev)

    def bindChallenge(self, challenge):
        self.left = fr_bind_low(self.left, challenge)
        self.right = fr_bind_low(self.right, challenge)
        self.split_eq.bind(challenge)
        self.current_round += 1

    def updateClaim(self, round_evals, challenge):
        self.current_claim = raf_update_claim(round_evals, challenge)

    def getFinalClaim(self):
        return f_mul(FR, self.left[:1], self.right[:1])[0]


# ---- batched sumcheck driver (src/zkvm/batched_sumcheck.zig) restated over the C oracle's field arithmetic
def _fe(v):
    return f_from_u64(FR, np.array([v], dtype=np.uint64))


def _mul(a, b):
    return f_mul(FR, _c(a).reshape(1, 4), _c(b).reshape(1, 4))[0]


def _add(a, b):
    return f_add(FR, _c(a).reshape(1, 4), _c(b).reshape(1, 4))[0]


def _sub(a, b):
    return f_sub(FR, _c(a).reshape(1, 4), _c(b).reshape(1, 4))[0]


class BatchedSumcheck:
    """BatchedSumcheckProver (batched_sumcheck.zig:77-262). Instances: objects with num_rounds, input_claim,
    computeRoundPoly(round) -> (4,4), bindChallenge(challenge)."""

    def __init__(self, instances, coeffs, inactive_scaling="proof_converter"):
        """inactive_scaling "proof_converter": 2^(start - round - 1) (src/zkvm/proof_converter.zig:3330-3343, the loop the prover runs);
        "batched_sumcheck_zig": 2^(start - round) (batched_sumcheck.zig:208-212 as written)"""
        self.minus_one = inactive_scaling == "proof_converter"
        self.instances, self.coeffs = instances, [_c(c) for c in coeffs]
        self.max_num_rounds = max(i.num_rounds for i in instances)
        self.current_round = 0
        self.challenges = []
        acc = np.zeros(4, dtype=np.uint64)
        for inst, c in zip(instances, self.coeffs):  # :161-173
            scaled = _c(inst.input_claim)
            for _ in range(self.max_num_rounds - inst.num_rounds):
                scaled = _add(scaled, scaled)
            acc = _add(acc, _mul(scaled, c))
        self.current_claim = acc

    def combinedEvals(self):  # :193-222
        comb = np.zeros((4, 4), dtype=np.uint64)
        for inst, c in zip(self.instances, self.coeffs):
            start = self.max_num_rounds - inst.num_rounds
            if self.current_round >= start:
                ev = inst.computeRoundPoly(self.current_round - start)
                for j in range(4):
                    comb[j] = _add(comb[j], _mul(ev[j], c))
            else:
                scaled = _c(inst.input_claim)
                for _ in range(start - self.current_round - (1 if self.minus_one else 0)):
                    scaled = _add(scaled, scaled)
                w = _mul(scaled, c)
                for j in range(4):
                    comb[j] = _add(comb[j], w)
        return comb

    def computeRoundPolynomial(self):
        return evals_to_compressed(self.combinedEvals())

    def bindChallenge(self, challenge):  # :229-241
        self.challenges.append(_c(challenge).copy())
        for inst in self.instances:
            if self.current_round >= self.max_num_rounds - inst.num_rounds:
                inst.bindChallenge(challenge)
        self.current_round += 1

    def updateClaim(self, round_evals, challenge):
        self.current_claim = raf_update_claim(round_evals, challenge)


def decompress_round_poly(compressed, current_claim):
    """batched_sumcheck.zig:380-400"""
    c0, c2, c3 = (_c(x) for x in compressed)
    c1 = _sub(_sub(_sub(_sub(current_claim, c0), c0), c2), c3)
    out = []
    for t in range(4):
        v = _add(c0, _mul(c1, _fe(t)[0]))
        v = _add(v, _mul(c2, _fe(t * t)[0]))
        v = _add(v, _mul(c3, _fe(t * t * t)[0]))
        out.append(v)
    return np.stack(out)


def r1cs_claimed_inputs(cycle_witnesses, r_cycle):
    """R1CSInputEvaluator.computeClaimedInputs (src/zkvm/r1cs/evaluation.zig:55-122); cycle_witnesses: (T, k, 4)"""
    rows = _c(cycle_witnesses)
    r = _c(np.asarray(r_cycle, dtype=np.uint64).reshape(-1, 4))
    out = np.empty((rows.shape[1], 4), dtype=np.uint64)
    rc = lib.zo_r1cs_claimed_inputs(_p(rows), C.c_size_t(rows.shape[0]), C.c_size_t(rows.shape[1]), _p(r), C.c_size_t(r.shape[0]), _p(out))
    if rc != 0:
        raise IndexError("eq_evals index out of bounds (r_cycle shorter than log2 of the cycle count)")
    return out


# ---- Stage-3 prover rounds (src/zkvm/spartan/stage3_prover.zig): tables in, round evaluations out
def _ptrs(tabs):
    arrs = [_c(t).reshape(-1, 4) for t in tabs]
    return arrs, (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


def shift_phase1_round(P, Q, current_prefix_size):
    """ShiftSumcheckProver.computeRoundEvalsPhase1 (:1351-1392); P, Q: four tables each -> (3,4): p(0), p(1), p(2)"""
    pa, pp = _ptrs(P)
    qa, qp = _ptrs(Q)
    out = np.empty((3, 4), dtype=np.uint64)
    lib.zo_shift_phase1_round(pp, qp, C.c_size_t(current_prefix_size), _p(out))
    return out


def shift_phase2_round(tabs, gamma_powers, previous_claim):
    """computeRoundEvalsPhase2 (:1399-1455); tabs = [eq_outer, eq_prod, upc, pc, virt, first, noop] -> (3,4)"""
    arrs, ptrs = _ptrs(tabs)
    out = np.empty((3, 4), dtype=np.uint64)
    lib.zo_shift_phase2_round(ptrs, C.c_size_t(arrs[0].shape[0]), _p(_c(gamma_powers)), _p(_c(previous_claim)), _p(out))
    return out


def instruction_input_round(tabs, gamma, previous_claim):
    """InstructionInputProver.computeRoundEvals (:2029-2100); tabs = [left_is_rs1, rs1_value, left_is_pc, unexpanded_pc, right_is_rs2,
    rs2_value, right_is_imm, imm, eq_outer, eq_product] -> (4,4)"""
    arrs, ptrs = _ptrs(tabs)
    out = np.empty((4, 4), dtype=np.uint64)
    lib.zo_instruction_input_round(ptrs, C.c_size_t(arrs[0].shape[0]), _p(_c(gamma)), _p(_c(previous_claim)), _p(out))
    return out


def registers_cr_round(phase2, tabs, gamma, previous_claim):
    """RegistersClaimReductionProver.computeRoundEvalsPhase1 / Phase2 (:2334-2389) -> (3,4)"""
    arrs, ptrs = _ptrs(tabs)
    out = np.empty((3, 4), dtype=np.uint64)
    lib.zo_registers_cr_round(C.c_int(1 if phase2 else 0), ptrs, C.c_size_t(arrs[0].shape[0]), _p(_c(gamma)), _p(_c(previous_claim)), _p(out))
    return out


def eq_plus_one_mle(x, y):
    """EqPlusOnePolynomial.mle (src/poly/mod.zig:407-435)"""
    x, y = _c(np.asarray(x, dtype=np.uint64).reshape(-1, 4)), _c(np.asarray(y, dtype=np.uint64).reshape(-1, 4))
    out = np.empty(4, dtype=np.uint64)
    lib.zo_eq_plus_one_mle(_p(x), _p(y), C.c_size_t(x.shape[0]), _p(out))
    return out


def eq_plus_one_table(r):
    """computeEqPlusOneEvals (src/poly/mod.zig:530-548): the general formula at every cube point, as the reference does it"""
    r = _c(np.asarray(r, dtype=np.uint64).reshape(-1, 4))
    out = np.empty((1 << r.shape[0], 4), dtype=np.uint64)
    lib.zo_eq_plus_one_table(_p(r), C.c_size_t(r.shape[0]), _p(out))
    return out


# ---- RamReadWriteCheckingProver (src/zkvm/ram/read_write_checking.zig:160-1323), restated. The sparse cycle-major / address-major
# entry algebra runs on Python integers (canonical values mod r; the reference's field type is Montgomery, the values are the same),
# the dense tables (eq_evals, inc, val_init) are Montgomery arrays folded by the C oracle's bind_low, and the Gruen split-eq
# machinery is the GruenSplitEq restatement above.
_R_P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
_R_MONT = (1 << 256) % _R_P
_R_MONT_INV = pow(_R_MONT, -1, _R_P)


def fr_to_int(a):
    a = np.asarray(a, dtype=np.uint64).reshape(4)
    return (int(a[0]) | int(a[1]) << 64 | int(a[2]) << 128 | int(a[3]) << 192) * _R_MONT_INV % _R_P


def fr_from_int(v):
    m = (v % _R_P) * _R_MONT % _R_P
    return np.array([(m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


class RamReadWriteCheckingProver:
    """accesses: [(timestamp, address, is_write, value)] in trace order (MemoryTrace.accesses); initial_ram: {address: u64}.
    Entry = [cycle, address, ra_coeff, val_coeff, prev_val, next_val] (CycleMajorEntry, :91-157)."""

    def __init__(self, accesses, gamma, r_cycle, log_k, log_t, phase1_num_rounds, start_address, initial_claim, initial_ram=None):
        P = _R_P
        self.gamma = fr_to_int(gamma)
        self.r_cycle = _c(np.asarray(r_cycle, dtype=np.uint64).reshape(-1, 4)).copy()
        self.log_k, self.log_t, self.phase1_num_rounds, self.start_address = log_k, log_t, phase1_num_rounds, start_address
        K, T = 1 << log_k, 1 << log_t
        inc = [0] * T
        val_init = [0] * K
        cur = {}
        for addr, val in (initial_ram or {}).items():  # :212-231, :253-267
            if addr >= start_address:
                idx = (addr - start_address) // 8
                if idx < K:
                    val_init[idx] = val % P
                    cur[idx] = val
        self.entries = []
        for ts, address, is_write, value in accesses:  # :269-330
            if ts >= T or address < start_address:
                continue
            idx = (address - start_address) // 8
            if idx >= K:
                continue
            prev = cur.get(idx, 0)
            if is_write:
                inc[ts] = (value - prev) % P
                cur[idx] = value
            val_coeff = prev if is_write else value
            self.entries.append([ts, idx, 1, val_coeff % P, prev, value])
        self.entries.sort(key=lambda e: (e[0], e[1]))  # :333-340
        self.inc = np.stack([fr_from_int(v) for v in inc]) if T <= 4096 else self._bulk(inc)
        self.val_init = self._bulk(val_init)
        self.eq_evals = fr_eq_table(self.r_cycle) if log_t else f_from_u64(FR, np.array([1], dtype=np.uint64))  # :345-348 computeEqBigEndian
        self.eq_size = T
        self.gruen = GruenSplitEq(self.r_cycle)  # :354
        self.current_claim = fr_to_int(initial_claim)
        self.round = 0
        self.challenges = []

    @staticmethod
    def _bulk(vals):
        """list of canonical ints (mostly zero) -> Montgomery array"""
        out = np.zeros((len(vals), 4), dtype=np.uint64)
        for i, v in enumerate(vals):
            if v:
                out[i] = fr_from_int(v)
        return out

    def numRounds(self):
        return self.log_k + self.log_t

    def isComplete(self):
        return self.round >= self.numRounds()

    def _in_cycle_phase(self):
        p1 = self.phase1_num_rounds
        return self.round < p1 or self.round >= p1 + self.log_k

    def computeRoundPolynomialCubic(self):  # :391-408 -> (4, 4) Montgomery
        ev = self._phase1() if self._in_cycle_phase() else self._phase2()
        return ev

    def _phase1(self):  # :410-536
        P, gamma = _R_P, self.gamma
        g = self.gruen
        e_out, e_in, head_in_bits = g.getWindowEqTables(1)  # getWindowEqTables(current_index, 1)
        qc = qq = 0
        ents, i = self.entries, 0
        inc_len = self.inc.shape[0]
        cache_out, cache_in, cache_inc = {}, {}, {}

        def eo(x):
            if x not in cache_out:
                cache_out[x] = fr_to_int(e_out[x]) if x < e_out.shape[0] else 1
            return cache_out[x]

        def ei(x):
            if x not in cache_in:
                cache_in[x] = fr_to_int(e_in[x]) if x < e_in.shape[0] else 1
            return cache_in[x]

        def incv(j):
            if j not in cache_inc:
                cache_inc[j] = fr_to_int(self.inc[j]) if j < inc_len else 0
            return cache_inc[j]
        while i < len(ents):
            e = ents[i]
            pair = e[0] // 2
            e_prefix = eo(pair >> head_in_bits) * ei(pair & ((1 << head_in_bits) - 1)) % P
            inc_0, inc_1 = incv(2 * pair), incv(2 * pair + 1)
            inc_inf = (inc_1 - inc_0) % P
            if e[0] % 2 == 0:
                nxt = ents[i + 1] if i + 1 < len(ents) else None
                if nxt is not None and nxt[0] // 2 == pair and nxt[1] == e[1] and nxt[0] % 2 == 1:  # :465-484
                    ra_0, ra_inf = e[2], (nxt[2] - e[2]) % P
                    val_0, val_inf = e[3], (nxt[3] - e[3]) % P
                    i += 2
                else:  # :485-494 only the even entry: the odd one is implicit
                    ra_0, ra_inf = e[2], (-e[2]) % P
                    val_0, val_inf = e[3], (e[5] - e[3]) % P
                    i += 1
            else:  # :496-505 only the odd entry
                ra_0, ra_inf = 0, e[2]
                val_0, val_inf = e[4] % P, (e[3] - e[4]) % P
                i += 1
            inner_0 = (val_0 + gamma * (inc_0 + val_0)) % P  # :510-511
            inner_inf = (val_inf + gamma * (inc_inf + val_inf)) % P
            qc = (qc + e_prefix * ra_0 % P * inner_0) % P
            qq = (qq + e_prefix * ra_inf % P * inner_inf) % P
        self.last_q = (qc, qq)
        return g.computeCubicRoundPoly(fr_from_int(qc), fr_from_int(qq), fr_from_int(self.current_claim))

    def _eq_addr(self, address, addr_round):  # the loops at :785-794 etc.
        P = _R_P
        acc = 1
        for i in range(addr_round):
            r_i = self.challenges[self.phase1_num_rounds + i]
            acc = acc * (r_i if (address >> i) & 1 else (1 - r_i)) % P
        return acc

    def _pair_groups(self, addr_round):
        """entries (address-major order) grouped by column pair: yields (col_pair, even_list, odd_list)"""
        ents, i = self.entries, 0
        while i < len(ents):
            col_pair = (ents[i][1] >> addr_round) // 2
            j = i
            while j < len(ents) and (ents[j][1] >> addr_round) // 2 == col_pair:
                j += 1
            k = i
            while k < j and (ents[k][1] >> addr_round) % 2 == 0:
                k += 1
            yield col_pair, ents[i:k], ents[k:j]
            i = j

    def _phase2(self):  # :538-769
        P, gamma = _R_P, self.gamma
        addr_round = self.round - self.phase1_num_rounds
        if addr_round == 0:
            self.entries.sort(key=lambda e: (e[1], e[0]))  # :555-562
        eq_cycle, inc_s = fr_to_int(self.eq_evals[0]), fr_to_int(self.inc[0])
        size = (1 << self.log_k) >> addr_round
        s0 = s2 = 0
        opg = (1 + gamma) % P
        for col_pair, even, odd in self._pair_groups(addr_round):
            ec = fr_to_int(self.val_init[2 * col_pair]) if 2 * col_pair < size else 0
            oc = fr_to_int(self.val_init[2 * col_pair + 1]) if 2 * col_pair + 1 < size else 0
            a = b = 0

            def contrib(address, ra_0, ra_2, val_0, val_2):
                eq_partial = eq_cycle * self._eq_addr(address, addr_round) % P
                c0 = eq_partial * ra_0 % P * ((val_0 * opg + gamma * inc_s) % P) % P
                c2 = eq_partial * ra_2 % P * ((val_2 * opg + gamma * inc_s) % P) % P
                return c0, c2
            while a < len(even) or b < len(odd):
                ee = even[a] if a < len(even) else None
                oe = odd[b] if b < len(odd) else None
                if ee is not None and oe is not None and ee[0] == oe[0]:  # :771-812
                    c0, c2 = contrib(ee[1], ee[2], (2 * oe[2] - ee[2]) % P, ee[3], (2 * oe[3] - ee[3]) % P)
                    ec, oc = ee[5] % P, oe[5] % P
                    a += 1
                    b += 1
                elif oe is None or (ee is not None and ee[0] < oe[0]):  # :815-855 even only
                    c0, c2 = contrib(ee[1], ee[2], (-ee[2]) % P, ee[3], (2 * oc - ee[3]) % P)
                    ec = ee[5] % P
                    a += 1
                else:  # :858-899 odd only
                    c0, c2 = contrib(oe[1], 0, 2 * oe[2] % P, ec, (2 * oe[3] - ec) % P)
                    oc = oe[5] % P
                    b += 1
                s0, s2 = (s0 + c0) % P, (s2 + c2) % P
        s1 = (self.current_claim - s0) % P  # :752
        s3 = (3 * s2 - 3 * s1 + s0) % P  # :755
        return np.stack([fr_from_int(v) for v in (s0, s1, s2, s3)])

    def bindChallenge(self, challenge):  # :902-970
        P = _R_P
        ch = _c(challenge).copy()
        r = fr_to_int(ch)
        self.challenges.append(r)
        p1 = self.phase1_num_rounds
        if self._in_cycle_phase() and self.eq_size > 1:
            half = self.eq_size // 2
            self.eq_evals = fr_bind_low(self.eq_evals[:self.eq_size], ch)
            self.inc = fr_bind_low(self.inc[:self.eq_size], ch)
            self.eq_size = half
            self.gruen.bind(ch)
            self._bind_entries(r)
        if p1 <= self.round < p1 + self.log_k:
            addr_round = self.round - p1
            size = (1 << self.log_k) >> addr_round
            if size > 1:  # :953-959 folds IN PLACE: [0, size/2) are the new values, [size/2, size) keep the old ones ...
                self.val_init[:size // 2] = fr_bind_low(self.val_init[:size], ch)
            # ... and :962 reads its checkpoints from that array with the OLD size (:974-975), as restated here
            self._bind_entries_address_major(r, addr_round, self.val_init, size)
        self.round += 1

    def _bind_entries(self, r):  # :1139-1185 with CycleMajorEntry.bindEntries (:110-156)
        P = _R_P
        ents, out, i = self.entries, [], 0
        while i < len(ents):
            e = ents[i]
            if e[0] % 2 == 0:
                nxt = ents[i + 1] if i + 1 < len(ents) else None
                if nxt is not None and nxt[0] // 2 == e[0] // 2 and nxt[1] == e[1] and nxt[0] % 2 == 1:
                    out.append([e[0] // 2, e[1], (e[2] + r * (nxt[2] - e[2])) % P, (e[3] + r * (nxt[3] - e[3])) % P, e[4], nxt[5]])
                    i += 2
                    continue
                out.append([e[0] // 2, e[1], (1 - r) * e[2] % P, (e[3] + r * (e[5] - e[3])) % P, e[4], e[5]])
            else:
                out.append([e[0] // 2, e[1], r * e[2] % P, (e[4] + r * (e[3] - e[4])) % P, e[4], e[5]])
            i += 1
        self.entries = out

    def _bind_entries_address_major(self, r, addr_round, val_init, size):  # :973-1086, :1088-1137
        P = _R_P
        out = []
        for col_pair, even, odd in self._pair_groups(addr_round):
            ec = fr_to_int(val_init[2 * col_pair]) if 2 * col_pair < size else 0
            oc = fr_to_int(val_init[2 * col_pair + 1]) if 2 * col_pair + 1 < size else 0
            a = b = 0
            while a < len(even) or b < len(odd):
                ee = even[a] if a < len(even) else None
                oe = odd[b] if b < len(odd) else None
                if ee is not None and oe is not None and ee[0] == oe[0]:
                    out.append([ee[0], ee[1] // 2, (ee[2] + r * (oe[2] - ee[2])) % P, (ee[3] + r * (oe[3] - ee[3])) % P, ee[4], oe[5]])
                    ec, oc = ee[5] % P, oe[5] % P
                    a += 1
                    b += 1
                elif oe is None or (ee is not None and ee[0] < oe[0]):
                    out.append([ee[0], ee[1] // 2, (1 - r) * ee[2] % P, (ee[3] + r * (oc - ee[3])) % P, ee[4], ee[5]])
                    ec = ee[5] % P
                    a += 1
                else:
                    out.append([oe[0], oe[1] // 2, r * oe[2] % P, (ec + r * (oe[3] - ec)) % P, oe[4], oe[5]])
                    oc = oe[5] % P
                    b += 1
        self.entries = out

    def updateClaim(self, evals, challenge):  # :1187-1204 (Lagrange interpolation through 0, 1, 2, 3)
        P = _R_P
        e = [fr_to_int(x) for x in np.asarray(evals, dtype=np.uint64).reshape(4, 4)]
        c = fr_to_int(challenge)
        inv = lambda v: pow(v % P, -1, P)  # noqa: E731
        L0 = (c - 1) * (c - 2) % P * (c - 3) % P * inv(-6) % P
        L1 = c * (c - 2) % P * (c - 3) % P * inv(2) % P
        L2 = c * (c - 1) % P * (c - 3) % P * inv(-2) % P
        L3 = c * (c - 1) % P * (c - 2) % P * inv(6) % P
        self.current_claim = (e[0] * L0 + e[1] * L1 + e[2] * L2 + e[3] * L3) % P

    def getOpeningClaims(self, r_sumcheck):  # :1210-1322 -> (ra_claim, val_claim, inc_claim) Montgomery
        P = _R_P
        rs = [fr_to_int(x) for x in np.asarray(r_sumcheck, dtype=np.uint64).reshape(-1, 4)]
        log_k, log_t, p1 = self.log_k, self.log_t, self.phase1_num_rounds
        p2, p3 = p1 + log_k, log_t - p1
        r_address, r_cyc = [0] * log_k, [0] * log_t
        for i in range(min(log_k, max(len(rs) - p1, 0))):
            r_address[log_k - 1 - i] = rs[p1 + i]
        for i in range(min(p1, len(rs))):
            if p3 + (p1 - 1 - i) < log_t:
                r_cyc[p3 + (p1 - 1 - i)] = rs[i]
        for i in range(min(p3, max(len(rs) - p2, 0))):
            r_cyc[p3 - 1 - i] = rs[p2 + i]

        def eq(rv, x):  # computeEq, :1353-1366
            acc, n = 1, len(rv)
            for i in range(n):
                acc = acc * (rv[i] if (x >> (n - 1 - i)) & 1 else (1 - rv[i])) % P
            return acc
        ra = 0
        val = fr_to_int(self.val_init[0])
        for e in self.entries:
            w = eq(r_address, e[1]) * eq(r_cyc, e[0]) % P
            ra = (ra + w * e[2]) % P
            val = (val + w * (e[3] - fr_to_int(self.val_init[e[1]]))) % P
        return fr_from_int(ra), fr_from_int(val), self.inc[0].copy()


# ---- Stage4GruenProver — RegistersReadWriteChecking (src/zkvm/spartan/stage4_gruen_prover.zig:65-1240, with src/zkvm/spartan/gruen_eq.zig:
# the same prefix-table split-eq structure as src/poly/split_eq.zig; E_in_current / E_out_current = getWindowEqTables(1), gruenPolyDeg3's
# four evaluations = computeCubicRoundPoly). Dense K x T tables (K = 128 registers), restated with the C oracle's vector field ops.
def _fmul(a, b):
    sh = np.broadcast(a[..., 0], b[..., 0]).shape
    a2 = np.ascontiguousarray(np.broadcast_to(a, sh + (4,))).reshape(-1, 4)
    b2 = np.ascontiguousarray(np.broadcast_to(b, sh + (4,))).reshape(-1, 4)
    return f_mul(FR, a2, b2).reshape(sh + (4,))


def _fadd(a, b):
    sh = np.broadcast(a[..., 0], b[..., 0]).shape
    a2 = np.ascontiguousarray(np.broadcast_to(a, sh + (4,))).reshape(-1, 4)
    b2 = np.ascontiguousarray(np.broadcast_to(b, sh + (4,))).reshape(-1, 4)
    return f_add(FR, a2, b2).reshape(sh + (4,))


def _fsub(a, b):
    sh = np.broadcast(a[..., 0], b[..., 0]).shape
    a2 = np.ascontiguousarray(np.broadcast_to(a, sh + (4,))).reshape(-1, 4)
    b2 = np.ascontiguousarray(np.broadcast_to(b, sh + (4,))).reshape(-1, 4)
    return f_sub(FR, a2, b2).reshape(sh + (4,))


def _fsum(a):
    """sum of all elements of an (..., 4) array -> (4,)"""
    v = np.ascontiguousarray(a).reshape(-1, 4)
    if v.shape[0] == 0:
        return np.zeros(4, dtype=np.uint64)
    while v.shape[0] > 1:
        if v.shape[0] % 2:
            v = np.concatenate([v, np.zeros((1, 4), dtype=np.uint64)])
        v = f_add(FR, np.ascontiguousarray(v[0::2]), np.ascontiguousarray(v[1::2]))
    return v[0]


def _fsum_axis0(a):
    """(K, n, 4) -> (n, 4): sum over the first axis"""
    v = np.ascontiguousarray(a)
    while v.shape[0] > 1:
        if v.shape[0] % 2:
            v = np.concatenate([v, np.zeros((1,) + v.shape[1:], dtype=np.uint64)])
        v = _fadd(v[0::2], v[1::2])
    return v[0]


class Stage4GruenProver:
    """steps: [(instruction u32, rd_value u64, is_noop)] (ExecutionTrace.steps); r_cycle in ROUND order (r_cycle[0] bound first, :283-288).
    Tables are (K, T, 4) Montgomery arrays with the reference's [k * T + j] indexing; the live region is [:current_K, :current_T]."""
    LOG_K, K = 7, 128

    def __init__(self, steps, gamma, r_cycle, phase1_num_rounds, phase2_num_rounds):
        n = len(steps)
        T = 1
        while T < n:
            T *= 2
        self.T, self.log_T = T, T.bit_length() - 1
        r_cycle = _c(np.asarray(r_cycle, dtype=np.uint64).reshape(-1, 4))
        assert r_cycle.shape[0] == self.log_T
        self.num_rounds = self.LOG_K + self.log_T
        self.gamma = _c(gamma).copy()
        self.gamma_sq = f_mul(FR, self.gamma.reshape(1, 4), self.gamma.reshape(1, 4))[0]
        K = self.K
        one = f_from_u64(FR, np.array([1], dtype=np.uint64))[0]
        val_u = np.zeros((K, T), dtype=np.uint64)
        wa = np.zeros((K, T, 4), dtype=np.uint64)
        ra = np.zeros((K, T, 4), dtype=np.uint64)
        r1 = np.zeros((K, T, 4), dtype=np.uint64)
        r2 = np.zeros((K, T, 4), dtype=np.uint64)
        inc_pre, inc_post, inc_set = np.zeros(T, dtype=np.uint64), np.zeros(T, dtype=np.uint64), np.zeros(T, dtype=bool)
        regs = [0] * 32
        for cycle, (instr, rd_value, is_noop) in enumerate(steps):  # :183-246
            val_u[:32, cycle] = regs
            if is_noop:
                continue
            rd, rs1, rs2, opcode = (instr >> 7) & 31, (instr >> 15) & 31, (instr >> 20) & 31, instr & 0x7F
            if opcode in (0x13, 0x03, 0x67, 0x1B, 0x33, 0x3B, 0x23, 0x63):
                r1[rs1, cycle] = one
                ra[rs1, cycle] = f_add(FR, ra[rs1, cycle].reshape(1, 4), self.gamma.reshape(1, 4))[0]
            if opcode in (0x33, 0x3B, 0x23, 0x63):
                r2[rs2, cycle] = one
                ra[rs2, cycle] = f_add(FR, ra[rs2, cycle].reshape(1, 4), self.gamma_sq.reshape(1, 4))[0]
            if opcode not in (0x23, 0x63) and rd != 0:
                wa[rd, cycle] = one
                inc_pre[cycle], inc_post[cycle], inc_set[cycle] = regs[rd], rd_value, True
                regs[rd] = rd_value
        for cycle in range(n, T):  # :249-258 padding cycles keep the final register file
            val_u[:32, cycle] = regs
        self.val = f_from_u64(FR, val_u.reshape(-1)).reshape(K, T, 4)
        self.inc = f_sub(FR, f_from_u64(FR, inc_post), f_from_u64(FR, inc_pre))  # F.fromU64(post) - F.fromU64(pre), zero where unset
        self.wa, self.ra, self.rs1_ra, self.rs2_ra = wa, ra, r1, r2
        self.gruen = GruenSplitEq(r_cycle[::-1].copy())  # big-endian for the split-eq structure (:283-288)
        self.current_T, self.current_K = T, K
        self.phase1_num_rounds, self.phase2_num_rounds = phase1_num_rounds, phase2_num_rounds
        self.merged_eq = None

    def computeInputClaim(self):  # :385-431 without Stage-3 claims: sum over (k, j) of eq(r_cycle, j) * combined
        eq = self.gruen.getFullEqTable()
        comb = _fadd(_fmul(self.ra, self.val), _fmul(self.wa, _fadd(self.val, self.inc[None, :, :])))
        return _fsum(_fmul(_fsum_axis0(comb), eq))

    def _live(self):
        K, T = self.current_K, self.current_T
        return self.ra[:K, :T], self.wa[:K, :T], self.val[:K, :T], self.inc[:T]

    def computeRoundEvals(self, rnd, current_claim):  # :1165-1190 -> (4, 4): p(0..3)
        p1, p2 = self.phase1_num_rounds, self.phase2_num_rounds
        claim = _c(current_claim)
        ra, wa, val, inc = self._live()
        two = f_from_u64(FR, np.array([2], dtype=np.uint64))[0]
        three = f_from_u64(FR, np.array([3], dtype=np.uint64))[0]
        if rnd < p1:  # phase1ComputeMessage :561-741
            e_out, e_in, head_in_bits = self.gruen.getWindowEqTables(1)
            half = self.current_T // 2
            i = np.arange(half)
            x_in, x_out = i & ((1 << head_in_bits) - 1), i >> head_in_bits
            one = f_from_u64(FR, np.array([1], dtype=np.uint64))
            eo = np.where((x_out < len(e_out))[:, None], _c(e_out)[np.minimum(x_out, len(e_out) - 1)], one)
            ei = np.where((x_in < len(e_in))[:, None], _c(e_in)[np.minimum(x_in, len(e_in) - 1)], one)
            E = _fmul(eo, ei)
            inc0, incs = inc[0::2], _fsub(inc[1::2], inc[0::2])
            rae, was, vae = ra[:, 0::2], wa[:, 0::2], val[:, 0::2]
            ras, wss, vas = _fsub(ra[:, 1::2], rae), _fsub(wa[:, 1::2], was), _fsub(val[:, 1::2], vae)
            c0 = _fadd(_fmul(rae, vae), _fmul(was, _fadd(vae, inc0[None])))
            cx = _fadd(_fmul(ras, vas), _fmul(wss, _fadd(vas, incs[None])))
            q0 = _fsum(_fmul(_fsum_axis0(c0), E))
            qx = _fsum(_fmul(_fsum_axis0(cx), E))
            self.last_q = (q0, qx)
            return self.gruen.computeCubicRoundPoly(q0, qx, claim)
        if rnd < p1 + p2 or self.current_T == 1:  # phase2ComputeMessage :764-852; phase 3 with no cycle left :955-1013
            eq = self.merged_eq[:self.current_T]
            rae, rao = ra[0::2], ra[1::2]
            wae, wao = wa[0::2], wa[1::2]
            vae, vao = val[0::2], val[1::2]
            c0 = _fadd(_fmul(rae, vae), _fmul(wae, _fadd(vae, inc[None])))
            ra2 = _fadd(rae, _fmul(two, _fsub(rao, rae)))
            wa2 = _fadd(wae, _fmul(two, _fsub(wao, wae)))
            va2 = _fadd(vae, _fmul(two, _fsub(vao, vae)))
            c2 = _fadd(_fmul(ra2, va2), _fmul(wa2, _fadd(va2, inc[None])))
            e0 = _fsum(_fmul(_fsum_axis0(c0), eq))
            e2 = _fsum(_fmul(_fsum_axis0(c2), eq))
            e1 = f_sub(FR, claim.reshape(1, 4), e0.reshape(1, 4))[0]
            # quadratic through (0, e0), (1, e1), (2, e2): p(3) = e0 - 3 e1 + 3 e2 (c3 = 0, :841-850)
            e3 = _fadd(_fsub(e0, _fmul(three, e1)), _fmul(three, e2))
            return np.stack([e0, e1, e2, e3])
        # phase3ComputeMessage with cycles remaining :854-953
        eq = self.merged_eq[:self.current_T]
        eqe, eqs = eq[0::2], _fsub(eq[1::2], eq[0::2])
        ince, incs = inc[0::2], _fsub(inc[1::2], inc[0::2])
        rae, was, vae = ra[:, 0::2], wa[:, 0::2], val[:, 0::2]
        ras, wss, vas = _fsub(ra[:, 1::2], rae), _fsub(wa[:, 1::2], was), _fsub(val[:, 1::2], vae)
        out = []
        for t, tf in ((0, None), (2, two), (3, three)):
            if tf is None:
                r_, w_, v_, i_, q_ = rae, was, vae, ince, eqe
            else:
                r_, w_, v_ = _fadd(rae, _fmul(tf, ras)), _fadd(was, _fmul(tf, wss)), _fadd(vae, _fmul(tf, vas))
                i_, q_ = _fadd(ince, _fmul(tf, incs)), _fadd(eqe, _fmul(tf, eqs))
            inner = _fsum_axis0(_fadd(_fmul(r_, v_), _fmul(w_, _fadd(v_, i_[None]))))
            out.append(_fsum(_fmul(q_, inner)))
        e0, e2, e3 = out
        e1 = f_sub(FR, claim.reshape(1, 4), e0.reshape(1, 4))[0]
        return np.stack([e0, e1, e2, e3])

    def bindChallenge(self, rnd, challenge):  # bindPolynomials :1047-1163
        ch = _c(challenge).reshape(1, 4)
        p1, p2 = self.phase1_num_rounds, self.phase2_num_rounds
        K, T = self.current_K, self.current_T

        def fold_cycle(t):
            lo, hi = t[:K, 0:T:2], t[:K, 1:T:2]
            t[:K, :T // 2] = _fadd(lo, _fmul(ch, _fsub(hi, lo)))  # lo (1 - c) + hi c

        def fold_vec(v):
            lo, hi = v[0:T:2], v[1:T:2]
            v[:T // 2] = _fadd(lo, _fmul(ch, _fsub(hi, lo)))
        if rnd < p1 or rnd >= p1 + p2:
            for t in (self.val, self.wa, self.ra, self.rs1_ra, self.rs2_ra):
                fold_cycle(t)
            fold_vec(self.inc)
            if rnd >= p1 + p2 and self.merged_eq is not None:
                fold_vec(self.merged_eq)
            self.current_T = T // 2
            if rnd < p1:
                self.gruen.bind(ch[0])
                if rnd == p1 - 1:  # gruen_eq.merge (:119-146)
                    self.merged_eq = self.gruen.getFullEqTable().copy()
        else:
            for t in (self.val, self.wa, self.ra, self.rs1_ra, self.rs2_ra):
                lo, hi = t[0:K:2, :T], t[1:K:2, :T]
                t[:K // 2, :T] = _fadd(lo, _fmul(ch, _fsub(hi, lo)))
            self.current_K = K // 2

    def getFinalClaims(self):  # :1219-1236
        return {"val_claim": self.val[0, 0].copy(), "rs1_ra_claim": self.rs1_ra[0, 0].copy(), "rs2_ra_claim": self.rs2_ra[0, 0].copy(),
                "rd_wa_claim": self.wa[0, 0].copy(), "inc_claim": self.inc[0].copy()}

    def finalCheck(self):
        """(eq_scalar, combined, expected) as bindChallenge prints them after the last round (:1196-1210)"""
        eq = self.merged_eq[0]
        comb = _fadd(_fmul(self.ra[0, 0], self.val[0, 0]), _fmul(self.wa[0, 0], _fadd(self.val[0, 0], self.inc[0])))
        return eq, comb, _fmul(eq, comb)


# ---------------------------------------------------------------- Spartan outer sumcheck, remaining rounds (linear phase)
# StreamingOuterProver (src/zkvm/spartan/streaming_outer.zig) after the UniSkip first round: Az / Bz of the two constraint groups are
# materialised per cycle, then every round is Gruen's (t'(0), t'(inf)) over adjacent pairs under the split-eq weights and a low-to-high
# fold. The 19 uniform constraints are the published Jolt R1CS (src/zkvm/r1cs/constraints.zig:248-531), restated here as data:
# condition * (left - right) = 0, each side a linear combination of the 43 per-cycle inputs (:39-92) plus a constant.
R1CS_INPUT_NAMES = [  # R1CSInputIndex (:39-92)
    "LeftInstructionInput", "RightInstructionInput", "Product", "WriteLookupOutputToRD", "WritePCtoRD", "ShouldBranch", "PC", "UnexpandedPC",
    "Imm", "RamAddress", "Rs1Value", "Rs2Value", "RdWriteValue", "RamReadValue", "RamWriteValue", "LeftLookupOperand", "RightLookupOperand",
    "NextUnexpandedPC", "NextPC", "NextIsVirtual", "NextIsFirstInSequence", "LookupOutput", "ShouldJump", "FlagAddOperands",
    "FlagSubtractOperands", "FlagMultiplyOperands", "FlagLoad", "FlagStore", "FlagJump", "FlagWriteLookupOutputToRD", "FlagVirtualInstruction",
    "FlagAssert", "FlagDoNotUpdateUnexpandedPC", "FlagAdvice", "FlagIsCompressed", "FlagIsFirstInSequence", "FlagIsRdNotZero", "FlagBranch",
    "FlagIsNoop", "FlagLeftOperandIsRs1", "FlagLeftOperandIsPC", "FlagRightOperandIsRs2", "FlagRightOperandIsImm"]
NUM_R1CS_INPUTS = len(R1CS_INPUT_NAMES)  # 43
_R1 = {n: i for i, n in enumerate(R1CS_INPUT_NAMES)}


def _lc(const=0, **terms):
    return ([(_R1[k], v) for k, v in terms.items()], const)


UNIFORM_CONSTRAINTS = [  # (condition, left, right), constraints.zig:248-531 in order
    (_lc(FlagLoad=1, FlagStore=1), _lc(RamAddress=1), _lc(Rs1Value=1, Imm=1)),                                           # 0
    (_lc(1, FlagLoad=-1, FlagStore=-1), _lc(RamAddress=1), _lc()),                                                      # 1
    (_lc(FlagLoad=1), _lc(RamReadValue=1), _lc(RamWriteValue=1)),                                                       # 2
    (_lc(FlagLoad=1), _lc(RamReadValue=1), _lc(RdWriteValue=1)),                                                        # 3
    (_lc(FlagStore=1), _lc(Rs2Value=1), _lc(RamWriteValue=1)),                                                          # 4
    (_lc(FlagAddOperands=1, FlagSubtractOperands=1, FlagMultiplyOperands=1), _lc(LeftLookupOperand=1), _lc()),          # 5
    (_lc(1, FlagAddOperands=-1, FlagSubtractOperands=-1, FlagMultiplyOperands=-1), _lc(LeftLookupOperand=1), _lc(LeftInstructionInput=1)),  # 6
    (_lc(FlagAddOperands=1), _lc(RightLookupOperand=1), _lc(LeftInstructionInput=1, RightInstructionInput=1)),          # 7
    (_lc(FlagSubtractOperands=1), _lc(RightLookupOperand=1), _lc(1 << 64, LeftInstructionInput=1, RightInstructionInput=-1)),  # 8
    (_lc(FlagMultiplyOperands=1), _lc(RightLookupOperand=1), _lc(Product=1)),                                           # 9
    (_lc(1, FlagAddOperands=-1, FlagSubtractOperands=-1, FlagMultiplyOperands=-1, FlagAdvice=-1), _lc(RightLookupOperand=1), _lc(RightInstructionInput=1)),  # 10
    (_lc(FlagAssert=1), _lc(LookupOutput=1), _lc(1)),                                                                   # 11
    (_lc(WriteLookupOutputToRD=1), _lc(RdWriteValue=1), _lc(LookupOutput=1)),                                           # 12
    (_lc(WritePCtoRD=1), _lc(RdWriteValue=1), _lc(4, UnexpandedPC=1, FlagIsCompressed=-2)),                             # 13
    (_lc(ShouldJump=1), _lc(NextUnexpandedPC=1), _lc(LookupOutput=1)),                                                  # 14
    (_lc(ShouldBranch=1), _lc(NextUnexpandedPC=1), _lc(UnexpandedPC=1, Imm=1)),                                         # 15
    (_lc(1, ShouldBranch=-1, FlagJump=-1), _lc(NextUnexpandedPC=1), _lc(4, UnexpandedPC=1, FlagDoNotUpdateUnexpandedPC=-4, FlagIsCompressed=-2)),  # 16
    (_lc(FlagVirtualInstruction=1), _lc(NextPC=1), _lc(1, PC=1)),                                                       # 17
    (_lc(NextIsVirtual=1, NextIsFirstInSequence=-1), _lc(1), _lc(FlagDoNotUpdateUnexpandedPC=1)),                       # 18
]
FIRST_GROUP_INDICES = [1, 2, 3, 4, 5, 6, 11, 14, 17, 18]  # :537-548
SECOND_GROUP_INDICES = [0, 7, 8, 9, 10, 12, 13, 15, 16]  # :553-563


def lagrange_evals_symmetric(r, size=10):
    """L_i(r), i < size, over the symmetric domain {-(size-1)/2, ...} = {-4..5} for size 10 (computeLagrangeEvalsAtR0,
    streaming_outer.zig:1157-1213; LagrangePoly.evals of r1cs/univariate_skip.zig) -> (size, 4) Montgomery"""
    rv, start = fr_to_int(r), -((size - 1) // 2)
    out = []
    for i in range(size):
        num = den = 1
        for j in range(size):
            if j != i:
                num = num * (rv - (start + j)) % _R_P
                den = den * (i - j) % _R_P
        out.append(fr_from_int(num * pow(den, _R_P - 2, _R_P) % _R_P))
    return np.stack(out)


def lagrange_kernel(x, y, size=10):
    """K(x, y) = sum_i L_i(x) L_i(y) (LagrangePoly.lagrangeKernel, r1cs/univariate_skip.zig:296-312)"""
    a, b = lagrange_evals_symmetric(x, size), lagrange_evals_symmetric(y, size)
    return fr_from_int(sum(fr_to_int(u) * fr_to_int(v) for u, v in zip(a, b)) % _R_P)


def _lc_eval(lc, w):
    """LinearCombination.evaluate (constraints.zig:183-198) over all cycles at once: w (n, 43, 4) -> (n, 4)"""
    terms, const = lc
    n = w.shape[0]
    acc = np.repeat(fr_from_int(const % _R_P).reshape(1, 4), n, axis=0)
    for idx, coeff in terms:
        scaled = _fmul(w[:, idx], fr_from_int(abs(coeff)))
        acc = _fadd(acc, scaled) if coeff >= 0 else _fsub(acc, scaled)
    return acc


class StreamingOuterProver:
    """the remaining rounds of StreamingOuterProver (streaming_outer.zig:120-212 init, 1135-1155 bindFirstRoundChallenge, 258-372
    materializeLinearPhasePolynomials, 381-465 buildTPrimePoly, 481-491 computeTEvals, 1215-1281 computeRemainingRoundPoly, 1681-1717
    bindRemainingRoundChallenge, 1723-1737 updateClaim). cycle_witnesses: (n, 43, 4); tau: num_cycle_vars + 2 challenges."""

    def __init__(self, cycle_witnesses, tau, lagrange_tau_r0=None):
        self.cycle_witnesses = _c(np.asarray(cycle_witnesses, dtype=np.uint64).reshape(-1, NUM_R1CS_INPUTS, 4))
        n = self.cycle_witnesses.shape[0]
        assert n > 0
        self.padded_trace_len = 1
        while self.padded_trace_len < n:
            self.padded_trace_len *= 2
        self.num_cycle_vars = self.padded_trace_len.bit_length() - 1
        tau = _c(np.asarray(tau, dtype=np.uint64).reshape(-1, 4))
        self.tau_high = tau[-1].copy()
        self.full_tau = tau.copy()
        self.split_eq = GruenSplitEq(tau[:-1], lagrange_tau_r0)  # tau_low (:165-167)
        self.current_claim = fr_from_int(0)
        self.current_round = 0
        self.challenges = []
        self.lagrange_evals_r0 = None
        self.az = self.bz = None

    def numRounds(self):  # :236-238
        return 1 + self.num_cycle_vars

    def bindFirstRoundChallenge(self, r0, uni_skip_claim):  # :1135-1155 (r0 is NOT bound in split_eq)
        self.current_round = 1
        self.current_claim = _c(uni_skip_claim).copy()
        self.lagrange_evals_r0 = lagrange_evals_symmetric(r0, 10)

    def materializeLinearPhasePolynomials(self):  # :258-372
        e_out, e_in, _ = self.split_eq.getWindowEqTables(1)
        poly_size = len(e_out) * len(e_in) * 2
        w = self.cycle_witnesses
        n = min(w.shape[0], poly_size // 2)
        zero = np.zeros((poly_size, 4), dtype=np.uint64)
        az, bz = zero.copy(), zero.copy()
        for g, group in enumerate((FIRST_GROUP_INDICES, SECOND_GROUP_INDICES[:10])):
            a = np.zeros((n, 4), dtype=np.uint64)
            b = np.zeros((n, 4), dtype=np.uint64)
            for t, ci in enumerate(group):
                cond, left, right = UNIFORM_CONSTRAINTS[ci]
                wt = self.lagrange_evals_r0[t]
                a = _fadd(a, _fmul(_lc_eval(cond, w[:n]), wt))
                b = _fadd(b, _fmul(_fsub(_lc_eval(left, w[:n]), _lc_eval(right, w[:n])), wt))
            az[g:2 * n:2], bz[g:2 * n:2] = a, b
        self.az, self.bz = az, bz

    def _t_evals(self):  # buildTPrimePoly with window 1 (:381-465) projected by E_active = [1] (:481-491)
        e_out, e_in, _ = self.split_eq.getWindowEqTables(1)
        e_out, e_in = _c(e_out), _c(e_in)
        bits = (len(e_in).bit_length() - 1) if len(e_in) > 1 else 0
        n_i = len(e_out) * len(e_in)
        i = np.arange(n_i)
        live = self.az.shape[0]
        pad = lambda t: np.concatenate([t, np.zeros((max(2 * n_i - live, 0), 4), dtype=np.uint64)])[:2 * n_i]
        az, bz = pad(self.az), pad(self.bz)
        weight = _fmul(e_out[i >> bits], e_in[i & ((1 << bits) - 1)])
        t0 = _fsum(_fmul(_fmul(az[0::2], bz[0::2]), weight))
        tinf = _fsum(_fmul(_fmul(_fsub(az[1::2], az[0::2]), _fsub(bz[1::2], bz[0::2])), weight))
        return t0, tinf

    def computeRemainingRoundPoly(self):  # :1215-1281
        if self.current_round == 1 and self.az is None:
            self.materializeLinearPhasePolynomials()
        t0, tinf = self._t_evals()
        self.last_t = (t0, tinf)
        return self.split_eq.computeCubicRoundPoly(t0, tinf, self.current_claim)

    def bindRemainingRoundChallenge(self, r):  # :1681-1717: split_eq first, then Az / Bz low-to-high
        self.challenges.append(_c(r).copy())
        self.split_eq.bind(r)
        self.az, self.bz = fr_bind_low(self.az, r), fr_bind_low(self.bz, r)
        self.current_round += 1

    def updateClaim(self, round_poly, challenge):  # :1723-1737: the cubic through the four evaluations at the challenge
        self.current_claim = raf_update_claim(round_poly, challenge)

    def getFinalEval(self):  # :1740-1742
        return self.current_claim


# ---------------------------------------------------------------- MultiStageProver stages 5 and 6 (src/zkvm/prover.zig:818-1112)
def compute_reg_eq(r, reg):
    """computeRegEq (prover.zig:961-972): prod_i (bit_i(reg) ? r[i] : 1 - r[i]), bit i <-> r[i]"""
    one = fr_from_int(1)
    acc = one
    for i, ri in enumerate(_c(r).reshape(-1, 4)):
        acc = _fmul(acc, ri if (reg >> i) & 1 else _fsub(one, ri))
    return acc.reshape(4)


def _high_half_rounds(evals, num_rounds, transcript, label):
    """the round loop stages 5 and 6 share (:902-944, 1055-1097): p(0), p(1) = sums of the two halves, the proof keeps [p(0), p(2) = 2 p(1) -
    p(0)], challengeScalar(label), f[j] = (1 - r) f[j] + r f[j + half], claim = (1 - r) p(0) + r p(1)"""
    cur = _c(evals).reshape(-1, 4).copy()
    one = fr_from_int(1)
    polys, chals, claims = [], [], []
    for _ in range(num_rounds):
        p0, p1 = fr_sum_halves(cur)
        polys.append(np.stack([p0, _fsub(_fadd(p1, p1), p0)]))
        ch = transcript.challenge_scalar(label)
        chals.append(ch)
        omr = _fsub(one, ch)
        half = cur.shape[0] // 2
        cur = _fadd(_fmul(cur[:half], omr), _fmul(cur[half:], ch))
        claims.append(_fadd(_fmul(omr, p0), _fmul(ch, p1)).reshape(4))
    z2, z1 = np.zeros((0, 2, 4), dtype=np.uint64), np.zeros((0, 4), dtype=np.uint64)
    return (np.stack(polys) if polys else z2, np.stack(chals) if chals else z1, np.stack(claims) if claims else z1,
            cur[0].copy() if cur.shape[0] else np.zeros(4, dtype=np.uint64))


def stage5_prove(instructions, log_t, transcript):
    """proveStage5 (prover.zig:829-958): r_register (5) and r_cycle_reg (log_t) challenges, eq_evals[j] = eq(r_register, rd(j)) over the
    trace steps (zero past them), initial claim = their sum, log2_ceil(trace_len) rounds. -> dict; None rounds for an empty trace"""
    r_register = np.stack([transcript.challenge_scalar(b"r_register") for _ in range(5)])
    r_cycle_reg = [transcript.challenge_scalar(b"r_cycle_reg") for _ in range(log_t)]
    instr = np.asarray(instructions, dtype=np.uint32)
    n_steps = len(instr)
    if n_steps == 0:
        return {"r_register": r_register, "r_cycle_reg": r_cycle_reg, "initial_claim": None}
    num_rounds = 0 if n_steps <= 1 else (n_steps - 1).bit_length()
    n = 1 << num_rounds
    table = np.stack([compute_reg_eq(r_register, reg) for reg in range(32)])
    eq_evals = np.zeros((n, 4), dtype=np.uint64)
    eq_evals[:n_steps] = table[(instr >> 7) & 31]
    polys, chals, claims, fin = _high_half_rounds(eq_evals, num_rounds, transcript, b"reg_eval_round")
    return {"r_register": r_register, "r_cycle_reg": r_cycle_reg, "initial_claim": _fsum(eq_evals), "round_polys": polys, "challenges": chals,
            "claims": claims, "final_claim": fin}


def stage6_prove(trace_len, transcript):
    """proveStage6 (prover.zig:990-1112): the booleanity batching challenge, violation_evals = 0 for every step (a valid trace is assumed,
    :1024-1033), the same round loop under "bool_round" """
    bool_challenge = transcript.challenge_scalar(b"booleanity")
    if trace_len == 0:
        return {"bool_challenge": bool_challenge, "initial_claim": None}
    num_rounds = 0 if trace_len <= 1 else (trace_len - 1).bit_length()
    viol = np.zeros((1 << num_rounds, 4), dtype=np.uint64)
    polys, chals, claims, fin = _high_half_rounds(viol, num_rounds, transcript, b"bool_round")
    return {"bool_challenge": bool_challenge, "initial_claim": np.zeros(4, dtype=np.uint64), "round_polys": polys, "challenges": chals,
            "claims": claims, "final_claim": fin}


class Stage4Prover(Stage4GruenProver):
    """the original Stage4Prover (src/zkvm/spartan/stage4_prover.zig:74-865): the same tables from the same trace rules (:183-277), a DENSE
    eq table over the cycles from the start (computeEqEvalsBE of the reversed r_cycle, :279-292), all log_T cycle rounds first and then the
    seven register rounds, every round's four evaluations computed directly from the tables (:601-723) — p(1) is not taken from the claim —
    and the full-coefficient round polynomial (:731-758)."""

    def __init__(self, steps, gamma, r_cycle, stage3_claims=None, batching_coeff=None):
        log_t = max(len(steps) - 1, 0).bit_length()
        super().__init__(steps, gamma, r_cycle, max(log_t, 1), self.LOG_K)
        self.eq_cycle_evals = self.gruen.getFullEqTable().copy()  # eq(r_cycle_be, .), r_cycle_be[0] <-> MSB
        self.stage3_claims = stage3_claims  # (rd_write_value, rs1_value, rs2_value) or None
        self.batching_coeff = fr_from_int(1) if batching_coeff is None else _c(batching_coeff).copy()

    def computeInputClaim(self):  # :569-599
        comb = _fadd(_fmul(self.ra, self.val), _fmul(self.wa, _fadd(self.val, self.inc[None, :, :])))
        return _fsum(_fmul(_fsum_axis0(comb), self.eq_cycle_evals))

    def computeRoundEvals(self, rnd, current_claim=None):  # :601-723 -> p(0..3)
        ra, wa, val, inc = self._live()
        eq = self.eq_cycle_evals[:self.current_T]
        ts = [fr_from_int(t) for t in range(4)]
        lerp = lambda a, b, t: _fadd(a, _fmul(t, _fsub(b, a)))  # (1 - t) a + t b
        out = []
        if rnd < self.log_T:
            for t in ts:
                r_, w_, v_ = lerp(ra[:, 0::2], ra[:, 1::2], t), lerp(wa[:, 0::2], wa[:, 1::2], t), lerp(val[:, 0::2], val[:, 1::2], t)
                i_, q_ = lerp(inc[0::2], inc[1::2], t), lerp(eq[0::2], eq[1::2], t)
                out.append(_fsum(_fmul(q_, _fsum_axis0(_fadd(_fmul(r_, v_), _fmul(w_, _fadd(v_, i_[None])))))))
        else:
            for t in ts:
                r_, w_, v_ = lerp(ra[0::2], ra[1::2], t), lerp(wa[0::2], wa[1::2], t), lerp(val[0::2], val[1::2], t)
                out.append(_fsum(_fmul(eq, _fsum_axis0(_fadd(_fmul(r_, v_), _fmul(w_, _fadd(v_, inc[None])))))))
        return np.stack(out)

    def computeRoundPolynomial(self, rnd, current_claim=None):  # :731-758 -> coefficients c0..c3
        P = _R_P
        e = [fr_to_int(x) for x in self.computeRoundEvals(rnd, current_claim)]
        c3 = (-e[0] + 3 * e[1] - 3 * e[2] + e[3]) * pow(6, P - 2, P) % P
        c2 = (2 * e[0] - 5 * e[1] + 4 * e[2] - e[3]) * pow(2, P - 2, P) % P
        c1 = (e[1] - e[0] - c2 - c3) % P
        return np.stack([fr_from_int(v) for v in (e[0], c1, c2, c3)])

    def bindChallenge(self, rnd, challenge):  # bindPolynomials :779-839
        ch = _c(challenge).reshape(1, 4)
        K, T = self.current_K, self.current_T
        if rnd < self.log_T:
            for t in (self.val, self.wa, self.ra, self.rs1_ra, self.rs2_ra):
                lo, hi = t[:K, 0:T:2], t[:K, 1:T:2]
                t[:K, :T // 2] = _fadd(lo, _fmul(ch, _fsub(hi, lo)))
            for v in (self.inc, self.eq_cycle_evals):
                lo, hi = v[0:T:2], v[1:T:2]
                v[:T // 2] = _fadd(lo, _fmul(ch, _fsub(hi, lo)))
            self.current_T = T // 2
        else:
            for t in (self.val, self.wa, self.ra, self.rs1_ra, self.rs2_ra):
                lo, hi = t[0:K:2, :T], t[1:K:2, :T]
                t[:K // 2, :T] = _fadd(lo, _fmul(ch, _fsub(hi, lo)))
            self.current_K = K // 2

    def finalCheck(self):  # the values prove() prints at the end (:533-553)
        eq = self.eq_cycle_evals[0]
        comb = _fadd(_fmul(self.ra[0, 0], self.val[0, 0]), _fmul(self.wa[0, 0], _fadd(self.val[0, 0], self.inc[0])))
        return eq, comb, _fmul(eq, comb)

    def prove(self, transcript):
        """prove (:395-567) with a Blake2bTranscript (any object with appendScalar / challengeScalar: the transcript is host logic with its own
        fixture, tests/golden/blake2b_transcript_preamble.json): batched compressed coefficients appended, challenges squeezed, batched claims tracked
        -> dict(round_polys (rounds, 4, 4) batched coefficients, challenges, final claims)"""
        P = _R_P
        b = fr_to_int(self.batching_coeff)
        if self.stage3_claims is not None:  # :404-437
            g = fr_to_int(self.gamma)
            rd, r1, r2 = (fr_to_int(x) for x in self.stage3_claims)
            unbatched = (rd + g * r1 + g * g * r2) % P
        else:
            unbatched = fr_to_int(self.computeInputClaim())
        claim = unbatched * b % P
        polys, chals = [], []
        for rnd in range(self.num_rounds):
            c = [fr_to_int(x) for x in self.computeRoundPolynomial(rnd, fr_from_int(claim))]
            for i in (0, 2, 3):
                transcript.appendScalar(fr_from_int(c[i] * b % P))
            ch = transcript.challengeScalar()
            chals.append(ch)
            x = fr_to_int(ch)
            claim = (c[0] + x * (c[1] + x * (c[2] + x * c[3]))) % P * b % P
            self.bindChallenge(rnd, ch)
            polys.append(np.stack([fr_from_int(v * b % P) for v in c]))
        out = self.getFinalClaims()
        out.update({"round_polys": np.stack(polys), "challenges": np.stack(chals), "final_claim": fr_from_int(claim)})
        return out


# ---- the UniSkip first round of the outer prover (streaming_outer.zig:523-597, r1cs/univariate_skip.zig)
def uniskip_targets(domain_size=10, degree=9):
    """uniskipTargets (univariate_skip.zig:188-225): the extended points outside the base window, interleaved -5, 6, -6, 7, ..."""
    base_left = -((domain_size - 1) // 2)
    base_right = base_left + domain_size - 1
    out, n, p = [], base_left - 1, base_right + 1
    while n >= -degree and p <= degree and len(out) < degree:
        out.append(n)
        if len(out) >= degree:
            break
        out.append(p)
        n -= 1
        p += 1
    while len(out) < degree and n >= -degree:
        out.append(n)
        n -= 1
    while len(out) < degree and p <= degree:
        out.append(p)
        p += 1
    return out


def _generalized_binomial(t, k):  # LagrangeHelper.generalizedBinomial (:398-428)
    if k == 0:
        return 1
    if t >= 0:
        if k > t:
            return 0
        num = den = 1
        for j in range(k):
            num *= t - j
            den *= j + 1
        return num // den
    sign = -1 if k & 1 else 1
    tt = -t + k - 1
    num = den = 1
    for j in range(k):
        num *= tt - j
        den *= j + 1
    return sign * (num // den)


def shift_coeffs(n, shift):
    """shiftCoeffsI32 (:435-448): alpha with p(shift) = sum_i alpha[i] p(i) for a polynomial of degree < n given on 0..n-1"""
    out = []
    for i in range(n):
        s1 = _generalized_binomial(shift, i)
        s2 = _generalized_binomial(shift - i - 1, (n - 1) - i)
        sign = -1 if ((n - 1 - i) & 1) else 1
        out.append(sign * s1 * s2)
    return out


UNISKIP_TARGETS = uniskip_targets()
COEFFS_PER_J = [shift_coeffs(10, t + 4) for t in UNISKIP_TARGETS]  # :469-476 (TARGET_SHIFTS = target - BASE_LEFT)


def _interpolate_int_domain(vals, left):
    """coefficients (ascending) of the polynomial through (left + i, vals[i]) — lagrangeInterpolate (streaming_outer.zig:728-799);
    exact arithmetic mod r, so any construction gives the same field elements"""
    P, n = _R_P, len(vals)
    coeffs = [0] * n
    for i, y in enumerate(vals):
        if y % P == 0:
            continue
        den, basis = 1, [1] + [0] * (n - 1)
        deg = 0
        for j in range(n):
            if j == i:
                continue
            den = den * (i - j) % P
            xj = left + j
            for k in range(deg + 1, 0, -1):  # multiply by (Y - x_j)
                basis[k] = (basis[k - 1] - xj * basis[k]) % P if k <= deg else basis[k - 1]
            basis[0] = (-xj * basis[0]) % P
            deg += 1
        scale = y * pow(den, P - 2, P) % P
        for k in range(n):
            coeffs[k] = (coeffs[k] + basis[k] * scale) % P
    return coeffs


def _outer_first_round(self):
    """computeFirstRoundPoly (:523-597) -> 28 coefficients (ascending) of s1(Y) = L(tau_high, Y) t1(Y); self.full_tau = the whole tau"""
    P = _R_P
    tau_low = self.full_tau[:-1]
    T2 = 1 << len(tau_low)
    weights = fr_eq_table(tau_low) if len(tau_low) else fr_from_int(1).reshape(1, 4)  # E_out[x_out] E_in[x_in], index = cycle * 2 + group
    w = self.cycle_witnesses
    n = min(w.shape[0], T2 // 2)
    base = []
    for group in (FIRST_GROUP_INDICES, SECOND_GROUP_INDICES):
        az = [_lc_eval(UNIFORM_CONSTRAINTS[ci][0], w[:n]) for ci in group]
        bz = [_fsub(_lc_eval(UNIFORM_CONSTRAINTS[ci][1], w[:n]), _lc_eval(UNIFORM_CONSTRAINTS[ci][2], w[:n])) for ci in group]
        base.append((az, bz))
    ext = []
    for j in range(len(UNISKIP_TARGETS)):
        total = np.zeros(4, dtype=np.uint64)
        for g, (az, bz) in enumerate(base):
            a = np.zeros((n, 4), dtype=np.uint64)
            b = np.zeros((n, 4), dtype=np.uint64)
            for i in range(len(az)):  # the second group uses the first nine of the ten coefficients (:631-657)
                c = COEFFS_PER_J[j][i]
                if c:
                    cf = fr_from_int(abs(c))
                    a = _fadd(a, _fmul(az[i], cf)) if c > 0 else _fsub(a, _fmul(az[i], cf))
                    b = _fadd(b, _fmul(bz[i], cf)) if c > 0 else _fsub(b, _fmul(bz[i], cf))
            total = _fadd(total, _fsum(_fmul(_fmul(a, b), weights[g:2 * n:2]))).reshape(4)
        ext.append(fr_to_int(total))
    self.last_extended_evals = ext
    t1 = [0] * 19
    for z, v in zip(UNISKIP_TARGETS, ext):
        t1[z + 9] = v
    t1_coeffs = _interpolate_int_domain(t1, -9)
    lag = [fr_to_int(x) for x in lagrange_evals_symmetric(self.tau_high, 10)]
    lag_coeffs = _interpolate_int_domain(lag, -4)
    s1 = [0] * 28
    for i in range(10):
        for j in range(19):
            s1[i + j] = (s1[i + j] + lag_coeffs[i] * t1_coeffs[j]) % P
    return np.stack([fr_from_int(v) for v in s1])


StreamingOuterProver.computeFirstRoundPoly = _outer_first_round


# ---------------------------------------------------------------- R1CS cycle inputs from the trace
def _sx(v, bits):
    return v - (1 << bits) if v >> (bits - 1) else v


def _imm_i(w):
    return _sx(w >> 20, 12)


def _imm_s(w):
    return _sx(((w >> 25) << 5) | ((w >> 7) & 0x1F), 12)


def _imm_b(w):
    return _sx(((w >> 31) << 12) | (((w >> 7) & 1) << 11) | (((w >> 25) & 0x3F) << 5) | (((w >> 8) & 0xF) << 1), 13)


def _imm_j(w):
    return _sx(((w >> 31) << 20) | (((w >> 12) & 0xFF) << 12) | (((w >> 20) & 1) << 11) | (((w >> 21) & 0x3FF) << 1), 21)


def _is_noop_instruction(step):  # isNoopInstruction (constraints.zig:569-595)
    if step is None:
        return False
    if step["is_noop"]:
        return True
    w = step["instruction"]
    return (w & 0x7F) == 0x13 and ((w >> 7) & 31) == 0 and ((w >> 15) & 31) == 0 and ((w >> 12) & 7) == 0 and (w >> 20) == 0


def r1cs_cycle_inputs(step, next_step):
    """R1CSCycleInputs.fromTraceStep (src/zkvm/r1cs/constraints.zig:930-1223) with deriveImmediate (:1226-1274), setFlagsFromInstruction
    (:1288-1398) and computeLookupOutput (:600-640) -> the 43 inputs of a cycle as integers mod r. step: tracer.TraceStep as a dict."""
    P, M64 = _R_P, (1 << 64) - 1
    v = [0] * NUM_R1CS_INPUTS
    I = _R1
    w = step["instruction"]
    op, f3, f7, rd = w & 0x7F, (w >> 12) & 7, (w >> 25) & 0x7F, (w >> 7) & 31
    is_load, is_store = op == 0x03, op == 0x23
    v[I["FlagLoad"]], v[I["FlagStore"]] = int(is_load), int(is_store)
    v[I["FlagIsCompressed"]] = int(step["is_compressed"])
    # deriveImmediate: I / S / B / J sign-extended, U as the unsigned upper bits, everything else (incl. the 32-bit ops) zero
    if op in (0x13, 0x03, 0x67):
        imm = _imm_i(w)
    elif op == 0x23:
        imm = _imm_s(w)
    elif op == 0x63:
        imm = _imm_b(w)
    elif op == 0x6F:
        imm = _imm_j(w)
    elif op in (0x37, 0x17):
        imm = w & 0xFFFFF000
    else:
        imm = 0
    v[I["Imm"]] = imm % P
    if op in (0x13, 0x03, 0x67, 0x1B, 0x33, 0x3B, 0x23, 0x63):  # :957-977
        v[I["Rs1Value"]] = step["rs1_value"]
    if op in (0x33, 0x3B, 0x23, 0x63):  # :986-993
        v[I["Rs2Value"]] = step["rs2_value"]
    v[I["RamAddress"]] = (step["rs1_value"] + imm) % P if (is_load or is_store) else 0  # :1001-1009
    mem = step["memory_value"] or 0
    writes_rd = not is_store and op != 0x63 and rd != 0
    if is_load:  # :1023-1047
        v[I["RamReadValue"]] = v[I["RamWriteValue"]] = v[I["RdWriteValue"]] = mem
    elif is_store:
        v[I["RamReadValue"]], v[I["RamWriteValue"]] = mem, step["rs2_value"]
    else:
        v[I["RdWriteValue"]] = step["rd_value"] if writes_rd else 0
    left_is_rs1 = int(op in (0x33, 0x13, 0x03, 0x67, 0x23, 0x63, 0x1B, 0x3B))  # :1059-1098
    left_is_pc = int(op in (0x17, 0x6F))
    right_is_rs2 = int(op in (0x33, 0x63, 0x3B))
    right_is_imm = int(op in (0x13, 0x03, 0x67, 0x23, 0x37, 0x17, 0x6F, 0x1B))
    v[I["FlagLeftOperandIsRs1"]], v[I["FlagLeftOperandIsPC"]] = left_is_rs1, left_is_pc
    v[I["FlagRightOperandIsRs2"]], v[I["FlagRightOperandIsImm"]] = right_is_rs2, right_is_imm
    left = (left_is_rs1 * v[I["Rs1Value"]] + left_is_pc * step["unexpanded_pc"]) % P  # :1106-1118
    right = (right_is_rs2 * v[I["Rs2Value"]] + right_is_imm * v[I["Imm"]]) % P
    v[I["LeftInstructionInput"]], v[I["RightInstructionInput"]], v[I["Product"]] = left, right, left * right % P
    # computeLookupOutput (:600-640)
    if op == 0x6F:
        lookup = (step["pc"] + _imm_j(w)) & M64
    elif op == 0x67:
        lookup = ((step["rs1_value"] + _imm_i(w)) & M64) & ~1
    elif op == 0x63:
        a, b = step["rs1_value"], step["rs2_value"]
        sa, sb = _sx(a, 64), _sx(b, 64)
        lookup = int({0: a == b, 1: a != b, 4: sa < sb, 5: sa >= sb, 6: a < b, 7: a >= b}.get(f3, False))
    else:
        lookup = step["rd_value"]
    v[I["LookupOutput"]] = lookup
    v[I["PC"]], v[I["UnexpandedPC"]] = step["pc"], step["unexpanded_pc"]
    if next_step is not None and not next_step["is_noop"]:  # :1150-1172 (a NoOp successor reads as zero)
        v[I["NextPC"]], v[I["NextUnexpandedPC"]] = next_step["pc"], next_step["unexpanded_pc"]
    # setFlagsFromInstruction (:1288-1398): circuit flags and the two lookup operands
    add = sub = mul = wl = jump = 0
    lo_l, lo_r = left, right  # "NOT Add + Sub + Mul": the operands pass through
    if op == 0x33:
        if f7 == 0x01:
            if f3 == 0:
                mul, lo_l, lo_r = 1, 0, left * right % P
        elif f7 == 0x20 and f3 == 0:
            sub, lo_l, lo_r = 1, 0, (left - right + (1 << 64)) % P
        else:
            add, lo_l, lo_r = 1, 0, (left + right) % P
        wl = 1
    elif op == 0x13:
        add, wl, lo_l, lo_r = 1, 1, 0, (left + right) % P
    elif op in (0x6F, 0x67):
        jump, add, lo_l, lo_r = 1, 1, 0, (left + right) % P
    elif op in (0x37, 0x17):
        add, wl, lo_l, lo_r = 1, 1, 0, (left + right) % P
    v[I["FlagAddOperands"]], v[I["FlagSubtractOperands"]], v[I["FlagMultiplyOperands"]] = add, sub, mul
    v[I["FlagWriteLookupOutputToRD"]], v[I["FlagJump"]] = wl, jump
    v[I["LeftLookupOperand"]], v[I["RightLookupOperand"]] = lo_l, lo_r
    v[I["ShouldJump"]] = jump * (0 if _is_noop_instruction(next_step) else 1)  # :1181-1185
    nz = int(rd != 0)
    v[I["WriteLookupOutputToRD"]], v[I["WritePCtoRD"]] = nz * wl, nz * jump  # :1199-1203
    v[I["ShouldBranch"]] = lookup * int(op == 0x63) % P
    v[I["FlagIsRdNotZero"]], v[I["FlagBranch"]] = nz, int(op == 0x63)
    return [x % P for x in v]


def r1cs_witness_from_trace(steps):
    """R1CSWitnessGenerator.generateWitness (:1469-1494) -> (n, 43, 4): a NoOp padding cycle is createNoopWitness (:1418-1438: only
    DoNotUpdateUnexpandedPC and IsNoop set), a real cycle fromTraceStep(step, the step after it)"""
    rows = []
    for i, st in enumerate(steps):
        if st["is_noop"]:
            row = [0] * NUM_R1CS_INPUTS
            row[_R1["FlagDoNotUpdateUnexpandedPC"]] = row[_R1["FlagIsNoop"]] = 1
        else:
            row = r1cs_cycle_inputs(st, steps[i + 1] if i + 1 < len(steps) else None)
        rows.append(row)
    return np.stack([np.stack([fr_from_int(x) for x in row]) for row in rows])


# ---------------------------------------------------------------- Stage 2: product virtualisation (UniSkip first round and the fused tables)
PRODUCT_VIRTUAL_TARGETS = uniskip_targets(5, 4)  # -3, 3, -4, 4 (univariate_skip.zig:56-59)
PRODUCT_VIRTUAL_COEFFS_PER_J = [shift_coeffs(5, t + 2) for t in PRODUCT_VIRTUAL_TARGETS]  # :78-84


def product_factors(w):
    """extractProductInputs (src/zkvm/spartan/product_remainder.zig:436-476) over all cycles: (n, 43, 4) -> eight (n, 4) columns —
    LeftInstructionInput, RightInstructionInput, IsRdNotZero, WriteLookupOutputToRDFlag, JumpFlag, LookupOutput, BranchFlag, NextIsNoop
    (the NEXT cycle's IsNoop flag; the last cycle reads 1)"""
    w = _c(w)
    n = w.shape[0]
    nxt = np.concatenate([w[1:, _R1["FlagIsNoop"]], fr_from_int(1).reshape(1, 4)]) if n else np.zeros((0, 4), dtype=np.uint64)
    return [w[:, _R1["LeftInstructionInput"]], w[:, _R1["RightInstructionInput"]], w[:, _R1["FlagIsRdNotZero"]], w[:, _R1["FlagWriteLookupOutputToRD"]],
            w[:, _R1["FlagJump"]], w[:, _R1["LookupOutput"]], w[:, _R1["FlagBranch"]], nxt]


def product_fused(f, weights):
    """fusedLeft / fusedRight (product_remainder.zig:113-134; computeProductVirtualExtendedEvals, univariate_skip.zig:648-665) under five
    weights (field elements): left = w0 LeftInput + (w1 + w2) IsRdNotZero + w3 LookupOutput + w4 Jump; right = w0 RightInput + w1 WLFlag +
    w2 Jump + w3 Branch + w4 (1 - NextIsNoop)"""
    one = fr_from_int(1).reshape(1, 4)
    wt = [_c(x).reshape(1, 4) for x in weights]
    left = _fadd(_fadd(_fadd(_fadd(_fmul(f[0], wt[0]), _fmul(f[2], wt[1])), _fmul(f[2], wt[2])), _fmul(f[5], wt[3])), _fmul(f[4], wt[4]))
    right = _fadd(_fadd(_fadd(_fadd(_fmul(f[1], wt[0]), _fmul(f[3], wt[1])), _fmul(f[4], wt[2])), _fmul(f[6], wt[3])), _fmul(_fsub(one, f[7]), wt[4]))
    return left, right


def product_virtual_extended_evals(w, tau):
    """computeProductVirtualExtendedEvals (univariate_skip.zig:607-678): t1 at -3, 3, -4, 4 = sum_x eq(tau[0..log n), x) fused_left fused_right"""
    w = _c(w)
    n = w.shape[0]
    log_n = max(n - 1, 0).bit_length()
    eq = fr_eq_table(_c(tau)[:log_n])[:n]
    f = product_factors(w)
    out = []
    for coeffs in PRODUCT_VIRTUAL_COEFFS_PER_J:
        left, right = product_fused(f, [fr_from_int(c % _R_P) for c in coeffs])
        out.append(_fsum(_fmul(_fmul(left, right), eq)))
    return np.stack(out)


def build_uniskip_first_round_poly(domain_size, degree, base_evals, extended_evals, tau_high):
    """buildUniskipFirstRoundPoly (univariate_skip.zig:486-546): t1 from its values on the base window (given, or zero) and at the
    targets, times the Lagrange kernel L(tau_high, .) over the base window -> 3 * degree + 1 coefficients"""
    P = _R_P
    targets = uniskip_targets(domain_size, degree)
    t1 = [0] * (2 * degree + 1)
    base_left = -((domain_size - 1) // 2)
    if base_evals is not None:
        for i, v in enumerate(base_evals):
            t1[base_left + i + degree] = fr_to_int(v)
    for z, v in zip(targets, extended_evals):
        t1[z + degree] = fr_to_int(v)
    t1c = _interpolate_int_domain(t1, -degree)
    lagc = _interpolate_int_domain([fr_to_int(x) for x in lagrange_evals_symmetric(tau_high, domain_size)], base_left)
    s1 = [0] * (3 * degree + 1)
    for i, a in enumerate(lagc):
        for j, b in enumerate(t1c):
            if i + j < len(s1):
                s1[i + j] = (s1[i + j] + a * b) % P
    return np.stack([fr_from_int(v) for v in s1])


def product_remainder_prover_from_witness(w, r0, tau, uni_skip_claim):
    """ProductVirtualRemainderProver.init (product_remainder.zig:166-243): Lagrange weights at r0 over {-2..2}, the fused left / right tables
    (zero past the trace), the split-eq structure over tau_low scaled by L(tau_high, r0)"""
    w = _c(w)
    n = w.shape[0]
    padded = 1
    while padded < n:
        padded *= 2
    weights = lagrange_evals_symmetric(r0, 5)
    left, right = product_fused(product_factors(w), weights)
    z = np.zeros((padded - n, 4), dtype=np.uint64)
    tau = _c(tau)
    return ProductRemainderProver(np.concatenate([left, z]), np.concatenate([right, z]), tau[:-1], lagrange_kernel(r0, tau[-1], 5), uni_skip_claim)


# ---------------------------------------------------------------- Stage 3 as a whole (src/zkvm/spartan/stage3_prover.zig:113-760)
# The three instances are BUILT here as the reference builds them — the prefix / suffix tables of ShiftPrefixSuffixProver.init
# (:979-1112), its transition to the second phase (:1506-1700), RegistersPrefixSuffixProver (:2189-2467), InstructionInputProver.init
# (:1945-2013) — from the cycle witnesses, Stage 1's r_cycle and Stage 2's product r_cycle; the round evaluations are the C
# restatements above. Tables are lists of canonical integers (mod r); challenges come from the caller (the reference's transcript is
# not restated: the captured run's challenges are fixture data).
def _s3_tab(ints):
    return np.stack([fr_from_int(v) for v in ints]) if len(ints) else np.zeros((0, 4), dtype=np.uint64)


def _s3_ints(tab):
    return [fr_to_int(x) for x in _c(tab).reshape(-1, 4)]


def _s3_eq(r):  # EqPolynomial.evals: r[0] is the most significant index bit
    return _s3_ints(fr_eq_table(_s3_tab(r))) if len(r) else [1]


def _s3_eqp1(r):  # computeEqPlusOneEvals (:1878-1895)
    return _s3_ints(eq_plus_one_table(_s3_tab(r))) if len(r) else [0]


def _s3_bind(t, r):  # new[i] = old[2i] + r (old[2i+1] - old[2i])
    return [(t[2 * i] + r * (t[2 * i + 1] - t[2 * i])) % _R_P for i in range(len(t) // 2)]


def _s3_eval_mle(t, point):  # evaluateMle (:1820-1838): the point's first entry binds the LOW index bit
    t = list(t)
    for r in point:
        if len(t) == 1:
            break
        t = _s3_bind(t, r)
    return t[0]


def _s3_prod(xs):
    r = 1
    for x in xs:
        r = r * x % _R_P
    return r


class Stage3ShiftProver:
    """ShiftPrefixSuffixProver (:928-1919)"""
    COLS = ("UnexpandedPC", "PC", "FlagVirtualInstruction", "FlagIsFirstInSequence", "FlagIsNoop")

    def __init__(self, w, r_outer, r_product, gamma_powers):
        """w: (T, 43) canonical integers; r_outer, r_product big-endian; gamma_powers[0..4]"""
        n, P = len(r_outer), _R_P
        split = n // 2
        self.w, self.T, self.g = w, len(w), list(gamma_powers)
        self.r_outer, self.r_product, self.suffix_n = list(r_outer), list(r_product), split
        self.prefix_n = n - split
        ps, ss = 1 << self.prefix_n, 1 << split
        o_hi, o_lo, p_hi, p_lo = r_outer[:split], r_outer[split:], r_product[:split], r_product[split:]
        # P_0 = eq+1(r_lo, .), P_1 = is_max(r_lo) at index 0 (:1013-1037); order: P_0_outer, P_1_outer, P_0_prod, P_1_prod
        self.P = [_s3_eqp1(o_lo), [_s3_prod(o_lo)] + [0] * (ps - 1), _s3_eqp1(p_lo), [_s3_prod(p_lo)] + [0] * (ps - 1)]
        s0o, s1o, s0p, s1p = _s3_eq(o_hi), _s3_eqp1(o_hi), _s3_eq(p_hi), _s3_eqp1(p_hi)
        iu, ip, iv, if_, inp = (_R1[c] for c in self.COLS)
        Q = [[0] * ps for _ in range(4)]
        for lo in range(ps):
            for hi in range(ss):
                x = lo + (hi << self.prefix_n)
                if x >= self.T:
                    continue
                row = w[x]
                v = (row[iu] + self.g[1] * row[ip] + self.g[2] * row[iv] + self.g[3] * row[if_]) % P
                omn = (1 - row[inp]) % P
                Q[0][lo] = (Q[0][lo] + v * s0o[hi]) % P
                Q[1][lo] = (Q[1][lo] + v * s1o[hi]) % P
                Q[2][lo] = (Q[2][lo] + omn * s0p[hi]) % P
                Q[3][lo] = (Q[3][lo] + omn * s1p[hi]) % P
        Q[2] = [q * self.g[4] % P for q in Q[2]]
        Q[3] = [q * self.g[4] % P for q in Q[3]]
        self.Q, self.size, self.challenges, self.phase2 = Q, ps, [], False
        self.tabs2 = None

    def phase1_tables(self):
        """P_0_outer, Q_0_outer, P_1_outer, Q_1_outer, P_0_prod, Q_0_prod, P_1_prod, Q_1_prod"""
        return [_s3_tab(t) for k in range(4) for t in (self.P[k], self.Q[k])]

    def phase2_tables(self):
        return [_s3_tab(t) for t in self.tabs2]

    def computeRoundEvals(self, previous_claim):
        if not self.phase2:  # :1351-1392
            return shift_phase1_round([_s3_tab(t) for t in self.P], [_s3_tab(t) for t in self.Q], self.size)
        return shift_phase2_round(self.phase2_tables(), _s3_tab(self.g), previous_claim)  # :1399-1455

    def bind(self, r):
        P = _R_P
        if self.phase2:  # :1782-1817
            self.tabs2 = [_s3_bind(t, r) for t in self.tabs2]
            return
        transition = self.size == 2  # :1474-1477
        self.P = [_s3_bind(t, r) for t in self.P]
        self.Q = [_s3_bind(t, r) for t in self.Q]
        self.size //= 2
        self.challenges.append(r)
        if not transition:
            return
        # transitionToPhase2 (:1506-1700)
        self.phase2 = True
        le = self.challenges
        be = le[::-1]
        ss = 1 << self.suffix_n
        tabs = []
        for rr in (self.r_outer, self.r_product):
            hi, lo = rr[:self.suffix_n], rr[self.suffix_n:]
            p0 = _s3_eqp1(lo)
            p1 = [_s3_prod(lo)] + [0] * (len(p0) - 1)
            e0, e1 = _s3_eval_mle(p0, le), _s3_eval_mle(p1, le)
            s0, s1 = _s3_eq(hi), _s3_eqp1(hi)
            tabs.append([(e0 * s0[j] + e1 * s1[j]) % P for j in range(ss)])
        eq = _s3_eq(be)
        pd = len(eq)
        cols = [_R1[c] for c in self.COLS]
        for c in cols:
            t = []
            for j in range(ss):
                acc = 0
                for i in range(pd):
                    x = j * pd + i
                    if x < self.T:
                        acc += eq[i] * self.w[x][c]
                t.append(acc % P)
            tabs.append(t)
        self.tabs2 = tabs  # eq+1_outer, eq+1_prod, unexpanded_pc, pc, is_virtual, is_first_in_sequence, is_noop

    def finalClaims(self):
        return dict(zip(("unexpanded_pc", "pc", "is_virtual", "is_first_in_sequence", "is_noop"), (t[0] for t in self.tabs2[2:])))


class Stage3RegistersProver:
    """RegistersPrefixSuffixProver (:2156-2495)"""

    def __init__(self, w, r_spartan, gamma):
        n, P = len(r_spartan), _R_P
        split = n // 2
        self.r_hi, self.r_lo, self.g = list(r_spartan[:split]), list(r_spartan[split:]), gamma
        pn = n - split
        ps, ss = 1 << pn, 1 << split
        self.P = _s3_eq(self.r_lo)
        suf = _s3_eq(self.r_hi)
        ird, i1, i2 = _R1["RdWriteValue"], _R1["Rs1Value"], _R1["Rs2Value"]
        T = len(w)
        self.wit = [[row[c] for row in w] for c in (ird, i1, i2)]
        g2 = gamma * gamma % P
        self.Q = [0] * ps
        for lo in range(ps):
            for hi in range(ss):
                x = lo + (hi << pn)
                if x < T:
                    self.Q[lo] = (self.Q[lo] + (w[x][ird] + gamma * w[x][i1] + g2 * w[x][i2]) * suf[hi]) % P
        self.size, self.phase2, self.challenges, self.eq2 = ps, False, [], None

    def computeRoundEvals(self, previous_claim):
        g = fr_from_int(self.g)
        if not self.phase2:  # :2334-2354
            return registers_cr_round(False, [_s3_tab(self.P), _s3_tab(self.Q)], g, previous_claim)
        return registers_cr_round(True, [_s3_tab(self.eq2)] + [_s3_tab(t) for t in self.wit], g, previous_claim)  # :2356-2386

    def bind(self, r):
        self.wit = [_s3_bind(t, r) for t in self.wit]
        if self.phase2:  # :2467-2481
            self.eq2 = _s3_bind(self.eq2, r)
            return
        transition = self.size == 2
        self.P, self.Q = _s3_bind(self.P, r), _s3_bind(self.Q, r)  # :2404-2426 (the witness tables are folded alongside)
        self.size //= 2
        self.challenges.append(r)
        if transition:  # :2427-2466: eq(r_lo, reversed prefix challenges) * eq(r_hi, .)
            self.phase2 = True
            e = fr_to_int(fr_eq_mle(_s3_tab(self.r_lo), _s3_tab(self.challenges[::-1])))
            self.eq2 = [x * e % _R_P for x in _s3_eq(self.r_hi)]

    def finalClaims(self):
        return dict(zip(("rd_write_value", "rs1_value", "rs2_value"), (t[0] for t in self.wit)))


class Stage3InstructionInputProver:
    """InstructionInputProver (:1921-2150): ten cycle-length tables folded low to high"""
    COLS = ("FlagLeftOperandIsRs1", "Rs1Value", "FlagLeftOperandIsPC", "UnexpandedPC", "FlagRightOperandIsRs2", "Rs2Value", "FlagRightOperandIsImm", "Imm")

    def __init__(self, w, r_outer, r_product, gamma):
        self.tabs = [[row[_R1[c]] for row in w] for c in self.COLS] + [_s3_eq(r_outer), _s3_eq(r_product)]
        self.g = gamma

    def tables(self):
        return [_s3_tab(t) for t in self.tabs]

    def computeRoundEvals(self, previous_claim):
        return instruction_input_round(self.tables(), fr_from_int(self.g), previous_claim)

    def bind(self, r):
        self.tabs = [_s3_bind(t, r) for t in self.tabs]

    def finalClaims(self):
        return [t[0] for t in self.tabs]


def stage3_evals_to_coeffs(ev):
    """evalsToCoeffs (:846-901), degree 2 or 3, canonical integers"""
    P = _R_P
    if len(ev) == 3:
        p0, p1, p2 = ev
        c2 = (p2 - 2 * p1 + p0) * pow(2, -1, P) % P
        return [p0, (p1 - p0 - c2) % P, c2]
    p0, p1, p2, p3 = ev
    d1, d2, d3 = p1 - p0, p2 - p1, p3 - p2
    dd1, dd2 = d2 - d1, d3 - d2
    c3 = (dd2 - dd1) * pow(6, -1, P) % P
    c2 = (dd1 * pow(2, -1, P) - 3 * c3) % P
    return [p0 % P, (d1 - c2 - c3) % P, c2, c3]


def stage3_input_claims(outer, product, shift_gammas, instr_gamma, reg_gamma):
    """computeShiftInputClaim / computeInstructionInputClaim / computeRegistersInputClaim (:775-844); outer / product: opening claims
    by polynomial name at Stage 1's and at the product sumcheck's r_cycle (canonical integers)"""
    P, g = _R_P, shift_gammas
    shift = (outer["NextUnexpandedPC"] + g[1] * outer["NextPC"] + g[2] * outer["NextIsVirtual"] + g[3] * outer["NextIsFirstInSequence"]
             + g[4] * (1 - product["NextIsNoop"])) % P
    instr = (outer["RightInstructionInput"] + instr_gamma * outer["LeftInstructionInput"]
             + instr_gamma * instr_gamma % P * (product["RightInstructionInput"] + instr_gamma * product["LeftInstructionInput"])) % P
    reg = (outer["RdWriteValue"] + reg_gamma * outer["Rs1Value"] + reg_gamma * reg_gamma % P * outer["Rs2Value"]) % P
    return shift, instr, reg


class Stage3Batch:
    """the round loop of generateStage3Proof (:327-560): three instances under three batching coefficients; a round's message is
    (c0, c2, c3) of the combined cubic; every instance's claim follows its own polynomial"""

    def __init__(self, shift, instr, reg, input_claims, coeffs):
        self.inst, self.claims, self.coeffs = (shift, instr, reg), list(input_claims), list(coeffs)
        self.combined = sum(c * k for c, k in zip(self.claims, self.coeffs)) % _R_P

    def computeRoundPolynomial(self):
        P = _R_P
        ev = [[fr_to_int(x) for x in p.computeRoundEvals(fr_from_int(c))] for p, c in zip(self.inst, self.claims)]
        self.evals = ev
        full = []
        for e in ev:  # a quadratic's value at 3: 3 p(2) - 3 p(1) + p(0)  (:415-417)
            full.append(e if len(e) == 4 else e + [(3 * e[2] - 3 * e[1] + e[0]) % P])
        comb = [sum(full[k][i] * self.coeffs[k] for k in range(3)) % P for i in range(4)]
        self.coeffs_poly = stage3_evals_to_coeffs(comb)
        return [self.coeffs_poly[0], self.coeffs_poly[2], self.coeffs_poly[3]]

    def bindChallenge(self, r):
        P = _R_P
        at = lambda cs: sum(c * pow(r, i, P) for i, c in enumerate(cs)) % P
        self.combined = at(self.coeffs_poly)
        self.claims = [at(stage3_evals_to_coeffs(e)) for e in self.evals]
        for p in self.inst:
            p.bind(r)


# ---------------------------------------------------------------- Dory's data-parallel G1 / Fr pieces (src/poly/commitment/dory.zig)
def dory_row_commitments(g1_xy, g1_inf, evals, num_columns):
    """computeRowCommitments (dory.zig:646-670): row r = MSM(g1_vec[0..len(row)], evals[r * cols .. min((r + 1) * cols, len)]);
    the last row may be shorter -> (rows, 8), (rows,)"""
    ev = _c(evals).reshape(-1, 4)
    rows = (ev.shape[0] + num_columns - 1) // num_columns
    out, inf = np.zeros((rows, 8), dtype=np.uint64), np.zeros(rows, dtype=np.uint8)
    for r in range(rows):
        row = ev[r * num_columns:(r + 1) * num_columns]
        k = row.shape[0]
        xy, i = msm_g1(_c(g1_xy)[:k], None if g1_inf is None else _c(g1_inf, np.uint8)[:k], row)
        out[r], inf[r] = xy, i
    return out, inf


def dory_vector_matrix_product(evals, left_vec, nu, sigma):
    """computeVectorMatrixProduct (dory.zig:622-642): v[col] = sum_row left_vec[row] * evals[row * 2^sigma + col] over 2^nu rows; rows
    past left_vec and entries past evals contribute nothing -> (2^sigma, 4)"""
    ev, lv = _c(evals).reshape(-1, 4), _c(left_vec).reshape(-1, 4)
    cols, rows = 1 << sigma, 1 << nu
    acc = [0] * cols
    e = [fr_to_int(x) for x in ev]
    l = [fr_to_int(x) for x in lv]
    for r in range(min(rows, len(l))):
        for c in range(cols):
            idx = r * cols + c
            if idx < len(e):
                acc[c] = (acc[c] + l[r] * e[idx]) % _R_P
    return np.stack([fr_from_int(v) for v in acc])


def dory_multilinear_lagrange_basis(point, out_len=None):
    """multilinearLagrangeBasis (dory.zig:544-588): output[i] = prod_level (bit_level(i) ? point[level] : 1 - point[level]), the index's
    LOW bit belongs to point[0]; an output shorter than 2^len(point) holds the first entries of the full table -> (out_len, 4)"""
    pt = [fr_to_int(x) for x in _c(np.asarray(point, dtype=np.uint64).reshape(-1, 4))]
    n = (1 << len(pt)) if out_len is None else out_len
    out = []
    for i in range(n):
        v = 1
        for lvl, p in enumerate(pt):
            v = v * (p if (i >> lvl) & 1 else (1 - p)) % _R_P
        out.append(v)
    return np.stack([fr_from_int(v) for v in out]) if out else np.zeros((0, 4), dtype=np.uint64)


def dory_evaluation_vectors(point, nu, sigma):
    """computeEvaluationVectors (dory.zig:590-620) -> (left_vec (2^nu, 4), right_vec (2^sigma, 4)); entries the reference does not write stay zero"""
    pt = _c(np.asarray(point, dtype=np.uint64).reshape(-1, 4))
    d = pt.shape[0]
    left, right = np.zeros((1 << nu, 4), dtype=np.uint64), np.zeros((1 << sigma, 4), dtype=np.uint64)
    one = fr_from_int(1)
    if d == 0:
        left[0], right[0] = one, one
    elif d <= sigma:
        right[:1 << d] = dory_multilinear_lagrange_basis(pt)
        left[0] = one
    elif d <= nu + sigma:
        right[:] = dory_multilinear_lagrange_basis(pt[:sigma])
        left[:1 << (d - sigma)] = dory_multilinear_lagrange_basis(pt[sigma:])
    else:
        right[:] = dory_multilinear_lagrange_basis(pt[:sigma])
        left[:] = dory_multilinear_lagrange_basis(pt[sigma:], 1 << nu)
    return left, right


# ---------------------------------------------------------------- the remaining fold sites (canonical integers; small restatements)
class SpartanOuterRounds:
    """SpartanOuterProver.computeStandardRoundPoly / bindChallenge (src/zkvm/spartan/outer.zig:364-407) over working_vals"""

    def __init__(self, working_vals):
        self.vals = [fr_to_int(x) for x in _c(working_vals).reshape(-1, 4)]
        self.challenges = []

    def computeStandardRoundPoly(self):
        P = _R_P
        if len(self.vals) <= 1:
            return [self.vals[0] if self.vals else 0, 0, 0]
        p0, p1 = sum(self.vals[0::2]) % P, sum(self.vals[1::2]) % P
        return [p0, p1, (2 * p1 - p0) % P]

    def bindChallenge(self, r):
        self.challenges.append(r)
        if len(self.vals) > 1:
            self.vals = [((1 - r) * self.vals[2 * i] + r * self.vals[2 * i + 1]) % _R_P for i in range(len(self.vals) // 2)]


class Phase1Prover:
    """Phase1Prover (src/zkvm/spartan/prefix_suffix.zig:35-147): P / Q pairs, g(0) = sum P[2i] Q[2i], g(1) = sum P[2i+1] Q[2i+1] (:95-112),
    bind every buffer low to high (:114-132)"""

    def __init__(self):
        self.pairs, self.challenges, self.current_size = [], [], 0

    def addPair(self, P, Q):
        P, Q = [fr_to_int(x) for x in _c(P).reshape(-1, 4)], [fr_to_int(x) for x in _c(Q).reshape(-1, 4)]
        assert len(P) == len(Q) and (self.current_size in (0, len(P)))
        self.current_size = len(P)
        self.pairs.append([P, Q])

    def shouldTransition(self):
        return self.current_size <= 2

    def computeRoundEvals(self):
        g0 = g1 = 0
        for P, Q in self.pairs:
            for i in range(self.current_size // 2):
                g0 += P[2 * i] * Q[2 * i]
                g1 += P[2 * i + 1] * Q[2 * i + 1]
        return [g0 % _R_P, g1 % _R_P]

    def bind(self, r):
        self.challenges.append(r)
        half = self.current_size // 2
        for pq in self.pairs:
            for t in range(2):
                v = pq[t]
                pq[t] = [(v[2 * i] + r * (v[2 * i + 1] - v[2 * i])) % _R_P for i in range(half)]
        self.current_size = half


def init_shift_q_buffers(unexpanded_pc, pc, is_virtual, is_first_in_sequence, is_noop, suffix_0_outer, suffix_1_outer, suffix_0_product,
                         suffix_1_product, gamma_powers, prefix_size):
    """initShiftQBuffers (src/zkvm/spartan/prefix_suffix.zig:149-232) -> (Q_0_outer, Q_1_outer, Q_0_product, Q_1_product), (prefix_size, 4) each"""
    P = _R_P
    cols = [[fr_to_int(x) for x in _c(t).reshape(-1, 4)] for t in (unexpanded_pc, pc, is_virtual, is_first_in_sequence, is_noop)]
    suf = [[fr_to_int(x) for x in _c(t).reshape(-1, 4)] for t in (suffix_0_outer, suffix_1_outer, suffix_0_product, suffix_1_product)]
    g = [fr_to_int(x) for x in _c(gamma_powers).reshape(-1, 4)]
    ss = len(suf[0])
    assert len(cols[0]) == prefix_size * ss and len(g) >= 5
    Q = [[0] * prefix_size for _ in range(4)]
    for hi in range(ss):
        for lo in range(prefix_size):
            x = lo + hi * prefix_size
            v = (cols[0][x] + g[1] * cols[1][x] + g[2] * cols[2][x] + g[3] * cols[3][x]) % P
            Q[0][lo] = (Q[0][lo] + v * suf[0][hi]) % P
            Q[1][lo] = (Q[1][lo] + v * suf[1][hi]) % P
            nf = (1 - cols[4][x]) % P
            Q[2][lo] = (Q[2][lo] + nf * suf[2][hi]) % P
            Q[3][lo] = (Q[3][lo] + nf * suf[3][hi]) % P
    Q[2] = [q * g[4] % P for q in Q[2]]
    Q[3] = [q * g[4] % P for q in Q[3]]
    return tuple(np.stack([fr_from_int(v) for v in q]) for q in Q)


class LassoPrefixPolynomial:
    """PrefixPolynomial (src/zkvm/lasso/prefix_suffix.zig:133-231): bind = the high-half fold new[i] = old[i] (1 - c) + old[i + half] c
    (:175-196); evaluate = sum_i evals[i] prod_j (bit_j(i) ? point[j] : 1 - point[j]) (:198-216)"""

    def __init__(self, evaluations):
        self.evaluations = [fr_to_int(x) for x in _c(evaluations).reshape(-1, 4)]
        self.num_vars = len(self.evaluations).bit_length() - 1

    def bind(self, c):
        h = len(self.evaluations) // 2
        out = LassoPrefixPolynomial.__new__(LassoPrefixPolynomial)
        out.evaluations = [(self.evaluations[i] * (1 - c) + self.evaluations[i + h] * c) % _R_P for i in range(h)]
        out.num_vars = self.num_vars - 1
        return out

    def evaluate(self, point):
        assert len(point) == self.num_vars
        r = 0
        for i, e in enumerate(self.evaluations):
            t = e
            for j, pj in enumerate(point):
                t = t * (pj if (i >> j) & 1 else (1 - pj)) % _R_P
            r += t
        return r % _R_P


# ---------------------------------------------------------------- Stage 4 of the standard path (src/zkvm/prover.zig:713-828)
def lt_table_int(r_cycle):
    """LtPolynomial.evaluateAtIndex over the cube (src/zkvm/ram/val_evaluation.zig:309-330), canonical integers; index bit i <-> r_cycle[i]"""
    P, v = _R_P, len(r_cycle)
    out = []
    for j in range(1 << v):
        res = 0
        for i in range(v):
            if not (j >> i) & 1:
                c = r_cycle[i]
                for k in range(i + 1, v):
                    c = c * (r_cycle[k] if (j >> k) & 1 else (1 - r_cycle[k])) % P
                res += c
        out.append(res % P)
    return out


def val_evaluation_tables(accesses, initial_ram, trace_len, k, r_address, r_cycle, start_address):
    """IncPolynomial.fromTrace (:92-165), WaPolynomial.fromTrace / evaluateAtCycle (:208-262), LtPolynomial (:289-330) as ValEvaluationProver.init
    tabulates them (:423-470): n = ceilPow2(max(trace_len, 1)) entries each. accesses: [(timestamp, address, is_write, value)];
    r_address / r_cycle: canonical integers -> (inc, wa, lt) lists of integers"""
    P = _R_P
    n = 1
    while n < max(trace_len, 1):
        n <<= 1
    inc, wa_addr = [0] * n, [None] * n
    last = {}
    for addr, val in (initial_ram or {}).items():
        if addr >= start_address and (addr - start_address) // 8 < k:
            last[addr] = val
    for ts, addr, is_write, value in accesses:
        if not is_write or addr < start_address or (addr - start_address) // 8 >= k or ts >= trace_len:
            continue
        old = last.get(addr, 0)
        inc[ts] = (value - old) % P
        last[addr] = value
        wa_addr[ts] = (addr - start_address) // 8
    def eq_at(r, idx):  # computeEqAtPoint (:790-802): index bit i <-> r[i]
        v = 1
        for i, ri in enumerate(r):
            v = v * (ri if (idx >> i) & 1 else (1 - ri)) % P
        return v
    wa = [0 if a is None else eq_at(r_address, a) for a in wa_addr]
    full = lt_table_int(r_cycle)
    lt = [full[j % len(full)] for j in range(n)]  # evaluateAtIndex reads len(r_cycle) index bits
    return inc, wa, lt


def stage4_prove(accesses, initial_ram, trace_len, log_k, log_t, start_address, transcript):
    """proveStage4 (prover.zig:713-828): log_k "r_address" and log_t "r_cycle_val" challenges, ValEvaluationProver over the memory trace
    (init_eval = 0), the initial claim, log2_ceil(trace_len) rounds of [p(0..3)] under "val_eval_round", the final claim"""
    r_address = [transcript.challenge_scalar(b"r_address") for _ in range(log_k)]
    r_cycle = [transcript.challenge_scalar(b"r_cycle_val") for _ in range(log_t)]
    out = {"r_address": r_address, "r_cycle": r_cycle}
    if trace_len == 0:
        return out
    inc, wa, lt = val_evaluation_tables(accesses, initial_ram, trace_len, 1 << log_k, [fr_to_int(x) for x in r_address], [fr_to_int(x) for x in r_cycle], start_address)
    claim = sum(a * b % _R_P * c for a, b, c in zip(inc, wa, lt)) % _R_P
    pr = ValEvaluationProver(_s3_tab(inc), _s3_tab(wa), _s3_tab(lt), fr_from_int(claim))
    out["initial_claim"] = fr_from_int(claim)
    num_rounds = 0 if trace_len <= 1 else (trace_len - 1).bit_length()
    polys, chals = [], []
    for _ in range(num_rounds):
        rp = pr.computeRoundPolynomial()
        polys.append(rp)
        ch = transcript.challenge_scalar(b"val_eval_round")
        chals.append(ch)
        pr.bindChallengeWithPoly(ch, rp)
    f = pr.getFinalClaims()
    out.update(round_polys=polys, challenges=chals, final_claim=_mul(_mul(f[0], f[1]), f[2]), final_openings=f)
    return out


class ExpandingTable:
    """ExpandingTable (src/zkvm/lasso/expanding_table.zig:27-190), canonical integers: bind doubles the table, new[2i] = v (1 - r),
    new[2i+1] = v r (:83-99) — after k binds the eq table of the challenges with the FIRST challenge on the index's top bit"""

    def __init__(self, max_rounds, initial=1):
        self.values, self.round, self.max_rounds = [initial % _R_P], 0, max_rounds

    def bind(self, r):
        assert self.round < self.max_rounds
        self.values = [x for v in self.values for x in (v * (1 - r) % _R_P, v * r % _R_P)]
        self.round += 1

    def sum(self):
        return sum(self.values) % _R_P

    def condense(self, weights, out_bits):
        """:144-161: out[i / chunk] += values[i] * weights[i], chunk = 2^(round - out_bits)"""
        assert len(weights) == len(self.values) and out_bits <= self.round
        chunk = 1 << (self.round - out_bits)
        out = [0] * (1 << out_bits)
        for i, (v, w) in enumerate(zip(self.values, weights)):
            out[i // chunk] = (out[i // chunk] + v * w) % _R_P
        return out


def lt_table(r_cycle):
    """the C restatement of LtPolynomial over the cube (zo_lt_table) -> (2^v, 4)"""
    r = _c(np.asarray(r_cycle, dtype=np.uint64).reshape(-1, 4))
    out = np.empty((1 << r.shape[0], 4), dtype=np.uint64)
    lib.zo_lt_table(_p(r), C.c_size_t(r.shape[0]), _p(out))
    return out


def weighted_colsum(table, rows, cols, weights):
    """the C restatement of the x_hi / x_lo double loop (zo_weighted_colsum) -> (m, cols, 4)"""
    t, w = _c(table), _c(weights)
    m = w.size // (4 * rows)
    out = np.empty((m, cols, 4), dtype=np.uint64)
    lib.zo_weighted_colsum(_p(t), C.c_size_t(rows), C.c_size_t(cols), _p(w), C.c_size_t(m), _p(out))
    return out
