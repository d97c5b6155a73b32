"""poly.py — zolt.poly: Eq / EqPlusOne / GruenSplitEq / Dense polynomials.

Part of the zolt_amd.api package (the host mirror of the reference's module API over libzolt_gpu.so); import zolt_amd.api,
which re-exports every name of every part."""
import numpy as np

from .. import lib
from ._base import *  # noqa: F401,F403
from .msm import *  # noqa: F401,F403
from .commitment import *  # noqa: F401,F403
from .wire import *  # noqa: F401,F403

# ---- polynomials
class EqPolynomial:
    def __init__(self, r):
        self.r = np.ascontiguousarray(r, dtype=np.uint64).reshape(-1, 4).copy()

    def evals(self):
        """EqPolynomial.evals (src/poly/mod.zig:240-242): 2^n table, index MSB <-> r[0]."""
        return lib.fr_eq_table(self.r)

    def evaluate(self, x):
        """EqPolynomial.evaluate (src/poly/mod.zig:214-227): eq(x, r) — host scalar code, like the reference's."""
        return EqPolynomial.mle(self.r, x)

    @staticmethod
    def mle(r, x):
        """EqPolynomial.mle (src/poly/mod.zig:311-321): prod_i (r_i x_i + (1 - r_i)(1 - x_i))."""
        r = np.ascontiguousarray(r, dtype=np.uint64).reshape(-1, 4)
        x = np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4)
        assert r.shape[0] == x.shape[0]  # std.debug.assert(r.len == x.len)
        acc = 1
        for ri, xi in zip(r, x):
            a, b = fr_to_int(ri), fr_to_int(xi)
            acc = acc * ((a * b + (1 - a) * (1 - b)) % R_MOD) % R_MOD
        return fr_from_int(acc)

    @staticmethod
    def evalsSliceWithScaling(r, scaling_factor=None):
        return lib.fr_eq_table(np.ascontiguousarray(r, dtype=np.uint64).reshape(-1, 4), scaling_factor)


class EqPlusOnePolynomial:
    """EqPlusOnePolynomial(F) (src/poly/mod.zig:332-446): eq+1(x, y) = 1 iff y = x + 1 on the cube (x[0] is the MSB). evaluate / mle are
    the reference's host scalar formula; the table over the cube comes from the device."""

    def __init__(self, x):
        self.x = np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4).copy()

    def evaluate(self, y):
        return EqPlusOnePolynomial.mle(self.x, y)

    @staticmethod
    def mle(x, y):
        """:407-435: sum over the flip position k of prod_{i<k} x_i (1 - y_i) * (1 - x_k) y_k * prod_{i>k} eq(x_i, y_i), bits counted from the LSB"""
        xs = [fr_to_int(v) for v in np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4)]
        ys = [fr_to_int(v) for v in np.ascontiguousarray(y, dtype=np.uint64).reshape(-1, 4)]
        l = len(xs)
        assert len(ys) == l
        result = 0
        for k in range(l):
            lower = 1
            for i in range(k):
                idx = l - 1 - i
                lower = lower * (xs[idx] * (1 - ys[idx]) % R_MOD) % R_MOD
            kth = (1 - xs[l - 1 - k]) * ys[l - 1 - k] % R_MOD
            higher = 1
            for i in range(k + 1, l):
                idx = l - 1 - i
                higher = higher * ((xs[idx] * ys[idx] + (1 - xs[idx]) * (1 - ys[idx])) % R_MOD) % R_MOD
            result = (result + lower * kth % R_MOD * higher) % R_MOD
        return fr_from_int(result)

    def evals(self):
        """the table over the cube (computeEqPlusOneEvals, :530-548)"""
        return lib.fr_eq_plus_one_table(self.x)


class EqPlusOnePrefixSuffixPoly:
    """EqPlusOnePrefixSuffixPoly(F).init (src/poly/mod.zig:462-528): r = (r_hi || r_lo) split at len / 2; prefix_0 = eq+1(r_lo, .),
    suffix_0 = eq(r_hi, .), prefix_1 = is_max(r_lo) at index 0, suffix_1 = eq+1(r_hi, .) — three table builds on the device."""

    def __init__(self, r):
        r = np.ascontiguousarray(r, dtype=np.uint64).reshape(-1, 4)
        assert r.shape[0] >= 2
        mid = r.shape[0] // 2
        r_hi, r_lo = r[:mid], r[mid:]
        self.prefix_0 = lib.fr_eq_plus_one_table(r_lo)
        self.suffix_0 = lib.fr_eq_table(r_hi)
        self.suffix_1 = lib.fr_eq_plus_one_table(r_hi)
        is_max = 1
        for v in r_lo:
            is_max = is_max * fr_to_int(v) % R_MOD
        self.prefix_1 = np.zeros_like(self.prefix_0)
        self.prefix_1[0] = fr_from_int(is_max)

    def prefixSize(self):
        return self.prefix_0.shape[0]

    def suffixSize(self):
        return self.suffix_0.shape[0]


class GruenSplitEqPolynomial:
    """GruenSplitEqPolynomial (src/poly/split_eq.zig:22-514). The prefix-table set is built on the device in one launch per
    half (zg_fr_eq_prefix_tables); bind / computeCubicRoundPoly are the reference's host scalar algebra; getFullEqTable and
    getEActiveForWindow are eq-table builds on the device."""

    def __init__(self, tau, scaling_factor=None):
        """init / initWithScaling (:51-183): m = len/2, w_out = tau[0..m), w_in = tau[m..len-1), tau[len-1] stays out of the tables"""
        self.tau = np.ascontiguousarray(tau, dtype=np.uint64).reshape(-1, 4).copy()
        n = self.tau.shape[0]
        self.current_index = n
        self.current_scalar = fr_from_int(1) if scaling_factor is None else np.ascontiguousarray(scaling_factor, dtype=np.uint64).copy()
        self._d_out = self._d_in = None  # the same table sets in HBM, built on first use by getWindowEqTablesDev
        if n == 0:  # :75-86: a valid object without tables (deinit is a no-op)
            self.E_out_vec, self.E_in_vec, self.num_x_out, self.num_x_in = [], [], 0, 0
            return
        m = n // 2
        self.num_x_out = m
        self.num_x_in = min(n - 1 - m, n - 1) if n > 1 else 0
        self.E_out_vec = list(lib.fr_eq_prefix_tables(self.tau[:m]))
        self.E_in_vec = list(lib.fr_eq_prefix_tables(self.tau[m:m + self.num_x_in]))

    init = classmethod(lambda cls, tau: cls(tau))
    initWithScaling = classmethod(lambda cls, tau, scaling_factor: cls(tau, scaling_factor))

    def bind(self, r):
        """bind (:213-248): current_scalar *= eq(tau[current_index-1], r); pops the largest E_in (then E_out) table, never table 0"""
        if self.current_index == 0:
            return
        t, rv = fr_to_int(self.tau[self.current_index - 1]), fr_to_int(r)
        eq_val = (t * rv + (1 - t) * (1 - rv)) % R_MOD
        self.current_scalar = fr_from_int(fr_to_int(self.current_scalar) * eq_val % R_MOD)
        self.current_index -= 1
        m = self.tau.shape[0] // 2
        if m < self.current_index:
            if len(self.E_in_vec) > 1:
                self.E_in_vec.pop()
        elif self.current_index > 0:
            if len(self.E_out_vec) > 1:
                self.E_out_vec.pop()

    def getFullEqTable(self):
        """getFullEqTable (:254-285): eq(tau[0..current_index), .) scaled by current_scalar, tau[0] <-> MSB"""
        return lib.fr_eq_table(self.tau[:self.current_index], self.current_scalar)

    def getTauHigh(self):
        """getTauHigh (:291-294)"""
        return self.tau[-1].copy() if self.tau.shape[0] else fr_from_int(0)

    def getWindowEqTables(self, num_unbound_vars, window_size):
        """getWindowEqTables (:312-343); the first argument is ignored, as in the reference. -> (E_out, E_in, head_in_bits)"""
        num_unbound = self.current_index
        head_len = max(num_unbound - min(window_size, num_unbound), 0)
        m = self.tau.shape[0] // 2
        head_out_bits = min(head_len, m)
        head_in_bits = max(head_len - head_out_bits, 0)
        one = fr_from_int(1).reshape(1, 4)  # (an object over no variables has no tables: the empty product)
        e_out = one if not self.E_out_vec else (self.E_out_vec[head_out_bits] if head_out_bits < len(self.E_out_vec) else self.E_out_vec[-1])
        e_in = one if not self.E_in_vec else (self.E_in_vec[head_in_bits] if head_in_bits < len(self.E_in_vec) else self.E_in_vec[-1])
        return e_out, e_in, head_in_bits

    def getWindowEqTablesDev(self, window_size):
        """getWindowEqTables for device consumers (zg_psc_round_gruen): (d_E_out, |E_out|, d_E_in, |E_in|) — pointers into the two
        prefix-table buffers zg_fr_eq_prefix_tables_dev filled (table k starts at element 2^k - 1); bind()'s pops only shorten the lists."""
        if self._d_out is None:
            m = self.tau.shape[0] // 2
            self._d_out = lib.DeviceBuffer(((2 << m) - 1) * 32)
            self._d_in = lib.DeviceBuffer(((2 << self.num_x_in) - 1) * 32)
            lib.fr_eq_prefix_tables_dev(self.tau[:m], self._d_out.ptr)
            lib.fr_eq_prefix_tables_dev(self.tau[m:m + self.num_x_in], self._d_in.ptr)
            lib.sync()  # the consumers read the tables on their sessions' own streams
        num_unbound = self.current_index
        head_len = max(num_unbound - min(window_size, num_unbound), 0)
        head_out_bits = min(head_len, self.tau.shape[0] // 2)
        head_in_bits = max(head_len - head_out_bits, 0)
        ko = min(head_out_bits, len(self.E_out_vec) - 1)
        ki = min(head_in_bits, len(self.E_in_vec) - 1)
        return self._d_out.ptr + ((1 << ko) - 1) * 32, 1 << ko, self._d_in.ptr + ((1 << ki) - 1) * 32, 1 << ki

    def deinit(self):
        for b in (self._d_out, self._d_in):
            if b is not None:
                b.free()
        self._d_out = self._d_in = None

    def getCurrentEqFactors(self):
        """getCurrentEqFactors (:441-452) -> (eq_0, eq_1)"""
        if self.current_index == 0:
            return self.current_scalar.copy(), self.current_scalar.copy()
        cs, t = fr_to_int(self.current_scalar), fr_to_int(self.tau[self.current_index - 1])
        return fr_from_int(cs * (1 - t) % R_MOD), fr_from_int(cs * t % R_MOD)

    def computeCubicRoundPoly(self, q_constant, q_quadratic_coeff, previous_claim):
        """computeCubicRoundPoly (:353-434): [s(0), s(1), s(2), s(3)] with s = l*q, q(1) recovered from the claim"""
        claim = fr_to_int(previous_claim)
        if self.current_index == 0:
            return np.stack([fr_from_int(claim), fr_from_int(0), fr_from_int(0), fr_from_int(0)])
        cs, t = fr_to_int(self.current_scalar), fr_to_int(self.tau[self.current_index - 1])
        c, e = fr_to_int(q_constant), fr_to_int(q_quadratic_coeff)
        l0, l1 = cs * (1 - t) % R_MOD, cs * t % R_MOD
        slope = (l1 - l0) % R_MOD
        l2, l3 = (l0 + 2 * slope) % R_MOD, (l0 + 3 * slope) % R_MOD
        q1 = 0 if l1 == 0 else (claim - l0 * c) * pow(l1, R_MOD - 2, R_MOD) % R_MOD
        q2 = (2 * q1 - c + 2 * e) % R_MOD
        q3 = (q2 + q1 - c + 4 * e) % R_MOD
        return np.stack([fr_from_int(l0 * c % R_MOD), fr_from_int(l1 * q1 % R_MOD), fr_from_int(l2 * q2 % R_MOD), fr_from_int(l3 * q3 % R_MOD)])

    def getEActiveForWindow(self, window_size):
        """getEActiveForWindow (:466-514): eq over the window's bits except the current one; [1] for windows of 0/1 or too wide"""
        if window_size <= 1 or window_size > self.current_index:
            return fr_from_int(1).reshape(1, 4)
        ws = self.current_index - window_size
        return lib.fr_eq_table(self.tau[ws:ws + window_size - 1])


class DensePolynomial:
    def __init__(self, evaluations):
        ev = np.ascontiguousarray(evaluations, dtype=np.uint64).reshape(-1, 4)
        n = ev.shape[0]
        assert n and n & (n - 1) == 0  # src/poly/mod.zig:36-37
        self.evaluations = ev.copy()
        self.num_vars = n.bit_length() - 1

    def len(self):
        return self.evaluations.shape[0]

    def evaluate(self, point):
        """DensePolynomial.evaluate (src/poly/mod.zig:73-92), index bit j <-> point[j]."""
        point = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1, 4)
        assert point.shape[0] == self.num_vars
        return lib.fr_dense_evaluate(self.evaluations, point)

    def add(self, other):
        """DensePolynomial.add (src/poly/mod.zig:94-110) -> new polynomial"""
        assert self.num_vars == other.num_vars
        return DensePolynomial(lib.field_op(lib.FR, lib.OP_ADD, self.evaluations, other.evaluations))

    def scale(self, scalar):
        """DensePolynomial.scale (src/poly/mod.zig:112-126) -> new polynomial"""
        return DensePolynomial(lib.fr_scale(self.evaluations, scalar))

    def bindFirst(self, value):
        """high-half fold into a NEW polynomial (src/poly/mod.zig:128-149)."""
        assert self.num_vars > 0
        return DensePolynomial(lib.fr_bind_high(self.evaluations, value))

    def bindLow(self, value):
        """adjacent-pair fold IN PLACE (src/poly/mod.zig:160-175)."""
        assert self.num_vars > 0
        self.evaluations = lib.fr_bind_low(self.evaluations, value)
        self.num_vars -= 1


__all__ = [_k for _k in dir() if not _k.startswith("__")]  # underscore helpers are shared between the parts too
