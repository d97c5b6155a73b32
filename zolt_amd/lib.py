"""ctypes binding of include/zolt_gpu.h. Arrays are numpy uint64 (host) or raw device
pointers (ints, e.g. torch.Tensor.data_ptr()). No torch import here."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ZOLT_GPU_LIB: load another build of the same library (A/B experiments on kernel variants); never a fallback
LIB_PATH = os.environ.get("ZOLT_GPU_LIB") or os.path.join(_HERE, "libzolt_gpu.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make -C zolt_amd/csrc` (or __graft_entry__.build()). "
        "zolt_amd has no CPU fallback."
    )
_lib = C.CDLL(LIB_PATH)

OK, ERR_INVALID, ERR_HIP, ERR_NOMEM, ERR_NO_DEVICE, ERR_VERIFY = 0, 1, 2, 3, 4, 5
FR, FP = 0, 1
OP_MUL, OP_ADD, OP_SUB, OP_NEG, OP_SQR, OP_INV, OP_FROM_MONT, OP_TO_MONT = range(8)
# self-test hooks (include/zolt_gpu_internal.h)
OP_MUL29, OP_SQR29, OP_X3_29, OP_INV_XGCD, OP_INV_SAFEGCD = 9, 10, 11, 12, 13
SC_HIGH_HALF, SC_LOW_PAIR = 0, 1

# every symbol include/zolt_gpu.h declares, with its ctypes signature: GENERATED from the header (tools/gen_bindings.py -> _abi.py), so the
# binding cannot drift from the C ABI; apply() fails with AttributeError if the loaded library lacks one of them
from . import _abi  # noqa: E402

SYMBOLS = _abi.SYMBOLS
# test / bench scaffolding of include/zolt_gpu_internal.h (not part of the drop-in boundary)
INTERNAL_SYMBOLS = _abi.INTERNAL_SYMBOLS
_abi.apply(_lib)
if _lib.zg_abi_version() >> 16 != _abi.ZG_ABI_MAJOR:
    raise ImportError(f"{LIB_PATH}: ABI major {_lib.zg_abi_version() >> 16}, this binding was generated for {_abi.ZG_ABI_MAJOR}")


class MsmConfig(C.Structure):
    _fields_ = [("window_bits", C.c_int), ("precompute_levels", C.c_int), ("expected_uses", C.c_int)]



_u64p = C.POINTER(C.c_uint64)
_u8p = C.POINTER(C.c_uint8)


class ZgError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = _lib.zg_last_error()
        super().__init__(f"{where}: error {code}: {msg.decode() if msg else ''}")


def _chk(rc, where):
    if rc != OK:
        raise ZgError(rc, where)


def _h(a):
    """host numpy array -> uint64* (None passes NULL)"""
    return None if a is None else a.ctypes.data_as(_u64p)


def _hb(a):
    return None if a is None else a.ctypes.data_as(_u8p)


def _d(ptr):
    """device address (int) -> void*"""
    return C.c_void_p(int(ptr) if ptr else None)


def _c(a, dtype=np.uint64):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


def version():
    return _lib.zg_version().decode()


def device_count():
    return int(_lib.zg_device_count())


def init(device=-1):
    _chk(_lib.zg_init(C.c_int(device)), "zg_init")


def init_devices(n=0):
    """one process, several GPUs (devices 0..n-1; n <= 0: all visible)"""
    _chk(_lib.zg_init_devices(C.c_int(n)), "zg_init_devices")


def n_devices():
    return int(_lib.zg_n_devices())


def shutdown():
    _lib.zg_shutdown()


import atexit as _atexit  # noqa: E402

_atexit.register(lambda: _lib.zg_shutdown())  # communicators / pooled sessions are released while HIP and RCCL are still alive


def abi_version():
    """(major, minor) of the loaded library's C ABI"""
    v = int(_lib.zg_abi_version())
    return v >> 16, v & 0xFFFF


def abi_features():
    return int(_lib.zg_abi_features())


def dev_trim():
    """return the device pool's idle blocks to the driver"""
    _chk(_lib.zg_dev_trim(), "zg_dev_trim")


def sync():
    _chk(_lib.zg_sync(), "zg_sync")


def last_error():
    m = _lib.zg_last_error()
    return m.decode() if m else ""


class DeviceBuffer:
    """Raw device memory through the C ABI itself (zg_dev_alloc / zg_memcpy_*): what a host without its own HIP binding uses
    (the Zig shim), and what keeps this module free of torch."""

    def __init__(self, nbytes):
        p = C.c_void_p()
        _chk(_lib.zg_dev_alloc(C.c_size_t(nbytes), C.byref(p)), "zg_dev_alloc")
        self.ptr, self.nbytes = p.value, nbytes

    @classmethod
    def from_host(cls, arr):
        a = np.ascontiguousarray(arr)
        b = cls(max(a.nbytes, 1))
        if a.nbytes:
            _chk(_lib.zg_memcpy_h2d(C.c_void_p(b.ptr), a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes)), "zg_memcpy_h2d")
        return b

    def to_host(self, dtype=np.uint64):
        out = np.empty(self.nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        _chk(_lib.zg_memcpy_d2h(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), C.c_size_t(out.nbytes)), "zg_memcpy_d2h")
        return out

    def free(self):
        if self.ptr:
            _chk(_lib.zg_dev_free(C.c_void_p(self.ptr)), "zg_dev_free")
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


PROF_NAMES = ["msm_digits", "msm_sort", "msm_accumulate", "msm_reduce", "eq_table", "sc_fold", "sc_sums", "combine"]


def profile_begin(max_records=4096):
    _chk(_lib.zg_profile_begin(C.c_int(max_records)), "zg_profile_begin")


def profile_end():
    """-> {kernel name: (total ms, launches)} measured with HIP events on the launch stream"""
    ms = (C.c_double * len(PROF_NAMES))()
    cnt = (C.c_uint64 * len(PROF_NAMES))()
    _chk(_lib.zg_profile_end(ms, cnt), "zg_profile_end")
    return {n: (float(ms[i]), int(cnt[i])) for i, n in enumerate(PROF_NAMES)}


# ---- field vectors
def field_op(field, op, a, b=None):
    a = _c(a)
    b = _c(b)
    out = np.empty_like(a)
    _chk(_lib.zg_field_op(C.c_int(field), C.c_int(op), _h(a), _h(b), _h(out), C.c_size_t(a.size // 4)), "zg_field_op")
    return out


# ---- bases / MSM
class Bases:
    """Device-resident bases (the SRS). Mirrors HyperKZG SetupParams.powers_of_tau_g1."""

    def __init__(self, handle, n):
        self._h = handle
        self.n = n

    @classmethod
    def upload(cls, xy, inf=None, window_bits=0, precompute_levels=0, expected_uses=0):
        xy = _c(xy)
        inf = _c(inf, np.uint8)
        n = xy.size // 8
        cfg = MsmConfig(window_bits, precompute_levels, expected_uses)
        h = C.c_void_p()
        _chk(_lib.zg_g1_bases_upload(_h(xy), _hb(inf), C.c_size_t(n), C.byref(cfg), C.byref(h)), "zg_g1_bases_upload")
        return cls(h, n)

    @classmethod
    def upload_dev(cls, d_xy, d_inf, n, stream=0, window_bits=0, precompute_levels=0):
        cfg = MsmConfig(window_bits, precompute_levels, 0)
        h = C.c_void_p()
        _chk(_lib.zg_g1_bases_upload_dev(_d(d_xy), _d(d_inf), C.c_size_t(n), C.byref(cfg), _d(stream), C.byref(h)),
             "zg_g1_bases_upload_dev")
        return cls(h, n)

    def msm_u64(self, values, n=None, off=0):
        """MSM over scalars given as u64 machine words (zg_msm_g1_u64): = msm(F.fromU64 of every word)"""
        v = np.ascontiguousarray(values, dtype=np.uint64).reshape(-1)
        n = v.size if n is None else n
        out = np.empty(8, dtype=np.uint64)
        inf = C.c_uint8(0)
        _chk(_lib.zg_msm_g1_u64(self._h, C.c_size_t(off), C.c_size_t(n), _h(v), _h(out), C.byref(inf)), "zg_msm_g1_u64")
        return out, int(inf.value)

    @classmethod
    def hyperkzg_setup(cls, base_xy, tau, n, want_points=True, window_bits=0, precompute_levels=0, expected_uses=0):
        """HyperKZG.setup's G1 side on the device (zg_hyperkzg_setup): -> (handle, points xy (n, 8) or None, inf (n,) or None)"""
        cfg = MsmConfig(window_bits, precompute_levels, expected_uses)
        xy = np.empty((n, 8), dtype=np.uint64) if want_points else None
        inf = np.zeros(n, dtype=np.uint8) if want_points else None
        h = C.c_void_p()
        _chk(_lib.zg_hyperkzg_setup(_h(_c(base_xy)), _h(_c(tau)), C.c_size_t(n), C.byref(cfg), _h(xy), _hb(inf), C.byref(h)), "zg_hyperkzg_setup")
        return cls(h, n), xy, inf

    def table_bytes(self):
        """bytes of HBM held for the bases: the table of precomputed multiples (levels x 64 B per base), or the plain bases"""
        return int(_lib.zg_g1_bases_table_bytes(self._h))

    def plan(self):
        """(window bits c, windows per scalar, table levels per base) the handle was built with"""
        c, w, l = C.c_int(), C.c_int(), C.c_int()
        _chk(_lib.zg_g1_bases_plan(self._h, C.byref(c), C.byref(w), C.byref(l)), "zg_g1_bases_plan")
        return c.value, w.value, l.value

    def free(self):
        if self._h:
            _chk(_lib.zg_g1_bases_free(self._h), "zg_g1_bases_free")
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def msm(self, scalars, off=0, n=None):
        scalars = _c(scalars)
        n = scalars.size // 4 if n is None else n
        out = np.empty(8, dtype=np.uint64)
        inf = C.c_uint8(0)
        _chk(_lib.zg_msm_g1(self._h, C.c_size_t(off), C.c_size_t(n), _h(scalars), _h(out), C.byref(inf)), "zg_msm_g1")
        return out, int(inf.value)

    def msm_dev(self, d_scalars, n, off=0, stream=0):
        out = np.empty(8, dtype=np.uint64)
        inf = C.c_uint8(0)
        _chk(_lib.zg_msm_g1_dev(self._h, C.c_size_t(off), C.c_size_t(n), _d(d_scalars), _d(stream), _h(out), C.byref(inf)),
             "zg_msm_g1_dev")
        return out, int(inf.value)

    def msm_dev_async(self, d_scalars, n, d_out_xy, d_out_inf, off=0, stream=0):
        _chk(_lib.zg_msm_g1_dev_async(self._h, C.c_size_t(off), C.c_size_t(n), _d(d_scalars), _d(stream), _d(d_out_xy),
                                      _d(d_out_inf)), "zg_msm_g1_dev_async")

    def msm_partial_dev(self, d_scalars, n, d_out_jac, off=0, stream=0):
        _chk(_lib.zg_msm_g1_partial_dev(self._h, C.c_size_t(off), C.c_size_t(n), _d(d_scalars), _d(stream), _d(d_out_jac)),
             "zg_msm_g1_partial_dev")

    def msm_partial_fast_dev(self, d_scalars, n, d_out_jac, off=0, stream=0):
        _chk(_lib.zg_msm_g1_partial_fast_dev(self._h, C.c_size_t(off), C.c_size_t(n), _d(d_scalars), _d(stream), _d(d_out_jac)),
             "zg_msm_g1_partial_fast_dev")

    def msm_batch_dev(self, d_scalars, n, k, d_out9, stream=0):
        """k vectors of n scalars back to back in HBM -> k records of 9 u64 (xy[8] + flag word) in HBM, asynchronous."""
        _chk(_lib.zg_msm_g1_batch_dev(self._h, C.c_size_t(n), _d(d_scalars), C.c_size_t(k), _d(stream), _d(d_out9)), "zg_msm_g1_batch_dev")

    def msm_batch(self, batches, n=None):
        batches = [_c(b) for b in batches]
        k = len(batches)
        n = (batches[0].size // 4 if k else 0) if n is None else n
        arr = (_u64p * max(k, 1))(*[_h(b) for b in batches])
        out = np.empty((k, 8), dtype=np.uint64)
        inf = np.zeros(k, dtype=np.uint8)
        _chk(_lib.zg_msm_g1_batch(self._h, C.c_size_t(n), arr, C.c_size_t(k), _h(out), _hb(inf)), "zg_msm_g1_batch")
        return out, inf


def pool_debug_stats():
    """ZG_POOL_DEBUG counters (include/zolt_gpu_internal.h): {'mode', 'hits', 'suspect_frees', 'blocks_verified', 'bytes_poisoned'}"""
    out = np.zeros(4, dtype=np.uint64)
    mode = int(_lib.zg_pool_debug_stats(_h(out)))
    return {"mode": mode, "hits": int(out[0]), "suspect_frees": int(out[1]), "blocks_verified": int(out[2]), "bytes_poisoned": int(out[3])}


def pool_debug_selftest():
    """breaks the pool's contract on purpose: 1 = ZG_POOL_DEBUG caught it, 0 = mode off; raises on an error"""
    rc = int(_lib.zg_pool_debug_selftest())
    if rc < 0:
        raise ZgError(-rc, "zg_pool_debug_selftest")
    return rc


def sharded_comm_sets_created():
    """test hook (include/zolt_gpu_internal.h): RCCL communicator sets created so far by the one-process multi-GPU path"""
    return int(_lib.zg_sharded_comm_sets_created())


def shard_bounds(n, shards, shard):
    """ParallelMSM's chunk `shard` of `shards` over n points -> (start, len); host arithmetic only (works without a GPU)"""
    st, ln = C.c_size_t(), C.c_size_t()
    _chk(_lib.zg_shard_bounds(C.c_size_t(n), C.c_int(shards), C.c_int(shard), C.byref(st), C.byref(ln)), "zg_shard_bounds")
    return st.value, ln.value


class ShardedBases:
    """The SRS sharded over the devices bound by init_devices (ParallelMSM's contiguous chunks, src/msm/mod.zig:609)."""
    EXCHANGE = {0: "none", 1: "rccl", 2: "p2p"}

    def __init__(self, handle, n):
        self._h = handle
        self.n = n

    @classmethod
    def upload(cls, xy, inf=None, window_bits=0, precompute_levels=0):
        xy = _c(xy)
        inf = _c(inf, np.uint8)
        n = xy.size // 8
        cfg = MsmConfig(window_bits, precompute_levels, 0)
        h = C.c_void_p()
        _chk(_lib.zg_g1_bases_upload_sharded(_h(xy), _hb(inf), C.c_size_t(n), C.byref(cfg), C.byref(h)), "zg_g1_bases_upload_sharded")
        return cls(h, n)

    def shards(self):
        """[(device, start, len)] per shard"""
        out = []
        for i in range(int(_lib.zg_g1_sbases_shards(self._h))):
            d, st, ln = C.c_int(), C.c_size_t(), C.c_size_t()
            _chk(_lib.zg_g1_sbases_shard(self._h, C.c_int(i), C.byref(d), C.byref(st), C.byref(ln)), "zg_g1_sbases_shard")
            out.append((d.value, st.value, ln.value))
        return out

    def exchange(self):
        return self.EXCHANGE[int(_lib.zg_g1_sbases_exchange(self._h))]

    def msm(self, scalars, n=None):
        scalars = _c(scalars)
        n = scalars.size // 4 if n is None else n
        out = np.empty(8, dtype=np.uint64)
        inf = C.c_uint8(0)
        _chk(_lib.zg_msm_g1_sharded(self._h, C.c_size_t(n), _h(scalars), _h(out), C.byref(inf)), "zg_msm_g1_sharded")
        return out, int(inf.value)

    def msm_dev(self, d_scalars_per_shard, n):
        arr = (C.c_void_p * len(d_scalars_per_shard))(*[int(p) if p else None for p in d_scalars_per_shard])
        out = np.empty(8, dtype=np.uint64)
        inf = C.c_uint8(0)
        _chk(_lib.zg_msm_g1_sharded_dev(self._h, C.c_size_t(n), arr, _h(out), C.byref(inf)), "zg_msm_g1_sharded_dev")
        return out, int(inf.value)

    def msm_batch(self, batches, n=None):
        batches = [_c(b) for b in batches]
        k = len(batches)
        n = (batches[0].size // 4 if k else 0) if n is None else n
        arr = (_u64p * max(k, 1))(*[_h(b) for b in batches])
        out = np.empty((k, 8), dtype=np.uint64)
        inf = np.zeros(k, dtype=np.uint8)
        _chk(_lib.zg_msm_g1_batch_sharded(self._h, C.c_size_t(n), arr, C.c_size_t(k), _h(out), _hb(inf)), "zg_msm_g1_batch_sharded")
        return out, inf

    def inflight(self):
        """sharded calls the handle keeps in flight (its slots)"""
        return int(_lib.zg_g1_sbases_inflight(self._h))

    def msm_dev_async(self, d_scalars_per_shard, n, ready_streams=None):
        """-> ticket; the result comes from wait(ticket). ready_streams: per shard the stream whose work fills its scalars (or None)"""
        arr = (C.c_void_p * len(d_scalars_per_shard))(*[int(p) if p else None for p in d_scalars_per_shard])
        rs = None
        if ready_streams is not None:
            rs = (C.c_void_p * len(ready_streams))(*[int(p) if p else None for p in ready_streams])
        t = C.c_uint64(0)
        _chk(_lib.zg_msm_g1_sharded_dev_async(self._h, C.c_size_t(n), arr, rs, C.byref(t)), "zg_msm_g1_sharded_dev_async")
        return int(t.value)

    def msm_batch_async(self, batches, n=None):
        """-> (ticket, k, keepalive); the host vectors must stay alive until wait(ticket)"""
        batches = [_c(b) for b in batches]
        k = len(batches)
        n = (batches[0].size // 4 if k else 0) if n is None else n
        arr = (_u64p * max(k, 1))(*[_h(b) for b in batches])
        t = C.c_uint64(0)
        _chk(_lib.zg_msm_g1_batch_sharded_async(self._h, C.c_size_t(n), arr, C.c_size_t(k), C.byref(t)), "zg_msm_g1_batch_sharded_async")
        return int(t.value), k, batches

    def wait(self, ticket, k=1):
        out = np.empty((k, 8), dtype=np.uint64)
        inf = np.zeros(k, dtype=np.uint8)
        _chk(_lib.zg_sharded_wait(self._h, C.c_uint64(ticket), _h(out), _hb(inf)), "zg_sharded_wait")
        return out, inf

    def free(self):
        if self._h:
            _chk(_lib.zg_g1_sbases_free(self._h), "zg_g1_sbases_free")
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def g1_affine_add_batch(a_xy, a_inf, b_xy, b_inf):
    """AffinePoint.add per pair (src/msm/mod.zig:74-103) -> (xy (n,8), inf (n,))"""
    a_xy, b_xy = _c(a_xy), _c(b_xy)
    a_inf, b_inf = _c(a_inf, np.uint8), _c(b_inf, np.uint8)
    n = a_xy.size // 8
    out = np.empty((n, 8), dtype=np.uint64)
    oinf = np.zeros(n, dtype=np.uint8)
    _chk(_lib.zg_g1_affine_add_batch(_h(a_xy), _hb(a_inf), _h(b_xy), _hb(b_inf), C.c_size_t(n), _h(out), _hb(oinf)), "zg_g1_affine_add_batch")
    return out, oinf


def fr_scale(a, s):
    a = _c(a)
    out = np.empty_like(a)
    _chk(_lib.zg_fr_scale(_h(a), C.c_size_t(a.size // 4), _h(_c(s)), _h(out)), "zg_fr_scale")
    return out


def combine_partials_dev(d_partials, k, stream=0):
    out = np.empty(8, dtype=np.uint64)
    inf = C.c_uint8(0)
    _chk(_lib.zg_g1_combine_partials_dev(_d(d_partials), C.c_size_t(k), _d(stream), _h(out), C.byref(inf)),
         "zg_g1_combine_partials_dev")
    return out, int(inf.value)


def combine_partials_dev_async(d_partials, k, d_out_xy, d_out_inf, stream=0):
    _chk(_lib.zg_g1_combine_partials_dev_async(_d(d_partials), C.c_size_t(k), _d(stream), _d(d_out_xy), _d(d_out_inf)),
         "zg_g1_combine_partials_dev_async")


def combine_partials_batch_dev_async(d_partials, ranks, rank_stride, m, d_out9, stream=0):
    """m results behind one exchange: rank r's m records at d_partials + r * rank_stride words; result j at d_out9 + 9 j"""
    _chk(_lib.zg_g1_combine_partials_batch_dev_async(_d(d_partials), C.c_size_t(ranks), C.c_size_t(rank_stride), C.c_size_t(m), _d(stream),
                                                     _d(d_out9)), "zg_g1_combine_partials_batch_dev_async")


def g1_scalar_mul_batch(xy, inf, scalars):
    xy, inf, scalars = _c(xy), _c(inf, np.uint8), _c(scalars)
    n = scalars.size // 4
    out = np.empty((n, 8), dtype=np.uint64)
    oinf = np.zeros(n, dtype=np.uint8)
    _chk(_lib.zg_g1_scalar_mul_batch(_h(xy), _hb(inf), _h(scalars), C.c_size_t(n), _h(out), _hb(oinf)), "zg_g1_scalar_mul_batch")
    return out, oinf


def g1_fixed_base_mul_batch(base_xy, scalars, base_inf=0):
    """scalars[i] * base for one shared base (HyperKZG.setup) -> (xy (n,8), inf (n,))"""
    base_xy, scalars = _c(base_xy), _c(scalars)
    n = scalars.size // 4
    out = np.empty((n, 8), dtype=np.uint64)
    oinf = np.zeros(n, dtype=np.uint8)
    _chk(_lib.zg_g1_fixed_base_mul_batch(_h(base_xy), C.c_uint8(base_inf), _h(scalars), C.c_size_t(n), _h(out), _hb(oinf)),
         "zg_g1_fixed_base_mul_batch")
    return out, oinf


def g1_is_on_curve_batch(xy, inf=None):
    xy, inf = _c(xy), _c(inf, np.uint8)
    n = xy.size // 8
    out = np.zeros(n, dtype=np.uint8)
    _chk(_lib.zg_g1_is_on_curve_batch(_h(xy), _hb(inf), C.c_size_t(n), _hb(out)), "zg_g1_is_on_curve_batch")
    return out


def fr_dense_evaluate(evals, point):
    evals, point = _c(evals), _c(point)
    out = np.empty(4, dtype=np.uint64)
    _chk(_lib.zg_fr_dense_evaluate(_h(evals), C.c_size_t(point.size // 4), _h(point), _h(out)), "zg_fr_dense_evaluate")
    return out


def hyperkzg_open(bases, evals, point, value):
    """HyperKZG.open on the device -> (quotient commitments (v,8), inf flags (v,), final_eval)."""
    evals, point, value = _c(evals), _c(point), _c(value)
    v = point.size // 4
    q = np.zeros((v, 8), dtype=np.uint64)
    qinf = np.zeros(v, dtype=np.uint8)
    fin = np.zeros(4, dtype=np.uint64)
    _chk(_lib.zg_hyperkzg_open(bases._h, _h(evals), C.c_size_t(evals.size // 4), _h(point), C.c_size_t(v), _h(value), _h(q),
                               _hb(qinf), _h(fin)), "zg_hyperkzg_open")
    return q, qinf, fin


def hyperkzg_open_dev(bases, d_evals, n_evals, point, value, stream=0):
    """HyperKZG.open with the table resident in HBM -> (quotient commitments (v,8), inf flags (v,), final_eval)."""
    point, value = _c(point), _c(value)
    v = point.size // 4
    q = np.zeros((v, 8), dtype=np.uint64)
    qinf = np.zeros(v, dtype=np.uint8)
    fin = np.zeros(4, dtype=np.uint64)
    _chk(_lib.zg_hyperkzg_open_dev(bases._h, _d(d_evals), C.c_size_t(n_evals), _h(point), C.c_size_t(v), _h(value), _d(stream), _h(q),
                                   _hb(qinf), _h(fin)), "zg_hyperkzg_open_dev")
    return q, qinf, fin


def hyperkzg_batch_open(bases, polys, point):
    """HyperKZG.batchOpen -> (quotient xy[nq,8], inf[nq], evaluations[k,4], final_eval, gamma)."""
    polys = [_c(p) for p in polys]
    point = _c(point)
    k, v = len(polys), point.size // 4
    ptrs = (C.c_void_p * max(k, 1))(*[p.ctypes.data if p.size else None for p in polys])
    lens = (C.c_size_t * max(k, 1))(*[p.size // 4 for p in polys])
    q = np.zeros((max(v, 1), 8), dtype=np.uint64)
    qi = np.zeros(max(v, 1), dtype=np.uint8)
    nq = C.c_size_t(0)
    ev = np.zeros((max(k, 1), 4), dtype=np.uint64)
    fin = np.zeros(4, dtype=np.uint64)
    gam = np.zeros(4, dtype=np.uint64)
    _chk(_lib.zg_hyperkzg_batch_open(bases._h, ptrs, lens, C.c_size_t(k), _h(point), C.c_size_t(v), _h(q), _hb(qi), C.byref(nq), _h(ev),
                                     _h(fin), _h(gam)), "zg_hyperkzg_batch_open")
    return q[:nq.value], qi[:nq.value], ev[:k], fin, gam


# ---- poly
def fr_eq_table(r, scale=None):
    r = _c(r)
    v = r.size // 4
    out = np.empty((1 << v, 4), dtype=np.uint64)
    _chk(_lib.zg_fr_eq_table(_h(r), C.c_size_t(v), _h(_c(scale)), _h(out)), "zg_fr_eq_table")
    return out


def fr_eq_table_dev(r, d_out, scale=None, stream=0):
    r = _c(r)
    _chk(_lib.zg_fr_eq_table_dev(_h(r), C.c_size_t(r.size // 4), _h(_c(scale)), _d(d_out), _d(stream)), "zg_fr_eq_table_dev")


def fr_eq_plus_one_table(r):
    """out[j] = EqPlusOnePolynomial.mle(r, bits(j)) (zg_fr_eq_plus_one_table)"""
    r = _c(np.asarray(r, dtype=np.uint64).reshape(-1, 4))
    out = np.empty((1 << r.shape[0], 4), dtype=np.uint64)
    _chk(_lib.zg_fr_eq_plus_one_table(_h(r), C.c_size_t(r.shape[0]), _h(out)), "zg_fr_eq_plus_one_table")
    return out


def fr_eq_plus_one_table_dev(r, d_out, stream=0):
    r = _c(np.asarray(r, dtype=np.uint64).reshape(-1, 4))
    _chk(_lib.zg_fr_eq_plus_one_table_dev(_h(r), C.c_size_t(r.shape[0]), _d(d_out), _d(stream)), "zg_fr_eq_plus_one_table_dev")


def fr_eq_prefix_tables(tau):
    """GruenSplitEqPolynomial's prefix tables (split_eq.zig:122-171) of tau[0..v): list of v+1 arrays, table k = eq(tau[0..k), .)"""
    tau = _c(np.asarray(tau, dtype=np.uint64).reshape(-1, 4))
    v = tau.shape[0]
    out = np.empty(((2 << v) - 1, 4), dtype=np.uint64)
    _chk(_lib.zg_fr_eq_prefix_tables(_h(tau), C.c_size_t(v), _h(out)), "zg_fr_eq_prefix_tables")
    return [out[(1 << k) - 1:(2 << k) - 1] for k in range(v + 1)]


def fr_eq_prefix_tables_dev(tau, d_out, stream=0):
    tau = _c(np.asarray(tau, dtype=np.uint64).reshape(-1, 4))
    _chk(_lib.zg_fr_eq_prefix_tables_dev(_h(tau), C.c_size_t(tau.shape[0]), _d(d_out), _d(stream)), "zg_fr_eq_prefix_tables_dev")


def fr_rows_mle(rows, r):
    """out[i] = sum_t eq(r, t) * rows[t][i] for a cycle-major (T, k, 4) matrix — R1CSInputEvaluator.computeClaimedInputs (zg_fr_rows_mle)"""
    rows = _c(rows)
    assert rows.ndim == 3 and rows.shape[2] == 4
    r = _c(np.asarray(r, dtype=np.uint64).reshape(-1, 4))
    out = np.empty((rows.shape[1], 4), dtype=np.uint64)
    _chk(_lib.zg_fr_rows_mle(_h(rows), C.c_size_t(rows.shape[0]), C.c_size_t(rows.shape[1]), _h(r), C.c_size_t(r.shape[0]), _h(out)), "zg_fr_rows_mle")
    return out


def fr_rows_mle_dev(d_rows, n_rows, k, r, stream=0):
    r = _c(np.asarray(r, dtype=np.uint64).reshape(-1, 4))
    out = np.empty((k, 4), dtype=np.uint64)
    _chk(_lib.zg_fr_rows_mle_dev(_d(d_rows), C.c_size_t(n_rows), C.c_size_t(k), _h(r), C.c_size_t(r.shape[0]), _d(stream), _h(out)), "zg_fr_rows_mle_dev")
    return out


def fr_rows_affine(rows, coeffs, ntab, g, n_pad=None, k=None):
    """tables[t][i * g + j] = C[t*g + j][k] + sum_col C[t*g + j][col] * rows[i][col] (zg_fr_rows_affine): rows (T, stride, 4) cycle-major,
    coeffs (ntab * g, k + 1, 4) with the constant last -> ntab arrays of (n_pad * g, 4). k > stride: a sliding window — the maps of row i
    read into the rows after it, and the last k - stride elements of `rows` are only ever read as such a tail (n_rows = (T * stride - k) //
    stride + 1)."""
    rows, coeffs = _c(rows), _c(coeffs)
    stride = rows.shape[1]
    k = stride if k is None else k
    assert rows.ndim == 3 and rows.shape[2] == 4 and coeffs.shape == (ntab * g, k + 1, 4) and k >= stride
    n_rows = rows.shape[0] if k == stride else (rows.shape[0] * stride - k) // stride + 1
    n_pad = n_rows if n_pad is None else n_pad
    outs = [np.empty((n_pad * g, 4), dtype=np.uint64) for _ in range(ntab)]
    ptrs = (C.c_void_p * ntab)(*[o.ctypes.data for o in outs])
    _chk(_lib.zg_fr_rows_affine(_h(rows), C.c_size_t(n_rows), C.c_size_t(k), C.c_size_t(stride), _h(coeffs), C.c_size_t(ntab), C.c_size_t(g),
                                C.c_size_t(n_pad), ptrs), "zg_fr_rows_affine")
    return outs


COL_ZERO, COL_U8, COL_U32, COL_U64, COL_I64, COL_I128, COL_U128, COL_FR, COL_BIT, COL_MUL, COL_LUT = range(11)
_COL_DTYPE = {COL_U8: np.uint8, COL_U32: np.uint32, COL_U64: np.uint64, COL_I64: np.int64, COL_I128: np.uint64, COL_U128: np.uint64, COL_FR: np.uint64,
              COL_MUL: np.uint64}  # (COL_MUL's optional addend is a 128-bit two's-complement column)


class Column(C.Structure):
    """zg_col_t: one typed integer column of zg_fr_rows_from_columns"""
    _fields_ = [("kind", C.c_uint32), ("a", C.c_uint32), ("b", C.c_uint32), ("data", C.c_void_p), ("aux", C.c_void_p)]


def _columns(cols, n_rows, device):
    """cols: list of (kind, data, a, b[, aux]) — data a numpy array (host) or a device address (device=True), None for COL_ZERO / COL_MUL;
    COL_LUT: data = the index per row (a bytes each), aux = the table of b elements"""
    arr = (Column * len(cols))()
    keep = []
    for i, spec in enumerate(cols):
        kind, data = spec[0], spec[1] if len(spec) > 1 else None
        a, b = (spec[2] if len(spec) > 2 else 0), (spec[3] if len(spec) > 3 else 0)
        aux = spec[4] if len(spec) > 4 else None
        ptr = aptr = None
        if aux is not None:
            if device:
                aptr = int(aux)
            else:
                t = np.ascontiguousarray(aux, dtype=np.uint64)
                assert t.size == 4 * b, (i, t.shape, b)
                keep.append(t)
                aptr = t.ctypes.data
        if data is not None:
            if device:
                ptr = int(data)
            else:
                dt = {1: np.uint8, 2: np.uint16, 4: np.uint32}[a] if kind == COL_LUT else (_COL_DTYPE.get(kind) or {1: np.uint8, 4: np.uint32, 8: np.uint64}[b])
                h = np.ascontiguousarray(data, dtype=dt)
                per = {COL_I128: 2, COL_U128: 2, COL_FR: 4, COL_MUL: 2}.get(kind, 1)
                assert h.size == n_rows * per, (i, kind, h.shape, n_rows)
                keep.append(h)
                ptr = h.ctypes.data
        arr[i] = Column(kind, a, b, ptr, aptr)
    return arr, keep


def fr_rows_from_columns(cols, n_rows, d_rows=None):
    """The witness matrix built on the device from integer columns (zg_fr_rows_from_columns): rows[row][c] = column c's value at `row`
    as a Montgomery element. d_rows: a device address to fill (returns None); None: the (n_rows, n_cols, 4) matrix comes back to the host."""
    arr, keep = _columns(cols, n_rows, device=False)
    buf = None
    if d_rows is None:
        buf = DeviceBuffer(max(n_rows * len(cols) * 32, 32))
        d_rows = buf.ptr
    _chk(_lib.zg_fr_rows_from_columns(arr, C.c_size_t(len(cols)), C.c_size_t(n_rows), _d(d_rows)), "zg_fr_rows_from_columns")
    del keep
    if buf is None:
        return None
    out = buf.to_host()[:n_rows * len(cols) * 4].reshape(n_rows, len(cols), 4)
    buf.free()
    return out


def fr_rows_from_columns_dev(cols, n_rows, d_rows, stream=0):
    """the same with the columns resident in HBM (data = device addresses); asynchronous"""
    arr, _ = _columns(cols, n_rows, device=True)
    _chk(_lib.zg_fr_rows_from_columns_dev(arr, C.c_size_t(len(cols)), C.c_size_t(n_rows), _d(d_rows), _d(stream)), "zg_fr_rows_from_columns_dev")


def fr_lt_table(r):
    """out[j] = LtPolynomial(r).evaluateAtIndex(j) over the cube (zg_fr_lt_table); index bit i <-> r[i]"""
    r = _c(np.asarray(r, dtype=np.uint64).reshape(-1, 4))
    out = np.empty((1 << r.shape[0], 4), dtype=np.uint64)
    _chk(_lib.zg_fr_lt_table(_h(r), C.c_size_t(r.shape[0]), _h(out)), "zg_fr_lt_table")
    return out


def fr_lt_table_dev(r, d_out, stream=0):
    r = _c(np.asarray(r, dtype=np.uint64).reshape(-1, 4))
    _chk(_lib.zg_fr_lt_table_dev(_h(r), C.c_size_t(r.shape[0]), _d(d_out), _d(stream)), "zg_fr_lt_table_dev")


def fr_write_tables_dev(n, cycle, word, pre, post, r_eq, d_inc, d_wa, stream=0):
    """ValEvaluation's inc / wa (n entries each, device) from the list of writes (zg_fr_write_tables_dev): inc[cycle] = F(post) - F(pre),
    wa[cycle] = eq(r_eq, word); r_eq (log_k, 4) in EqPolynomial order"""
    cycle, word, pre, post = _c(cycle, np.uint32), _c(word, np.uint32), _c(pre), _c(post)
    r_eq = _c(np.asarray(r_eq, dtype=np.uint64).reshape(-1, 4))
    m = cycle.size
    assert word.size == pre.size == post.size == m
    _chk(_lib.zg_fr_write_tables_dev(C.c_size_t(n), C.c_size_t(m), _hb(cycle), _hb(word), _h(pre), _h(post), _h(r_eq) if r_eq.size else None,
                                     C.c_size_t(r_eq.shape[0]), _d(d_inc), _d(d_wa), _d(stream)), "zg_fr_write_tables_dev")


def fr_weighted_colsum(table, rows, cols, weights):
    """out[k][c] = sum_r weights[k][r] * table[r * cols + c] (zg_fr_weighted_colsum): table (rows * cols, 4), weights (m, rows, 4), m <= 4
    -> (m, cols, 4)"""
    table, weights = _c(table), _c(weights)
    m = weights.size // (4 * rows)
    assert table.size == rows * cols * 4 and weights.size == m * rows * 4
    out = np.empty((m, cols, 4), dtype=np.uint64)
    _chk(_lib.zg_fr_weighted_colsum(_h(table), C.c_size_t(rows), C.c_size_t(cols), _h(weights), C.c_size_t(m), _h(out)), "zg_fr_weighted_colsum")
    return out


def fr_weighted_colsum_dev(d_table, rows, cols, d_weights, m, d_out, stream=0):
    _chk(_lib.zg_fr_weighted_colsum_dev(_d(d_table), C.c_size_t(rows), C.c_size_t(cols), _d(d_weights), C.c_size_t(m), _d(d_out), _d(stream)),
         "zg_fr_weighted_colsum_dev")


def fr_rows_affine_dev(d_rows, n_rows, k, coeffs, ntab, g, n_pad, d_tables, stream=0, stride=0):
    coeffs = _c(coeffs)
    assert coeffs.size == ntab * g * (k + 1) * 4 and len(d_tables) == ntab
    ptrs = (C.c_void_p * ntab)(*[int(p) for p in d_tables])
    _chk(_lib.zg_fr_rows_affine_dev(_d(d_rows), C.c_size_t(n_rows), C.c_size_t(k), C.c_size_t(stride), _h(coeffs), C.c_size_t(ntab), C.c_size_t(g),
                                    C.c_size_t(n_pad), ptrs, _d(stream)), "zg_fr_rows_affine_dev")


def fr_rows_affine_records_dev(d_rows, n_rows, k, coeffs, record, first, d_out, stream=0, stride=0):
    """d_out[i * record + first + c] = map_c(row_i) for the nout <= 16 maps of coeffs (nout, k + 1, 4) (zg_fr_rows_affine_records_dev)"""
    coeffs = _c(coeffs)
    nout = coeffs.shape[0]
    assert coeffs.shape == (nout, k + 1, 4)
    _chk(_lib.zg_fr_rows_affine_records_dev(_d(d_rows), C.c_size_t(n_rows), C.c_size_t(k), C.c_size_t(stride), _h(coeffs), C.c_size_t(nout), C.c_size_t(record),
                                            C.c_size_t(first), _d(d_out), _d(stream)), "zg_fr_rows_affine_records_dev")


def fr_rows_affine_prodsum_dev(d_rows, n_rows, k, coeffs, npairs, d_weights, g, stream=0, stride=0):
    """out[p] = sum_i W[i * g + p % g] * A_p(row_i) * B_p(row_i) (zg_fr_rows_affine_prodsum_dev) -> (npairs, 4)"""
    coeffs = _c(coeffs)
    assert coeffs.size == 2 * npairs * (k + 1) * 4
    out = np.empty((npairs, 4), dtype=np.uint64)
    _chk(_lib.zg_fr_rows_affine_prodsum_dev(_d(d_rows), C.c_size_t(n_rows), C.c_size_t(k), C.c_size_t(stride), _h(coeffs), C.c_size_t(npairs), _d(d_weights),
                                            C.c_size_t(g), _h(out), _d(stream)), "zg_fr_rows_affine_prodsum_dev")
    return out


def fr_bind_low(table, r):
    t = np.array(table, dtype=np.uint64, copy=True).reshape(-1, 4)
    n = t.shape[0]
    _chk(_lib.zg_fr_bind_low(_h(t), C.c_size_t(n), _h(_c(r))), "zg_fr_bind_low")
    return t[: n // 2].copy()


def fr_bind_high(table, r):
    t = _c(table).reshape(-1, 4)
    n = t.shape[0]
    out = np.empty((n // 2, 4), dtype=np.uint64)
    _chk(_lib.zg_fr_bind_high(_h(t), C.c_size_t(n), _h(_c(r)), _h(out)), "zg_fr_bind_high")
    return out


def fr_spartan_combine(eq, az, bz, cz):
    eq, az, bz, cz = _c(eq), _c(az), _c(bz), _c(cz)
    out = np.empty_like(eq)
    _chk(_lib.zg_fr_spartan_combine(_h(eq), _h(az), _h(bz), _h(cz), C.c_size_t(eq.size // 4), _h(out)), "zg_fr_spartan_combine")
    return out


def fr_spartan_combine_dev(d_eq, d_az, d_bz, d_cz, n, d_out, stream=0):
    _chk(_lib.zg_fr_spartan_combine_dev(_d(d_eq), _d(d_az), _d(d_bz), _d(d_cz), C.c_size_t(n), _d(d_out), _d(stream)),
         "zg_fr_spartan_combine_dev")


def fr_bit_split_sums(vals, idx128, bit):
    """sums of vals[j] split by bit `bit` of the u128 index (n,2) -> (sum0, sum1)"""
    vals, idx128 = _c(vals), _c(idx128)
    s0 = np.empty(4, dtype=np.uint64)
    s1 = np.empty(4, dtype=np.uint64)
    _chk(_lib.zg_fr_bit_split_sums(_h(vals), _h(idx128), C.c_size_t(vals.size // 4), C.c_uint(bit), _h(s0), _h(s1)), "zg_fr_bit_split_sums")
    return s0, s1


def fr_bit_split_sums_dev(d_vals, d_idx128, n, bit, stream=0):
    s0 = np.empty(4, dtype=np.uint64)
    s1 = np.empty(4, dtype=np.uint64)
    _chk(_lib.zg_fr_bit_split_sums_dev(_d(d_vals), _d(d_idx128), C.c_size_t(n), C.c_uint(bit), _d(stream), _h(s0), _h(s1)),
         "zg_fr_bit_split_sums_dev")
    return s0, s1


def selftest_handoff(blocks, threads=512, iters=200, busy=True):
    """zg_selftest_handoff: (words that differed, launches that elected exactly one last arriver)"""
    bad = C.c_uint64(0)
    done = C.c_uint64(0)
    _chk(_lib.zg_selftest_handoff(C.c_uint(blocks), C.c_uint(threads), C.c_uint(iters), C.c_int(1 if busy else 0), C.byref(bad), C.byref(done)),
         "zg_selftest_handoff")
    return int(bad.value), int(done.value)


class SumcheckVerificationFailed(RuntimeError):
    """error.SumcheckVerificationFailed (src/subprotocols/mod.zig:175-178)."""


def _run_sumcheck_call(fn, ptr, n, extra, where):
    v = max(n.bit_length() - 1, 0)
    claim = np.empty(4, dtype=np.uint64)
    rounds = np.empty((v, 2, 4), dtype=np.uint64)
    chal = np.empty((v, 4), dtype=np.uint64)
    fin = np.empty(4, dtype=np.uint64)
    res = C.c_uint8(0)
    rc = fn(ptr, C.c_size_t(n), *extra, _h(claim), _h(rounds) if v else None, _h(chal) if v else None, _h(fin), C.byref(res))
    if rc == ERR_VERIFY:
        raise SumcheckVerificationFailed(last_error())
    _chk(rc, where)
    return {"claim": claim, "rounds": rounds, "final_point": chal, "final_eval": fin, "result": bool(res.value)}


def run_sumcheck(evals):
    """runSumcheck (src/subprotocols/mod.zig:302-354), prover and toy verifier both on the device."""
    e = _c(evals)
    return _run_sumcheck_call(_lib.zg_run_sumcheck, _h(e), e.size // 4, (), "zg_run_sumcheck")


def run_sumcheck_dev(d_evals, n, stream=0):
    return _run_sumcheck_call(_lib.zg_run_sumcheck_dev, _d(d_evals), n, (_d(stream),), "zg_run_sumcheck_dev")


class SumcheckSession:
    """Device-resident Sumcheck(F).Prover table (src/subprotocols/mod.zig:50-134)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def open(cls, evals, layout=SC_HIGH_HALF):
        e = _c(evals)
        h = C.c_void_p()
        _chk(_lib.zg_sumcheck_open(_h(e), C.c_size_t(e.size // 4), C.c_int(layout), C.byref(h)), "zg_sumcheck_open")
        return cls(h)

    @classmethod
    def open_dev(cls, d_evals, n, layout=SC_HIGH_HALF, stream=0, borrow=False):
        """borrow: no copy — the session reads the caller's device table until its first bind has run (zg_sumcheck_open_dev_borrowed);
        the caller keeps it alive and unchanged until then"""
        h = C.c_void_p()
        if borrow:
            _chk(_lib.zg_sumcheck_open_dev_borrowed(_d(d_evals), C.c_size_t(n), C.c_int(layout), _d(stream), C.byref(h)), "zg_sumcheck_open_dev_borrowed")
        else:
            _chk(_lib.zg_sumcheck_open_dev(_d(d_evals), C.c_size_t(n), C.c_int(layout), _d(stream), C.byref(h)), "zg_sumcheck_open_dev")
        return cls(h)

    @classmethod
    def open_column(cls, col, n_rows, n, layout=SC_HIGH_HALF):
        """a session whose table is ONE integer column widened on the device: entries [0, n_rows) from `col` (a fr_rows_from_columns
        column spec, host data), [n_rows, n) zero (zg_sumcheck_open_column)"""
        arr, keep = _columns([col], n_rows, device=False)
        h = C.c_void_p()
        _chk(_lib.zg_sumcheck_open_column(arr, C.c_size_t(n_rows), C.c_size_t(n), C.c_int(layout), C.byref(h)), "zg_sumcheck_open_column")
        del keep
        return cls(h)

    @classmethod
    def open_spartan_dev(cls, r, d_az, d_bz, d_cz, layout=SC_HIGH_HALF, scale=None, stream=0):
        """f = eq(r, .) * (Az*Bz - Cz) built straight into a new session (round 0's sums included), zg_sumcheck_open_spartan_dev."""
        r = _c(r)
        h = C.c_void_p()
        _chk(_lib.zg_sumcheck_open_spartan_dev(_h(r), C.c_size_t(r.size // 4), _h(_c(scale)), _d(d_az), _d(d_bz), _d(d_cz), C.c_int(layout),
                                               _d(stream), C.byref(h)), "zg_sumcheck_open_spartan_dev")
        return cls(h)

    def round_sums(self):
        g0 = np.empty(4, dtype=np.uint64)
        g1 = np.empty(4, dtype=np.uint64)
        _chk(_lib.zg_sumcheck_round_sums(self._h, _h(g0), _h(g1)), "zg_sumcheck_round_sums")
        return g0, g1

    def bind(self, r):
        _chk(_lib.zg_sumcheck_bind(self._h, _h(_c(r))), "zg_sumcheck_bind")

    def __len__(self):
        return int(_lib.zg_sumcheck_len(self._h))

    def final(self):
        out = np.empty(4, dtype=np.uint64)
        _chk(_lib.zg_sumcheck_final(self._h, _h(out)), "zg_sumcheck_final")
        return out

    def read(self):
        out = np.empty((len(self), 4), dtype=np.uint64)
        _chk(_lib.zg_sumcheck_read(self._h, _h(out)), "zg_sumcheck_read")
        return out

    def gather(self, idx):
        """table[idx[i]] of the current table -> (n, 4)"""
        idx = np.ascontiguousarray(idx, dtype=np.uint64)
        out = np.empty((idx.size, 4), dtype=np.uint64)
        _chk(_lib.zg_sumcheck_gather(self._h, _h(idx), C.c_size_t(idx.size), _h(out)), "zg_sumcheck_gather")
        return out

    def raf_round(self, base, current_power):
        """RAF cubic round sums s(0), s(2) over this LOW_PAIR session's table (zg_sumcheck_raf_round)"""
        s0 = np.empty(4, dtype=np.uint64)
        s2 = np.empty(4, dtype=np.uint64)
        _chk(_lib.zg_sumcheck_raf_round(self._h, _h(_c(base)), C.c_uint64(current_power), _h(s0), _h(s2)), "zg_sumcheck_raf_round")
        return s0, s2

    def raf_claim(self, base, step=8):
        """sum_k t[k] * F.fromU64(base + step * k) over the current table: RafEvaluationProver.computeInitialClaim (zg_sumcheck_raf_claim)"""
        out = np.empty(4, dtype=np.uint64)
        _chk(_lib.zg_sumcheck_raf_claim(self._h, C.c_uint64(base), C.c_uint64(step), _h(out)), "zg_sumcheck_raf_claim")
        return out

    def bit_round(self, d_idx128, n_idx, bit):
        """LassoProver.computeAddressRoundPoly's sum_0 / sum_1 over the session's first n_idx entries (zg_sumcheck_bit_round)"""
        s0 = np.empty(4, dtype=np.uint64)
        s1 = np.empty(4, dtype=np.uint64)
        _chk(_lib.zg_sumcheck_bit_round(self._h, _d(d_idx128), C.c_size_t(n_idx), C.c_uint(bit), _h(s0), _h(s1)), "zg_sumcheck_bit_round")
        return s0, s1

    def bit_bind(self, d_idx128, n_idx, bit, r):
        """LassoProver.receiveChallenge's address branch in place; -> the new claim (zg_sumcheck_bit_bind)"""
        claim = np.empty(4, dtype=np.uint64)
        _chk(_lib.zg_sumcheck_bit_bind(self._h, _d(d_idx128), C.c_size_t(n_idx), C.c_uint(bit), _h(_c(r)), _h(claim)), "zg_sumcheck_bit_bind")
        return claim

    def round_sums_dev(self, d_out8):
        _chk(_lib.zg_sumcheck_round_sums_dev(self._h, _d(d_out8)), "zg_sumcheck_round_sums_dev")

    def read_dev(self, d_out):
        _chk(_lib.zg_sumcheck_read_dev(self._h, _d(d_out)), "zg_sumcheck_read_dev")

    def close(self):
        if self._h:
            _chk(_lib.zg_sumcheck_close(self._h), "zg_sumcheck_close")
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


PSC_PAIR_SUM = 256  # ZG_PSC_PAIR_SUM


class PscTerm(C.Structure):
    """zg_psc_term: one product term of zg_psc_round_expr"""
    _fields_ = [("n_prod", C.c_int), ("prod", C.c_int * 4), ("n_lin", C.c_int), ("lin", C.c_int * 4), ("lin_coeff", C.c_uint64 * 16)]


class ProductSumcheckSession:
    """k multilinear tables folded together (LowToHigh) with product-form round evaluations (zg_psc_*): the loops of
    ValEvaluationProver, ValFinalProver, OutputSumcheckProver, InstructionLookupsClaimReduction and ProductVirtualRemainderProver."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def open(cls, tables):
        tabs = [_c(t).reshape(-1, 4) for t in tables]
        n = tabs[0].shape[0]
        assert all(t.shape[0] == n for t in tabs)
        ptrs = (C.c_void_p * len(tabs))(*[t.ctypes.data for t in tabs])
        h = C.c_void_p()
        _chk(_lib.zg_psc_open(ptrs, C.c_size_t(len(tabs)), C.c_size_t(n), C.byref(h)), "zg_psc_open")
        return cls(h)

    def table_dev(self, table):
        """device address of a table's current entries (zg_psc_table_dev): valid until the next bind / close"""
        p = C.c_void_p()
        _chk(_lib.zg_psc_table_dev(self._h, C.c_size_t(table), C.byref(p)), "zg_psc_table_dev")
        return p.value

    @classmethod
    def open_dev(cls, d_tables, n, stream=0):
        ptrs = (C.c_void_p * len(d_tables))(*[int(p) for p in d_tables])
        h = C.c_void_p()
        _chk(_lib.zg_psc_open_dev(ptrs, C.c_size_t(len(d_tables)), C.c_size_t(n), _d(stream), C.byref(h)), "zg_psc_open_dev")
        return cls(h)

    def __len__(self):
        return int(_lib.zg_psc_len(self._h))

    def tables(self):
        return int(_lib.zg_psc_tables(self._h))

    def round_evals(self, prod_idx, lin_idx=(), lin_coeff=None):
        """[p(0), p(1), p(2), p(3)] of sum_g prod_j T[prod_idx[j]](t) * sum_m lin_coeff[m] T[lin_idx[m]](t)"""
        pi = (C.c_int * max(len(prod_idx), 1))(*prod_idx)
        li = (C.c_int * max(len(lin_idx), 1))(*lin_idx)
        co = _c(lin_coeff) if len(lin_idx) else None
        out = np.empty((4, 4), dtype=np.uint64)
        _chk(_lib.zg_psc_round_evals(self._h, pi, C.c_size_t(len(prod_idx)), li, _h(co), C.c_size_t(len(lin_idx)), _h(out)), "zg_psc_round_evals")
        return out

    def round_expr(self, terms):
        """[p(0..3)] of a SUM of product terms; terms: list of (prod_idx, lin_idx, lin_coeff) with lin_coeff (len(lin_idx), 4) or None,
        or (prod_idx, lin_idx, lin_coeff, True) for a ZG_PSC_PAIR_SUM term: (T[p0] T[p1] + T[p2] T[p3]) * L"""
        arr = (PscTerm * len(terms))()
        for t, term in zip(arr, terms):
            prod_idx, lin_idx, coeff = term[:3]
            t.n_prod, t.n_lin = len(prod_idx) | (PSC_PAIR_SUM if len(term) > 3 and term[3] else 0), len(lin_idx)
            for j, v in enumerate(prod_idx):
                t.prod[j] = v
            for m, v in enumerate(lin_idx):
                t.lin[m] = v
            if len(lin_idx):
                flat = _c(coeff).reshape(-1)
                for i in range(4 * len(lin_idx)):
                    t.lin_coeff[i] = int(flat[i])
        out = np.empty((4, 4), dtype=np.uint64)
        _chk(_lib.zg_psc_round_expr(self._h, arr, C.c_size_t(len(terms)), _h(out)), "zg_psc_round_expr")
        return out

    def set_points(self, points):
        """bit t of `points`: the round calls compute p(t); the other slots come back as zero (zg_psc_set_points)"""
        _chk(_lib.zg_psc_set_points(self._h, C.c_uint(points)), "zg_psc_set_points")

    def round_gruen(self, prod_idx, d_e_out, n_out, d_e_in, n_in):
        """Gruen's (t0, t_inf) under the split-eq weights E_out x E_in (device pointers)"""
        pi = (C.c_int * max(len(prod_idx), 1))(*prod_idx)
        t0 = np.empty(4, dtype=np.uint64)
        ti = np.empty(4, dtype=np.uint64)
        _chk(_lib.zg_psc_round_gruen(self._h, pi, C.c_size_t(len(prod_idx)), _d(d_e_out), C.c_size_t(n_out), _d(d_e_in), C.c_size_t(n_in),
                                     _h(t0), _h(ti)), "zg_psc_round_gruen")
        return t0, ti

    def bind(self, r):
        _chk(_lib.zg_psc_bind(self._h, _h(_c(r))), "zg_psc_bind")

    def read(self, table):
        out = np.empty((len(self), 4), dtype=np.uint64)
        _chk(_lib.zg_psc_read(self._h, C.c_size_t(table), _h(out)), "zg_psc_read")
        return out

    def gather(self, table, idx):
        """T[table][idx[i]] of the current tables -> (n, 4)"""
        idx = np.ascontiguousarray(idx, dtype=np.uint64)
        out = np.empty((idx.size, 4), dtype=np.uint64)
        _chk(_lib.zg_psc_gather(self._h, C.c_size_t(table), _h(idx), C.c_size_t(idx.size), _h(out)), "zg_psc_gather")
        return out

    def final(self):
        out = np.empty((self.tables(), 4), dtype=np.uint64)
        _chk(_lib.zg_psc_final(self._h, _h(out)), "zg_psc_final")
        return out

    def close(self):
        if self._h:
            _chk(_lib.zg_psc_close(self._h), "zg_psc_close")
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RegistersRwSession:
    """Stage4GruenProver's dense tables on the device (zg_rrw_*)"""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def open(cls, log_t, rs1, rs2, rd, reg_vals, inc, gamma):
        rs1, rs2, rd = _c(rs1, np.uint8), _c(rs2, np.uint8), _c(rd, np.uint8)
        reg_vals, inc, gamma = _c(reg_vals), _c(inc), _c(gamma)
        T = 1 << log_t
        assert rs1.size == rs2.size == rd.size == T and reg_vals.size == 32 * T and inc.size == 4 * T
        h = C.c_void_p()
        _chk(_lib.zg_rrw_open(C.c_size_t(log_t), _hb(rs1), _hb(rs2), _hb(rd), _h(reg_vals), _h(inc), _h(gamma), C.byref(h)), "zg_rrw_open")
        return cls(h)

    @classmethod
    def open_trace(cls, log_t, rs1, rs2, rd, rd_value, gamma):
        """the same session from the write column alone (zg_rrw_open_trace): the register file and inc are rebuilt on the device"""
        rs1, rs2, rd = _c(rs1, np.uint8), _c(rs2, np.uint8), _c(rd, np.uint8)
        rd_value, gamma = _c(rd_value), _c(gamma)
        T = 1 << log_t
        assert rs1.size == rs2.size == rd.size == rd_value.size == T
        h = C.c_void_p()
        _chk(_lib.zg_rrw_open_trace(C.c_size_t(log_t), _hb(rs1), _hb(rs2), _hb(rd), _h(rd_value), _h(gamma), C.byref(h)), "zg_rrw_open_trace")
        return cls(h)

    def cycles(self):
        return int(_lib.zg_rrw_cycles(self._h))

    def registers(self):
        return int(_lib.zg_rrw_registers(self._h))

    def round_cycle_gruen(self, d_e_out, n_out, d_e_in, n_in):
        a, b = np.empty(4, dtype=np.uint64), np.empty(4, dtype=np.uint64)
        _chk(_lib.zg_rrw_round_cycle_gruen(self._h, _d(d_e_out), C.c_size_t(n_out), _d(d_e_in), C.c_size_t(n_in), _h(a), _h(b)), "zg_rrw_round_cycle_gruen")
        return a, b

    def set_eq(self, eq):
        eq = _c(eq)
        _chk(_lib.zg_rrw_set_eq(self._h, _h(eq), C.c_size_t(eq.size // 4)), "zg_rrw_set_eq")

    def round_address(self, with_e1=False):
        """(e0, e2), or (e0, e1, e2) with the t = 1 value computed from the tables"""
        a, b, c = (np.empty(4, dtype=np.uint64) for _ in range(3))
        _chk(_lib.zg_rrw_round_address(self._h, _h(a), _h(c) if with_e1 else None, _h(b)), "zg_rrw_round_address")
        return (a, c, b) if with_e1 else (a, b)

    def round_cycle(self, with_e1=False):
        """(e0, e2, e3), or (e0, e1, e2, e3)"""
        a, b, c, d = (np.empty(4, dtype=np.uint64) for _ in range(4))
        _chk(_lib.zg_rrw_round_cycle(self._h, _h(a), _h(d) if with_e1 else None, _h(b), _h(c)), "zg_rrw_round_cycle")
        return (a, d, b, c) if with_e1 else (a, b, c)

    def bind_cycle(self, r):
        _chk(_lib.zg_rrw_bind_cycle(self._h, _h(_c(r))), "zg_rrw_bind_cycle")

    def bind_address(self, r):
        _chk(_lib.zg_rrw_bind_address(self._h, _h(_c(r))), "zg_rrw_bind_address")

    def final(self):
        """{val, rd_wa, ra, rs1_ra, rs2_ra, inc, eq}: entry [0][0] of every table"""
        out = np.empty((7, 4), dtype=np.uint64)
        _chk(_lib.zg_rrw_final(self._h, _h(out)), "zg_rrw_final")
        return dict(zip(("val", "rd_wa", "ra", "rs1_ra", "rs2_ra", "inc", "eq"), out))

    def close(self):
        if self._h:
            _chk(_lib.zg_rrw_close(self._h), "zg_rrw_close")
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RamRwSession:
    """RamReadWriteCheckingProver's entry list and dense tables behind zg_rwc_*: the cycle-phase walk and the field arithmetic on the
    device, the address-phase walk on the host inside the library"""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def open(cls, log_k, log_t, cycle, address, val_coeff, prev_val, next_val, inc, val_init, r_cycle):
        cycle, address = _c(cycle, np.uint32), _c(address, np.uint32)
        val_coeff, prev_val, next_val = _c(val_coeff), _c(prev_val), _c(next_val)
        inc, val_init, r_cycle = _c(inc), _c(val_init), _c(r_cycle)
        n = cycle.size
        assert address.size == val_coeff.size == prev_val.size == next_val.size == n
        assert inc.size == 4 << log_t and val_init.size == 4 << log_k and r_cycle.size == 4 * log_t
        h = C.c_void_p()
        _chk(_lib.zg_rwc_open(C.c_size_t(log_k), C.c_size_t(log_t), C.c_size_t(n), _hb(cycle), _hb(address), _h(val_coeff), _h(prev_val), _h(next_val),
                              _h(inc), _h(val_init), _h(r_cycle), C.byref(h)), "zg_rwc_open")
        return cls(h)

    @classmethod
    def open_writes(cls, log_k, log_t, cycle, address, val_coeff, prev_val, next_val, is_write, val_init, r_cycle):
        """zg_rwc_open_writes: inc is formed on the device from the entries marked as writes"""
        cycle, address, is_write = _c(cycle, np.uint32), _c(address, np.uint32), _c(is_write, np.uint8)
        val_coeff, prev_val, next_val = _c(val_coeff), _c(prev_val), _c(next_val)
        val_init, r_cycle = _c(val_init), _c(r_cycle)
        n = cycle.size
        assert address.size == val_coeff.size == prev_val.size == next_val.size == is_write.size == n
        assert val_init.size == 4 << log_k and r_cycle.size == 4 * log_t
        h = C.c_void_p()
        _chk(_lib.zg_rwc_open_writes(C.c_size_t(log_k), C.c_size_t(log_t), C.c_size_t(n), _hb(cycle), _hb(address), _h(val_coeff), _h(prev_val),
                                     _h(next_val), _hb(is_write), _h(val_init), _h(r_cycle), C.byref(h)), "zg_rwc_open_writes")
        return cls(h)

    def entries(self):
        return int(_lib.zg_rwc_entries(self._h))

    def cycles(self):
        return int(_lib.zg_rwc_cycles(self._h))

    def round_cycle(self, d_e_out, n_out, d_e_in, n_in, gamma):
        a, b = np.empty(4, dtype=np.uint64), np.empty(4, dtype=np.uint64)
        _chk(_lib.zg_rwc_round_cycle(self._h, _d(d_e_out), C.c_size_t(n_out), _d(d_e_in), C.c_size_t(n_in), _h(_c(gamma)), _h(a), _h(b)), "zg_rwc_round_cycle")
        return a, b

    def bind_cycle(self, r):
        _chk(_lib.zg_rwc_bind_cycle(self._h, _h(_c(r))), "zg_rwc_bind_cycle")

    def round_address(self, addr_round, challenges, gamma):
        ch = _c(np.asarray(challenges, dtype=np.uint64).reshape(-1, 4))
        assert ch.shape[0] == addr_round
        a, b = np.empty(4, dtype=np.uint64), np.empty(4, dtype=np.uint64)
        _chk(_lib.zg_rwc_round_address(self._h, C.c_size_t(addr_round), _h(ch) if addr_round else None, _h(_c(gamma)), _h(a), _h(b)), "zg_rwc_round_address")
        return a, b

    def bind_address(self, addr_round, r):
        _chk(_lib.zg_rwc_bind_address(self._h, C.c_size_t(addr_round), _h(_c(r))), "zg_rwc_bind_address")

    def opening(self, r_address, r_cycle):
        out = np.empty((3, 4), dtype=np.uint64)
        ra, rc = _c(np.asarray(r_address, dtype=np.uint64).reshape(-1, 4)), _c(np.asarray(r_cycle, dtype=np.uint64).reshape(-1, 4))
        _chk(_lib.zg_rwc_opening(self._h, _h(ra) if ra.size else None, _h(rc) if rc.size else None, _h(out)), "zg_rwc_opening")
        return out[0], out[1], out[2]

    def cycle_scalars(self):
        a, b = np.empty(4, dtype=np.uint64), np.empty(4, dtype=np.uint64)
        _chk(_lib.zg_rwc_cycle_scalars(self._h, _h(a), _h(b)), "zg_rwc_cycle_scalars")
        return a, b

    def read_entries(self, coefficients=True):
        """-> (cycle, address, ra_coeff, val_coeff, prev_val, next_val) arrays of the current list"""
        n = self.entries()
        cyc, adr = np.empty(n, dtype=np.uint32), np.empty(n, dtype=np.uint32)
        ra, val = np.empty((n, 4), dtype=np.uint64), np.empty((n, 4), dtype=np.uint64)
        prev, nxt = np.empty(n, dtype=np.uint64), np.empty(n, dtype=np.uint64)
        _chk(_lib.zg_rwc_read_entries(self._h, _hb(cyc), _hb(adr), _h(ra) if coefficients else None, _h(val) if coefficients else None, _h(prev), _h(nxt)),
             "zg_rwc_read_entries")
        return cyc, adr, ra, val, prev, nxt

    def close(self):
        if self._h:
            _chk(_lib.zg_rwc_close(self._h), "zg_rwc_close")
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedSumcheckSession:
    """Sumcheck(F).Prover over a table sharded across the devices bound by init_devices (zg_sumcheck_*_sharded)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def open(cls, evals, layout=SC_HIGH_HALF):
        e = _c(evals)
        h = C.c_void_p()
        _chk(_lib.zg_sumcheck_open_sharded(_h(e), C.c_size_t(e.size // 4), C.c_int(layout), C.byref(h)), "zg_sumcheck_open_sharded")
        return cls(h)

    def shards(self):
        return int(_lib.zg_sumcheck_shards(self._h))

    def round_sums(self):
        g0 = np.empty(4, dtype=np.uint64)
        g1 = np.empty(4, dtype=np.uint64)
        _chk(_lib.zg_sumcheck_round_sums_sharded(self._h, _h(g0), _h(g1)), "zg_sumcheck_round_sums_sharded")
        return g0, g1

    def bind(self, r):
        _chk(_lib.zg_sumcheck_bind_sharded(self._h, _h(_c(r))), "zg_sumcheck_bind_sharded")

    def __len__(self):
        return int(_lib.zg_sumcheck_len_sharded(self._h))

    def final(self):
        out = np.empty(4, dtype=np.uint64)
        _chk(_lib.zg_sumcheck_final_sharded(self._h, _h(out)), "zg_sumcheck_final_sharded")
        return out

    def close(self):
        if self._h:
            _chk(_lib.zg_sumcheck_close_sharded(self._h), "zg_sumcheck_close_sharded")
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
