"""Robustness of the boundary itself (argument sweeps over every export, an allocation the device cannot hold, shutdown / init cycles).
The C ABI "returns an error code, never throws" (MSM.compute has no error channel, src/msm/mod.zig:355-372: a shim must be
able to trust a return code): EVERY exported entry point is called with (a) all-zero arguments — NULL handles and pointers, sizes 0 — and
(b) NULL handles and pointers with every size / count argument = 16. Neither may crash the process; (b) must be refused (a size says there
is data, the pointer says there is none) unless the function has no pointer to refuse. Each sweep runs in a child process: a signal there
names the entry point that died. The list of entry points comes from the generated signature table (zolt_amd/_abi.py), so a new export is
swept without anybody remembering to add it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes, json, sys
sys.path.insert(0, %r)
from zolt_amd import _abi, lib
lib.init(0)
mode = %r
skip = {"zg_shutdown", "zg_init", "zg_init_devices", "zg_pool_debug_selftest", "zg_selftest_handoff"}  # lifecycle / self-tests with their own tests
out = {}
for name, (ret, args) in list(_abi.PROTOS.items()) + list(_abi.INTERNAL_PROTOS.items()):
    if name in skip:
        continue
    vals = []
    for a in args:
        if a is ctypes.c_void_p:
            vals.append(None)
        elif a is ctypes.c_double:
            vals.append(0.0)
        else:
            vals.append(16 if mode == "sized" and a in (ctypes.c_size_t, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint) else 0)
    print("CALL", name, flush=True)  # the last CALL line names the entry point if the process dies
    rc = getattr(lib._lib, name)(*vals)
    out[name] = rc if isinstance(rc, int) else (rc.decode()[:40] if isinstance(rc, bytes) else None)
print("RESULT " + json.dumps(out))
"""


def _sweep(mode):
    res = subprocess.run([sys.executable, "-c", CHILD % (ROOT, mode)], capture_output=True, text=True, timeout=600, cwd=ROOT)
    calls = [l.split()[1] for l in res.stdout.splitlines() if l.startswith("CALL ")]
    assert res.returncode == 0, f"the process died (rc {res.returncode}) inside {calls[-1] if calls else '?'}: {res.stderr[-800:]}"
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert len(out) >= 140
    return out


def test_every_entry_point_survives_all_zero_arguments():
    out = _sweep("zero")
    # nothing to do is either fine or an invalid argument: never a HIP error, never out of memory
    bad = {k: v for k, v in out.items() if isinstance(v, int) and k.startswith("zg_") and v in (2, 3) and not k.endswith(("_len", "_bytes", "_count", "_version", "_features"))}
    assert not bad, bad


def test_sizes_without_data_are_refused_not_dereferenced():
    from zolt_amd import _abi
    import ctypes
    out = _sweep("sized")
    for name, (ret, args) in _abi.PROTOS.items():
        if name not in out or ret is not ctypes.c_int:
            continue
        has_ptr = any(a is ctypes.c_void_p for a in args)
        has_size = any(a in (ctypes.c_size_t, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint) for a in args)
        if has_ptr and has_size:
            assert out[name] != 0, f"{name}: sizes of 16 with NULL pointers returned ZG_OK"


def test_an_allocation_the_device_cannot_hold_is_an_error_code_and_the_library_stays_usable():
    """400 GB on a 288 GB part: ZG_ERR_NOMEM (after the pool was trimmed and the driver asked once more), promptly, and the next call works"""
    import time

    import numpy as np
    from oracle import binding as ob
    from zolt_amd import lib
    lib.init(0)
    keep = [lib.DeviceBuffer(64 << 20) for _ in range(3)]  # something idle in the pool for the trim to return
    for b in keep:
        b.free()
    t0 = time.perf_counter()
    with pytest.raises(lib.ZgError) as err:
        lib.DeviceBuffer(400 << 30)
    assert err.value.code == lib.ERR_NOMEM and time.perf_counter() - t0 < 5.0, (err.value, time.perf_counter() - t0)
    gm = ob.g1_gen_multiples(2000)
    sc = ob.f_to_mont(ob.FR, np.random.default_rng(3).integers(0, 1 << 63, size=(2000, 4), dtype=np.uint64))
    h = lib.Bases.upload(gm)
    got, want = h.msm(sc), ob.msm_g1(gm, None, sc)
    h.free()
    assert got[1] == want[1] and np.array_equal(got[0], want[0])


LIFECYCLE = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
from oracle import binding as ob
from zolt_amd import api, lib
n = 3000
gm = ob.g1_gen_multiples(n)
sc = ob.f_to_mont(ob.FR, np.random.default_rng(4).integers(0, 1 << 63, size=(n, 4), dtype=np.uint64))
want = ob.msm_g1(gm, None, sc)
ev = sc[:1024]
wrun = ob.run_sumcheck(ev)
held = None
for cycle in range(4):
    lib.init(0)
    h = lib.Bases.upload(gm)
    got = h.msm(sc)
    assert got[1] == want[1] and np.array_equal(got[0], want[0]), cycle
    r = lib.run_sumcheck(ev)
    assert r["result"] and np.array_equal(r["final_eval"], wrun[3]), cycle
    s = lib.SumcheckSession.open(ev)
    g0, g1 = s.round_sums()
    w0, w1 = ob.fr_sum_halves(ev)
    assert np.array_equal(g0, w0) and np.array_equal(g1, w1), cycle
    if held is not None:  # a handle and a session that outlived a zg_shutdown: still usable, freed by their owners later
        hg = held[0].msm(sc)
        assert hg[1] == want[1] and np.array_equal(hg[0], want[0]), ("held handle", cycle)
        held[1].bind(ev[0])
        held[1].close()
        held[0].free()
        held = None
    if cycle %% 2 == 0:
        held = (h, s)  # keep them across the shutdown below
    else:
        s.close()
        h.free()
    lib.shutdown()
print("LIFECYCLE OK")
"""


def test_shutdown_and_init_again_with_handles_that_outlive_it():
    """zg_shutdown / zg_init cycles in one process: the pool's idle blocks, the streams and the pinned buffers go back, what a live handle or
    session still holds is forgotten (freed by its owner later), and the next cycle computes the same bytes."""
    res = subprocess.run([sys.executable, "-c", LIFECYCLE % ROOT], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0 and "LIFECYCLE OK" in res.stdout, (res.returncode, res.stdout[-300:], res.stderr[-1200:])
