"""CPU-only checks: the C-ABI library loads and exports every symbol include/zolt_gpu.h declares,
fails loudly without a GPU (no CPU fallback), and the host-side logic (sharding, host mirrors)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "zolt_gpu.h")).read()
    return sorted(set(re.findall(r"ZG_API[^;(]*?\b(zg_\w+)\s*\(", hdr)))


def test_header_symbols_all_exported():
    from zolt_amd import lib
    declared = _declared_symbols()
    assert len(declared) >= 36
    nm = subprocess.check_output(["nm", "-D", "--defined-only", lib.LIB_PATH], text=True)
    exported = set(re.findall(r" T (zg_\w+)", nm))
    missing = [s for s in declared if s not in exported]
    assert not missing, missing
    assert sorted(lib.SYMBOLS) == declared  # the Python binding covers the whole header
    # nothing else leaks out of the library
    assert all(s.startswith("zg_") for s in re.findall(r" T (\w+)", nm) if not s.startswith("_"))


def _split_params(arglist):
    """top-level comma split of a C / Zig parameter list (no nested parentheses in either header)"""
    arglist = re.sub(r"/\*.*?\*/", "", arglist, flags=re.S).strip()
    if arglist in ("", "void"):
        return []
    return [a.strip() for a in arglist.split(",")]


def test_zig_extern_declarations_match_the_header():
    """zig/gpu/ffi.zig cannot be compiled here (no Zig toolchain), so at least its extern list is held against include/zolt_gpu.h:
    every declared function exists in the header with the same number of parameters, pointer parameters are pointers on both
    sides, and the constants it copies (error codes, sumcheck layouts) have the header's values."""
    hdr = open(os.path.join(ROOT, "include", "zolt_gpu.h")).read()
    zig = open(os.path.join(ROOT, "zig", "gpu", "ffi.zig")).read()
    protos = {m.group(1): _split_params(m.group(2)) for m in re.finditer(r"ZG_API[^;(]*?\b(zg_\w+)\s*\(([^;]*?)\)\s*;", hdr, flags=re.S)}
    externs = re.findall(r"pub extern fn (zg_\w+)\((.*?)\) [\w\[\]:*?. ]+;", zig)
    assert len(externs) >= 25
    for name, args in externs:
        assert name in protos, name
        zargs, cargs = _split_params(args), protos[name]
        assert len(zargs) == len(cargs), (name, zargs, cargs)
        for za, ca in zip(zargs, cargs):
            c_ptr = "*" in ca or "[" in ca or "zg_bases_t" in ca or "zg_sc_t" in ca
            z_ptr = "*" in za or "Bases" in za or "Session" in za
            assert c_ptr == z_ptr, (name, za, ca)
    for zname, cname in (("ERR_VERIFY", "ZG_ERR_VERIFY"), ("SC_HIGH_HALF", "ZG_SC_HIGH_HALF"), ("SC_LOW_PAIR", "ZG_SC_LOW_PAIR"), ("OK", "ZG_OK")):
        zv = int(re.search(r"pub const %s: c_int = (\d+);" % zname, zig).group(1))
        cv = int(re.search(r"#define %s (\d+)" % cname, hdr).group(1))
        assert zv == cv, (zname, zv, cv)


def test_header_compiles_as_c_and_cpp(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "zolt_gpu.h"\nint main(void){ zg_msm_config c = {0,0}; (void)c; return ZG_OK; }\n')
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", inc, "-c", str(src), "-o", str(tmp_path / "t.o")])
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-x", "c++", "-I", inc, "-c", str(src), "-o", str(tmp_path / "t2.o")])


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU present: the no-device path cannot be observed")
def test_no_gpu_fails_loudly_never_falls_back():
    from zolt_amd import lib
    assert lib.device_count() == 0
    with pytest.raises(lib.ZgError) as e:
        lib.init()
    assert e.value.code == lib.ERR_NO_DEVICE
    with pytest.raises(lib.ZgError) as e:
        lib.field_op(lib.FR, lib.OP_MUL, np.zeros((1, 4), dtype=np.uint64), np.zeros((1, 4), dtype=np.uint64))
    assert e.value.code == lib.ERR_NO_DEVICE
    with pytest.raises(lib.ZgError):
        lib.Bases.upload(np.zeros((1, 8), dtype=np.uint64))
    with pytest.raises(lib.ZgError):
        lib.fr_eq_table(np.zeros((2, 4), dtype=np.uint64))
    with pytest.raises(lib.ZgError):
        lib.SumcheckSession.open(np.zeros((2, 4), dtype=np.uint64))
    with pytest.raises(lib.ZgError) as e:
        lib.run_sumcheck(np.zeros((4, 4), dtype=np.uint64))
    assert e.value.code == lib.ERR_NO_DEVICE
    with pytest.raises(lib.ZgError):
        lib.fr_spartan_combine(*[np.zeros((2, 4), dtype=np.uint64)] * 4)
    with pytest.raises(lib.ZgError):
        lib.g1_scalar_mul_batch(np.zeros((1, 8), dtype=np.uint64), np.zeros(1, dtype=np.uint8), np.zeros((1, 4), dtype=np.uint64))


def test_product_never_imports_oracle():
    """The product path must not reach the oracle (parity claims depend on it)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "zolt_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("no oracle", ""), os.path.join(dirpath, f)
    out = subprocess.check_output([sys.executable, "-c",
                                   "import sys; import zolt_amd.lib, zolt_amd.api; print([m for m in sys.modules if 'oracle' in m])"],
                                  cwd=ROOT, text=True)
    assert out.strip() == "[]"
    ldd = subprocess.check_output(["ldd", os.path.join(ROOT, "zolt_amd", "libzolt_gpu.so")], text=True)
    assert "oracle" not in ldd


def test_shard_bounds_match_parallel_msm_partition():
    """ParallelMSM: chunk = ceil(n/T), start = i*chunk, end = min(start+chunk, n) (src/msm/mod.zig:609,619-639)."""
    from zolt_amd import api
    assert api.shard_bounds(10, 4) == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert api.shard_bounds(8, 8) == [(i, i + 1) for i in range(8)]
    assert api.shard_bounds(3, 4) == [(0, 1), (1, 2), (2, 3), (3, 3)]  # trailing shard empty (threadWorkerEmpty, :667-670)
    assert api.shard_bounds(1 << 22, 8)[7] == (7 << 19, 1 << 22)
    for n in (0, 1, 5, 1000, 4097):
        for t in (1, 2, 3, 8):
            b = api.shard_bounds(n, t)
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(t - 1))


def test_host_scalar_helpers_and_toy_verifier():
    """fromU64 / toBytes conventions and the toy mixer (src/subprotocols/mod.zig:211-243) against the oracle."""
    from oracle import binding as ob
    from zolt_amd import api
    for v in (0, 1, 2, 0x12345678, (1 << 64) - 1):
        assert np.array_equal(api.fr_from_int(v), ob.f_from_u64(ob.FR, np.array([v], dtype=np.uint64))[0])
        assert api.fr_to_int(api.fr_from_int(v)) == v
    g = api.generator()
    assert ob.g1_is_on_curve(g)
    assert api.commitment_to_bytes(g, 0) == ob.commitment_to_bytes(g)
    assert api.commitment_to_bytes(g, 1) == bytes(64)
    claim = api.fr_from_int(36)
    coeffs = np.stack([api.fr_from_int(10), api.fr_from_int(16)])
    ver = api.Sumcheck.Verifier(claim)
    ch = ver.verifyRound(coeffs)
    assert np.array_equal(ch, ob.sumcheck_derive_challenge(0, claim, coeffs))
    bad = api.Sumcheck.Verifier(api.fr_from_int(35))
    with pytest.raises(api.SumcheckVerificationFailed):
        bad.verifyRound(coeffs)


def test_parse_zolt_proof_commitments(golden_dir):
    """the captured reference proof (logs/zolt_proof_regular.bin): container layout of src/zkvm/serialization.zig:283-306"""
    from zolt_amd import api
    data = open(os.path.join(golden_dir, "zolt_proof_regular.bin"), "rb").read()
    c = api.parse_zolt_proof_commitments(data)
    assert len(c) == 11
    assert c["bytecode.commitment"].hex().startswith("048184e5b9afa827")  # reproduced from fibonacci.elf
    assert c["memory.commitment"] == bytes(64)                              # identity
    assert c["register.commitment"].hex().startswith("1a5d881d")
    assert c["bytecode.read_ts_commitment"] == bytes(64)
    with pytest.raises(ValueError):
        api.parse_zolt_proof_commitments(b"JOLT" + data[4:])
