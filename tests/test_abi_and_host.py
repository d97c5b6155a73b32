"""CPU-only checks: the C-ABI library loads and exports every symbol include/zolt_gpu.h declares,
fails loudly without a GPU (no CPU fallback), and the host-side logic (sharding, host mirrors)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header="zolt_gpu.h"):
    hdr = open(os.path.join(ROOT, "include", header)).read()
    return sorted(set(re.findall(r"ZG_API[^;(]*?\b(zg_\w+)\s*\(", hdr)))


def test_header_symbols_all_exported():
    from zolt_amd import lib
    declared = _declared_symbols()
    internal = _declared_symbols("zolt_gpu_internal.h")
    assert len(declared) >= 70 and internal == ["zg_last_setup_times", "zg_pool_debug_selftest", "zg_pool_debug_stats", "zg_profile_begin", "zg_profile_end",
                                              "zg_sharded_comm_sets_created"]
    nm = subprocess.check_output(["nm", "-D", "--defined-only", lib.LIB_PATH], text=True)
    exported = set(re.findall(r" T (zg_\w+)", nm))
    assert exported == set(declared) | set(internal)  # the library exports exactly the two headers, nothing more, nothing less
    assert sorted(lib.SYMBOLS) == declared  # the Python binding covers the whole public header
    assert sorted(lib.INTERNAL_SYMBOLS) == internal
    # nothing else leaks out of the library
    assert all(s.startswith("zg_") for s in re.findall(r" T (\w+)", nm) if not s.startswith("_"))
    # the public header carries no test / bench scaffolding (self-test op codes and the profiler live in the internal one)
    pub = open(os.path.join(ROOT, "include", "zolt_gpu.h")).read()
    for name in ("zg_profile", "ZG_OP_MUL29", "ZG_OP_X3_29", "ZG_OP_INV_XGCD", "ZG_OP_INV_SAFEGCD", "ZG_PROF_"):
        assert name not in pub, name


def test_bindings_are_generated_from_the_header():
    """zolt_amd/_abi.py (ctypes signatures of every entry point) and zig/gpu/ffi.zig are both written by tools/gen_bindings.py from the
    header; neither is maintained by hand. The loaded library reports the ABI version of the header the binding was generated from."""
    assert subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_bindings.py"), "--check"]).returncode == 0
    from zolt_amd import _abi, lib
    hdr = open(os.path.join(ROOT, "include", "zolt_gpu.h")).read()
    protos = {m.group(1): _split_params(m.group(2)) for m in re.finditer(r"ZG_API[^;(]*?\b(zg_\w+)\s*\(([^;]*?)\)\s*;", hdr, flags=re.S)}
    assert sorted(_abi.PROTOS) == sorted(protos)
    for name, (ret, args) in _abi.PROTOS.items():
        assert len(args) == len(protos[name]), name
        fn = getattr(lib._lib, name)
        assert fn.argtypes == args and fn.restype == ret, name
    assert lib.abi_version() == (_abi.ZG_ABI_MAJOR, _abi.ZG_ABI_MINOR)
    assert lib.abi_features() & _abi.ZG_FEATURE_PROTOCOL_SESSIONS
    # the optional section really is optional: the header parses without it and loses exactly the two protocol-specific families
    core = subprocess.run(["gcc", "-E", "-DZG_NO_PROTOCOL_SESSIONS", os.path.join(ROOT, "include", "zolt_gpu.h")], capture_output=True, text=True, check=True).stdout
    core_syms = set(re.findall(r"\b(zg_\w+)\s*\(", core))
    dropped = set(protos) - core_syms
    assert dropped and all(n.startswith(("zg_rrw_", "zg_rwc_")) for n in dropped)
    assert not any(n.startswith(("zg_rrw_", "zg_rwc_")) for n in core_syms)


def test_zig_ffi_declares_every_type_it_names():
    """A static stand-in for the compiler this image lacks: every type name an extern of zig/gpu/ffi.zig uses is declared in the file
    (round 5 found `RegistersSession` / `RamRwSession` used by 24 externs and declared nowhere)."""
    zig = open(os.path.join(ROOT, "zig", "gpu", "ffi.zig")).read()
    declared = set(re.findall(r"^pub const (\w+) =", zig, flags=re.M))
    builtin = {"c_int", "c_uint", "usize", "u64", "u32", "u8", "f64", "anyopaque", "void", "const"}
    for name, args, ret in re.findall(r"pub extern fn (zg_\w+)\((.*?)\) ([\w\[\]:*?. ]+);", zig):
        for ty in re.findall(r":\s*([^,]+)", args) + [ret]:
            for ident in re.findall(r"[A-Za-z_]\w*", ty):
                assert ident in builtin or ident in declared, (name, ident)


def _split_params(arglist):
    """top-level comma split of a C / Zig parameter list (no nested parentheses in either header)"""
    arglist = re.sub(r"/\*.*?\*/", "", arglist, flags=re.S).strip()
    if arglist in ("", "void"):
        return []
    return [a.strip() for a in arglist.split(",")]


def test_zig_ffi_is_generated_from_the_header():
    """zig/gpu/ffi.zig cannot be compiled here (no Zig toolchain): it is GENERATED from include/zolt_gpu.h by tools/gen_zig_ffi.py,
    and this test re-runs the generator, so the file can neither go stale nor miss an export. Every public export must have an
    extern with the same number of parameters, pointers where the header has pointers, and the header's constants."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_zig_ffi
    zig = open(os.path.join(ROOT, "zig", "gpu", "ffi.zig")).read()
    assert zig == gen_zig_ffi.generate(), "zig/gpu/ffi.zig is stale: run python tools/gen_zig_ffi.py"
    hdr = open(os.path.join(ROOT, "include", "zolt_gpu.h")).read()
    protos = {m.group(1): _split_params(m.group(2)) for m in re.finditer(r"ZG_API[^;(]*?\b(zg_\w+)\s*\(([^;]*?)\)\s*;", hdr, flags=re.S)}
    externs = dict(re.findall(r"pub extern fn (zg_\w+)\((.*?)\) [\w\[\]:*?. ]+;", zig))
    assert sorted(externs) == sorted(protos) == _declared_symbols()  # EVERY export, no extras
    for name, args in externs.items():
        zargs, cargs = _split_params(args), protos[name]
        assert len(zargs) == len(cargs), (name, zargs, cargs)
        for za, ca in zip(zargs, cargs):
            c_ptr = "*" in ca or "[" in ca or any(h in ca for h in ("zg_bases_t", "zg_sc_t", "zg_sbases_t", "zg_ssc_t", "zg_psc_t", "zg_rrw_t", "zg_rwc_t"))
            z_ptr = "*" in za or any(h in za for h in ("Bases", "Session"))
            assert c_ptr == z_ptr, (name, za, ca)
    for cname, cval in re.findall(r"#define (ZG_(?:OK|ERR_\w+|SC_\w+|FIELD_\w+)) (\d+)", hdr):
        assert re.search(r"pub const %s: c_int = %s;" % (cname[3:], cval), zig), cname
    assert "expected_uses" in zig  # zg_msm_config's third field


def test_zig_backend_calls_match_the_generated_externs():
    """No Zig toolchain here, so zig/gpu/backend.zig is at least held statically to zig/gpu/ffi.zig (itself generated from the header):
    every `ffi.zg_*(...)` call names an existing extern and passes exactly as many arguments as the extern declares, every
    `ffi.CONST` exists, and braces / parentheses / brackets balance once comments and string literals are removed."""
    src = open(os.path.join(ROOT, "zig", "gpu", "backend.zig")).read()
    ffi = open(os.path.join(ROOT, "zig", "gpu", "ffi.zig")).read()
    ext = {m.group(1): m.group(2) for m in re.finditer(r"pub extern fn (zg_\w+)\((.*?)\) [\w\[\]:*?. ]+;", ffi)}
    consts = set(re.findall(r"pub const (\w+)", ffi))

    def split_top(text):
        out, depth, cur = [], 0, ""
        for ch in text:
            depth += ch in "([{"
            depth -= ch in ")]}"
            if ch == "," and depth == 0:
                out.append(cur)
                cur = ""
            else:
                cur += ch
        return out + ([cur] if cur.strip() else [])

    calls = 0
    for m in re.finditer(r"ffi\.(zg_\w+)\(", src):
        name, i = m.group(1), m.end()
        depth, j = 1, i
        while depth:
            depth += (src[j] == "(") - (src[j] == ")")
            j += 1
        assert name in ext, name
        want = len(ext[name].split(",")) if ext[name].strip() else 0
        assert len(split_top(src[i:j - 1])) == want, (name, src[i:j - 1][:100])
        calls += 1
    assert calls >= 50
    for c in set(re.findall(r"ffi\.([A-Z]\w+)", src)):
        assert c in consts, c
    code = re.sub(r'"(?:\\.|[^"\\])*"', '""', re.sub(r"//[^\n]*", "", src))
    for o, c in ("{}", "()", "[]"):
        assert code.count(o) == code.count(c), o


def test_zig_sources_use_no_undeclared_identifier(tmp_path):
    """One notch above counting brackets (no Zig toolchain in this image): tools/zig_lint.py holds every identifier USED in
    zig/gpu/backend.zig and the generated ffi.zig to a declaration (file / container level const, var, fn; parameter, local or capture of
    the enclosing function; keyword, primitive, builtin) and every direct call of one of the shim's free functions to its parameter count.
    The lint has to earn its keep: five one-token mutations of the shim — a misspelt
    local, a wrong parameter name, a missing helper, a wrong capture, a dropped file-level name — must each be reported."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import zig_lint
    for name in ("backend.zig", "ffi.zig"):
        found, n = zig_lint.lint(os.path.join(ROOT, "zig", "gpu", name))
        assert not found and n > 2000, found[:5]
    src = open(os.path.join(ROOT, "zig", "gpu", "backend.zig")).read()
    mutations = [("&f1.limbs, &FR_ONE) and", "&f_1.limbs, &FR_ONE) and", "f_1"),
                 ("available = ffi.zg_init(dev) == ffi.OK;", "available = ffi.zg_init(device) == ffi.OK;", "device"),
                 ("init_once.call();", "initOnce.call();", "initOnce"),
                 ("if (v.len > 0 and v[0] == '0') return;", "if (w.len > 0 and v[0] == '0') return;", "w"),
                 ("return @intCast(n_devices);", "return @intCast(n_device);", "n_device")]
    for k, (a, b, bad) in enumerate(mutations):
        assert src.count(a) == 1, a
        path = tmp_path / f"mutant{k}.zig"
        path.write_text(src.replace(a, b))
        found, _ = zig_lint.lint(str(path))
        assert len(found) == 1 and f"'{bad}' is used but never declared" in found[0], (k, found)
    # ... and the arity of direct calls to the shim's own free functions
    path = tmp_path / "mutant_arity.zig"
    path.write_text(src.replace("limbsOf(F, scalars)", "limbsOf(scalars)", 1))
    found, _ = zig_lint.lint(str(path))
    assert found == [f for f in found if "'limbsOf' takes 2 arguments, called with 1" in f] and len(found) == 1, found


def test_zig_backend_rules_out_the_stale_table_and_wrong_field_hazards():
    """zig/gpu/backend.zig (compile-unverified) is at least held to the three rules INTEGRATION.md states: a comptime type gate in
    front of every MSM path (the reference instantiates MSM(Fr, Fr), src/msm/mod.zig:853-873,911-936), no (ptr, len)-keyed
    handle cache (stale tables / use-after-evict), and one-shot uploads for ad-hoc slices that are freed before returning.
    Every ffi symbol it uses must exist in the generated ffi.zig."""
    be = open(os.path.join(ROOT, "zig", "gpu", "backend.zig")).read()
    code = "\n".join(l for l in be.splitlines() if not l.lstrip().startswith("//"))
    zig = open(os.path.join(ROOT, "zig", "gpu", "ffi.zig")).read()
    externs = set(re.findall(r"pub extern fn (zg_\w+)\(", zig))
    used = set(re.findall(r"ffi\.(zg_\w+)\(", code))
    assert used and used <= externs, used - externs
    for wrapper in ("zg_g1_fixed_base_mul_batch", "zg_hyperkzg_open", "zg_hyperkzg_batch_open", "zg_msm_g1_sharded", "zg_msm_g1_batch_sharded",
                    "zg_sumcheck_open_sharded", "zg_init_devices"):
        assert wrapper in used, wrapper  # setup / open / batchOpen / the multi-GPU entry points all have a wrapper now
    # rule 1: both one-shot MSM entry points start with the comptime gate
    for fn in ("msmComputeOneShot", "parallelMsmOneShot"):
        body = code[code.index("pub fn " + fn):]
        body = body[:body.index("\npub fn ", 10)] if "\npub fn " in body[10:] else body
        assert "if (comptime !isBn254Pair(F, G)) return null;" in body, fn
    assert "0xac96341c4ffffffb" in code and "0xd35d438dc58f0d9d" in code  # Fr / Fp Montgomery R: what the gate compares
    # rule 2: no address-keyed cache anywhere
    assert "@intFromPtr" not in code and "CacheEntry" not in code and "cache" not in code.lower()
    # rule 3: the one-shot handle is freed in the function that created it, and it skips the precompute table
    body = code[code.index("pub fn msmComputeOneShot"):code.index("pub fn parallelMsmOneShot")]
    assert "defer _ = ffi.zg_g1_bases_free(h);" in body and ".expected_uses = 1" in body


def test_zig_backend_size_gates_are_the_measured_crossovers():
    """Every host-pointer forwarding wrapper of zig/gpu/backend.zig refuses sizes below the CPU / GPU crossover measured by
    tools/crossover.py (profiles/r4_crossover.json), and its constant IS the measured gate: no wrapper forwards a size at which the
    GPU call was slower than the CPU body."""
    import json
    be = open(os.path.join(ROOT, "zig", "gpu", "backend.zig")).read()
    code = "\n".join(l for l in be.splitlines() if not l.lstrip().startswith("//"))
    cross = json.load(open(os.path.join(ROOT, "profiles", "r4_crossover.json")))  # re-measured in round 4 (new table-less MSM plan, new fold kernels)
    consts = {m.group(1): int(m.group(2)) for m in re.finditer(r"pub const (\w+_min_\w+): usize = (\d+);", code)}
    want = {"srs_commit_min_points": "srs_commit", "one_shot_min_points": "one_shot_msm", "eq_table_min_entries": "eq_table",
            "bind_low_min_entries": "bind_low", "bind_high_min_entries": "bind_high", "run_sumcheck_min_entries": "run_sumcheck",
            "open_min_entries": "hyperkzg_open", "lt_table_min_entries": "lt_table", "weighted_colsum_min_entries": "weighted_colsum"}
    assert set(want) <= set(consts)
    for const, key in want.items():
        assert consts[const] == cross["gates"][key], (const, consts[const], cross["gates"][key])
        pts = {p["n"]: p for p in cross["points"][key]}
        for n, p in pts.items():  # at and above the gate the GPU was faster wherever the CPU was timed
            if n >= consts[const] and p["cpu_us"] is not None:
                assert p["gpu_us"] < p["cpu_us"], (key, n)

    def body(name):
        i = code.index("pub fn " + name + "(")
        j = code.find("\n    pub fn ", i + 10) if code[i - 4:i] == "    " else code.find("\npub fn ", i + 10)
        return code[i:j if j > 0 else len(code)]
    gated = {"commit": "srs_commit_min_points", "batchCommit": "srs_commit_min_points", "open": "open_min_entries", "batchOpen": "open_min_entries", "msmComputeOneShot": "one_shot_min_points",
             "parallelMsmOneShot": "one_shot_min_points", "eqTable": "eq_table_min_entries", "bindLow": "bind_low_min_entries",
             "bindHigh": "bind_high_min_entries", "runSumcheck": "run_sumcheck_min_entries", "ltTable": "lt_table_min_entries",
             "weightedColsum": "weighted_colsum_min_entries"}
    for fn, const in gated.items():
        b = body(fn)
        assert const in b, (fn, const)
        assert b.index(const) < b.index("ffi.zg_"), fn  # the gate comes before the first library call


def test_header_compiles_as_c_and_cpp(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "zolt_gpu.h"\nint main(void){ zg_msm_config c = {0,0,0}; (void)c; return ZG_OK; }\n')
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


def test_shard_bounds_is_parallel_msm_partition():
    """zg_shard_bounds (host arithmetic, no device) = ParallelMSM's chunks (src/msm/mod.zig:609,619-621): chunk_size =
    ceil(n / T), start = i * chunk_size, end = min(start + chunk_size, n), empty chunk when start >= n. The chunks tile [0, n)
    in shard order — the order in which the gathered partials are combined — and agree with api.shard_bounds (the
    one-process-per-GPU path)."""
    from zolt_amd import api, lib
    for n in (0, 1, 7, 8, 1000, 1024, (1 << 20) + 1, 1 << 22):
        for shards in (1, 2, 3, 4, 7, 8, 16):
            chunk = (n + shards - 1) // shards
            pos = 0
            for i in range(shards):
                st, ln = lib.shard_bounds(n, shards, i)
                ref_start = i * chunk
                want = (min(ref_start, n), 0) if ref_start >= n else (ref_start, min(ref_start + chunk, n) - ref_start)
                assert (st, ln) == want, (n, shards, i)
                assert st == pos or ln == 0
                pos += ln
            assert pos == n
            assert [(a, b - a) for a, b in api.shard_bounds(n, shards)] == [lib.shard_bounds(n, shards, i) for i in range(shards)]
    with pytest.raises(lib.ZgError):
        lib.shard_bounds(10, 0, 0)
    with pytest.raises(lib.ZgError):
        lib.shard_bounds(10, 4, 4)


def test_eq_mle_host_mirror_matches_oracle_and_bigint():
    """EqPolynomial.mle / evaluate (src/poly/mod.zig:214-227,311-321) is host scalar code in the reference and in the mirror
    (zolt_amd/api/): checked here, without a GPU, against the C oracle and the big-int model."""
    from oracle import binding as ob
    from oracle import pymodel as pm
    from tests import util as U
    from zolt_amd import api
    for v in (0, 1, 3, 13):
        r = ob.f_to_mont(ob.FR, U.random_raw256(70 + v, v))
        x = ob.f_to_mont(ob.FR, U.random_raw256(80 + v, v))
        want = ob.fr_eq_mle(r, x)
        assert np.array_equal(api.EqPolynomial.mle(r, x), want)
        assert np.array_equal(api.EqPolynomial(r).evaluate(x), want)
        acc = 1
        for a, b in zip([U.fr_to_int(t) for t in r], [U.fr_to_int(t) for t in x]):
            acc = acc * ((a * b + (1 - a) * (1 - b)) % pm.R_MOD) % pm.R_MOD
        assert U.fr_to_int(want) == acc
    r = ob.f_to_mont(ob.FR, U.random_raw256(90, 4))
    tab = ob.fr_eq_table(r)
    for idx in range(16):  # at a boolean point mle is the table entry, index MSB <-> r[0]
        assert np.array_equal(api.EqPolynomial.mle(r, U.fr([(idx >> (3 - j)) & 1 for j in range(4)])), tab[idx])


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU present: the no-device path cannot be observed")
def test_bench_self_launch_without_gpu_fails_fast_and_loudly(tmp_path):
    """`python bench.py --gpus N` without a launcher starts its N ranks itself (before anything touches a GPU); without a device
    every rank must refuse (no CPU fallback) and the launcher must hand the failure back — a non-zero exit code and ONE JSON line that
    names the failing rank and carries its stderr (the ranks' own output goes to per-rank files) — not hang, and print no bench line."""
    import json
    env = dict(os.environ, ZOLT_BENCH_LOG_DIR=str(tmp_path))
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode != 0
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    err = json.loads(lines[0])
    assert "metric" not in err and "bench.py needs a GPU" in err["stderr_tail"] and err["rank"] in (0, 1)
    assert "bench.py needs a GPU" in open(os.path.join(tmp_path, f"rank{err['rank']}.stderr")).read()


def test_design_kernel_table_is_the_generated_one():
    """DESIGN.md section 4.0 carries the table tools/kernel_table.py generates from the committed rocprofv3 summaries: regenerated here and
    compared, so the figures in the document are the ones under profiles/ (and nobody edits the table by hand)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "kernel_table.py"), "r6"], capture_output=True, text=True, check=True).stdout.strip()
    assert out == open(os.path.join(root, "profiles", "r6_kernel_table.md")).read().strip()
    design = open(os.path.join(root, "DESIGN.md")).read()
    assert out in design


def test_bench_roofline_counters_come_from_the_committed_pmc_file():
    """Round-4 review: bench.py's roofline.traffic quoted numbers the file it cited no longer held. Now bench.py holds no counter values at
    all — it reads profiles/r6_pmc.json, which tools/collect_profiles.sh writes on the GPU box from rocprofv3 --pmc passes — and this test
    holds the two together: what bench.py would report for 2^20 and 2^22 points is exactly what the committed file contains."""
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    src = open(os.path.join(root, "bench.py")).read()
    assert "MEASURED_TRAFFIC" not in src and "PMC_INSTS_VALU_PER_LAUNCH" not in src
    assert not re.search(r"\d,\d{3},\d{3} KB", src), "a hand-typed counter value in bench.py"
    raw = json.load(open(bench.PMC_FILE))
    pmc = bench.load_pmc()
    assert set(pmc) == {"2^20", "2^22"}
    for size, logn, launches in (("2^20", 20, 1), ("2^22", 22, 4)):
        c = raw["sizes"][size]["counters"]
        assert raw["kernel"] == "msm_accumulate_chunk_kernel" and c["FETCH_SIZE"]["dispatches"] >= 40
        adds = (1 << logn) * 15 * (1.0 - 2.0 ** -17) / launches
        want = c["FETCH_SIZE"]["avg"] * 1024.0 + 2.0 * adds + c["WRITE_SIZE"]["avg"] * 1024.0
        assert bench.measured_traffic(pmc, logn, adds) == want
        assert pmc[size]["SQ_INSTS_VALU"] == c["SQ_INSTS_VALU"]["avg"]
        assert 2300 < c["SQ_INSTS_VALU"]["avg"] / (adds / 64.0) < 2450  # wave instructions per wave-wide mixed addition
        # the committed text summary of the same run holds the same averages (one decimal)
        summary = open(os.path.join(root, "profiles", "r6_rocprofv3_summary.txt" if logn == 20 else "r6_rocprofv3_summary_2^22.txt")).read()
        block = summary[summary.index("void zg::msm_accumulate_chunk_kernel<false>\n"):]
        assert "FETCH_SIZE               avg %16.1f" % c["FETCH_SIZE"]["avg"] in block[:2500]
        assert "WRITE_SIZE               avg %16.1f" % c["WRITE_SIZE"]["avg"] in block[:2500]
    assert bench.measured_traffic(pmc, 18, 1.0) is None  # no counters for a size: the line says null, never a guess
    cal = raw["calibration"]
    assert 0.95 < cal["gather"]["reported_over_actual"] < 1.10 and 0.45 < cal["stream"]["reported_over_actual"] < 0.55
