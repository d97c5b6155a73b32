#!/usr/bin/env python3
"""Regenerates every binding that is derived from include/zolt_gpu.h (+ zolt_gpu_internal.h), so that no mirror of the C ABI is kept by hand:

  zig/gpu/ffi.zig      the Zig `extern fn` declarations (tools/gen_zig_ffi.py does the work)
  zolt_amd/_abi.py     the Python side: SYMBOLS / INTERNAL_SYMBOLS, the header's constants, and ctypes restype / argtypes of every entry
                       point (zolt_amd/lib.py applies them at import: an argument count or scalar width that drifts from the header is a
                       TypeError at the call, not a silently truncated size_t)

The C++ host mirrors (zolt_amd/host/*.hpp) include the header itself: their signatures are checked by the compiler.

    python tools/gen_bindings.py            # rewrite both files
    python tools/gen_bindings.py --check    # exit 1 if either is stale   (tests/test_abi_and_host.py runs this)
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_zig_ffi  # noqa: E402

HDR = os.path.join(ROOT, "include", "zolt_gpu.h")
HDR_INT = os.path.join(ROOT, "include", "zolt_gpu_internal.h")
OUT_PY = os.path.join(ROOT, "zolt_amd", "_abi.py")

C_SCALARS = {"int": "c_int", "unsigned": "c_uint", "size_t": "c_size_t", "uint64_t": "c_uint64", "uint32_t": "c_uint32", "uint8_t": "c_uint8",
             "double": "c_double"}
RET = {"int": "c_int", "void": "None", "size_t": "c_size_t", "uint32_t": "c_uint32", "const char *": "c_char_p"}


def c_protos(text):
    """[(name, restype, [(param name, ctypes type)])] for every ZG_API declaration of a header"""
    out = []
    for m in re.finditer(r"ZG_API\s+([^;(]*?)\b(zg_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        ret, name = " ".join(m.group(1).split()), m.group(2)
        params = []
        for a in gen_zig_ffi.split_params(m.group(3)):
            am = re.match(r"^(.*?)(\w+)\s*((?:\[\w*\])?)$", a)
            ctype, pname, arr = am.group(1).strip(), am.group(2), am.group(3)
            base = ctype.replace("const", "").replace("*", "").strip()
            if arr or "*" in ctype or base.endswith("_t") and base.startswith("zg_"):
                params.append((pname, "c_void_p"))  # pointers, fixed-size arrays and opaque handles: an address
            else:
                params.append((pname, C_SCALARS[base]))
        out.append((name, RET[ret], params))
    return out


def generate_py():
    pub, internal = open(HDR).read(), open(HDR_INT).read()
    consts = re.findall(r"#define (ZG_[A-Z0-9_]+) (\d+)u?\b", pub + internal)
    lines = ['"""GENERATED from include/zolt_gpu.h and include/zolt_gpu_internal.h by tools/gen_bindings.py — do not edit.',
             "tests/test_abi_and_host.py re-runs the generator and compares. zolt_amd/lib.py applies these signatures to the loaded library.\"\"\"",
             "from ctypes import c_char_p, c_double, c_int, c_size_t, c_uint, c_uint8, c_uint32, c_uint64, c_void_p  # noqa: F401", ""]
    seen = set()
    for name, val in consts:
        if name in seen or name == "ZG_API":
            continue
        seen.add(name)
        lines.append(f"{name} = {val}")
    lines.append("")
    for var, text in (("PROTOS", pub), ("INTERNAL_PROTOS", internal)):
        lines.append(f"{var} = {{")
        for name, ret, params in c_protos(text):
            ps = ", ".join(t for _, t in params)
            lines.append(f'    "{name}": ({ret}, [{ps}]),  # ' + ", ".join(p for p, _ in params))
        lines.append("}")
        lines.append("")
    lines += ["SYMBOLS = list(PROTOS)", "INTERNAL_SYMBOLS = list(INTERNAL_PROTOS)", "", "",
              "def apply(lib):",
              '    """restype / argtypes of every declared entry point; a symbol the header declares and the library lacks raises AttributeError"""',
              "    for table in (PROTOS, INTERNAL_PROTOS):",
              "        for name, (ret, args) in table.items():",
              "            fn = getattr(lib, name)",
              "            fn.restype = ret",
              "            fn.argtypes = args", ""]
    return "\n".join(lines)


def main():
    check = "--check" in sys.argv
    rc = 0
    new_py = generate_py()
    cur_py = open(OUT_PY).read() if os.path.exists(OUT_PY) else ""
    if check:
        if cur_py != new_py:
            print("zolt_amd/_abi.py is stale: run python tools/gen_bindings.py")
            rc = 1
    elif cur_py != new_py:
        with open(OUT_PY, "w") as f:
            f.write(new_py)
        print("wrote", OUT_PY)
    sys.argv = [a for a in sys.argv if a != "--check"] + (["--check"] if check else [])
    return max(rc, gen_zig_ffi.main())


if __name__ == "__main__":
    sys.exit(main())
