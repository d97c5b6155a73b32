#!/usr/bin/env python3
"""Regenerates zig/gpu/ffi.zig from include/zolt_gpu.h: one `pub extern fn` per ZG_API declaration, the error codes and layout
constants, the handle types and zg_msm_config. tests/test_abi_and_host.py re-runs the generator and compares, so the Zig side
cannot drift from the header (this image has no Zig toolchain to compile it).

    python tools/gen_zig_ffi.py            # rewrite zig/gpu/ffi.zig
    python tools/gen_zig_ffi.py --check    # exit 1 if the file differs
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "zolt_gpu.h")
OUT = os.path.join(ROOT, "zig", "gpu", "ffi.zig")

HANDLES = {"zg_bases_t": "Bases", "zg_sc_t": "Session", "zg_sbases_t": "ShardedBases", "zg_ssc_t": "ShardedSession", "zg_psc_t": "ProductSession", "zg_rrw_t": "RegistersSession", "zg_rwc_t": "RamRwSession"}
SCALARS = {"int": "c_int", "unsigned": "c_uint", "size_t": "usize", "uint64_t": "u64", "uint32_t": "u32", "uint8_t": "u8", "double": "f64"}


def split_params(arglist):
    arglist = re.sub(r"/\*.*?\*/", "", arglist, flags=re.S).strip()
    if arglist in ("", "void"):
        return []
    return [" ".join(a.split()) for a in arglist.split(",")]


def zig_type(ctype, array):
    """ctype: the C type without the parameter name, e.g. 'const uint64_t *const *'; array: '[4]' suffix or ''"""
    t = ctype.replace(" *", "*").replace("* ", "*").strip()
    const = t.startswith("const ")
    base = t[6:] if const else t
    stars = base.count("*")
    base = base.replace("*", "").replace("const", "").strip()
    if array:  # fixed-size array parameter = pointer to that many elements
        n = array.strip("[]")
        el = SCALARS[base]
        if not n.isdigit():
            n = "8"  # ZG_PROF_NKERNELS (internal header only)
        return f"*const [{n}]{el}" if const else f"*[{n}]{el}"
    if base in HANDLES:
        return HANDLES[base] if stars == 0 else "*" + HANDLES[base]
    if base == "zg_msm_config":
        return "?*const MsmConfig"
    if base == "zg_psc_term":
        return "?[*]const PscTerm"
    if base == "zg_col_t":
        return "?[*]const Column"
    if base == "void":
        if stars == 2 and "*const*" in t.replace(" ", ""):
            return "?[*]const ?*anyopaque"  # array of stream handles
        if stars == 2:
            return "*?*anyopaque"
        return "?*const anyopaque" if const else "?*anyopaque"
    if base == "char":
        return "[*:0]const u8"
    el = SCALARS[base]
    if stars == 0:
        return el
    if stars == 2 and const and "*const*" not in t.replace(" ", ""):
        return f"*?[*]const {el}"  # out-parameter receiving a device address (const T **)
    if stars == 2:  # array of pointers to arrays (scalar batches; output tables when the pointed-to elements are not const)
        return f"?[*]const ?[*]const {el}" if const else f"?[*]const ?[*]{el}"
    if base in ("int", "size_t") and not const:
        return f"?*{el}"  # single out-parameter
    return f"?[*]const {el}" if const else f"?[*]{el}"


def parse_header(text):
    protos = []
    for m in re.finditer(r"ZG_API\s+([^;(]*?)\b(zg_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        ret, name, args = " ".join(m.group(1).split()), m.group(2), split_params(m.group(3))
        params = []
        for a in args:
            am = re.match(r"^(.*?)(\w+)\s*((?:\[\w*\])?)$", a)
            ctype, pname, arr = am.group(1).strip(), am.group(2), am.group(3)
            params.append((pname, zig_type(ctype, arr)))
        zret = {"int": "c_int", "void": "void", "size_t": "usize", "uint32_t": "u32", "const char *": "[*:0]const u8"}[ret]
        protos.append((name, params, zret))
    return protos


def generate():
    text = open(HDR).read()
    protos = parse_header(text)
    consts = re.findall(r"#define (ZG_(?:OK|ERR_\w+|FIELD_\w+|OP_\w+|SC_\w+|PSC_\w+|ABI_\w+|FEATURE_\w+|COL_\w+)) (\d+)u?\b", text)
    out = [
        "//! extern declarations of libzolt_gpu.so, for Zolt's src/gpu/ffi.zig.",
        "//! GENERATED from include/zolt_gpu.h by tools/gen_zig_ffi.py — do not edit; tests/test_abi_and_host.py holds the two together.",
        "//! COMPILE-UNVERIFIED: the build image has no Zig toolchain (Zig >= 0.14 syntax, build.zig.zon:5 of the reference).",
        "//! The same ABI is exercised for real by zolt_amd/host/zolt_host.hpp (C++) and zolt_amd/lib.py (ctypes).",
        "//!",
        "//! Field elements cross the boundary as they are: BN254Scalar / BN254BaseField are `struct { limbs: [4]u64 }`",
        "//! (src/field/mod.zig:131,583-584), Montgomery form, so `[]const F` is passed as `[*]const u64` via @ptrCast.",
        "",
    ] + [f"pub const {zname} = ?*opaque {{}}; // {cname}" for cname, zname in HANDLES.items()] + [  # every handle type the externs below name
        "",
        "pub const MsmConfig = extern struct { window_bits: c_int = 0, precompute_levels: c_int = 0, expected_uses: c_int = 0 };",
        "pub const Column = extern struct { kind: u32 = 0, a: u32 = 0, b: u32 = 0, data: ?*const anyopaque = null, aux: ?*const anyopaque = null }; // zg_col_t",
        "pub const PscTerm = extern struct { n_prod: c_int = 0, prod: [4]c_int = .{ 0, 0, 0, 0 }, n_lin: c_int = 0, lin: [4]c_int = .{ 0, 0, 0, 0 }, lin_coeff: [16]u64 = .{0} ** 16 }; // zg_psc_term",
        "",
    ]
    for name, val in consts:
        out.append(f"pub const {name[3:]}: {'u32' if name.startswith(('ZG_ABI_', 'ZG_FEATURE_', 'ZG_COL_')) else 'c_int'} = {val};")
    out.append("")
    for name, params, zret in protos:
        ps = ", ".join(f"{p}: {t}" for p, t in params)
        out.append(f"pub extern fn {name}({ps}) {zret};")
    return "\n".join(out) + "\n"


def main():
    new = generate()
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        if cur != new:
            print("zig/gpu/ffi.zig is stale: run tools/gen_zig_ffi.py")
            return 1
        return 0
    with open(OUT, "w") as f:
        f.write(new)
    print("wrote", OUT)
    return 0


if __name__ == "__main__":
    sys.exit(main())
