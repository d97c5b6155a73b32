#!/usr/bin/env python3
"""A stand-in for the compiler this image lacks, one notch above bracket counting: every identifier USED in zig/gpu/backend.zig (and the
generated ffi.zig) must be DECLARED — a file- or container-level `const` / `var` / `fn`, a parameter, a local or a capture of the
enclosing function — or be a keyword / primitive type / builtin. Field accesses (`.name`), declaration sites (`name:`) and block labels
are not uses. Scopes are flattened per function (a local declared anywhere in a function counts for the whole function): the lint
catches misspelt and missing names, not shadowing or order. No Zig toolchain exists in the build image; this does NOT make the shim
compile-verified.     python tools/zig_lint.py [file ...]      exit 1 and one line per finding"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYWORDS = set("""addrspace align allowzero and anyframe anytype asm async await break callconv catch comptime const continue defer else enum errdefer
error export extern fn for if inline linksection noalias noinline nosuspend opaque or orelse packed pub resume return struct suspend switch
test threadlocal try union unreachable usingnamespace var volatile while""".split())
PRIMITIVES = set("""bool void type anyopaque anyerror noreturn usize isize c_char c_short c_ushort c_int c_uint c_long c_ulong c_longlong c_ulonglong
f16 f32 f64 f80 f128 comptime_int comptime_float null undefined true false""".split())


def strip(src):
    """comments, string and char literals -> spaces (offsets kept)"""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if src.startswith("//", i):
            j = src.find("\n", i)
            j = n if j < 0 else j
            out.append(" " * (j - i))
            i = j
        elif c == '"':
            j = i + 1
            while j < n and src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            out.append('"' + " " * (j - i - 1) + '"')
            i = j + 1
        elif c == "'":
            j = i + 1
            while j < n and src[j] != "'":
                j += 2 if src[j] == "\\" else 1
            out.append(" " * (j + 1 - i))
            i = j + 1
        else:
            out.append(c)
            i += 1
    return "".join(out)


TOKEN = re.compile(r"@?[A-Za-z_]\w*|\d\w*|\.\.\.|\.\.|=>|[^\s\w]")


def tokens(code):
    return [(m.group(0), m.start()) for m in TOKEN.finditer(code)]


def match_close(toks, i, open_c, close_c):
    depth = 0
    for j in range(i, len(toks)):
        if toks[j][0] == open_c:
            depth += 1
        elif toks[j][0] == close_c:
            depth -= 1
            if depth == 0:
                return j
    return len(toks) - 1


def functions(toks):
    """[(name, params_open, params_close, body_open or None, body_close or None)]"""
    out = []
    for i, (t, _) in enumerate(toks):
        if t == "fn" and i + 2 < len(toks) and re.match(r"[A-Za-z_]", toks[i + 1][0]) and toks[i + 2][0] == "(":
            pc = match_close(toks, i + 2, "(", ")")
            j, depth, body = pc + 1, 0, None
            while j < len(toks):  # the return type may hold `struct { ... }`: the body is the first `{` that such a type does not open
                t2 = toks[j][0]
                if t2 == "struct" and toks[j + 1][0] == "{":
                    j = match_close(toks, j + 1, "{", "}") + 1
                    continue
                if t2 == "{":
                    body = j
                    break
                if t2 == ";":
                    break
                j += 1
            out.append((toks[i + 1][0], i + 2, pc, body, match_close(toks, body, "{", "}") if body is not None else None))
    return out


def lint(path):
    src = open(path).read()
    toks = tokens(strip(src))
    fns = functions(toks)
    in_fn = [None] * len(toks)  # innermost function (by index in fns) a token belongs to: params, return type, body
    for k, (_, po, pc, bo, bc) in enumerate(fns):
        for j in range(po, (bc if bc is not None else pc) + 1):
            in_fn[j] = k  # later (nested) functions overwrite
    ident = lambda s: re.match(r"[A-Za-z_]\w*$", s) is not None
    glob, local = set(n for n, *_ in fns), [set() for _ in fns]
    skip_to = -1
    for i, (t, _) in enumerate(toks):
        if i <= skip_to:
            continue
        nxt = toks[i + 1][0] if i + 1 < len(toks) else ""
        prev = toks[i - 1][0] if i else ""
        tgt = glob if in_fn[i] is None else local[in_fn[i]]
        if t in ("const", "var") and ident(nxt):
            tgt.add(nxt)
        elif ident(t) and nxt == ":" and prev != "." and t not in KEYWORDS and in_fn[i] is not None:
            tgt.add(t)  # a parameter or a block label (a container's FIELD declared outside any function is reached through `.` only)
        elif t == "|" and in_fn[i] is not None:  # capture list |a, *b| (opening bar: the previous token closes an expression)
            if prev in (")", "else", "catch") or ident(prev):
                j = i + 1
                while j < len(toks) and toks[j][0] != "|" and j - i <= 8:  # a capture list is a few names
                    if ident(toks[j][0]):
                        local[in_fn[i]].add(toks[j][0])
                    j += 1
                skip_to = j  # the closing bar is not another opening
    # enclosing functions' names are visible in nested functions (container methods inside a `fn (...) type { return struct {...} }`)
    parents = []
    for k, (_, po, *_r) in enumerate(fns):
        parents.append(in_fn[po - 1] if po >= 1 else None)
    def visible(k):
        s = set()
        while k is not None:
            s |= local[k]
            k = parents[k]
        return s
    in_error_set = [False] * len(toks)  # `error{ A, B }`: the members are declarations
    for i, (t, _) in enumerate(toks):
        if t == "error" and i + 1 < len(toks) and toks[i + 1][0] == "{":
            for j in range(i + 1, match_close(toks, i + 1, "{", "}")):
                in_error_set[j] = True
    findings = []
    for i, (t, off) in enumerate(toks):
        if t == "_" or in_error_set[i]:
            continue
        if not ident(t) or t in KEYWORDS or t in PRIMITIVES or re.match(r"[iu]\d+$", t):
            continue
        prev = toks[i - 1][0] if i else ""
        nxt = toks[i + 1][0] if i + 1 < len(toks) else ""
        if prev in (".", "fn", "const", "var") or prev == ":" and toks[i - 2][0] in ("break", "continue") or nxt == ":" and prev != "?":
            continue
        if t in glob or (in_fn[i] is not None and t in visible(in_fn[i])):
            continue
        line = src.count("\n", 0, off) + 1
        findings.append(f"{os.path.relpath(path, ROOT)}:{line}: '{t}' is used but never declared")
    # arity of direct calls `name(...)` to the file's own free functions (method calls go through `.` and are not judged)
    def n_args(open_i):
        close_i, depth, n, any_tok = match_close(toks, open_i, "(", ")"), 0, 0, False
        for j in range(open_i + 1, close_i):
            tj = toks[j][0]
            depth += tj in "([{"
            depth -= tj in ")]}"
            any_tok = True
            if tj == "," and depth == 0 and j + 1 < close_i:  # (a trailing comma adds no argument)
                n += 1
        return n + 1 if any_tok else 0
    arity = {}
    for name, po, pc, bo, bc in fns:
        if in_fn[po - 2] is None:  # declared outside any function body
            arity.setdefault(name, set()).add(n_args(po))
    for i, (t, off) in enumerate(toks):
        if t in arity and len(arity[t]) == 1 and i + 1 < len(toks) and toks[i + 1][0] == "(" and toks[i - 1][0] not in (".", "fn"):
            got, want = n_args(i + 1), next(iter(arity[t]))
            if got != want:
                findings.append(f"{os.path.relpath(path, ROOT)}:{src.count(chr(10), 0, off) + 1}: '{t}' takes {want} arguments, called with {got}")
    return findings, len([1 for t, _ in toks if ident(t)])


def main():
    files = sys.argv[1:] or [os.path.join(ROOT, "zig", "gpu", "backend.zig"), os.path.join(ROOT, "zig", "gpu", "ffi.zig")]
    bad = []
    for f in files:
        found, n = lint(f)
        bad += found
        print(f"{os.path.relpath(f, ROOT)}: {n} identifiers, {len(found)} undeclared")
    for b in bad:
        print(b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
