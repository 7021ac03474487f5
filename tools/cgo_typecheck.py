#!/usr/bin/env python3
"""The C-facing half of shim/go, type-checked by a real C compiler (there is no Go toolchain in this image).

cgo's rule at the boundary is strict: an argument of C.f(...) must have EXACTLY the Go type cgo derives from the parameter
(`*C.uint32_t` for `const uint32_t *`; no integer widening, no pointer conversions), a struct field must exist, a C function's
result must have the type of the Go parameter it is handed to.  Since Go 1.10 cgo translates `typedef X Y` into a type alias, so
"the same type" is C's notion after typedef resolution — which is what `__builtin_types_compatible_p` decides.  This tool

  1. reads every Go file of the overlay that imports "C", and the cgo preamble of each (compiled as it stands);
  2. infers the C type of every argument it can from the Go text — conversions `C.T(x)`, `(*C.T)(unsafe.Pointer(…))`, `nil`,
     `&v` / `v` / `r.field` of declared variables, parameters and struct fields, results of Go helpers (`ptr8`) and of nested
     C calls, untyped constants;
  3. takes the parameter types from the prototypes in include/*.h and the preamble (const dropped, as cgo does);
  4. emits one `_Static_assert(__builtin_types_compatible_p(argument type, parameter type))` per argument, per composite-literal
     field, per `v.field` access of a C struct and per C result handed to a Go helper, and runs `gcc -std=c11 -fsyntax-only`.

It proves less than a Go compiler (Go-side expressions it cannot type are counted as "unchecked", Go's own typing is not
modelled) and exactly what a C compiler can: no call in shim/go passes a pointer or an integer of the wrong C type, in the wrong
position, or to a parameter that does not exist.  tests/test_go_shim.py runs it and seeds the defects it must catch.

usage: cgo_typecheck.py [shim root] [-v]      exit code 0 = every emitted assertion holds"""
from __future__ import annotations

import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C_SCALARS = {"int", "uint", "long", "ulong", "longlong", "ulonglong", "char", "schar", "uchar", "short", "ushort", "float", "double",
             "size_t", "ssize_t", "uintptr_t", "intptr_t", "int8_t", "uint8_t", "int16_t", "uint16_t", "int32_t", "uint32_t",
             "int64_t", "uint64_t"}
GO_TO_C_NAME = {"uint": "unsigned int", "ulong": "unsigned long", "longlong": "long long", "ulonglong": "unsigned long long",
                "schar": "signed char", "uchar": "unsigned char", "ushort": "unsigned short"}
NIL, INTCONST = "<nil>", "<untyped integer constant>"


# ---- small lexical helpers -------------------------------------------------------------------------------------------------
def matching(text: str, open_at: int, pair: str = "()") -> int:
    depth = 0
    for k in range(open_at, len(text)):
        if text[k] == pair[0]:
            depth += 1
        elif text[k] == pair[1]:
            depth -= 1
            if depth == 0:
                return k
    return -1


def split_top(text: str, sep: str = ",") -> list[str]:
    out, depth, cur = [], 0, ""
    for ch in text:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def strip_go_comments(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", lambda m: " " * 0 + "\n" * m.group(0).count("\n"), text, flags=re.S)
    return "\n".join(l.split("//")[0] if '"' not in l.split("//")[0] or l.split("//")[0].count('"') % 2 == 0 else l
                     for l in text.splitlines())


def line_of(text: str, pos: int) -> int:
    return text.count("\n", 0, pos) + 1


# ---- the C side: prototypes with parameter types -----------------------------------------------------------------------------
def c_prototypes(text: str) -> dict[str, tuple[str, list[str]]]:
    """function name → (result type, [parameter types]) for every prototype / inline definition; const dropped"""
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = "\n".join(l.split("//")[0] for l in text.splitlines() if not l.lstrip().startswith("#"))
    text = re.sub(r'extern\s+"C"\s*\{', "", text)
    out = {}
    for m in re.finditer(r"\b(\w+)\s*\(", text):
        name = m.group(1)
        if name in ("if", "for", "while", "return", "sizeof", "switch", "defined"):
            continue
        end = matching(text, m.end() - 1)
        if end < 0:
            continue
        after = text[end + 1:end + 40].lstrip()
        if not (after.startswith(";") or after.startswith("{")):
            continue
        # the result type: the tokens between the previous ';' / '}' / start and the name
        start = max(text.rfind(";", 0, m.start()), text.rfind("}", 0, m.start()), text.rfind("{", 0, m.start())) + 1
        ret = text[start:m.start()].strip()
        if not ret or "typedef" in ret or "(" in ret or "=" in ret or ret.split()[-1] in ("return", "else"):
            continue
        ret = re.sub(r"\b(static|inline|extern|const|__attribute__\s*\(\(.*?\)\))\b", " ", ret)
        ret = " ".join(ret.replace("*", " * ").split())
        params = []
        plist = text[m.end():end].strip()
        if plist and plist != "void":
            for p in split_top(plist):
                p = re.sub(r"\bconst\b", " ", p).strip()
                arr = "[" in p
                p = re.sub(r"\[[^\]]*\]", "", p).strip()
                mm = re.match(r"^(.*?)(\b\w+)$", p, re.S)       # the last identifier is the parameter's name …
                ty = mm.group(1).strip() if mm and mm.group(1).strip() else p     # … unless the type stands alone
                ty = " ".join(ty.replace("*", " * ").split()) + (" *" if arr else "")
                params.append(ty)
        out.setdefault(name, (ret, params))
    return out


def c_typedef_names(text: str) -> set[str]:
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\}\s*(\w+)\s*;", text))
    names |= set(re.findall(r"typedef\s+struct\s+\w+\s+(\w+)\s*;", text))
    names |= set(re.findall(r"typedef\s+[\w\s\*]+?\(\s*\*\s*(\w+)\s*\)\s*\(", text))      # function-pointer typedefs
    names |= set(re.findall(r"typedef\s+[\w\s]+?\b(\w+)\s*;", text))
    return names


# ---- the Go side ---------------------------------------------------------------------------------------------------------------
def go_type_to_c(t: str, known: set[str]) -> str | None:
    """`*C.uint8_t` → `uint8_t *`; None for a type that is not a C type"""
    t = t.strip()
    stars = 0
    while t.startswith("*"):
        stars += 1
        t = t[1:].strip()
    if t == "unsafe.Pointer":
        base = "void *"
    elif t.startswith("C."):
        n = t[2:]
        if n.startswith("struct_"):
            base = "struct " + n[7:]
        elif n in C_SCALARS or n in known:
            base = GO_TO_C_NAME.get(n, n)
        else:
            return None
    else:
        return None
    return (base + " *" * stars).replace("* *", "**").replace("**", "* *") if stars else base


class GoFile:
    def __init__(self, path: str, include_dirs: list[str]):
        self.path = path
        self.raw = open(path).read()
        head = self.raw.split('import "C"')[0]
        self.preamble = head[head.rindex("/*") + 2:head.rindex("*/")] if "/*" in head else ""
        self.preamble_c = "\n".join(l for l in self.preamble.splitlines() if not l.lstrip().startswith("#cgo"))
        self.code = strip_go_comments(self.raw.split('import "C"', 1)[1])
        self.code_offset_lines = self.raw.split('import "C"', 1)[0].count("\n")
        hdr_text, todo, seen = "", re.findall(r'#include\s+"([^"]+)"', self.preamble), set()
        while todo:                                                           # (headers included by headers too)
            inc = todo.pop(0)
            if inc in seen:
                continue
            seen.add(inc)
            for d in include_dirs:
                if os.path.exists(os.path.join(d, inc)):
                    t = open(os.path.join(d, inc)).read()
                    hdr_text += t + "\n"
                    todo += re.findall(r'#include\s+"([^"]+)"', t)
                    break
        self.protos = c_prototypes(hdr_text + "\n" + self.preamble_c)
        self.ctypes = c_typedef_names(hdr_text + "\n" + self.preamble_c)


def parse_go_params(plist: str) -> list[tuple[str, str]]:
    """`a, b []byte, n int` → [(a, []byte), (b, []byte), (n, int)]"""
    items = split_top(plist)
    out, pending = [], []
    for it in items:
        parts = it.split(None, 1)
        if len(parts) == 1:
            pending.append(parts[0])
        else:
            for nme in pending:
                out.append((nme, parts[1].strip()))
            pending = []
            out.append((parts[0], parts[1].strip()))
    return out


def go_functions(code: str):
    """(name, receiver (name, type) | None, params [(name, type)], result text, body, position of the body) per top-level func"""
    out = []
    for m in re.finditer(r"^func\s*(\((\w+)\s+(\*?\w+)\)\s*)?(\w+)\(", code, re.M):
        close = matching(code, m.end() - 1)
        if close < 0:
            continue
        brace = code.find("{", close)
        result = code[close + 1:brace].strip()
        if result.startswith("(") and matching(code, close + 1 + code[close + 1:brace].index("(")) > brace:
            brace = code.find("{", matching(code, close + 1 + code[close + 1:brace].index("(")))
            result = code[close + 1:brace].strip()
        end = matching(code, brace, "{}")
        recv = (m.group(2), m.group(3)) if m.group(1) else None
        out.append((m.group(4), recv, parse_go_params(code[m.end():close]), result, code[brace:end + 1], brace))
    return out


def struct_fields(code: str) -> dict[str, dict[str, str]]:
    """Go struct name → {field: type} (single-line and block forms)"""
    out = {}
    for m in re.finditer(r"^type\s+(\w+)\s+struct\s*\{", code, re.M):
        end = matching(code, m.end() - 1, "{}")
        body = code[m.end():end]
        fields = {}
        for part in re.split(r"[;\n]", body):
            part = part.strip()
            if not part:
                continue
            for nme, ty in parse_go_params(part.replace("\t", " ")):
                fields[nme] = ty.split("`")[0].strip()
        out[m.group(1)] = fields
    return out


class Checker:
    def __init__(self, gf: GoFile, pkg_funcs: dict, pkg_structs: dict):
        self.gf, self.pkg_funcs, self.pkg_structs = gf, pkg_funcs, pkg_structs
        self.asserts: list[str] = []
        self.checked = self.unchecked = 0
        self.unchecked_notes: list[str] = []

    # -- types of Go expressions, as C type strings --------------------------------------------------------------------------
    def type_of(self, expr: str, env: dict[str, str]) -> str | None:
        e = expr.strip()
        while e.startswith("(") and matching(e, 0) == len(e) - 1:
            e = e[1:-1].strip()
        if e == "nil":
            return NIL
        if re.fullmatch(r"-?(0x[0-9a-fA-F_]+|\d[\d_]*)", e):
            return INTCONST
        m = re.match(r"^\((\*+C\.\w+|\*+unsafe\.Pointer)\)\(", e)          # (*C.T)(…)
        if m and matching(e, m.end() - 1) == len(e) - 1:
            return go_type_to_c(m.group(1), self.gf.ctypes)
        m = re.match(r"^C\.(\w+)\(", e)
        if m and matching(e, m.end() - 1) == len(e) - 1:
            n = m.group(1)
            if n in self.gf.protos:                                          # a nested C call: its result type
                return self.gf.protos[n][0]
            if n == "GoString":
                return None
            return go_type_to_c("C." + n, self.gf.ctypes)                    # a conversion
        m = re.match(r"^C\.(\w+)\{", e)                                      # a composite literal
        if m:
            return go_type_to_c("C." + m.group(1), self.gf.ctypes)
        if re.match(r"^unsafe\.Pointer\(", e) and matching(e, e.index("(")) == len(e) - 1:
            return "void *"
        if e.startswith("&"):
            inner = self.type_of(e[1:], env)
            return None if inner in (None, NIL, INTCONST) else inner + " *"
        m = re.match(r"^(\w+)\(", e)                                         # a Go helper of the package
        if m and matching(e, m.end() - 1) == len(e) - 1 and m.group(1) in self.pkg_funcs:
            return go_type_to_c(self.pkg_funcs[m.group(1)]["result"], self.gf.ctypes)
        if re.fullmatch(r"\w+", e):
            return go_type_to_c(env[e], self.gf.ctypes) if e in env else None
        m = re.fullmatch(r"(\w+)\.(\w+)", e)                                 # v.field
        if m and m.group(1) in env:
            owner = env[m.group(1)].lstrip("*")
            if owner in self.pkg_structs and m.group(2) in self.pkg_structs[owner]:
                return go_type_to_c(self.pkg_structs[owner][m.group(2)], self.gf.ctypes)
            cowner = go_type_to_c(env[m.group(1)], self.gf.ctypes)
            if cowner:                                                       # a field of a C struct: let the compiler say
                return "__typeof__(((%s *)0)->%s)" % (cowner.rstrip(" *"), m.group(2))
        return None

    def local_env(self, fn) -> dict[str, str]:
        name, recv, params, result, body, _ = fn
        env = {}
        if recv:
            env[recv[0]] = recv[1]
        for n, t in params:
            env[n] = t
        if result.startswith("("):                                           # named results
            for n, t in parse_go_params(result[1:-1]):
                if re.fullmatch(r"\w+", n):
                    env[n] = t
        for m in re.finditer(r"\bvar\s+([\w,\s]+?)\s+(\*?[\w\.\[\]]+)\s*(?:=|$)", body, re.M):
            for n in m.group(1).split(","):
                env[n.strip()] = m.group(2)
        for m in re.finditer(r"\b(\w+)\s*:=\s*C\.(\w+)\(", body):               # (a C call's result, however many lines the call takes)
            if m.group(2) in self.gf.protos:
                env[m.group(1)] = "<c>" + self.gf.protos[m.group(2)][0]
        for m in re.finditer(r"(?:^|\bif\s|\bswitch\s|;)\s*(\w+)\s*:=\s*(.+?)\s*(?:;|\{\s*$|$)", body, re.M):
            t = self.type_of(m.group(2), env)
            if t and t not in (NIL, INTCONST) and not t.startswith("__typeof__"):
                env[m.group(1)] = "<c>" + t                                  # already a C type
        return env

    def assert_same(self, got: str, want: str, where: str):
        want_c = want
        if got == NIL:
            cond = "__builtin_classify_type((%s)0) == 5" % want_c
            msg = "%s: nil passed where the prototype wants %s" % (where, want)
        elif got == INTCONST:
            cond = "__builtin_classify_type((%s)0) == 1 || __builtin_classify_type((%s)0) == 8" % (want_c, want_c)
            msg = "%s: an untyped constant passed where the prototype wants %s" % (where, want)
        else:
            cond = "__builtin_types_compatible_p(%s, %s)" % (got, want_c)
            msg = "%s: Go passes %s, C wants %s" % (where, got, want)
        self.asserts.append('_Static_assert(%s, "%s");' % (cond, msg.replace('"', "'").replace("\\", "")))
        self.checked += 1

    def run(self):
        code = self.gf.code
        rel = os.path.relpath(self.gf.path, ROOT)
        for fn in go_functions(code):
            name, recv, params, result, body, body_pos = fn
            env = self.local_env(fn)
            env = {k: v for k, v in env.items()}
            # `<c>`-tagged entries are C types already
            real_type_of = self.type_of

            def tof(expr, env=env):
                e = expr.strip()
                if re.fullmatch(r"\w+", e) and e in env and env[e].startswith("<c>"):
                    return env[e][3:]
                if e.startswith("&") and re.fullmatch(r"\w+", e[1:].strip()) and env.get(e[1:].strip(), "").startswith("<c>"):
                    return env[e[1:].strip()][3:] + " *"
                m2 = re.fullmatch(r"(\w+)\.(\w+)", e)
                if m2 and env.get(m2.group(1), "").startswith("<c>"):
                    return "__typeof__(((%s *)0)->%s)" % (env[m2.group(1)][3:].rstrip(" *"), m2.group(2))
                return real_type_of(expr, {k: v for k, v in env.items() if not v.startswith("<c>")})

            # 1. every C call: argument i against parameter i
            for m in re.finditer(r"\bC\.(\w+)\(", body):
                fname = m.group(1)
                if fname not in self.gf.protos:
                    continue
                end = matching(body, m.end() - 1)
                args = split_top(body[m.end():end])
                ret, ptypes = self.gf.protos[fname]
                line = line_of(code, body_pos + m.start()) + self.gf.code_offset_lines
                if len(args) != len(ptypes):
                    self.asserts.append('_Static_assert(0, "%s:%d: C.%s called with %d arguments, the prototype has %d");'
                                        % (rel, line, fname, len(args), len(ptypes)))
                    continue
                for k, (a, pt) in enumerate(zip(args, ptypes)):
                    got = tof(a)
                    if got is None:
                        self.unchecked += 1
                        self.unchecked_notes.append("%s:%d: C.%s argument %d `%s`" % (rel, line, fname, k + 1, a))
                        continue
                    self.assert_same(got, pt, "%s:%d: C.%s argument %d" % (rel, line, fname, k + 1))
            # 2. a C result handed to a Go helper of the package: its type against the helper's parameter
            for m in re.finditer(r"(?<![\w.])(?:\w+\.)?(\w+)\(\s*C\.(\w+)\(", body):
                helper, fname = m.group(1), m.group(2)
                if helper in self.pkg_funcs and fname in self.gf.protos and len(self.pkg_funcs[helper]["params"]) >= 1:
                    inner_open = body.index("(", m.start(2))
                    inner_end = matching(body, inner_open)
                    outer_end = matching(body, body.index("(", m.start(1)))
                    if body[inner_end + 1:outer_end].strip() == "":                  # the call is the helper's only argument
                        want = go_type_to_c(self.pkg_funcs[helper]["params"][0][1], self.gf.ctypes)
                        if want:
                            line = line_of(code, body_pos + m.start()) + self.gf.code_offset_lines
                            self.assert_same(self.gf.protos[fname][0], want, "%s:%d: result of C.%s handed to %s()" % (rel, line, fname, helper))
            # 3. composite literals of C structs: every field exists and has the type of the value
            for m in re.finditer(r"\bC\.(\w+)\{", body):
                cty = go_type_to_c("C." + m.group(1), self.gf.ctypes)
                if not cty:
                    continue
                end = matching(body, m.end() - 1, "{}")
                line = line_of(code, body_pos + m.start()) + self.gf.code_offset_lines
                for item in split_top(body[m.end():end]):
                    if ":" not in item:
                        continue
                    fld, val = item.split(":", 1)
                    got = tof(val)
                    want = "__typeof__(((%s *)0)->%s)" % (cty, fld.strip())
                    if got is None:
                        self.asserts.append("_Static_assert(sizeof(%s) > 0, \"\");" % want)      # the field exists, at least
                        self.unchecked += 1
                        continue
                    self.assert_same(got, want, "%s:%d: field %s of C.%s" % (rel, line, fld.strip(), m.group(1)))
            # 4. fields read from variables that hold C structs
            for var, ty in env.items():
                cty = ty[3:] if ty.startswith("<c>") else go_type_to_c(ty, self.gf.ctypes)
                if not cty or cty.rstrip(" *") in C_SCALARS or cty.startswith("void") or cty.rstrip(" *") in GO_TO_C_NAME.values():
                    continue
                if cty.count("*") > 1:
                    continue
                for m in re.finditer(r"(?<![\w.])%s\.(\w+)\b(?!\()" % re.escape(var), body):
                    self.asserts.append('_Static_assert(sizeof(((%s *)0)->%s) > 0, "%s: %s.%s");'
                                        % (cty.rstrip(" *"), m.group(1), rel, var, m.group(1)))
                    self.checked += 1


def main(argv: list[str]) -> int:
    verbose = "-v" in argv
    argv = [a for a in argv if a != "-v"]
    shim_root = argv[0] if argv else os.path.join(ROOT, "shim", "go")
    inc = [os.path.join(ROOT, "include")]
    total_checked = total_unchecked = 0
    failures = []
    for pkg_dir in sorted({os.path.dirname(p) for p in glob.glob(os.path.join(shim_root, "**", "*.go"), recursive=True)}):
        files = [GoFile(p, inc + [pkg_dir]) for p in sorted(glob.glob(os.path.join(pkg_dir, "*.go"))) if 'import "C"' in open(p).read()]
        if not files:
            continue
        pkg_funcs, pkg_structs = {}, {}
        for gf in files:                                                      # helpers and structs are visible package-wide
            for name, recv, params, result, body, _ in go_functions(gf.code):
                pkg_funcs.setdefault(name, {"params": params, "result": result})
            pkg_structs.update(struct_fields(gf.code))
        # //export functions: the prototype a preamble of the package declares for the trampoline must be the one cgo will
        # generate from the Go signature (a mismatch is a "conflicting types" error when cgo's _cgo_export.h is compiled)
        export_asserts = {}
        for gf in files:
            for m in re.finditer(r"^//export (\w+)\s*\n(?://[^\n]*\n)*func (\w+)\(", gf.raw, re.M):
                name = m.group(1)
                fn = next((f for f in go_functions(gf.code) if f[0] == name), None)
                decl = next(((g, g.protos[name]) for g in files if name in c_prototypes(g.preamble_c)), None)
                rel = os.path.relpath(gf.path, ROOT)
                if fn is None or m.group(2) != name:
                    export_asserts.setdefault(gf.path, []).append('_Static_assert(0, "%s: //export %s is not followed by func %s");' % (rel, name, name))
                    continue
                if decl is None:
                    continue                                                   # (only called through a pointer cgo creates: nothing to compare)
                owner, (ret, ptypes) = decl
                got_params = [go_type_to_c(t, owner.ctypes) for _, t in fn[2]]
                got_ret = go_type_to_c(fn[3], owner.ctypes) if fn[3] else "void"
                lst = export_asserts.setdefault(owner.path, [])
                if len(got_params) != len(ptypes) or None in got_params or got_ret is None:
                    lst.append('_Static_assert(0, "%s: //export %s%s does not match the declaration %s(%s)");'
                               % (rel, name, tuple(t for _, t in fn[2]), name, ", ".join(ptypes)))
                    continue
                for k, (g_, w_) in enumerate(zip(got_params + [got_ret], ptypes + [ret])):
                    what = "result" if k == len(ptypes) else "parameter %d" % (k + 1)
                    lst.append('_Static_assert(__builtin_types_compatible_p(%s, %s), "%s: //export %s %s is %s, the preamble declares %s");'
                               % (g_, w_, rel, name, what, g_, w_))
        for gf in files:
            ck = Checker(gf, pkg_funcs, pkg_structs)
            ck.run()
            ck.asserts += export_asserts.get(gf.path, [])
            ck.checked += len(export_asserts.get(gf.path, []))
            total_checked += ck.checked
            total_unchecked += ck.unchecked
            src = "#include <stddef.h>\n#include <stdint.h>\n" + gf.preamble_c + "\n" + "\n".join(ck.asserts) + "\nint main(void) { return 0; }\n"
            with tempfile.TemporaryDirectory() as td:
                cpath = os.path.join(td, "cgo_check.c")
                open(cpath, "w").write(src)
                cmd = ["gcc", "-std=gnu11", "-fsyntax-only", "-Wall", "-Werror=implicit-function-declaration",
                       "-Werror=incompatible-pointer-types", "-Werror=int-conversion", "-Wno-unused-function"] + ["-I" + d for d in inc + [pkg_dir]] + [cpath]
                out = subprocess.run(cmd, capture_output=True, text=True)
            rel = os.path.relpath(gf.path, ROOT)
            if out.returncode != 0:
                msgs = re.findall(r"static assertion failed: \"(.*?)\"", out.stderr)
                others = [l for l in out.stderr.splitlines() if "error:" in l and "static assertion failed" not in l]
                failures += msgs + ["%s: %s" % (rel, l.split("error:", 1)[1].strip()) for l in others]
            print("%s: %d assertions, %d arguments not typed by this tool%s" % (rel, ck.checked, ck.unchecked, "" if out.returncode == 0 else "  FAILED"))
            if verbose:
                for n in ck.unchecked_notes:
                    print("   unchecked:", n)
    for f in failures:
        print("ERROR:", f)
    print("cgo boundary: %d assertions compiled, %d arguments left to the Go compiler, %d problems" % (total_checked, total_unchecked, len(failures)))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
