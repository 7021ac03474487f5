#!/usr/bin/env python3
"""Go-to-Go calls of shim/go, checked for ARITY against the definitions a compiler would bind them to (no Go toolchain here).

tools/check_go_shim.py proves that every identifier exists and tools/cgo_typecheck.py type-checks the C boundary; what is
left for "the Go side has never met a compiler" is the typing between the Go packages of the overlay and the reference.  This
tool resolves what it can without implementing Go's type system:

  packages     shim/go/<pkg>/*.go overlaid on /root/reference/<pkg>/*.go (core, messages, messages/proto → proto), plus
               shim/go/ibftgpu and shim/go/hoststore
  definitions  every func, every method by receiver type, every interface's methods (embedded interfaces followed), every
               struct's fields with their types
  environment  per function: receiver, parameters, named results, `var x T`, `x := T{…}` / `&T{…}`, `x, ok := y.(T)`,
               `x[, err] := pkg.Func(…)` / `recv.Method(…)` (first result's type)
  calls        `pkg.Func(…)` and `a.b.c.Method(…)` whose receiver chain resolves through that environment and the field
               tables: the number of arguments must fit the parameter list (variadic honoured), and where the call is the
               whole right-hand side of an assignment, the number of names on the left must be the number of results

A call whose receiver does not resolve is counted, not judged.  Exit code 0 = every resolved call fits.
Also: the methods of the types the overlay assigns to interfaces (*hoststore.Store → core.Messages, core.hostStore;
*messages.Messages → core.Messages, core.batchStore) compared with those interfaces parameter type by parameter type.
usage: go_arity_check.py [shim root] [-v]"""
from __future__ import annotations

import glob
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cgo_typecheck as T  # noqa: E402  (lexical helpers: matching, split_top, strip_go_comments, go_functions, parse_go_params)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
IMPORT_TO_PKG = {"github.com/0xPolygon/go-ibft/messages/proto": "proto", "github.com/0xPolygon/go-ibft/messages": "messages",
                 "github.com/0xPolygon/go-ibft/ibftgpu": "ibftgpu", "github.com/0xPolygon/go-ibft/core": "core",
                 "github.com/0xPolygon/go-ibft/hoststore": "hoststore"}


def norm_type(t: str) -> str:
    """a Go type with the parameter names inside func types dropped and white space removed"""
    t = " ".join(t.split())
    out, k = "", 0
    while True:
        j = t.find("func(", k)
        if j < 0:
            return (out + t[k:]).replace(" ", "")
        close = T.matching(t, j + 4)
        inner = t[j + 5:close].strip().rstrip(",")
        out += t[k:j] + "func(" + ",".join(type_list(inner)) + ")"
        k = close + 1


def type_list(plist: str) -> list[str]:
    """`a, b T, c func(x U) V` → [T, T, func(U)V]; `T, U` (unnamed) → [T, U]"""
    plist = " ".join(plist.split()).rstrip(",").strip()
    if not plist:
        return []
    items = T.split_top(plist)
    named = T.parse_go_params(plist)
    # unnamed lists: every item is a type (an item like `x T` has a space outside brackets; `[]*p.T`, `func(A) B` need care)
    def is_named(it):
        parts = it.split(None, 1)
        return len(parts) == 2 and re.fullmatch(r"\w+", parts[0]) is not None and parts[0] not in ("func", "chan", "map", "struct", "interface")
    if any(is_named(it) for it in items) and len(named) == len(items):
        return [norm_type(t) for _, t in named]
    return [norm_type(it) for it in items]


def result_types(result: str) -> list[str]:
    result = result.strip()
    if not result:
        return []
    if result.startswith("(") and T.matching(result, 0) == len(result) - 1:
        return type_list(result[1:-1])
    return [norm_type(result)]


def n_results(result: str) -> int:
    result = result.strip()
    if not result:
        return 0
    if result.startswith("("):
        return len(T.split_top(result[1:T.matching(result, 0)]))
    return 1


def first_result_type(result: str) -> str | None:
    result = result.strip()
    if not result:
        return None
    if result.startswith("("):
        first = T.split_top(result[1:T.matching(result, 0)])[0]
        parts = first.split(None, 1)
        return parts[1] if len(parts) == 2 and re.fullmatch(r"\w+", parts[0]) and not parts[0][0].isupper() and parts[0] not in ("bool", "error", "int", "uint64", "string") else first
    return result


class Package:
    def __init__(self, name: str):
        self.name = name
        self.funcs: dict[str, tuple] = {}                     # name → (nparams, variadic, nresults, first result type)
        self.methods: dict[tuple[str, str], tuple] = {}       # (type, name) → same
        self.interfaces: dict[str, dict[str, tuple]] = {}     # type → {method: (…)}; "" keys = embedded type names
        self.fields: dict[str, dict[str, str]] = {}           # struct → {field: type}
        self.embedded: dict[str, list[str]] = {}              # struct / interface → embedded type names
        self.aliases: dict[str, str] = {}                     # import alias → package name, per file merged (last wins)
        self.types: set[str] = set()                          # every declared type name

    def add_source(self, code: str):
        for m in re.finditer(r'^\s*(?:(\w+)\s+)?"([^"]+)"\s*$', code, re.M):
            pkg = IMPORT_TO_PKG.get(m.group(2))
            if pkg:
                self.aliases[m.group(1) or pkg] = pkg
        self.types |= set(re.findall(r"^type\s+(\w+)\s", code, re.M))
        for blk in re.finditer(r"^type\s*\(", code, re.M):
            self.types |= set(re.findall(r"^\s*(\w+)\s+\S", code[blk.end():T.matching(code, blk.end() - 1)], re.M))
        for name, recv, params, result, body, _ in T.go_functions(code):
            sig = (len(params), any(t.startswith("...") for _, t in params), n_results(result), first_result_type(result),
                   [norm_type(t) for _, t in params], result_types(result))
            if recv:
                self.methods[(recv[1].lstrip("*"), name)] = sig
            else:
                self.funcs[name] = sig
        for m in re.finditer(r"^type\s+(\w+)\s+interface\s*\{", code, re.M):
            end = T.matching(code, m.end() - 1, "{}")
            meths, emb = {}, []
            ibody, k = code[m.end():end], 0
            while k < len(ibody):                                              # (method declarations may span lines)
                mm = re.compile(r"[ \t]*(\w+)\(").match(ibody, k)
                if mm and (k == 0 or ibody[k - 1] == "\n"):
                    close = T.matching(ibody, mm.end() - 1)
                    plist = " ".join(ibody[mm.end():close].split()).rstrip(",").strip()
                    items = T.split_top(plist) if plist else []
                    params = T.parse_go_params(plist) if plist else []
                    if len(params) != len(items):                              # unnamed parameter types: `F(uint64, []byte) bool`
                        params = [("_", t) for t in items]
                    eol = ibody.find("\n", close)
                    eol = len(ibody) if eol < 0 else eol
                    res = ibody[close + 1:eol].strip()
                    meths[mm.group(1)] = (len(params), any(t.startswith("...") for _, t in params), n_results(res), first_result_type(res),
                                          type_list(plist), result_types(res))
                    k = eol + 1
                    continue
                eol = ibody.find("\n", k)
                eol = len(ibody) if eol < 0 else eol
                line = ibody[k:eol].strip()
                if re.fullmatch(r"[\w\.]+", line):
                    emb.append(line)
                k = eol + 1
            self.interfaces[m.group(1)] = meths
            self.embedded[m.group(1)] = emb
        for m in re.finditer(r"^type\s+(\w+)\s+struct\s*\{", code, re.M):
            end = T.matching(code, m.end() - 1, "{}")
            fields, emb = {}, []
            for part in re.split(r"[;\n]", code[m.end():end]):
                part = part.strip().split("`")[0].strip()
                if not part:
                    continue
                if re.fullmatch(r"\*?[\w\.]+", part):
                    emb.append(part.lstrip("*"))
                    fields[part.lstrip("*").split(".")[-1]] = part
                    continue
                for nme, ty in T.parse_go_params(part.replace("\t", " ")):
                    fields[nme] = ty
            self.fields[m.group(1)] = fields
            self.embedded[m.group(1)] = emb


class World:
    def __init__(self, shim_root: str):
        self.pkgs: dict[str, Package] = {}
        self.files: list[tuple[str, str, str]] = []           # (path, package, code) of the shim files (the ones judged)
        layout = {"core": "core", "messages": "messages", "ibftgpu": "ibftgpu", "hoststore": "hoststore"}
        for d, pkg in layout.items():
            for p in sorted(glob.glob(os.path.join(shim_root, d, "*.go"))):
                code = T.strip_go_comments(open(p).read())
                self.pkg(pkg).add_source(code)
                self.files.append((p, pkg, code))
        self.have_ref = os.path.isdir(REF)
        if self.have_ref:
            for d, pkg in (("core", "core"), ("messages", "messages"), ("messages/proto", "proto")):
                for p in sorted(glob.glob(os.path.join(REF, d, "*.go"))):
                    if p.endswith("_test.go"):
                        continue
                    self.pkg(pkg).add_source(T.strip_go_comments(open(p, errors="ignore").read()))

    def pkg(self, name: str) -> Package:
        return self.pkgs.setdefault(name, Package(name))

    def resolve_type(self, t: str | None, here: str) -> tuple[str, str] | None:
        """`*ibftgpu.Ctx` seen from package `here` → ("ibftgpu", "Ctx"); None for builtins / slices / maps / unknown packages"""
        if not t:
            return None
        t = t.strip().lstrip("*").strip()
        if not re.fullmatch(r"[\w\.]+", t):
            return None
        if "." in t:
            q, n = t.split(".", 1)
            pkg = self.pkg(here).aliases.get(q, q)
            return (pkg, n) if pkg in self.pkgs else None
        return (here, t) if (t in self.pkg(here).fields or t in self.pkg(here).interfaces
                             or any(k[0] == t for k in self.pkg(here).methods)) else None

    def method(self, ty: tuple[str, str], name: str, depth: int = 0):
        pkg, tn = ty
        P = self.pkgs.get(pkg)
        if P is None or depth > 4:
            return None
        if (tn, name) in P.methods:
            return P.methods[(tn, name)]
        if tn in P.interfaces and name in P.interfaces[tn]:
            return P.interfaces[tn][name]
        for e in P.embedded.get(tn, []):
            et = self.resolve_type(e, pkg)
            if et:
                got = self.method(et, name, depth + 1)
                if got:
                    return got
        return None

    def field(self, ty: tuple[str, str], name: str, depth: int = 0) -> tuple[str, str] | None:
        pkg, tn = ty
        P = self.pkgs.get(pkg)
        if P is None or depth > 4:
            return None
        f = P.fields.get(tn, {})
        if name in f:
            return self.resolve_type(f[name], pkg)
        for e in P.embedded.get(tn, []):
            et = self.resolve_type(e, pkg)
            if et:
                got = self.field(et, name, depth + 1)
                if got:
                    return got
        return None

    def known_type(self, ty: tuple[str, str]) -> bool:
        P = self.pkgs.get(ty[0])
        return bool(P) and (ty[1] in P.fields or ty[1] in P.interfaces or any(k[0] == ty[1] for k in P.methods))


BUILTIN_SHAPED = re.compile(r"^(\[\]|\*)*(byte|uint8|uint16|uint32|uint64|int|int32|int64|bool|string|uintptr|float64|error)$")
typed_args = 0


def fits(nargs: int, sig) -> bool:
    np_, variadic = sig[0], sig[1]
    return nargs >= np_ - 1 if variadic else nargs == np_


def check(shim_root: str, verbose: bool = False) -> tuple[list[str], int, int]:
    global typed_args
    W = World(shim_root)
    errors, judged, unresolved = [], 0, 0
    typed_args = 0
    for path, pkg, code in W.files:
        rel = os.path.relpath(path, ROOT)
        P = W.pkg(pkg)
        for fn in T.go_functions(code):
            name, recv, params, result, body, body_pos = fn
            env: dict[str, tuple[str, str]] = {}
            plain: dict[str, str] = {}                                         # names whose declared type is made of builtins only
            for n_, t_ in params + (T.parse_go_params(result.strip()[1:-1]) if result.strip().startswith("(") else []):
                if BUILTIN_SHAPED.match(norm_type(t_)):
                    plain[n_] = norm_type(t_).replace("uint8", "byte")
            for m_ in re.finditer(r"\bvar\s+(\w+)\s+([\[\]\*\w]+)", body):
                if BUILTIN_SHAPED.match(m_.group(2)):
                    plain[m_.group(1)] = m_.group(2).replace("uint8", "byte")
            for m_ in re.finditer(r"\b(\w+)\s*:=\s*make\(\s*([\[\]\*\w]+)\s*,", body):
                if BUILTIN_SHAPED.match(m_.group(2)):
                    plain[m_.group(1)] = m_.group(2).replace("uint8", "byte")

            def bind(n, t):
                rt = W.resolve_type(t, pkg)
                if rt:
                    env[n] = rt
            if recv:
                bind(recv[0], recv[1])
            for n, t in params:
                bind(n, t)
            if result.strip().startswith("("):
                for n, t in T.parse_go_params(result.strip()[1:-1]):
                    bind(n, t)
            for m in re.finditer(r"\bvar\s+(\w+)\s+(\*?[\w\.]+)", body):
                bind(m.group(1), m.group(2))
            for m in re.finditer(r"\b(\w+)(?:\s*,\s*\w+)?\s*:=\s*&?([\w\.]+)\{", body):
                bind(m.group(1), m.group(2))
            for m in re.finditer(r"\b(\w+)(?:\s*,\s*\w+)?\s*:=[^\n]*?\.\(\s*(\*?[\w\.]+)\s*\)\s*(?:;[^\n]*)?(?:\{\s*)?$", body, re.M):
                bind(m.group(1), m.group(2))                                   # x[, ok] := <expr>.(T)

            def type_of_chain(chain: list[str]) -> tuple[str, str] | None:
                if not chain or chain[0] not in env:
                    return None
                ty = env[chain[0]]
                for f in chain[1:]:
                    ty = W.field(ty, f)
                    if ty is None:
                        return None
                return ty

            def signature_of_call(qual: list[str], fname: str):
                """the definition `qual.fname(` binds to: ("sig", …) | ("unknown",) | ("unresolved",)"""
                if not qual:
                    return ("sig", P.funcs[fname]) if fname in P.funcs else ("unresolved",)
                if len(qual) == 1 and qual[0] not in env:
                    target = P.aliases.get(qual[0])
                    if target and target in W.pkgs:
                        TP = W.pkgs[target]
                        if fname in TP.funcs:
                            return ("sig", TP.funcs[fname])
                        if fname in TP.fields or fname in TP.interfaces:      # a conversion / composite type, not a call
                            return ("unresolved",)
                        return ("unknown",) if (target != "proto" and (W.have_ref or target in ("ibftgpu", "hoststore"))) else ("unresolved",)
                    return ("unresolved",)
                ty = type_of_chain(qual)
                if ty is None or not W.known_type(ty):
                    return ("unresolved",)
                sig = W.method(ty, fname)
                if sig:
                    return ("sig", sig)
                if W.field(ty, fname) is not None or fname in W.pkgs[ty[0]].fields.get(ty[1], {}):
                    return ("unresolved",)                                     # a func-typed field
                def all_of_it(t, depth=0):       # is every method of t in sight?  (reference packages only with the reference)
                    if t is None or not W.known_type(t) or depth > 4 or not (W.have_ref or t[0] in ("ibftgpu", "hoststore")):
                        return False
                    return all(all_of_it(W.resolve_type(e, t[0]), depth + 1) for e in W.pkgs[t[0]].embedded.get(t[1], []))
                complete = all_of_it(ty)
                return ("unknown", ty) if complete else ("unresolved",)

            # first results bind further names: x[, err] := <call>
            for m in re.finditer(r"\b(\w+)(?:\s*,\s*[\w_]+)*\s*:=\s*((?:\w+\.)*)(\w+)\(", body):
                got = signature_of_call([q for q in m.group(2).split(".") if q], m.group(3))
                if got[0] == "sig" and got[1][3]:
                    owner_pkg = pkg
                    q = [x for x in m.group(2).split(".") if x]
                    if len(q) == 1 and q[0] not in env and P.aliases.get(q[0]) in W.pkgs:
                        owner_pkg = P.aliases[q[0]]
                    elif q:
                        ty = type_of_chain(q)
                        owner_pkg = ty[0] if ty else pkg
                    rt = W.resolve_type(got[1][3], owner_pkg)
                    if rt and m.group(1) not in env:
                        env[m.group(1)] = rt

            for m in re.finditer(r"(?<![\w\.\)\]])((?:\w+\.)*)(\w+)\(", body):
                fname, qual = m.group(2), [q for q in m.group(1).split(".") if q]
                if fname in ("func", "if", "for", "switch", "return", "make", "len", "cap", "append", "new", "copy", "panic", "delete",
                             "string", "int", "uint64", "uint32", "byte", "uint", "int32", "int64", "uintptr", "bool", "float64"):
                    continue
                if qual and qual[0] in ("C", "unsafe", "fmt", "errors", "sync", "runtime", "big", "binary", "bytes", "sort", "time",
                                        "math", "atomic", "context", "cgo", "goproto", "protobuf"):
                    continue
                got = signature_of_call(qual, fname)
                line = T.line_of(code, body_pos + m.start())
                if got[0] == "unresolved":
                    if qual:
                        unresolved += 1
                        if verbose:
                            print("   unresolved: %s:%d %s" % (rel, line, ".".join(qual + [fname])))
                    continue
                if got[0] == "unknown":
                    errors.append("%s:%d: %s is not defined on %s" % (rel, line, ".".join(qual + [fname]),
                                                                      ".".join(got[1]) if len(got) > 1 else "its package"))
                    continue
                sig = got[1]
                end = T.matching(body, m.end() - 1)
                inner = body[m.end():end].strip()
                args = T.split_top(inner) if inner else []
                judged += 1
                spread = len(args) == 1 and re.match(r"^(?:\w+\.)*\w+\(.*\)$", args[0], re.S)      # f(g()) with a multi-value g
                if not fits(len(args), sig) and not (spread and sig[0] > 1):
                    errors.append("%s:%d: %s called with %d arguments, the definition takes %d%s"
                                  % (rel, line, ".".join(qual + [fname]), len(args), sig[0], " (variadic)" if sig[1] else ""))
                # arguments that are plain names of a builtin-shaped declared type, against builtin-shaped parameter types
                if len(args) == len(sig[4]) and not sig[1]:
                    for k_, (a_, pt_) in enumerate(zip(args, sig[4])):
                        pt_ = pt_.replace("uint8", "byte")
                        if a_ in plain and BUILTIN_SHAPED.match(pt_):
                            typed_args += 1
                            if plain[a_] != pt_:
                                errors.append("%s:%d: %s argument %d: %s is %s, the parameter is %s"
                                              % (rel, line, ".".join(qual + [fname]), k_ + 1, a_, plain[a_], pt_))
                # the call as the whole right-hand side of an assignment
                ls = body.rfind("\n", 0, m.start()) + 1
                prefix, rest = body[ls:m.start()], body[end + 1:body.find("\n", end) if body.find("\n", end) > 0 else len(body)]
                mm = re.match(r"^\s*(?:\}\s*else\s+)?(?:if\s+|switch\s+)?([\w\.\[\]\*_, ]+?)\s*:?=\s*$", prefix)
                if mm and re.match(r"^\s*(;.*)?(\{\s*)?$", rest):
                    nl = len(T.split_top(mm.group(1)))
                    if nl != sig[2] and not (nl == 1 and sig[2] == 1):
                        errors.append("%s:%d: %d names receive the %d results of %s" % (rel, line, nl, sig[2], ".".join(qual + [fname])))
    return errors, judged, unresolved


# what the overlay promises to be assignable: T implements I  ⇔  `var _ I = (*T)(nil)` compiles
IMPLEMENTS = [(("hoststore", "Store"), ("core", "Messages")), (("hoststore", "Store"), ("core", "hostStore")),
              (("messages", "Messages"), ("core", "batchStore")), (("messages", "Messages"), ("core", "Messages"))]


def interface_methods(W: World, it: tuple[str, str], depth: int = 0) -> dict[str, tuple]:
    P = W.pkgs.get(it[0])
    if P is None or it[1] not in P.interfaces or depth > 4:
        return {}
    out = dict(P.interfaces[it[1]])
    for e in P.embedded.get(it[1], []):
        et = W.resolve_type(e, it[0])
        if et:
            out.update(interface_methods(W, et, depth + 1))
    return out


def qualify(types: list[str], pkg: str, W: World) -> list[str]:
    """type names written inside package `pkg` made absolute, so that `Messages` in package messages = `messages.Messages`"""
    P = W.pkg(pkg)
    local = set(P.fields) | set(P.interfaces) | {k[0] for k in P.methods} | P.types
    out = []
    for t in types:
        t = re.sub(r"(?<![\w\.])(\w+)\.(\w+)", lambda m: P.aliases.get(m.group(1), m.group(1)) + "." + m.group(2), t)
        t = re.sub(r"(?<![\w\.])([A-Z]\w*)(?![\w\.])", lambda m: pkg + "." + m.group(1) if m.group(1) in local else m.group(1), t)
        out.append(t)
    return out


def check_implements(shim_root: str) -> tuple[list[str], int]:
    W = World(shim_root)
    errors, compared = [], 0
    for ty, it in IMPLEMENTS:
        want = interface_methods(W, it)
        if not want or not W.known_type(ty):
            continue                                                           # (the reference is absent: nothing to compare with)
        for name, isig in sorted(want.items()):
            msig = W.method(ty, name)
            if msig is None:
                if W.have_ref or ty[0] in ("ibftgpu", "hoststore") and not W.pkgs[ty[0]].embedded.get(ty[1]):
                    errors.append("*%s.%s lacks %s of %s.%s" % (ty[0], ty[1], name, it[0], it[1]))
                continue
            # where the method was found decides how its unqualified names read
            owner = ty[0]
            if (ty[1], name) not in W.pkgs[ty[0]].methods:
                for e in W.pkgs[ty[0]].embedded.get(ty[1], []):
                    et = W.resolve_type(e, ty[0])
                    if et and W.method(et, name):
                        owner = et[0]
            got = (qualify(msig[4], owner, W), qualify(msig[5], owner, W))
            exp = (qualify(isig[4], it[0], W), qualify(isig[5], it[0], W))
            compared += 1
            if got != exp:
                errors.append("*%s.%s.%s has the signature (%s) (%s), %s.%s wants (%s) (%s)"
                              % (ty[0], ty[1], name, ", ".join(got[0]), ", ".join(got[1]), it[0], it[1], ", ".join(exp[0]), ", ".join(exp[1])))
    return errors, compared


def main(argv: list[str]) -> int:
    verbose = "-v" in argv
    argv = [a for a in argv if a != "-v"]
    shim_root = argv[0] if argv else os.path.join(ROOT, "shim", "go")
    errors, judged, unresolved = check(shim_root, verbose)
    ierrors, compared = check_implements(shim_root)
    errors += ierrors
    if not os.path.isdir(REF):
        print("note: /root/reference absent — calls into the reference's own types are not judged")
    for e in errors:
        print("ERROR:", e)
    print("go arity: %d calls judged against their definitions (%d arguments of builtin-shaped types compared with their parameters), "
          "%d with a receiver this tool cannot type; %d method signatures compared with the interfaces their types are assigned to; "
          "%d problems" % (judged, typed_args, unresolved, compared, len(errors)))
    return 1 if errors else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
