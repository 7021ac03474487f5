#!/usr/bin/env python3
"""CI-less sanity check of shim/go (no Go toolchain in this image): every identifier the shim uses must be
DEFINED somewhere a Go compiler would find it —

  i.<method>(            method or field of core.IBFT: shim/go/core or /root/reference/core
  i.state.<m>( / i.messages.<m>( / i.backend.<m>(   the reference's state / Messages interface / Backend
  ibftgpu.<Symbol>       exported by shim/go/ibftgpu
  messages.<Func>, proto.<Type>                     shim/go/messages or the reference packages
  C.<name>               declared in include/ibftgpu.h / include/ibft_host.h / the file's own cgo preamble, and every
                         C.<function>(...) call passes as many arguments as the prototype has parameters
  hoststore              *Store has every method of core.Messages and of core.hostStore; the "C call sequence" each method
                         documents is the one its body makes (tests/test_hoststore_sequence.py replays those sequences)
  import paths           the overlay layout documented in shim/go/ibftgpu/ibftgpu.go
  compile errors         an import nothing uses, a local variable declared and never read

Exit code 0 = consistent.  The reference checks are skipped (with a note) when /root/reference is absent."""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def read(paths):
    return "\n".join(open(p, errors="ignore").read() for p in paths)


def _split_args(text: str) -> int:
    """number of top-level comma-separated items in an argument / parameter list"""
    text = text.strip()
    if not text or text == "void":
        return 0
    depth, n = 0, 1
    for ch in text:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        elif ch == "," and depth == 0:
            n += 1
    return n


def _matching(text: str, open_at: int) -> int:
    depth = 0
    for k in range(open_at, len(text)):
        if text[k] == "(":
            depth += 1
        elif text[k] == ")":
            depth -= 1
            if depth == 0:
                return k
    return -1


def c_prototypes(header: str) -> dict:
    """function name → parameter count, for every prototype / inline definition in C text"""
    text = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    text = "\n".join(l.split("//")[0] for l in text.splitlines() if not l.lstrip().startswith("#"))
    out = {}
    for m in re.finditer(r"\b(\w+)\s*\(", text):
        name = m.group(1)
        if name in ("if", "for", "while", "return", "sizeof", "switch") or text[max(0, m.start() - 2):m.start()].endswith("(*"):
            continue
        before = text[:m.start()].rstrip()
        if not before or before[-1] in "=(,!&|+-*/<>?:" and not before.endswith("*"):   # a call, not a declaration
            continue
        end = _matching(text, m.end() - 1)
        if end < 0:
            continue
        after = text[end + 1:end + 40].lstrip()
        if not (after.startswith(";") or after.startswith("{")):
            continue
        if re.search(r"typedef[^;]*$", before):   # function-pointer typedef
            continue
        out.setdefault(name, _split_args(text[m.end():end]))
    return out


def go_c_calls(code: str):
    """(name, argument count) of every C.<name>(...) call in Go text (type conversions included: one argument)"""
    out = []
    for m in re.finditer(r"\bC\.(\w+)\(", code):
        end = _matching(code, m.end() - 1)
        if end > 0:
            out.append((m.group(1), _split_args(code[m.end():end])))
    return out


def go_functions_with_doc(src: str):
    """(name, doc comment, body) of every top-level func of a Go file"""
    out = []
    for m in re.finditer(r"((?:^//[^\n]*\n)*)^func (?:\([^)]*\) )?(\w+)\(", src, re.M):
        brace = src.find("{", _matching(src, m.end() - 1))   # (the result list may be parenthesised too: the body's brace follows it)
        line_end = src.find("\n", brace)
        if src.count("(", _matching(src, m.end() - 1) + 1, brace) != src.count(")", _matching(src, m.end() - 1) + 1, brace):
            brace = src.find("{", line_end)
        depth, k = 0, brace
        while k < len(src):
            if src[k] == "{":
                depth += 1
            elif src[k] == "}":
                depth -= 1
                if depth == 0:
                    break
            k += 1
        doc = "\n".join(l[2:].strip() for l in m.group(1).splitlines())
        out.append((m.group(2), doc, src[brace:k + 1]))
    return out


def main(shim_root: str | None = None) -> int:
    shim_root = shim_root or os.path.join(ROOT, "shim", "go")

    def code(text):  # comments say things like "proto.Unmarshal": only code is checked
        text = re.sub(r"/\*.*?\*/", lambda m: m.group(0) if "#include" in m.group(0) else "", text, flags=re.S)
        return "\n".join(l.split("//")[0] for l in text.splitlines())
    shim = {p: code(open(p).read()) for p in glob.glob(os.path.join(shim_root, "**", "*.go"), recursive=True)}
    all_shim = "\n".join(shim.values())
    have_ref = os.path.isdir(REF)
    ref_core = read(glob.glob(os.path.join(REF, "core", "*.go"))) if have_ref else ""
    ref_msgs = read(glob.glob(os.path.join(REF, "messages", "*.go"))) if have_ref else ""
    ref_proto = read(glob.glob(os.path.join(REF, "messages", "proto", "*.go"))) if have_ref else ""
    hdr = open(os.path.join(ROOT, "include", "ibftgpu.h")).read()
    errors = []

    def defined_method(recv_pat, name, *texts):
        pat = re.compile(r"func \(\w+ \*?%s\) %s\(" % (recv_pat, re.escape(name)))
        return any(pat.search(t) for t in texts)

    core_shim = "\n".join(v for p, v in shim.items() if os.sep + "core" + os.sep in p)
    msgs_shim = "\n".join(v for p, v in shim.items() if os.sep + "messages" + os.sep in p)
    gpu_shim = "\n".join(v for p, v in shim.items() if os.sep + "ibftgpu" + os.sep in p)

    # methods called on the IBFT receiver
    for name in sorted(set(re.findall(r"\bi\.(\w+)\(", core_shim))):
        if not defined_method("IBFT", name, core_shim, ref_core) and have_ref:
            errors.append(f"core: i.{name}() is defined neither in shim/go/core nor in the reference")
    # fields of IBFT used with a method call
    for field, name in sorted(set(re.findall(r"\bi\.(state|messages|backend)\.(\w+)\(", core_shim))):
        if not have_ref:
            continue
        if field == "state" and not defined_method("state", name, ref_core):
            errors.append(f"core: i.state.{name}() not in the reference's state")
        if field == "messages" and not (re.search(r"\b%s\(" % name, ref_core.split("type Messages interface")[1].split("}")[0])
                                        or defined_method("Messages", name, msgs_shim)):
            errors.append(f"core: i.messages.{name}() not in core.Messages nor in shim/go/messages")
        if field == "backend" and not re.search(r"\b%s\(" % name, ref_core):
            errors.append(f"core: i.backend.{name}() not in the reference's Backend interfaces")
    # package-qualified identifiers
    for name in sorted(set(re.findall(r"(?<![\w.])ibftgpu\.(\w+)", all_shim))):
        if not re.search(r"(func|type|var|const)\s+(\([^)]*\)\s*)?%s\b|^\s+%s\s*=|^\s+%s\b" % (name, name, name), gpu_shim, re.M):
            errors.append(f"ibftgpu.{name} is not exported by shim/go/ibftgpu")
    for name in sorted(set(re.findall(r"(?<![\w.])messages\.(\w+)", core_shim))):
        if have_ref and not re.search(r"func %s\(|type %s\b" % (name, name), ref_msgs + msgs_shim):
            errors.append(f"messages.{name} not found")
    for name in sorted(set(re.findall(r"(?<![\w.])proto\.(\w+)", all_shim))):
        if have_ref and not re.search(r"\b%s\b" % name, ref_proto):
            errors.append(f"proto.{name} not found in messages/proto")
    # C symbols: every C.<name> of a cgo file must be declared in the headers its preamble includes (or in the preamble
    # itself), and every C.<function>(...) call must pass as many arguments as the prototype has parameters — the part of a
    # type check that can be done without a compiler
    host_hdr = open(os.path.join(ROOT, "include", "ibft_host.h")).read()
    protos = c_prototypes(hdr + "\n" + host_hdr)
    builtin = {"int", "size_t", "uint8_t", "uint32_t", "uint64_t", "int32_t", "uintptr_t", "GoString", "free", "char", "double"}
    for path, text in shim.items():
        if 'import "C"' not in text:
            continue
        raw = open(path).read()
        preamble = raw[raw.index("/*"):raw.index('import "C"')] if "/*" in raw.split('import "C"')[0] else ""
        visible = hdr + (host_hdr if "ibft_host.h" in preamble else "") + preamble
        local = c_prototypes(preamble)
        rel = os.path.relpath(path, ROOT)
        for name in sorted(set(re.findall(r"\bC\.(\w+)", text))):
            if name in builtin:
                continue
            if not re.search(r"\b%s\b" % name, visible):
                errors.append(f"{rel}: C.{name} is not declared in the headers / preamble of this file")
        for name, argc in go_c_calls(text):
            want = local.get(name, protos.get(name))
            if want is not None and want != argc:
                errors.append(f"{rel}: C.{name} called with {argc} arguments, the prototype has {want}")
    # the optional interfaces of package core that another package of the overlay implements
    hs = "\n".join(v for p_, v in shim.items() if os.sep + "hoststore" + os.sep in p_)
    m_if = re.search(r"type hostStore interface \{(.*?)\n\}", core_shim, re.S)
    if m_if and hs:
        for name in re.findall(r"^\s*(\w+)\(", m_if.group(1), re.M):
            if not defined_method("Store", name, hs):
                errors.append(f"core.hostStore.{name} is not a method of *hoststore.Store")
    for name in ("AddMessage", "PruneByHeight", "GetValidMessages", "GetExtendedRCC", "GetMostRoundChangeMessages"):
        if hs and not defined_method("Store", name, hs):   # core.Messages (core/ibft.go:23-46); the subscription half is embedded
            errors.append(f"*hoststore.Store lacks {name} of core.Messages")
    if hs and "*messages.Messages" not in hs:
        errors.append("*hoststore.Store does not embed *messages.Messages (Subscribe / Unsubscribe / SignalEvent)")
    # the documented C call sequence of every method is the one its body makes (tests/test_hoststore_sequence.py replays it)
    hs_path = os.path.join(shim_root, "hoststore", "hoststore.go")
    hs_src = open(hs_path).read() if os.path.exists(hs_path) else ""
    for fn, doc, body in go_functions_with_doc(hs_src):
        m_seq = re.search(r"C call sequence:\s*(.*?)\.\s*$", doc, re.S)
        made = [n for n, _ in go_c_calls(body) if n.startswith("ibft_host_")]
        for helper, _, hbody in go_functions_with_doc(hs_src):   # one level of helpers of the same file
            if helper != fn and re.search(r"\b%s\(" % helper, body):
                made += [n for n, _ in go_c_calls(hbody) if n.startswith("ibft_host_")]
        if m_seq is None:
            if made:
                errors.append(f"hoststore.{fn}: calls {made} but documents no C call sequence")
            continue
        documented = re.findall(r"ibft_host_\w+", m_seq.group(1))   # ("… (on failure: x, y)" counts too)
        if sorted(set(documented)) != sorted(set(made)):
            errors.append(f"hoststore.{fn}: documented sequence {documented} != calls made {made}")
    # methods on shim types used by the shim itself (ctx.X / g.X)
    for name in sorted(set(re.findall(r"\(\*Ctx\)\.(\w+)", all_shim))):
        if not defined_method("Ctx", name, gpu_shim):
            errors.append(f"(*Ctx).{name} undefined")
    # import paths follow the overlay layout
    for imp in sorted(set(re.findall(r'"(github\.com/0xPolygon/go-ibft/[^"]+)"', all_shim))):
        if imp.split("go-ibft/")[1] not in ("messages", "messages/proto", "ibftgpu", "core"):
            errors.append(f"import path {imp} does not exist in the overlay layout")
    # every exported binding the docs mention exists
    for doc in ("INTEGRATION.md", "DESIGN.md"):
        text = open(os.path.join(ROOT, doc)).read()
        for name in sorted(set(re.findall(r"\bctx\.(\w+)\(", text)) | set(re.findall(r"\(\*Ctx\)\.(\w+)", text))):
            if not defined_method("Ctx", name, gpu_shim):
                errors.append(f"{doc} mentions ctx.{name}() which shim/go/ibftgpu does not define")
    # what the Go compiler refuses outright: an import nothing uses, a local variable declared and never read
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import cgo_typecheck as CT
    for path in sorted(shim):
        rel = os.path.relpath(path, ROOT)
        text = CT.strip_go_comments(open(path).read())
        nostr = re.sub(r'`[^`]*`', '``', re.sub(r'"(?:[^"\\\n]|\\.)*"', '""', text))
        imports = []
        for m in re.finditer(r'^import\s*\((.*?)^\)', text, re.M | re.S):
            imports += re.findall(r'^\s*(?:(\w+|_|\.)\s+)?"([^"]+)"', m.group(1), re.M)
        imports += re.findall(r'^import\s+(?:(\w+)\s+)?"([^"]+)"\s*$', text, re.M)
        for alias, ipath in imports:
            name = alias or ipath.split("/")[-1]
            if ipath == "C" or alias in ("_", "."):
                continue
            if not re.search(r"(?<![\w\.])%s\." % re.escape(name), nostr):
                errors.append(f"{rel}: \"{ipath}\" imported and not used")
        for fname, _, _, _, body, _ in CT.go_functions(text):
            b = re.sub(r'"(?:[^"\\\n]|\\.)*"', '""', body)
            declared = set()
            for m in re.finditer(r"(?:^|[;{]|\bif\s|\bfor\s|\bswitch\s)\s*([\w, ]+?)\s*:=", b, re.M):
                declared |= {n.strip() for n in m.group(1).split(",") if re.fullmatch(r"[A-Za-z]\w*", n.strip())}
            for m in re.finditer(r"\bvar\s+([\w, ]+?)\s+[\*\[\w]", b):
                declared |= {n.strip() for n in m.group(1).split(",") if re.fullmatch(r"[A-Za-z]\w*", n.strip())}
            for n in sorted(declared):
                if len(re.findall(r"(?<![\w\.])%s\b" % re.escape(n), b)) < 2:
                    errors.append(f"{rel}: {fname}: {n} declared and not used")
    if not have_ref:
        print("note: /root/reference absent — only the intra-repo checks ran")
    for e in errors:
        print("ERROR:", e)
    print(f"shim/go: {len(shim)} files, {len(errors)} problems")
    return 1 if errors else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else None))
