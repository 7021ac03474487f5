#!/usr/bin/env python3
"""CI-less sanity check of shim/go (no Go toolchain in this image): every identifier the shim uses must be
DEFINED somewhere a Go compiler would find it —

  i.<method>(            method or field of core.IBFT: shim/go/core or /root/reference/core
  i.state.<m>( / i.messages.<m>( / i.backend.<m>(   the reference's state / Messages interface / Backend
  ibftgpu.<Symbol>       exported by shim/go/ibftgpu
  messages.<Func>, proto.<Type>                     shim/go/messages or the reference packages
  C.<name>               declared in include/ibftgpu.h
  import paths           the overlay layout documented in shim/go/ibftgpu/ibftgpu.go

Exit code 0 = consistent.  The reference checks are skipped (with a note) when /root/reference is absent."""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def read(paths):
    return "\n".join(open(p, errors="ignore").read() for p in paths)


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
    # C symbols
    cnames = set(re.findall(r"\bC\.(\w+)", gpu_shim))
    for name in sorted(cnames):
        if name in ("int", "size_t", "uint8_t", "uint32_t", "uint64_t", "int32_t", "GoString"):
            continue
        if not re.search(r"\b%s\b" % name, hdr):
            errors.append(f"C.{name} is not declared in include/ibftgpu.h")
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
    if not have_ref:
        print("note: /root/reference absent — only the intra-repo checks ran")
    for e in errors:
        print("ERROR:", e)
    print(f"shim/go: {len(shim)} files, {len(errors)} problems")
    return 1 if errors else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else None))
