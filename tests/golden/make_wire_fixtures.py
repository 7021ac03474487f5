"""Generate tests/golden/wire_vectors.json: protobuf wire bytes of go-ibft messages
produced by the google.protobuf runtime from a descriptor that restates
/root/reference/messages/proto/messages.proto:7-110 (no protoc in this image, so the
FileDescriptorProto is built programmatically).  These pin oracle/wire.py and the C++
host encoder (go-ibft_amd/host/proto.cpp) — including PayloadNoSig
(/root/reference/messages/proto/helper.go:12-27 = serialize with `signature` cleared).
Run from the repo root:  python tests/golden/make_wire_fixtures.py
"""
import json
import os
import sys

from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

HERE = os.path.dirname(os.path.abspath(__file__))
F = descriptor_pb2.FieldDescriptorProto


def build_messages():
    fd = descriptor_pb2.FileDescriptorProto(name="ibft_messages.proto", package="ibftgold", syntax="proto3")
    en = fd.enum_type.add(name="MessageType")
    for i, n in enumerate(["PREPREPARE", "PREPARE", "COMMIT", "ROUND_CHANGE"]):
        en.value.add(name=n, number=i)

    def msg(name, fields, oneofs=()):
        m = fd.message_type.add(name=name)
        for o in oneofs:
            m.oneof_decl.add(name=o)
        for (fname, num, ftype, tname, label, oneof) in fields:
            f = m.field.add(name=fname, number=num, type=ftype, label=label)
            if tname:
                f.type_name = ".ibftgold." + tname
            if oneof is not None:
                f.oneof_index = oneof
        return m

    OPT, REP = F.LABEL_OPTIONAL, F.LABEL_REPEATED
    msg("View", [("height", 1, F.TYPE_UINT64, None, OPT, None), ("round", 2, F.TYPE_UINT64, None, OPT, None)])
    msg("IbftMessage", [
        ("view", 1, F.TYPE_MESSAGE, "View", OPT, None), ("from", 2, F.TYPE_BYTES, None, OPT, None),
        ("signature", 3, F.TYPE_BYTES, None, OPT, None), ("type", 4, F.TYPE_ENUM, "MessageType", OPT, None),
        ("preprepareData", 5, F.TYPE_MESSAGE, "PrePrepareMessage", OPT, 0),
        ("prepareData", 6, F.TYPE_MESSAGE, "PrepareMessage", OPT, 0),
        ("commitData", 7, F.TYPE_MESSAGE, "CommitMessage", OPT, 0),
        ("roundChangeData", 8, F.TYPE_MESSAGE, "RoundChangeMessage", OPT, 0)], oneofs=["payload"])
    msg("PrePrepareMessage", [("proposal", 1, F.TYPE_MESSAGE, "Proposal", OPT, None),
                              ("proposalHash", 2, F.TYPE_BYTES, None, OPT, None),
                              ("certificate", 3, F.TYPE_MESSAGE, "RoundChangeCertificate", OPT, None)])
    msg("PrepareMessage", [("proposalHash", 1, F.TYPE_BYTES, None, OPT, None)])
    msg("CommitMessage", [("proposalHash", 1, F.TYPE_BYTES, None, OPT, None),
                          ("committedSeal", 2, F.TYPE_BYTES, None, OPT, None)])
    msg("RoundChangeMessage", [("lastPreparedProposal", 1, F.TYPE_MESSAGE, "Proposal", OPT, None),
                               ("latestPreparedCertificate", 2, F.TYPE_MESSAGE, "PreparedCertificate", OPT, None)])
    msg("PreparedCertificate", [("proposalMessage", 1, F.TYPE_MESSAGE, "IbftMessage", OPT, None),
                                ("prepareMessages", 2, F.TYPE_MESSAGE, "IbftMessage", REP, None)])
    msg("RoundChangeCertificate", [("roundChangeMessages", 1, F.TYPE_MESSAGE, "IbftMessage", REP, None)])
    msg("Proposal", [("rawProposal", 1, F.TYPE_BYTES, None, OPT, None), ("round", 2, F.TYPE_UINT64, None, OPT, None)])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    names = ["View", "IbftMessage", "PrePrepareMessage", "PrepareMessage", "CommitMessage", "RoundChangeMessage",
             "PreparedCertificate", "RoundChangeCertificate", "Proposal"]
    return {n: message_factory.GetMessageClass(pool.FindMessageTypeByName("ibftgold." + n)) for n in names}


def main():
    M = build_messages()
    Msg = M["IbftMessage"]
    vectors = []

    def add(name, m):
        full = m.SerializeToString(deterministic=True)
        ns = Msg()
        ns.CopyFrom(m)
        ns.ClearField("signature")
        vectors.append({"name": name, "wire": full.hex(), "payload_no_sig": ns.SerializeToString(deterministic=True).hex()})

    addr = bytes(range(1, 21))
    h32 = bytes(range(32, 64))
    seal = bytes(range(100, 165))
    sig = bytes(range(7, 72))

    def prepare(h, r, frm=addr, hsh=h32, s=sig):
        m = Msg(type=1, signature=s)
        setattr(m, "from", frm)
        m.view.height = h
        m.view.round = r
        m.prepareData.proposalHash = hsh
        return m

    def commit(h, r, frm=addr, hsh=h32, sl=seal, s=sig):
        m = Msg(type=2, signature=s)
        setattr(m, "from", frm)
        m.view.height = h
        m.view.round = r
        m.commitData.proposalHash = hsh
        m.commitData.committedSeal = sl
        return m

    add("prepare_h1_r0", prepare(1, 0))
    add("prepare_h300_r2", prepare(300, 2))
    add("prepare_h0_r0_empty_view", prepare(0, 0))
    add("commit_h1_r0", commit(1, 0))
    add("commit_big_height", commit(2**63 + 5, 2**32))
    add("commit_empty_fields", commit(1, 0, hsh=b"", sl=b""))
    m = Msg(type=2)
    setattr(m, "from", addr)
    add("commit_type_nil_payload_nil_view", m)
    m = Msg(type=0, signature=sig)
    setattr(m, "from", addr)
    m.view.height = 5
    m.preprepareData.proposal.rawProposal = b"block bytes" * 7
    m.preprepareData.proposal.round = 0
    m.preprepareData.proposalHash = h32
    add("preprepare_r0", m)
    # round-change carrying a prepared certificate, nested in a preprepare certificate
    pp = Msg(type=0, signature=sig)
    setattr(pp, "from", addr)
    pp.view.height = 9
    pp.view.round = 1
    pp.preprepareData.proposal.rawProposal = b"raw"
    pp.preprepareData.proposal.round = 1
    pp.preprepareData.proposalHash = h32
    rc = Msg(type=3, signature=sig)
    setattr(rc, "from", bytes(range(50, 70)))
    rc.view.height = 9
    rc.view.round = 2
    rc.roundChangeData.lastPreparedProposal.rawProposal = b"raw"
    rc.roundChangeData.lastPreparedProposal.round = 1
    rc.roundChangeData.latestPreparedCertificate.proposalMessage.CopyFrom(pp)
    for i in range(3):
        rc.roundChangeData.latestPreparedCertificate.prepareMessages.append(prepare(9, 1, frm=bytes([i + 1]) * 20))
    add("round_change_with_pc", rc)
    rc0 = Msg(type=3, signature=sig)
    setattr(rc0, "from", addr)
    rc0.view.height = 9
    rc0.view.round = 2
    rc0.roundChangeData.SetInParent()
    add("round_change_empty_body", rc0)
    np_ = Msg(type=0, signature=sig)
    setattr(np_, "from", addr)
    np_.view.height = 9
    np_.view.round = 2
    np_.preprepareData.proposal.rawProposal = b"raw"
    np_.preprepareData.proposal.round = 2
    np_.preprepareData.proposalHash = h32
    np_.preprepareData.certificate.roundChangeMessages.append(rc)
    np_.preprepareData.certificate.roundChangeMessages.append(rc0)
    add("preprepare_with_rcc", np_)
    json.dump(vectors, open(os.path.join(HERE, "wire_vectors.json"), "w"), indent=1)
    print("wrote", len(vectors), "vectors")


if __name__ == "__main__":
    main()
