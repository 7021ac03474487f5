"""§8f ranks 1 + 2 together, host side, WITHOUT a device: certificates judged on arrival.  IngestWire sends PREPREPARE /
ROUND_CHANGE messages to the batch backend as they arrived (VerifyCertificatesWire — here the loop backend, which decodes
and asks the callback Verifier in the device's row order); the verdicts about the NESTED messages wait in tables keyed by
the stored message objects, and handleRoundChangeMessage / handlePrePrepare (core/ibft.go:470-512, 792-813 → validateProposal
:683-788 → validPC :1162-1231, proposalMatchesCertificate :516-551) must then

  * decide exactly like the per-message reference walk, and
  * need no further batch call — everything they ask was settled when the carrying message arrived.
"""
import pytest

import go_ibft_amd.hostlib as H
from oracle import wire as W
from test_host_roundchange import KINDS, PP, World, fake_hash, rc_message


def build_traffic(w, height, raw, rounds=(1, 2, 3)):
    """ROUND_CHANGE messages (honest and Byzantine certificates) and, per round, PREPREPAREs whose RoundChangeCertificate
    is made of that round's ROUND_CHANGE messages"""
    per_round = {}
    for rnd in rounds:
        senders = list(w.addrs)
        w.rng.shuffle(senders)
        msgs = []
        # round 1: an honest quorum (an RCC exists); round 2: one Byzantine certificate among honest ones; round 3: anything
        count = w.n if rnd < 3 else w.rng.randrange(w.n // 2, w.n + 1)
        for k, a in enumerate(senders[:count]):
            kind = w.rng.choice(KINDS) if rnd == 3 or (rnd == 2 and k == 0) else w.rng.choice([None, None, "no_certificate"])
            cert_round = w.rng.randrange(0, rnd)
            if kind == "no_certificate":
                m = rc_message(w, height, rnd, a)
            elif kind == "proposal_without_certificate":
                m = rc_message(w, height, rnd, a, raw=raw, cert=None, cert_round=cert_round)
            elif kind == "other_proposal":
                m = rc_message(w, height, rnd, a, raw=raw, cert=w.certificate(height, cert_round, b"another"), cert_round=cert_round)
            else:
                # the mock backend marks bad signatures by message bytes, and honest certificates of one (height, round,
                # proposal) share their nested messages: a Byzantine certificate gets a proposal of its own
                if kind in ("short_hash_everywhere", "commit_among_prepares"):
                    w.irregular = True   # shapes whose hash questions are not settled on arrival: the walk asks the backend once more
                r2 = raw if kind is None else raw + b"|" + kind.encode() + a
                m = rc_message(w, height, rnd, a, raw=r2, cert=w.certificate(height, cert_round, r2, kind), cert_round=cert_round)
            msgs.append(m)
        per_round[rnd] = msgs
    proposals = {}
    for rnd in rounds:
        prop = w.proposer(height, rnd)
        proposals[rnd] = W.IbftMessage(view=W.View(height, rnd), sender=prop, type=PP, signature=b"sig-pp-" + prop,
                                       payload=W.preprepare_body(W.Proposal(raw, rnd), fake_hash(raw, rnd),
                                                                 W.round_change_certificate(per_round[rnd])))
    return per_round, proposals


@pytest.mark.parametrize("seed", range(10))
@pytest.mark.parametrize("mode", ["certs", "certs_objects", "certs_fail", "certs_off"])
def test_certificates_on_arrival_equal_stock(seed, mode):
    """certs: a ROUND_CHANGE message's certificate is judged from the backend's rows on arrival and stays undecoded
    (ibft_host_use_rc_rows, the default); certs_objects: decoded, verdicts noted in the nested objects, the walk over them."""
    w = World(n=7 + seed % 4, seed=100 + seed)
    height, raw = 3, b"block-%d" % seed
    stock, fast = w.host(), w.host()
    fast.use_loop_batch(16 if mode == "certs_fail" else 0)
    fast.use_batch(True)
    fast.use_certs(mode != "certs_off")
    fast.use_rc_rows(mode != "certs_objects")
    me = b"someone else"  # this node proposes nothing
    for h in (stock, fast):
        h.set_id(me)
        h.set_state(height, 0, None)
    per_round, proposals = build_traffic(w, height, raw)
    batches = []
    for rnd, msgs in per_round.items():
        wires = [m.encode() for m in msgs]
        batches += [wires[:3], wires[3:] + [proposals[rnd].encode()], wires[:2]]  # the last one: re-deliveries
    for wires in batches:
        if not wires:
            continue
        ra = stock.ingest_wire(wires)[0]
        rb = fast.ingest_wire(wires)[0]
        assert ra == rb, (seed, mode)
    calls, rows, _ = fast.cert_stats()
    if mode in ("certs", "certs_objects"):
        assert calls >= 3 and rows > 20 and fast.loop_batch_cert_calls() == calls
        assert fast.rc_from_rows > 5 if mode == "certs" else fast.rc_from_rows == 0
        assert (fast.pp_from_rows >= 1 or getattr(w, "irregular", False)) if mode == "certs" else fast.pp_from_rows == 0
    else:
        assert calls == 0 and fast.rc_from_rows == 0
    for view_round in (1, 2, 3):
        for h in (stock, fast):
            h.set_state(height, view_round, None)
        before = fast.loop_batch_calls()
        a = stock.handle_round_change(height, view_round)
        b = fast.handle_round_change(height, view_round)
        assert sorted(a) == sorted(b), (seed, mode, view_round)
        assert a or view_round == 3  # the honest rounds do produce an extended RCC
        if mode in ("certs", "certs_objects") and not getattr(w, "irregular", False):
            assert fast.loop_batch_calls() == before, "the certificate walk asked the backend again"
        pa = stock.handle_preprepare(height, view_round)
        pb = fast.handle_preprepare(height, view_round)
        assert pa == pb, (seed, mode, view_round)
        if mode in ("certs", "certs_objects") and not getattr(w, "irregular", False):
            assert fast.loop_batch_calls() == before
            if pb is not None:
                assert fast.cert_stats()[2] > 0  # sender verdicts came from the arrival-time tables
    assert fast.fallbacks() == 0 or mode == "certs_fail"
    stock.close(); fast.close()


def test_tables_follow_the_store():
    """a validator-set change and a height prune drop the arrival-time verdicts; the walks still decide correctly"""
    w = World(n=7, seed=7)
    height, raw = 3, b"blk"
    fast = w.host()
    fast.use_loop_batch(0)
    fast.use_batch(True)
    fast.set_id(b"x")
    fast.set_state(height, 0, None)
    per_round, proposals = build_traffic(w, height, raw, rounds=(1,))
    wires = [m.encode() for m in per_round[1]] + [proposals[1].encode()]
    fast.ingest_wire(wires)
    fast.set_state(height, 1, None)
    first = fast.handle_round_change(height, 1)
    # the same set again as a new validator set: tables cleared, the walk batches what it needs itself
    assert fast.vm_init({a: 1 for a in w.addrs})
    before = fast.loop_batch_calls()
    again = fast.handle_round_change(height, 1)
    assert sorted(first) == sorted(again)
    if first:
        assert fast.loop_batch_calls() > before
    fast.close()


@pytest.mark.parametrize("kind", [k for k in dict.fromkeys(KINDS) if k is not None])
def test_every_certificate_rule_from_rows(kind):
    """One ROUND_CHANGE set per rule of validPC / AreValidPCMessages / proposalMatchesCertificate: honest messages plus ONE whose
    certificate breaks that rule.  The verdict computed from the backend's rows on arrival (no nested message decoded), the
    walk over decoded objects with arrival-time verdicts, and the reference's per-message walk keep exactly the same messages."""
    n = 10
    w = World(n=n, seed=KINDS.index(kind))
    height, rnd, raw = 3, 2, b"the block"
    msgs = []
    for k, a in enumerate(w.addrs):
        cert_round = 1 if k % 2 else 0
        if k != 4:
            msgs.append(rc_message(w, height, rnd, a, raw=raw, cert=w.certificate(height, cert_round, raw), cert_round=cert_round))
        elif kind == "no_certificate":
            msgs.append(rc_message(w, height, rnd, a))
        elif kind == "proposal_without_certificate":
            msgs.append(rc_message(w, height, rnd, a, raw=raw, cert=None, cert_round=cert_round))
        elif kind == "other_proposal":
            msgs.append(rc_message(w, height, rnd, a, raw=raw, cert=w.certificate(height, cert_round, b"another"), cert_round=cert_round))
        else:
            r2 = b"%02d-block|" % KINDS.index(kind) + kind.encode()   # (fake_hash looks at the first 8 bytes: nested messages of its own)
            msgs.append(rc_message(w, height, rnd, a, raw=r2, cert=w.certificate(height, cert_round, r2, kind), cert_round=cert_round))
    wires = [m.encode() for m in msgs]
    hosts = {"stock": w.host(), "rows": w.host(), "objects": w.host()}
    for name, h in hosts.items():
        h.set_id(b"someone else")
        h.set_state(height, rnd, None)
        if name != "stock":
            h.use_loop_batch(0)
            h.use_batch(True)
            h.use_rc_rows(name == "rows")
    res = {name: h.ingest_wire(wires)[0] for name, h in hosts.items()}
    assert res["stock"] == res["rows"] == res["objects"]
    out = {name: sorted(h.handle_round_change(height, rnd)) for name, h in hosts.items()}
    assert out["stock"] == out["rows"] == out["objects"]
    accepted_by_default = kind in ("no_certificate", "unknown_field_in_prepare")
    assert (wires[4] in out["stock"]) == accepted_by_default and len(out["stock"]) == n - (0 if accepted_by_default else 1)
    irregular = kind == "short_hash_everywhere"
    assert hosts["rows"].rc_from_rows == n - (1 if irregular else 0) and hosts["objects"].rc_from_rows == 0
    # the messages handed out again are the bytes that came in, whether or not their certificates were ever decoded
    assert sorted(hosts["rows"].store_get_valid(height, rnd, 3)) == sorted(wires)
    for h in hosts.values():
        h.close()


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_a_flood_of_forged_carriers_does_not_buy_their_trees(mode):
    """judging a carrier's tree on arrival lets anybody buy N² signature checks with one message.  With
    cert_roots_first (2 = adaptive, the default) the carriers' envelopes are judged first while forged carriers keep arriving:
    the nested messages of a forged carrier are then never looked at.  Same verdicts in every mode."""
    n = 10
    w = World(n=n, seed=77)
    height, rnd, raw = 3, 1, b"the block"
    honest = [rc_message(w, height, rnd, a, raw=raw, cert=w.certificate(height, 0, raw), cert_round=0) for a in w.addrs]
    forged = []
    for k in range(40):                                   # a stranger replays honest certificates under envelopes of its own
        m = rc_message(w, height, rnd, w.addrs[k % n], raw=raw, cert=w.certificate(height, 0, raw), cert_round=0)
        m.signature = b"forged-%d" % k
        w.bad_wires.add(m.encode())
        forged.append(m)
    nested_asked = []
    ver = w.verifier()
    inner = ver["is_valid_validator"]
    ver["is_valid_validator"] = lambda wire: (nested_asked.append(wire), inner(wire))[1]
    stock, fast = w.host(), H.Host()
    assert fast.vm_init({a: 1 for a in w.addrs})
    fast.set_verifier(**ver)
    fast.use_loop_batch(0)
    fast.use_batch(True)
    fast.cert_roots_first(mode)
    for h in (stock, fast):
        h.set_id(b"someone else")
        h.set_state(height, rnd, None)
    batches = [[m.encode() for m in forged[k:k + 8]] for k in range(0, 40, 8)] + [[m.encode() for m in honest]]
    for wires in batches:
        assert stock.ingest_wire(wires)[0] == fast.ingest_wire(wires)[0]
    assert sorted(stock.handle_round_change(height, rnd)) == sorted(fast.handle_round_change(height, rnd))
    q = 2 * n // 3 + 1
    per_tree = 1 + q                                       # the carrier, the PREPREPARE, q − 1 PREPAREs
    everything = (40 + n) * per_tree
    if mode == 0:
        assert len(nested_asked) == everything and fast.roots_first_calls == 0
    elif mode == 1:
        assert len(nested_asked) == 40 + n + n * per_tree and fast.roots_first_calls == len(batches)
    else:   # the first batch is expanded in full (nothing was known), from then on forged carriers cost one check each; the
            # honest batch that follows the flood still pays the extra call, the one after it would not
        assert len(nested_asked) == 8 * per_tree + 32 + n + n * per_tree and fast.roots_first_calls == len(batches) - 1
        for _ in range(3):
            fast.ingest_wire([honest[0].encode()])         # re-deliveries: no carriers asked, the count decays
        fresh = rc_message(w, height, rnd + 1, w.addrs[0], raw=raw, cert=w.certificate(height, 0, raw), cert_round=0)
        calls = fast.roots_first_calls
        fast.ingest_wire([fresh.encode()])
        assert fast.roots_first_calls == calls             # calm again: one call per micro-batch
    stock.close(); fast.close()


RCC_SHAPES = ["honest", "duplicate_rc_sender", "rc_of_another_round", "rc_of_another_height", "rc_forged_envelope", "prepare_in_the_rcc",
              "too_few_rcs", "no_rcc", "one_invalid_pc_is_ignored", "two_prepared_rounds", "highest_round_prepared_another_block",
              "nobody_prepared_anything", "pc_round_at_the_limit"]


@pytest.mark.parametrize("shape", RCC_SHAPES)
def test_every_proposal_rule_from_rows(shape):
    """validateProposal (core/ibft.go:683-788) rule by rule: a PREPREPARE of round 2 whose RoundChangeCertificate has ONE
    property — decided from the backend's rows on arrival (the certificate never decoded), by the walk over decoded objects
    with arrival-time verdicts, and by the reference's per-message walk: the same answer."""
    from test_host_roundchange import PR
    n = 10
    w = World(n=n, seed=300 + RCC_SHAPES.index(shape))
    height, rnd, raw = 3, 2, b"the block"
    q = 2 * n // 3 + 1
    rcs = []
    for k, a in enumerate(w.addrs[:q]):
        cert_round = 0
        craw = raw
        if shape == "two_prepared_rounds":
            cert_round = k % 2                              # rounds 0 and 1: the proposal must carry the hash prepared in round 1
        if shape == "highest_round_prepared_another_block" and k == 3:
            cert_round, craw = 1, b"another block"         # the highest prepared round is about another block: the proposal's hash fails
        if shape == "pc_round_at_the_limit" and k == 3:
            cert_round, craw = 2, b"another block"         # round ≥ the proposal's round: not a valid PC, so it is ignored
        if shape == "nobody_prepared_anything":
            rcs.append(rc_message(w, height, rnd, a))
            continue
        kind = "bad_prepare_signature" if (shape == "one_invalid_pc_is_ignored" and k == 2) else None
        r2 = craw if kind is None else b"9-block|own"
        rcs.append(rc_message(w, height, rnd, a, raw=r2, cert=w.certificate(height, cert_round, r2, kind), cert_round=cert_round))
    if shape == "duplicate_rc_sender":
        rcs.append(rcs[0])
    elif shape == "rc_of_another_round":
        rcs[1] = rc_message(w, height, rnd + 1, rcs[1].sender, raw=raw, cert=w.certificate(height, 0, raw), cert_round=0)
    elif shape == "rc_of_another_height":
        rcs[1] = rc_message(w, height + 1, rnd, rcs[1].sender, raw=raw, cert=w.certificate(height, 0, raw), cert_round=0)
    elif shape == "rc_forged_envelope":
        w.bad_wires.add(rcs[1].encode())
    elif shape == "prepare_in_the_rcc":
        rcs[1] = W.IbftMessage(view=W.View(height, rnd), sender=rcs[1].sender, type=PR, signature=b"sig", payload=W.prepare_body(b"h" * 32))
    elif shape == "too_few_rcs":
        rcs = rcs[: q - 1]
    prop = w.proposer(height, rnd)
    # the block is re-proposed; its hash is that of (block, new round) — the certificate's hashes are those of the prepared rounds
    pp = W.IbftMessage(view=W.View(height, rnd), sender=prop, type=PP, signature=b"sig-pp-" + prop,
                       payload=W.preprepare_body(W.Proposal(raw, rnd), fake_hash(raw, rnd),
                                                 None if shape == "no_rcc" else W.round_change_certificate(rcs)))
    hosts = {"stock": w.host(), "rows": w.host(), "objects": w.host()}
    for name, h in hosts.items():
        h.set_id(b"someone else")
        h.set_state(height, rnd, None)
        if name != "stock":
            h.use_loop_batch(0)
            h.use_batch(True)
            h.use_rc_rows(name == "rows")
    res = {name: h.ingest_wire([pp.encode()])[0] for name, h in hosts.items()}
    assert res["stock"] == res["rows"] == res["objects"] == [2]
    out = {name: h.handle_preprepare(height, rnd) for name, h in hosts.items()}
    assert out["stock"] == out["rows"] == out["objects"]
    accepted = shape in ("honest", "one_invalid_pc_is_ignored", "two_prepared_rounds", "nobody_prepared_anything", "pc_round_at_the_limit")
    assert (out["stock"] is not None) == accepted and (out["stock"] in (None, pp.encode()))
    assert hosts["rows"].pp_from_rows == 1 and hosts["objects"].pp_from_rows == 0
    for h in hosts.values():
        h.close()
