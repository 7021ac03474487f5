"""core/drop_test.go over the host mirror: nodes that go offline (neither send nor run, core/helpers_test.go:101-110), nodes
whose Transport drops half of what they multicast (`faulty`, drop_test.go:134-140), and clusters that lose more than f.

* TestDropMaxFaulty (:282-326): heights 1…5 with everybody, then f nodes stop; heights 6…10 still commit — through a round
  change whenever the round-robin proposer is one of the stopped nodes.
* TestDropMaxFaultyPlusOne (:224-274): with f + 1 nodes stopped no height commits however often the timer fires (nobody
  inserts, every online node keeps moving rounds); once they are back, heights 6…10 commit.
* TestDropAllAndRecover (:16-81): the same with every node stopped.
* TestMaxFaultyDroppingMessages (:105-148): f nodes drop each of their messages with probability ½ (here: a keyed hash of
  (node, type, view), so that the three modes see the same drops); n + 1 heights commit (five there).

Every scenario runs in stock / batch / ingest mode and the modes must agree on what every node inserted (block and
seals), height by height."""
import hashlib

import pytest

from oracle import wire as W
from test_host_cluster import MODES, SEAL, rc_cluster

RC = W.ROUND_CHANGE


def max_faulty(n):                                     # core/consensus_test.go:112-114
    return (n - 1) // 3


def drop_cluster(n, mode):
    proposals = [b"valid ethereum block"] * n          # buildValidEthereumBlock: every proposer builds the same bytes
    return rc_cluster(n, mode, proposals, set(proposals), lambda h, r: (h + r) % n)


def progress(c, height, max_ticks=8):
    """one RunSequence everywhere: round 0, then the timer until every online node has inserted (or max_ticks)"""
    online = [nd for nd in c.nodes if not nd.offline]
    inserted = c.run_height(height)
    ticks = 0
    while ticks < max_ticks and not all(nd.inserted is not None for nd in online):
        inserted = c.tick()
        ticks += 1
    return inserted, ticks


def set_offline(c, k, flag):                           # stopN / startN (core/helpers_test.go:249-259)
    for nd in c.nodes[:k]:
        nd.offline = flag
        if flag:
            nd.inbox = []


def check_height(c, inserted, n_online):
    q = c.nodes[0].host.vm_quorum()
    done = [x for nd, x in zip(c.nodes, inserted) if not nd.offline]
    assert len(done) == n_online and all(x is not None for x in done), inserted
    assert all(x[0] == b"valid ethereum block" and len(x[1]) >= q for x in done)
    assert all(sig.startswith(SEAL) for x in done for _, sig in x[1])
    assert all(x is None for nd, x in zip(c.nodes, inserted) if nd.offline)


def drop_max_faulty(n, mode):
    c = drop_cluster(n, mode)
    f, log = max_faulty(n), []
    try:
        for h in range(1, 6):
            inserted, ticks = progress(c, h)
            assert ticks == 0
            check_height(c, inserted, n)
            log.append(inserted)
        set_offline(c, f, True)
        for h in range(6, 11):
            inserted, ticks = progress(c, h)
            # rounds whose proposer is stopped are lost: the first online proposer (h + r) % n ≥ f ends the height
            want = next(r for r in range(n) if (h + r) % n >= f)
            assert ticks == want, (h, ticks, want)
            check_height(c, inserted, n - f)
            assert all(nd.round == want for nd in c.nodes[f:])
            log.append(inserted)
    finally:
        c.close()
    return log


@pytest.mark.parametrize("n", [4, 6, 7, 10])
def test_drop_max_faulty(n):
    logs = [drop_max_faulty(n, mode) for mode in MODES]
    assert logs[0] == logs[1] == logs[2]


def drop_more_than_faulty(n, mode, stopped):
    c = drop_cluster(n, mode)
    log = []
    try:
        for h in range(1, 4):
            inserted, _ = progress(c, h)
            check_height(c, inserted, n)
        set_offline(c, stopped, True)
        inserted, ticks = progress(c, 4, max_ticks=2 * n)
        assert ticks == 2 * n and all(x is None for x in inserted)                 # "cluster cannot reach height"
        assert all(nd.round == 2 * n and nd.state != "fin" for nd in c.nodes[stopped:])
        assert all(nd.host.store_num(4, 2 * n, RC) == n - stopped for nd in c.nodes[stopped:])   # the survivors' ROUND_CHANGEs, < quorum
        set_offline(c, stopped, False)
        for h in range(4, 9):                                                       # RunSequence(4) again, with everybody
            inserted, ticks = progress(c, h)
            assert ticks == 0
            check_height(c, inserted, n)
            log.append(inserted)
    finally:
        c.close()
    return log


@pytest.mark.parametrize("n", [4, 6, 9])
def test_drop_max_faulty_plus_one_stalls_then_recovers(n):
    logs = [drop_more_than_faulty(n, mode, max_faulty(n) + 1) for mode in MODES]
    assert logs[0] == logs[1] == logs[2]


@pytest.mark.parametrize("mode", MODES)
def test_drop_all_and_recover(mode):
    n = 6
    c = drop_cluster(n, mode)
    try:
        inserted, _ = progress(c, 1)
        check_height(c, inserted, n)
        set_offline(c, n, True)
        inserted, ticks = progress(c, 2, max_ticks=3)
        assert inserted == [None] * n
        set_offline(c, n, False)
        for h in range(2, 11):
            inserted, ticks = progress(c, h)
            assert ticks == 0
            check_height(c, inserted, n)
    finally:
        c.close()


def dropping(n, mode, key):
    c = drop_cluster(n, mode)
    f = max_faulty(n)

    def censor(nd, type_, view):                       # currentNode.faulty && rand.Intn(100) < 50 (drop_test.go:135)
        if nd.index >= f:
            return False
        d = hashlib.sha256(b"%s|%d|%d|%d|%d" % (key, nd.index, type_, view[0], view[1])).digest()
        return d[0] & 1 == 1
    c.censor = censor
    log, total_ticks = [], 0
    try:
        for h in range(1, n + 2):                      # every node — the faulty ones too — is round-0 proposer at least once
            inserted, ticks = progress(c, h, max_ticks=2 * n)
            total_ticks += ticks
            # n − f nodes never drop: whatever the faulty ones withhold, every node ends the height
            check_height(c, inserted, n)
            log.append(inserted)
    finally:
        c.close()
    return log, total_ticks


@pytest.mark.parametrize("n,key", [(4, b"a"), (6, b"f"), (7, b"c"), (10, b"d"), (13, b"e")])
def test_max_faulty_dropping_messages(n, key):
    runs = [dropping(n, mode, key) for mode in MODES]
    assert runs[0] == runs[1] == runs[2]
    assert runs[0][1] >= 1                             # some round was lost to a withheld PREPREPARE and changed
