"""The mirror's entry points from several threads at once (every one of them takes the mirror's mutex; the queue has its own):
transport threads push, the worker ingests and signals, the signal callback and the main thread call handle_* / store_* /
set_state meanwhile.  The assertions are about the END state (everything pushed is ingested, the walks answer like a stock
mirror fed serially); tests/test_host_sanitize.py runs this file under ThreadSanitizer, which is where interleavings matter."""
import threading

import numpy as np

import go_ibft_amd.hostlib as H
from oracle import wire as W
from test_host_roundchange import _commit_world, PR, CM


def test_threads_push_while_others_walk_the_store():
    w, proposal, prepares, commits = _commit_world(n=60, bad=(3, 17))
    wires = [m.encode() for m in prepares + commits]
    ref, q = w.host(), w.host()
    for h in (ref, q):
        h.set_state(1, 0, proposal.encode())
        h.use_loop_batch(0)
        h.use_batch(True)
        h.enable_quorum_index()
    expect, *_ = ref.ingest_wire(wires)
    q.queue_start(max_rows=32, linger_us=0)
    stop = threading.Event()
    walks = []

    def on_signal(t, hh, rr):                                 # from the worker thread
        walks.append((t, q.store_num(hh, rr, t)))
    q.queue_on_signal(on_signal)

    def producer(chunk):
        for k in range(0, len(chunk), 4):
            part = chunk[k:k + 4]
            q.queue_push(np.frombuffer(b"".join(part), dtype=np.uint8),
                         np.concatenate([[0], np.cumsum([len(x) for x in part])]).astype(np.uint32))

    def walker():                                              # the round goroutine's calls, while messages still arrive
        while not stop.is_set():
            q.store_num(1, 0, PR)
            q.store_get_valid(1, 0, PR, lambda wire: True)
            q.seen_entries()
            q.lean_stats(1, 0, CM)
    threads = [threading.Thread(target=producer, args=(wires[i::4],)) for i in range(4)] + [threading.Thread(target=walker)]
    for t in threads:
        t.start()
    for t in threads[:-1]:
        t.join()
    st = q.queue_drain()
    stop.set()
    threads[-1].join()
    assert st.pushed == st.ingested == len(wires)
    assert st.stored == sum(1 for x in expect if x > 0)
    for t in (PR, CM):
        assert q.store_num(1, 0, t) == ref.store_num(1, 0, t)
    a, b = ref.handle_prepare(1, 0), q.handle_prepare(1, 0)
    assert (a[0], sorted(a[1])) == (b[0], sorted(b[1]))
    a, b = ref.handle_commit(1, 0), q.handle_commit(1, 0)
    assert (a[0], sorted(a[1])) == (b[0], sorted(b[1]))
    assert walks
    q.close(); ref.close()
