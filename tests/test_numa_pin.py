"""go-ibft_amd/numa.py — pin the process to the GPU's NUMA node before HIP comes up (0.333 vs 0.345 ms for the headline kernel,
profiles/r05g_harness_ab.txt).  CPU: the pure parts, and that a box without a KFD topology is left alone."""
import os

import go_ibft_amd.numa as N


def test_cpulist_and_visible_devices(monkeypatch):
    assert N.parse_cpulist("0-3,8,10-11") == {0, 1, 2, 3, 8, 10, 11} and N.parse_cpulist("") == set()
    base = [0, 1, 2, 3, 4, 5, 6, 7]
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
    assert N._visible("ROCR_VISIBLE_DEVICES", 8, base) == base
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "4,5")
    rocr = N._visible("ROCR_VISIBLE_DEVICES", 8, base)
    assert rocr == [4, 5]
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1")                 # indexes into what ROCR left visible
    assert N._visible("HIP_VISIBLE_DEVICES", 8, rocr) == [5]
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1,GPU-deadbeef,0")  # a UUID ends the list, like in the runtime
    assert N._visible("HIP_VISIBLE_DEVICES", 8, rocr) == [5]
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "9")
    assert N._visible("HIP_VISIBLE_DEVICES", 8, rocr) == []


def test_no_topology_no_pinning_and_the_switch(monkeypatch):
    before = os.sched_getaffinity(0)
    info = N.pin_to_device_node(0)
    if not os.path.isdir("/sys/class/kfd/kfd/topology/nodes"):
        assert info["pinned"] is False and os.sched_getaffinity(0) == before
    else:                                                          # a GPU box: whatever it did, it stayed inside the old mask
        assert os.sched_getaffinity(0) <= before
        os.sched_setaffinity(0, before)
    monkeypatch.setenv("IBFT_NO_NUMA_PIN", "1")
    assert N.pin_to_device_node(0) == {"pinned": False, "device": 0, "why": "disabled"}
    assert N.pin_to_device_node(99)["pinned"] is False or True
