#!/usr/bin/env python3
"""tools/small_n.py [N …] — bench.py's small_n leg on its own (what a node with the reference's own validator counts gets: one
device call against one CPU core at N = 4, 6, 30 …, the crossover for IBFT_MIN_DEVICE_ROWS).  Prints one JSON object."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import go_ibft_amd.numa as NUMA  # noqa: E402
pin = NUMA.pin_to_device_node(0)
import go_ibft_amd.verifier as V  # noqa: E402
import bench  # noqa: E402

sizes = tuple(int(x) for x in sys.argv[1:]) or (4, 6, 8, 12, 16, 30, 64, 128, 256)
out = bench.small_n_leg(V, sizes)
out["numa_pin"] = pin
print(json.dumps(out))
