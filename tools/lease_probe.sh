#!/bin/bash
# tools/lease_probe.sh TAG — one lease: instruction-fetch and scattered-read probes next to the kernels whose time differs between leases
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out/profiles
{ echo "## tools/scatter_probe"; timeout 120 tools/scatter_probe; echo "## tools/table_ab.py 16384 65536 (cold kernels)"; timeout 120 python tools/table_ab.py 16384 65536 2>/dev/null; } > gpurun_out/profiles/$1_lease_probe.txt 2>&1
cat gpurun_out/profiles/$1_lease_probe.txt
