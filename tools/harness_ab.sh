cd $GRAFT_REPO_ROOT
echo "# KFD topology:"; python3 -c "
import sys; sys.path.insert(0,'.')
import go_ibft_amd.numa as N
print(N.kfd_gpu_nodes()); print('device 0 ->', N.device_cpulist(0))"
for f in /sys/class/drm/renderD*/device; do echo "$f numa $(cat $f/numa_node) cpus $(cat $f/local_cpulist)"; done
echo "# N = 4 096 cold kernel, one fresh process per line: unpinned (the scheduler's choice) / pinned by go_ibft_amd/numa.py / forced onto each node"
for rep in 1 2 3; do
  echo -n "unpinned: "; timeout 120 python tools/harness_ab.py none 200 2>&1 | tail -1
  echo -n "IBFT_PIN=1: "; IBFT_PIN=1 timeout 120 python tools/harness_ab.py none 200 2>&1 | tail -1
  echo -n "IBFT_PIN=1 torch_first: "; IBFT_PIN=1 timeout 120 python tools/harness_ab.py torch_first 200 2>&1 | tail -1
  echo -n "node0: "; timeout 120 taskset -c 0-63,128-191 python tools/harness_ab.py none 200 2>&1 | tail -1
  echo -n "node1: "; timeout 120 taskset -c 64-127,192-255 python tools/harness_ab.py none 200 2>&1 | tail -1
done
