cd $GRAFT_REPO_ROOT
for i in 1 2; do for m in none torch_first lib_first; do timeout 120 python tools/harness_ab.py $m 2>&1 | tail -1; done; done
