set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_tree.py -x -q -k "mode_decision_matches_oracle" 2>&1 | tail -15 > gpurun_out/r04a_tree_tests.log
cat gpurun_out/r04a_tree_tests.log
for C in 1 4 8 16; do
  echo "== fused walk C=$C" >> gpurun_out/r04a_probe_tree.log
  XEVE_HIP_WALK_C=$C timeout 300 python tools/probe_tree.py --chains=64,1024,4096 >> gpurun_out/r04a_probe_tree.log 2>&1
done
echo "== composed walk" >> gpurun_out/r04a_probe_tree.log
XEVE_HIP_WALK=0 timeout 300 python tools/probe_tree.py --chains=64,1024,4096 >> gpurun_out/r04a_probe_tree.log 2>&1
cat gpurun_out/r04a_probe_tree.log
