mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_tree.py -x -q -k "mode_decision_matches_oracle" 2>&1 | tail -3 > gpurun_out/r04c_tests.log
cat gpurun_out/r04c_tests.log
rm -f gpurun_out/r04c_prof.log
for cfg in "1 0" "1 1" "8 1"; do
  set -- $cfg
  echo "== C=$1 count_only=$2" >> gpurun_out/r04c_prof.log
  XEVE_HIP_WALK_PROF=1 XEVE_HIP_WALK_C=$1 XEVE_HIP_WALK_COUNT=$2 timeout 300 python tools/probe_walk.py --chains=64 >> gpurun_out/r04c_prof.log 2>&1
done
for C in 4 8; do
  echo "== fused walk C=$C count-only" >> gpurun_out/r04c_prof.log
  XEVE_HIP_WALK_C=$C XEVE_HIP_WALK_COUNT=1 timeout 300 python tools/probe_tree.py --chains=1024,4096 >> gpurun_out/r04c_prof.log 2>&1
done
cat gpurun_out/r04c_prof.log
