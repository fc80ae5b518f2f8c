# round 6, call e: the winner's reconstruction as one back-half launch per component, a side stream per node size -- parity, then step times (per-level streams / one side stream / one stream)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
timeout 900 python -m pytest tests/test_hip_mc_cu.py tests/test_hip_me.py tests/test_hip_rdo.py tests/test_hip_skip.py tests/test_hip_inter.py tests/test_hip_tree.py tests/test_zz_tree_golden_gpu.py tests/test_enc_batches.py -m gpu -x -q > gpurun_out/r06e_tests.log 2>&1; echo "tests rc $?"; tail -n 4 gpurun_out/r06e_tests.log
P="tools/probe_enc.py --width 1280 --height 512 --gops 668 --frames 2 --chunk 17"
for v in "side1:XEVE_HIP_TREE_SIDE=1" "side2:XEVE_HIP_TREE_SIDE=2" "side0:XEVE_HIP_TREE_SIDE=0"; do
  n=${v%%:*}; e=${v#*:}
  env $e timeout 300 python $P > gpurun_out/r06e_probe_$n.log 2>&1; echo "$n rc $?"; grep -E "steps|md5" gpurun_out/r06e_probe_$n.log | cut -c1-220
done
