# round 6, call b: the composed walk's side stream -- tree / encoder parity on every walk setting, then the step time at the bench's width with and without it
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
timeout 900 python -m pytest tests/test_hip_tree.py tests/test_zz_tree_golden_gpu.py tests/test_walk_choice_gpu.py tests/test_enc_batches.py -m gpu -x -q > gpurun_out/r06b_tree_tests.log 2>&1; echo "tree tests rc $?"; tail -n 4 gpurun_out/r06b_tree_tests.log
P="tools/probe_enc.py --width 1280 --height 512 --gops 668 --frames 2 --chunk 17"
for v in "side0:XEVE_HIP_TREE_SIDE=0" "side1:XEVE_HIP_TREE_SIDE=1" "side1_spec:XEVE_HIP_TREE_SIDE=1 XEVE_HIP_RDO_SPEC=1000000" "side0_spec:XEVE_HIP_TREE_SIDE=0 XEVE_HIP_RDO_SPEC=1000000"; do
  n=${v%%:*}; e=${v#*:}
  env $e timeout 300 python $P > gpurun_out/r06b_probe_$n.log 2>&1; echo "$n rc $?"; grep -E "steps|md5" gpurun_out/r06b_probe_$n.log | cut -c1-220
done
