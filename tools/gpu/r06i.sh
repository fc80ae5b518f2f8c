# round 6, call i: second fold batch (skip decision + copy, shared RDOQ estimates, dropped blocks left to the reader, candidates in the first kernel, intra folds),
# the composed walk as the library's choice at every width -- parity, then step times
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
timeout 1200 python -m pytest tests/test_hip_mc_cu.py tests/test_hip_me.py tests/test_hip_sbac.py tests/test_hip_rdo.py tests/test_hip_skip.py tests/test_hip_inter.py tests/test_hip_intra.py tests/test_hip_tree.py tests/test_zz_tree_golden_gpu.py tests/test_walk_choice_gpu.py tests/test_enc_batches.py -m gpu -x -q > gpurun_out/r06i_tests.log 2>&1; echo "tests rc $?"; tail -n 4 gpurun_out/r06i_tests.log
P="tools/probe_enc.py --width 1280 --height 512 --frames 2 --chunk 17"
for v in "full:XEVE_HIP_TREE_SIDE=1:668" "one_stream:XEVE_HIP_TREE_SIDE=0:668" "g1:XEVE_HIP_TREE_SIDE=1:1" "g64:XEVE_HIP_TREE_SIDE=1:64"; do
  n=${v%%:*}; r=${v#*:}; e=${r%%:*}; g=${r#*:}
  env $e timeout 300 python $P --gops $g > gpurun_out/r06i_probe_$n.log 2>&1; echo "$n rc $?"; grep -E "steps|md5" gpurun_out/r06i_probe_$n.log | cut -c1-200
done
