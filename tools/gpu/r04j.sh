mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_zz_tree_golden_gpu.py tests/test_hip_tree.py -x -q 2>&1 | tail -3 > gpurun_out/r04j_tree_tests.log
cat gpurun_out/r04j_tree_tests.log
for C in 8 4; do
echo "== C=$C" >> gpurun_out/r04j_probe_720p.log
XEVE_HIP_WALK_C=$C timeout 900 python tools/probe_enc.py --width 1280 --height 720 --gops 448 --threads 8 --frames 2 --chunk 23 >> gpurun_out/r04j_probe_720p.log 2>&1
done
grep -E "==|steps" gpurun_out/r04j_probe_720p.log
echo "== 896 gops C=8" >> gpurun_out/r04j_probe_720p.log
timeout 900 python tools/probe_enc.py --width 1280 --height 720 --gops 896 --threads 8 --frames 2 --chunk 23 2>&1 | grep steps | tee -a gpurun_out/r04j_probe_720p.log
