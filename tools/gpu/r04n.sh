mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_zz_tree_golden_gpu.py tests/test_hip_tree.py -x -q 2>&1 | tail -3 > gpurun_out/r04n_tree_tests.log
cat gpurun_out/r04n_tree_tests.log
XEVE_HIP_WALK_PROF=1 timeout 900 python tools/probe_enc.py --width 832 --height 480 --gops 64 --threads 8 --frames 2 --chunk 27 > gpurun_out/r04n_prof_480p_g64.log 2>&1
head -30 gpurun_out/r04n_prof_480p_g64.log
timeout 900 python tools/probe_enc.py --width 1280 --height 720 --gops 896 --threads 8 --frames 2 --chunk 23 2>&1 | grep steps | tee gpurun_out/r04n_probe_720p_g896.log
