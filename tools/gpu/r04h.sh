mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_zz_tree_golden_gpu.py tests/test_hip_tree.py -x -q 2>&1 | tail -3 > gpurun_out/r04i_tree_tests.log
cat gpurun_out/r04i_tree_tests.log
XEVE_HIP_WALK_PROF=1 timeout 900 python tools/probe_enc.py --width 832 --height 480 --gops 64 --threads 8 --frames 2 --chunk 27 > gpurun_out/r04i_prof_480p_g64.log 2>&1
head -24 gpurun_out/r04i_prof_480p_g64.log
timeout 900 python tools/probe_enc.py --width 1280 --height 720 --gops 448 --threads 8 --frames 2 --chunk 23 > gpurun_out/r04i_probe_720p_g448.log 2>&1
tail -6 gpurun_out/r04i_probe_720p_g448.log
