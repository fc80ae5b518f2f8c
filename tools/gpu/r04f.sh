mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_zz_tree_golden_gpu.py tests/test_hip_tree.py -x -q 2>&1 | tail -4 > gpurun_out/r04f_tree_tests.log
cat gpurun_out/r04f_tree_tests.log
timeout 1200 python -m pytest tests/test_enc_gpu.py -x -q -k "not 2160 and not 1080 and not 720" 2>&1 | tail -4 > gpurun_out/r04f_enc_tests.log
cat gpurun_out/r04f_enc_tests.log
timeout 900 python tools/probe_enc.py --width 1280 --height 720 --gops 448 --threads 8 --frames 2 --chunk 8 > gpurun_out/r04f_probe_720p_g448.log 2>&1
tail -16 gpurun_out/r04f_probe_720p_g448.log
