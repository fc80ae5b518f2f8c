mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_enc_gpu.py tests/test_walk_choice_gpu.py tests/test_gop_shard.py -x -q -m gpu -k "not full_eight and not 17f" --durations=12 2>&1 | tail -30 > gpurun_out/r04q_tests.log
cat gpurun_out/r04q_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/r04q_smoke.log
