mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_enc_gpu.py tests/test_gop_shard.py -x -q -m gpu -k "full_eight or 17f or bench_with_two" 2>&1 | tail -15 > gpurun_out/r04o_new_tests.log
cat gpurun_out/r04o_new_tests.log
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r04o_bench_fused.json 2> gpurun_out/r04o_bench_fused.err
tail -5 gpurun_out/r04o_bench_fused.err
cat gpurun_out/r04o_bench_fused.json
