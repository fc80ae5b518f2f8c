mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
timeout 455 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err
tail -2 gpurun_out/r04_bench_final.err; cut -c1-300 gpurun_out/r04_bench_final.json
