# round 5, final call: the WHOLE default GPU suite (junit -> tests/golden/gpu_suite_durations.json), smoke, the bench as the driver runs it, and a kernel trace window
# inside the same bench command's first B picture
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
t0=$(date +%s)
timeout 1300 python -m pytest tests -q -m gpu --durations=30 -p no:cacheprovider --junitxml=gpurun_out/r05g_suite.xml > gpurun_out/r05g_suite.log 2>&1
echo "suite rc $? in $(( $(date +%s) - t0 )) s" | tee -a gpurun_out/r05g_suite.log
tail -4 gpurun_out/r05g_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05g_smoke.log 2>&1; tail -1 gpurun_out/r05g_smoke.log
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
tail -2 gpurun_out/r05_bench.err; cut -c1-400 gpurun_out/r05_bench.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -P 150:4:1 -d /tmp/prof_bench -o b -- python $R/bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-1080p > $R/gpurun_out/r05_bench_under_rocprof.json 2> $R/gpurun_out/r05_bench_under_rocprof.err
find /tmp/prof_bench -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r05_bench_kernel_stats.csv \;
head -6 $R/gpurun_out/r05_bench_kernel_stats.csv | cut -c1-200
