mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import torch" > /dev/null 2>&1
# 1. the bench as the driver runs it
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err
tail -3 gpurun_out/r04_bench.err; cut -c1-600 gpurun_out/r04_bench.json
# 2. kernel trace + stats of the same job (no extras), a 4 s window inside the first B picture of the timed region
cd /tmp
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -P 130:4:1 -d /tmp/prof_bench -o b -- python $R/bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $R/gpurun_out/r04_bench_under_rocprof.json 2> $R/gpurun_out/r04_bench_under_rocprof.err
find /tmp/prof_bench -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r04_bench_kernel_stats.csv \;
ls -la /tmp/prof_bench/* | head; head -12 $R/gpurun_out/r04_bench_kernel_stats.csv
# 3. PMC: HBM read bytes of the search kernel, one batch of the bench's width, two pictures
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "k_me_epzs" --output-format csv -d /tmp/pmc_fetch -o f -- python $R/bench.py --steps 3 --warmup 1 --batches 1 --pictures 2 --no-secondary --no-cpu-baseline > $R/gpurun_out/r04_pmc_fetch_bench.json 2> $R/gpurun_out/r04_pmc_fetch.err
tail -2 $R/gpurun_out/r04_pmc_fetch.err
# 4. PMC: HBM read bytes of the fused walk's kernel at the same width (448 GOPs x 8 chains, the I picture and 40 steps of the B picture)
XEVE_HIP_WALK=1 timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "k_walk" --output-format csv -d /tmp/pmc_walk -o w -- python $R/tools/probe_enc.py --width 3840 --height 2160 --gops 448 --threads 8 --frames 2 --chunk 40 --max-steps 342 > $R/gpurun_out/r04_pmc_walk_probe.log 2>&1
tail -3 $R/gpurun_out/r04_pmc_walk_probe.log
cd $R
python tools/pmc_summary.py gpurun_out/r04_pmc_search.json /tmp/pmc_fetch 2>&1 | tail -1
python tools/pmc_summary.py gpurun_out/r04_pmc_walk.json /tmp/pmc_walk 2>&1 | tail -1
