# round 6, call g: bench (one side stream = the default after r06e) and a kernel trace of a mid-picture B step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
export XEVE_HIP_TREE_SIDE=2
timeout 1200 python bench.py --no-1080p > gpurun_out/r06g_bench.json 2> gpurun_out/r06g_bench.err; echo "bench rc $?"; cut -c1-400 gpurun_out/r06g_bench.json; tail -n 3 gpurun_out/r06g_bench.err
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o tr -- python $GRAFT_REPO_ROOT/tools/probe_enc.py --width 1280 --height 512 --gops 668 --frames 2 --chunk 17 > $GRAFT_REPO_ROOT/gpurun_out/r06g_probe.log 2>&1; echo "trace rc $?"
cd $GRAFT_REPO_ROOT
f=$(find /tmp/tr -name '*kernel_trace.csv' | head -1); ls -la $f
python tools/trace_step.py $f gpurun_out/r06g_step 16
