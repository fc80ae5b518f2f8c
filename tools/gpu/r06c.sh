# round 6, call c: kernel trace of a mid-picture B step at the bench's width with the side stream, then the bench itself
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o tr -- python $GRAFT_REPO_ROOT/tools/probe_enc.py --width 1280 --height 512 --gops 668 --frames 2 --chunk 17 > $GRAFT_REPO_ROOT/gpurun_out/r06c_probe.log 2>&1; echo "trace rc $?"
cd $GRAFT_REPO_ROOT; grep -E "steps" gpurun_out/r06c_probe.log | cut -c1-200
f=$(find /tmp/tr -name '*kernel_trace.csv' | head -1); ls -la $f
python tools/trace_step.py $f gpurun_out/r06c_step 16
rm -rf /tmp/tr
timeout 1200 python bench.py > gpurun_out/r06c_bench.json 2> gpurun_out/r06c_bench.err; echo "bench rc $?"; cut -c1-1200 gpurun_out/r06c_bench.json; tail -n 3 gpurun_out/r06c_bench.err
