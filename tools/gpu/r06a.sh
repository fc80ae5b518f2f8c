# round 6, call a: the GPU tests the driver's r05 run never reached (after the stale refusal test), then a kernel trace of the composed walk at the bench's width
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
timeout 1000 python -m pytest tests/test_enc_gpu.py tests/test_e2e_real_sizes.py -m gpu -x -q --durations=30 > gpurun_out/r06a_tail_tests.log 2>&1; echo "tests rc $?"; tail -n 5 gpurun_out/r06a_tail_tests.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o tr -- python $GRAFT_REPO_ROOT/tools/probe_enc.py --width 512 --height 512 --gops 668 --frames 2 --chunk 11 > $GRAFT_REPO_ROOT/gpurun_out/r06a_probe.log 2>&1; echo "trace rc $?"
cd $GRAFT_REPO_ROOT; tail -n 6 gpurun_out/r06a_probe.log
f=$(find /tmp/tr -name '*kernel_trace.csv' | head -1); ls -la $f
python tools/trace_step.py $f gpurun_out/r06a_step 3
