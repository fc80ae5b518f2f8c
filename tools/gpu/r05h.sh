# round 5: preset placebo on the device (the fused walk with 4x4 inter CUs, two reference pictures per list, the raster search), beside the preset-slow cases and the
# medium single runs as a regression check of the same library
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
t0=$(date +%s)
timeout 400 python -m pytest tests/test_enc_gpu.py -q -m gpu -p no:cacheprovider --durations=12 -k "placebo or slow or single_runs_on_the_gpu or p_slices" > gpurun_out/r05h_tests.log 2>&1
echo "tests rc $? in $(( $(date +%s) - t0 )) s" | tee -a gpurun_out/r05h_tests.log
tail -25 gpurun_out/r05h_tests.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05h_smoke.log 2>&1; tail -1 gpurun_out/r05h_smoke.log
