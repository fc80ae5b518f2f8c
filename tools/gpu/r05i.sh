# round 5, last call: the seconds of the GPU tests added since the full run (junit -> tools/gpu_suite_durations.py --merge), presets slow / placebo at 1920x1080 against the
# reference's bitstreams (gpu_full cases), and the fused walk's step times at 1920x1080 with 1024 chains for presets slow / placebo / medium (noise; the IDR picture and the
# first steps of the first B picture)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
timeout 150 python -m pytest tests/test_enc_gpu.py -q -m gpu -p no:cacheprovider -k "placebo and not 1920x1080" --junitxml=gpurun_out/r05i_new_tests.xml > gpurun_out/r05i_new_tests.log 2>&1
echo "new tests rc $?"; tail -2 gpurun_out/r05i_new_tests.log
t0=$(date +%s)
XEVE_GPU_FULL=1 timeout 240 python -m pytest tests/test_enc_gpu.py -q -m gpu -p no:cacheprovider -s --durations=4 -k "1920x1080_on_the_gpu" > gpurun_out/r05i_presets_1080p.log 2>&1
echo "1080p presets rc $? in $(( $(date +%s) - t0 )) s" | tee -a gpurun_out/r05i_presets_1080p.log; tail -8 gpurun_out/r05i_presets_1080p.log
for p in slow placebo medium; do
  timeout 110 python tools/probe_enc.py --width 1920 --height 1080 --gops 128 --threads 8 --frames 8 --chunk 30 --max-steps 120 --preset $p > gpurun_out/r05i_probe_$p.log 2>&1
  echo "probe $p rc $?"; grep wall_ms gpurun_out/r05i_probe_$p.log | cut -c1-120
done
