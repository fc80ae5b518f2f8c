# round 5, call 1: the WHOLE default GPU suite with per-test times (junit XML -> tests/golden/gpu_suite_durations.json), smoke, then the fused walk's experiment builds
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
t0=$(date +%s)
timeout 1500 python -m pytest tests -q -m gpu --durations=40 -p no:cacheprovider --junitxml=gpurun_out/r05a_suite.xml > gpurun_out/r05a_suite.log 2>&1
echo "suite rc $? in $(( $(date +%s) - t0 )) s" | tee -a gpurun_out/r05a_suite.log
tail -5 gpurun_out/r05a_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05a_smoke.log 2>&1; tail -2 gpurun_out/r05a_smoke.log
timeout 600 bash tools/gpu/r05_variants.sh > gpurun_out/r05a_variants_run.log 2>&1; tail -12 gpurun_out/r05_variants.log
