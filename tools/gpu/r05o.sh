# round 5: the ALF statistics' host form (kernel level) and the Main-profile encoder with classification, statistics and both filters on the GPU
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 80 python -m pytest tests/test_alf.py tests/test_main_profile.py -q -m gpu -p no:cacheprovider -k "host_forms or alf_kernels" --durations=4 --junitxml=gpurun_out/r05o_alf.xml > gpurun_out/r05o_alf.log 2>&1
echo "rc $?"; tail -n 20 gpurun_out/r05o_alf.log
