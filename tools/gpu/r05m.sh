# round 5: the adaptive loop filter's kernels (Main profile, xeve_hip_alf_*) against oracle and goldens
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 80 python -m pytest tests/test_alf.py tests/test_abi_symbols.py -q -m gpu -p no:cacheprovider --durations=5 --junitxml=gpurun_out/r05m_alf.xml > gpurun_out/r05m_alf.log 2>&1
echo "rc $?"; tail -n 25 gpurun_out/r05m_alf.log
