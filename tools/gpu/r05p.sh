# round 5: affine motion compensation of a CU (Main profile, xeve_hip_affine_mc_jobs) against oracle and the reference's goldens
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 80 python -m pytest tests/test_affine.py tests/test_abi_symbols.py -q -m gpu -p no:cacheprovider --durations=4 --junitxml=gpurun_out/r05p_affine.xml > gpurun_out/r05p_affine.log 2>&1
echo "rc $?"; tail -n 25 gpurun_out/r05p_affine.log
