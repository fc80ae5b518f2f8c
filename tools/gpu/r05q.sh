# round 5: the forward ATS table (Main profile, xeve_trans_map_tbl_hip) against oracle and the reference's goldens
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 60 python -m pytest tests/test_main_ats_fwd.py -q -m gpu -p no:cacheprovider --junitxml=gpurun_out/r05q_ats.xml > gpurun_out/r05q_ats.log 2>&1
echo "rc $?"; tail -n 12 gpurun_out/r05q_ats.log
