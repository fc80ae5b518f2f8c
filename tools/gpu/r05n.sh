# round 5: the unmodified Main-profile encoder with its ALF object's function pointers bound to the HIP host forms (oracle/ref_shim_alf.c): byte-identical bitstreams
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 70 python -m pytest tests/test_main_profile.py -q -m gpu -p no:cacheprovider -k "alf_kernels" --durations=3 --junitxml=gpurun_out/r05n_alf_insitu.xml > gpurun_out/r05n_alf_insitu.log 2>&1
echo "rc $?"; tail -n 20 gpurun_out/r05n_alf_insitu.log
