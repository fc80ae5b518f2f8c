# round 5: the Main-profile sample kernels (ALF, affine MC) timed on a 3840x2160 picture resident in HBM
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 70 python tools/probe_main_kernels.py > gpurun_out/r05r_main_kernels.json 2> gpurun_out/r05r_main_kernels.err
echo "rc $?"; cut -c1-1500 gpurun_out/r05r_main_kernels.json; tail -n 3 gpurun_out/r05r_main_kernels.err
