mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
XEVE_HIP_WALK=0 timeout 300 python tools/gpu/r04s.py 256 128 4 2>&1 | tail -4
timeout 300 python tools/gpu/r04s.py 832 480 160 2>&1 | tail -4
