mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
XEVE_HIP_WALK_PROF=1 timeout 900 python tools/probe_enc.py --width 832 --height 480 --gops 64 --threads 8 --frames 2 --chunk 27 > gpurun_out/r04l_prof_480p_g64.log 2>&1
head -16 gpurun_out/r04l_prof_480p_g64.log
