mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for C in 1; do
  echo "== C=$C" > gpurun_out/r04b_prof.log
  XEVE_HIP_WALK_PROF=1 XEVE_HIP_WALK_C=$C timeout 300 python tools/probe_walk.py --chains=64 > gpurun_out/r04b_prof.log 2>&1
done
cat gpurun_out/r04b_prof.log
