# round 5: the fused walk with a team's serial stages packed into ONE wave that sits on a different SIMD for each of a CU's four teams (walk.hip k_walk: spread) against
# round 4's layout (dealt over the team's four waves).  1920x1080 (104 steps per picture): the IDR picture's steps, then 52 steps of the B picture; 896 chains (one per
# team) and 3584 chains (four per team).  The md5 of a WHOLE small run under both layouts is the gate.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
for g in 112 448; do
  for v in "0 0" "1 1" "0 1"; do
    set -- $v
    echo "== gops $g spread $1 deal $2" | tee -a gpurun_out/r05c_spread.log
    XEVE_HIP_WALK=1 XEVE_HIP_WALK_SPREAD=$1 XEVE_HIP_WALK_DEAL=$2 timeout 300 python tools/probe_enc.py --width 1920 --height 1080 --gops $g --threads 8 --frames 2 --chunk 26 --max-steps 156 2>&1 | grep -E '"steps"' | cut -c1-120 | tee -a gpurun_out/r05c_spread.log
  done
done
for v in "0 0" "1 1"; do
  set -- $v
  echo "== whole run 352x288, 6 GOPs x 4 frames, spread $1 deal $2" | tee -a gpurun_out/r05c_spread.log
  XEVE_HIP_WALK=1 XEVE_HIP_WALK_SPREAD=$1 XEVE_HIP_WALK_DEAL=$2 timeout 300 python tools/probe_enc.py --width 352 --height 288 --gops 6 --threads 8 --frames 4 --chunk 64 2>&1 | grep -E 'md5' | cut -c1-300 | tee -a gpurun_out/r05c_spread.log
done
