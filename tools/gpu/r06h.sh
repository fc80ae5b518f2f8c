# round 6, call h: stream priorities for the side stream; the fused walk against the composed walk at small widths
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
P="tools/probe_enc.py --width 1280 --height 512 --frames 2 --chunk 17"
for v in "base:XEVE_HIP_TREE_SIDE=1:668" "side_low:XEVE_HIP_TREE_SIDE=1 XEVE_HIP_TREE_SIDE_PRIO=1:668" "side_low_main_high:XEVE_HIP_TREE_SIDE=1 XEVE_HIP_TREE_SIDE_PRIO=1 XEVE_HIP_ENC_PRIO=1:668" "fused_8:XEVE_HIP_WALK=1:8" "fused_1:XEVE_HIP_WALK=1:1" "composed_1:XEVE_HIP_WALK=0:1" "fused_64:XEVE_HIP_WALK=1:64" "composed_64:XEVE_HIP_WALK=0:64" "fused_128:XEVE_HIP_WALK=1:128" "composed_128:XEVE_HIP_WALK=0:128"; do
  n=${v%%:*}; r=${v#*:}; e=${r%%:*}; g=${r#*:}
  env $e timeout 300 python $P --gops $g > gpurun_out/r06h_probe_$n.log 2>&1; echo "$n rc $?"; grep -E "steps|md5" gpurun_out/r06h_probe_$n.log | cut -c1-200
done
