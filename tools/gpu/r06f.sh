# round 6, call f: is the composed walk's step bound by the host's launch calls or by the device?  graph replay of the one-stream walk; the same walk at 64 chains
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
P="tools/probe_enc.py --width 1280 --height 512 --frames 2 --chunk 17"
for v in "graph_side0:XEVE_HIP_TREE_SIDE=0 XEVE_HIP_TREE_GRAPH=1:668" "narrow_side0:XEVE_HIP_TREE_SIDE=0 XEVE_HIP_WALK=0:8" "narrow_side2:XEVE_HIP_TREE_SIDE=2 XEVE_HIP_WALK=0:8" "narrow_graph:XEVE_HIP_TREE_SIDE=0 XEVE_HIP_WALK=0 XEVE_HIP_TREE_GRAPH=1:8" "mid_side2:XEVE_HIP_TREE_SIDE=2 XEVE_HIP_WALK=0:167" "mid_side0:XEVE_HIP_TREE_SIDE=0 XEVE_HIP_WALK=0:167"; do
  n=${v%%:*}; r=${v#*:}; e=${r%%:*}; g=${r#*:}
  env $e timeout 300 python $P --gops $g > gpurun_out/r06f_probe_$n.log 2>&1; echo "$n rc $?"; grep -E "steps|md5" gpurun_out/r06f_probe_$n.log | cut -c1-200
done
