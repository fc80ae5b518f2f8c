#!/bin/bash
# the lane-serial node kernel against the composed kernels on content with small levels (few bins per CU)
mkdir -p gpurun_out
{
for lane in 0 1; do for content in texture noise; do
  echo "== XEVE_HIP_TREE_LANE=$lane content=$content"
  XEVE_HIP_TREE_LANE=$lane timeout 200 python tools/probe_tree.py --chains=1,2048 --content=$content 2>&1 | grep chains
done; done
} > gpurun_out/r02_tree_lane_texture.log 2>&1
cat gpurun_out/r02_tree_lane_texture.log
