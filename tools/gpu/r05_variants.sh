# next round's first measurement: the fused walk's experiment switches (tools/build_walk_variants.sh built the libraries), one short encode each -- the same md5 is the gate
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in default stages pairs stages_pairs wg5; do
  unset XEVE_HIP_LIB_PATH
  if [ $v != default ]; then export XEVE_HIP_LIB_PATH=$PWD/xeve_amd/lib/exp/libxeve_hip_$v.so; fi
  echo "== $v" | tee -a gpurun_out/r05_variants.log
  XEVE_HIP_WALK=1 timeout 120 python tools/probe_enc.py --width 1280 --height 720 --gops 448 --threads 8 --frames 2 --chunk 46 2>&1 | grep -E '"steps"|md5' | cut -c1-260 | tee -a gpurun_out/r05_variants.log
done
