cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in default ni; do
  if [ $v = ni ]; then export XEVE_HIP_LIB_PATH=$PWD/xeve_amd/lib/exp/libxeve_hip_ni.so; fi
  echo "== $v" | tee -a gpurun_out/r04z_noinline_coder.log
  timeout 24 python tools/probe_enc.py --width 832 --height 480 --gops 16 --threads 8 --frames 2 --chunk 27 2>&1 | grep -E '"steps"|md5' | cut -c1-260 | tee -a gpurun_out/r04z_noinline_coder.log
done
