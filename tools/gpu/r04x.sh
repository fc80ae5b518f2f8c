mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
python -c "import torch" > /dev/null 2>&1
timeout 50 python tools/gpu/two_procs.py 1280 720 448 2 2 2>&1 | grep procs | tee gpurun_out/r04x_two_procs.log
timeout 50 python tools/gpu/two_procs.py 1280 720 448 2 1 2>&1 | grep procs | tee -a gpurun_out/r04x_two_procs.log
