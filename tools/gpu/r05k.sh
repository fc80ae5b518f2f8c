# round 5: the bench's job at 1920x1080 for presets placebo (256 GOPs) and slow (512 GOPs) once more after bench.py learned that these presets run on the fused walk
# (its k_walk timer and roofline instead of the composed walk's search timer)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
timeout 85 python bench.py --preset placebo --width 1920 --height 1080 --gops 256 --batches 1 --pictures 2 --steps 4 --warmup 4 --no-secondary --no-cpu-baseline --no-1080p > gpurun_out/r05k_bench_placebo_1080p.json 2> gpurun_out/r05k_bench_placebo_1080p.err
echo "placebo rc $?"; cut -c1-200 gpurun_out/r05k_bench_placebo_1080p.json
timeout 85 python bench.py --preset slow --width 1920 --height 1080 --gops 512 --batches 1 --pictures 2 --steps 4 --warmup 4 --no-secondary --no-cpu-baseline --no-1080p > gpurun_out/r05k_bench_slow_1080p.json 2> gpurun_out/r05k_bench_slow_1080p.err
echo "slow rc $?"; cut -c1-200 gpurun_out/r05k_bench_slow_1080p.json
