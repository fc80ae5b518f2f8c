# round 5: the bench's own job at presets slow (3840x2160, one batch of what HBM holds, IDR + first B picture) and placebo (1920x1080, 256 GOPs), each checked against the
# reference's bitstream of the same clip at that preset after every picture run (tests/golden/cfg4_8f_v1.json)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
timeout 280 python bench.py --preset slow --pictures 2 --steps 4 --warmup 4 --no-secondary --no-cpu-baseline --no-1080p > gpurun_out/r05j_bench_slow_2160p.json 2> gpurun_out/r05j_bench_slow_2160p.err
echo "slow rc $?"; cut -c1-300 gpurun_out/r05j_bench_slow_2160p.json; tail -n 3 gpurun_out/r05j_bench_slow_2160p.err
timeout 100 python bench.py --preset placebo --width 1920 --height 1080 --gops 256 --batches 1 --pictures 2 --steps 4 --warmup 4 --no-secondary --no-cpu-baseline --no-1080p > gpurun_out/r05j_bench_placebo_1080p.json 2> gpurun_out/r05j_bench_placebo_1080p.err
echo "placebo rc $?"; cut -c1-300 gpurun_out/r05j_bench_placebo_1080p.json; tail -n 3 gpurun_out/r05j_bench_placebo_1080p.err
