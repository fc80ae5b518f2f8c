mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 80 python bench.py --width 832 --height 480 --gops 160 --batches 2 --pictures 2 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | cut -c1-2200 | tee gpurun_out/r04w_bench_480p.json
timeout 80 python -m pytest tests/test_enc_gpu.py -x -q -m gpu -k "gops_128x64_noise or tiny_ldb_fast or wide_batch or gops_cif_noise_m8" 2>&1 | tail -2
