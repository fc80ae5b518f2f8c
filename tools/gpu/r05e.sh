# round 5, call 4: the measurement record at the bench's own configuration (ONE batch of 668 closed GOPs x 8 frames at 3840x2160, composed walk)
#  1. bench.py --pictures 0: WHOLE 8-frame GOPs timed and checked against the reference's file
#  2. PMC passes (each in a run of its own, --kernel-trace only): SQ + matrix-core counters, then FETCH_SIZE, over the IDR picture and 40 steps of the first B picture
#  3. rocprofv3 --kernel-trace --stats of the same steps
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
timeout 1200 python bench.py --steps 20 --warmup 5 --pictures 0 --no-secondary --no-cpu-baseline --no-1080p > gpurun_out/r05_bench_whole_gop.json 2> gpurun_out/r05_bench_whole_gop.err
tail -2 gpurun_out/r05_bench_whole_gop.err
python -c "import json;d=json.load(open('gpurun_out/r05_bench_whole_gop.json'));print('whole gop',d['value'],d['ms_per_step'],d['config']['gops_in_lockstep'],d['config']['pictures_run'],d['bitstream_check'])"
P="tools/probe_enc.py --width 3840 --height 2160 --gops 668 --threads 8 --frames 8 --chunk 38 --max-steps 342"
cd /tmp
XEVE_HIP_WALK=0 timeout 900 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --kernel-include-regex "k_me_epzs|k_cu_bits|k_rdo_mfma|k_dct_mfma" --output-format csv -d /tmp/pmc_sq -o s -- python $R/$P > $R/gpurun_out/r05_pmc_sq_probe.log 2>&1
tail -2 $R/gpurun_out/r05_pmc_sq_probe.log
XEVE_HIP_WALK=0 timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "k_me_epzs" --output-format csv -d /tmp/pmc_fetch -o f -- python $R/$P > $R/gpurun_out/r05_pmc_fetch_probe.log 2>&1
tail -2 $R/gpurun_out/r05_pmc_fetch_probe.log
XEVE_HIP_WALK=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_one -o b -- python $R/$P > $R/gpurun_out/r05_stats_probe.log 2>&1
find /tmp/prof_one -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r05_one_batch_kernel_stats.csv \;
head -8 $R/gpurun_out/r05_one_batch_kernel_stats.csv | cut -c1-160
cd $R
python tools/pmc_summary.py gpurun_out/r05_pmc_all.json /tmp/pmc_sq /tmp/pmc_fetch 2>&1 | tail -1
