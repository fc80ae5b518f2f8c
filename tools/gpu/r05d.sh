# round 5, call 3: the library's own job records count pairs of samples (xh_common.h XH_OFF2_HALF) -> ONE batch per GPU.  The parity tests of every kernel that builds or
# reads such a record, then the bench's job as one batch of 668 GOPs (round 4, same command, two batches of 448 + 220: 8.23 frames/s), composed and fused.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
timeout 900 python -m pytest -q -x -m gpu -p no:cacheprovider tests/test_hip_batched.py tests/test_rdoq.py tests/test_hip_intra.py tests/test_hip_rdo.py tests/test_hip_skip.py tests/test_hip_inter.py tests/test_hip_me.py tests/test_workload.py tests/test_hip_tree.py tests/test_zz_tree_golden_gpu.py tests/test_walk_choice_gpu.py tests/test_enc_gpu.py -k "not real_picture and not full_eight" > gpurun_out/r05d_tests.log 2>&1
tail -4 gpurun_out/r05d_tests.log
B="--steps 5 --warmup 5 --pictures 2 --no-secondary --no-cpu-baseline --no-1080p"
for w in composed fused; do
  timeout 700 python bench.py $B --walk $w > gpurun_out/r05d_one_batch_$w.json 2> gpurun_out/r05d_one_batch_$w.err
  tail -2 gpurun_out/r05d_one_batch_$w.err
  python -c "import json;d=json.load(open('gpurun_out/r05d_one_batch_$w.json'));print('$w',d['value'],d['ms_per_step'],d['config']['gops_in_lockstep'],d['encode'],d['bitstream_check']['byte_identical_to_the_reference'],d['bitstream_check']['pictures_of_the_golden_gop_matched'],d['roofline']['avg_launch_ms'],d['roofline']['algorithmic_bytes_per_launch'],d['roofline']['frac'])"
done
