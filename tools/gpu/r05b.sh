# round 5, call 2: do more launch chains side by side pay once every stream has a hardware queue of its own?  (HIP maps streams onto GPU_MAX_HW_QUEUES = 4 queues by
# default; a batch encoder has three streams, so from two batches on, launch chains share queues and wait for each other in submission order.)  1920x1080 as the proxy
# (same launch chains per step, 104 steps per picture instead of 302): the IDR picture + the first B picture, composed walk.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
B="--width 1920 --height 1080 --steps 6 --warmup 4 --pictures 2 --no-secondary --no-cpu-baseline --walk composed"
run() { # name, env, extra args
  env $2 timeout 400 python bench.py $B $3 > gpurun_out/r05b_$1.json 2> gpurun_out/r05b_$1.err
  python - <<P
import json
try:
    d=json.load(open('gpurun_out/r05b_$1.json'))
    print('$1', d['value'], 'fps', d['ms_per_step'], 'ms/slice', d['config']['gops_in_lockstep'], d['encode'], d['bitstream_check'].get('byte_identical_to_the_reference'), d['bitstream_check'].get('pictures_of_the_golden_gop_matched'))
except Exception as e:
    print('$1 failed', e); print(open('gpurun_out/r05b_$1.err').read()[-600:])
P
}
run b2_q4 "XEVE_NOP=1" "--batches 2"
run b2_q8 "GPU_MAX_HW_QUEUES=8" "--batches 2"
run b3_q12 "GPU_MAX_HW_QUEUES=12" "--batches 3 --gops 700"
run b4_q16 "GPU_MAX_HW_QUEUES=16" "--batches 4 --gops 520"
