mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
B="--steps 5 --warmup 5 --pictures 2 --no-secondary --no-cpu-baseline"
XEVE_HIP_WALK_AUTO_MAX=2048 XEVE_HIP_WALK_C=4 timeout 600 python bench.py $B > gpurun_out/r04u_hybrid_c4.json 2> gpurun_out/r04u_hybrid_c4.err
tail -2 gpurun_out/r04u_hybrid_c4.err
python -c "import json;d=json.load(open('gpurun_out/r04u_hybrid_c4.json'));print('hybrid c4',d['value'],d['ms_per_step'],d['config']['walk'],d['encode'],d['bitstream_check']['byte_identical_to_the_reference'],d['bitstream_check']['all_seeded_gops_same_bytes'])"
