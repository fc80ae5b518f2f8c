mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
B="--steps 5 --warmup 5 --pictures 2 --no-secondary --no-cpu-baseline"
timeout 600 python bench.py $B > gpurun_out/r04t_base.json 2> gpurun_out/r04t_base.err
python -c "import json;d=json.load(open('gpurun_out/r04t_base.json'));print('base',d['value'],d['ms_per_step'],d['encode'],d['bitstream_check']['byte_identical_to_the_reference'])"
XEVE_HIP_TREE_GRAPH=1 timeout 600 python bench.py $B > gpurun_out/r04t_graph2.json 2> gpurun_out/r04t_graph2.err
tail -2 gpurun_out/r04t_graph2.err
python -c "import json;d=json.load(open('gpurun_out/r04t_graph2.json'));print('graph2',d['value'],d['ms_per_step'],d['encode'],d['bitstream_check']['byte_identical_to_the_reference'])"
XEVE_HIP_TREE_GRAPH=1 timeout 600 python bench.py $B --gops 167 --batches 4 > gpurun_out/r04t_graph4.json 2> gpurun_out/r04t_graph4.err
tail -2 gpurun_out/r04t_graph4.err
python -c "import json;d=json.load(open('gpurun_out/r04t_graph4.json'));print('graph4',d['value'],d['ms_per_step'],d['config']['gops_in_lockstep'],d['encode'],d['bitstream_check']['byte_identical_to_the_reference'])"
