# next round, before the offsets work: what ONE stream gives (448 GOPs alone) against 448 + 220 in two -- the figure that says what a 668-GOP batch is worth
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import torch" > /dev/null 2>&1
B="--steps 5 --warmup 5 --pictures 2 --no-secondary --no-cpu-baseline"
for n in 1 2; do
  timeout 600 python bench.py $B --batches $n > gpurun_out/r05_batches_$n.json 2> gpurun_out/r05_batches_$n.err
  python -c "import json;d=json.load(open('gpurun_out/r05_batches_$n.json'));print('batches $n',d['value'],d['config']['gops_in_lockstep'],d['ms_per_step'],d['encode'])"
done
