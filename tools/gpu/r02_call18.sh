#!/bin/bash
set -x
export TMPDIR=/tmp
O=gpurun_out/r02c18
mkdir -p $O
for lp in none search search,cu_bits; do
  python bench.py --steps 15 --no-secondary --no-cpu-baseline --live-prof $lp > $O/b_$lp.json 2> $O/b_$lp.err
  python -c "import json;d=json.load(open('$O/b_$lp.json'));print('$lp', d['ms_per_step'], d['value'], d['roofline'].get('frac'))"
done
