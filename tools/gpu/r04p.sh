mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
(cat /sys/fs/cgroup/cpu.max; nproc; cat /sys/fs/cgroup/cpuset.cpus.effective; lscpu | head -20) > gpurun_out/r04p_host.log 2>&1
# the inter picture alone at the bench's batch size: step time + stage profile, chains per team 4 (auto) / 8, jobs dealt round-robin / packed
for cfg in "0 0" "8 0" "8 1" "0 1"; do
  set -- $cfg
  XEVE_HIP_WALK_C=$1 XEVE_HIP_WALK_DEAL=$2 timeout 600 python tools/probe_enc.py --width 3840 --height 2160 --gops 448 --threads 8 --frames 2 --chunk 20 --prof-after 302 --max-steps 342 > gpurun_out/r04p_probe_c$1_d$2.log 2>&1
  grep -E '"steps": \[(0|20|302|322),' gpurun_out/r04p_probe_c$1_d$2.log
  grep -A14 "walk profile" gpurun_out/r04p_probe_c$1_d$2.log | head -16
done
timeout 1200 python bench.py --steps 20 --warmup 5 --walk composed --no-secondary --no-cpu-baseline > gpurun_out/r04p_bench_composed.json 2> gpurun_out/r04p_bench_composed.err
tail -3 gpurun_out/r04p_bench_composed.err
cut -c1-1800 gpurun_out/r04p_bench_composed.json
