#!/bin/bash
# tools/gpu/job.sh -- the ONE script behind every gpurun call of the builder (round 6 on; rounds 4-5 kept a script per call: git history).  Runs on the GPU box from the
# repository root; everything it keeps goes to gpurun_out/<tag>_*.  usage: gpurun --timeout S -- 'bash tools/gpu/job.sh <tag> <step> [<step> ...]', a step being one of
#   tests:<pytest arguments>              pytest -m gpu -x -q over the given files / -k expressions (empty: the whole GPU suite, junit + durations kept)
#   probe:<name>:<env assignments>:<gops> tools/probe_enc.py at 1280x512 (34 lockstep steps per picture, all 8 row chains busy mid-picture), IDR + one B picture
#   bench:<bench.py arguments>            python bench.py ...
#   trace:<gops>:<step from the end>      rocprofv3 --kernel-trace of the probe, distilled by tools/trace_step.py (one lockstep step: sequence, per-kernel totals, queues)
#   stats:<gops>                          rocprofv3 --kernel-trace --stats of a 3840x2160 encode's IDR picture + 40 B steps (the per-kernel summary committed under profiles/)
#   smoke:                                __graft_entry__.smoke()
#   par:<name>:<env assignments>:<args>   tools/probe_par.py (G GOPs as K encoders on K host threads), args comma-separated
#   pmc:<gops>                            the PMC passes (SQ counters; FETCH_SIZE) over the same 3840x2160 steps, each in a run of its own, distilled by tools/pmc_summary.py
# (steps are separated by spaces: quote the whole argument list once, e.g. 'tests:tests/test_hip_tree.py probe:side0:XEVE_HIP_TREE_SIDE=0:668')
cd "$GRAFT_REPO_ROOT" || exit 2
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
TAG=$1; shift
PROBE="tools/probe_enc.py --width 1280 --height 512 --frames 2 --chunk 17"
PROBE4K="tools/probe_enc.py --width 3840 --height 2160 --threads 8 --frames 8 --chunk 38 --max-steps 342" # (stats / pmc: the bench's own picture -- the IDR picture and 40 steps of the first B picture)
for step in "$@"; do
  kind=${step%%:*}; arg=${step#*:}
  case $kind in
    tests)
      if [ -z "$arg" ] || [ "$arg" = "tests" ]; then
        timeout 1500 python -m pytest tests -m gpu -x -q --junitxml=gpurun_out/${TAG}_gpu_suite.xml --durations=0 > gpurun_out/${TAG}_gpu_suite.log 2>&1; echo "suite rc $?"; tail -n 5 gpurun_out/${TAG}_gpu_suite.log
      else
        n=$(echo "$arg" | tr -c 'A-Za-z0-9_' '_' | cut -c1-40)
        timeout 1500 python -m pytest ${arg//,/ } -m gpu -x -q > gpurun_out/${TAG}_tests_$n.log 2>&1; echo "tests rc $?"; tail -n 4 gpurun_out/${TAG}_tests_$n.log
      fi ;;
    probe)
      n=${arg%%:*}; r=${arg#*:}; e=${r%%:*}; g=${r#*:}
      env ${e//,/ } timeout 300 python $PROBE --gops $g > gpurun_out/${TAG}_probe_$n.log 2>&1; echo "probe $n rc $?"; grep -E '"steps"|md5' gpurun_out/${TAG}_probe_$n.log | cut -c1-210 ;;
    bench)
      timeout 1500 python bench.py ${arg//,/ } > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc $?"; cut -c1-300 gpurun_out/${TAG}_bench.json; tail -n 2 gpurun_out/${TAG}_bench.err ;;
    trace)
      g=${arg%%:*}; back=${arg#*:}
      (cd /tmp && rm -rf /tmp/tr && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o tr -- python $R/$PROBE --gops $g > $R/gpurun_out/${TAG}_trace_probe.log 2>&1); echo "trace rc $?"
      f=$(find /tmp/tr -name '*kernel_trace.csv' | head -1); python tools/trace_step.py $f gpurun_out/${TAG}_step $back; rm -rf /tmp/tr ;;
    stats)
      (cd /tmp && rm -rf /tmp/st && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -o s -- python $R/$PROBE4K --gops $arg > $R/gpurun_out/${TAG}_stats_probe.log 2>&1); echo "stats rc $?"
      find /tmp/st -name '*kernel_stats.csv' -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \; ; head -6 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-150; rm -rf /tmp/st ;;
    pmc)
      (cd /tmp && rm -rf /tmp/pmc_sq /tmp/pmc_fetch
       timeout 1500 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --kernel-include-regex "k_me_epzs|k_cu_bits|k_rdo_mfma|k_dct_mfma" --output-format csv -d /tmp/pmc_sq -o s -- python $R/$PROBE4K --gops $arg > $R/gpurun_out/${TAG}_pmc_sq_probe.log 2>&1; echo "pmc sq rc $?"
       timeout 1500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "k_me_epzs" --output-format csv -d /tmp/pmc_fetch -o f -- python $R/$PROBE4K --gops $arg > $R/gpurun_out/${TAG}_pmc_fetch_probe.log 2>&1; echo "pmc fetch rc $?")
      python tools/pmc_summary.py gpurun_out/${TAG}_pmc_all.json /tmp/pmc_sq /tmp/pmc_fetch 2>&1 | tail -1; rm -rf /tmp/pmc_sq /tmp/pmc_fetch ;;
    par)
      n=${arg%%:*}; r=${arg#*:}; e=${r%%:*}; g=${r#*:}
      env ${e//,/ } timeout 1200 python tools/probe_par.py ${g//,/ } > gpurun_out/${TAG}_par_$n.log 2>&1; echo "par $n rc $?"; grep parts gpurun_out/${TAG}_par_$n.log | cut -c1-400 ;;
    smoke)
      timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc $?"; tail -n 2 gpurun_out/${TAG}_smoke.log ;;
    *) echo "unknown step $step" ;;
  esac
done
