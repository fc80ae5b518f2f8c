#!/bin/bash
# PMC passes (each in its own run, --kernel-trace only, as the profiling rules require) over one bench step, levels one after the other on one stream
set -x
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02pmc
mkdir -p $O
cd /tmp
P="python $GRAFT_REPO_ROOT/tools/probe_step.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_serial -o s -- $P 3 --serial > $O/stats_serial.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_struct -o s -- $P 3 --serial --structured > $O/stats_struct.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/p_fetch -o p -- $P 1 --serial > $O/p_fetch.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/p_tcc -o p -- $P 1 --serial > $O/p_tcc.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/p_sq -o p -- $P 1 --serial > $O/p_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $O/p_sq2 -o p -- $P 1 --serial > $O/p_sq2.log 2>&1
XEVE_HIP_SBAC_REG=0 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/p_sq_reg0 -o p -- $P 1 --serial > $O/p_sq_reg0.log 2>&1
XEVE_HIP_SBAC_REG=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_serial_reg0 -o s -- $P 3 --serial > $O/stats_serial_reg0.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p_mfma -o p -- $P 3 --mfma > $O/p_mfma.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/p_mfma_fetch -o p -- $P 3 --mfma > $O/p_mfma_fetch.log 2>&1
tail -2 $O/*.log | cut -c1-200
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $O/pmc_serial.json $O/p_fetch $O/p_tcc $O/p_sq $O/p_sq2
python tools/pmc_summary.py $O/pmc_serial_reg0.json $O/p_sq_reg0
python tools/pmc_summary.py $O/pmc_mfma.json $O/p_mfma $O/p_mfma_fetch
