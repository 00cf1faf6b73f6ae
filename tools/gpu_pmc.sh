#!/bin/bash
# SQ counter passes over one bench step (-> gpurun_out/pmc_*.txt via tools/rocpd_summary.py)
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM" \
           "SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace -d $REPO/gpurun_out/pmc_$i -o bench -- python $REPO/bench.py --utts ${UTTS:-1024} --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --no-other ${WL:+--workload $WL} > $REPO/gpurun_out/pmc_$i.log 2>&1
  python $REPO/tools/rocpd_summary.py $REPO/gpurun_out/pmc_$i/bench_results.db | grep -E "^k_|^void k_|counter|^# " > $REPO/gpurun_out/pmc_$i.txt
  rm -rf $REPO/gpurun_out/pmc_$i
done
cat $REPO/gpurun_out/pmc_1.txt | head -5
