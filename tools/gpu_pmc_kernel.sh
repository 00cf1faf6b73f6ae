#!/bin/bash
# SQ counters of ONE kernel (regex $1) over one bench step: two --pmc passes -> gpurun_out/pmck_<tag>_{1,2}.txt
#   bash tools/gpu_pmc_kernel.sh k_synth_tile [tag]
K=$1; TAG=${2:-$1}
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "$K" -d $REPO/gpurun_out/pmck_$i -o bench -- python $REPO/bench.py --utts ${UTTS:-1024} --steps 1 --warmup 1 --no-cpu-baseline --no-e2e > $REPO/gpurun_out/pmck_$i.log 2>&1
  python $REPO/tools/rocpd_summary.py $REPO/gpurun_out/pmck_$i/bench_results.db | grep -E "$K" > $REPO/gpurun_out/pmck_${TAG}_$i.txt
  rm -rf $REPO/gpurun_out/pmck_$i
done
cat $REPO/gpurun_out/pmck_${TAG}_1.txt $REPO/gpurun_out/pmck_${TAG}_2.txt
