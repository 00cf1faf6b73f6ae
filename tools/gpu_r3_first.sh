#!/bin/bash
# Round-3 first GPU visit: (1) two ranks sharing the one device, self-launched and under torchrun, each under
# `timeout 120`; (2) tools/ubench/mfma_valu: do f32 MFMA and VALU cycles add on one SIMD?
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
echo "== 2 ranks, self-launched, shared device =="
LLSM_BENCH_SHARE_DEVICE=1 timeout 120 python bench.py --gpus 2 --utts 256 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e \
  > gpurun_out/r3_2rank_self.json 2> gpurun_out/r3_2rank_self.err; echo "rc=$?"
tail -c 600 gpurun_out/r3_2rank_self.json; tail -5 gpurun_out/r3_2rank_self.err
echo "== 2 ranks under torch.distributed.run, shared device =="
LLSM_BENCH_SHARE_DEVICE=1 timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29621 \
  bench.py --gpus 2 --utts 256 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e \
  > gpurun_out/r3_2rank_torchrun.json 2> gpurun_out/r3_2rank_torchrun.err; echo "rc=$?"
tail -c 600 gpurun_out/r3_2rank_torchrun.json; tail -5 gpurun_out/r3_2rank_torchrun.err
echo "== mfma + valu =="
timeout 120 tools/ubench/mfma_valu | tee gpurun_out/r3_mfma_valu.txt
