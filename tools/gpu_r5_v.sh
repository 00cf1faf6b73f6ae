#!/bin/bash
# Round 5, visit V: device timeline of the object path in its default mode (2: block transfers by the copy engine) and in
# mode 1 (kernels over the link), at the HEAD of the round; then a last layer-0 soak under the alternative conventions.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
for mode in 2 1; do
  LLSM_PACKED_FRAMES=$mode timeout 600 rocprofv3 --kernel-trace -d $REPO/gpurun_out/prof_chunk -o chunk -- python $REPO/tools/bench_chunk_api.py --workers 8 --block 32 --reps 3 --batch-delete 1 > $REPO/gpurun_out/prof_chunk.log 2>&1
  tail -1 $REPO/gpurun_out/prof_chunk.log | cut -c100-500
  ( echo "LLSM_PACKED_FRAMES=$mode, 8 workers, blocks of 32, 1024 utterances, 4 call pairs traced"; python $REPO/tools/chunk_api_timeline.py $(find $REPO/gpurun_out/prof_chunk -name "*.db" | head -1) ) | tee $REPO/gpurun_out/r05_zz_chunk_api_timeline_mode$mode.txt
  rm -rf $REPO/gpurun_out/prof_chunk
done
cd $REPO
( time SOAK_ONLY=alt timeout 900 python tools/fuzz_soak.py 95000 15000 ) > gpurun_out/r05_zz3_soak_alt.txt 2>&1
grep -E "alternative conventions|^FAIL|^real" gpurun_out/r05_zz3_soak_alt.txt | cut -c1-400 | head
find gpurun_out -name "parity_*.json" -delete
