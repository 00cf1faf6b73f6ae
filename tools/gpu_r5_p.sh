#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $REPO/gpurun_out/prof_chunk -o chunk -- python $REPO/tools/bench_chunk_api.py --workers 8 --block 32 --reps 3 --batch-delete 1 > $REPO/gpurun_out/prof_chunk.log 2>&1
tail -1 $REPO/gpurun_out/prof_chunk.log | cut -c100-500
python $REPO/tools/chunk_api_timeline.py $(find $REPO/gpurun_out/prof_chunk -name "*.db" | head -1) | tee $REPO/gpurun_out/r05_p_chunk_api_timeline.txt
rm -rf $REPO/gpurun_out/prof_chunk
