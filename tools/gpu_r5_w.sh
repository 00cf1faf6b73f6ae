#!/bin/bash
# Round 5, visit W: the object path's default mode with the copy engine's transfers in the trace (--memory-copy-trace beside
# --kernel-trace; no counters): how much of a call pair the link is busy.
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d $REPO/gpurun_out/prof_chunk -o chunk -- python $REPO/tools/bench_chunk_api.py --workers 8 --block 32 --reps 3 --batch-delete 1 > $REPO/gpurun_out/prof_chunk.log 2>&1
tail -1 $REPO/gpurun_out/prof_chunk.log | cut -c100-520
DB=$(find $REPO/gpurun_out/prof_chunk -name "*.db" | head -1)
python - "$DB" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
print([r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")][:60])
PY
( echo "LLSM_PACKED_FRAMES=2 (default), 8 workers, blocks of 32, 1024 utterances, 4 call pairs traced, kernel + memory-copy trace"; python $REPO/tools/chunk_api_timeline.py $DB ) | tee $REPO/gpurun_out/r05_zz_chunk_api_timeline_mode2_copies.txt | tail -12
rm -rf $REPO/gpurun_out/prof_chunk
