#!/bin/bash
# One GPU-box visit, assembled from legs (replaces the one-shot tools/gpu_r*_*.sh scripts of rounds 3 - 5):
#
#   gpurun --timeout 1500 -- 'bash tools/visit.sh TAG leg [leg ...]'
#
# Every leg writes to gpurun_out/TAG_<leg>...; summaries worth keeping are copied to profiles/rNN_* by hand afterwards.
# Legs (arguments after ':' are comma-free, separated by ':'; a leg is one shell word):
#   tests[:expr]            pytest -m gpu (optionally -k expr)
#   smoke                   __graft_entry__.smoke()
#   bench[:args]            python bench.py --gpus 1 <args with '+' for spaces> (default: --steps 20 --warmup 5)
#   wl:NAME                 bench.py --workload NAME (sweep, rt64, rt64pbp, l1) without the cpu / e2e legs
#   kbench:ABL1:ABL2...     tools/kbench.py per-kernel HIP-event times, product build first, then each ablation build
#                           (exp_build/lib_<ABL>.so, made here beforehand with `tools/kbench.py --build --ablate ...`);
#                           KB_UTTS / KB_STEPS override 1024 / 5
#   stats[:WL]              rocprofv3 --kernel-trace --stats of bench.py (workload WL) -> TAG_kernel_stats[_WL].txt
#   traffic[:WL]            separate --pmc FETCH_SIZE / WRITE_SIZE passes -> TAG_traffic[_WL].json (gfx950 correction applied)
#   pmc[:WL]                three SQ counter passes (no other tracing) -> TAG_pmc[_WL]_{1,2,3}.txt
#   pmck:REGEX              the first two SQ counter sets restricted to the kernels matching REGEX -> TAG_pmck_{1,2}.txt
#   soak:N[:START[:ONLY]]   tools/fuzz_soak.py over N fresh seeds from START (ONLY: layer0 / l1rt / hmpp / alt / coder, '+'-joined; default layer0)
#   objpath                 tools/bench_chunk_api.py (8 workers, blocks of 32, with deletion) + tools/bench_dropin.py
#   rt                      tools/bench_rt.py capacity sweep
#   ubench:NAME             tools/ubench/NAME (a micro-benchmark binary built here beforehand) -> TAG_ubench_NAME.txt
#   py:SCRIPT[:args]        python SCRIPT args ('+' for spaces), output -> TAG_py_<basename>.log
#   with:VAR=VALUE          export VAR for the legs that follow (e.g. with:LLSM_AMD_LIB=exp_build/lib_KAL_BREAK_1.so: the suite
#                           against a deliberately broken build -- it must FAIL); unset:VAR removes it again
TAG=$1; shift
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
REPO=$PWD
O=$REPO/gpurun_out/$TAG
SETS=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE"
      "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM"
      "SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INST_LEVEL_VMEM")
BARGS="--no-cpu-baseline --no-e2e --no-other"
prof() {   # prof <dir> <rocprofv3 options...> -- <bench args...>   (run from /tmp, as the guide asks)
  local d=$1; shift
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 "$@" > ${O}_$d.log 2>&1)
}
for leg in "$@"; do
  IFS=: read -r name a1 a2 a3 a4 a5 a6 <<< "$leg"
  echo "== $leg =="
  case $name in
    tests)
      # (per-test limit: a hung test of visit v5 sat out the whole 1800 s and took the rest of the visit with it)
      timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout ${PYTEST_TIMEOUT:-300} ${a1:+-k "$a1"} 2>&1 | tee ${O}_pytest.log | grep -E "^E  |passed|failed|FAILED|rror|Timeout" | cut -c1-300 | head -40 ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee ${O}_smoke.log ;;
    bench)
      timeout 900 python bench.py --gpus 1 $( [ -n "$a1" ] && echo "${a1//+/ }" || echo "--steps 20 --warmup 5") 2>${O}_bench.err | tee ${O}_bench.json | cut -c1-400 ;;
    wl)
      timeout 400 python bench.py --workload $a1 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e 2>${O}_bench_$a1.err | tee ${O}_bench_$a1.json | cut -c1-260 ;;
    kbench)
      abl=""; for a in $a1 $a2 $a3 $a4 $a5 $a6; do abl="$abl $a"; done
      timeout 1200 python tools/kbench.py --utts ${KB_UTTS:-1024} --steps ${KB_STEPS:-5} --ablate $abl 2>&1 | tee -a ${O}_kbench.txt | cut -c1-700 ;;
    stats)
      prof stats${a1:+_$a1} --kernel-trace --stats -d ${O}_d -o bench -- python $REPO/bench.py --steps 5 --warmup 1 $BARGS ${a1:+--workload $a1}
      python tools/rocpd_summary.py $(find ${O}_d -name "*.db" | head -1) | grep -E "^kernel|k_|copy|Copy" > ${O}_kernel_stats${a1:+_$a1}.txt
      rm -rf ${O}_d; head -18 ${O}_kernel_stats${a1:+_$a1}.txt ;;
    traffic)
      prof tf --pmc FETCH_SIZE --kernel-trace -d ${O}_df -o bench -- python $REPO/bench.py --steps 2 --warmup 1 $BARGS ${a1:+--workload $a1}
      prof tw --pmc WRITE_SIZE --kernel-trace -d ${O}_dw -o bench -- python $REPO/bench.py --steps 2 --warmup 1 $BARGS ${a1:+--workload $a1}
      python tools/rocpd_traffic.py $(find ${O}_df -name "*.db" | head -1) $(find ${O}_dw -name "*.db" | head -1) > ${O}_traffic${a1:+_$a1}.json
      rm -rf ${O}_df ${O}_dw; head -c 500 ${O}_traffic${a1:+_$a1}.json; echo ;;
    pmc)
      for i in 1 2 3; do
        prof pmc$i --pmc ${SETS[$((i-1))]} --kernel-trace -d ${O}_dp -o bench -- python $REPO/bench.py --steps 1 --warmup 1 $BARGS ${a1:+--workload $a1}
        python tools/rocpd_summary.py $(find ${O}_dp -name "*.db" | head -1) | grep -E "^k_|^void k_|counter|^# " > ${O}_pmc${a1:+_$a1}_$i.txt
        rm -rf ${O}_dp
      done; head -4 ${O}_pmc${a1:+_$a1}_1.txt | cut -c1-400 ;;
    pmck)
      for i in 1 2; do
        prof pmck$i --pmc ${SETS[$((i-1))]} --kernel-trace --kernel-include-regex "$a1" -d ${O}_dk -o bench -- python $REPO/bench.py --steps 1 --warmup 1 $BARGS
        python tools/rocpd_summary.py $(find ${O}_dk -name "*.db" | head -1) | grep -E "$a1|counter|^# " > ${O}_pmck_$i.txt
        rm -rf ${O}_dk; cat ${O}_pmck_$i.txt | cut -c1-600
      done ;;
    soak)
      a3=${a3:-layer0}
      SOAK_ONLY=${a3//+/,} timeout 3000 python tools/fuzz_soak.py ${a2:-700000} ${a1:-1000} 2>&1 | tee -a ${O}_soak.log | tail -8 ;;
    objpath)
      timeout 300 python tools/bench_chunk_api.py --workers 8 --block 32 --reps 5 --batch-delete 1 2>/dev/null | tee ${O}_chunk_api.json | cut -c1-640
      timeout 300 python tools/bench_dropin.py 2>/dev/null | tee ${O}_dropin.json | cut -c1-400 ;;
    rt)
      timeout 600 python tools/bench_rt.py 2>&1 | tee ${O}_rt.log | tail -12 ;;
    py)
      timeout 1500 python $a1 ${a2//+/ } 2>&1 | tee ${O}_py_$(basename $a1 .py).log | tail -${PY_TAIL:-20} ;;
    ubench)                 # ubench:NAME -- tools/ubench/NAME (built here beforehand with hipcc)
      timeout 300 tools/ubench/$a1 2>&1 | tee ${O}_ubench_$a1.txt ;;
    with) export "$a1" ;;
    unset) unset "$a1" ;;
    *) echo "unknown leg $leg" ;;
  esac
done
