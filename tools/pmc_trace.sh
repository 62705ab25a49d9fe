#!/bin/bash
# PMC passes over tools/trace_bench.py (config 3, SPP from env, default 64) or the script named by SCRIPT:
# usage tools/pmc_trace.sh <tag> [extra env assignments...]
# Each pass collects a few counters (separate runs, --kernel-trace only, as gpurun requires); tools/pmc_summary.py sums
# them per kernel. Output under gpurun_out/pmc_<tag>/.
set -o pipefail
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
export SPP=${SPP:-64}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
for SET in "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -o p$i -- python3 ${SCRIPT:-tools/trace_bench.py} > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; exit 1; }
  echo "pass $i done: $SET"
done
python3 tools/pmc_summary.py $OUT | tee $OUT/summary.txt
find $OUT -name "*.csv" -size +2M -delete   # raw per-dispatch tables are big; the summary is what is kept
