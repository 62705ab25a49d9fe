#!/bin/bash
# round 3, k_shade split at the light-sampling boundary: libvar_split.so = -DPB_SHADE_SPLIT=1.
# Parity of the variant (render + golden + li tests run against it), same-box A/B on config 3 and config 5's geometry, then the
# counters of both forms of k_shade (rocprofv3 --pmc, separate passes).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out
V=$PWD/pbrt-rs_amd/pbrt_hip/libvar_split.so
PBRT_HIP_LIB=$V timeout -k 10 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_golden.py tests/test_gpu_li.py -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r3_split_tests.txt
grep -q " passed" gpurun_out/r3_split_tests.txt && ! grep -q "failed\|error" gpurun_out/r3_split_tests.txt || exit 1
{
for rep in 1 2; do
echo "== config 3, 64 spp: one launch per bounce"; SPP=64 timeout -k 10 300 python tools/trace_bench.py || exit 1
echo "== config 3, 64 spp: split"; SPP=64 timeout -k 10 300 python tools/trace_bench.py $V || exit 1
done
echo "== config 5 (1080p x 16 spp): one launch per bounce"; W=1920 H=1080 SPP=16 NO_COUNT=1 timeout -k 10 300 python tools/bench_c5.py || exit 1
echo "== config 5 (1080p x 16 spp): split"; PBRT_LIB=$V W=1920 H=1080 SPP=16 NO_COUNT=1 timeout -k 10 300 python tools/bench_c5.py || exit 1
} 2>&1 | grep -v "amdgpu.ids\|scene + host\|wide records:" | tee gpurun_out/r3_split_ab.txt
OUT=gpurun_out/split_pmc; rm -rf $OUT; mkdir -p $OUT
for NAME in one split; do
  L=""; [ $NAME = split ] && L=$V
  SPP=64 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$NAME -o s -- python3 tools/trace_bench.py $L > $OUT/stats_$NAME.log 2>&1 || exit 1
  SPP=64 timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq_$NAME -o p -- python3 tools/trace_bench.py $L > $OUT/sq_$NAME.log 2>&1 || exit 1
  SPP=64 timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f_$NAME -o p -- python3 tools/trace_bench.py $L > $OUT/f_$NAME.log 2>&1 || exit 1
  SPP=64 timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w_$NAME -o p -- python3 tools/trace_bench.py $L > $OUT/w_$NAME.log 2>&1 || exit 1
  echo "===== $NAME"
  python3 - $OUT/stats_$NAME <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f, newline="")):
    if "k_shade" in r["Name"]: print(f"   {r['Name'][:40]:40s} calls {r['Calls']:>4s} total {float(r['TotalDurationNs'])/1e6:9.2f} ms avg {float(r['AverageNs'])/1e6:8.3f} ms")
PY
  for d in sq f w; do python3 tools/pmc_summary.py $OUT/${d}_$NAME | grep -A9 "k_shade"; done
done 2>&1 | tee gpurun_out/r3_split_pmc.txt
find $OUT -name "*.csv" -size +1M -delete
