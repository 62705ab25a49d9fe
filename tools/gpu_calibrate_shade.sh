#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of rocprofv3 on k_shade's access pattern: pbrt_hip_probe_state_stream with a known
# byte count, one --pmc pass per counter (the program directly after `--`), per kind of access and for the whole mix, at the
# queue densities of a config-3 frame. Output: gpurun_out/shade_calib/report.txt (-> profiles/r04_fetch_size_calibration_shade.txt)
set -o pipefail
OUT=gpurun_out/shade_calib
rm -rf $OUT; mkdir -p $OUT
for D in 1.0 0.7; do
  for P in 1 2 4 8 15 16 31; do
    dir=$OUT/p${P}_d${D}; mkdir -p $dir
    PARTS=$P DENSITY=$D python3 tools/calibrate_shade_fetch.py 2>/dev/null | tail -1 > $dir/known.json || exit 1
    for C in FETCH_SIZE WRITE_SIZE; do
      PARTS=$P DENSITY=$D timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $dir/$C -o c -- python3 tools/calibrate_shade_fetch.py > $dir/$C.log 2>&1 || { tail -5 $dir/$C.log; exit 1; }
    done
    echo "parts $P density $D done"
  done
done
python3 tools/calibrate_shade_fetch.py report $OUT | tee $OUT/report.txt
find $OUT -name "*.csv" -size +1M -delete
