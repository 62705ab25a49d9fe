#!/bin/bash
mkdir -p gpurun_out/variants
i=0
for D in "$@"; do
  i=$((i+1))
  PB_DEFS="$D" PB_OUT=$PWD/gpurun_out/variants/libc5_$i.so bash pbrt-rs_amd/build.sh 2>&1 | grep -E "error"
  echo "variant $i: $D"
  PBRT_LIB=$PWD/gpurun_out/variants/libc5_$i.so W=1920 H=1080 SPP=8 timeout -k 10 300 python tools/bench_c5.py 2>&1 | grep -v "amdgpu.ids\|scene +"
done
