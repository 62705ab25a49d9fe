#!/bin/bash
# usage: tools/sweep.sh "DEFS1" "DEFS2" ...   (run on the GPU box)
mkdir -p gpurun_out/variants
i=0
for D in "$@"; do
  i=$((i+1))
  PB_DEFS="$D" PB_OUT=$PWD/gpurun_out/variants/lib_$i.so bash pbrt-rs_amd/build.sh 2>&1 | grep -E "error" 
  echo "variant $i: $D"
  timeout -k 10 300 python tools/trace_bench.py $PWD/gpurun_out/variants/lib_$i.so 2>&1 | grep -v amdgpu.ids
done
