#!/bin/bash
# usage (in the container): tools/sweep_prebuilt.sh build "DEFS1" "DEFS2" ...  -> pbrt-rs_amd/pbrt_hip/libvar_N.so
#        (on the GPU box):  tools/sweep_prebuilt.sh run                        -> trace_bench on each variant, SPP from env
if [ "$1" = build ]; then
  shift; i=0
  for D in "$@"; do
    i=$((i+1))
    PB_DEFS="$D" PB_OUT=$PWD/pbrt-rs_amd/pbrt_hip/libvar_$i.so bash pbrt-rs_amd/build.sh 2>&1 | grep -E "error"
    echo "$D" > pbrt-rs_amd/pbrt_hip/libvar_$i.txt
  done
else
  for f in pbrt-rs_amd/pbrt_hip/libvar_*.so; do
    echo "== $(cat ${f%.so}.txt)"
    timeout -k 10 200 python tools/trace_bench.py $PWD/$f 2>&1 | grep -v amdgpu.ids || exit 1
  done
fi
