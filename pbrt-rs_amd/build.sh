#!/bin/bash
# Builds libpbrt_hip.so (gfx950 code object + host C ABI) in-tree. The translation units are compiled side by side
# (PB_JOBS at a time, default 8) and linked once; PB_DEFS adds -D switches, PB_OUT names another output (variants).
set -e
cd "$(dirname "$0")/csrc"
OUT=${PB_OUT:-../pbrt_hip/libpbrt_hip.so}
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt --offload-arch=gfx950 -Wall -Wno-unused-function"
OBJ=$(mktemp -d /tmp/pbrt_hip_build.XXXXXX)
trap 'rm -rf "$OBJ"' EXIT
SRCS="pbrt_hip.hip render.hip hlbvh_gpu.hip wide_gpu.hip probe.hip host_bvh.cpp host_wide.cpp host_film.cpp film_reduce.cpp"
pids=()
for f in $SRCS; do
  /opt/rocm/bin/hipcc $FLAGS $PB_DEFS -c -o "$OBJ/${f%.*}.o" "$f" &
  pids+=($!)
  while [ "$(jobs -rp | wc -l)" -ge "${PB_JOBS:-8}" ]; do wait -n || exit 1; done
done
for p in "${pids[@]}"; do wait "$p" || exit 1; done
/opt/rocm/bin/hipcc $FLAGS -shared -o "$OUT" "$OBJ"/*.o -ldl -lpthread "$@"
