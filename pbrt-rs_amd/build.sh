#!/bin/bash
# Builds libpbrt_hip.so (gfx950 code object + host C ABI) in-tree.
set -e
cd "$(dirname "$0")/csrc"
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt --offload-arch=gfx950 -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS $PB_DEFS -shared -o ${PB_OUT:-../pbrt_hip/libpbrt_hip.so} pbrt_hip.hip render.hip hlbvh_gpu.hip wide_gpu.hip probe.hip host_bvh.cpp host_wide.cpp host_film.cpp film_reduce.cpp -ldl -lpthread "$@"
