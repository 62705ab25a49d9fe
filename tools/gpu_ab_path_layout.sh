#!/bin/bash
# PbrtRenderParams.samples_per_wave: 1 (64 pixels of one sample index per wave, rounds 1-4) against 16 (the default) and 64,
# same box, alternating: config 3 (tools/trace_bench.py at 64 spp, with the per-launch times) and config 5 (4K x 32 spp).
for rep in 1 2 3; do
for K in 1 16 64; do
  echo -n "config 3 samples_per_wave $K: "; SPW=$K PBRT_HIP_TRACE_LOG=1 SPP=64 timeout -k 10 300 python tools/trace_bench.py 2>&1 | grep "k_trace\|libpbrt" | tail -7 | awk '/k_trace/ {printf "%s ", $5} /libpbrt/ {print "| " $0}'
done
done
for rep in 1 2; do
for K in 1 16 64; do
  echo -n "config 5 samples_per_wave $K: "; SPW=$K W=3840 H=2160 SPP=32 NO_COUNT=1 timeout -k 10 300 python tools/bench_c5.py 2>&1 | tail -1
done
done
