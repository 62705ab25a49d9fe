#!/bin/bash
# Where does one 64-byte line per wide record / triangle start to pay? tools/trace_bench.py (1920x1080 x SPP spp of the random
# triangle cloud) at several tree sizes, packed (LAYOUT=1) against lines (LAYOUT=2), same box, two rounds.
# -> profiles/r05_line_aligned.txt (the 8 MiB rule of wide_bvh.h: kWideLineAlignBytes)
for round in 1 2; do
for tris in 25000 50000 100000 200000 400000 1000000; do
  for L in 1 2; do
    echo -n "tris $tris LAYOUT=$L "; TRIS=$tris LAYOUT=$L SPP=${SPP:-32} timeout -k 10 300 python tools/trace_bench.py 2>&1 | tail -1
  done
done
done
