#!/bin/bash
# Tile-count sweep: does a partly filled last round cost a whole round?  (profiles/r05_experiments.md)
# usage: sweep_tiles.sh <gemm_bench binary> ; N = 3072 (12 N tiles), M = 256 x Mt -> 12 Mt tiles on 256 CUs
B=${1:-tools/micro/_ab/gemm_bench_base}
for rep in 1 2; do
for K in 3072 15360; do
  for epi in 0 3; do
    args=""
    for mt in 16 21 22 24 26 28 30 32 34; do args="$args $((mt*256)) $K 3072 $epi"; done
    echo "== rep $rep K=$K epi=$epi (tiles 192 252 264 288 312 336 360 384 408)"
    GEMM_DATA=gauss $B c $args | grep "us "
  done
done
done
