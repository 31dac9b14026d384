#!/bin/bash
# Runs the single-GPU bench with the batch seeds ranks 1..7 would use (bench.py seeds its batches with 1000 + RANK);
# a rank other than 0 prints no JSON line, so the exit status and the elapsed time are the result.
set -e
for r in 1 2 3 4 5 6 7; do
  s=$(date +%s)
  RANK=$r JOVA_BENCH_WATCHDOG=60 timeout -k 10 120 python bench.py --no-cpu-baseline --no-graph-stress --steps 4 --warmup 1
  echo "rank-seed $r ok in $(( $(date +%s) - s )) s"
done
