#!/bin/bash
# FFN linear1 forward with / without an environment switch, twice each, one box: x3w_ko.sh VAR=VALUE ...
set -u
for rep in 1 2; do
for v in default "$@"; do
  if [ $v = default ]; then E=""; else E="$v"; fi
  env $E timeout -k 10 120 python tools/scratch/x3w_time.py 221 30 || exit 1
  env $E timeout -k 10 120 python tools/scratch/x3w_time.py 279 30 || exit 1
done
done
