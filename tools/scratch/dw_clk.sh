#!/bin/bash
# shader clock and cycles per slab of the dW kernel (lab builds with -DDW_CLOCK), one box: dw_clk.sh <variant> ...
set -u
export DW_CLOCK=1
for v in "$@"; do
  unset JOVA_LIB DW_ZERO
  lib=${v%%+*}; [ "$v" != "$lib" ] && export DW_ZERO=1
  export JOVA_LIB=$GRAFT_REPO_ROOT/tools/scratch/libs/lib_$lib.so
  timeout -k 10 120 python tools/scratch/dw_time.py 221 30 || exit 1
done
