#!/bin/bash
# knock-out timings of the dW kernel, one box: dw_ko.sh <variant> ...   (variant "zeros" = default library, all-zero operands)
set -u
for rep in 1 2; do
for v in default "$@"; do
  unset JOVA_LIB DW_ZERO
  if [ $v = zeros ]; then export DW_ZERO=1; elif [ $v != default ]; then export JOVA_LIB=$GRAFT_REPO_ROOT/tools/scratch/libs/lib_$v.so; fi
  timeout -k 10 120 python tools/scratch/dw_time.py 221 30 || exit 1
done
done
