#!/bin/bash
# bench A/B of the default library against tools/scratch/libs/lib_<name>.so, twice each, one box: ab_lib.sh <name> ...
set -u
O=gpurun_out/r3y; mkdir -p $O
for rep in 1 2; do
for v in default "$@"; do
  unset JOVA_LIB; [ $v != default ] && export JOVA_LIB=$GRAFT_REPO_ROOT/tools/scratch/libs/lib_$v.so
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-graph-stress > $O/b_$v.json 2> $O/b_$v.err || exit 1
  python - <<PY
import json
d=json.loads(open("$O/b_$v.json").read().strip().splitlines()[-1])
print("$v", d["value"], d["ms_per_step"], d["roofline"]["avg_us"], d["roofline"]["frac"])
PY
done
done
