#!/bin/bash
# bench A/B of an environment switch, twice each, one box: ab_env.sh VAR=value
set -u
O=gpurun_out/r3z; mkdir -p $O
for rep in 1 2; do
for v in off on; do
  if [ $v = on ]; then export "$1"; else unset ${1%%=*}; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-graph-stress > $O/b_$v.json 2> $O/b_$v.err || exit 1
  python - <<PY
import json
d=json.loads(open("$O/b_$v.json").read().strip().splitlines()[-1])
print("$1 $v", d["value"], d["ms_per_step"], d["roofline"]["avg_us"], d["roofline"]["frac"])
PY
done
done
