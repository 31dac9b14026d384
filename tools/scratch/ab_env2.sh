#!/bin/bash
# bench A/B of an environment switch, twice each, one box: ab_env2.sh VAR=VALUE
set -u
O=gpurun_out/r3z; mkdir -p $O
for rep in 1 2; do
for v in default "$@"; do
  if [ $v = default ]; then E=""; else E="$v"; fi
  env $E timeout -k 10 300 python bench.py --no-cpu-baseline --no-graph-stress > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/b.json").read().strip().splitlines()[-1])
print("$v", d["value"], d["ms_per_step"], d["roofline"]["avg_us"], d["roofline"]["frac"])
PY
done
done
