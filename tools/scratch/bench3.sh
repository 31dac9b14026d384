#!/bin/bash
# the bench line with the three bf16-pieces forms of the big forward / data-gradient products, twice each, one box
set -u
O=gpurun_out/r3s; mkdir -p $O
for rep in 1 2; do
  for v in inkernel v1 v2; do
    unset JOVA_GEMM_NO_X3W JOVA_GEMM_X3W_V1
    [ $v = inkernel ] && export JOVA_GEMM_NO_X3W=1
    [ $v = v1 ] && export JOVA_GEMM_X3W_V1=1
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-graph-stress > $O/b_${v}_$rep.json 2> $O/b_${v}_$rep.err || exit 1
    python - <<PY
import json
d=json.loads(open("$O/b_${v}_$rep.json").read().strip().splitlines()[-1])
print("$v $rep", d["value"], d["ms_per_step"], d["roofline"]["avg_us"], d["roofline"]["frac"], "fp32-only", d["fp32_mfma_only"]["ms_per_step"])
PY
  done
done
