#!/bin/bash
# tests of the image forms 2 and 3, then the bench line with each, twice, on one box
set -u
O=gpurun_out/r3t; mkdir -p $O
for f in 3 2; do
  JOVA_GEMM_X3W_FORM=$f timeout -k 10 300 python -u -m pytest tests/test_gpu_ops.py -x -q --timeout 120 --timeout-method thread -k "bf16" > $O/pytest_f$f.log 2>&1 || { tail -20 $O/pytest_f$f.log; exit 1; }
  tail -1 $O/pytest_f$f.log
done
for rep in 1 2; do
  for f in 2 3; do
    JOVA_GEMM_X3W_FORM=$f timeout -k 10 300 python bench.py --no-cpu-baseline --no-graph-stress > $O/b_f${f}_$rep.json 2> $O/b_f${f}_$rep.err || exit 1
    python - <<PY
import json
d=json.loads(open("$O/b_f${f}_$rep.json").read().strip().splitlines()[-1])
print("form $f rep $rep", d["value"], d["ms_per_step"], d["roofline"]["avg_us"], d["roofline"]["frac"], "fp32-only", d["fp32_mfma_only"]["ms_per_step"])
PY
  done
done
