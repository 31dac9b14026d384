#!/bin/bash
# kernel trace of tools/scratch/x3w_shapes.py for the default library and the variant builds under tools/scratch/libs
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3o; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 python3 -c "import torch; print(torch.cuda.get_device_name(0))" || exit 1
for v in default "$@"; do
  unset JOVA_LIB JOVA_GEMM_NO_X3W
  if [ $v = inkernel ]; then export JOVA_GEMM_NO_X3W=1; elif [ $v != default ]; then export JOVA_LIB=$R/tools/scratch/libs/lib_$v.so; fi
  timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $O/kt_$v -- python3 $R/tools/scratch/x3w_shapes.py > $O/kt_$v.log 2>&1 || { echo "trace $v failed"; tail -5 $O/kt_$v.log; exit 1; }
  echo "== $v"; python3 $R/tools/kt_groups.py $O/kt_$v 12 8 | grep -v "at::native" | cut -c1-150
  rm -rf $O/kt_$v
done
