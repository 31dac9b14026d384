#!/bin/bash
# kernel trace of tools/one_gemm.py ffn1 in the three forms (image, in-kernel split, fp32)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3m; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 python3 -c "import torch; print(torch.cuda.get_device_name(0))" || exit 1
for mode in img x3 f32; do
  unset JOVA_GEMM_NO_X3W JOVA_GEMM_NO_BF16X3
  [ $mode = x3 ] && export JOVA_GEMM_NO_X3W=1
  [ $mode = f32 ] && export JOVA_GEMM_NO_BF16X3=1
  timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$mode -- python3 $R/tools/one_gemm.py ffn1 > $O/kt_$mode.log 2>&1 || { echo "trace $mode failed"; tail -5 $O/kt_$mode.log; exit 1; }
  f=$(find $O/kt_$mode -name "*kernel_stats.csv" | head -1)
  echo "== $mode"; head -6 $f | cut -c1-150
done
