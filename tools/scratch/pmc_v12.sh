#!/bin/bash
# clock and MFMA-busy of the two image forms on the FFN linear1 launch (one_gemm.py ffn1): PMC pass + kernel trace each
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 python3 -c "import torch; print(torch.cuda.get_device_name(0))" || exit 1
for v in v2 v1; do
  unset JOVA_GEMM_X3W_V1; [ $v = v1 ] && export JOVA_GEMM_X3W_V1=1
  timeout -s KILL 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_LEVEL_VMEM --output-format csv -d $O/p_$v -- python3 $R/tools/one_gemm.py ffn1 > $O/p_$v.log 2>&1 || { echo "pmc $v failed"; exit 1; }
  timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k_$v -- python3 $R/tools/one_gemm.py ffn1 > $O/k_$v.log 2>&1 || { echo "kt $v failed"; exit 1; }
  echo "== $v"; python3 $R/tools/pmc_dump.py gemm_x3w $O/p_$v; f=$(find $O/k_$v -name "*kernel_stats.csv" | head -1); head -2 $f | cut -c1-120
  rm -rf $O/p_$v $O/k_$v
done
