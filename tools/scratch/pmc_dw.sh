#!/bin/bash
# PMC passes over one FFN weight-gradient launch (tools/one_gemm.py dw): big-tile kernel and the 128 x 128 kernel
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 python3 -c "import torch; print(torch.cuda.get_device_name(0))" || exit 1
for v in big; do
  unset JOVA_GEMM_NO_DW256; [ $v = old ] && export JOVA_GEMM_NO_DW256=1
  timeout -s KILL 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d $O/a_$v -- python3 $R/tools/one_gemm.py dw > $O/a_$v.log 2>&1 || { echo fail; exit 1; }
  timeout -s KILL 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM --output-format csv -d $O/b_$v -- python3 $R/tools/one_gemm.py dw > $O/b_$v.log 2>&1 || { echo fail; exit 1; }
  timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k_$v -- python3 $R/tools/one_gemm.py dw > $O/k_$v.log 2>&1 || { echo fail; exit 1; }
  echo "== $v"; python3 $R/tools/pmc_dump.py gemm_bf16x3 $O/a_$v $O/b_$v; head -2 $(find $O/k_$v -name "*kernel_stats.csv" | head -1) | cut -c1-110
  rm -rf $O/a_$v $O/b_$v $O/k_$v
done
