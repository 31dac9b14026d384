#!/bin/bash
# PMC passes over the FFN linear1 launch in the image form (gemm_x3w_kernel); one rocprofv3 run per counter group.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3n
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 python3 -c "import torch; print(torch.cuda.get_device_name(0))" || exit 1
run() {
  timeout -s KILL 150 rocprofv3 --pmc $2 --output-format csv -d $O/$1 -- python3 $R/tools/one_gemm.py ffn1 > $O/$1.log 2>&1 || { echo "pass $1 failed"; tail -5 $O/$1.log; exit 1; }
  echo "pass $1 done"
}
run w_a "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"
run w_b "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_COEXEC_CYCLES"
run w_c "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum"
run w_d "FETCH_SIZE"
run w_e "WRITE_SIZE"
run w_f "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY"
python3 $R/tools/pmc_dump.py gemm_x3w_kernel $O/w_a $O/w_b $O/w_c $O/w_d $O/w_e $O/w_f > $O/x3w.json
rm -rf $O/w_a $O/w_b $O/w_c $O/w_d $O/w_e $O/w_f
cat $O/x3w.json
