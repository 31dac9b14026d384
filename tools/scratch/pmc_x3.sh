#!/bin/bash
# PMC passes over the FFN linear1 launch (tools/one_gemm.py ffn1): bf16-pieces form (default) and fp32 MFMA form.
# One rocprofv3 run per counter group; the program sits directly after `--`.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3l
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 python3 -c "import torch; print(torch.cuda.get_device_name(0))" || exit 1
run() {  # name, counters
  timeout -s KILL 150 rocprofv3 --pmc $2 --output-format csv -d $O/$1 -- python3 $R/tools/one_gemm.py ffn1 > $O/$1.log 2>&1 || { echo "pass $1 failed"; tail -5 $O/$1.log; exit 1; }
  echo "pass $1 done"
}
for mode in x3 f32; do
  if [ $mode = f32 ]; then export JOVA_GEMM_NO_BF16X3=1; fi
  run ${mode}_a "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"
  run ${mode}_b "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_COEXEC_CYCLES"
  run ${mode}_c "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum"
  run ${mode}_d "FETCH_SIZE"
  run ${mode}_e "WRITE_SIZE"
  python3 $R/tools/pmc_dump.py gemm_f32_fast_kernel $O/${mode}_a $O/${mode}_b $O/${mode}_c $O/${mode}_d $O/${mode}_e > $O/${mode}.json
  find $O -name "*.csv" -size +2M -delete
done
rm -rf $O/*_a $O/*_b $O/*_c $O/*_d $O/*_e
cat $O/x3.json $O/f32.json
