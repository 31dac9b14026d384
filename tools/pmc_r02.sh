# three counter passes over the config-4 graph kernels (GPU box); each pass its own run
R=/root/repo
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc2_fetch -- python3 $R/tools/one_graph.py > /dev/null 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc2_write -- python3 $R/tools/one_graph.py > /dev/null 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc2_sq -- python3 $R/tools/one_graph.py > /dev/null 2>&1 || exit 1
cd $R && python3 tools/pmc_graph_collect.py gpurun_out/pmc2_fetch gpurun_out/pmc2_write gpurun_out/pmc2_sq gpurun_out/r02_pmc_graph.json
rm -rf gpurun_out/pmc2_fetch gpurun_out/pmc2_write gpurun_out/pmc2_sq
