#!/bin/bash
# Round-3 measurement set, one box: bench lines of the five workloads, kernel stats + probe check of the Davis bench under
# rocprofv3, PMC passes (graph kernels; the probed FFN launch).  Summaries land in gpurun_out/r3f_final/ for copying
# into profiles/.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3f_final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 python3 -c "import torch; print(torch.cuda.get_device_name(0))" || exit 1
cd $R
timeout -k 10 400 python3 bench.py > $O/bench_davis.json 2> $O/bench_davis.err || { echo "davis bench failed"; tail -5 $O/bench_davis.err; exit 1; }
echo "davis done"
for w in kiba metz_psc davis_b256_default long1024; do
  timeout -k 10 400 python3 bench.py --workload $w --no-cpu-baseline --no-graph-stress > $O/bench_$w.json 2> $O/bench_$w.err || { echo "$w bench failed"; tail -5 $O/bench_$w.err; exit 1; }
  echo "$w done"
done
cd /tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu-baseline --no-graph-stress > $O/bench_davis_under_rocprof.json 2> $O/kt.err || { echo "trace failed"; tail -5 $O/kt.err; exit 1; }
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats_davis.csv
python3 $R/tools/probe_vs_trace.py $O/kt $O/bench_davis_under_rocprof.json 5 30 $O/ffn1_probe_vs_trace.json
python3 $R/tools/kt_groups.py $O/kt 1 40 > $O/kernel_groups_davis.txt
rm -rf $O/kt
echo "trace done"
pass() {  # dir, counters, program args...
  local dir=$1 ctr=$2; shift 2
  timeout -s KILL 200 rocprofv3 --pmc $ctr --output-format csv -d $dir -- python3 "$@" > $dir.log 2>&1 || { echo "pmc pass $dir failed"; tail -5 $dir.log; exit 1; }
}
pass $O/g_fetch "FETCH_SIZE" $R/tools/one_graph.py
pass $O/g_write "WRITE_SIZE" $R/tools/one_graph.py
pass $O/g_sq "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT" $R/tools/one_graph.py
python3 $R/tools/pmc_graph_collect.py $O/g_fetch $O/g_write $O/g_sq $O/pmc_graph.json
rm -rf $O/g_fetch $O/g_write $O/g_sq
echo "graph pmc done"
pass $O/x_fetch "FETCH_SIZE" $R/tools/one_gemm.py ffn1
pass $O/x_write "WRITE_SIZE" $R/tools/one_gemm.py ffn1
pass $O/x_sq "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT" $R/tools/one_gemm.py ffn1
python3 $R/tools/pmc_ffn_collect.py $O/x_fetch $O/x_write $O/x_sq $O/pmc_x3w_ffn1.json gemm_x3w_kernel
python3 $R/tools/pmc_dump.py gemm_x3w_kernel $O/x_sq > $O/pmc_x3w_ffn1_sq.json
rm -rf $O/x_fetch $O/x_write $O/x_sq
echo "ffn pmc done"
