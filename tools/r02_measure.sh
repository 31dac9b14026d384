# round-2 measurement set (GPU box): bench lines for every workload, kernel trace + stats of the default bench, the
# two-rank gloo diagnosis.  Outputs under gpurun_out/r02/.
R=/root/repo; O=$R/gpurun_out/r02; mkdir -p $O
cd $R
timeout -k 10 400 python bench.py > $O/bench_davis.json 2> $O/bench_davis.err || exit 1
echo davis done
for w in kiba metz_psc davis_b256_default long1024; do
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-graph-stress > $O/bench_$w.json 2> $O/bench_$w.err || exit 1
  echo $w done
done
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu-baseline --no-graph-stress > $O/bench_davis_under_rocprof.json 2> /dev/null || exit 1
cd $R
cp $O/kt/*/*kernel_stats.csv $O/bench_davis_kernel_stats.csv
python tools/kt_groups.py $O/kt 1 40 > $O/bench_davis_kernel_groups.txt
rm -rf $O/kt
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf_fetch -- python3 $R/tools/one_gemm.py ffn1 > /dev/null 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pf_write -- python3 $R/tools/one_gemm.py ffn1 > /dev/null 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pf_sq -- python3 $R/tools/one_gemm.py ffn1 > /dev/null 2>&1 || exit 1
cd $R
python tools/pmc_ffn_collect.py $O/pf_fetch $O/pf_write $O/pf_sq $O/r02_pmc_ffn1.json
rm -rf $O/pf_fetch $O/pf_write $O/pf_sq
bash tools/pmc_r02.sh
JOVA_DIAG_WATCHDOG=60 timeout -k 10 150 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/diag_two_rank.py cpu staged nonsat devtensor devtensor_big devtensor_loop > $O/diag_two_rank.log 2>&1
echo "diag rc=$?" >> $O/diag_two_rank.log
tail -5 $O/diag_two_rank.log
