cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3f
for SKT in 1024 4096; do
  rm -rf gpurun_out/r3f/kts
  JOVA_GEMM_SKT=$SKT JOVA_GEMM_SKMINK=128 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3f/kts -- python3 tools/bench_small_gemm.py 4 > /dev/null 2>&1
  echo "== SKT=$SKT"; python tools/kt_groups.py gpurun_out/r3f/kts 5 60
done > gpurun_out/r3f/kt_small.txt 2>&1
rm -rf gpurun_out/r3f/kts
