cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3f
export JOVA_GEMM_SKT=4096 JOVA_GEMM_SKMINK=128
for CFG in "512 256 64" "2048 256 64" "2048 128 128" "4096 128 256" "4096 64 256" "8192 64 512"; do
  set -- $CFG
  rm -rf gpurun_out/r3f/kts
  JOVA_DW_T=$1 JOVA_DW_G=$2 JOVA_DW_C=$3 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3f/kts -- python3 tools/bench_small_gemm.py 4 > /dev/null 2>&1
  echo "== DW T=$1 G=$2 C=$3"; python tools/kt_groups.py gpurun_out/r3f/kts 5 14 | grep -v "at::native"
done > gpurun_out/r3f/kt_small2.txt 2>&1
rm -rf gpurun_out/r3f/kts
