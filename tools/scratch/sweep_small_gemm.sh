mkdir -p gpurun_out/r3f
for W in 0 2 3 4; do for SKT in 1024 4096; do for MINK in 256 128; do
  echo "== W=$W SKT=$SKT MINK=$MINK"; JOVA_GEMM_W=$W JOVA_GEMM_SKT=$SKT JOVA_GEMM_SKMINK=$MINK timeout -k 10 100 python tools/bench_small_gemm.py 2>/dev/null | grep -v amdgpu
done; done; done > gpurun_out/r3f/sweep1.txt 2>&1
