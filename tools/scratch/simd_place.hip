// Experiment: are the waves of small workgroups spread over all 4 SIMDs of a CU?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void mfma_loop(float* out, int iters) {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float a = threadIdx.x * 0.001f, b = 1.0f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* out;
    hipMalloc(&out, 4 * 1024 * 1024 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    struct { int threads, blocks; } cfg[] = {{64, 4096}, {128, 2048}, {256, 1024}, {512, 512}, {64, 1024}, {128, 512}, {256, 256}, {128, 1024}, {64, 2048}};
    for (auto c : cfg) {
        hipLaunchKernelGGL(mfma_loop, dim3(c.blocks), dim3(c.threads), 0, 0, out, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop, dim3(c.blocks), dim3(c.threads), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)c.blocks * (c.threads / 64) * iters * 16 * 4096.0;
        printf("threads %4d blocks %5d (waves %5d): %8.3f ms  %7.1f TFLOP/s\n", c.threads, c.blocks, c.blocks * c.threads / 64, ms, flops / ms / 1e9);
    }
    return 0;
}
