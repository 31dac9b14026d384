#include <hip/hip_runtime.h>
__device__ __forceinline__ void glds16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__global__ void two_arrays(const float* x, float* out, int n) {
    __shared__ __attribute__((aligned(16))) float b0[4096], b1[4096];
    const int tid = threadIdx.x, wave = tid >> 6;
    glds16(x + tid * 4, b0 + wave * 256);
    float acc = 0.f;
    for (int t = 0; t < n; t += 2) {
        __syncthreads();
        glds16(x + (t + 1) * 1024 + tid * 4, b1 + wave * 256);
        acc += b0[(tid * 7) & 1023];
        __syncthreads();
        glds16(x + (t + 2) * 1024 + tid * 4, b0 + wave * 256);
        acc += b1[(tid * 7) & 1023];
    }
    out[tid] = acc;
}
__global__ void one_array(const float* x, float* out, int n) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* b0 = sm; float* b1 = sm + 4096;
    const int tid = threadIdx.x, wave = tid >> 6;
    glds16(x + tid * 4, b0 + wave * 256);
    float acc = 0.f;
    for (int t = 0; t < n; t += 2) {
        __syncthreads();
        glds16(x + (t + 1) * 1024 + tid * 4, b1 + wave * 256);
        acc += b0[(tid * 7) & 1023];
        __syncthreads();
        glds16(x + (t + 2) * 1024 + tid * 4, b0 + wave * 256);
        acc += b1[(tid * 7) & 1023];
    }
    out[tid] = acc;
}
__global__ void dyn_buffer(const float* x, float* out, int n, int ppt) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, wave = tid >> 6;
    glds16(x + tid * 4, sm + wave * 256);
    float acc = 0.f;
    for (int t = 0; t < n; ++t) {
        float* cur = sm + (t & 1) * ppt * 2048;
        float* nxt = sm + ((t + 1) & 1) * ppt * 2048;
        __syncthreads();
        glds16(x + (t + 1) * 1024 + tid * 4, nxt + wave * 256);
        acc += cur[(tid * 7) & 1023];
    }
    out[tid] = acc;
}
template <int B> __device__ __forceinline__ float stepf(const float* x, float* sm, int t, int tid, int wave, int ppt) {
    float* cur = sm + B * ppt * 2048;
    float* nxt = sm + (1 - B) * ppt * 2048;
    __syncthreads();
    glds16(x + (t + 1) * 1024 + tid * 4, nxt + wave * 256);
    return cur[(tid * 7) & 1023];
}
__global__ void unrolled2(const float* x, float* out, int n, int ppt) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, wave = tid >> 6;
    glds16(x + tid * 4, sm + wave * 256);
    float acc = 0.f;
    for (int t = 0; t < n; t += 2) {
        acc += stepf<0>(x, sm, t, tid, wave, ppt);
        acc += stepf<1>(x, sm, t + 1, tid, wave, ppt);
    }
    out[tid] = acc;
}
template <int G>
__device__ __forceinline__ float4 fragx(const float* slots, int s, int h) {
    return *reinterpret_cast<const float4*>(slots + s * 64 + (((2 * G + h) ^ (s & 15)) << 2));
}
__global__ void real_like(const float* x, float* out, int n, int ppt) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* slots = sm;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, h = lane >> 5;
    glds16(x + tid * 4, slots + wave * 256);
    float acc = 0.f;
    for (int t = 0; t < n; ++t) {
        const int pb = (t & 1) * ppt;
        __syncthreads();
        glds16(x + (t + 1) * 1024 + tid * 4, slots + (((t + 1) & 1) * ppt * 32 + wave * 4) * 64);
        float4 a = fragx<0>(slots, pb * 32 + i, h);
        float4 b = fragx<1>(slots, pb * 32 + i, h);
        acc += a.x + a.y + a.z + a.w + b.x + b.w;
    }
    out[tid] = acc;
}
__global__ void real_like2(const float* x, float* out, int n, int ppt, const int* idx) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* slots = sm;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, h = lane >> 5;
    glds16(x + tid * 4, slots + wave * 256);
    float acc = 0.f;
    int nxt = idx[tid];
    for (int t = 0; t < n; ++t) {
        const int pb = (t & 1) * ppt;
        __syncthreads();
        glds16(x + (long long)nxt * 64 + tid * 4, slots + (((t + 1) & 1) * ppt * 32 + wave * 4) * 64);
        nxt = idx[(t + 2) * 128 + tid];
        float4 a = fragx<0>(slots, pb * 32 + i, h);
        float4 b = fragx<1>(slots, pb * 32 + i, h);
        acc += a.x + a.y + a.z + a.w + b.x + b.w;
    }
    out[tid] = acc;
}
template <int B> __device__ __forceinline__ float step2(const float* x, float* cur, float* nxt, int t, int tid, int wave, int i, int h) {
    __syncthreads();
    glds16(x + (t + 1) * 1024 + tid * 4, nxt + wave * 256);
    float4 a = fragx<0>(cur, i, h);
    float4 b = fragx<1>(cur, i, h);
    return a.x + a.y + a.z + a.w + b.x + b.w;
}
__global__ void static_two(const float* x, float* out, int n) {
    __shared__ __attribute__((aligned(16))) float b0[4096];
    __shared__ __attribute__((aligned(16))) float b1[4096];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, h = lane >> 5;
    glds16(x + tid * 4, b0 + wave * 256);
    float acc = 0.f;
    for (int t = 0; t < n; t += 2) {
        acc += step2<0>(x, b0, b1, t, tid, wave, i, h);
        acc += step2<1>(x, b1, b0, t + 1, tid, wave, i, h);
    }
    out[tid] = acc;
}
