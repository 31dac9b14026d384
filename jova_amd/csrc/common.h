// Shared device/host helpers for the jova_hip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define JOVA_OK 0
#define JOVA_ERR_ARG 1
#define JOVA_ERR_HIP 2
#define JOVA_ERR_LIMIT 3

void jova_set_error(const char* fmt, ...);

#define JOVA_CHECK_ARG(cond, ...)                                                    \
    do {                                                                             \
        if (!(cond)) {                                                               \
            jova_set_error(__VA_ARGS__);                                             \
            return JOVA_ERR_ARG;                                                     \
        }                                                                            \
    } while (0)

#define JOVA_CHECK_LAUNCH(name)                                                      \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) {                                                      \
            jova_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));    \
            return JOVA_ERR_HIP;                                                     \
        }                                                                            \
    } while (0)

#define JOVA_CHECK_HIP(expr)                                                         \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess) {                                                      \
            jova_set_error("%s: %s", #expr, hipGetErrorString(e_));                  \
            return JOVA_ERR_HIP;                                                     \
        }                                                                            \
    } while (0)

// Zero fill as a KERNEL on the stream (rows of row_bytes at a pitch; bytes and pitch multiples of 4).  The library does
// not use hipMemsetAsync: as memset nodes of a captured hipGraph small memsets did not take effect on every replay on
// this ROCm (the persistent nonsat kernel then met stale barrier words), a kernel node always does.
int jova_zero_fill(void* p, size_t row_bytes, size_t rows, size_t pitch_bytes, hipStream_t s);
#define JOVA_ZERO(ptr, bytes, stream)                                                               \
    do {                                                                                            \
        int rc_ = jova_zero_fill((ptr), (size_t)(bytes), 1, (size_t)(bytes), (stream));             \
        if (rc_ != JOVA_OK) return rc_;                                                             \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// C/D fragment of v_mfma_f32_32x32x2_f32: register r of lane l holds D[row][col] with
//   col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)
__device__ __forceinline__ int mfma32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// One 32x32 output tile accumulated over K (even) from LDS operands:
//   acc[i][j] += sum_k A(i,k) * B(k,j),  A(i,k) = a[i*a_si + k*a_sk],  B(k,j) = b[k*b_sk + j*b_sj]
// Lane l supplies A(l&31, k + (l>>5)) and B(k + (l>>5), l&31) per step (guide section 3, f32 operand map).
__device__ __forceinline__ void mfma_tile_lds(const float* a, int a_si, int a_sk, const float* b, int b_sk, int b_sj,
                                              int K, f32x16& acc, int lane) {
    const int i = lane & 31, h = lane >> 5;
    const float* ap = a + i * a_si + h * a_sk;
    const float* bp = b + h * b_sk + i * b_sj;
    // 8 k-pairs of operands are read ahead of their 8 dependent MFMAs so the LDS round trip is paid once per group
    for (int k0 = 0; k0 < K; k0 += 16) {
        float av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = min(k0 + 2 * u, K - 2);
            av[u] = ap[k * a_sk];
            bv[u] = bp[k * b_sk];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float x = (k0 + 2 * u < K) ? av[u] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, bv[u], acc, 0, 0, 0);
        }
    }
}
