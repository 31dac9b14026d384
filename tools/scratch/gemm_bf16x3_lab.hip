// Lab harness (not part of the library): C[M,N] = A[M,K] B[N,K]^T in fp32 ACCURACY from bf16 matrix instructions.
// Every fp32 operand is split on the fly into three bf16 pieces x = h + m + l (round-to-nearest at each level, the
// residual of each level is exact in fp32); a product keeps the six piece products of weight >= 2^-16
// (hh, hm, mh, mm, hl, lh; each is exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16) and drops ml, lm, ll
// (<= 2^-23 of the product).  Question: how much of the 16 x rate of the bf16 pipe survives the 6 x instruction count, the
// split arithmetic and 1.5 x the LDS bytes -- against 126 TFLOP/s of the fp32 kernel on the FFN shape.
//   hipcc -O3 --offload-arch=gfx950 -o gemm_bf16x3_lab gemm_bf16x3_lab.hip && ./gemm_bf16x3_lab [M N K]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int BM = 128, BN = 128, BK = 32, LDR = 40;       // LDR: bf16 per LDS row (80 bytes: 16-lane groups of 16-byte reads are conflict-free)

__device__ __forceinline__ float bf16_up(__bf16 v) {
    return __uint_as_float(((unsigned)(*reinterpret_cast<unsigned short*>(&v))) << 16);
}

// x[4] -> three bf16x4 pieces
__device__ __forceinline__ void split4(const float4 x, bf16x4& h, bf16x4& m, bf16x4& l) {
    const float v[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const __bf16 hi = (__bf16)v[i];
        const float r1 = v[i] - bf16_up(hi);
        const __bf16 mi = (__bf16)r1;
        const float r2 = r1 - bf16_up(mi);
        h[i] = hi;
        m[i] = mi;
        l[i] = (__bf16)r2;
    }
}

template <int NTERMS>                  // 6: hh hm mh mm hl lh; 3: hh hm mh (bf16x2-like accuracy, for comparison)
__global__ void __launch_bounds__(256, 2)
gemm_bf16x3_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) __bf16 sA[3][BM][LDR];
    __shared__ __attribute__((aligned(16))) __bf16 sB[3][BN][LDR];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, h = lane >> 5;
    const int ntn = N / BN;
    const int m0 = (blockIdx.x / ntn) * BM, n0 = (blockIdx.x % ntn) * BN;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float4 ra[4], rb[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = tid + 256 * j, row = f >> 3, c4 = f & 7;
            ra[j] = *reinterpret_cast<const float4*>(A + (long long)(m0 + row) * K + k0 + c4 * 4);
            rb[j] = *reinterpret_cast<const float4*>(B + (long long)(n0 + row) * K + k0 + c4 * 4);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = tid + 256 * j, row = f >> 3, c4 = f & 7;
            bf16x4 p0, p1, p2;
            split4(ra[j], p0, p1, p2);
            *reinterpret_cast<bf16x4*>(&sA[0][row][c4 * 4]) = p0;
            *reinterpret_cast<bf16x4*>(&sA[1][row][c4 * 4]) = p1;
            *reinterpret_cast<bf16x4*>(&sA[2][row][c4 * 4]) = p2;
            split4(rb[j], p0, p1, p2);
            *reinterpret_cast<bf16x4*>(&sB[0][row][c4 * 4]) = p0;
            *reinterpret_cast<bf16x4*>(&sB[1][row][c4 * 4]) = p1;
            *reinterpret_cast<bf16x4*>(&sB[2][row][c4 * 4]) = p2;
        }
    };
    gload(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        __syncthreads();
        lstore();
        __syncthreads();
        if (k0 + BK < K) gload(k0 + BK);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[3][2], fb[3][2];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    fa[p][t] = *reinterpret_cast<const bf16x8*>(&sA[p][wm + 32 * t + i][ks * 16 + 8 * h]);
                    fb[p][t] = *reinterpret_cast<const bf16x8*>(&sB[p][wn + 32 * t + i][ks * 16 + 8 * h]);
                }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    if (NTERMS == 6) {
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][a], fb[2][b], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2][a], fb[0][b], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][a], fb[1][b], acc[a][b], 0, 0, 0);
                    }
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][a], fb[1][b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][a], fb[0][b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][a], fb[0][b], acc[a][b], 0, 0, 0);
                }
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                C[(long long)m * N + n0 + wn + 32 * b + i] = acc[a][b][r];
            }
}

// v2: k16 slabs, two LDS buffers, ONE barrier per slab: the split + store of slab s+1 sits in the same basic block as
// the MFMAs of slab s, so the scheduler can put the VALU / LDS-write work into the MFMA shadow of the same wave.
constexpr int LD16 = 24;                                    // 48-byte rows: conflict-free 16-byte fragment reads
__device__ __forceinline__ void split8(const float4 x0, const float4 x1, bf16x8& h, bf16x8& m, bf16x8& l) {
    const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 hi = (__bf16)v[i];
        const float r1 = v[i] - bf16_up(hi);
        const __bf16 mi = (__bf16)r1;
        h[i] = hi;
        m[i] = mi;
        l[i] = (__bf16)(r1 - bf16_up(mi));
    }
}
// diagnostics (wrong results on purpose, same memory traffic): LAB_NOSPLIT = one conversion instead of the three-level
// split; LAB_NOSTORE = no split and no LDS stores; LAB_NOREAD = fragments read once, outside the loop
#ifdef LAB_NOSPLIT
__device__ __forceinline__ void cheap8(const float4 x0, const float4 x1, bf16x8& h, bf16x8& m, bf16x8& l) {
    const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = (__bf16)v[i];
    m = h;
    l = h;
}
#define SPLIT8 cheap8
#else
#define SPLIT8 split8
#endif
__global__ void __launch_bounds__(256, 2)
gemm_bf16x3_v2_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) __bf16 sA[2][3][BM][LD16];
    __shared__ __attribute__((aligned(16))) __bf16 sB[2][3][BN][LD16];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, h = lane >> 5;
    const int ntn = N / BN;
    const int m0 = (blockIdx.x / ntn) * BM, n0 = (blockIdx.x % ntn) * BN;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int row = tid >> 1, half = tid & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float4 ra[2][2], rb[2][2];                              // two slabs in flight (slot = slab & 1)
    const float* Ap = A + (long long)(m0 + row) * K + 8 * half;
    const float* Bp = B + (long long)(n0 + row) * K + 8 * half;
#ifdef LAB_MAP4
    // four lanes per row (64 contiguous bytes), rows r and r + 64 per thread: a load instruction touches 16 rows instead of 32
    const int rq = tid >> 2, cq = tid & 3;
    const float* Aq = A + (long long)(m0 + rq) * K + 4 * cq;
    const float* Bq = B + (long long)(n0 + rq) * K + 4 * cq;
#define GLOAD(SL, s_)                                                                   \
    do {                                                                                \
        ra[SL][0] = *reinterpret_cast<const float4*>(Aq + (s_) * 16);                   \
        ra[SL][1] = *reinterpret_cast<const float4*>(Aq + 64ll * K + (s_) * 16);        \
        rb[SL][0] = *reinterpret_cast<const float4*>(Bq + (s_) * 16);                   \
        rb[SL][1] = *reinterpret_cast<const float4*>(Bq + 64ll * K + (s_) * 16);        \
    } while (0)
#define LSTORE(SL, buf)                                                                 \
    do {                                                                                \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                 \
            bf16x4 p0, p1, p2;                                                          \
            split4(ra[SL][j], p0, p1, p2);                                              \
            *reinterpret_cast<bf16x4*>(&sA[buf][0][rq + 64 * j][4 * cq]) = p0;          \
            *reinterpret_cast<bf16x4*>(&sA[buf][1][rq + 64 * j][4 * cq]) = p1;          \
            *reinterpret_cast<bf16x4*>(&sA[buf][2][rq + 64 * j][4 * cq]) = p2;          \
            split4(rb[SL][j], p0, p1, p2);                                              \
            *reinterpret_cast<bf16x4*>(&sB[buf][0][rq + 64 * j][4 * cq]) = p0;          \
            *reinterpret_cast<bf16x4*>(&sB[buf][1][rq + 64 * j][4 * cq]) = p1;          \
            *reinterpret_cast<bf16x4*>(&sB[buf][2][rq + 64 * j][4 * cq]) = p2;          \
        }                                                                               \
    } while (0)
#else
#define GLOAD(SL, s_)                                                                   \
    do {                                                                                \
        ra[SL][0] = *reinterpret_cast<const float4*>(Ap + (s_) * 16);                   \
        ra[SL][1] = *reinterpret_cast<const float4*>(Ap + (s_) * 16 + 4);               \
        rb[SL][0] = *reinterpret_cast<const float4*>(Bp + (s_) * 16);                   \
        rb[SL][1] = *reinterpret_cast<const float4*>(Bp + (s_) * 16 + 4);               \
    } while (0)
#if defined(LAB_NOSTORE)
#define LSTORE(SL, buf) do { dummy += ra[SL][0].x + ra[SL][1].w + rb[SL][0].y + rb[SL][1].z; } while (0)
#else
#define LSTORE(SL, buf)                                                                 \
    do {                                                                                \
        bf16x8 p0, p1, p2;                                                              \
        SPLIT8(ra[SL][0], ra[SL][1], p0, p1, p2);                                       \
        *reinterpret_cast<bf16x8*>(&sA[buf][0][row][8 * half]) = p0;                    \
        *reinterpret_cast<bf16x8*>(&sA[buf][1][row][8 * half]) = p1;                    \
        *reinterpret_cast<bf16x8*>(&sA[buf][2][row][8 * half]) = p2;                    \
        SPLIT8(rb[SL][0], rb[SL][1], p0, p1, p2);                                       \
        *reinterpret_cast<bf16x8*>(&sB[buf][0][row][8 * half]) = p0;                    \
        *reinterpret_cast<bf16x8*>(&sB[buf][1][row][8 * half]) = p1;                    \
        *reinterpret_cast<bf16x8*>(&sB[buf][2][row][8 * half]) = p2;                    \
    } while (0)
#endif
#endif   /* LAB_MAP4 */
#define FRAG_READ_DO(CUR)                                                               \
        _Pragma("unroll") for (int p = 0; p < 3; ++p)                                   \
            _Pragma("unroll") for (int t = 0; t < 2; ++t) {                             \
                fa[p][t] = *reinterpret_cast<const bf16x8*>(&sA[CUR][p][wm + 32 * t + i][8 * h]); \
                fb[p][t] = *reinterpret_cast<const bf16x8*>(&sB[CUR][p][wn + 32 * t + i][8 * h]); \
            }
#ifdef LAB_NOREAD
#define FRAG_READ(CUR)
#else
#define FRAG_READ(CUR) FRAG_READ_DO(CUR)
#endif
#ifdef LAB_NOLOAD
#define GLOAD_LOOP(SL, s_) do { } while (0)
#else
#define GLOAD_LOOP(SL, s_) GLOAD(SL, s_)
#endif
#define STEP(CUR, s_)                                                                   \
    do {                                                                                \
        LSTORE(1 - CUR, 1 - CUR);       /* (past the end: stale registers into the idle buffer) */ \
        GLOAD_LOOP(1 - CUR, min((s_) + 3, ns - 1)); /* branch-free body: one basic block for the scheduler */ \
        FRAG_READ(CUR)                                                                  \
        _Pragma("unroll") for (int t = 0; t < 6; ++t)                                   \
            _Pragma("unroll") for (int a = 0; a < 2; ++a)                               \
                _Pragma("unroll") for (int b = 0; b < 2; ++b)                           \
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[t]][a], fb[PB[t]][b], acc[a][b], 0, 0, 0); \
        SCHED_HINT                                                                      \
        __syncthreads();                                                                \
    } while (0)
#ifdef LAB_SCHED
#define SCHED_HINT                                                                      \
    _Pragma("unroll") for (int q = 0; q < 24; ++q) {                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                              \
        __builtin_amdgcn_sched_group_barrier(0x002, LAB_SCHED, 0);                      \
        if ((q & 3) == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);            \
    }
#else
#define SCHED_HINT
#endif
    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
    bf16x8 fa[3][2], fb[3][2];
    float dummy = 0.f;
    const int ns = K / 16;                                  // even (K % 32 == 0)
    GLOAD(0, 0);
    LSTORE(0, 0);
    GLOAD(1, 1);
    GLOAD(0, min(2, ns - 1));
    __syncthreads();
#ifdef LAB_NOREAD
    FRAG_READ_DO(0)
#endif
    for (int s = 0; s < ns; s += 2) {
        STEP(0, s);
        STEP(1, s + 1);
    }
#undef GLOAD
#undef LSTORE
#undef STEP
    if (dummy == 123.456f) acc[0][0][0] += dummy;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                C[(long long)m * N + n0 + wn + 32 * b + i] = acc[a][b][r];
            }
}

// v3: producer / consumer waves.  Waves 0-3 multiply (fragment reads + MFMAs only), waves 4-7 feed (global load -> split ->
// LDS store); one barrier per k16 slab, two LDS buffers.  The feed chain's waits no longer sit in the issue stream of the
// waves that multiply.
__global__ void __launch_bounds__(512)
gemm_bf16x3_v3_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) __bf16 sA[2][3][BM][LD16];
    __shared__ __attribute__((aligned(16))) __bf16 sB[2][3][BN][LD16];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, h = lane >> 5;
    const int ntn = N / BN;
    const int m0 = (blockIdx.x / ntn) * BM, n0 = (blockIdx.x % ntn) * BN;
    const int ns = K / 16;
    if (wave >= 4) {                                        // ---- feeders
        const int p = tid - 256, row = p >> 1, half = p & 1;
        const float* Ap = A + (long long)(m0 + row) * K + 8 * half;
        const float* Bp = B + (long long)(n0 + row) * K + 8 * half;
        float4 ra[2][2], rb[2][2];
#define GLOAD3(SL, s_)                                                                  \
    do {                                                                                \
        ra[SL][0] = *reinterpret_cast<const float4*>(Ap + (s_) * 16);                   \
        ra[SL][1] = *reinterpret_cast<const float4*>(Ap + (s_) * 16 + 4);               \
        rb[SL][0] = *reinterpret_cast<const float4*>(Bp + (s_) * 16);                   \
        rb[SL][1] = *reinterpret_cast<const float4*>(Bp + (s_) * 16 + 4);               \
    } while (0)
#define LSTORE3(SL, buf)                                                                \
    do {                                                                                \
        bf16x8 p0, p1, p2;                                                              \
        split8(ra[SL][0], ra[SL][1], p0, p1, p2);                                       \
        *reinterpret_cast<bf16x8*>(&sA[buf][0][row][8 * half]) = p0;                    \
        *reinterpret_cast<bf16x8*>(&sA[buf][1][row][8 * half]) = p1;                    \
        *reinterpret_cast<bf16x8*>(&sA[buf][2][row][8 * half]) = p2;                    \
        split8(rb[SL][0], rb[SL][1], p0, p1, p2);                                       \
        *reinterpret_cast<bf16x8*>(&sB[buf][0][row][8 * half]) = p0;                    \
        *reinterpret_cast<bf16x8*>(&sB[buf][1][row][8 * half]) = p1;                    \
        *reinterpret_cast<bf16x8*>(&sB[buf][2][row][8 * half]) = p2;                    \
    } while (0)
        GLOAD3(0, 0);
        GLOAD3(1, 1);
        LSTORE3(0, 0);
        GLOAD3(0, min(2, ns - 1));
        __syncthreads();
        for (int s = 0; s < ns; s += 2) {
            LSTORE3(1, 1);
            GLOAD3(1, min(s + 3, ns - 1));
            __syncthreads();
            LSTORE3(0, 0);
            GLOAD3(0, min(s + 4, ns - 1));
            __syncthreads();
        }
#undef GLOAD3
#undef LSTORE3
        return;
    }
    // ---- multipliers
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
    __syncthreads();
    for (int s = 0; s < ns; ++s) {
        const int cur = s & 1;
        bf16x8 fa[3][2], fb[3][2];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fa[p][t] = *reinterpret_cast<const bf16x8*>(&sA[cur][p][wm + 32 * t + i][8 * h]);
                fb[p][t] = *reinterpret_cast<const bf16x8*>(&sB[cur][p][wn + 32 * t + i][8 * h]);
            }
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[t]][a], fb[PB[t]][b], acc[a][b], 0, 0, 0);
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                C[(long long)m * N + n0 + wn + 32 * b + i] = acc[a][b][r];
            }
}

// v4: v2 with a 256 x 128 workgroup tile (8 waves of 64 x 64): a quarter fewer operand bytes per MAC from L2.
__global__ void __launch_bounds__(512, 1)
gemm_bf16x3_v4_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) __bf16 sA[2][3][256][LD16];
    __shared__ __attribute__((aligned(16))) __bf16 sB[2][3][128][LD16];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, h = lane >> 5;
    const int ntn = N / 128;
    const int m0 = (blockIdx.x / ntn) * 256, n0 = (blockIdx.x % ntn) * 128;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int row = tid >> 1, half = tid & 1;               // A: 256 rows x 2 halves; B: rows 0..127 by threads 0..255
    const bool feedB = tid < 256;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float4 ra[2][2], rb[2][2];
    const float* Ap = A + (long long)min(m0 + row, M - 1) * K + 8 * half;
    const float* Bp = B + (long long)(n0 + (row & 127)) * K + 8 * half;
#define GLOAD4(SL, s_)                                                                  \
    do {                                                                                \
        ra[SL][0] = *reinterpret_cast<const float4*>(Ap + (s_) * 16);                   \
        ra[SL][1] = *reinterpret_cast<const float4*>(Ap + (s_) * 16 + 4);               \
        if (feedB) {                                                                    \
            rb[SL][0] = *reinterpret_cast<const float4*>(Bp + (s_) * 16);               \
            rb[SL][1] = *reinterpret_cast<const float4*>(Bp + (s_) * 16 + 4);           \
        }                                                                               \
    } while (0)
#define LSTORE4(SL, buf)                                                                \
    do {                                                                                \
        bf16x8 p0, p1, p2;                                                              \
        split8(ra[SL][0], ra[SL][1], p0, p1, p2);                                       \
        *reinterpret_cast<bf16x8*>(&sA[buf][0][row][8 * half]) = p0;                    \
        *reinterpret_cast<bf16x8*>(&sA[buf][1][row][8 * half]) = p1;                    \
        *reinterpret_cast<bf16x8*>(&sA[buf][2][row][8 * half]) = p2;                    \
        if (feedB) {                                                                    \
            split8(rb[SL][0], rb[SL][1], p0, p1, p2);                                   \
            *reinterpret_cast<bf16x8*>(&sB[buf][0][row][8 * half]) = p0;                \
            *reinterpret_cast<bf16x8*>(&sB[buf][1][row][8 * half]) = p1;                \
            *reinterpret_cast<bf16x8*>(&sB[buf][2][row][8 * half]) = p2;                \
        }                                                                               \
    } while (0)
#define STEP4(CUR, s_)                                                                  \
    do {                                                                                \
        LSTORE4(1 - CUR, 1 - CUR);                                                      \
        GLOAD4(1 - CUR, min((s_) + 3, ns - 1));                                         \
        bf16x8 fa[3][2], fb[3][2];                                                      \
        _Pragma("unroll") for (int p = 0; p < 3; ++p)                                   \
            _Pragma("unroll") for (int t = 0; t < 2; ++t) {                             \
                fa[p][t] = *reinterpret_cast<const bf16x8*>(&sA[CUR][p][wm + 32 * t + i][8 * h]); \
                fb[p][t] = *reinterpret_cast<const bf16x8*>(&sB[CUR][p][wn + 32 * t + i][8 * h]); \
            }                                                                           \
        _Pragma("unroll") for (int t = 0; t < 6; ++t)                                   \
            _Pragma("unroll") for (int a = 0; a < 2; ++a)                               \
                _Pragma("unroll") for (int b = 0; b < 2; ++b)                           \
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[t]][a], fb[PB[t]][b], acc[a][b], 0, 0, 0); \
        __syncthreads();                                                                \
    } while (0)
    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
    const int ns = K / 16;
    GLOAD4(0, 0);
    GLOAD4(1, 1);
    LSTORE4(0, 0);
    GLOAD4(0, min(2, ns - 1));
    __syncthreads();
    for (int s = 0; s < ns; s += 2) {
        STEP4(0, s);
        STEP4(1, s + 1);
    }
#undef GLOAD4
#undef LSTORE4
#undef STEP4
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < M) C[(long long)m * N + n0 + wn + 32 * b + i] = acc[a][b][r];
            }
}

// v5: v2 with 32-byte LDS rows (halves swapped on rows 8..15 mod 16: conflict-free fragment reads) = 49 KB per workgroup,
// and at most 168 VGPRs: THREE workgroups per CU to cover the per-slab dependency chain.
__device__ __forceinline__ int swz5(int row, int half) { return (half ^ ((row >> 3) & 1)) * 8; }
__global__ void __launch_bounds__(256, 3)
gemm_bf16x3_v5_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) __bf16 sA[2][3][BM][16];
    __shared__ __attribute__((aligned(16))) __bf16 sB[2][3][BN][16];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, h = lane >> 5;
    const int ntn = N / BN;
    const int m0 = (blockIdx.x / ntn) * BM, n0 = (blockIdx.x % ntn) * BN;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int row = tid >> 1, half = tid & 1, wpos = swz5(row, half);
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float4 ra[2][2], rb[2][2];
    const float* Ap = A + (long long)(m0 + row) * K + 8 * half;
    const float* Bp = B + (long long)(n0 + row) * K + 8 * half;
#define GLOAD5(SL, s_)                                                                  \
    do {                                                                                \
        ra[SL][0] = *reinterpret_cast<const float4*>(Ap + (s_) * 16);                   \
        ra[SL][1] = *reinterpret_cast<const float4*>(Ap + (s_) * 16 + 4);               \
        rb[SL][0] = *reinterpret_cast<const float4*>(Bp + (s_) * 16);                   \
        rb[SL][1] = *reinterpret_cast<const float4*>(Bp + (s_) * 16 + 4);               \
    } while (0)
#define LSTORE5(SL, buf)                                                                \
    do {                                                                                \
        bf16x8 p0, p1, p2;                                                              \
        split8(ra[SL][0], ra[SL][1], p0, p1, p2);                                       \
        *reinterpret_cast<bf16x8*>(&sA[buf][0][row][wpos]) = p0;                        \
        *reinterpret_cast<bf16x8*>(&sA[buf][1][row][wpos]) = p1;                        \
        *reinterpret_cast<bf16x8*>(&sA[buf][2][row][wpos]) = p2;                        \
        split8(rb[SL][0], rb[SL][1], p0, p1, p2);                                       \
        *reinterpret_cast<bf16x8*>(&sB[buf][0][row][wpos]) = p0;                        \
        *reinterpret_cast<bf16x8*>(&sB[buf][1][row][wpos]) = p1;                        \
        *reinterpret_cast<bf16x8*>(&sB[buf][2][row][wpos]) = p2;                        \
    } while (0)
#define STEP5(CUR, s_)                                                                  \
    do {                                                                                \
        LSTORE5(1 - CUR, 1 - CUR);                                                      \
        GLOAD5(1 - CUR, min((s_) + 3, ns - 1));                                         \
        bf16x8 fa[3][2], fb[3][2];                                                      \
        _Pragma("unroll") for (int p = 0; p < 3; ++p)                                   \
            _Pragma("unroll") for (int t = 0; t < 2; ++t) {                             \
                fa[p][t] = *reinterpret_cast<const bf16x8*>(&sA[CUR][p][wm + 32 * t + i][swz5(wm + 32 * t + i, h)]); \
                fb[p][t] = *reinterpret_cast<const bf16x8*>(&sB[CUR][p][wn + 32 * t + i][swz5(wn + 32 * t + i, h)]); \
            }                                                                           \
        _Pragma("unroll") for (int t = 0; t < 6; ++t)                                   \
            _Pragma("unroll") for (int a = 0; a < 2; ++a)                               \
                _Pragma("unroll") for (int b = 0; b < 2; ++b)                           \
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[t]][a], fb[PB[t]][b], acc[a][b], 0, 0, 0); \
        __syncthreads();                                                                \
    } while (0)
    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
    const int ns = K / 16;
    GLOAD5(0, 0);
    GLOAD5(1, 1);
    LSTORE5(0, 0);
    GLOAD5(0, min(2, ns - 1));
    __syncthreads();
    for (int s = 0; s < ns; s += 2) {
        STEP5(0, s);
        STEP5(1, s + 1);
    }
#undef GLOAD5
#undef LSTORE5
#undef STEP5
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                C[(long long)m * N + n0 + wn + 32 * b + i] = acc[a][b][r];
            }
}

// plain fp32 FMA reference on the device for a sample of rows (one thread per output), summed in k order
__global__ void ref_rows_kernel(const float* A, const float* B, float* Cr, double* Cd, int N, int K, const int* rows, int nrows) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)nrows * N) return;
    const int r = rows[e / N], n = (int)(e % N);
    float s = 0.f;
    double d = 0.0;
    for (int k = 0; k < K; ++k) {
        s = fmaf(A[(long long)r * K + k], B[(long long)n * K + k], s);
        d += (double)A[(long long)r * K + k] * (double)B[(long long)n * K + k];
    }
    Cr[e] = s;
    Cd[e] = d;
}

int main(int argc, char** argv) {
    int M = 35712, N = 2048, K = 512;
    if (argc >= 4) { M = atoi(argv[1]); N = atoi(argv[2]); K = atoi(argv[3]); }
    if (M % BM || N % BN || K % BK) { printf("M %% 128, N %% 128, K %% 32 must be 0\n"); return 1; }
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((st >> 11) * (1.0 / 9007199254740992.0)); };
    for (auto& v : hA) { const float u = rnd(), w = rnd(); v = sqrtf(-2.f * logf(u + 1e-12f)) * cosf(6.2831853f * w); }   // N(0,1)
    for (auto& v : hB) { const float u = rnd(), w = rnd(); v = 0.05f * sqrtf(-2.f * logf(u + 1e-12f)) * cosf(6.2831853f * w); }
    float *dA, *dB, *dC, *dCr;
    double* dCd;
    int* dRows;
    const int nrows = 64;
    std::vector<int> rows(nrows);
    for (int r = 0; r < nrows; ++r) rows[r] = (int)((long long)r * (M - 1) / (nrows - 1));
    CHECK(hipMalloc(&dA, sizeof(float) * hA.size()));
    CHECK(hipMalloc(&dB, sizeof(float) * hB.size()));
    CHECK(hipMalloc(&dC, sizeof(float) * (size_t)M * N));
    CHECK(hipMalloc(&dCr, sizeof(float) * (size_t)nrows * N));
    CHECK(hipMalloc(&dCd, sizeof(double) * (size_t)nrows * N));
    CHECK(hipMalloc(&dRows, sizeof(int) * nrows));
    CHECK(hipMemcpy(dA, hA.data(), sizeof(float) * hA.size(), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dB, hB.data(), sizeof(float) * hB.size(), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dRows, rows.data(), sizeof(int) * nrows, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(ref_rows_kernel, dim3((nrows * N + 255) / 256), dim3(256), 0, 0, dA, dB, dCr, dCd, N, K, dRows, nrows);
    CHECK(hipDeviceSynchronize());
    std::vector<float> hCr((size_t)nrows * N), hC((size_t)nrows * N);
    std::vector<double> hCd((size_t)nrows * N);
    CHECK(hipMemcpy(hCr.data(), dCr, sizeof(float) * hCr.size(), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hCd.data(), dCd, sizeof(double) * hCd.size(), hipMemcpyDeviceToHost));
    double scale = 0.0, e32 = 0.0;
    for (size_t e = 0; e < hCd.size(); ++e) { scale = fmax(scale, fabs(hCd[e])); e32 = fmax(e32, fabs((double)hCr[e] - hCd[e])); }
    printf("M %d N %d K %d   fp32 FMA chain vs fp64: max err / max|C| = %.3e\n", M, N, K, e32 / scale);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const dim3 grid((M / BM) * (N / BN));
    for (int variant = 0; variant < 6; ++variant) {
        auto launch = [&]() {
            if (variant == 0) hipLaunchKernelGGL(gemm_bf16x3_kernel<6>, grid, dim3(256), 0, 0, dA, dB, dC, M, N, K);
            else if (variant == 1) hipLaunchKernelGGL(gemm_bf16x3_kernel<3>, grid, dim3(256), 0, 0, dA, dB, dC, M, N, K);
            else if (variant == 2) hipLaunchKernelGGL(gemm_bf16x3_v2_kernel, grid, dim3(256), 0, 0, dA, dB, dC, M, N, K);
            else if (variant == 3) hipLaunchKernelGGL(gemm_bf16x3_v3_kernel, grid, dim3(512), 0, 0, dA, dB, dC, M, N, K);
            else if (variant == 4) hipLaunchKernelGGL(gemm_bf16x3_v4_kernel, dim3(((M + 255) / 256) * (N / 128)), dim3(512), 0, 0, dA, dB, dC, M, N, K);
            else hipLaunchKernelGGL(gemm_bf16x3_v5_kernel, grid, dim3(256), 0, 0, dA, dB, dC, M, N, K);
        };
        for (int w = 0; w < 20; ++w) launch();
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        const int reps = 20;
        for (int w = 0; w < reps; ++w) launch();
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipDeviceSynchronize());
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        double err = 0.0;
        for (int r = 0; r < nrows; ++r) {
            CHECK(hipMemcpy(hC.data() + (size_t)r * N, dC + (size_t)rows[r] * N, sizeof(float) * N, hipMemcpyDeviceToHost));
        }
        for (size_t e = 0; e < hCd.size(); ++e) err = fmax(err, fabs((double)hC[e] - hCd[e]));
        printf("%s%d piece products: %8.1f us  %7.1f TFLOP/s (fp32-equivalent)   vs fp64: max err / max|C| = %.3e\n",
               variant == 2 ? "v2 " : (variant == 3 ? "v3 " : (variant == 4 ? "v4 " : (variant == 5 ? "v5 " : ""))), variant == 1 ? 3 : 6, us, 2.0 * M * N * K / us * 1e-6, err / scale);
    }
    return 0;
}
