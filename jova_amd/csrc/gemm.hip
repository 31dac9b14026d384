// fp32 MFMA GEMM for the dense contractions of the DTI hot path (view projections, PCNN, QKV, attention score / PV
// products, out-proj, FFN, FC head; forward, data-gradient and weight-gradient forms), gfx950.
//
//   C[z][m][n] = epilogue( alpha * sum_k A[z](m,k) * B[z](k,n) )
//   A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]; one of each stride pair must be 1
//   (K-contiguous "x[m][k]" / "W[n][k]" operands, or row-contiguous operands for the transposed products).
//   epilogue: + bias[n], + R[m*ldr + n] (residual), ReLU; or split-K partial sums added with float atomics.
//   z = z1 * nb2 + z2 selects the batch with two independent strides (needed for the [S,B,3L] QKV layout).
//
// v_mfma_f32_32x32x2_f32 is exact fp32 (k-ordered fma chain) at 64 FLOP/clk/SIMD, 64 cycles per instruction, so the
// kernel is MFMA-issue bound as long as the staging is hidden: 128x128x16 tiles, 4 waves as 2x2 of 64x64, register
// prefetch of the next k-tile under the current tile's 32 MFMAs, several workgroups per CU for latency hiding.
#include "common.h"
#include "jova_hip.h"

#define BM 128
#define BN 128
#define BK 16
#define LDS_KC (BK + 1)      // [row][k] image, odd stride
#define LDS_RC (BM + 4)      // [k][row] image, 16-byte aligned rows

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    const float* R;
    int M, N, K;
    long long sam, sak, sbk, sbn;
    long long ldc, ldr;
    int nb2;
    long long sA1, sA2, sB1, sB2, sC1, sC2;
    float alpha;
    int relu;
    int splitk;
    int vecA, vecB;
};

__device__ __forceinline__ float4 load4(const float* __restrict__ p, long long stride, int nvalid, bool vec) {
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vec && nvalid >= 4) return *reinterpret_cast<const float4*>(p);
    if (nvalid > 0) r.x = p[0];
    if (nvalid > 1) r.y = p[stride];
    if (nvalid > 2) r.z = p[2 * stride];
    if (nvalid > 3) r.w = p[3 * stride];
    return r;
}

// Operand tile loader.  KC = true: the operand is K-contiguous (element (row,k) at base[row*srow + k]);
// thread t covers rows t/4 and t/4+64, k-quad (t%4)*4.  KC = false: row-contiguous (element at base[k*sk + row]);
// thread t covers k = t/32 and t/32+8, row-quad (t%32)*4.
template <bool KC>
__device__ __forceinline__ void load_operand(const float* __restrict__ base, long long srow, long long sk, int row0,
                                             int nrows, int k0, int kend, bool vec, int tid, float4 (&v)[2]) {
    if (KC) {
        const int kq = (tid & 3) * 4;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r = (tid >> 2) + 64 * p;
            const int row = row0 + r;
            const int nv = row < nrows ? kend - (k0 + kq) : 0;
            v[p] = load4(base + (long long)row * srow + k0 + kq, 1, nv, vec);
        }
    } else {
        const int rq = (tid & 31) * 4;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int k = k0 + (tid >> 5) + 8 * p;
            const int nv = k < kend ? nrows - (row0 + rq) : 0;
            v[p] = load4(base + (long long)k * sk + row0 + rq, 1, nv, vec);
        }
    }
}

template <bool KC>
__device__ __forceinline__ void store_operand(float* __restrict__ S, int tid, const float4 (&v)[2]) {
    if (KC) {
        const int kq = (tid & 3) * 4;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float* d = S + ((tid >> 2) + 64 * p) * LDS_KC + kq;
            d[0] = v[p].x; d[1] = v[p].y; d[2] = v[p].z; d[3] = v[p].w;
        }
    } else {
        const int rq = (tid & 31) * 4;
#pragma unroll
        for (int p = 0; p < 2; ++p)
            *reinterpret_cast<float4*>(S + ((tid >> 5) + 8 * p) * LDS_RC + rq) = v[p];
    }
}

template <bool KC>
__device__ __forceinline__ float frag(const float* __restrict__ S, int row, int k) {
    return KC ? S[row * LDS_KC + k] : S[k * LDS_RC + row];
}

template <bool A_KC, bool B_KC>
__global__ void __launch_bounds__(256)
gemm_f32_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[A_KC ? BM * LDS_KC : BK * LDS_RC];
    __shared__ __attribute__((aligned(16))) float Bs[B_KC ? BN * LDS_KC : BK * LDS_RC];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int z = blockIdx.z / g.splitk, ks = blockIdx.z % g.splitk;
    const int z1 = z / g.nb2, z2 = z % g.nb2;
    const float* A = g.A + z1 * g.sA1 + z2 * g.sA2;
    const float* B = g.B + z1 * g.sB1 + z2 * g.sB2;
    // K range of this split (multiples of BK)
    const int ktiles = (g.K + BK - 1) / BK;
    const int per = (ktiles + g.splitk - 1) / g.splitk;
    const int kbeg = ks * per * BK;
    const int kend = min(g.K, (ks + 1) * per * BK);
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int i = lane & 31, h = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float4 va[2], vb[2];
    if (kbeg < kend) {
        load_operand<A_KC>(A, g.sam, g.sak, m0, g.M, kbeg, kend, g.vecA, tid, va);
        load_operand<B_KC>(B, g.sbn, g.sbk, n0, g.N, kbeg, kend, g.vecB, tid, vb);
    }
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();
        store_operand<A_KC>(As, tid, va);
        store_operand<B_KC>(Bs, tid, vb);
        __syncthreads();
        if (k0 + BK < kend) {
            load_operand<A_KC>(A, g.sam, g.sak, m0, g.M, k0 + BK, kend, g.vecA, tid, va);
            load_operand<B_KC>(B, g.sbn, g.sbk, n0, g.N, k0 + BK, kend, g.vecB, tid, vb);
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float a0 = frag<A_KC>(As, wm + i, kk + h);
            const float a1 = frag<A_KC>(As, wm + 32 + i, kk + h);
            const float b0 = frag<B_KC>(Bs, wn + i, kk + h);
            const float b1 = frag<B_KC>(Bs, wn + 32 + i, kk + h);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }

    float* C = g.C + z1 * g.sC1 + z2 * g.sC2;
    const float* R = g.R ? g.R + z1 * g.sC1 + z2 * g.sC2 : nullptr;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int n = n0 + wn + 32 * b + i;
            if (n >= g.N) continue;
            const float bias = (g.bias && ks == 0) ? g.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * a + mfma32_row(r, lane);
                if (m >= g.M) continue;
                float v = g.alpha * acc[a][b][r] + bias;
                if (g.splitk > 1) {
                    atomicAdd(&C[(long long)m * g.ldc + n], v);
                } else {
                    if (R) v += R[(long long)m * g.ldr + n];
                    if (g.relu) v = fmaxf(v, 0.f);
                    C[(long long)m * g.ldc + n] = v;
                }
            }
        }
}

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

extern "C" int jova_gemm_f32(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn,
                             float* C, int64_t ldc, int M, int N, int K, float alpha, const float* bias,
                             const float* residual, int64_t ldr, int relu, int splitk, int nb1, int nb2, int64_t sA1,
                             int64_t sA2, int64_t sB1, int64_t sB2, int64_t sC1, int64_t sC2, void* stream) {
    JOVA_CHECK_ARG(A && B && C, "gemm: null pointer");
    JOVA_CHECK_ARG(M >= 0 && N >= 0 && K >= 0 && nb1 >= 1 && nb2 >= 1, "gemm: bad shape M=%d N=%d K=%d", M, N, K);
    JOVA_CHECK_ARG(sam == 1 || sak == 1, "gemm: A needs a unit stride (sam=%lld sak=%lld)", (long long)sam, (long long)sak);
    JOVA_CHECK_ARG(sbk == 1 || sbn == 1, "gemm: B needs a unit stride (sbk=%lld sbn=%lld)", (long long)sbk, (long long)sbn);
    JOVA_CHECK_ARG(splitk >= 1 && (splitk == 1 || (!residual && !relu)), "gemm: split-K excludes residual/relu");
    if (M == 0 || N == 0) return JOVA_OK;
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.R = residual;
    g.M = M; g.N = N; g.K = K;
    g.sam = sam; g.sak = sak; g.sbk = sbk; g.sbn = sbn; g.ldc = ldc; g.ldr = ldr;
    g.nb2 = nb2; g.sA1 = sA1; g.sA2 = sA2; g.sB1 = sB1; g.sB2 = sB2; g.sC1 = sC1; g.sC2 = sC2;
    g.alpha = alpha; g.relu = relu; g.splitk = splitk;
    const bool a_kc = (sak == 1), b_kc = (sbk == 1);
    const int64_t a_ld = a_kc ? sam : sak, b_ld = b_kc ? sbn : sbk;
    g.vecA = aligned16(A) && (a_ld % 4 == 0) && (sA1 % 4 == 0) && (sA2 % 4 == 0);
    g.vecB = aligned16(B) && (b_ld % 4 == 0) && (sB1 % 4 == 0) && (sB2 % 4 == 0);
    const long long gz = (long long)nb1 * nb2 * splitk;
    JOVA_CHECK_ARG(gz <= 65535, "gemm: batch*splitk %lld exceeds grid.z", gz);
    dim3 grid(cdiv(M, BM), cdiv(N, BN), (unsigned)gz);
    JOVA_CHECK_ARG(grid.y <= 65535, "gemm: N too large");
    hipStream_t s = (hipStream_t)stream;
    if (a_kc && b_kc) hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, dim3(256), 0, s, g);
    else if (a_kc && !b_kc) hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, dim3(256), 0, s, g);
    else if (!a_kc && b_kc) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, dim3(256), 0, s, g);
    JOVA_CHECK_LAUNCH("gemm_f32");
    return JOVA_OK;
}
