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
// kernel is MFMA-issue bound as long as the staging is hidden: 128x128x32 tiles, 4 waves as 2x2 of 64x64, register
// prefetch of the next k-tile under the current tile's 64 MFMAs, several workgroups per CU for latency hiding.
#include "common.h"
#include "jova_hip.h"

#define BM 128
#define BN 128
#define BK 32
#define LDS_KC (BK + 1)      // [row][k] image, odd stride: conflict-free ds_read_b32 down a column
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

// Branch-free guarded loads: out-of-range elements read the (always valid) tensor base, so every global load is
// unconditional and all of a tile's loads are in flight together (a branch per element would put an
// s_waitcnt vmcnt(0) behind each load: guide section 5, "Three .s-level traps" (c)).
// A staged quad: raw loaded data + how many of its 4 elements are inside the operand.  The zeroing select is applied
// when the quad is written to LDS (one iteration later), never right after the load, so the loads stay in flight
// under the MFMAs of the current tile.
struct Quad {
    float4 v;
    int nvalid;
};

template <bool VEC>
__device__ __forceinline__ Quad load4(const float* __restrict__ base, long long off, int nvalid) {
    Quad q;
    q.nvalid = nvalid;
    if (VEC) {      // VEC kernels are only launched when a quad is all-in or all-out
        q.v = *reinterpret_cast<const float4*>(base + (nvalid >= 4 ? off : 0));
    } else {
        q.v.x = base[nvalid > 0 ? off : 0];
        q.v.y = base[nvalid > 1 ? off + 1 : 0];
        q.v.z = base[nvalid > 2 ? off + 2 : 0];
        q.v.w = base[nvalid > 3 ? off + 3 : 0];
    }
    return q;
}

__device__ __forceinline__ float4 masked(const Quad& q) {
    float4 r;
    r.x = q.nvalid > 0 ? q.v.x : 0.f;
    r.y = q.nvalid > 1 ? q.v.y : 0.f;
    r.z = q.nvalid > 2 ? q.v.z : 0.f;
    r.w = q.nvalid > 3 ? q.v.w : 0.f;
    return r;
}

// Operand tile loader.  KC = true: the operand is K-contiguous (element (row,k) at base[row*srow + k]);
// thread t covers rows t/8 + 32p (p < 4), k-quad (t%8)*4: 8 lanes read one full 128-byte line of a row.
// KC = false: row-contiguous (element at base[k*sk + row]); thread t covers k = t/32 + 8p, row-quad (t%32)*4.
template <bool KC, bool VEC>
__device__ __forceinline__ void load_operand(const float* __restrict__ base, long long srow, long long sk, int row0,
                                             int nrows, int k0, int kend, int tid, Quad (&v)[4]) {
    if (KC) {
        const int kq = (tid & 7) * 4;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = row0 + (tid >> 3) + 32 * p;
            const int nv = row < nrows ? kend - (k0 + kq) : 0;
            v[p] = load4<VEC>(base, (long long)row * srow + k0 + kq, nv);
        }
    } else {
        const int rq = (tid & 31) * 4;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int k = k0 + (tid >> 5) + 8 * p;
            const int nv = k < kend ? nrows - (row0 + rq) : 0;
            v[p] = load4<VEC>(base, (long long)k * sk + row0 + rq, nv);
        }
    }
}

template <bool KC>
__device__ __forceinline__ void store_operand(float* __restrict__ S, int tid, const Quad (&v)[4]) {
    if (KC) {
        const int kq = (tid & 7) * 4;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float4 m = masked(v[p]);
            float* d = S + ((tid >> 3) + 32 * p) * LDS_KC + kq;
            d[0] = m.x; d[1] = m.y; d[2] = m.z; d[3] = m.w;
        }
    } else {
        const int rq = (tid & 31) * 4;
#pragma unroll
        for (int p = 0; p < 4; ++p)
            *reinterpret_cast<float4*>(S + ((tid >> 5) + 8 * p) * LDS_RC + rq) = masked(v[p]);
    }
}

template <bool KC>
__device__ __forceinline__ float frag(const float* __restrict__ S, int row, int k) {
    return KC ? S[row * LDS_KC + k] : S[k * LDS_RC + row];
}

template <bool A_KC, bool B_KC, bool VA, bool VB>
__global__ void __launch_bounds__(256, 3)      // 3 workgroups per CU: <= 168 VGPR+AGPR
gemm_f32_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[A_KC ? BM * LDS_KC : BK * LDS_RC];
    __shared__ __attribute__((aligned(16))) float Bs[B_KC ? BN * LDS_KC : BK * LDS_RC];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // Tile order: n fastest (the tiles that share an A row-panel run back to back), and the linear tile id is
    // remapped so that each XCD (workgroups b, b+8, ... share one) walks a contiguous chunk of that order: the
    // A panel is then fetched once into ONE L2 instead of eight.
    const int ntn = (g.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    int pid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = pid & 7, loc = pid >> 3;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int m0 = (pid / ntn) * BM, n0 = (pid % ntn) * BN;
    const int z = blockIdx.z / g.splitk, ks = blockIdx.z % g.splitk;
    const int z1 = z / g.nb2, z2 = z % g.nb2;
    const float* A = g.A + z1 * g.sA1 + z2 * g.sA2;
    const float* B = g.B + z1 * g.sB1 + z2 * g.sB2;
    // K range of this split (multiples of BK)
    const int ktiles = (g.K + BK - 1) / BK;
    const int per = (ktiles + g.splitk - 1) / g.splitk;
    const int kbeg = ks * per * BK;
    const int kend = min(g.K, (ks + 1) * per * BK);
    const int nt = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int i = lane & 31, h = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // Register prefetch: tile t+1 is loaded into registers while tile t is multiplied out of LDS.  (Measured on
    // MI355X: double-buffered LDS with one barrier per tile and hand-pipelined ds_read_b128 fragments were both
    // slower than this form at the 2-3 workgroups/CU it runs at -- profiles/r01_gemm_notes.md.)
    Quad va[4], vb[4];
    if (nt > 0) {
        load_operand<A_KC, VA>(A, g.sam, g.sak, m0, g.M, kbeg, kend, tid, va);
        load_operand<B_KC, VB>(B, g.sbn, g.sbk, n0, g.N, kbeg, kend, tid, vb);
    }
    for (int t = 0; t < nt; ++t) {
        __syncthreads();
        store_operand<A_KC>(As, tid, va);
        store_operand<B_KC>(Bs, tid, vb);
        __syncthreads();
        if (t + 1 < nt) {
            load_operand<A_KC, VA>(A, g.sam, g.sak, m0, g.M, kbeg + (t + 1) * BK, kend, tid, va);
            load_operand<B_KC, VB>(B, g.sbn, g.sbk, n0, g.N, kbeg + (t + 1) * BK, kend, tid, vb);
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
            const bool nok = n < g.N;
            const int nc = nok ? n : g.N - 1;
            const float bias = (g.bias && ks == 0) ? g.bias[nc] : 0.f;
            const int mbase = m0 + wm + 32 * a;
            if (g.splitk > 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mbase + mfma32_row(r, lane);
                    if (m < g.M && nok) atomicAdd(&C[(long long)m * g.ldc + n], g.alpha * acc[a][b][r] + bias);
                }
            } else {
                float rv[16];
                if (R) {            // unconditional (clamped) residual loads: all 16 in flight together
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = min(mbase + mfma32_row(r, lane), g.M - 1);
                        rv[r] = R[(long long)m * g.ldr + nc];
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) rv[r] = 0.f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mbase + mfma32_row(r, lane);
                    float v = g.alpha * acc[a][b][r] + bias + rv[r];
                    if (g.relu) v = fmaxf(v, 0.f);
                    if (m < g.M && nok) C[(long long)m * g.ldc + n] = v;
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
    // 16-byte loads need aligned quads that are entirely inside or outside the operand: the contiguous extent
    // (K for a K-contiguous operand, M / N for a row-contiguous one) must be a multiple of 4 as well
    const bool va = aligned16(A) && (a_ld % 4 == 0) && (sA1 % 4 == 0) && (sA2 % 4 == 0) && ((a_kc ? K : M) % 4 == 0);
    const bool vb = aligned16(B) && (b_ld % 4 == 0) && (sB1 % 4 == 0) && (sB2 % 4 == 0) && ((b_kc ? K : N) % 4 == 0);
    g.vecA = va; g.vecB = vb;
    const long long gz = (long long)nb1 * nb2 * splitk;
    JOVA_CHECK_ARG(gz <= 65535, "gemm: batch*splitk %lld exceeds grid.z", gz);
    dim3 grid((unsigned)((long long)cdiv(M, BM) * cdiv(N, BN)), 1, (unsigned)gz);
    hipStream_t s = (hipStream_t)stream;
#define GEMM_LAUNCH(AK, BK_, VA_, VB_) hipLaunchKernelGGL((gemm_f32_kernel<AK, BK_, VA_, VB_>), grid, dim3(256), 0, s, g)
#define GEMM_VEC(AK, BK_)                                                     \
    do {                                                                      \
        if (va && vb) GEMM_LAUNCH(AK, BK_, true, true);                       \
        else if (va) GEMM_LAUNCH(AK, BK_, true, false);                       \
        else if (vb) GEMM_LAUNCH(AK, BK_, false, true);                       \
        else GEMM_LAUNCH(AK, BK_, false, false);                              \
    } while (0)
    if (a_kc && b_kc) GEMM_VEC(true, true);
    else if (a_kc && !b_kc) GEMM_VEC(true, false);
    else if (!a_kc && b_kc) GEMM_VEC(false, true);
    else GEMM_VEC(false, false);
    JOVA_CHECK_LAUNCH("gemm_f32");
    return JOVA_OK;
}
