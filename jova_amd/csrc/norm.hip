// Normalisation / softmax / dropout / column-reduction kernels of the DTI hot path (all HBM-bound), gfx950.
#include "common.h"
#include "jova_hip.h"

// ------------------------------------------------------------------------------------------------ column sums
// Column tile of 64 x row strip per workgroup; thread (ty, tx) walks rows ty, ty+4, ...; partials meet in LDS,
// one fp64 atomic per column per workgroup.  MODE 0: sum x, sum x^2.  MODE 1: sum g, sum g*xhat (BN backward),
// with g = dy masked by (y > 0) when y is given.  MODE 2: sum x only (bias gradients) into a float buffer.
template <int MODE>
__global__ void __launch_bounds__(256)
colreduce_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ y,
                 const float* __restrict__ mean_rstd, long long n_rows, int c, long long ld, int rows_per_block,
                 double* __restrict__ out_d, float* __restrict__ out_f) {
    __shared__ double red[2][4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + tx;
    const long long r0 = (long long)blockIdx.y * rows_per_block;
    const long long r1 = min(n_rows, r0 + rows_per_block);
    double s0 = 0.0, s1 = 0.0;
    if (col < c) {
        float mu = 0.f, rs = 0.f;
        if (MODE == 1) { mu = mean_rstd[col]; rs = mean_rstd[c + col]; }
        // fp64 accumulation element by element: sum(x^2)/n - mean^2 cancels catastrophically in fp32 when
        // |mean| >> std (e.g. the 279-segment view sums that feed JointAttention's BatchNorm)
        // four rows in flight per thread (independent loads first, fp64 adds after)
        auto add = [&](float xv, float gv, float yv) {
            if (MODE == 0) {
                const double v = (double)xv;
                s0 += v;
                s1 += v * v;
            } else if (MODE == 1) {
                const float g = (y && !(yv > 0.f)) ? 0.f : gv;
                s0 += (double)g;
                s1 += (double)g * (double)((xv - mu) * rs);
            } else {
                s0 += (double)xv;
            }
        };
        long long r = r0 + ty;
        for (; r + 12 < r1; r += 16) {
            float xv[4], gv[4], yv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long o = (r + 4 * u) * ld + col;
                xv[u] = x[o];
                gv[u] = MODE == 1 ? dy[o] : 0.f;
                yv[u] = (MODE == 1 && y) ? y[o] : 1.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) add(xv[u], gv[u], yv[u]);
        }
        for (; r < r1; r += 4) {
            const long long o = r * ld + col;
            add(x[o], MODE == 1 ? dy[o] : 0.f, (MODE == 1 && y) ? y[o] : 1.f);
        }
    }
    red[0][ty][tx] = s0;
    red[1][ty][tx] = s1;
    __syncthreads();
    if (ty == 0 && col < c) {
        const double a = (red[0][0][tx] + red[0][1][tx]) + (red[0][2][tx] + red[0][3][tx]);
        const double b = (red[1][0][tx] + red[1][1][tx]) + (red[1][2][tx] + red[1][3][tx]);
        if (MODE == 2) {
            atomicAdd(&out_f[col], (float)a);
        } else {
            atomicAdd(&out_d[col], a);
            atomicAdd(&out_d[c + col], b);
        }
    }
}

static inline void colreduce_grid(long long n_rows, int c, dim3& grid, int& rpb) {
    const int cb = cdiv(c, 64);
    int rb = (int)(2048 / cb);
    if (rb < 1) rb = 1;
    // at least 16 rows per workgroup (4 row lanes x 4 rows in flight); small tensors (the [N atoms, 64] BatchNorms of the
    // graph stack) then spread over ~n_rows / 16 workgroups instead of 16 workgroups walking 256 rows each (32 us)
    long long want = (n_rows + 15) / 16;
    if (want < 1) want = 1;
    if (rb > want) rb = (int)want;
    rpb = (int)((n_rows + rb - 1) / rb);
    if (rpb < 1) rpb = 1;
    grid = dim3(cb, (unsigned)((n_rows + rpb - 1) / rpb));
}

// ------------------------------------------------------------------------------------------------ BatchNorm1d
// stats (fp64 [2][C]: sum, sum of squares) -> mean_rstd (fp32 [2][C]); running stats updated like nn.BatchNorm1d
// (momentum, unbiased variance); then y = (x - mean) * rstd * gamma + beta, optional ReLU.
__global__ void bn_finalize_kernel(const double* __restrict__ stats, long long n_rows, int c, float eps,
                                   float momentum, float* __restrict__ mean_rstd, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, long long* __restrict__ num_batches) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col == 0 && num_batches) *num_batches += 1;
    if (col >= c) return;
    const double n = (double)n_rows;
    const double mean = stats[col] / n;
    double var = stats[c + col] / n - mean * mean;
    if (var < 0.0) var = 0.0;
    mean_rstd[col] = (float)mean;
    mean_rstd[c + col] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unb = n_rows > 1 ? var * n / (n - 1.0) : var;
        running_mean[col] = (1.f - momentum) * running_mean[col] + momentum * (float)mean;
        running_var[col] = (1.f - momentum) * running_var[col] + momentum * (float)unb;
    }
}

__global__ void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean_rstd,
                                const float* __restrict__ gamma, const float* __restrict__ beta, long long n_rows,
                                int c, int relu, float* __restrict__ y) {
    const long long total = n_rows * c;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const int col = (int)(e % c);
        float v = (x[e] - mean_rstd[col]) * mean_rstd[c + col] * gamma[col] + beta[col];
        if (relu) v = fmaxf(v, 0.f);
        y[e] = v;
    }
}

// dx = gamma * rstd * (g - sum_g / n - xhat * sum_gxhat / n);  dgamma = sum_gxhat, dbeta = sum_g
__global__ void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                    const float* __restrict__ y, const float* __restrict__ mean_rstd,
                                    const float* __restrict__ gamma, const double* __restrict__ sums, long long n_rows,
                                    int c, int mask_input, float* __restrict__ dx, float* __restrict__ dgamma,
                                    float* __restrict__ dbeta) {
    const long long total = n_rows * c;
    const float inv_n = 1.f / (float)n_rows;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const int col = (int)(e % c);
        float g = dy[e];
        if (y && !(y[e] > 0.f)) g = 0.f;
        const float rs = mean_rstd[c + col];
        const float xv = x[e];
        const float xhat = (xv - mean_rstd[col]) * rs;
        const float sg = (float)sums[col], sgx = (float)sums[c + col];
        float d = gamma[col] * rs * (g - sg * inv_n - xhat * sgx * inv_n);
        if (mask_input && !(xv > 0.f)) d = 0.f;      // x is a ReLU output: its backward rides along (dx -> dz)
        dx[e] = d;
        if (e < c) {       // first row's threads also publish the parameter gradients (accumulating)
            dbeta[col] += sg;
            dgamma[col] += sgx;
        }
    }
}

// ------------------------------------------------------------------------------------------------ LayerNorm
// One wave per row.  y = LN(x + r) * gamma + beta; saves mean / rstd per row.  Row length d <= 64 * LN_MAXV.
// VEC4 (d % 4 == 0, 16-byte aligned rows): lane l owns columns 4 l + 256 k .. + 3 and moves them as float4.
#define LN_MAXV 32
template <int LNV, bool VEC4>
__device__ __forceinline__ int ln_col(int lane, int k) { return VEC4 ? 4 * lane + 256 * (k >> 2) + (k & 3) : lane + 64 * k; }

template <int LNV, bool VEC4>
__device__ __forceinline__ void ln_load(const float* __restrict__ p, int d, int lane, float* v) {
    if (VEC4) {
#pragma unroll
        for (int k4 = 0; k4 < LNV / 4; ++k4) {
            const int c = 4 * lane + 256 * k4;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < d) t = *reinterpret_cast<const float4*>(p + c);
            v[4 * k4] = t.x; v[4 * k4 + 1] = t.y; v[4 * k4 + 2] = t.z; v[4 * k4 + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < LNV; ++k) {
            const int c = lane + 64 * k;
            v[k] = c < d ? p[c] : 0.f;
        }
    }
}
template <int LNV, bool VEC4>
__device__ __forceinline__ void ln_store(float* __restrict__ p, int d, int lane, const float* v) {
    if (VEC4) {
#pragma unroll
        for (int k4 = 0; k4 < LNV / 4; ++k4) {
            const int c = 4 * lane + 256 * k4;
            if (c < d) *reinterpret_cast<float4*>(p + c) = make_float4(v[4 * k4], v[4 * k4 + 1], v[4 * k4 + 2], v[4 * k4 + 3]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < LNV; ++k) {
            const int c = lane + 64 * k;
            if (c < d) p[c] = v[k];
        }
    }
}

template <int LNV, bool VEC4>
__global__ void __launch_bounds__(256)
layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ r, const float* __restrict__ gamma,
                     const float* __restrict__ beta, long long n_rows, int d, float eps, float* __restrict__ y,
                     float* __restrict__ mean_rstd) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + wave;
    if (row >= n_rows) return;
    float v[LNV], t[LNV];
    ln_load<LNV, VEC4>(x + row * d, d, lane, v);
    if (r) {
        ln_load<LNV, VEC4>(r + row * d, d, lane, t);
#pragma unroll
        for (int k = 0; k < LNV; ++k) v[k] += t[k];
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < LNV; ++k) s += v[k];                     // (columns past d hold zeros)
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < LNV; ++k) {
        if (ln_col<LNV, VEC4>(lane, k) < d) { const float u = v[k] - mean; q += u * u; }
    }
    const float rstd = 1.f / sqrtf(wave_sum(q) / (float)d + eps);
    float g[LNV], b[LNV];
    ln_load<LNV, VEC4>(gamma, d, lane, g);
    ln_load<LNV, VEC4>(beta, d, lane, b);
#pragma unroll
    for (int k = 0; k < LNV; ++k) t[k] = (v[k] - mean) * rstd * g[k] + b[k];
    ln_store<LNV, VEC4>(y + row * d, d, lane, t);
    if (lane == 0) { mean_rstd[2 * row] = mean; mean_rstd[2 * row + 1] = rstd; }
}

// dz (gradient w.r.t. z = x + r) from dy; xhat recomputed from z (re-read: z = x + r).  dgamma / dbeta partials of a
// workgroup's rows are kept in registers, combined over its four waves through LDS and added with float atomics.
template <int LNV, bool VEC4>
__global__ void __launch_bounds__(256)
layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ r, const float* __restrict__ dy,
                     const float* __restrict__ gamma, const float* __restrict__ mean_rstd, long long n_rows, int d,
                     int rows_per_block, float* __restrict__ dz, float* __restrict__ dgamma,
                     float* __restrict__ dbeta) {
    extern __shared__ float lds[];          // [2][4][d]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float pg[LNV], pb[LNV], gm[LNV];
#pragma unroll
    for (int k = 0; k < LNV; ++k) pg[k] = pb[k] = 0.f;
    ln_load<LNV, VEC4>(gamma, d, lane, gm);
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = min(n_rows, r0 + rows_per_block);
    // the next row's three loads are in flight while this row's two wave reductions run (one row per wave at a time left
    // the kernel at 4.6 TB/s: latency-bound)
    float nz[LNV], nt[LNV], ndy[LNV], nmean = 0.f, nrstd = 0.f;
    auto fetch = [&](long long row) {
        if (row < r1) {
            nmean = mean_rstd[2 * row];
            nrstd = mean_rstd[2 * row + 1];
            ln_load<LNV, VEC4>(x + row * d, d, lane, nz);
            if (r) ln_load<LNV, VEC4>(r + row * d, d, lane, nt);
            ln_load<LNV, VEC4>(dy + row * d, d, lane, ndy);
        }
    };
    fetch(r0 + wave);
    for (long long row = r0 + wave; row < r1; row += 4) {
        const float mean = nmean, rstd = nrstd;
        float z[LNV], t[LNV], dyv[LNV];
#pragma unroll
        for (int k = 0; k < LNV; ++k) {
            z[k] = r ? nz[k] + nt[k] : nz[k];
            dyv[k] = ndy[k];
        }
        fetch(row + 4);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < LNV; ++k) {
            const bool ok = ln_col<LNV, VEC4>(lane, k) < d;
            const float xh = ok ? (z[k] - mean) * rstd : 0.f;
            const float g = dyv[k] * gm[k];                     // (zero past d: dy and gamma load as zeros there)
            z[k] = xh;
            t[k] = g;
            s1 += g;
            s2 += g * xh;
            pg[k] += dyv[k] * xh;
            pb[k] += dyv[k];
        }
        s1 = wave_sum(s1) / (float)d;
        s2 = wave_sum(s2) / (float)d;
#pragma unroll
        for (int k = 0; k < LNV; ++k) t[k] = rstd * (t[k] - s1 - z[k] * s2);
        ln_store<LNV, VEC4>(dz + row * d, d, lane, t);
    }
    float* lg = lds;
    float* lb = lds + 4 * d;
#pragma unroll
    for (int k = 0; k < LNV; ++k) {
        const int c = ln_col<LNV, VEC4>(lane, k);
        if (c < d) { lg[wave * d + c] = pg[k]; lb[wave * d + c] = pb[k]; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += 256) {
        atomicAdd(&dgamma[c], (lg[c] + lg[d + c]) + (lg[2 * d + c] + lg[3 * d + c]));
        atomicAdd(&dbeta[c], (lb[c] + lb[d + c]) + (lb[2 * d + c] + lb[3 * d + c]));
    }
}

// ------------------------------------------------------------------------------------------------ softmax
// One wave per row of length n (<= 64 * SM_MAXV), in place capable.
#define SM_MAXV 32          // rows to 2048 elements (long-protein batches: S = 1 + A_max + P_max up to ~1100)
template <int SMV>
__global__ void __launch_bounds__(256)
softmax_fwd_kernel(const float* __restrict__ x, long long n_rows, int n, long long ld, float* __restrict__ y) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + wave;
    if (row >= n_rows) return;
    float v[SMV];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < SMV; ++k) {
        const int c = lane + 64 * k;
        v[k] = c < n ? x[row * ld + c] : -INFINITY;
        m = fmaxf(m, v[k]);
    }
    m = wave_max(m);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < SMV; ++k) {
        const int c = lane + 64 * k;
        v[k] = c < n ? expf(v[k] - m) : 0.f;
        s += v[k];
    }
    const float inv = 1.f / wave_sum(s);
#pragma unroll
    for (int k = 0; k < SMV; ++k) {
        const int c = lane + 64 * k;
        if (c < n) y[row * ld + c] = v[k] * inv;
    }
}

// ds = p * (dp' - sum(dp' * p)), dp' = dp (or dp masked and scaled by the dropout that followed the softmax);
// in place capable (ds may alias dp)
template <int SMV>
__global__ void __launch_bounds__(256)
softmax_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dp, const float* __restrict__ pdrop,
                   float keep_scale, long long n_rows, int n, long long ld, float* __restrict__ ds) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + wave;
    if (row >= n_rows) return;
    float pv[SMV], dv[SMV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < SMV; ++k) {
        const int c = lane + 64 * k;
        pv[k] = dv[k] = 0.f;
        if (c < n) {
            pv[k] = p[row * ld + c];
            dv[k] = dp[row * ld + c];
            // dropout between softmax and its consumer: pdrop = dropout(p) as the forward produced it; an element
            // was kept iff pdrop != 0 (p == 0 elements contribute nothing either way)
            if (pdrop) dv[k] = pdrop[row * ld + c] != 0.f ? dv[k] * keep_scale : 0.f;
            s += pv[k] * dv[k];
        }
    }
    s = wave_sum(s);
#pragma unroll
    for (int k = 0; k < SMV; ++k) {
        const int c = lane + 64 * k;
        if (c < n) ds[row * ld + c] = pv[k] * (dv[k] - s);
    }
}

// ------------------------------------------------------------------------------------------------ dropout
// Philox4x32-10 keyed by (seed), counter = (element / 4, offset): the mask is a pure function of
// (seed, offset, element index), so backward regenerates it instead of storing it.
__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * ctr.x;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * ctr.z;
        ctr = make_uint4((unsigned)(p1 >> 32) ^ ctr.y ^ key.x, (unsigned)p1, (unsigned)(p0 >> 32) ^ ctr.w ^ key.y,
                         (unsigned)p0);
        key.x += 0x9E3779B9u;
        key.y += 0xBB67AE85u;
    }
    return ctr;
}

// y[r][c] = x[r][c] * keep / (1 - p) over a [n_rows, n] view with leading dimension ld (x may alias y)
__global__ void dropout_kernel(const float* __restrict__ x, long long n_rows, int n, long long ld, float p,
                               unsigned long long seed, unsigned long long offset, float* __restrict__ y) {
    const float scale = 1.f / (1.f - p);
    const long long total = n_rows * n;
    const long long quads = (total + 3) / 4;
    for (long long qd = (long long)blockIdx.x * blockDim.x + threadIdx.x; qd < quads;
         qd += (long long)gridDim.x * blockDim.x) {
        const uint4 rnd = philox4x32_10(make_uint4((unsigned)qd, (unsigned)(qd >> 32), (unsigned)offset,
                                                   (unsigned)(offset >> 32)),
                                        make_uint2((unsigned)seed, (unsigned)(seed >> 32)));
        const unsigned rr[4] = {rnd.x, rnd.y, rnd.z, rnd.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long e = qd * 4 + k;
            if (e < total) {
                const long long row = e / n;
                const int c = (int)(e - row * n);
                const float u = (float)(rr[k] >> 8) * (1.f / 16777216.f);
                y[row * ld + c] = u >= p ? x[row * ld + c] * scale : 0.f;
            }
        }
    }
}

// Rows padded to a multiple of four floats (the attention probabilities: ld = S rounded up) and 16-byte aligned: the
// padded tensor is one flat run of float4s, no division, 16-byte accesses; the pad columns are scaled / zeroed along
// (nobody reads them).  The element -> random number map differs from the generic kernel's; forward and backward of
// one dropout always take the same kernel (same shape), which is all that has to agree.
__global__ void __launch_bounds__(256)
dropout_v4_kernel(const float* __restrict__ x, long long quads, float p, unsigned long long seed,
                  unsigned long long offset, float* __restrict__ y) {
    const float scale = 1.f / (1.f - p);
    for (long long qd = (long long)blockIdx.x * blockDim.x + threadIdx.x; qd < quads;
         qd += (long long)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(x)[qd];
        const uint4 rnd = philox4x32_10(make_uint4((unsigned)qd, (unsigned)(qd >> 32), (unsigned)offset,
                                                   (unsigned)(offset >> 32)),
                                        make_uint2((unsigned)seed, (unsigned)(seed >> 32)));
        const float k = 1.f / 16777216.f;
        float4 o;
        o.x = (float)(rnd.x >> 8) * k >= p ? v.x * scale : 0.f;
        o.y = (float)(rnd.y >> 8) * k >= p ? v.y * scale : 0.f;
        o.z = (float)(rnd.z >> 8) * k >= p ? v.z * scale : 0.f;
        o.w = (float)(rnd.w >> 8) * k >= p ? v.w * scale : 0.f;
        reinterpret_cast<float4*>(y)[qd] = o;
    }
}

// ================================================================================================ C ABI
extern "C" int jova_colstats(const float* x, int64_t n_rows, int c, double* stats, void* stream) {
    JOVA_CHECK_ARG(x && stats && n_rows >= 0 && c >= 1, "colstats: bad argument");
    hipStream_t s = (hipStream_t)stream;
    JOVA_ZERO(stats, sizeof(double) * 2 * c, s);
    if (n_rows == 0) return JOVA_OK;
    dim3 grid;
    int rpb;
    colreduce_grid(n_rows, c, grid, rpb);
    hipLaunchKernelGGL(colreduce_kernel<0>, grid, dim3(256), 0, s, x, nullptr, nullptr, nullptr, (long long)n_rows, c,
                       (long long)c, rpb, stats, nullptr);
    JOVA_CHECK_LAUNCH("colstats");
    return JOVA_OK;
}

extern "C" int jova_colsum_add(const float* x, int64_t n_rows, int c, int64_t ld, float* out, void* stream) {
    JOVA_CHECK_ARG(x && out && n_rows >= 0 && c >= 1 && ld >= c, "colsum_add: bad argument");
    if (n_rows == 0) return JOVA_OK;
    dim3 grid;
    int rpb;
    colreduce_grid(n_rows, c, grid, rpb);
    hipLaunchKernelGGL(colreduce_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, x, nullptr, nullptr, nullptr,
                       (long long)n_rows, c, (long long)ld, rpb, nullptr, out);
    JOVA_CHECK_LAUNCH("colsum_add");
    return JOVA_OK;
}

extern "C" int jova_batchnorm_fwd(const float* x, int64_t n_rows, int c, const double* stats, const float* gamma,
                                  const float* beta, float eps, float momentum, float* running_mean,
                                  float* running_var, int64_t* num_batches, int relu, float* y, float* mean_rstd,
                                  void* stream) {
    JOVA_CHECK_ARG(x && stats && gamma && beta && y && mean_rstd && n_rows >= 1 && c >= 1, "batchnorm_fwd: bad argument");
    JOVA_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "batchnorm_fwd: running stats must come in pairs");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(c, 256)), dim3(256), 0, s, stats, (long long)n_rows, c, eps,
                       momentum, mean_rstd, running_mean, running_var, (long long*)num_batches);
    JOVA_CHECK_LAUNCH("bn_finalize");
    const long long total = n_rows * c;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(blocks), dim3(256), 0, s, x, mean_rstd, gamma, beta, (long long)n_rows, c,
                       relu, y);
    JOVA_CHECK_LAUNCH("bn_apply");
    return JOVA_OK;
}

extern "C" int jova_batchnorm_bwd(const float* x, const float* dy, const float* y_or_null, const float* mean_rstd,
                                  const float* gamma, int64_t n_rows, int c, int relu_input_mask, float* dx,
                                  float* dgamma, float* dbeta, double* sums_ws, void* stream) {
    JOVA_CHECK_ARG(x && dy && mean_rstd && gamma && dx && dgamma && dbeta && sums_ws && n_rows >= 1 && c >= 1,
                   "batchnorm_bwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    JOVA_ZERO(sums_ws, sizeof(double) * 2 * c, s);
    dim3 grid;
    int rpb;
    colreduce_grid(n_rows, c, grid, rpb);
    hipLaunchKernelGGL(colreduce_kernel<1>, grid, dim3(256), 0, s, x, dy, y_or_null, mean_rstd, (long long)n_rows, c,
                       (long long)c, rpb, sums_ws, nullptr);
    JOVA_CHECK_LAUNCH("bn_bwd_reduce");
    const long long total = n_rows * c;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks), dim3(256), 0, s, x, dy, y_or_null, mean_rstd, gamma, sums_ws,
                       (long long)n_rows, c, relu_input_mask, dx, dgamma, dbeta);
    JOVA_CHECK_LAUNCH("bn_bwd_apply");
    return JOVA_OK;
}

extern "C" int jova_layernorm_fwd(const float* x, const float* residual, const float* gamma, const float* beta,
                                  int64_t n_rows, int d, float eps, float* y, float* mean_rstd, void* stream) {
    JOVA_CHECK_ARG(x && gamma && beta && y && mean_rstd && n_rows >= 0, "layernorm_fwd: bad argument");
    JOVA_CHECK_ARG(d >= 1 && d <= 64 * LN_MAXV, "layernorm_fwd: d %d not in [1,%d]", d, 64 * LN_MAXV);
    if (n_rows == 0) return JOVA_OK;
    const bool v4 = (d & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)residual) | ((uintptr_t)gamma) | ((uintptr_t)beta) |
                                      ((uintptr_t)y)) & 15) == 0;
#define LN_FWD(V) do { if (v4) hipLaunchKernelGGL((layernorm_fwd_kernel<V, true>), dim3(cdiv(n_rows, 4)), dim3(256), 0,  \
                                    (hipStream_t)stream, x, residual, gamma, beta, (long long)n_rows, d, eps, y, mean_rstd); \
                       else hipLaunchKernelGGL((layernorm_fwd_kernel<V, false>), dim3(cdiv(n_rows, 4)), dim3(256), 0,     \
                                    (hipStream_t)stream, x, residual, gamma, beta, (long long)n_rows, d, eps, y, mean_rstd); } while (0)
    if (d <= 256) LN_FWD(4); else if (d <= 512) LN_FWD(8); else if (d <= 1024) LN_FWD(16); else LN_FWD(32);
    JOVA_CHECK_LAUNCH("layernorm_fwd");
    return JOVA_OK;
}

extern "C" int jova_layernorm_bwd(const float* x, const float* residual, const float* dy, const float* gamma,
                                  const float* mean_rstd, int64_t n_rows, int d, float* dz, float* dgamma,
                                  float* dbeta, void* stream) {
    JOVA_CHECK_ARG(x && dy && gamma && mean_rstd && dz && dgamma && dbeta && n_rows >= 0, "layernorm_bwd: bad argument");
    JOVA_CHECK_ARG(d >= 1 && d <= 64 * LN_MAXV, "layernorm_bwd: d %d not in [1,%d]", d, 64 * LN_MAXV);
    if (n_rows == 0) return JOVA_OK;
    int rpb = (int)((n_rows + 1023) / 1024);
    if (rpb < 4) rpb = 4;
    const size_t lds = sizeof(float) * 8 * d;
    const bool v4 = (d & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)residual) | ((uintptr_t)gamma) | ((uintptr_t)dy) |
                                      ((uintptr_t)dz)) & 15) == 0;
#define LN_BWD(V) do { if (v4) hipLaunchKernelGGL((layernorm_bwd_kernel<V, true>), dim3(cdiv(n_rows, rpb)), dim3(256), lds, \
                          (hipStream_t)stream, x, residual, dy, gamma, mean_rstd, (long long)n_rows, d, rpb, dz, dgamma, dbeta); \
                       else hipLaunchKernelGGL((layernorm_bwd_kernel<V, false>), dim3(cdiv(n_rows, rpb)), dim3(256), lds,   \
                          (hipStream_t)stream, x, residual, dy, gamma, mean_rstd, (long long)n_rows, d, rpb, dz, dgamma, dbeta); } while (0)
    if (d <= 256) LN_BWD(4); else if (d <= 512) LN_BWD(8); else if (d <= 1024) LN_BWD(16); else LN_BWD(32);
    JOVA_CHECK_LAUNCH("layernorm_bwd");
    return JOVA_OK;
}

extern "C" int jova_softmax_fwd(const float* x, int64_t n_rows, int n, int64_t ld, float* y, void* stream) {
    JOVA_CHECK_ARG(x && y && n_rows >= 0 && ld >= n, "softmax_fwd: bad argument");
    JOVA_CHECK_ARG(n >= 1 && n <= 64 * SM_MAXV, "softmax_fwd: row length %d not in [1,%d]", n, 64 * SM_MAXV);
    if (n_rows == 0) return JOVA_OK;
    if (n <= 512)
        hipLaunchKernelGGL(softmax_fwd_kernel<8>, dim3(cdiv(n_rows, 4)), dim3(256), 0, (hipStream_t)stream, x,
                           (long long)n_rows, n, (long long)ld, y);
    else if (n <= 1024)
        hipLaunchKernelGGL(softmax_fwd_kernel<16>, dim3(cdiv(n_rows, 4)), dim3(256), 0, (hipStream_t)stream, x,
                           (long long)n_rows, n, (long long)ld, y);
    else
        hipLaunchKernelGGL(softmax_fwd_kernel<32>, dim3(cdiv(n_rows, 4)), dim3(256), 0, (hipStream_t)stream, x,
                           (long long)n_rows, n, (long long)ld, y);
    JOVA_CHECK_LAUNCH("softmax_fwd");
    return JOVA_OK;
}

extern "C" int jova_softmax_bwd(const float* p, const float* dp, const float* p_dropped_or_null, float keep_scale,
                                int64_t n_rows, int n, int64_t ld, float* ds, void* stream) {
    JOVA_CHECK_ARG(p && dp && ds && n_rows >= 0 && ld >= n, "softmax_bwd: bad argument");
    JOVA_CHECK_ARG(n >= 1 && n <= 64 * SM_MAXV, "softmax_bwd: row length %d not in [1,%d]", n, 64 * SM_MAXV);
    if (n_rows == 0) return JOVA_OK;
    if (n <= 512)
        hipLaunchKernelGGL(softmax_bwd_kernel<8>, dim3(cdiv(n_rows, 4)), dim3(256), 0, (hipStream_t)stream, p, dp,
                           p_dropped_or_null, keep_scale, (long long)n_rows, n, (long long)ld, ds);
    else if (n <= 1024)
        hipLaunchKernelGGL(softmax_bwd_kernel<16>, dim3(cdiv(n_rows, 4)), dim3(256), 0, (hipStream_t)stream, p, dp,
                           p_dropped_or_null, keep_scale, (long long)n_rows, n, (long long)ld, ds);
    else
        hipLaunchKernelGGL(softmax_bwd_kernel<32>, dim3(cdiv(n_rows, 4)), dim3(256), 0, (hipStream_t)stream, p, dp,
                           p_dropped_or_null, keep_scale, (long long)n_rows, n, (long long)ld, ds);
    JOVA_CHECK_LAUNCH("softmax_bwd");
    return JOVA_OK;
}

extern "C" int jova_dropout(const float* x, int64_t n_rows, int n, int64_t ld, float p, uint64_t seed,
                            uint64_t offset, float* y, void* stream) {
    JOVA_CHECK_ARG(x && y && n_rows >= 0 && n >= 1 && ld >= n, "dropout: bad argument");
    JOVA_CHECK_ARG(p >= 0.f && p < 1.f, "dropout: p %f not in [0,1)", (double)p);
    if (n_rows == 0) return JOVA_OK;
    if ((ld & 3) == 0 && ld - n < 4 && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0) {
        const long long q4 = n_rows * ld / 4;
        const int b4 = (int)((q4 + 255) / 256 < 16384 ? (q4 + 255) / 256 : 16384);
        hipLaunchKernelGGL(dropout_v4_kernel, dim3(b4), dim3(256), 0, (hipStream_t)stream, x, q4, p,
                           (unsigned long long)seed, (unsigned long long)offset, y);
        JOVA_CHECK_LAUNCH("dropout_v4");
        return JOVA_OK;
    }
    const long long quads = (n_rows * n + 3) / 4;
    const int blocks = (int)((quads + 255) / 256 < 8192 ? (quads + 255) / 256 : 8192);
    hipLaunchKernelGGL(dropout_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (long long)n_rows, n,
                       (long long)ld, p, (unsigned long long)seed, (unsigned long long)offset, y);
    JOVA_CHECK_LAUNCH("dropout");
    return JOVA_OK;
}
