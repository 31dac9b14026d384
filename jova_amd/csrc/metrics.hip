// Per-batch regression metrics of the evaluation phase, computed where the predictions already are (HBM), so the
// evaluation loop needs no device-to-host copy per batch: RMSE, squared Pearson correlation, concordance index and the
// score Jova.evaluate forms from them.  Reference: jova/metrics/all_metrics.py:80-104, jova/metrics/cindex_measure.py:15-38
// with count_swapped -> ext_src/c_swapped.c:324-365, proj/dti/train_joint_attn.py:335-343.
//
// The concordance index is a count over label/prediction pairs.  The reference counts them with a red-black tree in
// O(n log n) on one core; an evaluation batch has 128-256 pairs and a whole Davis/KIBA test fold 6k-24k, i.e. at most
// 3e8 comparisons, which the brute-force tiled count below finishes in well under a millisecond while staying exact
// (integer counters; half-weights of tied predictions are carried as 2x integers).
#include "common.h"
#include "jova_hip.h"

namespace {

constexpr int MT = 256;     // samples per tile == threads per block

// ws layout (16 x 8 bytes): [0] m, [1] sum y, [2] sum p, [3] sum (y-p)^2, [4] sum ym^2, [5] sum pm^2, [6] sum ym*pm,
//                           [8] (u64) 2*swapped, [9] (u64) pairs with different labels
__device__ __forceinline__ double untransform(float v, double scale, double shift) {
    return __dadd_rn(__dmul_rn((double)v, scale), shift);       // two roundings, as numpy's z * y_std + y_mean
}

__device__ __forceinline__ double block_sum(double v, double* red) {
    for (int o = 32; o; o >>= 1) v += __shfl_down(v, o, 64);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < MT / 64; ++i) t += red[i];
    return t;                                                     // valid on thread 0
}

__global__ void __launch_bounds__(MT)
metrics_sums_kernel(const float* __restrict__ y, const float* __restrict__ p, const float* __restrict__ w, int n,
                    double scale, double shift, double* __restrict__ ws) {
    __shared__ double red[MT / 64];
    double m = 0.0, sy = 0.0, sp = 0.0, se = 0.0;
    for (int i = blockIdx.x * MT + threadIdx.x; i < n; i += gridDim.x * MT) {
        if (w && w[i] == 0.f) continue;
        const double a = untransform(y[i], scale, shift), b = untransform(p[i], scale, shift);
        m += 1.0; sy += a; sp += b; se += (a - b) * (a - b);
    }
    m = block_sum(m, red); sy = block_sum(sy, red); sp = block_sum(sp, red); se = block_sum(se, red);
    if (threadIdx.x == 0) {
        atomicAdd(&ws[0], m); atomicAdd(&ws[1], sy); atomicAdd(&ws[2], sp); atomicAdd(&ws[3], se);
    }
}

// Block (bi, bj), bj >= bi: thread t owns sample i = bi*MT + t and meets the samples of tile bj (through LDS) with
// index j > i.  Diagonal blocks also add the centred second moments of their own samples.
__global__ void __launch_bounds__(MT)
metrics_pairs_kernel(const float* __restrict__ y, const float* __restrict__ p, const float* __restrict__ w, int n,
                     double scale, double shift, double* __restrict__ ws) {
    const int bi = blockIdx.x, bj = blockIdx.y;
    if (bj < bi) return;
    __shared__ double ty[MT], tp[MT];
    __shared__ int tv[MT];
    __shared__ double red[MT / 64];
    const int t = threadIdx.x;
    const int j0 = bj * MT + t;
    const bool vj = j0 < n && !(w && w[j0] == 0.f);
    ty[t] = vj ? untransform(y[j0], scale, shift) : 0.0;
    tp[t] = vj ? untransform(p[j0], scale, shift) : 0.0;
    tv[t] = vj;
    const int i = bi * MT + t;
    const bool vi = i < n && !(w && w[i] == 0.f);
    const double yi = vi ? untransform(y[i], scale, shift) : 0.0, pi = vi ? untransform(p[i], scale, shift) : 0.0;
    __syncthreads();
    unsigned long long sw2 = 0, pairs = 0;
    if (vi) {
        const int first = bi == bj ? t + 1 : 0;
        for (int k = first; k < MT; ++k) {
            if (!tv[k]) continue;
            const double yj = ty[k], pj = tp[k];
            if (yi == yj) continue;                               // same label: neither a pair nor a swap
            pairs += 1;
            if (pi == pj) sw2 += 1;                               // tied predictions, different labels: half a swap
            else if ((pi < pj) == (yi > yj)) sw2 += 2;            // ordered against the labels
        }
    }
    double ey = 0.0, ep = 0.0, eyp = 0.0;
    if (bi == bj && vi) {
        const double m = ws[0];
        const double ym = yi - ws[1] / m, pm = pi - ws[2] / m;
        ey = ym * ym; ep = pm * pm; eyp = ym * pm;
    }
    const double s2 = block_sum((double)sw2, red), pr = block_sum((double)pairs, red);   // < 2^53: exact in double
    if (threadIdx.x == 0) {
        unsigned long long* wc = reinterpret_cast<unsigned long long*>(ws);
        if (s2 != 0.0) atomicAdd(&wc[8], (unsigned long long)s2);
        if (pr != 0.0) atomicAdd(&wc[9], (unsigned long long)pr);
    }
    if (bi == bj) {
        ey = block_sum(ey, red); ep = block_sum(ep, red); eyp = block_sum(eyp, red);
        if (threadIdx.x == 0) { atomicAdd(&ws[4], ey); atomicAdd(&ws[5], ep); atomicAdd(&ws[6], eyp); }
    }
}

__global__ void metrics_final_kernel(const double* __restrict__ ws, double* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const unsigned long long* wc = reinterpret_cast<const unsigned long long*>(ws);
    const double m = ws[0], nan = __longlong_as_double(0x7ff8000000000000LL);
    const double sw = 0.5 * (double)wc[8], pairs = (double)wc[9];
    const double rms = m > 0.0 ? sqrt(ws[3] / m) : nan;
    const double ci = pairs > 0.0 ? 1.0 - sw / pairs : nan;
    const double den = sqrt(ws[4] * ws[5]);
    double r2 = nan;
    if (m > 1.0 && den > 0.0) { const double r = fmin(1.0, fmax(-1.0, ws[6] / den)); r2 = r * r; }
    // np.nanmean([ci, r2]) - rms
    double mean = nan;
    if (ci == ci && r2 == r2) mean = 0.5 * (ci + r2);
    else if (ci == ci) mean = ci;
    else if (r2 == r2) mean = r2;
    out[0] = rms; out[1] = ci; out[2] = r2; out[3] = mean - rms; out[4] = sw; out[5] = pairs;
}

}  // namespace

extern "C" int jova_dti_metrics(const float* y_true, const float* y_pred, const float* w_or_null, int n, double y_std,
                                double y_mean, double* ws16, double* out6, void* stream) {
    JOVA_CHECK_ARG(ws16 && out6 && n >= 0 && (n == 0 || (y_true && y_pred)), "dti_metrics: bad argument");
    JOVA_CHECK_ARG(n <= (1 << 20), "dti_metrics: more than 2^20 samples");
    hipStream_t s = (hipStream_t)stream;
    JOVA_ZERO(ws16, 16 * sizeof(double), s);
    if (n > 0) {
        const int nt = (n + MT - 1) / MT;
        hipLaunchKernelGGL(metrics_sums_kernel, dim3(nt < 1024 ? nt : 1024), dim3(MT), 0, s, y_true, y_pred, w_or_null, n,
                           y_std, y_mean, ws16);
        JOVA_CHECK_LAUNCH("dti_metrics(sums)");
        hipLaunchKernelGGL(metrics_pairs_kernel, dim3(nt, nt), dim3(MT), 0, s, y_true, y_pred, w_or_null, n, y_std, y_mean,
                           ws16);
        JOVA_CHECK_LAUNCH("dti_metrics(pairs)");
    }
    hipLaunchKernelGGL(metrics_final_kernel, dim3(1), dim3(64), 0, s, ws16, out6);
    JOVA_CHECK_LAUNCH("dti_metrics(final)");
    return JOVA_OK;
}
