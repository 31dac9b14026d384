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

// Multi-task form (output_dim = len(tasks) > 1, proj/dti/train_joint_attn.py:1027): y / p / w are [n, n_tasks] row-major,
// blockIdx.z (sums: blockIdx.y) is the task: sample i of task t sits at [i * stride + t] and the task owns ws + 16 t.
__global__ void __launch_bounds__(MT)
metrics_sums_kernel(const float* __restrict__ y, const float* __restrict__ p, const float* __restrict__ w, int n,
                    int stride, double scale, double shift, double* __restrict__ ws) {
    __shared__ double red[MT / 64];
    const int task = blockIdx.y;
    y += task; p += task; if (w) w += task;
    ws += 16 * task;
    double m = 0.0, sy = 0.0, sp = 0.0, se = 0.0;
    for (int i = blockIdx.x * MT + threadIdx.x; i < n; i += gridDim.x * MT) {
        if (w && w[(size_t)i * stride] == 0.f) continue;
        const double a = untransform(y[(size_t)i * stride], scale, shift), b = untransform(p[(size_t)i * stride], scale, shift);
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
                     int stride, double scale, double shift, double* __restrict__ ws) {
    const int bi = blockIdx.x, bj = blockIdx.y, task = blockIdx.z;
    if (bj < bi) return;
    y += task; p += task; if (w) w += task;
    ws += 16 * task;
    __shared__ double ty[MT], tp[MT];
    __shared__ int tv[MT];
    __shared__ double red[MT / 64];
    const int t = threadIdx.x;
    const int j0 = bj * MT + t;
    const bool vj = j0 < n && !(w && w[(size_t)j0 * stride] == 0.f);
    ty[t] = vj ? untransform(y[(size_t)j0 * stride], scale, shift) : 0.0;
    tp[t] = vj ? untransform(p[(size_t)j0 * stride], scale, shift) : 0.0;
    tv[t] = vj;
    const int i = bi * MT + t;
    const bool vi = i < n && !(w && w[(size_t)i * stride] == 0.f);
    const double yi = vi ? untransform(y[(size_t)i * stride], scale, shift) : 0.0;
    const double pi = vi ? untransform(p[(size_t)i * stride], scale, shift) : 0.0;
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

struct TaskMetrics { double rms, ci, r2, m, sw, pairs; };

__device__ TaskMetrics task_metrics(const double* ws) {
    const unsigned long long* wc = reinterpret_cast<const unsigned long long*>(ws);
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    TaskMetrics t;
    t.m = ws[0];
    t.sw = 0.5 * (double)wc[8];
    t.pairs = (double)wc[9];
    t.rms = t.m > 0.0 ? sqrt(ws[3] / t.m) : nan;
    t.ci = t.pairs > 0.0 ? 1.0 - t.sw / t.pairs : nan;
    const double den = sqrt(ws[4] * ws[5]);
    t.r2 = nan;
    if (t.m > 1.0 && den > 0.0) { const double r = fmin(1.0, fmax(-1.0, ws[6] / den)); t.r2 = r * r; }
    return t;
}

// np.nanmean([ci, r2]) - rms   (Jova.evaluate, proj/dti/train_joint_attn.py:335-343)
__device__ double dti_score(double rms, double ci, double r2) {
    double mean = __longlong_as_double(0x7ff8000000000000LL);
    if (ci == ci && r2 == r2) mean = 0.5 * (ci + r2);
    else if (ci == ci) mean = ci;
    else if (r2 == r2) mean = r2;
    return mean - rms;
}

__global__ void metrics_final_kernel(const double* __restrict__ ws, double* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const TaskMetrics t = task_metrics(ws);
    out[0] = t.rms; out[1] = t.ci; out[2] = t.r2; out[3] = dti_score(t.rms, t.ci, t.r2); out[4] = t.sw; out[5] = t.pairs;
}

// Metric.compute_metric with the constructor default arithmetic_mean = True and task_averager = np.nanmean
// (jova/metrics/all_metrics.py:147, :517-600; the DTI driver builds Metric(fn, np.nanmean), train_joint_attn.py:299-301):
// task t's value is weighted by  included_tasks * n_t / N  and the weighted values are nan-averaged, which comes to
//     sum over tasks with a finite value of  n_t * value_t / N,
// n_t = samples of task t with w != 0, N = the sum of n_t over all tasks EXCEPT those the metric excludes up front: for
// the concordance index a task whose kept labels are all equal (:273-279; no pair).  A task without samples is NaN and
// weighs nothing; a task whose r2 is undefined (constant input) is NaN but its samples stay in N, as in the reference.
__global__ void metrics_tasks_final_kernel(const double* __restrict__ ws, int n_tasks, double* __restrict__ per_task,
                                           double* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    double n_all = 0.0, n_ci = 0.0, s_rms = 0.0, s_ci = 0.0, s_r2 = 0.0, sw = 0.0, pairs = 0.0;
    bool any_rms = false, any_ci = false, any_r2 = false;
    for (int t = 0; t < n_tasks; ++t) {
        const TaskMetrics k = task_metrics(ws + 16 * t);
        if (per_task) {
            double* o = per_task + 6 * t;
            o[0] = k.rms; o[1] = k.ci; o[2] = k.r2; o[3] = k.m; o[4] = k.sw; o[5] = k.pairs;
        }
        n_all += k.m;
        if (k.pairs > 0.0) n_ci += k.m;
        if (k.rms == k.rms) { s_rms += k.m * k.rms; any_rms = true; }
        if (k.ci == k.ci) { s_ci += k.m * k.ci; any_ci = true; }
        if (k.r2 == k.r2) { s_r2 += k.m * k.r2; any_r2 = true; }
        sw += k.sw;
        pairs += k.pairs;
    }
    const double rms = any_rms ? s_rms / n_all : nan, ci = any_ci ? s_ci / n_ci : nan, r2 = any_r2 ? s_r2 / n_all : nan;
    out[0] = rms; out[1] = ci; out[2] = r2; out[3] = dti_score(rms, ci, r2); out[4] = sw; out[5] = pairs;
}

}  // namespace

// the pair count is brute force (n^2 / 2 comparisons per task): 2^18 samples = 3.4e10, a second of GPU time; the
// reference's own use is a batch (<= 256) or a test fold (<= 24 k)
#define METRICS_MAX_N (1 << 18)

static int metrics_launch(const float* y_true, const float* y_pred, const float* w_or_null, int n, int n_tasks,
                          double y_std, double y_mean, double* ws, hipStream_t s) {
    JOVA_ZERO(ws, (size_t)n_tasks * 16 * sizeof(double), s);
    if (n > 0) {
        const int nt = (n + MT - 1) / MT;
        hipLaunchKernelGGL(metrics_sums_kernel, dim3(nt < 1024 ? nt : 1024, n_tasks), dim3(MT), 0, s, y_true, y_pred,
                           w_or_null, n, n_tasks, y_std, y_mean, ws);
        JOVA_CHECK_LAUNCH("dti_metrics(sums)");
        hipLaunchKernelGGL(metrics_pairs_kernel, dim3(nt, nt, n_tasks), dim3(MT), 0, s, y_true, y_pred, w_or_null, n,
                           n_tasks, y_std, y_mean, ws);
        JOVA_CHECK_LAUNCH("dti_metrics(pairs)");
    }
    return JOVA_OK;
}

extern "C" int jova_dti_metrics(const float* y_true, const float* y_pred, const float* w_or_null, int n, double y_std,
                                double y_mean, double* ws16, double* out6, void* stream) {
    JOVA_CHECK_ARG(ws16 && out6 && n >= 0 && (n == 0 || (y_true && y_pred)), "dti_metrics: bad argument");
    JOVA_CHECK_ARG(n <= METRICS_MAX_N, "dti_metrics: more than 2^18 samples");
    hipStream_t s = (hipStream_t)stream;
    const int rc = metrics_launch(y_true, y_pred, w_or_null, n, 1, y_std, y_mean, ws16, s);
    if (rc != JOVA_OK) return rc;
    hipLaunchKernelGGL(metrics_final_kernel, dim3(1), dim3(64), 0, s, ws16, out6);
    JOVA_CHECK_LAUNCH("dti_metrics(final)");
    return JOVA_OK;
}

extern "C" int jova_dti_metrics_tasks(const float* y_true, const float* y_pred, const float* w_or_null, int n, int n_tasks,
                                      double y_std, double y_mean, double* ws, double* per_task_or_null, double* out6,
                                      void* stream) {
    JOVA_CHECK_ARG(ws && out6 && n >= 0 && n_tasks >= 1 && n_tasks <= 65535 && (n == 0 || (y_true && y_pred)),
                   "dti_metrics_tasks: bad argument");
    JOVA_CHECK_ARG(n <= METRICS_MAX_N, "dti_metrics_tasks: more than 2^18 samples");
    hipStream_t s = (hipStream_t)stream;
    const int rc = metrics_launch(y_true, y_pred, w_or_null, n, n_tasks, y_std, y_mean, ws, s);
    if (rc != JOVA_OK) return rc;
    hipLaunchKernelGGL(metrics_tasks_final_kernel, dim3(1), dim3(64), 0, s, ws, n_tasks, per_task_or_null, out6);
    JOVA_CHECK_LAUNCH("dti_metrics_tasks(final)");
    return JOVA_OK;
}
