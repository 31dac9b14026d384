// Protein embedding (Prot2Vec), view pooling, loss, optimiser and small element-wise kernels, gfx950.
#include "common.h"
#include "jova_hip.h"

// ------------------------------------------------------------------------------------------------ Prot2Vec
// The padded index cube [B, P_max, w] of the reference (jova/nn/models.py:233-242) is never materialised: all
// protein profiles stay resident in HBM as one int32 table [rows, w]; a batch is (row offset, segment count) per
// pair and the pad index (= vocab, the zero row of the embedding) is implied for p >= count.
//   out[p][b][j*e + c] = relu(E[idx(b,p,j)][c])
__global__ void prot2vec_fwd_kernel(const int* __restrict__ table, const int* __restrict__ prot_off,
                                    const int* __restrict__ prot_len, const float* __restrict__ emb, int bsz,
                                    int p_max, int w, int e, int vocab, int relu, float* __restrict__ out,
                                    long long* __restrict__ idx_out) {
    const long long total = (long long)p_max * bsz * w * e;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(t % e);
        long long q = t / e;
        const int j = (int)(q % w);
        q /= w;
        const int b = (int)(q % bsz), p = (int)(q / bsz);
        int idx = vocab;
        if (p < prot_len[b]) idx = table[((long long)prot_off[b] + p) * w + j];
        idx = min(max(idx, 0), vocab);
        float v = emb[(long long)idx * e + c];
        if (relu) v = fmaxf(v, 0.f);
        out[t] = v;
        if (idx_out && c == 0) idx_out[((long long)b * p_max + p) * w + j] = idx;
    }
}

// dE[idx][c] += dout * (out > 0); the pad row receives nothing (nn.Embedding padding_idx)
__global__ void prot2vec_bwd_kernel(const int* __restrict__ table, const int* __restrict__ prot_off,
                                    const int* __restrict__ prot_len, const float* __restrict__ dout,
                                    const float* __restrict__ out, int bsz, int p_max, int w, int e, int vocab,
                                    int relu, float* __restrict__ demb) {
    const long long total = (long long)p_max * bsz * w * e;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(t % e);
        long long q = t / e;
        const int j = (int)(q % w);
        q /= w;
        const int b = (int)(q % bsz), p = (int)(q / bsz);
        if (p >= prot_len[b]) continue;
        int idx = table[((long long)prot_off[b] + p) * w + j];
        if (idx < 0 || idx >= vocab) continue;
        if (relu && !(out[t] > 0.f)) continue;
        atomicAdd(&demb[(long long)idx * e + c], dout[t]);
    }
}

// ------------------------------------------------------------------------------------------------ view pooling
// x [S, B, L] with views occupying consecutive segment ranges seg_off[v] .. seg_off[v+1]:
//   out[b][v*L + c] = sum_{s in view v} x[s][b][c]          (JointAttention: split + sum(0) + cat, models.py:966-971)
__global__ void viewpool_fwd_kernel(const float* __restrict__ x, const int* __restrict__ seg_off, int n_views, int bsz,
                                    int l, float* __restrict__ out) {
    const long long total = (long long)bsz * n_views * l;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int c = (int)(t % l);
    const int v = (int)((t / l) % n_views);
    const int b = (int)(t / ((long long)l * n_views));
    double s = 0.0;      // fp64: the BatchNorm that follows subtracts a batch mean much larger than the spread
    for (int sgm = seg_off[v]; sgm < seg_off[v + 1]; ++sgm) s += (double)x[((long long)sgm * bsz + b) * l + c];
    out[t] = (float)s;
}

// float4 form (l % 4 == 0): a workgroup owns (b, v) and 128 columns; its 8 thread groups take every 8th segment of the
// view (fp64 partial sums), the partials are added in group order through LDS.
__global__ void __launch_bounds__(256)
viewpool_fwd_v4_kernel(const float* __restrict__ x, const int* __restrict__ seg_off, int n_views, int bsz, int l,
                       float* __restrict__ out) {
    __shared__ double part[8][128];
    const int cblocks = (l + 127) / 128;
    const int cbk = blockIdx.x % cblocks;
    const int bv = blockIdx.x / cblocks;
    const int v = bv % n_views, b = bv / n_views;
    const int q = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int c = cbk * 128 + 4 * q;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (c < l) {
        const int s_end = seg_off[v + 1];
        const float* xp = x + (long long)b * l + c;
        const long long sstride = (long long)bsz * l;
        int sgm = seg_off[v] + grp;
        for (; sgm + 24 < s_end; sgm += 32) {          // four segments of this group in flight
            const float4 a0 = *reinterpret_cast<const float4*>(xp + sgm * sstride);
            const float4 a1 = *reinterpret_cast<const float4*>(xp + (sgm + 8) * sstride);
            const float4 a2 = *reinterpret_cast<const float4*>(xp + (sgm + 16) * sstride);
            const float4 a3 = *reinterpret_cast<const float4*>(xp + (sgm + 24) * sstride);
            s0 += (double)a0.x; s1 += (double)a0.y; s2 += (double)a0.z; s3 += (double)a0.w;
            s0 += (double)a1.x; s1 += (double)a1.y; s2 += (double)a1.z; s3 += (double)a1.w;
            s0 += (double)a2.x; s1 += (double)a2.y; s2 += (double)a2.z; s3 += (double)a2.w;
            s0 += (double)a3.x; s1 += (double)a3.y; s2 += (double)a3.z; s3 += (double)a3.w;
        }
        for (; sgm < s_end; sgm += 8) {
            const float4 a0 = *reinterpret_cast<const float4*>(xp + sgm * sstride);
            s0 += (double)a0.x; s1 += (double)a0.y; s2 += (double)a0.z; s3 += (double)a0.w;
        }
    }
    part[grp][4 * q + 0] = s0; part[grp][4 * q + 1] = s1; part[grp][4 * q + 2] = s2; part[grp][4 * q + 3] = s3;
    __syncthreads();
    if (threadIdx.x < 128) {
        const int cc = cbk * 128 + threadIdx.x;
        if (cc < l) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) t += part[k][threadIdx.x];
            out[((long long)b * n_views + v) * l + cc] = (float)t;
        }
    }
}

__global__ void __launch_bounds__(256)
viewpool_bwd_v4_kernel(const float* __restrict__ dout, const int* __restrict__ seg_off, int n_views, int bsz, int l4,
                       int s_total, float* __restrict__ dx) {
    const long long total = (long long)s_total * bsz * l4;        // float4s
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const unsigned sb = (unsigned)(t / l4);
        const int c = (int)(t - (long long)sb * l4);
        const int sgm = (int)(sb / (unsigned)bsz), b = (int)(sb - (unsigned)sgm * (unsigned)bsz);
        int v = 0;
        for (int k = 1; k < n_views; ++k) v += (sgm >= seg_off[k]);
        reinterpret_cast<float4*>(dx)[t] = reinterpret_cast<const float4*>(dout)[((long long)b * n_views + v) * l4 + c];
    }
}

__global__ void viewpool_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ seg_off, int n_views,
                                    int bsz, int l, int s_total, float* __restrict__ dx) {
    const long long total = (long long)s_total * bsz * l;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(t % l);
        const int b = (int)((t / l) % bsz);
        const int sgm = (int)(t / ((long long)l * bsz));
        int v = 0;
        for (int k = 1; k < n_views; ++k) v += (sgm >= seg_off[k]);
        dx[t] = dout[((long long)b * n_views + v) * l + c];
    }
}

// ------------------------------------------------------------------------------------------------ loss
// loss = mean((out * w - y)^2) over n elements (nn.MSELoss on outputs*weights, train_joint_attn.py:464-465);
// dout = 2 * w * (out * w - y) / n.  Single workgroup: n is the batch size.
__global__ void __launch_bounds__(256)
mse_kernel(const float* __restrict__ o, const float* __restrict__ w, const float* __restrict__ y, int n,
           float* __restrict__ loss, float* __restrict__ dout) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float d = o[i] * w[i] - y[i];
        acc += (double)d * d;
        if (dout) dout[i] = 2.f * w[i] * d / (float)n;
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *loss = (float)(((red[0] + red[1]) + (red[2] + red[3])) / (double)n);
}

// ------------------------------------------------------------------------------------------------ optimiser
// torch.optim.SGD(lr, momentum, weight_decay, nesterov=False): g += wd * p; buf = mu * buf + g; p -= lr * buf
// Every optimizer kernel takes two extras.  `skip`: device flag of jova_nonfinite_flag; when it is set the parameters
// and the optimizer state are left as they are (the reference leaves its loop at a NaN loss BEFORE backward / step,
// proj/dti/train_joint_attn.py:468-470, so its weights stay at the last finite values; here the decision stays on the
// device).  `zero_g`: the gradient is cleared once it has been consumed, which spares the step a separate fill of the
// flat gradient buffer (optimizer.zero_grad(), :428).
__global__ void sgd_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ mom, long long n,
                           float lr, float wd, float mu, float gscale, int first, const int* __restrict__ skip,
                           int zero_g) {
    const bool upd = !(skip && *skip);
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n;
         e += (long long)gridDim.x * blockDim.x) {
        if (upd) {
            float d = g[e] * gscale + wd * p[e];
            if (mom) {
                d = first ? d : mu * mom[e] + d;
                mom[e] = d;
            }
            p[e] -= lr * d;
        }
        if (zero_g) g[e] = 0.f;
    }
}

// torch.optim.Adamax(lr, betas, eps, weight_decay): the reference's default optimiser when no tuned CSV is found
__global__ void adamax_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                              float* __restrict__ u, long long n, float lr, float b1, float b2, float eps, float wd,
                              float gscale, float bias_corr1, const int* __restrict__ skip, int zero_g) {
    const bool upd = !(skip && *skip);
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n;
         e += (long long)gridDim.x * blockDim.x) {
        if (upd) {
            const float d = g[e] * gscale + wd * p[e];
            const float mm = b1 * m[e] + (1.f - b1) * d;
            const float uu = fmaxf(b2 * u[e], fabsf(d) + eps);
            m[e] = mm;
            u[e] = uu;
            p[e] -= (lr / bias_corr1) * mm / uu;
        }
        if (zero_g) g[e] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------------ element-wise
__global__ void relu_fwd_kernel(const float* __restrict__ x, long long n, float* __restrict__ y) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x)
        y[e] = fmaxf(x[e], 0.f);
}
__global__ void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, long long n,
                                float* __restrict__ dx) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x)
        dx[e] = y[e] > 0.f ? dy[e] : 0.f;
}
__global__ void axpy_kernel(const float* __restrict__ x, float a, long long n, float* __restrict__ y) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x)
        y[e] += a * x[e];
}

static inline int ew_blocks(long long n) { return (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192); }

// ================================================================================================ C ABI
extern "C" int jova_prot2vec_fwd(const int32_t* table, const int32_t* prot_off, const int32_t* prot_len,
                                 const float* emb, int batch, int p_max, int window, int e, int vocab, int relu,
                                 float* out, int64_t* idx_out_or_null, void* stream) {
    JOVA_CHECK_ARG(table && prot_off && prot_len && emb && out, "prot2vec_fwd: null pointer");
    JOVA_CHECK_ARG(batch >= 1 && p_max >= 0 && window >= 1 && e >= 1 && vocab >= 1, "prot2vec_fwd: bad shape");
    const long long total = (long long)p_max * batch * window * e;
    if (total == 0) return JOVA_OK;
    hipLaunchKernelGGL(prot2vec_fwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, table, prot_off,
                       prot_len, emb, batch, p_max, window, e, vocab, relu, out, (long long*)idx_out_or_null);
    JOVA_CHECK_LAUNCH("prot2vec_fwd");
    return JOVA_OK;
}

extern "C" int jova_prot2vec_bwd(const int32_t* table, const int32_t* prot_off, const int32_t* prot_len,
                                 const float* dout, const float* out, int batch, int p_max, int window, int e,
                                 int vocab, int relu, float* demb, void* stream) {
    JOVA_CHECK_ARG(table && prot_off && prot_len && dout && out && demb, "prot2vec_bwd: null pointer");
    JOVA_CHECK_ARG(batch >= 1 && p_max >= 0 && window >= 1 && e >= 1 && vocab >= 1, "prot2vec_bwd: bad shape");
    const long long total = (long long)p_max * batch * window * e;
    if (total == 0) return JOVA_OK;
    hipLaunchKernelGGL(prot2vec_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, table, prot_off,
                       prot_len, dout, out, batch, p_max, window, e, vocab, relu, demb);
    JOVA_CHECK_LAUNCH("prot2vec_bwd");
    return JOVA_OK;
}

extern "C" int jova_viewpool_fwd(const float* x, const int32_t* seg_off, int n_views, int batch, int l, float* out,
                                 void* stream) {
    JOVA_CHECK_ARG(x && seg_off && out && n_views >= 1 && batch >= 1 && l >= 1, "viewpool_fwd: bad argument");
    const long long total = (long long)batch * n_views * l;
    if ((l & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)out)) & 15) == 0)
        hipLaunchKernelGGL(viewpool_fwd_v4_kernel, dim3(batch * n_views * ((l + 127) / 128)), dim3(256), 0,
                           (hipStream_t)stream, x, seg_off, n_views, batch, l, out);
    else
        hipLaunchKernelGGL(viewpool_fwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, seg_off,
                           n_views, batch, l, out);
    JOVA_CHECK_LAUNCH("viewpool_fwd");
    return JOVA_OK;
}

extern "C" int jova_viewpool_bwd(const float* dout, const int32_t* seg_off, int n_views, int batch, int l,
                                 int s_total, float* dx, void* stream) {
    JOVA_CHECK_ARG(dout && seg_off && dx && n_views >= 1 && batch >= 1 && l >= 1 && s_total >= 0, "viewpool_bwd: bad argument");
    const long long total = (long long)s_total * batch * l;
    if (total == 0) return JOVA_OK;
    if ((l & 3) == 0 && ((((uintptr_t)dout) | ((uintptr_t)dx)) & 15) == 0 && (long long)s_total * batch < (1ll << 32))
        hipLaunchKernelGGL(viewpool_bwd_v4_kernel, dim3(ew_blocks(total / 4)), dim3(256), 0, (hipStream_t)stream, dout,
                           seg_off, n_views, batch, l / 4, s_total, dx);
    else
        hipLaunchKernelGGL(viewpool_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, dout, seg_off,
                           n_views, batch, l, s_total, dx);
    JOVA_CHECK_LAUNCH("viewpool_bwd");
    return JOVA_OK;
}

extern "C" int jova_mse_loss(const float* outputs, const float* weights, const float* target, int n, float* loss,
                             float* dout_or_null, void* stream) {
    JOVA_CHECK_ARG(outputs && weights && target && loss && n >= 1, "mse_loss: bad argument");
    hipLaunchKernelGGL(mse_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, outputs, weights, target, n, loss,
                       dout_or_null);
    JOVA_CHECK_LAUNCH("mse_loss");
    return JOVA_OK;
}

// adam / rmsprop / adagrad / adadelta on the flat buffers, torch.optim semantics (the reference builds its optimizer from
// torch.optim by name, proj/dti/train_joint_attn.py:276-296); weight decay is the coupled L2 form (d = g + wd p)
__global__ void optim_kernel(int kind, float* __restrict__ p, float* __restrict__ g, float* __restrict__ s1,
                             float* __restrict__ s2, long long n, float lr, float wd, float gscale, float h0, float h1,
                             float h2, float c0, float c1, const int* __restrict__ skip, int zero_g) {
    const bool upd = !(skip && *skip);
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n;
         e += (long long)gridDim.x * blockDim.x) {
        float d = g[e] * gscale + wd * p[e];
        if (zero_g) g[e] = 0.f;
        if (!upd) continue;
        if (kind == 2) {                         // adam: h0 = beta1, h1 = beta2, h2 = eps, c0 = 1 - b1^t, c1 = sqrt(1 - b2^t)
            const float m = h0 * s1[e] + (1.f - h0) * d;
            const float v = h1 * s2[e] + (1.f - h1) * d * d;
            s1[e] = m;
            s2[e] = v;
            p[e] -= (lr / c0) * m / (sqrtf(v) / c1 + h2);
        } else if (kind == 3) {                  // rmsprop: h0 = alpha, h1 = eps, h2 = momentum (s2 = momentum buffer)
            const float sq = h0 * s1[e] + (1.f - h0) * d * d;
            s1[e] = sq;
            const float avg = sqrtf(sq) + h1;
            if (h2 > 0.f) {
                const float b = h2 * s2[e] + d / avg;
                s2[e] = b;
                p[e] -= lr * b;
            } else {
                p[e] -= lr * d / avg;
            }
        } else if (kind == 4) {                  // adagrad: h1 = eps, c0 = lr / (1 + (t - 1) lr_decay)
            const float sum = s1[e] + d * d;
            s1[e] = sum;
            p[e] -= c0 * d / (sqrtf(sum) + h1);
        } else if (kind == 5) {                  // adadelta: h0 = rho, h1 = eps (s1 = square_avg, s2 = acc_delta)
            const float sq = h0 * s1[e] + (1.f - h0) * d * d;
            const float delta = sqrtf(s2[e] + h1) / sqrtf(sq + h1) * d;
            s1[e] = sq;
            s2[e] = h0 * s2[e] + (1.f - h0) * delta * delta;
            p[e] -= lr * delta;
        } else if (kind == 6) {                  // asgd: h0 = eta_t, h1 = lambd, h2 = mu_t (s1 = ax, the averaged iterate)
            float pn = p[e] * (1.f - h1 * h0);
            pn = pn - h0 * d;
            p[e] = pn;
            const float ax = s1[e];
            s1[e] = h2 != 1.f ? ax + (pn - ax) * h2 : pn;
        } else {                                 // rprop: h0 = eta-, h1 = eta+, c0 / c1 = step size bounds (s1 = previous
            const float prod = d * s1[e];        // gradient, s2 = per-element step size); no weight decay (wd = 0)
            const float fac = prod > 0.f ? h1 : (prod < 0.f ? h0 : 1.f);
            const float ss = fminf(fmaxf(s2[e] * fac, c0), c1);
            s2[e] = ss;
            if (prod < 0.f) d = 0.f;
            const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            p[e] -= sg * ss;
            s1[e] = d;
        }
    }
}

extern "C" int jova_optim_step(int kind, float* params, float* grads, float* state1, float* state2_or_null,
                               int64_t n, float lr, float weight_decay, float grad_scale, int step, float h0, float h1,
                               float h2, float h3, float h4, const int* skip_flag_or_null, int zero_grads, void* stream) {
    JOVA_CHECK_ARG(kind >= 2 && kind <= 7,
                   "optim_step: kind %d (2 adam, 3 rmsprop, 4 adagrad, 5 adadelta, 6 asgd, 7 rprop)", kind);
    JOVA_CHECK_ARG(n >= 0 && step >= 1 && (n == 0 || (params && grads && state1)), "optim_step: bad argument");
    if (n == 0) return JOVA_OK;
    JOVA_CHECK_ARG(state2_or_null || kind == 4 || kind == 6 || (kind == 3 && h2 <= 0.f),
                   "optim_step: this optimizer needs state2");
    float c0 = 1.f, c1 = 1.f;
    if (kind == 2) {
        c0 = (float)(1.0 - pow((double)h0, (double)step));
        c1 = (float)sqrt(1.0 - pow((double)h1, (double)step));
    } else if (kind == 4) {
        c0 = (float)((double)lr / (1.0 + (double)(step - 1) * (double)h0));
    } else if (kind == 6) {
        // torch.optim.ASGD: eta and mu are functions of the step count alone (its state tensors `eta`, `mu`):
        // eta_t = lr / (1 + lambd lr (t-1))^alpha, mu_t = 1 / max(1, (t-1) - t0); h0 lambd, h1 alpha, h2 t0
        const double lambd = h0, alpha = h1, t0 = h2, tm1 = (double)(step - 1);
        const double eta = (double)lr / pow(1.0 + lambd * (double)lr * tm1, alpha);
        const double mu = 1.0 / fmax(1.0, tm1 - t0);
        h0 = (float)eta;
        h1 = (float)lambd;
        h2 = (float)mu;
    } else if (kind == 7) {
        JOVA_CHECK_ARG(weight_decay == 0.f, "optim_step: rprop takes no weight decay");
        c0 = h3;                                 // step_sizes = (min, max)
        c1 = h4;
    }
    hipLaunchKernelGGL(optim_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, kind, params, grads, state1,
                       state2_or_null, (long long)n, lr, weight_decay, grad_scale, h0, h1, h2, c0, c1, skip_flag_or_null,
                       zero_grads);
    JOVA_CHECK_LAUNCH("optim_step");
    return JOVA_OK;
}

// flag |= 1 if any of x[0..n) is NaN or infinite (the reference's fail-fast reads loss.item() every batch,
// train_joint_attn.py:468-470; here the flag stays on the device and is read once per epoch)
__global__ void nonfinite_flag_kernel(const float* __restrict__ x, long long n, int* __restrict__ flag) {
    int bad = 0;                                 // bit 0: a NaN (what the reference tests for), bit 1: an infinity
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const float v = x[e];
        bad |= (v != v) ? 1 : (isinf(v) ? 2 : 0);
    }
    if (bad) atomicOr(flag, bad);
}
extern "C" int jova_nonfinite_flag(const float* x, int64_t n, int* flag, void* stream) {
    JOVA_CHECK_ARG(flag && n >= 0 && (n == 0 || x), "nonfinite_flag: bad argument");
    if (n == 0) return JOVA_OK;
    hipLaunchKernelGGL(nonfinite_flag_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, (long long)n, flag);
    JOVA_CHECK_LAUNCH("nonfinite_flag");
    return JOVA_OK;
}

extern "C" int jova_sgd_step(float* params, float* grads, float* momentum_buf_or_null, int64_t n, float lr,
                             float weight_decay, float momentum, float grad_scale, int first_step,
                             const int* skip_flag_or_null, int zero_grads, void* stream) {
    JOVA_CHECK_ARG(params && grads && n >= 0, "sgd_step: bad argument");
    if (n == 0) return JOVA_OK;
    hipLaunchKernelGGL(sgd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, params, grads,
                       momentum_buf_or_null, (long long)n, lr, weight_decay, momentum, grad_scale, first_step,
                       skip_flag_or_null, zero_grads);
    JOVA_CHECK_LAUNCH("sgd_step");
    return JOVA_OK;
}

extern "C" int jova_adamax_step(float* params, float* grads, float* exp_avg, float* exp_inf, int64_t n, float lr,
                                float beta1, float beta2, float eps, float weight_decay, float grad_scale, int step,
                                const int* skip_flag_or_null, int zero_grads, void* stream) {
    JOVA_CHECK_ARG(params && grads && exp_avg && exp_inf && n >= 0 && step >= 1, "adamax_step: bad argument");
    if (n == 0) return JOVA_OK;
    const float bc = 1.f - powf(beta1, (float)step);
    hipLaunchKernelGGL(adamax_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                       exp_inf, (long long)n, lr, beta1, beta2, eps, weight_decay, grad_scale, bc, skip_flag_or_null,
                       zero_grads);
    JOVA_CHECK_LAUNCH("adamax_step");
    return JOVA_OK;
}

extern "C" int jova_relu_fwd(const float* x, int64_t n, float* y, void* stream) {
    JOVA_CHECK_ARG(n >= 0 && (n == 0 || (x && y)), "relu_fwd: bad argument");
    if (n == 0) return JOVA_OK;
    hipLaunchKernelGGL(relu_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, (long long)n, y);
    JOVA_CHECK_LAUNCH("relu_fwd");
    return JOVA_OK;
}

extern "C" int jova_relu_bwd(const float* dy, const float* y, int64_t n, float* dx, void* stream) {
    JOVA_CHECK_ARG(n >= 0 && (n == 0 || (dy && y && dx)), "relu_bwd: bad argument");
    if (n == 0) return JOVA_OK;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, dy, y, (long long)n, dx);
    JOVA_CHECK_LAUNCH("relu_bwd");
    return JOVA_OK;
}

__global__ void scale_dev_kernel(const float* __restrict__ x, const float* __restrict__ a, long long n,
                                 float* __restrict__ y) {
    const float s = *a;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x)
        y[e] = x[e] * s;
}
extern "C" int jova_scale_dev(const float* x, const float* a_dev, int64_t n, float* y, void* stream) {
    JOVA_CHECK_ARG(n >= 0 && a_dev && (n == 0 || (x && y)), "scale_dev: bad argument");
    if (n == 0) return JOVA_OK;
    hipLaunchKernelGGL(scale_dev_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, a_dev, (long long)n, y);
    JOVA_CHECK_LAUNCH("scale_dev");
    return JOVA_OK;
}

extern "C" int jova_zero(float* p, int64_t n, void* stream) {
    JOVA_CHECK_ARG(n >= 0 && (n == 0 || p), "zero: bad argument");
    if (n == 0) return JOVA_OK;
    return jova_zero_fill(p, (size_t)n * sizeof(float), 1, (size_t)n * sizeof(float), (hipStream_t)stream);
}

extern "C" int jova_axpy(const float* x, float a, int64_t n, float* y, void* stream) {
    JOVA_CHECK_ARG(x && y && n >= 0, "axpy: bad argument");
    if (n == 0) return JOVA_OK;
    hipLaunchKernelGGL(axpy_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, a, (long long)n, y);
    JOVA_CHECK_LAUNCH("axpy");
    return JOVA_OK;
}

// ------------------------------------------------------------------------------------------------ 2-D copy with zero fill
// dst[r][c] (+)= (r < src_rows && c < src_cols) ? src[r][c] : 0 for r < dst_rows, c < dst_cols: pads an operand to the
// aligned GEMM's 128-multiples (ProteinCNN's 242-wide layers), crops a result back, adds a padded weight gradient's block
__global__ void copy2d_kernel(const float* __restrict__ src, long long src_rows, int src_cols, long long ld_src,
                              float* __restrict__ dst, long long dst_rows, int dst_cols, long long ld_dst, int accumulate) {
    const long long total = dst_rows * dst_cols;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long r = e / dst_cols;
        const int c = (int)(e - r * dst_cols);
        const float v = (r < src_rows && c < src_cols) ? src[r * ld_src + c] : 0.f;
        float* d = dst + r * ld_dst + c;
        *d = accumulate ? *d + v : v;
    }
}
extern "C" int jova_copy2d(const float* src, int64_t src_rows, int src_cols, int64_t ld_src, float* dst, int64_t dst_rows,
                           int dst_cols, int64_t ld_dst, int accumulate, void* stream) {
    JOVA_CHECK_ARG(src_rows >= 0 && dst_rows >= 0 && src_cols >= 0 && dst_cols >= 1 && ld_src >= src_cols && ld_dst >= dst_cols &&
                   (dst_rows == 0 || dst) && (src_rows == 0 || src_cols == 0 || src), "copy2d: bad argument");
    if (dst_rows == 0) return JOVA_OK;
    hipLaunchKernelGGL(copy2d_kernel, dim3(ew_blocks(dst_rows * dst_cols)), dim3(256), 0, (hipStream_t)stream, src,
                       (long long)src_rows, src_cols, (long long)ld_src, dst, (long long)dst_rows, dst_cols,
                       (long long)ld_dst, accumulate);
    JOVA_CHECK_LAUNCH("copy2d");
    return JOVA_OK;
}

// ------------------------------------------------------------------------------------------------ GraphNeuralNet pieces
// nn.Embedding forward / backward (jova/nn/models.py:362, :376: the fingerprint embedding of GraphNeuralNet) and the
// residual update x + (A hs) / (M - 1) of update_fingerprints (:364-372); 16 lanes per row, float4 when d % 4 == 0
__global__ void embedding_fwd_kernel(const float* __restrict__ w, const long long* __restrict__ idx, long long n, int d,
                                     long long vocab, float* __restrict__ out, int* __restrict__ bad) {
    const long long total = n * d;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long r = e / d;
        const int c = (int)(e - r * d);
        const long long k = idx[r];
        if (k < 0 || k >= vocab) {               // nn.Embedding raises: flag it (read by the caller off the hot loop), emit NaN
            if (c == 0) atomicOr(bad, 1);
            out[e] = __int_as_float(0x7fc00000);
        } else {
            out[e] = w[k * d + c];
        }
    }
}
__global__ void embedding_bwd_kernel(const float* __restrict__ dout, const long long* __restrict__ idx, long long n, int d,
                                     long long vocab, float* __restrict__ dw) {
    const long long total = n * d;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long r = e / d;
        const int c = (int)(e - r * d);
        const long long k = idx[r];
        if (k >= 0 && k < vocab) atomicAdd(&dw[k * d + c], dout[e]);
    }
}
// out = x (or 0) + y * s[row],  s[row] = 1 / (m[row] - 1) when m is given, else 1
__global__ void rowscale_add_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ m,
                                    long long rows, int cols, float* __restrict__ out) {
    const long long total = rows * cols;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long r = e / cols;
        float v = y[e];
        if (m) v = v / (m[r] - 1.f);
        out[e] = x ? x[e] + v : v;
    }
}

extern "C" int jova_embedding_fwd(const float* weight, const int64_t* idx, int64_t n, int d, int64_t vocab, float* out,
                                  int* bad_flag, void* stream) {
    JOVA_CHECK_ARG(n >= 0 && d >= 1 && vocab >= 1 && bad_flag && (n == 0 || (weight && idx && out)), "embedding_fwd: bad argument");
    if (n == 0) return JOVA_OK;
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3(ew_blocks(n * d)), dim3(256), 0, (hipStream_t)stream, weight,
                       (const long long*)idx, (long long)n, d, (long long)vocab, out, bad_flag);
    JOVA_CHECK_LAUNCH("embedding_fwd");
    return JOVA_OK;
}
extern "C" int jova_embedding_bwd(const float* dout, const int64_t* idx, int64_t n, int d, int64_t vocab, float* dweight,
                                  void* stream) {
    JOVA_CHECK_ARG(n >= 0 && d >= 1 && vocab >= 1 && (n == 0 || (dout && idx && dweight)), "embedding_bwd: bad argument");
    if (n == 0) return JOVA_OK;
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(ew_blocks(n * d)), dim3(256), 0, (hipStream_t)stream, dout,
                       (const long long*)idx, (long long)n, d, (long long)vocab, dweight);
    JOVA_CHECK_LAUNCH("embedding_bwd");
    return JOVA_OK;
}
extern "C" int jova_rowscale_add(const float* x_or_null, const float* y, const float* m_or_null, int64_t rows, int cols,
                                 float* out, void* stream) {
    JOVA_CHECK_ARG(rows >= 0 && cols >= 1 && (rows == 0 || (y && out)), "rowscale_add: bad argument");
    if (rows == 0) return JOVA_OK;
    hipLaunchKernelGGL(rowscale_add_kernel, dim3(ew_blocks(rows * cols)), dim3(256), 0, (hipStream_t)stream, x_or_null, y,
                       m_or_null, (long long)rows, cols, out);
    JOVA_CHECK_LAUNCH("rowscale_add");
    return JOVA_OK;
}

// ------------------------------------------------------------------------------------------------ Weave pieces (section 8 f3)
// out[r][:] = src[idx[r]][:]  (atom_features[atom_to_pair], jova/nn/layers.py:141-142) and its adjoint
// out[idx[r]][:] += src[r][:]  (torch_scatter's scatter_add(src, index, dim=0), layers.py:135, :191)
__global__ void gather_rows_kernel(const float* __restrict__ src, const long long* __restrict__ idx, long long n, int d,
                                   long long m, float* __restrict__ out) {
    const long long total = n * d;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long r = e / d;
        const int c = (int)(e - r * d);
        const long long k = idx[r];
        out[e] = (k >= 0 && k < m) ? src[k * d + c] : __int_as_float(0x7fc00000);
    }
}
__global__ void scatter_add_rows_kernel(const float* __restrict__ src, const long long* __restrict__ idx, long long n,
                                        int d, long long m, float* __restrict__ out) {
    const long long total = n * d;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long r = e / d;
        const int c = (int)(e - r * d);
        const long long k = idx[r];
        if (k >= 0 && k < m) atomicAdd(&out[k * d + c], src[e]);
    }
}
// WeaveGather.gaussian_histogram (layers.py:198-207): for the 11 fuzzy memberships (mean, std) v_k = log N(x; m_k, s_k) /
// log N(m_k; m_k, s_k); out[.., 11 c + k] = (v_k + eps) / (sum_k v_k + eps)
__constant__ float c_gh_mean[11] = {-1.645f, -1.080f, -0.739f, -0.468f, -0.228f, 0.f, 0.228f, 0.468f, 0.739f, 1.080f, 1.645f};
__constant__ float c_gh_std[11] = {0.283f, 0.170f, 0.134f, 0.118f, 0.114f, 0.114f, 0.114f, 0.118f, 0.134f, 0.170f, 0.283f};
__global__ void gauss_hist_kernel(const float* __restrict__ x, const float* __restrict__ dout, long long n, float eps,
                                  float* __restrict__ out, float* __restrict__ dx) {
    const float half_log_2pi = 0.91893853320467274178f;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const float xv = x[e];
        float v[11], dvdx[11], tot = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float m = c_gh_mean[k], s = c_gh_std[k];
            const float logs = logf(s);
            const float dmax = -logs - half_log_2pi;                          // log_prob at the mean
            const float lp = -((xv - m) * (xv - m)) / (2.f * s * s) - logs - half_log_2pi;
            v[k] = lp / dmax;
            dvdx[k] = -(xv - m) / (s * s) / dmax;
            tot += v[k];
        }
        const float T = tot + eps;
        if (out) {
#pragma unroll
            for (int k = 0; k < 11; ++k) out[e * 11 + k] = (v[k] + eps) / T;
        }
        if (dx) {
            float dot = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) dot += dout[e * 11 + k] * (v[k] + eps);
            float g = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) g += (dout[e * 11 + k] / T - dot / (T * T)) * dvdx[k];
            dx[e] = g;
        }
    }
}

// out[i][:] (+)= sum over e in [row_ptr[i], row_ptr[i+1]) of val[e] * x[col[e]][:]: the product A x of a sparse A in CSR form.
// GraphNeuralNet.update_fingerprints multiplies by the batch's block-diagonal adjacency (jova/nn/models.py:366-371), which
// the reference keeps dense ([N, N], 67 MB at N = 4.1 k: 99.9 % zeros); 16 lanes per row, columns strided over the lanes.
__global__ void __launch_bounds__(256)
spmm_csr_kernel(const int* __restrict__ row_ptr, const int* __restrict__ col, const float* __restrict__ val,
                const float* __restrict__ x, long long n_rows, int d, long long n_src, float* __restrict__ out) {
    const int l16 = threadIdx.x & 15;
    const long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (row >= n_rows) return;
    const int e0 = row_ptr[row], e1 = row_ptr[row + 1];
    for (int c = l16; c < d; c += 16) {
        float acc = 0.f;
        for (int e = e0; e < e1; ++e) {
            const long long j = col[e];
            const float v = val ? val[e] : 1.f;
            acc += (j >= 0 && j < n_src) ? v * x[j * d + c] : __int_as_float(0x7fc00000);
        }
        out[row * d + c] = acc;
    }
}

extern "C" int jova_spmm_csr(const int32_t* row_ptr, const int32_t* col, const float* val_or_null, const float* x,
                             int64_t n_rows, int d, int64_t n_src, float* out, void* stream) {
    JOVA_CHECK_ARG(n_rows >= 0 && d >= 1 && n_src >= 0 && (n_rows == 0 || (row_ptr && x && out)), "spmm_csr: bad argument");
    if (n_rows == 0) return JOVA_OK;
    hipLaunchKernelGGL(spmm_csr_kernel, dim3(cdiv(n_rows, 16)), dim3(256), 0, (hipStream_t)stream, row_ptr, col, val_or_null,
                       x, (long long)n_rows, d, (long long)n_src, out);
    JOVA_CHECK_LAUNCH("spmm_csr");
    return JOVA_OK;
}

extern "C" int jova_gather_rows(const float* src, const int64_t* idx, int64_t n, int d, int64_t m, float* out, void* stream) {
    JOVA_CHECK_ARG(n >= 0 && d >= 1 && m >= 0 && (n == 0 || (src && idx && out)), "gather_rows: bad argument");
    if (n == 0) return JOVA_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(ew_blocks(n * d)), dim3(256), 0, (hipStream_t)stream, src,
                       (const long long*)idx, (long long)n, d, (long long)m, out);
    JOVA_CHECK_LAUNCH("gather_rows");
    return JOVA_OK;
}
extern "C" int jova_scatter_add_rows(const float* src, const int64_t* idx, int64_t n, int d, int64_t m, float* out,
                                     void* stream) {
    JOVA_CHECK_ARG(n >= 0 && d >= 1 && m >= 0 && (n == 0 || (src && idx && out)), "scatter_add_rows: bad argument");
    if (n == 0) return JOVA_OK;
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(ew_blocks(n * d)), dim3(256), 0, (hipStream_t)stream, src,
                       (const long long*)idx, (long long)n, d, (long long)m, out);
    JOVA_CHECK_LAUNCH("scatter_add_rows");
    return JOVA_OK;
}
extern "C" int jova_gauss_hist_fwd(const float* x, int64_t n, float eps, float* out, void* stream) {
    JOVA_CHECK_ARG(n >= 0 && (n == 0 || (x && out)), "gauss_hist_fwd: bad argument");
    if (n == 0) return JOVA_OK;
    hipLaunchKernelGGL(gauss_hist_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, (const float*)nullptr,
                       (long long)n, eps, out, (float*)nullptr);
    JOVA_CHECK_LAUNCH("gauss_hist_fwd");
    return JOVA_OK;
}
extern "C" int jova_gauss_hist_bwd(const float* x, const float* dout, int64_t n, float eps, float* dx, void* stream) {
    JOVA_CHECK_ARG(n >= 0 && (n == 0 || (x && dout && dx)), "gauss_hist_bwd: bad argument");
    if (n == 0) return JOVA_OK;
    hipLaunchKernelGGL(gauss_hist_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, dout, (long long)n, eps,
                       (float*)nullptr, dx);
    JOVA_CHECK_LAUNCH("gauss_hist_bwd");
    return JOVA_OK;
}

// ------------------------------------------------------------------------------------------------ zero fill (see common.h)
__global__ void zero_words_kernel(unsigned* __restrict__ p, long long row_words, long long rows, long long pitch_words) {
    const long long total = row_words * rows;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x)
        p[(e / row_words) * pitch_words + e % row_words] = 0u;
}
__global__ void zero_quads_kernel(uint4* __restrict__ p, long long n_quads) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n_quads; e += (long long)gridDim.x * blockDim.x)
        p[e] = make_uint4(0u, 0u, 0u, 0u);
}
int jova_zero_fill(void* p, size_t row_bytes, size_t rows, size_t pitch_bytes, hipStream_t s) {
    JOVA_CHECK_ARG((row_bytes & 3) == 0 && (pitch_bytes & 3) == 0 && pitch_bytes >= row_bytes && (p || row_bytes * rows == 0),
                   "zero_fill: bad argument");
    if (row_bytes == 0 || rows == 0) return JOVA_OK;
    if ((rows == 1 || pitch_bytes == row_bytes) && ((row_bytes * rows) & 15) == 0 && (((uintptr_t)p) & 15) == 0) {
        const long long nq = (long long)(row_bytes * rows / 16);
        hipLaunchKernelGGL(zero_quads_kernel, dim3(ew_blocks(nq)), dim3(256), 0, s, (uint4*)p, nq);
    } else {
        const long long rw = (long long)(row_bytes / 4);
        hipLaunchKernelGGL(zero_words_kernel, dim3(ew_blocks(rw * (long long)rows)), dim3(256), 0, s, (unsigned*)p, rw,
                           (long long)rows, (long long)(pitch_bytes / 4));
    }
    JOVA_CHECK_LAUNCH("zero_fill");
    return JOVA_OK;
}

// ------------------------------------------------------------------------------------------------ error channel
#include <stdarg.h>
static thread_local char g_err[512] = "";
void jova_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* jova_last_error(void) { return g_err; }
extern "C" int jova_abi_version(void) { return JOVA_HIP_ABI_VERSION; }
