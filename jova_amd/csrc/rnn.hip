// ProteinRNN (jova/nn/models.py:657-682): a single-layer, one-directional nn.LSTM over the protein segments with zero
// initial state, section 8(f3).  The input part of the gates for ALL time steps is one GEMM (gx = x W_ih^T + b_ih + b_hh,
// done by the caller with jova_gemm_f32); what is sequential is here: one workgroup per sequence, thread j owns gate row j
// (PyTorch's order i | f | g | o), W_hh in LDS (row-major for the forward, transposed for the backward), h in LDS.
// Two barriers per step; the hidden size of the reference is 27 (train_joint_attn.py:1051), the kernels take H <= 64.
#include "common.h"
#include "jova_hip.h"

namespace {

constexpr int LSTM_MAX_H = 64;

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + __expf(-v)); }

// gx [T, B, 4H] (bias included), w_hh [4H, H] -> h [T, B, H], gates (activated i, f, g, o) [T, B, 4H], c [T, B, H]
__global__ void __launch_bounds__(256)
lstm_fwd_kernel(const float* __restrict__ gx, const float* __restrict__ w_hh, int T, int B, int H,
                float* __restrict__ h_out, float* __restrict__ gates, float* __restrict__ c_out) {
    __shared__ float w[4 * LSTM_MAX_H * (LSTM_MAX_H + 1)];
    __shared__ float hs[LSTM_MAX_H];
    __shared__ float g[4 * LSTM_MAX_H];
    const int b = blockIdx.x, j = threadIdx.x, G = 4 * H, ld = H + 1;
    for (int e = j; e < G * H; e += blockDim.x) w[(e / H) * ld + (e % H)] = w_hh[e];
    if (j < H) hs[j] = 0.f;
    float c = 0.f;
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const size_t row = (size_t)t * B + b;
        if (j < G) {
            float a = gx[row * G + j];
            const float* wr = w + j * ld;
            for (int k = 0; k < H; ++k) a += wr[k] * hs[k];
            const float v = (j >= 2 * H && j < 3 * H) ? tanhf(a) : sigmoidf_(a);
            g[j] = v;
            gates[row * G + j] = v;
        }
        __syncthreads();
        if (j < H) {
            c = g[H + j] * c + g[j] * g[2 * H + j];
            const float h = g[3 * H + j] * tanhf(c);
            hs[j] = h;
            h_out[row * H + j] = h;
            c_out[row * H + j] = c;
        }
        __syncthreads();
    }
}

// dh_in [T, B, H] (gradient w.r.t. every h_t from above) -> dgx [T, B, 4H] (gradient w.r.t. the gate pre-activations)
__global__ void __launch_bounds__(256)
lstm_bwd_kernel(const float* __restrict__ dh_in, const float* __restrict__ w_hh, const float* __restrict__ gates,
                const float* __restrict__ c_seq, int T, int B, int H, float* __restrict__ dgx) {
    __shared__ float wt[LSTM_MAX_H * (4 * LSTM_MAX_H + 1)];      // transposed: wt[k][j] = w_hh[j][k]
    __shared__ float dg[4 * LSTM_MAX_H];
    __shared__ float dhs[LSTM_MAX_H];
    const int b = blockIdx.x, j = threadIdx.x, G = 4 * H, ld = G + 1;
    for (int e = j; e < G * H; e += blockDim.x) wt[(e % H) * ld + (e / H)] = w_hh[e];
    if (j < H) dhs[j] = 0.f;
    float dc = 0.f;
    __syncthreads();
    for (int t = T - 1; t >= 0; --t) {
        const size_t row = (size_t)t * B + b;
        if (j < H) {
            const float gi = gates[row * G + j], gf = gates[row * G + H + j], gg = gates[row * G + 2 * H + j],
                        go = gates[row * G + 3 * H + j];
            const float c = c_seq[row * H + j];
            const float c_prev = t > 0 ? c_seq[(row - B) * H + j] : 0.f;
            const float tc = tanhf(c);
            const float dh = dh_in[row * H + j] + dhs[j];
            const float d_o = dh * tc;
            dc += dh * go * (1.f - tc * tc);
            const float d_i = dc * gg, d_f = dc * c_prev, d_g = dc * gi;
            dc *= gf;
            const float pi = d_i * gi * (1.f - gi), pf = d_f * gf * (1.f - gf), pg = d_g * (1.f - gg * gg),
                        po = d_o * go * (1.f - go);
            dg[j] = pi; dg[H + j] = pf; dg[2 * H + j] = pg; dg[3 * H + j] = po;
            dgx[row * G + j] = pi; dgx[row * G + H + j] = pf; dgx[row * G + 2 * H + j] = pg; dgx[row * G + 3 * H + j] = po;
        }
        __syncthreads();
        if (j < H) {
            float a = 0.f;
            const float* wr = wt + j * ld;
            for (int q = 0; q < G; ++q) a += wr[q] * dg[q];
            dhs[j] = a;
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int jova_lstm_fwd(const float* gx, const float* w_hh, int seq_len, int batch, int hidden, float* h_out,
                             float* gates, float* c_out, void* stream) {
    JOVA_CHECK_ARG(seq_len >= 0 && batch >= 0 && hidden >= 1, "lstm_fwd: bad shape");
    JOVA_CHECK_ARG(hidden <= LSTM_MAX_H, "lstm_fwd: hidden size %d exceeds %d", hidden, LSTM_MAX_H);
    if (seq_len == 0 || batch == 0) return JOVA_OK;
    JOVA_CHECK_ARG(gx && w_hh && h_out && gates && c_out, "lstm_fwd: null pointer");
    hipLaunchKernelGGL(lstm_fwd_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, gx, w_hh, seq_len, batch, hidden,
                       h_out, gates, c_out);
    JOVA_CHECK_LAUNCH("lstm_fwd");
    return JOVA_OK;
}

extern "C" int jova_lstm_bwd(const float* dh, const float* w_hh, const float* gates, const float* c_seq, int seq_len,
                             int batch, int hidden, float* dgx, void* stream) {
    JOVA_CHECK_ARG(seq_len >= 0 && batch >= 0 && hidden >= 1, "lstm_bwd: bad shape");
    JOVA_CHECK_ARG(hidden <= LSTM_MAX_H, "lstm_bwd: hidden size %d exceeds %d", hidden, LSTM_MAX_H);
    if (seq_len == 0 || batch == 0) return JOVA_OK;
    JOVA_CHECK_ARG(dh && w_hh && gates && c_seq && dgx, "lstm_bwd: null pointer");
    hipLaunchKernelGGL(lstm_bwd_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, dh, w_hh, gates, c_seq, seq_len,
                       batch, hidden, dgx);
    JOVA_CHECK_LAUNCH("lstm_bwd");
    return JOVA_OK;
}
