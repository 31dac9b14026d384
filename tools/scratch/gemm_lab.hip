// Dev tool: times jova_gemm_f32 on the shapes of the Davis joint-attention step, standalone (no Python).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include -I jova_amd/csrc [-D variants] tools/scratch/gemm_lab.hip -o tools/scratch/gemm_lab
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "../../jova_amd/csrc/gemm.hip"

void jova_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fprintf(stderr, "\n");
}

static float* dalloc(size_t n, bool rnd) {
    float* p;
    if (hipMalloc(&p, n * sizeof(float)) != hipSuccess) { fprintf(stderr, "alloc failed\n"); exit(1); }
    if (rnd) {
        std::vector<float> h(n);
        unsigned s = 12345u + (unsigned)n;
        for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xffff) / 32768.f - 1.f; }
        hipMemcpy(p, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
    } else hipMemset(p, 0, n * sizeof(float));
    return p;
}

template <typename F>
static double timeit(F fn, int reps = 10) {
    fn(); fn();
    hipDeviceSynchronize();
    std::vector<float> t(reps);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(a, 0); fn(); hipEventRecord(b, 0);
        hipEventSynchronize(b);
        hipEventElapsedTime(&t[r], a, b);
    }
    std::sort(t.begin(), t.end());
    return t[reps / 2] * 1e3;   // us
}

int main(int argc, char** argv) {
    const int S = 279, Bt = 128, T = S * Bt, L = 512, I = 2048, H = 2, Sp = 280;
    float* x = dalloc((size_t)T * L, true);
    float* w1 = dalloc((size_t)I * L, true);
    float* b1 = dalloc(I, true);
    float* hid = dalloc((size_t)T * I, true);
    float* w2 = dalloc((size_t)L * I, true);
    float* hidp = dalloc((size_t)T * (I + 32), true);
    float* w2p = dalloc((size_t)I * (I + 32), true);
    float* o = dalloc((size_t)T * L, false);
    float* dw1 = dalloc((size_t)I * L, false);
    float* qkv = dalloc((size_t)T * 3 * L, true);
    float* probs = dalloc((size_t)Bt * H * S * Sp, true);
    float* win = dalloc((size_t)3 * L * L, true);
    const char* only = argc > 1 ? argv[1] : "";
    struct Case { const char* name; double flops; std::function<int()> fn; };
    std::vector<Case> cases = {
        {"ffn1_fwd NT [T,512]x[2048,512]^T +bias+relu", 2. * T * L * I,
         [&] { return jova_gemm_f32(x, L, 1, w1, 1, L, hid, I, T, I, L, 1.f, b1, nullptr, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, nullptr); }},
        {"ffn2_fwd NT [T,2048]x[512,2048]^T +res", 2. * T * L * I,
         [&] { return jova_gemm_f32(hid, I, 1, w2, 1, I, o, L, T, L, I, 1.f, nullptr, x, L, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, nullptr); }},
        {"ffn2pad NT same with ld = K+32 (A, B)", 2. * T * L * I,
         [&] { return jova_gemm_f32(hidp, I + 32, 1, w2p, 1, I + 32, o, L, T, L, I, 1.f, nullptr, x, L, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, nullptr); }},
        {"ffn1pad NT same with ld = K+32 (A, B)", 2. * T * L * I,
         [&] { return jova_gemm_f32(hidp, L + 32, 1, w2p, 1, L + 32, hid, I, T, I, L, 1.f, b1, nullptr, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, nullptr); }},
        {"ffn1padA NT only A padded", 2. * T * L * I,
         [&] { return jova_gemm_f32(hidp, L + 32, 1, w1, 1, L, hid, I, T, I, L, 1.f, b1, nullptr, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, nullptr); }},
        {"ffn1padB NT only B padded", 2. * T * L * I,
         [&] { return jova_gemm_f32(x, L, 1, w2p, 1, L + 32, hid, I, T, I, L, 1.f, b1, nullptr, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, nullptr); }},
        {"ffn1padC NT only C padded (ldc+32)", 2. * T * L * I,
         [&] { return jova_gemm_f32(x, L, 1, w1, 1, L, hidp, I + 32, T, I, L, 1.f, b1, nullptr, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, nullptr); }},
        {"qkv_fwd  NT [T,512]x[1536,512]^T", 2. * T * L * 3 * L,
         [&] { return jova_gemm_f32(x, L, 1, win, 1, L, qkv, 3 * L, T, 3 * L, L, 1.f, nullptr, nullptr, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, nullptr); }},
        {"ffn2_dX  NN [T,512]x[512,2048]", 2. * T * L * I,
         [&] { return jova_gemm_f32(x, L, 1, w2, I, 1, hid, I, T, I, L, 1.f, nullptr, nullptr, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, nullptr); }},
        {"ffn1_dX  NN [T,2048]x[2048,512]", 2. * T * L * I,
         [&] { return jova_gemm_f32(hid, I, 1, w1, L, 1, o, L, T, L, I, 1.f, nullptr, nullptr, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, nullptr); }},
        {"ffn1_dW  TN [2048,T]x[T,512] splitk", 2. * T * L * I,
         [&] { const int sk = 8;
               return jova_gemm_f32(hid, 1, I, x, L, 1, dw1, L, I, L, T, 1.f, nullptr, nullptr, 0, 0, sk, 1, 1, 0, 0, 0, 0, 0, 0, nullptr); }},
        {"scores   NT batched 256x[279,256]x[279,256]^T", 2. * Bt * H * S * S * 256,
         [&] { return jova_gemm_f32(qkv, (long long)Bt * 3 * L, 1, qkv + L, 1, (long long)Bt * 3 * L, probs, Sp, S, S, 256, 1.f, nullptr, nullptr, 0, 0, 1,
                                    Bt, H, 3 * L, 256, 3 * L, 256, (long long)H * S * Sp, (long long)S * Sp, nullptr); }},
        {"PV       NN batched 256x[279,279]x[279,256]", 2. * Bt * H * S * S * 256,
         [&] { return jova_gemm_f32(probs, Sp, 1, qkv + 2 * L, (long long)Bt * 3 * L, 1, o, (long long)Bt * L, S, 256, S, 1.f, nullptr, nullptr, 0, 0, 1,
                                    Bt, H, (long long)H * S * Sp, (long long)S * Sp, 3 * L, 256, L, 256, nullptr); }},
    };
    for (int r = 0; r < 400; ++r) cases[0].fn();      // ~0.3 s of load first: the first launches run at a lower clock
    hipDeviceSynchronize();
    for (auto& c : cases) {
        if (*only && !strstr(c.name, only)) continue;
        if (c.fn() != 0) { printf("%s: launch error\n", c.name); continue; }
        const double us = timeit(c.fn);
        printf("%-50s %8.1f us  %6.1f TFLOP/s  %.1f%%", c.name, us, c.flops / us / 1e6, c.flops / us / 1e6 / 1.573);
#ifdef GEMM_CLOCK_PROBE
        static unsigned long long pr[4096];
        hipMemcpyFromSymbol(pr, HIP_SYMBOL(gemm_probe), sizeof(pr));
        unsigned long long t0 = ~0ull, t1 = 0; double avg = 0; unsigned long long dmin = ~0ull, dmax = 0;
        for (int w = 0; w < 512; ++w) { t0 = std::min(t0, pr[4*w]); t1 = std::max(t1, pr[4*w+1]); }
        int late = 0;
        for (int w = 0; w < 512; ++w) { unsigned long long d = pr[4*w+1] - pr[4*w]; avg += d / 512.0; dmin = std::min(dmin, d); dmax = std::max(dmax, d); if (pr[4*w] - t0 > 1000) ++late; }
        printf("   span %.1f us; wg life avg %.1f min %.1f max %.1f us; %d started >10us late", (t1 - t0) / 100.0, avg / 100.0, dmin / 100.0, dmax / 100.0, late);
        if (getenv("DUMP")) { printf("\n"); for (int w = 0; w < 512; ++w) printf("wg %d start %.1f end %.1f cyc %llu hwid %08llx xcc %llu\n", w, (pr[4*w]-t0)/100.0, (pr[4*w+1]-t0)/100.0, pr[4*w+2], pr[4*w+3] & 0xffffffffull, pr[4*w+3] >> 32); }
#endif
        printf("\n");
    }
    return 0;
}
