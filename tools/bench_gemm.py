"""GEMM micro-benchmark on the shapes of the Davis joint-attention step (dev tool; GPU box only)."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from jova_amd.nn import functional as F_

dev = torch.device('cuda:0')
T, L, I, B, H, S = 279 * 128, 512, 2048, 128, 2, 279
Sp = 280


def timeit(fn, reps=10):
    fn(); fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


x = torch.randn(T, L, device=dev); w1 = torch.randn(I, L, device=dev); b1 = torch.randn(I, device=dev)
h = torch.randn(T, I, device=dev); w2 = torch.randn(L, I, device=dev); o = torch.empty(T, L, device=dev)
hid = torch.empty(T, I, device=dev); dw1 = torch.zeros(I, L, device=dev); dw2 = torch.zeros(L, I, device=dev)
qkv = torch.randn(T, 3 * L, device=dev); probs = torch.empty(B * H, S, Sp, device=dev)
win = torch.randn(3 * L, L, device=dev)
cases = [
    ("ffn1 fwd  NT [T,512]x[2048,512]^T", lambda: F_.linear_fwd(x, w1, b1, hid, relu=True), 2. * T * L * I),
    ("ffn2 fwd  NT [T,2048]x[512,2048]^T", lambda: F_.linear_fwd(h, w2, None, o), 2. * T * L * I),
    ("qkv  fwd  NT [T,512]x[1536,512]^T", lambda: F_.linear_fwd(x, win, None, qkv), 2. * T * L * 3 * L),
    ("ffn2 dX   NN [T,512]x[512,2048]", lambda: F_.linear_bwd_data(x, w2, hid), 2. * T * L * I),
    ("ffn1 dX   NN [T,2048]x[2048,512]", lambda: F_.linear_bwd_data(h, w1, o), 2. * T * L * I),
    ("ffn1 dW   TN [2048,T]x[T,512] splitk", lambda: F_.linear_bwd_weight(h, x, dw1, None), 2. * T * L * I),
    ("ffn2 dW   TN [512,T]x[T,2048] splitk", lambda: F_.linear_bwd_weight(x, h, dw2, None), 2. * T * L * I),
    ("scores    NT batched 256x[279,256]x[279,256]^T",
     lambda: F_.gemm(F_._addr(qkv), B * 3 * L, 1, F_._addr(qkv, L), 1, B * 3 * L, F_._addr(probs), Sp, S, S, 256,
                     nb1=B, nb2=H, sA=(3 * L, 256), sB=(3 * L, 256), sC=(H * S * Sp, S * Sp)), 2. * B * H * S * S * 256),
    ("PV        NN batched 256x[279,279]x[279,256]",
     lambda: F_.gemm(F_._addr(probs), Sp, 1, F_._addr(qkv, 2 * L), B * 3 * L, 1, F_._addr(o), B * L, S, 256, S,
                     nb1=B, nb2=H, sA=(H * S * Sp, S * Sp), sB=(3 * L, 256), sC=(L, 256)), 2. * B * H * S * S * 256),
]
for name, fn, flops in cases:
    ms = timeit(fn)
    print("%-52s %8.1f us  %6.1f TFLOP/s" % (name, ms * 1e3, flops / ms / 1e9))
