"""Per-(b,h) attention products with the tokens laid out [S, B, 3L] (as now) and [B, S, 3L]: is the 0.8 MB row stride
of the first what holds them at ~70 TFLOP/s?  GPU box only; read the durations from a kernel trace."""
import os, sys, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from jova_amd.nn import functional as F_

dev = torch.device('cuda:0')
S, bsz, H, hd = 241, 128, 2, 256
L = H * hd
Sp = (S + 3) // 4 * 4
qkv = torch.randn(S * bsz, 3 * L, device=dev)
probs = torch.empty(bsz * H, S, Sp, device=dev)
o = torch.empty(S * bsz, L, device=dev)
a = F_._addr
for rep in range(3):
    # scores, seq-major then batch-major
    F_.gemm(a(qkv), bsz * 3 * L, 1, a(qkv, L), 1, bsz * 3 * L, a(probs), Sp, S, S, hd, alpha=0.0625, nb1=bsz, nb2=H,
            sA=(3 * L, hd), sB=(3 * L, hd), sC=(H * S * Sp, S * Sp))
    F_.gemm(a(qkv), 3 * L, 1, a(qkv, L), 1, 3 * L, a(probs), Sp, S, S, hd, alpha=0.0625, nb1=bsz, nb2=H,
            sA=(S * 3 * L, hd), sB=(S * 3 * L, hd), sC=(H * S * Sp, S * Sp))
    # o = P V, seq-major then batch-major
    F_.gemm(a(probs), Sp, 1, a(qkv, 2 * L), bsz * 3 * L, 1, a(o), bsz * L, S, hd, S, nb1=bsz, nb2=H,
            sA=(H * S * Sp, S * Sp), sB=(3 * L, hd), sC=(L, hd))
    F_.gemm(a(probs), Sp, 1, a(qkv, 2 * L), 3 * L, 1, a(o), L, S, hd, S, nb1=bsz, nb2=H,
            sA=(H * S * Sp, S * Sp), sB=(S * 3 * L, hd), sC=(S * L, hd))
    # dv = P^T dO (A row-contiguous), seq-major then batch-major
    F_.gemm(a(probs), 1, Sp, a(o), bsz * L, 1, a(qkv, 2 * L), bsz * 3 * L, S, hd, S, nb1=bsz, nb2=H,
            sA=(H * S * Sp, S * Sp), sB=(L, hd), sC=(3 * L, hd))
    F_.gemm(a(probs), 1, Sp, a(o), L, 1, a(qkv, 2 * L), 3 * L, S, hd, S, nb1=bsz, nb2=H,
            sA=(H * S * Sp, S * Sp), sB=(S * L, hd), sC=(S * 3 * L, hd))
torch.cuda.synchronize()
