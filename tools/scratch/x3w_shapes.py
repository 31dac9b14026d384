"""The four big product shapes of the Davis step (FFN linear1 / linear2, forward and data gradient), 12 launches each
(kernel trace target; distinct grids tell them apart: T = 221 * 128 tokens)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from jova_amd.nn import functional as F_
dev = torch.device('cuda:0')
T, L, I = 221 * 128, 512, 2048
x = torch.randn(T, L, device=dev); w1 = torch.randn(I, L, device=dev) * 0.05; b1 = torch.randn(I, device=dev)
w2 = torch.randn(L, I, device=dev) * 0.05; b2 = torch.randn(L, device=dev)
hid = torch.empty(T, I, device=dev); out = torch.empty(T, L, device=dev)
dh = torch.empty(T, I, device=dev); dx = torch.empty(T, L, device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for _ in range(n):
    F_.linear_fwd(x, w1, b1, hid, relu=True)              # 1: [T,512] x [2048,512]^T
    F_.linear_fwd(hid, w2, b2, out, residual=x)           # 2: [T,2048] x [512,2048]^T
    F_.linear_bwd_data(out, w2, dh, relu_mask=hid)        # 3: [T,512] x [512,2048]
    F_.linear_bwd_data(dh, w1, dx)                        # 4: [T,2048] x [2048,512]
torch.cuda.synchronize()
