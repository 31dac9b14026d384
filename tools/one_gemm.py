import sys, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from jova_amd.nn import functional as F_
dev = torch.device('cuda:0')
T, L, I = 279 * 128, 512, 2048
which = sys.argv[1] if len(sys.argv) > 1 else "ffn1"
x = torch.randn(T, L, device=dev); w1 = torch.randn(I, L, device=dev); b1 = torch.randn(I, device=dev)
hid = torch.empty(T, I, device=dev); h = torch.randn(T, I, device=dev); dw1 = torch.zeros(I, L, device=dev)
for _ in range(5):
    if which == "ffn1":
        F_.linear_fwd(x, w1, b1, hid, relu=True)
    else:
        F_.linear_bwd_weight(h, x, dw1, None)
torch.cuda.synchronize()
