"""FFN linear1 forward ([T,512] x [2048,512]^T + bias, ReLU; image form) timed with HIP events over n launches:
python tools/scratch/x3w_time.py [S] [n].  JOVA_LIB selects a lab build."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from jova_amd.nn import functional as F_
dev = torch.device('cuda:0')
S = int(sys.argv[1]) if len(sys.argv) > 1 else 221
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
T, L, I = S * 128, 512, 2048
x = torch.randn(T, L, device=dev); w1 = torch.randn(I, L, device=dev) * 0.05; b1 = torch.randn(I, device=dev)
hid = torch.empty(T, I, device=dev)
for _ in range(5):
    F_.linear_fwd(x, w1, b1, hid, relu=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(n):
    F_.linear_fwd(x, w1, b1, hid, relu=True)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / n * 1e3
print(os.path.basename(os.environ.get("JOVA_LIB", "default")), "S=%d FFN1 fwd %.1f us incl. the image launch (%.0f TF-eq)" % (S, us, 2.0 * T * I * L / us / 1e6), flush=True)
