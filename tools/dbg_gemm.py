import sys, torch
sys.path.insert(0, '.')
from jova_amd.nn import functional as F_
dev = torch.device('cuda:0')
torch.set_printoptions(linewidth=200, precision=3, sci_mode=False)
M, N, K = 5, 7, 3
a = torch.arange(M * K, dtype=torch.float32).view(M, K) + 1
b = torch.zeros(K, N); b[0, 0] = 1; b[1, 2] = 1; b[2, 5] = 1
A = a.to(dev); B = b.t().contiguous().to(dev)
C = torch.full((M, N), float('nan'), device=dev)
F_.gemm(F_._addr(A), K, 1, F_._addr(B), 1, K, F_._addr(C), N, M, N, K)
print(C.cpu()); print(a @ b)
