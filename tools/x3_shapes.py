"""Backward products of the Davis step, one launch each, for a kernel trace: fp32 MFMA kernel (JOVA_GEMM_NO_BF16X3=1)
against the bf16-pieces kernel (JOVA_GEMM_BF16X3_ALL=1).  GPU box only."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from jova_amd.nn import functional as F_

dev = torch.device('cuda:0')
T = 30848
for rep in range(4):
    for rows, n_out, k_in in [(T, 2048, 512), (T, 512, 2048), (T, 1536, 512), (T, 512, 512), (23168, 256, 256)]:
        dy, x = torch.randn(rows, n_out, device=dev), torch.randn(rows, k_in, device=dev)
        w = torch.randn(n_out, k_in, device=dev)
        dw, db, dx = torch.zeros(n_out, k_in, device=dev), torch.zeros(n_out, device=dev), torch.empty(rows, k_in, device=dev)
        F_.linear_bwd_data(dy, w, dx)
        F_.linear_bwd_weight(dy, x, dw, db)
torch.cuda.synchronize()
