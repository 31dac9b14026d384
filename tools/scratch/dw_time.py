"""The two big weight-gradient shapes of the Davis step (FFN: dW[2048,512] over T tokens; QKV: dW[1536,512]), timed with
HIP events over n launches each: python tools/scratch/dw_time.py [S] [n].  JOVA_LIB selects a lab build."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from jova_amd.nn import functional as F_
dev = torch.device('cuda:0')
S = int(sys.argv[1]) if len(sys.argv) > 1 else 221
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
T, L = S * 128, 512
ZERO = os.environ.get('DW_ZERO') == '1'
x = torch.zeros(T, L, device=dev) if ZERO else torch.randn(T, L, device=dev)
out = []
for I in (2048, 1536):
    dy = torch.zeros(T, I, device=dev) if ZERO else torch.randn(T, I, device=dev); dw = torch.zeros(I, L, device=dev); db = torch.zeros(I, device=dev)
    for _ in range(5):
        F_.linear_bwd_weight(dy, x, dw, db)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        F_.linear_bwd_weight(dy, x, dw, db)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    if os.environ.get("DW_CLOCK") == "1":
        import ctypes, numpy as np
        from jova_amd import _lib
        buf = (ctypes.c_uint64 * 1024)()
        assert _lib.lib().jova_debug_dw_clk(buf, 1024) == 0
        a = np.array(buf[:], dtype=np.float64).reshape(256, 4)
        assert _lib.lib().jova_debug_dw_clk2(buf, 1024) == 0
        b = np.array(buf[:], dtype=np.float64).reshape(256, 4) * 0.01
        t0 = b[:, 0].min()
        out.append("entry spread %.1f, prologue %.1f, loop %.1f (min %.1f max %.1f), epilogue %.1f, first entry to last exit %.1f us"
                   % (b[:, 0].max() - t0, (b[:, 1] - b[:, 0]).mean(), (b[:, 2] - b[:, 1]).mean(), (b[:, 2] - b[:, 1]).min(),
                      (b[:, 2] - b[:, 1]).max(), (b[:, 3] - b[:, 2]).mean(), b[:, 3].max() - t0))
        out.append("clock %.2f GHz, %.0f cycles/slab, launch span %.1f us" % ((a[:, 0] / a[:, 1]).mean() * 0.1, (a[:, 0] / a[:, 2]).mean(),
                   (a[:, 3].max() + a[a[:, 3].argmax(), 1] - a[:, 3].min()) * 0.01))
    out.append("dW[%d,512] %.1f us (%.0f TF-eq)" % (I, us, 2.0 * T * I * L / us / 1e6))
print(os.path.basename(os.environ.get("JOVA_LIB", "default")), "S=%d%s" % (S, " zeros" if ZERO else ""), " | ".join(out), flush=True)
