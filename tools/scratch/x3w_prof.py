"""Phase cycles of gemm_x3w2_kernel from the X3W_PROFILE build (JOVA_LIB=tools/scratch/libs/lib_prof.so), per shape."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from jova_amd.nn import functional as F_
from jova_amd import _lib
dev = torch.device('cuda:0')
T, L, I = 221 * 128, 512, 2048
x = torch.randn(T, L, device=dev); w1 = torch.randn(I, L, device=dev) * 0.05; b1 = torch.randn(I, device=dev)
w2 = torch.randn(L, I, device=dev) * 0.05; b2 = torch.randn(L, device=dev)
hid = torch.empty(T, I, device=dev); out = torch.empty(T, L, device=dev)
dh = torch.empty(T, I, device=dev); dx = torch.empty(T, L, device=dev)
L_ = _lib.lib()
L_.jova_debug_x3w_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
n = 512 * 4 * 8
names = ["wait", "split", "barrier", "issue", "products", "loop total", "epilogue"]
shapes = [("fwd1 N=2048 K=512 bias relu", lambda: F_.linear_fwd(x, w1, b1, hid, relu=True), 512),
          ("fwd2 N=512 K=2048 residual", lambda: F_.linear_fwd(hid, w2, b2, out, residual=x), 2048),
          ("dx2 N=2048 K=512 relu mask", lambda: F_.linear_bwd_data(out, w2, dh, relu_mask=hid), 512),
          ("dx1 N=512 K=2048", lambda: F_.linear_bwd_data(dh, w1, dx), 2048)]
for name, fn, K in shapes:
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * n)()
    assert L_.jova_debug_x3w_prof(buf, n) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(512, 4, 8).astype(np.float64)
    a = a[(a[:, 0, 5] > 0) & (a[:, 0, 5] < 1e9)]
    nkb = K // 32
    print("%s: %.0f us (instrumented), %d workgroups recorded, %d blocks per tile" % (name, e0.elapsed_time(e1) * 1e3, len(a), nkb))
    print("   " + "  ".join("%s %.0f" % (nm, np.median(a[:, :, k]) / (nkb if k < 6 else 1)) for k, nm in enumerate(names)),
          " | tile %.0f cycles" % np.median(a[:, :, 5] + a[:, :, 6]))
    t0 = a[:, 0, 7]
    print("   start spread of the recorded workgroups: %.0f cycles; tile ends spread %.0f" % (t0.max() - t0.min(), (t0 + a[:, 0, 5] + a[:, 0, 6]).max() - (t0 + a[:, 0, 5] + a[:, 0, 6]).min()))
