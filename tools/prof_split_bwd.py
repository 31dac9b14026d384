"""Per-phase cycle stamps of graphconv_bwd_split_kernel (profiling build: JOVA_LIB=jova_amd/libjova_hip_prof.so)."""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import ctypes, numpy as np, torch
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'one_conv_fwd.py')).read().split("for layer, x, ld, fin in")[0])
layer, x, ld, fin = conv1, x1, x1.shape[1], 75
w = [l.weight for l in layer.linear_layers]
dw = [torch.zeros_like(t) for t in w]; db = [torch.zeros(64, device=dev) for _ in w]
dz = torch.randn(n, 64, device=dev)
dws = torch.empty(int(lib().jova_graphconv_bwd_scratch_floats()), device=dev)
for _ in range(5):
    call("jova_graphconv_bwd", P(x), 0, ld, n, fin, 64, ds, adj, ptr_array(w), P(dz), None, ptr_array(dw), ptr_array(db),
         None, None, P(topo.sym_flag), dsh, P(dws), None, st)
torch.cuda.synchronize()
buf = np.zeros(256 * 8 * 8 + 512, dtype=np.uint64)
lib().jova_debug_wave_prof(ctypes.c_void_p(buf.ctypes.data), buf.size)
a = buf[:256 * 64].reshape(256, 8, 8).astype(np.float64)
print("role deg  waves tiles/wave | per tile: wait   read   mfma+side | total/tile | wave life us (mean, max)")
for role in (0, 1):
    for d in range(11):
        m = (a[:, :, 7] == d) & (a[:, :, 6] > 0) & (a[:, :, 2] == role)
        if m.sum() == 0: continue
        r = a[m]; nt = r[:, 6]
        print("%4s %3d %6d %8.1f | %9.0f %6.0f (planes %5.0f) %9.0f | %8.0f | %6.1f %6.1f" % (
            "rel" if role else "self", d, m.sum(), nt.mean(), r[:, 0].sum() / nt.sum(), r[:, 1].sum() / nt.sum(), r[:, 5].sum() / nt.sum(),
            r[:, 3].sum() / nt.sum(), (r[:, 0] + r[:, 1] + r[:, 3]).sum() / nt.sum(), r[:, 4].mean() / 100, r[:, 4].max() / 100))
