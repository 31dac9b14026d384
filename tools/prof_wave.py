"""Per-phase cycle stamps of the wave-private conv forward (profiling build: JOVA_LIB=jova_amd/libjova_hip_prof.so)."""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import ctypes, numpy as np, torch
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'one_conv_fwd.py')).read().split("for layer, x, ld, fin in")[0])
layer, x, ld, fin = (conv2, x2, 64, 64) if 'l1' not in sys.argv else (conv1, x1, x1.shape[1], 75)
w = [l.weight for l in layer.linear_layers]; b = [l.bias for l in layer.linear_layers]
ws = torch.empty(int(lib().jova_graphconv_workspace_floats(fin, 64)), device=dev)
for _ in range(5):
    call("jova_graphconv_fwd", P(x), 0, ld, n, fin, 64, ds, adj, ptr_array(w), ptr_array(b), 1, P(out), None, P(ws), dsh, st)
torch.cuda.synchronize()
buf = np.zeros(256 * 8 * 8 + 512, dtype=np.uint64)
lib().jova_debug_wave_prof(ctypes.c_void_p(buf.ctypes.data), buf.size)
a = buf[:256 * 64].reshape(256, 8, 8).astype(np.float64)
wall = buf[256 * 64:].reshape(256, 2).astype(np.float64)
wall = wall[wall[:, 0] > 0]
print('workgroup entry spread %.1f us, first entry -> last exit %.1f us, mean life %.1f us; prologue (entry -> loop) mean %.0f cycles' % ((wall[:, 0].max() - wall[:, 0].min()) / 100, (wall[:, 1].max() - wall[:, 0].min()) / 100, (wall[:, 1] - wall[:, 0]).mean() / 100, a[:, :, 2][a[:, :, 6] > 0].mean()))
print("deg  waves tiles/wave | per tile: wait  read  (n/a)  mfma+side epi | total/tile   wave life (cycles)")
for d in range(11):
    m = (a[:, :, 7] == d) & (a[:, :, 6] > 0)
    if m.sum() == 0: continue
    r = a[m]
    nt = r[:, 6]
    print("%3d %6d %8.1f | %9.0f %5.0f %6.0f %5.0f %5.0f | %8.0f   %8.0f (max %.0f)" % (
        d, m.sum(), nt.mean(), *(r[:, k].sum() / nt.sum() for k in range(5)), r[:, 5].sum() / nt.sum(), r[:, 5].mean(), r[:, 5].max()))
