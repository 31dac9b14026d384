"""Per-phase cycle stamps of the wave-private conv backward (profiling build: JOVA_LIB=jova_amd/libjova_hip_prof.so,
built with -DJOVA_WAVE_PROFILE).  python tools/prof_wave_bwd.py [l1]"""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import ctypes, numpy as np, torch
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'one_conv_fwd.py')).read().split("for layer, x, ld, fin in")[0])
first = 'l1' in sys.argv
layer, x, ld, fin = (conv2, x2, 64, 64) if not first else (conv1, x1, x1.shape[1], 75)
w = [l.weight for l in layer.linear_layers]
dw = [torch.zeros_like(t) for t in w]; db = [torch.zeros(64, device=dev) for _ in w]
dz = torch.randn(n, 64, device=dev)
dx = None if first else torch.empty(n, 64, device=dev); gb = None if first else torch.empty(n, 64, device=dev)
dws = torch.empty(int(lib().jova_graphconv_bwd_scratch_floats()), device=dev)
sym = topo.sym_flag
for _ in range(5):
    call("jova_graphconv_bwd", P(x), 0, ld, n, fin, 64, ds, adj, ptr_array(w), P(dz), None, ptr_array(dw), ptr_array(db),
         P(dx), P(gb), P(sym), dsh, P(dws), None, st)
torch.cuda.synchronize()
buf = np.zeros(256 * 8 * 8 + 512, dtype=np.uint64)
lib().jova_debug_wave_prof(ctypes.c_void_p(buf.ctypes.data), buf.size)
a = buf[:256 * 64].reshape(256, 8, 8).astype(np.float64)
wl = buf[256 * 64:].reshape(256, 2)
wl = wl[wl[:, 0] > 0]
ent, loop, end, fin = [(v.astype(np.int64) & 0xFFFFFFFF).astype(np.float64) / 100.0 for v in (wl[:, 0], wl[:, 0] >> np.uint64(32), wl[:, 1], wl[:, 1] >> np.uint64(32))]
t0 = ent.min()
print("wall clock (us, wave 0 of each workgroup): entry %.1f..%.1f | prologue mean %.1f max %.1f | loop mean %.1f max %.1f | reduction+flush mean %.1f max %.1f | last exit %.1f" % (
    0.0, ent.max() - t0, (loop - ent).mean(), (loop - ent).max(), (end - loop).mean(), (end - loop).max(), (fin - end).mean(), (fin - end).max(), fin.max() - t0))
print("deg  waves tiles/wave | per tile: wait  phaseA  B1read  B1mfma  B2read  B2mfma | total/tile  per wave total (max)")
for d in range(11):
    m = (a[:, :, 7] == d) & (a[:, :, 6] > 0)
    if m.sum() == 0: continue
    r = a[m]
    nt = r[:, 6]
    tot = r[:, :6].sum(1)
    print("%3d %6d %8.1f | %9.0f %7.0f %7.0f %7.0f %7.0f %7.0f | %8.0f   %8.0f (%.0f)" % (
        d, m.sum(), nt.mean(), *(r[:, k].sum() / nt.sum() for k in range(6)), tot.sum() / nt.sum(), tot.mean(), tot.max()))
