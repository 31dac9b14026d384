"""dev tool: time the graph-conv backward on the config-4 batch (GPU box only).
JOVA_LIB=<instrumented build> adds the in-kernel timeline of the last launch."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from jova_amd import synth
from jova_amd.data.collate import process_gconv_view_data
from jova_amd.feat.mol_graphs import ConvMol
from jova_amd.nn.layers import GraphConvLayer
dev = torch.device('cuda:0')
pool = synth.SyntheticPool("graph4k", seed=1)
mols = [ConvMol(f, a, smiles=str(i)) for i, (f, a) in enumerate(pool.mols)]
glist = process_gconv_view_data(mols, dev)
n = glist[0].shape[0]
act = None if "nomask" in sys.argv else "relu"
conv1, conv2 = GraphConvLayer(75, 64, activation=act).to(dev), GraphConvLayer(64, 64, activation=act).to(dev)
for m in (conv1, conv2):
    for prm in m.parameters():
        prm.grad = torch.zeros_like(prm)
x2 = torch.randn(n, 64, device=dev).relu_().requires_grad_(True)
g64 = torch.randn(n, 64, device=dev)
if "f80" in sys.argv:        # experiment: layer 1 with 80 real columns (aligned dW rows)
    conv1 = GraphConvLayer(80, 64, activation=act).to(dev)
    for prm in conv1.parameters():
        prm.grad = torch.zeros_like(prm)
    x80 = torch.randn(n, 80, device=dev)
    y1 = conv1([x80] + glist[1:])
    y2 = conv2([x2] + glist[1:])
else:
    y1, y2 = conv1(glist), conv2([x2] + glist[1:])
which = [("l1", y1), ("l2", y2)]
if "l1last" in sys.argv:
    which = which[::-1]
for name, y in which:
    for r in range(6):
        x2.grad = None
        y.backward(g64, retain_graph=True)
    torch.cuda.synchronize()
if "fwd" in sys.argv:
    with torch.no_grad():
        for r in range(4):
            (conv1(glist) if "l1last" in sys.argv else conv2([x2] + glist[1:]))
    torch.cuda.synchronize()
if os.environ.get("JOVA_LIB"):
    import ctypes
    from jova_amd import _lib
    L = _lib.lib()
    buf = (ctypes.c_ulonglong * (4096 * 8))()
    L.jova_debug_bwd_prof(buf, 4096 * 8)
    a = np.array(buf[:], dtype=np.uint64).reshape(4096, 8)


    a = a[a[:, 6] > 0]
    t0 = a[:, 0].min()
    ent, ls, le, fl = [(a[:, k] - t0).astype(np.float64) / 100.0 for k in range(4)]
    steps = a[:, 6].astype(np.float64)
    print("last launch:", which[-1][0], "WGs", len(a), "steps", int(steps.sum()))
    print("us: kernel span %.1f; prologue mean %.2f max %.2f; loop mean %.2f max %.2f; flush mean %.2f max %.2f" % (
        fl.max(), (ls - ent).mean(), (ls - ent).max(), (le - ls).mean(), (le - ls).max(), (fl - le).mean(), (fl - le).max()))
    print("cycles per step: wait %.0f  compute %.0f;  loop us per step %.2f" % (
        a[:, 4].sum() / steps.sum(), a[:, 5].sum() / steps.sum(), (le - ls).sum() / steps.sum()))
    hw = a[:, 7]
    key = (((hw >> 32) & 0xF) * 256 + ((hw >> 8) & 0xFF)).astype(np.int64)
    uniq, cnt = np.unique(key, return_counts=True)
    busy = np.array([(fl - ent)[key == k].sum() for k in uniq])
    print("distinct CUs %d, WGs per CU min %d max %d; per-CU summed WG life us: min %.1f med %.1f max %.1f" % (
        len(uniq), cnt.min(), cnt.max(), busy.min(), np.median(busy), busy.max()))
    print("entry pct", np.percentile(ent, [0, 25, 50, 75, 90, 100]).round(1), "end pct", np.percentile(fl, [0, 25, 50, 75, 90, 100]).round(1))
