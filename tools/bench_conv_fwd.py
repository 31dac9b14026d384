"""conv forward only, config-4 batch (dev tool)."""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import ctypes, torch
from jova_amd import synth
from jova_amd.feat.mol_graphs import ConvMol
from jova_amd.data.collate import process_gconv_view_data
from jova_amd.nn import functional as F_
from jova_amd.nn.layers import GraphConvLayer
from jova_amd._lib import P, call, lib, ptr_array, stream
dev = torch.device('cuda:0')
pool = synth.SyntheticPool("graph4k", seed=1)
mols = [ConvMol(f, a, smiles=str(i)) for i, (f, a) in enumerate(pool.mols)]
glist = process_gconv_view_data(mols, dev)
topo = F_.topology_of(glist)
n = glist[0].shape[0]
conv1, conv2 = GraphConvLayer(75, 64).to(dev), GraphConvLayer(64, 64).to(dev)
x1 = topo.atoms_f32(glist[0]); x2 = torch.randn(n, 64, device=dev).relu_()
out = torch.empty(n, 64, device=dev)
st = stream(); ds, adj = P(topo.deg_slice), topo.adj_ptrs
dsh = ctypes.c_void_p(topo.deg_slice_host.ctypes.data)
stats = torch.zeros(2, 64, dtype=torch.float64, device=dev)
def run(layer, x, ld, fin, ws, with_stats):
    w = [l.weight for l in layer.linear_layers]; b = [l.bias for l in layer.linear_layers]
    pw, pb = ptr_array(w), ptr_array(b)
    f = lambda: call("jova_graphconv_fwd", P(x), 0, ld, n, fin, 64, ds, adj, pw, pb, 1, P(out), P(stats) if with_stats else None, P(ws), dsh, st)
    f(); torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): f()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / 20 * 1e3
ws1 = torch.empty(int(lib().jova_graphconv_workspace_floats(75, 64)), device=dev)
ws2 = torch.empty(int(lib().jova_graphconv_workspace_floats(64, 64)), device=dev)
print("dbg=%s  l1 %.1f us  l2 %.1f us  l2+stats %.1f us" % (os.environ.get("JOVA_WAVE_DBG", "0"), run(conv1, x1, x1.shape[1], 75, ws1, False),
      run(conv2, x2, 64, 64, ws2, False), run(conv2, x2, 64, 64, ws2, True)))
