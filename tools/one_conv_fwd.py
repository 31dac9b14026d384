"""Four launches of each conv forward on the config-4 batch (for rocprofv3 --pmc passes)."""
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
for layer, x, ld, fin in ((conv1, x1, x1.shape[1], 75), (conv2, x2, 64, 64)):
    w = [l.weight for l in layer.linear_layers]; b = [l.bias for l in layer.linear_layers]
    ws = torch.empty(int(lib().jova_graphconv_workspace_floats(fin, 64)), device=dev)
    for _ in range(4):
        call("jova_graphconv_fwd", P(x), 0, ld, n, fin, 64, ds, adj, ptr_array(w), ptr_array(b), 1, P(out), None, P(ws), dsh, st)
    torch.cuda.synchronize()
