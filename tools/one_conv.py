"""Runs graph-conv forward (layer-2 shape, or layer 1 with 'l1') a few times on the config-4 batch: profiling target."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from jova_amd import synth
from jova_amd.feat.mol_graphs import ConvMol
from jova_amd.data.collate import process_gconv_view_data
from jova_amd.nn.layers import GraphConvLayer
dev = torch.device('cuda:0')
pool = synth.SyntheticPool("graph4k", seed=1)
mols = [ConvMol(f, a, smiles=str(i)) for i, (f, a) in enumerate(pool.mols)]
glist = process_gconv_view_data(mols, dev)
n = glist[0].shape[0]
l1 = 'l1' in sys.argv
conv = GraphConvLayer(75 if l1 else 64, 64).to(dev)
gl = glist if l1 else [torch.randn(n, 64, device=dev).relu_()] + glist[1:]
mode = 'bwd' if 'bwd' in sys.argv else 'fwd'
if mode == 'fwd':
    with torch.no_grad():
        for _ in range(6): conv(gl)
else:
    for prm in conv.parameters(): prm.grad = torch.zeros_like(prm)
    if not l1: gl[0].requires_grad_(True)
    g = torch.randn(n, 64, device=dev)
    for _ in range(6):
        conv(gl).backward(g)
torch.cuda.synchronize()
