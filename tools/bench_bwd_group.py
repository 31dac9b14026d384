"""dev tool: conv backward on the config-4 batch for several JOVA_BWD_GROUP values (run under rocprofv3; GPU box only)"""
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
conv2 = GraphConvLayer(64, 64).to(dev)
for prm in conv2.parameters():
    prm.grad = torch.zeros_like(prm)
x2 = torch.randn(n, 64, device=dev).relu_().requires_grad_(True)
g64 = torch.randn(n, 64, device=dev)
y2 = conv2([x2] + glist[1:])
for grp in sys.argv[1:]:
    os.environ["JOVA_BWD_GROUP"] = grp
    for r in range(5):
        x2.grad = None
        y2.backward(g64, retain_graph=True)
    torch.cuda.synchronize()
