import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import helpers as H
import torch.nn as nn
from jova_amd.nn.layers import Linear, BatchNorm1d, ReLU, Dropout
from jova_amd.nn import functional as F_
dev = torch.device('cuda:0')
torch.manual_seed(0)
dims = [1536, 267, 637]
layers = []
p = dims[0]
for d in dims[1:]:
    layers += [Linear(p, d), BatchNorm1d(d), ReLU(), Dropout(0.0)]; p = d
layers.append(Linear(p, 1))
net = nn.Sequential(*layers)
ref = nn.Sequential(*[nn.Linear(l.in_features, l.out_features) if isinstance(l, nn.Linear) else (nn.BatchNorm1d(l.num_features) if isinstance(l, nn.BatchNorm1d) else (nn.ReLU() if isinstance(l, nn.ReLU) else nn.Dropout(0.0))) for l in layers]).double()
ref.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in net.state_dict().items()})
x = torch.randn(128, 1536).relu()
y = torch.randn(128, 1)
net = net.to(dev).train(); ref.train()
xg = x.to(dev).requires_grad_(True); xr = x.double().requires_grad_(True)
out = net(xg); loss = F_.MSELossFunction.apply(out, torch.ones(128,1,device=dev), y.to(dev)); loss.backward()
o2 = ref(xr); l2 = ((o2 - y.double())**2).mean(); l2.backward()
print('out err', H.rel_err(out.detach().cpu().numpy(), o2.detach().numpy()), 'loss', loss.item(), float(l2))
print('dx err', H.rel_err(xg.grad.cpu().numpy(), xr.grad.numpy()))
for (k, a), (_, b) in zip(net.named_parameters(), ref.named_parameters()):
    print(k, tuple(a.shape), 'grad err %.2e' % H.rel_err(a.grad.cpu().numpy(), b.grad.numpy()))
