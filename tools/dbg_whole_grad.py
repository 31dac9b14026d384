"""Whole-tensor gradient distances of the HIP step and of the fp32 oracle from the fp64 oracle, per parameter tensor, next
to the bound the full-size parity test derives from the reference's own runs.  usage: python tools/dbg_whole_grad.py <tag>"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H                                            # noqa: E402
import test_full_size_parity as T                              # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "davis_b128"
device = torch.device("cuda:0")
from jova_amd.data.collate import collate_pairs               # noqa: E402
from jova_amd.feat.mol_graphs import ConvMol                  # noqa: E402
from jova_amd.train import Trainer                            # noqa: E402

z, spec, hparams, pool, mi, pi, y, w = H.full_setup(tag)
net = H.full_net(spec, hparams, pool).to(device).train()
cms = [ConvMol(f, a, smiles=str(i)) for i, (f, a) in enumerate(pool.mols)]
batch = collate_pairs(pool, cms, mi, pi, y, w, device, net[0].models[2][0], with_psc="psc" in spec["views"])
tr = Trainer(net, hparams)
tr.flat.zero_grad()
loss, out = tr.forward_loss(batch)
loss.backward()
g_hip = {k: p.grad.detach().cpu().double().numpy() for k, p in net.named_parameters()}
torch.set_num_threads(16)
_, _, _, _, g64 = T._oracle_run(tag)
_, _, _, _, g32 = T._oracle_run(tag, torch.float32)
stats = [z[k] for k in z.files if k.startswith("gstat.")]
e_all = float(np.sqrt(sum(float(s[3]) ** 2 for s in stats) / sum(float(s[0]) ** 2 for s in stats)))
gmax = max(float(s[1]) for s in stats)
print("e_all %.3e" % e_all)
for k in z.files:
    if not k.startswith("gstat."):
        continue
    name = k[6:]
    l2, mx, dmax, dl2 = (float(v) for v in z[k])
    if mx < 1e-9 * gmax:
        continue
    bound = max(1e-5, 2 * max(dmax / mx, e_all))
    dh = np.abs(g_hip[name] - g64[name]).reshape(-1)
    d32 = np.abs(g32[name] - g64[name]).reshape(-1)
    flag = "  <-- OVER" if dh.max() / mx > bound else ""
    print("%-50s n=%8d  hip max %.2e l2 %.2e | oracle32 max %.2e l2 %.2e | ref32 max %.2e l2 %.2e | bound %.2e  over(hip) %d over(o32) %d%s"
          % (name, dh.size, dh.max() / mx, np.sqrt((dh ** 2).sum()) / l2, d32.max() / mx, np.sqrt((d32 ** 2).sum()) / l2,
             dmax / mx, dl2 / l2, bound, int((dh / mx > bound).sum()), int((d32 / mx > bound).sum()), flag))
    if flag:
        idx = np.argsort(-dh)[:6]
        print("      worst elements", [(int(i), "%.2e" % (dh[i] / mx), "o32 %.2e" % (d32[i] / mx)) for i in idx])
