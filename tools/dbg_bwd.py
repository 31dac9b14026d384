import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import helpers as H
from jova_amd import synth
from jova_amd.data.collate import collate_pairs
from jova_amd.feat.mol_graphs import ConvMol
from jova_amd.train import TUNED_DAVIS_HPARAMS, Trainer, create_integrated_net
from jova_amd.nn import functional as F_
from oracle import mol_graphs as og
from oracle import nn_ref as R
device = torch.device('cuda:0')
hp = dict(TUNED_DAVIS_HPARAMS); hp["dprob"] = 0.0
pool = synth.SyntheticPool("davis", seed=8)
cms = [ConvMol(f, a, smiles=str(i)) for i, (f, a) in enumerate(pool.mols)]
torch.manual_seed(8)
net = create_integrated_net(hp, pool.protein_profile)
sd0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
net = net.to(device).train()
mi, pi, y, w = pool.sample_pairs(np.random.RandomState(64), 128)
torch.set_num_threads(16)
b = og.agglomerate([og.SortedMol(*pool.mols[i]) for i in mi]); gl = og.gconv_input_list(b)
glist = [torch.from_numpy(gl[0]), torch.from_numpy(gl[1]), torch.from_numpy(gl[2]), gl[3]] + [torch.from_numpy(a) for a in gl[4:]]
idx = og.embedding_indices([pool.prot_names[i] for i in pi], pool.protein_profile, fill_val=pool.vocab)
named = dict(net.named_parameters())
def mk(dt):
    return {k: (v.detach().clone().to(dt).requires_grad_("running" not in k) if v.is_floating_point() else v) for k, v in sd0.items()}
def head(x, p):
    k = 2
    for _ in range(2):
        x = R.linear(x, p["%d.weight" % k], p["%d.bias" % k])
        x = torch.relu(R.batch_norm_train(x, p["%d.weight" % (k + 1)], p["%d.bias" % (k + 1)]))
        k += 4
    return R.linear(x, p["%d.weight" % k], p["%d.bias" % k])
p64 = mk(torch.float64)
with torch.no_grad():
    v_g, _ = R.gconv_net(glist, p64, prefix="0.models.1.0.")
    v_p = R.pcnn2d(R.prot2vec(torch.from_numpy(idx), p64["0.models.2.0.embedding.weight"]), [p64["0.models.2.1.lin_kernels.%d.weight" % i] for i in range(3)], [p64["0.models.2.1.lin_kernels.%d.bias" % i] for i in range(3)])
    v_e = torch.from_numpy(pool.ecfp[mi]).double().unsqueeze(0)
views64 = [v.clone().requires_grad_(True) for v in (v_e, v_g, v_p)]
ja64 = R.joint_attention(views64, p64, 2, prefix="1.")
ja_leaf = ja64.detach().clone().requires_grad_(True)
o64 = head(ja_leaf, p64); l64 = R.mse_loss(o64, torch.from_numpy(w).double(), torch.from_numpy(y).double()); l64.backward()
dja64 = ja_leaf.grad.clone()
# cpu32 head
p32 = mk(torch.float32)
ja_leaf32 = ja64.detach().float().requires_grad_(True)
o32 = head(ja_leaf32, p32); l32 = R.mse_loss(o32, torch.from_numpy(w).float(), torch.from_numpy(y).float()); l32.backward()
# gpu head
jg = ja64.detach().float().to(device).requires_grad_(True)
x = jg
for i in range(2, len(net)): x = net[i](x)
lg = F_.MSELossFunction.apply(x, torch.from_numpy(w).float().to(device), torch.from_numpy(y).float().to(device)); lg.backward()
print('HEAD: out err gpu %.2e cpu32 %.2e' % (H.rel_err(x.detach().cpu().numpy(), o64.detach().numpy()), H.rel_err(o32.detach().numpy(), o64.detach().numpy())))
print('HEAD: dja err gpu %.2e cpu32 %.2e' % (H.rel_err(jg.grad.cpu().numpy(), dja64.numpy()), H.rel_err(ja_leaf32.grad.numpy(), dja64.numpy())))
for k in [k for k in sd0 if k[0] in '23456789' or k.startswith('10.')]:
    if p64[k].is_floating_point() and p64[k].grad is not None:
        g = p64[k].grad.numpy(); fl = 0.05 * np.abs(g).max()
        print('  %-12s gpu %.2e cpu32 %.2e' % (k, H.rel_err(named[k].grad.cpu().numpy(), g, fl), H.rel_err(p32[k].grad.numpy(), g, fl)))
# JA backward with oracle dja
for p_ in named.values():
    if p_.grad is not None: p_.grad.zero_()
ja64.backward(dja64)
vg = [v.detach().float().to(device).requires_grad_(True) for v in (v_e, v_g, v_p)]
jag = net[1](vg); jag.backward(dja64.float().to(device))
p32b = mk(torch.float32)
v32 = [v.detach().float().requires_grad_(True) for v in (v_e, v_g, v_p)]
R.joint_attention(v32, p32b, 2, prefix='1.').backward(dja64.float())
print('JA: fwd err gpu %.2e' % H.rel_err(jag.detach().cpu().numpy(), ja64.detach().numpy()))
for i in range(3):
    g = views64[i].grad.numpy()
    print('  dview%d gpu %.2e cpu32 %.2e' % (i, H.rel_err(vg[i].grad.cpu().numpy(), g), H.rel_err(v32[i].grad.numpy(), g)))
for k in [k for k in sd0 if k.startswith('1.')]:
    if p64[k].is_floating_point() and p64[k].grad is not None:
        g = p64[k].grad.numpy(); fl = 0.05 * np.abs(g).max() + 1e-30
        print('  %-40s gpu %.2e cpu32 %.2e  |g|max %.2e' % (k, H.rel_err(named[k].grad.cpu().numpy(), g, fl), H.rel_err(p32b[k].grad.numpy(), g, fl), np.abs(g).max()))
