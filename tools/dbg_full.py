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
batch = collate_pairs(pool, cms, mi, pi, y, w, device, net[0].models[2][0])
tr = Trainer(net, hp); tr.flat.zero_grad()
loss, out = tr.forward_loss(batch); loss.backward()
print('gpu nonsat ctrl', F_.NonsatFunction.last_n_eval.cpu().tolist())
torch.set_num_threads(16)
b = og.agglomerate([og.SortedMol(*pool.mols[i]) for i in mi]); gl = og.gconv_input_list(b)
glist = [torch.from_numpy(gl[0]), torch.from_numpy(gl[1]), torch.from_numpy(gl[2]), gl[3]] + [torch.from_numpy(a) for a in gl[4:]]
idx = og.embedding_indices([pool.prot_names[i] for i in pi], pool.protein_profile, fill_val=pool.vocab)
named = dict(net.named_parameters())
# stage-wise forward comparison (fp64 oracle)
p64 = {k: (v.detach().clone().double() if v.is_floating_point() else v) for k, v in sd0.items()}
with torch.no_grad():
    views_gpu = net[0](batch.model_input())
    ja_gpu = net[1](views_gpu)
    v_g, _ = R.gconv_net(glist, p64, prefix="0.models.1.0.")
    v_p = R.pcnn2d(R.prot2vec(torch.from_numpy(idx), p64["0.models.2.0.embedding.weight"]), [p64["0.models.2.1.lin_kernels.%d.weight" % i] for i in range(3)], [p64["0.models.2.1.lin_kernels.%d.bias" % i] for i in range(3)])
    v_e = torch.from_numpy(pool.ecfp[mi]).double().unsqueeze(0)
    print('gconv view err', H.rel_err(views_gpu[1].cpu().numpy(), v_g.numpy()))
    print('prot view err', H.rel_err(views_gpu[2].cpu().numpy(), v_p.numpy()))
    ja = R.joint_attention([v_e, v_g, v_p], p64, 2, prefix="1.")
    print('JA out err', H.rel_err(ja_gpu.cpu().numpy(), ja.numpy()))
    p32 = {k: (v.detach().clone() if v.is_floating_point() else v) for k, v in sd0.items()}
    print('JA cpu32-vs-64 err', H.rel_err(R.joint_attention([v_e.float(), v_g.float(), v_p.float()], p32, 2, prefix='1.').numpy(), ja.numpy()))
    # JA on oracle inputs but GPU kernels
    ja2 = net[1]([v_e.float().to(device), v_g.float().to(device), v_p.float().to(device)])
    print('JA(gpu kernels on oracle views) err', H.rel_err(ja2.cpu().numpy(), ja.numpy()))
res = {}
for dt in (torch.float32, torch.float64):
    p = {k: (v.detach().clone().to(dt).requires_grad_("running" not in k) if v.is_floating_point() else v) for k, v in sd0.items()}
    want, nev = R.dti_forward(p, dict(ecfp=torch.from_numpy(pool.ecfp[mi]).to(dt), glist=glist, prot_idx=torch.from_numpy(idx)), 2, 3, 2)
    wl = R.mse_loss(want, torch.from_numpy(w).to(dt), torch.from_numpy(y).to(dt)); wl.backward()
    print(dt, 'nonsat evals', nev, 'loss', float(wl), 'gpu loss', loss.item(), 'out err', H.rel_err(out.detach().cpu().numpy(), want.detach().numpy()))
    res[dt] = p
gmax = max(float(v.grad.abs().max()) for v in res[torch.float64].values() if v.requires_grad and v.grad is not None)
worst = []
for k, v in res[torch.float64].items():
    if v.requires_grad and v.grad is not None:
        g64 = v.grad.numpy(); g32 = res[torch.float32][k].grad.numpy(); gg = named[k].grad.cpu().numpy()
        worst.append((H.rel_err(gg, g64, 0.05 * gmax), H.rel_err(g32, g64, 0.05 * gmax), H.rel_err(gg, g32, 0.05*gmax), k))
worst.sort(reverse=True)
for e in worst[:12]: print("gpu-vs-64 %.2e  cpu32-vs-64 %.2e  gpu-vs-cpu32 %.2e  %s" % e)
