import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import helpers as H
from jova_amd import synth
from jova_amd.data.collate import collate_pairs
from jova_amd.feat.mol_graphs import ConvMol
from jova_amd.train import TUNED_DAVIS_HPARAMS, Trainer, create_integrated_net
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
acts = {}
for i in (1, 4, 8):
    net[i].register_forward_hook(lambda m, a, o, i=i: acts.__setitem__(i, o.detach().cpu()))
with torch.no_grad(): net(batch.model_input())
torch.set_num_threads(16)
b = og.agglomerate([og.SortedMol(*pool.mols[i]) for i in mi]); gl = og.gconv_input_list(b)
glist = [torch.from_numpy(gl[0]), torch.from_numpy(gl[1]), torch.from_numpy(gl[2]), gl[3]] + [torch.from_numpy(a) for a in gl[4:]]
idx = og.embedding_indices([pool.prot_names[i] for i in pi], pool.protein_profile, fill_val=pool.vocab)
p = {k: (v.detach().clone().double() if v.is_floating_point() else v) for k, v in sd0.items()}
with torch.no_grad():
    v_g, _ = R.gconv_net(glist, p, prefix="0.models.1.0.")
    v_p = R.pcnn2d(R.prot2vec(torch.from_numpy(idx), p["0.models.2.0.embedding.weight"]), [p["0.models.2.1.lin_kernels.%d.weight" % i] for i in range(3)], [p["0.models.2.1.lin_kernels.%d.bias" % i] for i in range(3)])
    v_e = torch.from_numpy(pool.ecfp[mi]).double().unsqueeze(0)
    x = R.joint_attention([v_e, v_g, v_p], p, 2, prefix="1.")
    ref = {1: x}
    k = 2
    for _ in range(2):
        x = R.linear(x, p["%d.weight" % k], p["%d.bias" % k])
        pre = R.batch_norm_train(x, p["%d.weight" % (k + 1)], p["%d.bias" % (k + 1)])
        x = torch.relu(pre); ref[k + 2] = x; ref[('pre', k + 2)] = pre
        k += 4
for i in (1, 4, 8):
    g = acts[i].double(); r = ref[i]
    mism = ((g > 0) != (r > 0))
    print('stage', i, 'shape', tuple(r.shape), 'max abs err %.2e' % float((g - r).abs().max()), 'mask mismatches', int(mism.sum()))
    if i != 1 and mism.any():
        print('   pre-activation at mismatches', ref[('pre', i)][mism].tolist()[:8])
