#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE modules on CPU.

Build-container only: imports /root/reference/jova/{nn,feat} with the recipe of
SURVEY.md Appendix A (packages registered without running their __init__, keras
stubbed) and dumps inputs + weights + outputs + gradients of every hot-path
module as small compressed fixtures.  Nothing of the reference travels: the
fixtures are pure data.  Re-run with ``python tools/make_golden.py`` (deterministic).
"""
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


_pkg("jova", REF + "/jova").cuda = False
for sub in ("utils", "nn", "feat"):
    _pkg("jova." + sub, "%s/jova/%s" % (REF, sub))
for n in ("keras", "keras.utils", "keras.utils.data_utils"):
    sys.modules[n] = types.ModuleType(n)
sys.modules["keras.utils.data_utils"].get_file = lambda *a, **k: None
import jova.nn.layers as L          # noqa: E402
import jova.nn.models as M          # noqa: E402
import jova.feat.mol_graphs as G    # noqa: E402
from torch import nn                # noqa: E402

from jova_amd import synth          # noqa: E402

torch.set_num_threads(4)


# ------------------------------------------------------------------ helpers
def pack_mols(mols):
    """list of (features, adjacency) -> flat arrays that np.savez can hold."""
    feats = np.vstack([f for f, _ in mols]).astype(np.int16)
    atom_off = np.cumsum([0] + [f.shape[0] for f, _ in mols]).astype(np.int64)
    nbr_flat, nbr_off = [], [0]
    for _, adj in mols:
        for nb in adj:
            nbr_flat.extend(nb)
            nbr_off.append(len(nbr_flat))
    return dict(mol_feats=feats, mol_atom_off=atom_off,
                mol_nbr_flat=np.array(nbr_flat, dtype=np.int32), mol_nbr_off=np.array(nbr_off, dtype=np.int64))


def ref_batch(mols):
    cms = [G.ConvMol(f, a, smiles="m%d" % i) for i, (f, a) in enumerate(mols)]
    mm = G.ConvMol.agglomerate_mols(cms)
    glist = [torch.from_numpy(mm.get_atom_features()), torch.from_numpy(mm.deg_slice),
             torch.tensor(mm.membership), [m.get_num_atoms() for m in cms]]
    glist += [torch.from_numpy(a) for a in mm.get_deg_adjacency_lists()[1:]]     # data.py:291-296
    return mm, glist


def index_arrays(mm, glist):
    d = dict(nodes=mm.get_atom_features().astype(np.int16), deg_slice=np.asarray(mm.deg_slice),
             membership=np.asarray(mm.membership, dtype=np.int64), n_atoms=np.asarray(glist[3], dtype=np.int64))
    d["nodes_dtype"] = np.array(str(mm.get_atom_features().dtype))
    d["deg_slice_dtype"] = np.array(str(np.asarray(mm.deg_slice).dtype))
    for k, a in enumerate(mm.get_deg_adjacency_lists()):
        d["adj_%d" % k] = a
    return d


def sd_arrays(module, prefix="sd."):
    return {prefix + k: v.detach().numpy().copy() for k, v in module.state_dict().items()}


def grad_arrays(module, prefix="grad."):
    return {prefix + k: (p.grad.detach().numpy().copy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32))
            for k, p in module.named_parameters()}


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-28s %7.1f KB" % (name, os.path.getsize(path) / 1024.))


def mol_sets():
    sets = {}
    r = np.random.RandomState(1)
    sets["toy4"] = [synth.make_molecule(r, n) for n in (7, 9, 8, 9)]
    r = np.random.RandomState(8)
    # single atoms (degree 0), a 2-atom molecule, and capped graphs -> empty buckets 5..10
    sets["deg0_b16"] = [synth.make_molecule(r, n) for n in (1, 12, 1, 2, 19, 23, 5, 3, 1, 30, 46, 21, 2, 9, 14, 1)]
    r = np.random.RandomState(64)
    sets["hubs_b8"] = [synth.make_molecule(r, n, deg_cap=4, hub_frac=0.15) for n in (40, 61, 25, 90, 33, 100, 20, 77)]
    r = np.random.RandomState(1)
    sets["pair_b2"] = [synth.make_molecule(r, n) for n in (19, 46)]
    r = np.random.RandomState(8)
    sets["davis_b128"] = [synth.make_molecule(r, n) for n in synth.SHAPES["davis"]["atoms"](r, 128)]
    return sets


# ------------------------------------------------------------------ G1: index goldens
def gen_index(sets):
    for name, mols in sets.items():
        mm, glist = ref_batch(mols)
        save("index_" + name, **pack_mols(mols), **index_arrays(mm, glist))


# ------------------------------------------------------------------ G2: per-op goldens
def gen_graphconv(sets):
    mols = sets["hubs_b8"] + sets["deg0_b16"][:6]
    mm, glist = ref_batch(mols)
    arrays = dict(pack_mols(mols))
    n = glist[0].shape[0]
    for tag, in_dim, act in (("l1", 75, "relu"), ("l2", 64, "relu"), ("lin", 64, None)):
        torch.manual_seed({"l1": 1, "l2": 8, "lin": 64}[tag])
        layer = L.GraphConvLayer(in_dim, 64, activation=act)
        inp = list(glist)
        if in_dim != 75:
            x = torch.randn(n, in_dim).relu_().requires_grad_(True)
            inp[0] = x
        out = layer(inp)
        gout = torch.randn(out.shape)
        (out * gout).sum().backward()
        arrays.update({tag + ".out": out.detach().numpy(), tag + ".gout": gout.numpy()})
        arrays.update(sd_arrays(layer, tag + ".sd."))
        arrays.update(grad_arrays(layer, tag + ".grad."))
        if in_dim != 75:
            arrays[tag + ".x"] = x.detach().numpy()
            arrays[tag + ".gx"] = x.grad.numpy()
    save("op_graphconv", **arrays)


def gen_graphpool(sets):
    mols = sets["hubs_b8"] + sets["deg0_b16"][:6]
    mm, glist = ref_batch(mols)
    n = glist[0].shape[0]
    torch.manual_seed(8)
    # post-ReLU style input with many exact ties (zeros and a coarse value grid)
    x = (torch.randn(n, 64) * 2).round().div(2).relu_().requires_grad_(True)
    inp = list(glist)
    inp[0] = x
    out = L.GraphPool()(inp)
    gout = torch.randn(out.shape)
    (out * gout).sum().backward()
    save("op_graphpool", **pack_mols(mols), x=x.detach().numpy(), out=out.detach().numpy(),
         gout=gout.numpy(), gx=x.grad.numpy())


def gen_gather_nonsat(sets):
    arrays = {}
    for tag, setname, g, sigma in (("a", "deg0_b16", 24, 1.0), ("b", "hubs_b8", 16, 4.0), ("c", "toy4", 8, 0.5)):
        mols = sets[setname]
        n_atoms = [f.shape[0] for f, _ in mols]
        torch.manual_seed(len(n_atoms))
        x = (torch.randn(sum(n_atoms), g) * sigma).requires_grad_(True)
        glist = [x, None, None, n_atoms]
        out = L.GraphGather2D(activation="nonsat")(glist, len(n_atoms))
        gout = torch.randn(out.shape)
        (out * gout).sum().backward()
        arrays.update({tag + ".x": x.detach().numpy(), tag + ".n_atoms": np.array(n_atoms),
                       tag + ".out": out.detach().numpy(), tag + ".gout": gout.numpy(), tag + ".gx": x.grad.numpy()})
        # tanh flavour (ctor default) pins the padding/stacking alone
        out_t = L.GraphGather2D()([x.detach(), None, None, n_atoms], len(n_atoms))
        arrays[tag + ".out_tanh"] = out_t.numpy()
    save("op_gather2d_nonsat", **arrays)


def make_profile(seed, n_prot, lo, hi, vocab):
    r = np.random.RandomState(seed)
    names = ["P%03d" % i for i in range(n_prot)]
    return names, {nm: synth.make_protein(r, r.randint(lo, hi + 1), vocab) for nm in names}


def gen_protein():
    vocab, e, w = 60, 6, 11
    names, profile = make_profile(1, 5, 3, 9, vocab)
    batch = [("davis", names[i]) for i in (0, 3, 3, 1, 4, 2, 0)]
    torch.manual_seed(1)
    p2v = M.Prot2Vec(protein_profile=profile, vocab_size=vocab, embedding_dim=e, batch_first=False)
    idx = M._construct_embedding_indices(batch, profile, torch.device("cpu"), vocab)
    out = p2v(batch)
    gout = torch.randn(out.shape)
    (out * gout).sum().backward()
    arrays = dict(idx=idx.numpy(), out=out.detach().numpy(), gout=gout.numpy(),
                  batch_prot=np.array([int(nm[1][1:]) for nm in batch]),
                  **sd_arrays(p2v), **grad_arrays(p2v))
    for nm in names:
        arrays["profile." + nm] = profile[nm]
    save("op_prot2vec", **arrays)

    torch.manual_seed(8)
    pcnn = M.ProteinCNN2D(dim=e, window=w, num_layers=3)
    x = torch.randn(9, 7, e * w).relu_().requires_grad_(True)
    out = pcnn(x)
    gout = torch.randn(out.shape)
    (out * gout).sum().backward()
    save("op_pcnn2d", x=x.detach().numpy(), out=out.detach().numpy(), gout=gout.numpy(), gx=x.grad.numpy(),
         **sd_arrays(pcnn), **grad_arrays(pcnn))


def gen_joint_attention():
    torch.manual_seed(64)
    d_dims, latent, heads, bsz = [40, 12, 6], 32, 2, 5
    ja = M.JointAttention(d_dims=d_dims, latent_dim=latent, num_heads=heads, num_layers=1, dprob=0.,
                          inner_layer_dim=48)
    segs = [1, 9, 13]
    xs = [torch.randn(s, bsz, d).requires_grad_(True) for s, d in zip(segs, d_dims)]
    out = ja(xs)
    gout = torch.randn(out.shape)
    (out * gout).sum().backward()
    arrays = dict(out=out.detach().numpy(), gout=gout.numpy(), heads=np.array(heads))
    for i, x in enumerate(xs):
        arrays["x%d" % i] = x.detach().numpy()
        arrays["gx%d" % i] = x.grad.numpy()
    save("op_joint_attention", **arrays, **sd_arrays(ja), **grad_arrays(ja))


# ------------------------------------------------------------------ G3: end to end
def build_reference_net(hp, profile):
    """Restates create_integrated_net (train_joint_attn.py:188-243) for ecfp8-gconv__pcnn2d with reference classes."""
    dim, dp = hp["gconv_dim"], hp["dprob"]
    gconv = nn.Sequential(M.GraphConvSequential(
        L.GraphConvLayer(in_dim=75, out_dim=64), nn.BatchNorm1d(64), nn.ReLU(), L.GraphPool(),
        L.GraphConvLayer(in_dim=64, out_dim=64), nn.BatchNorm1d(64), nn.ReLU(), L.GraphPool(),
        nn.Linear(64, dim), nn.BatchNorm1d(dim), nn.ReLU(), nn.Dropout(dp),
        L.GraphGather2D(activation="nonsat")))
    ecfp = nn.Sequential(L.Unsqueeze(dim=0))
    prot = nn.Sequential(M.Prot2Vec(protein_profile=profile, vocab_size=hp["vocab"], embedding_dim=hp["emb"],
                                    batch_first=False),
                         M.ProteinCNN2D(dim=hp["emb"], window=hp["window"], num_layers=hp["pcnn_layers"]))
    layers = [M.NwayForward(models=[ecfp, gconv, prot]),
              M.JointAttention(d_dims=[hp["ecfp_dim"], dim, hp["emb"]], latent_dim=hp["latent"],
                               num_heads=hp["heads"], num_layers=hp["attn_layers"], dprob=dp)]
    p = 3 * hp["latent"]
    for d in hp["lin_dims"]:
        layers += [nn.Linear(p, d), nn.BatchNorm1d(d), nn.ReLU(), nn.Dropout(dp)]
        p = d
    layers.append(nn.Linear(p, 1))
    return nn.Sequential(*layers)


def gen_e2e(sets):
    hp = dict(gconv_dim=20, dprob=0., vocab=60, emb=4, window=11, pcnn_layers=3, ecfp_dim=1024, latent=32, heads=2,
              attn_layers=1, lin_dims=[24, 20], lr=4.70e-3, wd=1.91e-2)
    mols = sets["deg0_b16"][3:11]
    names, profile = make_profile(8, 6, 3, 10, hp["vocab"])
    r = np.random.RandomState(64)
    prot_ids = r.randint(0, len(names), size=len(mols))
    batch_prot = [("davis", names[i]) for i in prot_ids]
    ecfp = (r.uniform(size=(len(mols), 1024)) < 0.06).astype(np.float32)
    y = r.normal(size=(len(mols), 1))
    w = np.ones((len(mols), 1))
    torch.manual_seed(1)
    net = build_reference_net(hp, profile)
    net.train()
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    mm, glist = ref_batch(mols)
    opt = torch.optim.SGD(net.parameters(), lr=hp["lr"], weight_decay=hp["wd"])
    arrays = dict(pack_mols(mols), prot_ids=prot_ids, ecfp=ecfp, y=y, w=w, hp=np.array(json.dumps(hp)))
    for nm in names:
        arrays["profile." + nm] = profile[nm]
    arrays.update({"sd0." + k: v.numpy() for k, v in sd0.items()})
    for step in range(2):
        opt.zero_grad()
        out = net([torch.from_numpy(ecfp), (glist, len(mols)), batch_prot])
        loss = nn.MSELoss()(out * torch.from_numpy(w).float(), torch.from_numpy(y).float())   # :459-465
        loss.backward()
        arrays["step%d.out" % step] = out.detach().numpy()
        arrays["step%d.loss" % step] = np.array(loss.item())
        if step == 0:
            arrays.update(grad_arrays(net, "step0.grad."))
        opt.step()
        if step == 0:
            arrays.update(sd_arrays(net, "step0.sd."))
    save("e2e_small", **arrays)


def gen_manifest():
    """State-dict keys/shapes at the tuned Davis config (soek_res ... pcnn2d_ecfp8_gconv ... gp.csv:1-2)."""
    hp = dict(gconv_dim=194, dprob=0.211, vocab=9048, emb=22, window=11, pcnn_layers=3, ecfp_dim=1024, latent=512,
              heads=2, attn_layers=1, lin_dims=[267, 637])
    net = build_reference_net(hp, {})
    man = {k: list(v.shape) for k, v in net.state_dict().items()}
    n_train = sum(p.numel() for p in net.parameters() if p.requires_grad)
    with open(os.path.join(OUT, "manifest_tuned.json"), "w") as f:
        json.dump(dict(hparams=hp, n_trainable=n_train, state_dict=man), f, indent=0, sort_keys=True)
    print("manifest_tuned.json  trainable params:", n_train)


def gen_segment_kats():
    """The reference's only known-answer tests on this path: jova/utils/tests/test_math.py:17-55 (values copied as data)."""
    save("kat_segment",
         seg_sum_data=np.array([5, 1, 7, 2, 3, 4, 1, 3]), seg_ids=np.array([0, 0, 0, 1, 2, 2, 3, 3]),
         seg_sum_expect=np.array([13, 2, 7, 4]), seg_max_expect=np.array([7, 2, 4, 3]))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    sets = mol_sets()
    gen_index(sets)
    gen_graphconv(sets)
    gen_graphpool(sets)
    gen_gather_nonsat(sets)
    gen_protein()
    gen_joint_attention()
    gen_e2e(sets)
    gen_manifest()
    gen_segment_kats()
