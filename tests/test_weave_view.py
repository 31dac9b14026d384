"""§8(f3): the weave view (jova/nn/layers.py:72-262, jova/nn/models.py:245-307) -- collator layout, oracle and HIP path
against the reference-made golden tests/golden/op_weave.npz (tools/make_golden_weave.py: the reference's classes with
torch_scatter's scatter_add bound to its published semantics, the one dependency the reference leaves unresolved)."""
import numpy as np
import pytest
import torch

import helpers as H

TAGS = {"up2d": (True, "2D"), "noup2d": (False, "2D"), "up1d": (True, "1D")}


class _Prot:
    def __init__(self, name):
        self._n = name

    def get_name(self):
        return self._n


def _mols(z):
    from jova_amd.feat.mol_graphs import WeaveMol
    return [WeaveMol(z["mol%d.nodes" % k], z["mol%d.pairs" % k], smiles="m%d" % k) for k in range(len(z["n_atoms"]))]


def test_weave_collator_equals_the_reference_layout():
    from jova_amd.data import data as D
    z = H.load("op_weave")
    mols = _mols(z)
    X = np.empty((len(mols), 3, 1), dtype=object)
    for k, m in enumerate(mols):
        X[k, 0, 0] = (m, _Prot("P0"))
    old = D._device
    D._device = lambda: torch.device("cpu")
    try:
        mol_data, pt, names, _ = D.process_weave_view_data(X, {"P0": np.zeros((1, 4), np.float32)}, 0, cuda_prot=False)
    finally:
        D._device = old
    af, pf, ps, a2p, asp, n_list = mol_data
    assert af.dtype == torch.float32 and np.array_equal(af.numpy(), z["atom_feat"])
    assert pf.dtype == torch.float32 and np.array_equal(pf.numpy(), z["pair_feat"])
    assert ps.dtype == torch.int32 and np.array_equal(ps.numpy(), z["pair_split"])
    assert a2p.dtype == torch.int64 and np.array_equal(a2p.numpy(), z["atom_to_pair"])
    assert asp.dtype == torch.int32 and np.array_equal(asp.numpy(), z["atom_split"])
    assert n_list == z["n_atoms"].tolist() and pt.shape == (len(mols), 4)


@pytest.mark.parametrize("tag", sorted(TAGS))
def test_weave_oracle_vs_reference_golden(tag):
    from oracle import nn_ref as R
    z = H.load("op_weave")
    up, wtype = TAGS[tag]
    p = {k[len(tag) + 4:]: torch.from_numpy(z[k]).double() for k in z.files if k.startswith(tag + ".sd.")}
    out = R.weave_model(p, torch.from_numpy(z["atom_feat"]).double(), torch.from_numpy(z["pair_feat"]).double(),
                        torch.from_numpy(z["pair_split"]), torch.from_numpy(z["atom_to_pair"]),
                        torch.from_numpy(z["atom_split"]), z["n_atoms"].tolist(), 2, up, wtype == "2D")
    H.assert_close(out.float(), z[tag + ".out"], tol=1e-5, what=tag)


def _net(z, tag):
    from jova_amd.nn.models import WeaveModel
    from jova_amd.utils.args import WeaveGatherArgs, WeaveLayerArgs
    up, wtype = TAGS[tag]
    dim = 20
    wargs = (WeaveLayerArgs(n_atom_input_feat=75, n_pair_input_feat=14, n_atom_output_feat=50, n_pair_output_feat=50,
                            update_pair=up, activation='relu', batch_norm=True, dropout=-1),
             WeaveLayerArgs(n_atom_input_feat=50, n_pair_input_feat=50 if up else 14, n_atom_output_feat=dim,
                            n_pair_output_feat=50, update_pair=up, batch_norm=True, dropout=-1, activation='relu'))
    wg = WeaveGatherArgs(conv_out_depth=dim, gaussian_expand=True, n_depth=dim)
    net = WeaveModel(wargs, weave_gath_arg=wg, update_pair=up, weave_type=wtype)
    sd = {k[len(tag) + 4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(tag + ".sd.")}
    return net, sd


@pytest.mark.parametrize("tag", sorted(TAGS))
def test_weave_state_dict_keys_are_the_reference_ones(tag):
    z = H.load("op_weave")
    net, sd = _net(z, tag)
    assert sorted(net.state_dict().keys()) == sorted(sd.keys())
    net.load_state_dict(sd)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", sorted(TAGS))
def test_weave_model_equals_the_reference_golden(device, tag):
    z = H.load("op_weave")
    net, sd = _net(z, tag)
    net.load_state_dict(sd)
    net = net.to(device).train()
    inp = [torch.from_numpy(z["atom_feat"]).to(device), torch.from_numpy(z["pair_feat"]).to(device),
           torch.from_numpy(z["pair_split"]).to(device), torch.from_numpy(z["atom_to_pair"]).to(device),
           torch.from_numpy(z["atom_split"]).to(device), z["n_atoms"].tolist()]
    out = net(inp)
    (out * torch.from_numpy(z[tag + ".gout"]).to(device)).sum().backward()
    H.assert_close(out, z[tag + ".out"], tol=1e-5, what="out")
    # gradients run through two BatchNorms over 24 atoms: the bound is derived from the reference's own fp32 run, per
    # tensor err(hip, oracle64) <= max(1e-5, 2 err(reference32, oracle64)), err = max|a - b| / max(|oracle64|, floor)
    from oracle import nn_ref as R
    up, wtype = TAGS[tag]
    p64 = {k: v.double().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
    o64 = R.weave_model(p64, torch.from_numpy(z["atom_feat"]).double(), torch.from_numpy(z["pair_feat"]).double(),
                        torch.from_numpy(z["pair_split"]), torch.from_numpy(z["atom_to_pair"]),
                        torch.from_numpy(z["atom_split"]), z["n_atoms"].tolist(), 2, up, wtype == "2D")
    (o64 * torch.from_numpy(z[tag + ".gout"]).double()).sum().backward()
    gmax = max(float(v.grad.abs().max()) for v in p64.values() if v.grad is not None)
    for k, p in net.named_parameters():
        want64 = p64[k].grad.numpy() if p64[k].grad is not None else np.zeros(tuple(p.shape))
        ref32 = z[tag + ".grad." + k]
        got = (p.grad if p.grad is not None else torch.zeros_like(p)).cpu().numpy()
        e_ref = H.rel_err(ref32, want64, 1e-3 * gmax)
        e = H.rel_err(got, want64, 1e-3 * gmax)
        assert e <= max(1e-5, 2 * e_ref), "grad %s: err %.3e, reference fp32 %.3e" % (k, e, e_ref)


@pytest.mark.gpu
def test_weave_view_inside_the_joint_model(device):
    """ecfp8 + weave + pcnn2d through create_integrated_net (create_weave_net, train_joint_attn.py:119-152)."""
    from jova_amd.train import create_integrated_net
    z = H.load("op_weave")
    bsz = len(z["n_atoms"])
    rng = np.random.RandomState(2)
    prof = {"P%d" % k: rng.randint(0, 60, size=(rng.randint(3, 8), 11)) for k in range(3)}
    hparams = {"attn_heads": 2, "attn_layers": 1, "lin_dims": [20], "output_dim": 1, "latent_dim": 32, "dprob": 0.0,
               "prot": {"model_types": ["pcnn2d"], "vocab_size": 60, "window": 11, "pcnn_num_layers": 2, "embedding_dim": 4},
               "weave": {"dim": 24, "update_pairs": False}, "ecfp8": {"dim": 64}}
    torch.manual_seed(4)
    net = create_integrated_net(hparams, prof, c_views=("ecfp8", "weave")).to(device).train()
    ecfp = torch.from_numpy((rng.uniform(size=(bsz, 64)) < 0.1).astype(np.float32)).to(device)
    winp = [torch.from_numpy(z["atom_feat"]).to(device), torch.from_numpy(z["pair_feat"]).to(device),
            torch.from_numpy(z["pair_split"]).to(device), torch.from_numpy(z["atom_to_pair"]).to(device),
            torch.from_numpy(z["atom_split"]).to(device), z["n_atoms"].tolist()]
    out = net([ecfp, winp, [("davis", "P%d" % (k % 3)) for k in range(bsz)]])
    assert out.shape == (bsz, 1) and torch.isfinite(out).all()
    out.sum().backward()
    w = net[0].models[1][0].weave_seq[0].linear_AA.weight
    assert w.grad is not None and torch.isfinite(w.grad).all() and float(w.grad.abs().max()) > 0
