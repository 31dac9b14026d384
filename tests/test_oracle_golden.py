"""Pins the oracle (oracle/*.py) against golden vectors exported from the imported reference.

CPU only.  Integer layout: bit-exact.  Floating point: 1e-5 of the tensor's scale (fp32 oracle vs fp32 reference);
the fp64 oracle must agree with the fp32 reference to the same bound, which is what makes it usable as the yard-stick
for the HIP path.
"""
import numpy as np
import pytest
import torch

import helpers as H
from oracle import mol_graphs as og
from oracle import nn_ref as R

INDEX_SETS = ["toy4", "deg0_b16", "hubs_b8", "pair_b2", "davis_b128"]


@pytest.mark.parametrize("name", INDEX_SETS)
def test_index_layout_bit_exact(name):
    z = H.load("index_" + name)
    b = H.oracle_batch(H.unpack_mols(z))
    assert str(z["nodes_dtype"]) == "int64" and b["nodes"].dtype == np.int64
    assert np.array_equal(b["nodes"], z["nodes"].astype(np.int64))
    assert b["deg_slice"].dtype == np.dtype(str(z["deg_slice_dtype"]))
    assert np.array_equal(b["deg_slice"], z["deg_slice"])
    assert np.array_equal(b["membership"], z["membership"])
    assert list(b["n_atoms"]) == list(z["n_atoms"])
    for d in range(11):
        ref = z["adj_%d" % d]
        assert b["deg_adj_lists"][d].dtype == ref.dtype == np.int32
        assert b["deg_adj_lists"][d].shape == ref.shape
        assert np.array_equal(b["deg_adj_lists"][d], ref)


def test_index_edge_cases_present():
    """The fixtures really contain the edge cases: degree-0 atoms, empty buckets, degree up to 10."""
    z = H.load("index_deg0_b16")
    assert z["deg_slice"][0, 1] >= 4 and (z["deg_slice"][5:, 1] == 0).all()
    z = H.load("index_hubs_b8")
    assert z["deg_slice"][10, 1] > 0 and (z["deg_slice"][5:, 1] > 0).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("tag", ["l1", "l2", "lin"])
def test_graphconv(tag, dtype):
    z = H.load("op_graphconv")
    mols = H.unpack_mols(z)
    p = H.sd(z, tag + ".sd.", dtype)
    w = [p["linear_layers.%d.weight" % k].requires_grad_(True) for k in range(21)]
    b = [p["linear_layers.%d.bias" % k].requires_grad_(True) for k in range(21)]
    x = None
    if tag != "l1":
        x = torch.from_numpy(z[tag + ".x"]).to(dtype).requires_grad_(True)
    gl = H.oracle_glist(mols, x)
    out = R.graph_conv(gl[0], gl[1], gl[4:], w, b, relu=(tag != "lin"))
    H.assert_close(out, z[tag + ".out"], what="out")
    (out * torch.from_numpy(z[tag + ".gout"]).to(dtype)).sum().backward()
    for k in range(21):
        H.assert_close(w[k].grad, z["%s.grad.linear_layers.%d.weight" % (tag, k)], what="dW%d" % k)
        H.assert_close(b[k].grad, z["%s.grad.linear_layers.%d.bias" % (tag, k)], what="db%d" % k)
    if x is not None:
        H.assert_close(x.grad, z[tag + ".gx"], what="dx")


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_graphpool_ties(dtype):
    z = H.load("op_graphpool")
    x = torch.from_numpy(z["x"]).to(dtype).requires_grad_(True)
    gl = H.oracle_glist(H.unpack_mols(z), x)
    out = R.graph_pool(gl[0], gl[1], gl[4:])
    assert np.array_equal(out.detach().numpy().astype(np.float32), z["out"])
    (out * torch.from_numpy(z["gout"]).to(dtype)).sum().backward()
    H.assert_close(x.grad, z["gx"], tol=1e-6, what="dx (tie routing)")


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_gather2d_nonsat(tag):
    z = H.load("op_gather2d_nonsat")
    n_atoms = [int(v) for v in z[tag + ".n_atoms"]]
    x = torch.from_numpy(z[tag + ".x"]).requires_grad_(True)
    pad = R.gather2d(x, n_atoms)
    assert np.array_equal(torch.tanh(pad).detach().numpy(), z[tag + ".out_tanh"])
    out, n_eval = R.nonsat(pad)
    assert 2 <= n_eval <= 12
    assert np.array_equal(out.detach().numpy(), z[tag + ".out"])          # same op order on CPU -> identical
    (out * torch.from_numpy(z[tag + ".gout"])).sum().backward()
    H.assert_close(x.grad, z[tag + ".gx"], tol=1e-6, what="dx")
    # fp64 oracle takes the same number of evaluations (decision margin is wide on these fixtures)
    out64, n_eval64 = R.nonsat(R.gather2d(x.detach().double(), n_atoms))
    assert n_eval64 == n_eval
    H.assert_close(out64, z[tag + ".out"], what="fp64 out")


def test_prot2vec():
    z = H.load("op_prot2vec")
    prof = H.profile(z)
    names = ["P%03d" % i for i in z["batch_prot"]]
    idx = og.embedding_indices(names, prof, fill_val=60)
    assert idx.dtype == np.int64 and np.array_equal(idx, z["idx"])
    emb = torch.from_numpy(z["sd.embedding.weight"]).requires_grad_(True)
    out = R.prot2vec(torch.from_numpy(idx), emb)
    assert np.array_equal(out.detach().numpy(), z["out"])
    (out * torch.from_numpy(z["gout"])).sum().backward()
    g = emb.grad.clone()
    g[60] = 0                                   # padding_idx row receives no gradient in nn.Embedding
    H.assert_close(g, z["grad.embedding.weight"], tol=1e-6, what="dE")
    assert np.all(z["grad.embedding.weight"][60] == 0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_pcnn2d(dtype):
    z = H.load("op_pcnn2d")
    p = H.sd(z, "sd.", dtype)
    ws = [p["lin_kernels.%d.weight" % i].requires_grad_(True) for i in range(3)]
    bs = [p["lin_kernels.%d.bias" % i].requires_grad_(True) for i in range(3)]
    x = torch.from_numpy(z["x"]).to(dtype).requires_grad_(True)
    out = R.pcnn2d(x, ws, bs)
    H.assert_close(out, z["out"], what="out")
    (out * torch.from_numpy(z["gout"]).to(dtype)).sum().backward()
    H.assert_close(x.grad, z["gx"], what="dx")
    for i in range(3):
        H.assert_close(ws[i].grad, z["grad.lin_kernels.%d.weight" % i], what="dW")
        H.assert_close(bs[i].grad, z["grad.lin_kernels.%d.bias" % i], what="db")


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_joint_attention(dtype):
    z = H.load("op_joint_attention")
    p = {k: v.requires_grad_(True) if v.is_floating_point() else v for k, v in H.sd(z, "sd.", dtype).items()}
    xs = [torch.from_numpy(z["x%d" % i]).to(dtype).requires_grad_(True) for i in range(3)]
    out = R.joint_attention(xs, p, int(z["heads"]))
    H.assert_close(out, z["out"], what="out")
    (out * torch.from_numpy(z["gout"]).to(dtype)).sum().backward()
    for i in range(3):
        H.assert_close(xs[i].grad, z["gx%d" % i], tol=2e-5, what="dx%d" % i)
    floor = H.grad_floor(z, "grad.")
    for k in z.files:
        if k.startswith("grad."):
            H.assert_close(p[k[5:]].grad, z[k], tol=2e-5, what=k, floor=floor)


def _e2e_batch(z, dtype):
    mols = H.unpack_mols(z)
    prof = H.profile(z)
    names = sorted(prof)
    idx = og.embedding_indices([names[i] for i in z["prot_ids"]], prof, fill_val=H.hp(z)["vocab"])
    return dict(ecfp=torch.from_numpy(z["ecfp"]).to(dtype), glist=H.oracle_glist(mols),
                prot_idx=torch.from_numpy(idx))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_e2e_step(dtype):
    z = H.load("e2e_small")
    hp = H.hp(z)
    p = {k: (v.requires_grad_(True) if v.is_floating_point() and "running" not in k else v)
         for k, v in H.sd(z, "sd0.", dtype).items()}
    batch = _e2e_batch(z, dtype)
    out, n_eval = R.dti_forward(p, batch, hp["heads"], hp["pcnn_layers"], len(hp["lin_dims"]))
    H.assert_close(out, z["step0.out"], tol=2e-5, what="out")
    loss = R.mse_loss(out, torch.from_numpy(z["w"]).to(dtype), torch.from_numpy(z["y"]).to(dtype))
    assert abs(float(loss.detach()) - float(z["step0.loss"])) <= 1e-5 * abs(float(z["step0.loss"]))
    loss.backward()
    floor = H.grad_floor(z, "step0.grad.")
    for k in z.files:
        if k.startswith("step0.grad."):
            name = k[len("step0.grad."):]
            g = p[name].grad
            if g is None:
                g = torch.zeros_like(p[name])
            if name.endswith("embedding.weight"):
                g = g.clone()
                g[hp["vocab"]] = 0
            # the tiny fixture model (batch 8, BatchNorm/LayerNorm chains) is ill-conditioned: the reference's own
            # fp32 gradients sit up to 1.3e-4 from the fp64 oracle on the LayerNorm-cancelled tensors
            H.assert_close(g, z[k], tol=5e-4, what=k, floor=floor)
    # SGD step of torch.optim.SGD(lr, weight_decay): w <- w - lr * (g + wd * w)
    for k in z.files:
        if k.startswith("step0.sd.") and "running" not in k and "num_batches" not in k:
            name = k[len("step0.sd."):]
            g = p[name].grad if p[name].grad is not None else torch.zeros_like(p[name])
            if name.endswith("embedding.weight"):
                g = g.clone()
                g[hp["vocab"]] = 0
            new = p[name].detach() - hp["lr"] * (g + hp["wd"] * p[name].detach())
            # update error = weight rounding (1e-6 of the weights) + lr * gradient error (bound used above)
            gmax = max(float(np.abs(z["step0.grad." + name]).max()), floor)
            bound = 1e-6 * float(np.abs(z[k]).max()) + hp["lr"] * 5e-4 * gmax
            assert float(np.abs(new.numpy() - z[k]).max()) <= bound, k


def test_manifest_matches_survey():
    import json
    import os
    man = json.load(open(os.path.join(H.GOLDEN, "manifest_tuned.json")))
    assert man["n_trainable"] == 4900207
    assert man["state_dict"]["0.models.2.0.embedding.weight"] == [9049, 22]
    assert man["state_dict"]["1.attention_models.0.in_proj_weight"] == [1536, 512]


def test_segment_kats():
    """The reference's own known-answer tests (jova/utils/tests/test_math.py:17-55) against the oracle's reductions."""
    z = H.load("kat_segment")
    data, ids = torch.from_numpy(z["seg_sum_data"]), torch.from_numpy(z["seg_ids"])
    out = torch.zeros(4, dtype=data.dtype).index_add_(0, ids, data)
    assert out.tolist() == z["seg_sum_expect"].tolist()
    mx = [int(data[ids == s].max()) for s in range(4)]
    assert mx == z["seg_max_expect"].tolist()


def test_gnn2d_oracle_vs_reference_golden():
    """oracle/nn_ref.gnn2d against the reference's GraphNeuralNet2D (tests/golden/op_gnn2d.npz, tools/make_golden_gnn.py):
    outputs and parameter gradients, 'mean' and 'sum' updates."""
    import torch
    from oracle import nn_ref as R
    z = H.load("op_gnn2d")
    n_fp, dim, layers = [int(v) for v in z["cfg"]]
    for update in ("mean", "sum"):
        emb = torch.from_numpy(z["sd.embed_fingerprint.weight"]).double().requires_grad_(True)
        ws = [torch.from_numpy(z["sd.W_gnn.%d.weight" % k]).double().requires_grad_(True) for k in range(layers)]
        bs = [torch.from_numpy(z["sd.W_gnn.%d.bias" % k]).double().requires_grad_(True) for k in range(layers)]
        out = R.gnn2d(torch.from_numpy(z["fingerprints"]), torch.from_numpy(z["A"]), torch.from_numpy(z["M"]),
                      z["axis"].tolist(), emb, ws, bs, update)
        (out * torch.from_numpy(z[update + ".gout"]).double()).sum().backward()
        H.assert_close(out.detach().float(), z[update + ".out"], tol=1e-5, what=update + " out")
        H.assert_close(emb.grad.float(), z[update + ".grad.embed_fingerprint.weight"], tol=2e-5, what=update + " dE")
        for k in range(layers):
            H.assert_close(ws[k].grad.float(), z[update + ".grad.W_gnn.%d.weight" % k], tol=2e-5, what="dW%d" % k)
            H.assert_close(bs[k].grad.float(), z[update + ".grad.W_gnn.%d.bias" % k], tol=2e-5, what="db%d" % k)
