"""§8(f3), first encoder: GraphNeuralNet2D (jova/nn/models.py:352-418) on the HIP kernels against the reference-made golden
(tests/golden/op_gnn2d.npz), its collator against the reference's layout, and the view inside the joint model."""
import numpy as np
import pytest
import torch

import helpers as H


class _Mol:
    def __init__(self, adjacency, fingerprints):
        self.adjacency, self.fingerprints = adjacency, fingerprints


class _Prot:
    def __init__(self, name):
        self._n = name

    def get_name(self):
        return self._n


def _golden_mols(z):
    mols, o = [], 0
    for n in z["axis"].tolist():
        mols.append(_Mol(z["A"][o:o + n, o:o + n].copy(), z["fingerprints"][o:o + n].copy()))
        o += n
    return mols


def test_gnn_collator_layout_equals_the_reference_description():
    """process_gnn_view_data (data.py:309-335): fingerprints = concatenation, adjacency = block diagonal, M = molecule size
    repeated per atom as [N, 1], axis = sizes; the golden's A / M / ids were laid out that way for the reference model."""
    from jova_amd.data import data as D
    z = H.load("op_gnn2d")
    mols = _golden_mols(z)
    X = np.empty((len(mols), 3, 1), dtype=object)
    for k, m in enumerate(mols):
        X[k, 0, 0] = (m, _Prot("P%d" % (k % 2)))
    prot = {"P0": np.zeros((1, 5), np.float32), "P1": np.ones((1, 5), np.float32)}
    old = D._device
    D._device = lambda: torch.device("cpu")
    try:
        (fp, A, M, axis), pt, names, x_data = D.process_gnn_view_data(X, prot, 0, cuda_prot=False)
    finally:
        D._device = old
    assert fp.dtype == torch.int64 and np.array_equal(fp.numpy(), z["fingerprints"])
    assert A.dtype == torch.float32 and np.array_equal(A.numpy(), z["A"])
    assert M.shape == (len(z["fingerprints"]), 1) and np.array_equal(M.numpy(), z["M"])
    assert axis == z["axis"].tolist() and names == ["P%d" % (k % 2) for k in range(len(mols))]
    assert pt.shape == (len(mols), 5) and pt.dtype == torch.float32


def test_gnn_state_dict_keys_are_the_reference_ones():
    from jova_amd.nn.models import GraphNeuralNet2D
    z = H.load("op_gnn2d")
    n_fp, dim, layers = [int(v) for v in z["cfg"]]
    net = GraphNeuralNet2D(n_fp, dim, num_layers=layers)
    want = sorted(k[3:] for k in z.files if k.startswith("sd."))
    assert sorted(net.state_dict().keys()) == want
    assert all(tuple(net.state_dict()[k].shape) == z["sd." + k].shape for k in want)


@pytest.mark.gpu
@pytest.mark.parametrize("update", ["mean", "sum"])
def test_gnn2d_equals_the_reference_golden(device, update):
    from jova_amd.nn.models import GraphNeuralNet2D
    z = H.load("op_gnn2d")
    n_fp, dim, layers = [int(v) for v in z["cfg"]]
    net = GraphNeuralNet2D(n_fp, dim, num_layers=layers, update=update)
    net.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")})
    net = net.to(device)
    inp = (torch.from_numpy(z["fingerprints"]).to(device), torch.from_numpy(z["A"]).to(device),
           torch.from_numpy(z["M"]).to(device), z["axis"].tolist())
    out = net(inp)
    (out * torch.from_numpy(z[update + ".gout"]).to(device)).sum().backward()
    H.assert_close(out, z[update + ".out"], tol=1e-5, what="out")
    for k, p in net.named_parameters():
        H.assert_close(p.grad, z[update + ".grad." + k], tol=2e-5, what="grad " + k)


@pytest.mark.gpu
@pytest.mark.parametrize("update", ["mean", "sum"])
def test_gnn2d_on_the_sparse_block_adjacency_equals_the_reference_golden(device, update):
    """The collator's GPU form of the adjacency (BlockAdjacency: CSR of every molecule's block, jova_spmm_csr forward and
    backward) against the same reference-made golden as the dense form, its to_dense() against the reference's matrix, and
    an ASYMMETRIC adjacency (A != A^T) against the dense path, so the backward really uses A^T."""
    from jova_amd.data import data as D
    from jova_amd.nn import functional as F_
    from jova_amd.nn.models import GraphNeuralNet2D
    z = H.load("op_gnn2d")
    n_fp, dim, layers = [int(v) for v in z["cfg"]]
    mols = _golden_mols(z)
    X = np.empty((len(mols), 3, 1), dtype=object)
    for k, m in enumerate(mols):
        X[k, 0, 0] = (m, _Prot("P0"))
    (fp, A, M, axis), _, _, _ = D.process_gnn_view_data(X, {"P0": np.zeros((1, 5), np.float32)}, 0, cuda_prot=False)
    assert isinstance(A, F_.BlockAdjacency) and A.is_cuda and fp.is_cuda
    assert np.array_equal(A.to_dense().cpu().numpy(), z["A"]) and np.array_equal(M.cpu().numpy(), z["M"])
    assert np.array_equal(fp.cpu().numpy(), z["fingerprints"]) and axis == z["axis"].tolist()
    net = GraphNeuralNet2D(n_fp, dim, num_layers=layers, update=update)
    net.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")})
    net = net.to(device)
    out = net((fp, A, M, axis))
    (out * torch.from_numpy(z[update + ".gout"]).to(device)).sum().backward()
    H.assert_close(out, z[update + ".out"], tol=1e-5, what="out")
    for k, p in net.named_parameters():
        H.assert_close(p.grad, z[update + ".grad." + k], tol=2e-5, what="grad " + k)
    # asymmetric, weighted adjacency: sparse and dense paths agree (forward and the A^T of the backward)
    rng = np.random.RandomState(3)
    for m in mols:
        m.adjacency = (m.adjacency * rng.uniform(0.5, 1.5, size=m.adjacency.shape)).astype(np.float32)
        m.adjacency[0, -1] = 0.0
    (fp, A2, M, axis), _, _, _ = D.process_gnn_view_data(X, {"P0": np.zeros((1, 5), np.float32)}, 0, cuda_prot=False)
    dense = A2.to_dense()
    assert not torch.equal(dense, dense.t())
    grads = []
    for adj in (A2, dense):
        net.zero_grad()
        o = net((fp, adj, M, axis))
        (o * torch.from_numpy(z[update + ".gout"]).to(device)).sum().backward()
        grads.append((o.detach().clone(), {k: p.grad.detach().clone() for k, p in net.named_parameters()}))
    H.assert_close(grads[0][0], grads[1][0], tol=2e-6, what="sparse vs dense out")
    for k in grads[0][1]:
        H.assert_close(grads[0][1][k], grads[1][1][k], tol=1e-5, what="sparse vs dense grad " + k)


@pytest.mark.gpu
def test_gnn_embedding_flags_an_id_outside_the_table(device):
    from jova_amd.nn import functional as F_
    w = torch.randn(5, 4, device=device)
    out = F_.EmbeddingFunction.apply(w, torch.tensor([0, 4, 5], device=device))
    assert torch.isnan(out[2]).all() and torch.equal(out[:2], w[[0, 4]])
    with pytest.raises(IndexError):
        F_.EmbeddingFunction.raise_if_bad()
    F_.EmbeddingFunction.raise_if_bad()                                  # the flag was cleared


@pytest.mark.gpu
def test_gnn_view_inside_the_joint_model_vs_oracle(device):
    """ecfp8 + gnn + pcnn2d through create_integrated_net (create_gnn_net, train_joint_attn.py:177-184): forward output and
    one fingerprint-embedding gradient against the fp64 oracle composed from its pieces."""
    from jova_amd.train import create_integrated_net
    from oracle import nn_ref as R
    z = H.load("op_gnn2d")
    n_fp, dim, layers = [int(v) for v in z["cfg"]]
    bsz = len(z["axis"])
    rng = np.random.RandomState(2)
    prof = {"P%d" % k: rng.randint(0, 60, size=(rng.randint(3, 8), 11)) for k in range(4)}
    hparams = {"attn_heads": 2, "attn_layers": 1, "lin_dims": [20], "output_dim": 1, "latent_dim": 32, "dprob": 0.0,
               "prot": {"model_types": ["pcnn2d"], "vocab_size": 60, "window": 11, "pcnn_num_layers": 2, "embedding_dim": 4},
               "gnn": {"dim": dim, "fingerprint_size": n_fp, "num_layers": layers}, "ecfp8": {"dim": 64}}
    torch.manual_seed(4)
    net = create_integrated_net(hparams, prof, c_views=("ecfp8", "gnn")).to(device).train()
    ecfp = torch.from_numpy((rng.uniform(size=(bsz, 64)) < 0.1).astype(np.float32)).to(device)
    gnn_in = (torch.from_numpy(z["fingerprints"]).to(device), torch.from_numpy(z["A"]).to(device),
              torch.from_numpy(z["M"]).to(device), z["axis"].tolist())
    prots = [("davis", "P%d" % (k % 4)) for k in range(bsz)]
    out = net([ecfp, gnn_in, prots])
    assert out.shape == (bsz, 1) and torch.isfinite(out).all()
    out.sum().backward()
    emb = net[0].models[1][0].embed_fingerprint.weight
    assert emb.grad is not None and torch.isfinite(emb.grad).all() and float(emb.grad.abs().max()) > 0
    # the gnn view's own output inside the model equals the oracle's on the same weights
    sd = {k: v.detach().cpu().double() for k, v in net[0].models[1][0].state_dict().items()}
    want = R.gnn2d(torch.from_numpy(z["fingerprints"]), torch.from_numpy(z["A"]).double(), torch.from_numpy(z["M"]).double(),
                   z["axis"].tolist(), sd["embed_fingerprint.weight"], [sd["W_gnn.%d.weight" % k] for k in range(layers)],
                   [sd["W_gnn.%d.bias" % k] for k in range(layers)])
    with torch.no_grad():
        got = net[0].models[1][0](gnn_in)
    H.assert_close(got, want.float(), tol=1e-5, what="gnn view")
