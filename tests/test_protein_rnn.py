"""§8(f3): ProteinRNN (jova/nn/models.py:657-682) -- oracle and HIP LSTM against the reference-made golden
(tests/golden/op_protein_rnn.npz, tools/make_golden_rnn.py)."""
import numpy as np
import pytest
import torch

import helpers as H

KEYS = ("model.weight_ih_l0", "model.weight_hh_l0", "model.bias_ih_l0", "model.bias_hh_l0")


def test_protein_rnn_oracle_vs_reference_golden():
    from oracle import nn_ref as R
    z = H.load("op_protein_rnn")
    p = [torch.from_numpy(z["sd." + k]).double().requires_grad_(True) for k in KEYS]
    x = torch.from_numpy(z["x"]).double().requires_grad_(True)
    out = R.protein_rnn(x, *p)
    (out * torch.from_numpy(z["gout"]).double()).sum().backward()
    H.assert_close(out.detach().float(), z["out"], tol=1e-5, what="out")
    H.assert_close(x.grad.float(), z["gx"], tol=2e-5, what="dx")
    for k, t in zip(KEYS, p):
        H.assert_close(t.grad.float(), z["grad." + k], tol=2e-5, what=k)


def test_protein_rnn_state_dict_keys_and_unsupported_variants():
    from jova_amd.nn.models import ProteinRNN
    z = H.load("op_protein_rnn")
    T, B, K, Hd = [int(v) for v in z["cfg"]]
    net = ProteinRNN(in_dim=K, hidden_dim=Hd, dropout=0.2)
    assert sorted(net.state_dict().keys()) == sorted(KEYS)
    assert all(tuple(net.state_dict()[k].shape) == z["sd." + k].shape for k in KEYS)
    with pytest.raises(NotImplementedError):
        ProteinRNN(in_dim=K, hidden_dim=Hd, dropout=0.0, num_layers=2)
    with pytest.raises(NotImplementedError):
        ProteinRNN(in_dim=K, hidden_dim=Hd, dropout=0.0, bidrectional=True)


@pytest.mark.gpu
def test_protein_rnn_equals_the_reference_golden(device):
    from jova_amd.nn.models import ProteinRNN
    z = H.load("op_protein_rnn")
    T, B, K, Hd = [int(v) for v in z["cfg"]]
    net = ProteinRNN(in_dim=K, hidden_dim=Hd, dropout=0.2)
    net.load_state_dict({k: torch.from_numpy(z["sd." + k]) for k in KEYS})
    net = net.to(device)
    x = torch.from_numpy(z["x"]).to(device).requires_grad_(True)
    out = net(x)
    (out * torch.from_numpy(z["gout"]).to(device)).sum().backward()
    H.assert_close(out, z["out"], tol=1e-5, what="out")
    H.assert_close(x.grad, z["gx"], tol=2e-5, what="dx")
    for k, p in net.named_parameters():
        H.assert_close(p.grad, z["grad." + k], tol=2e-5, what=k)


@pytest.mark.gpu
def test_protein_rnn_long_sequence_and_limits(device):
    """P = 1024 steps, B = 128 (the long-protein shape): against the fp64 oracle; hidden > 64 is rejected loudly."""
    from jova_amd.nn.models import ProteinRNN
    from oracle import nn_ref as R
    torch.manual_seed(2)
    net = ProteinRNN(in_dim=22, hidden_dim=27, dropout=0.0)
    x = torch.randn(1024, 16, 22) * 0.5
    p = [net.state_dict()[k].double() for k in KEYS]
    want = R.protein_rnn(x.double(), *p)
    out = net.to(device)(x.to(device))
    H.assert_close(out, want.float(), tol=2e-5, what="long sequence")
    big = ProteinRNN(in_dim=8, hidden_dim=65, dropout=0.0).to(device)
    with pytest.raises(RuntimeError):
        big(torch.randn(3, 2, 8, device=device))


@pytest.mark.gpu
def test_rnn_protein_view_inside_the_joint_model(device):
    """ecfp8 + gconv + rnn through create_integrated_net (train_joint_attn.py:77-86, :212-213): one training step runs and
    every LSTM parameter receives a finite, non-zero gradient."""
    from test_gpu_ops import _e2e_setup
    from jova_amd.train import create_integrated_net
    z, hp, hparams, net0, batch = _e2e_setup(device)
    hparams = dict(hparams)
    hparams["prot"] = dict(hparams["prot"], model_types=["rnn"], rnn_hidden_state_dim=27)
    torch.manual_seed(3)
    net = create_integrated_net(hparams, H.profile(z)).to(device).train()
    out = net(batch.model_input())
    assert out.shape[0] == batch.batch_size and torch.isfinite(out).all()
    out.sum().backward()
    lstm = net[0].models[2][1].model
    for k, p in lstm.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().max()) > 0, k
