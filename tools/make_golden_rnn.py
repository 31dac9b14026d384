"""Writes tests/golden/op_protein_rnn.npz with the reference's ProteinRNN (jova/nn/models.py:657-682, imported from
/root/reference through tools/make_golden.py): the DTI driver's configuration (one layer, one direction, hidden 27,
in_dim = embedding 6 x window 11 here), input as Prot2Vec hands it over ([P, B, in], padded segments are zero rows)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as MG                                             # noqa: E402
from make_golden import M as RM                                      # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def main():
    r = np.random.RandomState(12)
    T, B, K, H = 37, 9, 66, 27
    x = r.normal(size=(T, B, K)).astype(np.float32)
    for b in range(B):                                               # ragged proteins: zero rows past their length
        x[r.randint(T // 2, T + 1):, b] = 0.0
    torch.manual_seed(9)
    net = RM.ProteinRNN(in_dim=K, hidden_dim=H, dropout=0.2, batch_first=False)
    arrays = dict(x=x, cfg=np.array([T, B, K, H]))
    for k, v in net.state_dict().items():
        arrays["sd." + k] = v.numpy().copy()
    xt = torch.from_numpy(x).requires_grad_(True)
    out = net(xt)
    g = torch.from_numpy(r.normal(size=tuple(out.shape)).astype(np.float32))
    (out * g).sum().backward()
    arrays.update(out=out.detach().numpy(), gout=g.numpy(), gx=xt.grad.numpy())
    for k, p in net.named_parameters():
        arrays["grad." + k] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "op_protein_rnn.npz"), **arrays)
    print(out.shape, float(out.abs().max()), sorted(net.state_dict().keys()))


if __name__ == "__main__":
    main()
