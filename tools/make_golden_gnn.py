"""Writes tests/golden/op_gnn2d.npz with the reference's GraphNeuralNet2D (jova/nn/models.py:352-418, imported from
/root/reference through tools/make_golden.py): inputs as its collator builds them (process_gnn_view_data,
jova/data/data.py:309-335: fingerprint ids, block-diagonal adjacency, M, axis), outputs and gradients."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as MG                                             # noqa: E402
from make_golden import M as RM                                      # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def main():
    r = np.random.RandomState(31)
    sizes = [7, 12, 3, 9, 2, 16]
    n_fp, dim, layers = 40, 24, 3
    fps, blocks = [], []
    for n in sizes:
        a = np.triu((r.uniform(size=(n, n)) < 0.3).astype(np.float32), 1)
        a = a + a.T + np.eye(n, dtype=np.float32)                    # the reference's adjacency carries the self loop
        blocks.append(a)
        fps.append(r.randint(0, n_fp, size=n))
    N = sum(sizes)
    A = np.zeros((N, N), np.float32)
    o = 0
    for b in blocks:
        A[o:o + len(b), o:o + len(b)] = b
        o += len(b)
    Mv = np.concatenate([np.repeat(n, n) for n in sizes]).astype(np.float32).reshape(-1, 1)
    fp = np.concatenate(fps).astype(np.int64)
    arrays = dict(fingerprints=fp, A=A, M=Mv, axis=np.array(sizes), cfg=np.array([n_fp, dim, layers]))
    for update in ("mean", "sum"):
        torch.manual_seed(5)
        net = RM.GraphNeuralNet2D(num_fingerprints=n_fp, embedding_dim=dim, num_layers=layers, update=update)
        if update == "mean":
            for k, v in net.state_dict().items():
                arrays["sd." + k] = v.numpy().copy()
        out = net((torch.from_numpy(fp), torch.from_numpy(A), torch.from_numpy(Mv), sizes))
        g = torch.from_numpy(r.normal(size=tuple(out.shape)).astype(np.float32))
        (out * g).sum().backward()
        arrays[update + ".out"] = out.detach().numpy()
        arrays[update + ".gout"] = g.numpy()
        for k, p in net.named_parameters():
            arrays[update + ".grad." + k] = p.grad.numpy().copy()
        print(update, out.shape, float(out.abs().max()))
    np.savez_compressed(os.path.join(OUT, "op_gnn2d.npz"), **arrays)


if __name__ == "__main__":
    main()
