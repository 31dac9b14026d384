"""Writes tests/golden/op_weave.npz with the reference's weave classes (jova/nn/layers.py:72-262, jova/nn/models.py:245-307,
imported from /root/reference through tools/make_golden.py).  The reference's WeaveLayer / WeaveGather call scatter_add
from torch_scatter, whose import it has commented out (layers.py:17) and which is not installed here; this script binds
that name in the reference module to the function's published semantics (out[index[r]] += src[r] along dim 0, out sized
max(index) + 1, fill_value 0) -- the only line that is not the reference's own code.  Inputs are laid out as its collator
does (process_weave_view_data, jova/data/data.py:215-269)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as MG                                             # noqa: E402
from make_golden import L as RL, M as RM                             # noqa: E402
import jova.utils.args as RA                                         # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def scatter_add(src, index, dim=0, fill_value=0):
    assert dim == 0
    out = torch.full((int(index.max()) + 1,) + tuple(src.shape[1:]), float(fill_value), dtype=src.dtype)
    return out.index_add(0, index, src)


RL.scatter_add = scatter_add
# the reference passes a Python float to Normal.log_prob (layers.py:201), which torch >= 1.8 rejects when argument
# validation is on; with validation off the same arithmetic runs
torch.distributions.Distribution.set_default_validate_args(False)


def collate(mols):
    """process_weave_view_data's molecule part, restated with the same numpy calls."""
    atom_feat, pair_feat, atom_split, atom_to_pair, pair_split, n_list = [], [], [], [], [], []
    start = 0
    for im, (nodes, pairs) in enumerate(mols):
        n = nodes.shape[0]
        n_list.append(n)
        atom_split.extend([im] * n)
        C0, C1 = np.meshgrid(np.arange(n), np.arange(n))
        atom_to_pair.append(np.transpose(np.array([C1.flatten() + start, C0.flatten() + start])))
        pair_split.extend(C1.flatten() + start)
        start += n
        atom_feat.append(nodes)
        pair_feat.append(np.reshape(pairs, (n * n, pairs.shape[2])))
    return (np.concatenate(atom_feat, 0).astype(np.float32), np.concatenate(pair_feat, 0).astype(np.float32),
            np.array(pair_split, dtype=np.int32), np.ascontiguousarray(np.concatenate(atom_to_pair, 0).astype(np.int64)),
            np.array(atom_split, dtype=np.int32), n_list)


def main():
    r = np.random.RandomState(17)
    sizes = [5, 9, 3, 7]
    mols = [((r.uniform(size=(n, 75)) < 0.15).astype(np.float32), (r.uniform(size=(n, n, 14)) < 0.2).astype(np.float32))
            for n in sizes]
    af, pf, ps, a2p, asp, n_list = collate(mols)
    arrays = dict(atom_feat=af, pair_feat=pf, pair_split=ps, atom_to_pair=a2p, atom_split=asp, n_atoms=np.array(n_list))
    for k, (nodes, pairs) in enumerate(mols):
        arrays["mol%d.nodes" % k], arrays["mol%d.pairs" % k] = nodes, pairs
    for tag, update_pairs, wtype in (("up2d", True, "2D"), ("noup2d", False, "2D"), ("up1d", True, "1D")):
        dim = 20
        wargs = (RA.WeaveLayerArgs(n_atom_input_feat=75, n_pair_input_feat=14, n_atom_output_feat=50, n_pair_output_feat=50,
                                   update_pair=update_pairs, activation='relu', batch_norm=True, dropout=-1),
                 RA.WeaveLayerArgs(n_atom_input_feat=50, n_pair_input_feat=50 if update_pairs else 14,
                                   n_atom_output_feat=dim, n_pair_output_feat=50, update_pair=update_pairs, batch_norm=True,
                                   dropout=-1, activation='relu'))
        wg = RA.WeaveGatherArgs(conv_out_depth=dim, gaussian_expand=True, n_depth=dim)
        torch.manual_seed(21)
        net = RM.WeaveModel(wargs, weave_gath_arg=wg, update_pair=update_pairs, weave_type=wtype)
        net.train()
        for k, v in net.state_dict().items():
            arrays[tag + ".sd." + k] = v.numpy().copy()
        inp = [torch.from_numpy(af), torch.from_numpy(pf), torch.from_numpy(ps), torch.from_numpy(a2p), torch.from_numpy(asp),
               n_list]
        out = net(inp)
        g = torch.from_numpy(r.normal(size=tuple(out.shape)).astype(np.float32))
        (out * g).sum().backward()
        arrays[tag + ".out"], arrays[tag + ".gout"] = out.detach().numpy(), g.numpy()
        for k, p in net.named_parameters():
            arrays[tag + ".grad." + k] = p.grad.numpy().copy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
        print(tag, tuple(out.shape), float(out.abs().max()))
    np.savez_compressed(os.path.join(OUT, "op_weave.npz"), **arrays)


if __name__ == "__main__":
    main()
