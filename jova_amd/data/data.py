"""Reference-shaped batch collators: same names, arguments and return tuples as /root/reference/jova/data/data.py
(batch_collator :166-187, process_ecfp_view_data :190-212, process_gconv_view_data :272-306, DtiDataset :108-134), so
the reference's train loop (proj/dti/train_joint_attn.py:412-453) can call them unchanged.

Inputs are duck-typed like the reference's: a pair is ``(mol, prot)``; a gconv ``mol`` is a ConvMol
(jova_amd.feat.mol_graphs), an ecfp ``mol`` has ``get_array()``; ``prot`` has ``get_name()``;
``prot_desc_dict[name]`` is the protein's descriptor of shape [1, psc_dim].

What differs is underneath: the batch merge is native (jova_collate_gconv) and every view goes up in one pinned,
asynchronous copy; tensors land on the current CUDA device when there is one (the reference's ``cuda()`` helper,
jova/utils/train_helpers or data.py ``cuda``), else stay on the CPU.
"""
import numpy as np
import torch

from .collate import gconv_input_list

__all__ = ["DtiDataset", "batch_collator", "process_ecfp_view_data", "process_gconv_view_data", "cuda"]


def _device():
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


def cuda(tensor):
    """Counterpart of the reference's cuda(): moves to the GPU when one is present."""
    dev = _device()
    if dev.type != "cuda":
        return tensor
    return tensor.pin_memory().to(dev, non_blocking=True) if not tensor.is_cuda else tensor


class DtiDataset(torch.utils.data.Dataset):
    """data.py:108-134: item = ([x of every view], [y of every view], [w of every view])."""

    def __init__(self, x_s, y_s, w_s):
        assert len(x_s) == len(y_s) == len(w_s), "Number of views in x_s must be equal to that of y_s."
        self.x_s, self.y_s, self.w_s = x_s, y_s, w_s

    def __len__(self):
        return len(self.x_s[0])

    def __getitem__(self, index):
        return ([v[index] for v in self.x_s], [v[index] for v in self.y_s], [v[index] for v in self.w_s])


def _as_batch_array(batch):
    """np.array(batch) of the reference (:167) without numpy trying to descend into the molecule objects:
    shape (batch_size, 3 = x / y / w, n_views), dtype object."""
    if isinstance(batch, np.ndarray) and batch.dtype == object and batch.ndim == 3:
        return batch
    n_views = len(batch[0][0])
    arr = np.empty((len(batch), 3, n_views), dtype=object)
    for b, item in enumerate(batch):
        for k in range(3):
            for v in range(n_views):
                arr[b, k, v] = item[k][v]
    return arr


def _protein_tensor(x_data, prot_desc_dict, cuda_prot):
    prots = [pair[1] for pair in x_data]
    prot_names = [prot.get_name() for prot in prots]
    prot_desc = np.array([prot_desc_dict[name] for name in prot_names])
    prot_desc = prot_desc.reshape(prot_desc.shape[0], prot_desc.shape[2])
    t = torch.from_numpy(np.ascontiguousarray(prot_desc)).float()
    return (cuda(t) if cuda_prot else t), prot_names


def process_ecfp_view_data(X, prot_desc_dict, idx, cuda_prot):
    """data.py:190-212 -> (mols_tensor [B, n_bits] float, prots_tensor [B, psc_dim] float, prot_names, x_data)."""
    x_data = X[:, 0, idx]
    mols_tensor = torch.from_numpy(np.array([pair[0].get_array() for pair in x_data]))
    prots_tensor, prot_names = _protein_tensor(x_data, prot_desc_dict, cuda_prot)
    return cuda(mols_tensor.float()), prots_tensor, prot_names, x_data


def process_gconv_view_data(X, prot_desc_dict, idx, cuda_prot):
    """data.py:272-306 -> (mol_data, prots_tensor, prot_names, x_data) with
    mol_data = [atoms int64 [N,75], deg_slice int64 [11,2], membership int64 [N], n_atoms_list, adj_1 .. adj_10 int32]."""
    x_data = X[:, 0, idx]
    mols = [pair[0] for pair in x_data]
    mol_data = gconv_input_list(mols, _device())
    prots_tensor, prot_names = _protein_tensor(x_data, prot_desc_dict, cuda_prot)
    return mol_data, prots_tensor, prot_names, x_data


def process_weave_view_data(X, prot_desc_dict, idx, cuda_prot):
    """data.py:215-269 -> (mol_data, prots_tensor, prot_names, x_data) with
    mol_data = [atom features float [N, F], pair features float [sum n^2, Fp], pair_split int32 [sum n^2] (the atom a pair
    belongs to), atom_to_pair int64 [sum n^2, 2], atom_split int32 [N] (the molecule of an atom), n_atoms_list]."""
    x_data = X[:, 0, idx]
    mols = [pair[0] for pair in x_data]
    n_atoms_list = [int(m.get_num_atoms()) for m in mols]
    atom_feat = np.concatenate([np.asarray(m.get_atom_features(), dtype=np.float32) for m in mols], axis=0)
    pair_feat = np.concatenate([np.asarray(m.get_pair_features(), dtype=np.float32).reshape(n * n, -1)
                                for m, n in zip(mols, n_atoms_list)], axis=0)
    atom_split = np.repeat(np.arange(len(mols), dtype=np.int32), n_atoms_list)
    starts = np.concatenate([[0], np.cumsum(n_atoms_list)[:-1]])
    a2p, psplit = [], []
    for n, st in zip(n_atoms_list, starts):
        r = np.repeat(np.arange(n), n) + st                       # C1.flatten(): the row atom of pair (i, j)
        c = np.tile(np.arange(n), n) + st                         # C0.flatten()
        a2p.append(np.stack([r, c], axis=1))
        psplit.append(r)
    atom_to_pair = np.concatenate(a2p, axis=0).astype(np.int64)
    pair_split = np.concatenate(psplit).astype(np.int32)
    mol_data = [cuda(torch.from_numpy(atom_feat)), cuda(torch.from_numpy(pair_feat)), cuda(torch.from_numpy(pair_split)),
                cuda(torch.from_numpy(atom_to_pair)), cuda(torch.from_numpy(atom_split)), n_atoms_list]
    prots_tensor, prot_names = _protein_tensor(x_data, prot_desc_dict, cuda_prot)
    return mol_data, prots_tensor, prot_names, x_data


def process_gnn_view_data(X, prot_desc_dict, idx, cuda_prot):
    """data.py:309-335 -> (mol_data, prots_tensor, prot_names, x_data) with
    mol_data = (fingerprints int64 [N], block-diagonal adjacency float [N, N], M float [N, 1], axis = atoms per molecule).
    The block-diagonal matrix, M and the ids are laid out in ONE pinned buffer and go up in one copy (the reference
    uploads every molecule's fingerprint tensor by itself and builds the block matrix with torch ops)."""
    x_data = X[:, 0, idx]
    mols = [pair[0] for pair in x_data]
    axis = [len(m.fingerprints) for m in mols]
    n = int(sum(axis))
    dev = _device()
    pin = dev.type == "cuda"
    buf = torch.empty(n * n + n + 2 * n, dtype=torch.float32, pin_memory=pin)     # A | M | ids (int64 = 2 floats each)
    a = buf[:n * n].view(n, n).numpy()
    a[:] = 0.0
    mv = buf[n * n:n * n + n].numpy()
    ids = buf[n * n + n:].view(torch.int64).numpy()
    o = 0
    for m, k in zip(mols, axis):
        a[o:o + k, o:o + k] = np.asarray(m.adjacency, dtype=np.float32)
        mv[o:o + k] = k
        ids[o:o + k] = np.asarray(m.fingerprints, dtype=np.int64)
        o += k
    d = buf.to(dev, non_blocking=True)
    mol_data = (d[n * n + n:].view(torch.int64), d[:n * n].view(n, n), d[n * n:n * n + n].view(n, 1), axis)
    prots_tensor, prot_names = _protein_tensor(x_data, prot_desc_dict, cuda_prot)
    return mol_data, prots_tensor, prot_names, x_data


def batch_collator(batch, prot_desc_dict, spec, cuda_prot=True):
    """data.py:166-187 -> (batch_size, {view: ((view data ...), y [B] object array, w [B] object array)}).
    ``spec``: a dict {view name: active?} (iteration order = view index in the batch) or a single view name.
    Views: ecfp4 / ecfp8 / gconv (the hot path), gnn and weave (section 8 f3)."""
    batch = _as_batch_array(batch)
    funcs = {"ecfp4": process_ecfp_view_data, "ecfp8": process_ecfp_view_data, "gconv": process_gconv_view_data,
             "gnn": process_gnn_view_data, "weave": process_weave_view_data}
    active_views = [k for k in spec if spec[k]] if isinstance(spec, dict) else [spec]
    data = {}
    for i, v_name in enumerate(active_views):
        if v_name not in funcs:
            raise NotImplementedError("view %r is not built (SURVEY.md section 8); have %s"
                                      % (v_name, sorted(funcs)))
        data[v_name] = (funcs[v_name](batch, prot_desc_dict, i, cuda_prot), batch[:, 1, i], batch[:, 2, i])
    return len(batch), data
