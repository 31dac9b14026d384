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
import os

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


def process_gnn_view_data(X, prot_desc_dict, idx, cuda_prot, dense=None):
    """data.py:309-335 -> (mol_data, prots_tensor, prot_names, x_data) with
    mol_data = (fingerprints int64 [N], block-diagonal adjacency, M float [N, 1], axis = atoms per molecule).

    The adjacency: the reference builds the dense block-diagonal float matrix [N, N] (67 MB at N = 4.1 k, 99.9 % zeros)
    and multiplies by it in every layer.  ``dense=False`` (the default on the GPU) hands the model a
    nn.functional.BlockAdjacency instead: the non-zeros of every molecule's matrix in CSR form (A and A^T), which
    GraphNeuralNet multiplies with jova_spmm_csr; ``.to_dense()`` gives the reference's tensor.  ``dense=True`` (the
    default on the CPU, JOVA_GNN_DENSE=1 anywhere) is the reference's layout: block matrix, M and ids laid out in ONE pinned
    buffer that goes up in one copy (the reference uploads every molecule's fingerprint tensor by itself and builds the block
    matrix with torch ops)."""
    x_data = X[:, 0, idx]
    mols = [pair[0] for pair in x_data]
    axis = [len(m.fingerprints) for m in mols]
    n = int(sum(axis))
    dev = _device()
    pin = dev.type == "cuda"
    if dense is None:
        dense = dev.type != "cuda" or bool(os.environ.get("JOVA_GNN_DENSE"))
    if not dense:
        from ..nn.functional import BlockAdjacency
        rows, cols, vals, o = [], [], [], 0
        for m, k in zip(mols, axis):
            a = np.asarray(m.adjacency, dtype=np.float32)
            r, c = np.nonzero(a)
            rows.append(r + o)
            cols.append(c + o)
            vals.append(a[r, c])
            o += k
        rows = np.concatenate(rows) if rows else np.zeros(0, np.int64)
        cols = np.concatenate(cols) if cols else np.zeros(0, np.int64)
        vals = np.concatenate(vals).astype(np.float32) if vals else np.zeros(0, np.float32)
        nnz = int(rows.size)

        def csr(r, c, v):                                   # r sorted or not: a stable sort by row keeps column order
            order = np.argsort(r, kind="stable")
            ptr = np.zeros(n + 1, np.int32)
            np.cumsum(np.bincount(r, minlength=n), out=ptr[1:])
            return ptr, c[order].astype(np.int32), v[order]
        p0, c0, v0 = csr(rows, cols, vals)
        p1, c1, v1 = csr(cols, rows, vals)
        # one pinned image: ptr | col | val | ptr_t | col_t | val_t | M | ids (int64 = two 4-byte words each)
        words = 2 * (n + 1) + 4 * nnz + n + (n & 1)
        buf = torch.empty(words + 2 * n, dtype=torch.int32, pin_memory=pin)
        hb = buf.numpy()
        offs, o = [], 0
        for arr in (p0, c0, v0, p1, c1, v1):
            hb[o:o + arr.size] = arr.view(np.int32)
            offs.append((o, arr.size))
            o += arr.size
        o_m = o
        mv = np.repeat(np.asarray(axis, dtype=np.float32), axis)
        hb[o_m:o_m + n] = mv.view(np.int32)
        ids = buf[words:].view(torch.int64).numpy()
        ids[:] = np.concatenate([np.asarray(m.fingerprints, dtype=np.int64) for m in mols]) if mols else 0
        d = buf.to(dev, non_blocking=True)
        part = [d[a:a + k] for a, k in offs]
        f32 = lambda t: t.view(torch.float32)              # noqa: E731
        adj = BlockAdjacency(n, part[0], part[1], f32(part[2]), part[3], part[4], f32(part[5]))
        adj._keepalive = (buf, d)
        mol_data = (d[words:].view(torch.int64), adj, f32(d[o_m:o_m + n]).view(n, 1), axis)
        prots_tensor, prot_names = _protein_tensor(x_data, prot_desc_dict, cuda_prot)
        return mol_data, prots_tensor, prot_names, x_data
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
