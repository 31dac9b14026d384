"""Batch collation: counterpart of /root/reference/jova/data/data.py batch_collator :166-187,
process_ecfp_view_data :190-212 and process_gconv_view_data :272-306.

The reference builds, per step and in Python, the list ``[atoms int64 [N,75], deg_slice [11,2], membership [N],
n_atoms_lst, adj_1 .. adj_10]`` and moves each piece to the GPU separately.  Here the same list is produced by the
vectorised ConvMol.agglomerate_mols, packed into ONE pinned host buffer and uploaded with a single async copy; the
per-batch topology (symmetry flag, molecule offsets) is derived once on the device.
"""
import ctypes

import numpy as np
import torch

from ..feat.mol_graphs import ConvMol
from ..nn import functional as F_


def adjacency_is_symmetric(deg_slice, adj_lists, n_atoms):
    """Host restatement of jova_adj_symmetry: every neighbour index in range, no self loops, no duplicate neighbours,
    and i in nbr(j) whenever j in nbr(i).  The collator has the lists on the host anyway; handing the answer to the
    Topology spares the training step its one device->host read per batch."""
    src_parts, dst_parts = [], []
    for d, a in enumerate(adj_lists, start=1):
        a = np.asarray(a)
        if a.size == 0:
            continue
        start = int(deg_slice[d][0])
        src_parts.append(np.repeat(np.arange(start, start + a.shape[0], dtype=np.int64), d))
        dst_parts.append(a.reshape(-1).astype(np.int64))
    if not src_parts:
        return True
    src, dst = np.concatenate(src_parts), np.concatenate(dst_parts)
    if (dst < 0).any() or (dst >= n_atoms).any() or (src == dst).any():
        return False
    fwd = np.sort(src * n_atoms + dst)
    if (fwd[1:] == fwd[:-1]).any():
        return False
    return bool(np.array_equal(fwd, np.sort(dst * n_atoms + src)))


def process_gconv_view_data(mols, device):
    """list of ConvMol -> the reference's gconv input list with device tensors (data.py:291-296)."""
    mm = ConvMol.agglomerate_mols(mols)
    n_atoms_lst = [m.get_num_atoms() for m in mols]
    adj = mm.get_deg_adjacency_lists()[1:]
    membership = np.asarray(mm.membership, dtype=np.int64)
    # one packed upload: int64 region (atoms, deg_slice, membership) then int32 region (adjacency, molecule offsets)
    mol_off = np.concatenate([[0], np.cumsum(n_atoms_lst)]).astype(np.int32)
    i64_parts = [mm.nodes.astype(np.int64, copy=False).reshape(-1), mm.deg_slice.reshape(-1), membership]
    ds_host = np.ascontiguousarray(mm.deg_slice, dtype=np.int64).copy()
    n_blocks = (int(mm.nodes.shape[0]) + 15) // 16
    order = np.zeros(n_blocks, dtype=np.int32)                   # XCD-aware block order of the gather kernels
    if device.type == "cuda" and n_blocks:
        F_.call("jova_block_order", ctypes.c_void_p(ds_host.ctypes.data), int(mm.nodes.shape[0]),
                ctypes.c_void_p(order.ctypes.data))
    i32_parts = [a.reshape(-1) for a in adj] + [mol_off, order]
    n64 = sum(p.size for p in i64_parts)
    n32 = sum(p.size for p in i32_parts)
    host = torch.empty(n64 * 8 + n32 * 4, dtype=torch.uint8, pin_memory=torch.cuda.is_available())
    h64 = host[:n64 * 8].view(torch.int64).numpy()
    h32 = host[n64 * 8:].view(torch.int32).numpy()
    o = 0
    for p in i64_parts:
        h64[o:o + p.size] = p
        o += p.size
    o = 0
    for p in i32_parts:
        h32[o:o + p.size] = p
        o += p.size
    dev = host.to(device, non_blocking=True)
    d64 = dev[:n64 * 8].view(torch.int64)
    d32 = dev[n64 * 8:].view(torch.int32)
    n, f = mm.nodes.shape
    atoms = d64[:n * f].view(n, f)
    deg_slice = d64[n * f:n * f + 22].view(11, 2)
    memb = d64[n * f + 22:]
    glist = [atoms, deg_slice, memb, n_atoms_lst]
    o = 0
    for d, a in enumerate(adj, start=1):
        glist.append(d32[o:o + a.size].view(a.shape[0], d))
        o += a.size
    if device.type == "cuda":
        topo = F_.Topology(deg_slice, glist[4:], n_atoms_lst, n)
        topo._mol_off = d32[o:o + len(n_atoms_lst) + 1]
        topo._deg_slice_host = ds_host
        topo._sym_host = adjacency_is_symmetric(ds_host, adj, n)
        o += len(n_atoms_lst) + 1
        topo._block_order = d32[o:o + n_blocks]
        deg_slice._jova_topology = topo
    return glist


class DtiBatch(object):
    """One collated mini-batch, resident on the device."""

    def __init__(self, ecfp, glist, prot_names, y, w, prot_input=None, psc=None):
        self.ecfp, self.glist, self.prot_names, self.y, self.w = ecfp, glist, prot_names, y, w
        self.prot_input = prot_input
        self.psc = psc
        self.batch_size = ecfp.shape[0]

    def model_input(self):
        """X = compound views + protein views, as assembled at train_joint_attn.py:441-453: embedding-based protein
        views take the name list (element 2 of the collator's tuple), the ``psc`` view the descriptor tensor
        (element 1); order = c_views + p_views, here ecfp8, gconv, pcnn2d[, psc]."""
        prot = self.prot_input if self.prot_input is not None else self.prot_names
        x = [self.ecfp, (self.glist, self.batch_size), prot]
        if self.psc is not None:
            x.append(self.psc)
        return x


def collate_pairs(pool, conv_mols, mol_ids, prot_ids, y, w, device, prot2vec=None, with_psc=False):
    """Synthetic-pool pairs -> DtiBatch.  ``prot2vec`` (optional module) pre-resolves protein names to table rows;
    ``with_psc`` adds the [B,8421] descriptor tensor of the ``psc`` view (process_ecfp_view_data, data.py:190-212)."""
    glist = process_gconv_view_data([conv_mols[i] for i in mol_ids], device)
    ecfp = torch.from_numpy(pool.ecfp[mol_ids]).to(device, non_blocking=True)
    names = [("synthetic", pool.prot_names[i]) for i in prot_ids]
    yt = torch.from_numpy(np.asarray(y, dtype=np.float32)).to(device, non_blocking=True)
    wt = torch.from_numpy(np.asarray(w, dtype=np.float32)).to(device, non_blocking=True)
    prot_input = None
    if prot2vec is not None and device.type == "cuda":
        prot_input = prot2vec._resident_table(device).batch(names, device)
    psc = torch.from_numpy(pool.psc[prot_ids]).to(device, non_blocking=True) if with_psc else None
    return DtiBatch(ecfp, glist, names, yt, wt, prot_input, psc)
