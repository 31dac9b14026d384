"""Batch collation: counterpart of /root/reference/jova/data/data.py batch_collator :166-187,
process_ecfp_view_data :190-212 and process_gconv_view_data :272-306.

The reference builds, per step and in Python, the list ``[atoms int64 [N,75], deg_slice [11,2], membership [N],
n_atoms_lst, adj_1 .. adj_10]`` and moves each piece to the GPU separately.  Here the same list is produced by the
vectorised ConvMol.agglomerate_mols, packed into ONE pinned host buffer and uploaded with a single async copy; the
per-batch topology (symmetry flag, molecule offsets) is derived once on the device.
"""
import contextlib
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


def _native_handles(mol):
    """Host addresses of a ConvMol's degree-sorted arrays (cached on the molecule: they never change)."""
    h = getattr(mol, "_jova_native", None)
    if h is None:
        feats = np.ascontiguousarray(mol.atom_features, dtype=np.int64)
        deg = np.ascontiguousarray(mol.deg_list, dtype=np.int64)
        nbr = np.ascontiguousarray(mol.nbr_flat, dtype=np.int64)
        h = (feats, deg, nbr, feats.ctypes.data, deg.ctypes.data, nbr.ctypes.data, int(mol.n_atoms), int(nbr.size))
        mol._jova_native = h
    return h


class MolTable(object):
    """The degree-sorted arrays of a fixed set of ConvMols as flat address / size tables, so that a batch is one numpy
    fancy-index away from the arguments of jova_collate_gconv (no per-molecule Python in the training loop)."""

    def __init__(self, mols):
        self.mols = list(mols)
        hs = [_native_handles(m) for m in self.mols]
        self.ptrs = np.array([[h[3], h[4], h[5]] for h in hs], dtype=np.uint64).reshape(len(hs), 3)
        self.n_atoms = np.array([h[6] for h in hs], dtype=np.int32)
        self.n_edges = np.array([h[7] for h in hs], dtype=np.int64)
        self.n_feat = int(hs[0][0].shape[1]) if hs else 75

    def select(self, ids):
        ids = np.asarray(ids, dtype=np.int64)
        return np.ascontiguousarray(self.ptrs[ids].T), np.ascontiguousarray(self.n_atoms[ids]), \
            np.ascontiguousarray(self.n_edges[ids])


def gconv_input_list(mols, device, stream=None, ids=None, extras=()):
    """list of ConvMol (or a MolTable plus ``ids``) -> the reference's gconv input list with device tensors
    (data.py:291-296).

    The batch merge (ConvMol.agglomerate_mols, mol_graphs.py:272-383) runs in native code (jova_collate_gconv) and
    writes straight into ONE pinned host buffer that goes up with a single asynchronous copy (on ``stream`` when given):
    int64 region (atoms, deg_slice, membership) then 4-byte region (adjacency, molecule offsets, XCD-aware block order,
    then ``extras``: further float32 / int32 arrays of the batch, returned as device views in ``glist.extras``).
    The per-batch topology (symmetry, host copy of deg_slice, molecule offsets, block order) rides along."""
    if isinstance(mols, MolTable):
        ptrs, cnt, n_nbr = mols.select(ids)
        f = mols.n_feat
    else:
        hs = [_native_handles(m) for m in mols]
        ptrs = np.array([[h[3], h[4], h[5]] for h in hs], dtype=np.uint64).reshape(len(hs), 3).T.copy()
        cnt = np.array([h[6] for h in hs], dtype=np.int32)
        n_nbr = np.array([h[7] for h in hs], dtype=np.int64)
        f = int(hs[0][0].shape[1]) if hs else 75
    n_mols = int(cnt.size)
    e = int(n_nbr.sum())
    n = int(cnt.sum())
    n_atoms_lst = cnt.tolist()
    n_blocks = (n + 15) // 16
    n64 = n * f + 22 + n
    n32 = e + n_mols + 1 + n_blocks
    extras = [np.ascontiguousarray(x) for x in extras]
    assert all(x.dtype.itemsize == 4 for x in extras), "extras must be 4-byte arrays"
    # derived per-batch images the kernels want, made on the host and shipped in the same upload: the fp32 copy of the
    # atom features (rows padded to ld floats) and the reverse edge positions
    ld = F_._atoms_ld(f)
    with_derived = device.type == "cuda"
    o_f32 = (n32 + 3) // 4 * 4
    o_rev = o_f32 + (n * ld if with_derived else 0)
    n32 = o_rev + ((e + 3) // 4 if with_derived else 0)
    n32 = (n32 + 3) // 4 * 4
    x_off = [n32]
    for x in extras:
        x_off.append(x_off[-1] + (x.size + 3) // 4 * 4)          # every extra starts 16-byte aligned
    # (the 4-byte region starts on a 16-byte boundary too)
    pad64 = (-n64) % 2
    host = torch.empty((n64 + pad64) * 8 + x_off[-1] * 4, dtype=torch.uint8, pin_memory=(device.type == "cuda"))
    base = host.data_ptr()
    p_nodes, p_ds, p_memb = base, base + 8 * n * f, base + 8 * (n * f + 22)
    p_adj = base + 8 * (n64 + pad64)
    p_off, p_order = p_adj + 4 * e, p_adj + 4 * (e + n_mols + 1)
    sym = ctypes.c_int(1)
    V = ctypes.c_void_p
    if n_mols == 0:
        ptrs = np.zeros((3, 1), np.uint64)
    F_.call("jova_collate_gconv", n_mols, V(cnt.ctypes.data), V(ptrs[0].ctypes.data), V(ptrs[1].ctypes.data),
            V(ptrs[2].ctypes.data), V(n_nbr.ctypes.data), f, V(p_nodes), V(p_ds), V(p_memb), V(p_adj), V(p_off), V(p_order) if n_blocks else None,
            ctypes.byref(sym), V(p_adj + 4 * o_f32) if with_derived else None, ld,
            V(p_adj + 4 * o_rev) if with_derived else None)
    h32 = host[(n64 + pad64) * 8:].view(torch.int32).numpy()
    for x, o in zip(extras, x_off):
        h32[o:o + x.size] = x.reshape(-1).view(np.int32)
    ds_host = host[8 * n * f:8 * (n * f + 22)].view(torch.int64).numpy().reshape(11, 2).copy()
    with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
        dev = host.to(device, non_blocking=True)
        d64 = dev[:n64 * 8].view(torch.int64)
        d32 = dev[(n64 + pad64) * 8:].view(torch.int32)
        atoms = d64[:n * f].view(n, f)
        deg_slice = d64[n * f:n * f + 22].view(11, 2)
        memb = d64[n * f + 22:]
        glist = _GconvList([atoms, deg_slice, memb, n_atoms_lst])
        o = 0
        for d in range(1, 11):
            n_d = int(ds_host[d, 1])
            glist.append(d32[o:o + n_d * d].view(n_d, d))
            o += n_d * d
        glist.extras = [(d32[xo:xo + x.size].view(torch.float32) if x.dtype == np.float32 else d32[xo:xo + x.size])
                        .view(x.shape) for x, xo in zip(extras, x_off)]
        if device.type == "cuda":
            topo = F_.Topology(deg_slice, glist[4:], n_atoms_lst, n, sym_host=bool(sym.value))
            topo._mol_off = d32[o:o + n_mols + 1]
            topo._deg_slice_host = ds_host
            o += n_mols + 1
            topo._block_order = d32[o:o + n_blocks]
            topo._atoms_f32 = ((atoms.data_ptr(), tuple(atoms.shape)),
                               d32[o_f32:o_f32 + n * ld].view(torch.float32).view(n, ld), atoms)
            if sym.value:
                topo._rev_pos = d32[o_rev:o_rev + (e + 3) // 4].view(torch.uint8) if e else \
                    torch.zeros(1, dtype=torch.uint8, device=device)
            topo._keepalive = (host, dev)             # the pinned image outlives its asynchronous upload
            deg_slice._jova_topology = topo
    return glist


class _GconvList(list):
    """The gconv input list (a plain list for every consumer) that can also carry the batch's other uploaded arrays."""
    extras = ()


process_gconv_view_data = gconv_input_list            # (older name; the reference-shaped collators: jova_amd.data.data)


class DtiBatch(object):
    """One collated mini-batch, resident on the device."""

    def __init__(self, ecfp, glist, prot_names, y, w, prot_input=None, psc=None):
        self.ecfp, self.glist, self.prot_names, self.y, self.w = ecfp, glist, prot_names, y, w
        self.prot_input = prot_input
        self.psc = psc
        self.batch_size = ecfp.shape[0]

    def device_tensors(self):
        """Every device buffer the batch owns (BatchPrefetcher marks them as used by the consumer's stream)."""
        out = []
        topo = getattr(self.glist[1], "_jova_topology", None) if torch.is_tensor(self.glist[1]) else None
        if topo is not None:
            out += [topo._keepalive[1], topo.sym_flag]               # the one packed upload + the symmetry flag
        else:
            out += [t for t in self.glist if torch.is_tensor(t)]
        for t in (self.ecfp, self.y, self.w, self.psc) + tuple(self.prot_input or ()):
            if torch.is_tensor(t) and (topo is None or t.untyped_storage().data_ptr() !=
                                       topo._keepalive[1].untyped_storage().data_ptr()):
                out.append(t)
        return [t for t in out if t.is_cuda]

    def model_input(self):
        """X = compound views + protein views, as assembled at train_joint_attn.py:441-453: embedding-based protein
        views take the name list (element 2 of the collator's tuple), the ``psc`` view the descriptor tensor
        (element 1); order = c_views + p_views, here ecfp8, gconv, pcnn2d[, psc]."""
        prot = self.prot_input if self.prot_input is not None else self.prot_names
        x = [self.ecfp, (self.glist, self.batch_size), prot]
        if self.psc is not None:
            x.append(self.psc)
        return x


def collate_pairs(pool, conv_mols, mol_ids, prot_ids, y, w, device, prot2vec=None, with_psc=False, stream=None):
    """Synthetic-pool pairs -> DtiBatch.  ``prot2vec`` (optional module) pre-resolves protein names to table rows;
    ``with_psc`` adds the [B,8421] descriptor tensor of the ``psc`` view (process_ecfp_view_data, data.py:190-212).
    ``stream``: the upload is enqueued there (BatchPrefetcher's copy stream) instead of on the current stream.
    Everything of the batch -- graph index tensors, fingerprints, labels, weights, protein rows, descriptors -- travels in
    ONE pinned buffer and one asynchronous copy."""
    cached = getattr(pool, "_jova_mol_table", None)
    if cached is None or cached[0] is not conv_mols:
        cached = (conv_mols, MolTable(conv_mols))
        pool._jova_mol_table = cached
    table = cached[1]
    names = [("synthetic", pool.prot_names[i]) for i in prot_ids]
    extras = [pool.ecfp[mol_ids], np.asarray(y, dtype=np.float32), np.asarray(w, dtype=np.float32)]
    p_max = None
    if prot2vec is not None and device.type == "cuda":
        meta = prot2vec._resident_table(device).batch_meta(names)            # int32 [B, 2] = (first row, segments)
        extras += [np.ascontiguousarray(meta[:, 0]), np.ascontiguousarray(meta[:, 1])]
        p_max = int(meta[:, 1].max())
    if with_psc:
        extras.append(pool.psc[prot_ids])
    glist = gconv_input_list(table, device, stream, ids=mol_ids, extras=extras)
    ex = list(glist.extras)
    ecfp, yt, wt = ex[0], ex[1], ex[2]
    prot_input = (ex[3], ex[4], p_max) if p_max is not None else None
    psc = ex[-1] if with_psc else None
    return DtiBatch(ecfp, glist, names, yt, wt, prot_input, psc)


class BatchPrefetcher(object):
    """Collates batches on a background thread, ``depth`` batches ahead of the training loop.

    ``make_batch(k, stream)`` builds batch k and enqueues its host->device copies on ``stream`` (a side stream, so the
    copies overlap the kernels of the step that is running); the iterator makes torch's current stream wait for the
    batch's copy event before handing it out.  The native merge and the pinned-buffer fill release the GIL, so the
    training loop's Python keeps running.  Counterpart of nothing in the reference, which collates inline in the loop
    (proj/dti/train_joint_attn.py:412)."""

    def __init__(self, make_batch, n_batches, device, depth=2):
        import queue
        import threading
        self.make_batch, self.n, self.device = make_batch, int(n_batches), device
        self.q = queue.Queue(maxsize=max(1, depth))
        self.stream = torch.cuda.Stream(device=device) if device.type == "cuda" else None
        self.error = None
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def _run(self):
        try:
            if self.device.type == "cuda":
                torch.cuda.set_device(self.device)
            for k in range(self.n):
                b = self.make_batch(k, self.stream)
                ev = None
                if self.stream is not None:
                    ev = torch.cuda.Event()
                    ev.record(self.stream)
                self.q.put((b, ev))
        except BaseException as exc:          # surfaced on the consumer side
            self.error = exc
        self.q.put(None)

    def __iter__(self):
        while True:
            item = self.q.get()
            if item is None:
                if self.error is not None:
                    raise self.error
                return
            b, ev = item
            if ev is not None:
                cur = torch.cuda.current_stream()
                cur.wait_event(ev)
                # The buffers were allocated on the copy stream: tell the caching allocator that the consumer's stream
                # reads them, or a freed batch's memory is handed to the next upload while this step's kernels still
                # read its index tensors (garbage molecule offsets -> out-of-bounds writes).
                for t in b.device_tensors():
                    t.record_stream(cur)
            yield b
