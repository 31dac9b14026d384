"""Shared test helpers: golden-fixture access and oracle glue (CPU only)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def unpack_mols(z):
    """Inverse of tools/make_golden.py:pack_mols -> list of (features int64 [n,75], adjacency list of lists)."""
    feats = z["mol_feats"].astype(np.int64)
    aoff, nflat, noff = z["mol_atom_off"], z["mol_nbr_flat"], z["mol_nbr_off"]
    mols = []
    for m in range(len(aoff) - 1):
        a0, a1 = int(aoff[m]), int(aoff[m + 1])
        adj = [[int(v) for v in nflat[noff[a]:noff[a + 1]]] for a in range(a0, a1)]
        mols.append((feats[a0:a1], adj))
    return mols


def oracle_batch(mols):
    from oracle import mol_graphs as og
    return og.agglomerate([og.SortedMol(f, a) for f, a in mols])


def oracle_glist(mols, atoms=None):
    """Torch version of the data.py:291-296 input list built by the oracle."""
    from oracle import mol_graphs as og
    b = oracle_batch(mols)
    gl = og.gconv_input_list(b)
    out = [torch.from_numpy(gl[0]) if atoms is None else atoms, torch.from_numpy(gl[1]), torch.from_numpy(gl[2]), gl[3]]
    out += [torch.from_numpy(a) for a in gl[4:]]
    return out


def sd(z, prefix, dtype=None):
    d = {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}
    if dtype is not None:
        d = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in d.items()}
    return d


def profile(z):
    return {k[len("profile."):]: z[k] for k in z.files if k.startswith("profile.")}


def hp(z):
    return json.loads(str(z["hp"]))


def rel_err(a, b, floor=0.0):
    """max |a-b| / max(max|b|, floor)  (the 1e-5 'relative fp32' figure of BASELINE.json is read against the
    tensor's own scale).  ``floor`` is for gradient tensors that are analytically ~0 (e.g. the affine of a LayerNorm
    that feeds a BatchNorm): they are pure rounding noise in the reference too, so they are judged on the scale of
    the module's other gradients (callers pass floor = 5 % of the largest gradient tensor)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    denom = max(np.abs(b).max() if b.size else 0.0, floor, 1e-30)
    return float(np.abs(a - b).max() / denom) if b.size else 0.0


def grad_floor(z, prefix):
    """5 % of the largest gradient magnitude among fixture entries starting with prefix."""
    return 0.05 * max(float(np.abs(z[k]).max()) for k in z.files if k.startswith(prefix) and z[k].size)


def assert_close(a, b, tol=1e-5, what="", floor=0.0):
    if torch.is_tensor(a):
        a = a.detach().cpu().numpy()
    if torch.is_tensor(b):
        b = b.detach().cpu().numpy()
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, a.shape, b.shape)
    e = rel_err(a, b, floor)
    assert e <= tol, "%s: rel err %.3e > %.1e" % (what, e, tol)
