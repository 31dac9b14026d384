"""ORACLE (test infrastructure only) -- integer layout of batched molecular graphs.

CPU restatement, in plain Python/numpy loops, of how the reference orders atoms
and re-indexes adjacency when it builds a graph-convolution batch.  It follows

* ``ConvMol.__init__`` / ``ConvMol._deg_sort``   /root/reference/jova/feat/mol_graphs.py:54-201
* ``ConvMol.agglomerate_mols``                  /root/reference/jova/feat/mol_graphs.py:272-383
* ``process_gconv_view_data`` (input list)      /root/reference/jova/data/data.py:272-306
* ``_construct_embedding_indices``              /root/reference/jova/nn/models.py:214-242

Pinned against golden vectors exported from the imported reference
(tests/golden/index_*.npz, made by tools/make_golden.py).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product path (jova_amd/) never does.
"""
import numpy as np

MAX_DEG = 10
MIN_DEG = 0


class SortedMol(object):
    """One molecule after the per-molecule degree sort (mol_graphs.py:132-201, :82-100)."""

    def __init__(self, atom_features, adj_list, max_deg=MAX_DEG, min_deg=MIN_DEG):
        n = atom_features.shape[0]
        degree = [len(nbrs) for nbrs in adj_list]
        # stable sort of atoms by degree: position p of the new order holds old atom order[p]
        order = sorted(range(n), key=lambda i: (degree[i], i))          # == np.lexsort((old, deg)) :140
        new_of_old = {old: new for new, old in enumerate(order)}          # :154
        self.n_atoms = n
        self.atom_features = atom_features[order, :]                       # :145
        self.deg_list = [degree[i] for i in order]                        # :148
        self.canon_adj = [[new_of_old[k] for k in adj_list[i]] for i in order]   # :157-159
        n_buckets = max_deg + 1 - min_deg
        self.deg_adj_lists = []
        self.deg_slice = np.zeros((n_buckets, 2), dtype=np.int32)          # :184
        for deg in range(min_deg, max_deg + 1):
            rows = [self.canon_adj[i] for i in range(n) if self.deg_list[i] == deg]
            if rows:
                self.deg_adj_lists.append(np.array(rows).reshape(len(rows), deg))   # :176
            else:
                self.deg_adj_lists.append(np.zeros((0, deg), dtype=np.int32))       # :180
            b = deg - min_deg
            self.deg_slice[b, 1] = len(rows)
            if deg > min_deg:
                self.deg_slice[b, 0] = self.deg_slice[b - 1, 0] + self.deg_slice[b - 1, 1]   # :194-197
        self.deg_slice[:, 0] *= (self.deg_slice[:, 1] != 0)                # empty buckets start at 0  :200
        self.deg_id_list = np.array(self.deg_list) - min_deg               # :83
        deg_start = np.concatenate([[0], np.cumsum(self.deg_slice[:, 1])])  # :92
        # index of every atom inside its own degree bucket                  :95-100
        self.deg_block_indices = np.array(
            [i - deg_start[self.deg_list[i]] for i in range(n)], dtype=np.int64)

    def atoms_with_deg(self, b):
        s, c = self.deg_slice[b]
        return self.atom_features[s:s + c, :]                              # :119-123


def agglomerate(mols, max_deg=MAX_DEG, min_deg=MIN_DEG):
    """Merge SortedMols into one batch graph (mol_graphs.py:272-383).

    Returns dict(nodes [N,F], deg_slice int64 [11,2], membership int64 [N],
    deg_adj_lists list of int32 [n_d, d] for d = min_deg..max_deg, n_atoms list).
    Atom order is degree-major, molecule-minor; empty buckets keep their
    running start (no zeroing here, unlike the per-molecule slice).
    """
    n_buckets = max_deg + 1 - min_deg
    n_mols = len(mols)
    blocks = [m.atoms_with_deg(b) for b in range(n_buckets) for m in mols]           # :290-293
    nodes = np.vstack(blocks)                                                          # :296
    size = [[int(m.deg_slice[b, 1]) for m in mols] for b in range(n_buckets)]        # :300-301
    deg_sizes = [sum(row) for row in size]                                             # :304
    deg_start = [0] * n_buckets
    for b in range(1, n_buckets):
        deg_start[b] = deg_start[b - 1] + deg_sizes[b - 1]                             # :308
    deg_slice = np.array(list(zip(deg_start, deg_sizes)), dtype=np.int64)              # :314
    membership = [k for b in range(n_buckets) for k in range(n_mols) for _ in range(size[b][k])]   # :317-321
    # first final row of (bucket b, molecule k)                                       # :327, :340
    first_row = [[0] * n_mols for _ in range(n_buckets)]
    for b in range(n_buckets):
        run = deg_start[b]
        for k in range(n_mols):
            first_row[b][k] = run
            run += size[b][k]
    deg_adj_lists = [np.zeros((deg_sizes[b], b + min_deg), dtype=np.int32) for b in range(n_buckets)]   # :355-358
    for b in range(n_buckets):
        row = 0
        for k, m in enumerate(mols):
            local = m.deg_adj_lists[b]
            for i in range(local.shape[0]):
                for j in range(local.shape[1]):
                    a = int(local[i, j])                                               # atom id inside molecule k
                    final = first_row[m.deg_id_list[a]][k] + m.deg_block_indices[a]   # to_final_id :342-352
                    deg_adj_lists[b][row, j] = final
                row += 1
    return dict(nodes=nodes, deg_slice=deg_slice,
                membership=np.array(membership, dtype=np.int64),
                deg_adj_lists=deg_adj_lists,
                n_atoms=[m.n_atoms for m in mols])


def gconv_input_list(batch):
    """Model input list of data.py:291-296 (numpy arrays instead of device tensors):
    [atoms, deg_slice, membership, n_atoms_lst, adj_1 .. adj_10]."""
    return [batch["nodes"], batch["deg_slice"], batch["membership"], list(batch["n_atoms"])] + \
        list(batch["deg_adj_lists"][1:])


def embedding_indices(prot_names, protein_profile, fill_val):
    """Padded 3-gram index cube [B, P_max, window] (models.py:233-242)."""
    rows = [np.asarray(protein_profile[name], dtype=np.int64) for name in prot_names]
    p_max = max(r.shape[0] for r in rows)
    out = np.full((len(rows), p_max, rows[0].shape[1]), fill_val, dtype=np.int64)
    for b, r in enumerate(rows):
        out[b, :r.shape[0], :] = r
    return out
