"""Batch-graph builder: same public surface and the same integer layout as the reference's
/root/reference/jova/feat/mol_graphs.py (ConvMol :47-270, ConvMol.agglomerate_mols :272-383, MultiConvMol :386-424),
written as flat numpy array passes instead of per-neighbour Python loops (the reference spends 15 ms per 128
molecules in `to_final_id`).

Layout contract (bit-exact with the reference, checked against tests/golden/index_*.npz):
  * per molecule: atoms stably sorted by degree; adjacency re-indexed into that order
  * batch: atoms ordered degree-major, molecule-minor; deg_slice int64 [11,2] = (running start, size) with NO
    zeroing of empty buckets' starts; membership[i] = molecule of final atom i; deg_adj_lists[d] int32 [n_d, d]
"""
import numpy as np

__all__ = ['ConvMol', 'MultiConvMol']


class ConvMol(object):
    """One molecule, atoms re-sorted by increasing degree (original order kept inside a degree)."""

    def __init__(self, atom_features, adj_list, smiles=None, max_deg=10, min_deg=0):
        if min_deg != 0:
            raise NotImplementedError("min_deg must be 0 (the reference's default, the only value its models use)")
        self.smiles = smiles
        self.max_deg = max_deg
        self.min_deg = min_deg
        self.n_atoms, self.n_feat = atom_features.shape
        n = self.n_atoms
        degree = np.fromiter((len(nb) for nb in adj_list), dtype=np.int64, count=n)
        if n and degree.max() > max_deg:
            raise IndexError("atom with %d neighbours exceeds max_deg=%d" % (int(degree.max()), max_deg))
        order = np.argsort(degree, kind='stable')                    # new position -> old atom
        new_of_old = np.empty(n, dtype=np.int64)
        new_of_old[order] = np.arange(n)
        self.atom_features = atom_features[order, :]
        self.deg_list = degree[order]
        old_ptr = np.concatenate([[0], np.cumsum(degree)])
        flat_old = np.fromiter((k for nb in adj_list for k in nb), dtype=np.int64, count=int(old_ptr[-1]))
        # neighbour lists in the new atom order, values remapped to new ids
        self.nbr_ptr = np.concatenate([[0], np.cumsum(self.deg_list)])
        take = (old_ptr[order][:, None] + np.arange(max_deg)[None, :])
        mask = np.arange(max_deg)[None, :] < self.deg_list[:, None]
        self.nbr_flat = new_of_old[flat_old[take[mask]]] if flat_old.size else np.zeros(0, dtype=np.int64)
        counts = np.bincount(self.deg_list, minlength=max_deg + 1).astype(np.int64)
        starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
        self.deg_start = np.concatenate([[0], np.cumsum(counts)])
        self.deg_block_indices = np.arange(n) - starts[self.deg_list]
        self.deg_id_list = self.deg_list - min_deg
        slice_ = np.zeros((max_deg + 1, 2), dtype=np.int32)
        slice_[:, 1] = counts
        slice_[:, 0] = starts * (counts != 0)                        # empty buckets start at 0 (mol_graphs.py:200)
        self.deg_slice = slice_
        self.membership = n * [0]

    # --- reference accessors -------------------------------------------------------------------------------
    @property
    def canon_adj_list(self):
        return [[int(v) for v in self.nbr_flat[self.nbr_ptr[i]:self.nbr_ptr[i + 1]]] for i in range(self.n_atoms)]

    @property
    def deg_adj_lists(self):
        out = []
        for d in range(self.min_deg, self.max_deg + 1):
            s, c = int(self.deg_start[d]), int(self.deg_slice[d, 1])
            if c == 0 or d == 0:
                out.append(np.zeros((c if d == 0 else 0, d), dtype=np.int32))
            else:
                out.append(self.nbr_flat[self.nbr_ptr[s]:self.nbr_ptr[s + c]].reshape(c, d))
        return out

    def __eq__(self, other):
        if isinstance(self, other.__class__):
            assert self.smiles is not None
            return self.smiles == other.smiles
        return False

    def __hash__(self):
        assert self.smiles is not None
        return hash(self.smiles)

    def get_smiles(self):
        return self.smiles

    def get_atoms_with_deg(self, deg):
        s, c = self.deg_slice[deg - self.min_deg]
        return self.atom_features[s:s + c, :]

    def get_num_atoms_with_deg(self, deg):
        return self.deg_slice[deg - self.min_deg, 1]

    def get_num_atoms(self):
        return self.n_atoms

    def get_atom_features(self):
        return self.atom_features

    def get_adjacency_list(self):
        return self.canon_adj_list

    def get_deg_adjacency_lists(self):
        return self.deg_adj_lists

    def get_deg_slice(self):
        return self.deg_slice

    @staticmethod
    def agglomerate_mols(mols, max_deg=10, min_deg=0):
        """Concatenate ConvMols into one batch graph (same result as the reference's agglomerate_mols)."""
        nb = max_deg + 1 - min_deg
        n_mols = len(mols)
        size = np.stack([m.deg_slice[:, 1].astype(np.int64) for m in mols], axis=1) if n_mols else \
            np.zeros((nb, 0), dtype=np.int64)                                          # [bucket, molecule]
        deg_sizes = size.sum(axis=1)
        deg_start = np.concatenate([[0], np.cumsum(deg_sizes)[:-1]])
        first_row = deg_start[:, None] + np.cumsum(size, axis=1) - size               # first final row of (b, k)
        n_per_mol = np.array([m.n_atoms for m in mols], dtype=np.int64)
        atom_base = np.concatenate([[0], np.cumsum(n_per_mol)])
        total = int(atom_base[-1])
        mol_of = np.repeat(np.arange(n_mols), n_per_mol)
        deg_of = np.concatenate([m.deg_list for m in mols]) if n_mols else np.zeros(0, dtype=np.int64)
        blk_of = np.concatenate([m.deg_block_indices for m in mols]) if n_mols else np.zeros(0, dtype=np.int64)
        final = first_row[deg_of, mol_of] + blk_of                                     # final row of every atom
        feats = np.concatenate([m.atom_features for m in mols], axis=0)
        nodes = np.empty_like(feats)
        nodes[final] = feats
        membership = np.empty(total, dtype=np.int64)
        membership[final] = mol_of
        # neighbours: local sorted ids -> position in the concatenation -> final rows
        nbr_local = np.concatenate([m.nbr_flat for m in mols]) if n_mols else np.zeros(0, dtype=np.int64)
        nbr_final = final[nbr_local + np.repeat(atom_base[:-1], [m.nbr_flat.size for m in mols])] \
            if nbr_local.size else np.zeros(0, dtype=np.int64)
        nbr_ptr = np.concatenate([[0], np.cumsum(deg_of)])
        deg_adj_lists = []
        for d in range(min_deg, max_deg + 1):
            b = d - min_deg
            rows = np.zeros((int(deg_sizes[b]), d), dtype=np.int32)
            if d > 0 and deg_sizes[b] > 0:
                sel = np.nonzero(deg_of == d)[0]
                src = nbr_ptr[sel][:, None] + np.arange(d)[None, :]
                rows[final[sel] - deg_start[b]] = nbr_final[src]
            deg_adj_lists.append(rows)
        deg_slice = np.stack([deg_start, deg_sizes], axis=1).astype(np.int64)
        return MultiConvMol(nodes, deg_adj_lists, deg_slice, membership.tolist(), n_mols,
                            smiles_list=[m.get_smiles() for m in mols])


class MultiConvMol(object):
    """Batch of molecules produced by ConvMol.agglomerate_mols."""

    def __init__(self, nodes, deg_adj_lists, deg_slice, membership, num_mols, smiles_list=None):
        self.nodes = nodes
        self.deg_adj_lists = deg_adj_lists
        self.deg_slice = deg_slice
        self.membership = membership
        self.num_mols = num_mols
        self.num_atoms = nodes.shape[0]
        self.smiles_list = smiles_list

    def __eq__(self, other):
        if isinstance(self, other.__class__):
            assert self.smiles_list is not None
            return self.smiles_list == other.smiles_list
        return False

    def __hash__(self):
        assert self.smiles_list is not None
        return hash(str(self.smiles_list))

    def get_deg_adjacency_lists(self):
        return self.deg_adj_lists

    def get_atom_features(self):
        return self.nodes

    def get_num_atoms(self):
        return self.num_atoms

    def get_num_molecules(self):
        return self.num_mols


class WeaveMol(object):
    """Molecule struct of the weave view (jova/feat/mol_graphs.py:427-462): atom features [n, F], pair features [n, n, Fp]."""

    def __init__(self, nodes, pairs, smiles=None):
        self.nodes = nodes
        self.pairs = pairs
        self.smiles = smiles
        self.num_atoms = self.nodes.shape[0]
        self.n_features = self.nodes.shape[1]

    def __eq__(self, other):
        if isinstance(self, other.__class__):
            assert self.smiles is not None
            return self.smiles == other.smiles
        return False

    def __hash__(self):
        assert self.smiles is not None
        return hash(self.smiles)

    def get_pair_features(self):
        return self.pairs

    def get_atom_features(self):
        return self.nodes

    def get_num_atoms(self):
        return self.num_atoms

    def get_num_features(self):
        return self.n_features
