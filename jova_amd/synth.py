"""Seeded synthetic DTI inputs.

No RDKit and no datasets exist on the build or GPU boxes, so every test, golden
fixture and benchmark runs on synthetic molecules / proteins whose *shape*
follows the reference's featurisers:

* molecule  -> (atom_features int64 [n, 75], adjacency list-of-lists), the pair
  that `ConvMol.__init__` takes (reference jova/feat/mol_graphs.py:54); the 75
  columns mimic `atom_features` of jova/feat/graph_features.py:112-177
  (44 element one-hot + 11 degree + 7 valence + 2 raw ints + 5 hybridisation
  + 1 aromatic + 5 numH).
* protein   -> int ndarray [n_seg, 11] of 3-gram ids in [0, vocab) whose last
  row tail is padded with `vocab` (layout of proj/dti/build_prot_vocabs.py:38-75).

Everything is drawn from ``numpy.random.RandomState(seed)`` so a seed fully
determines a batch.  Pure numpy: usable by the oracle tooling, tests and bench.
"""
from __future__ import annotations

import numpy as np

N_ATOM_FEAT = 75
VOCAB = 9048
WINDOW = 11

_ONE_HOT_BLOCKS = (44, 11, 7)       # element, degree, implicit valence
_ONE_HOT_TAIL = (5, 1, 5)           # hybridisation, aromatic flag, numH


def make_molecule(rng, n_atoms, deg_cap=4, hub_frac=0.0, hub_max_deg=10):
    """Random connected molecular graph.

    Spanning tree grown atom by atom (parent drawn among earlier atoms that
    still have free valence), then floor(n/8) ring closures.  Adjacency is
    symmetric, has no self loops and no duplicate bonds.  With ``hub_frac`` > 0
    that share of atoms gets a target degree drawn from 5..hub_max_deg and new
    atoms attach preferentially to the first unfilled hub (config 4: every
    degree bucket 5..10 holds at least 1 % of the atoms).
    """
    n = int(n_atoms)
    cap = np.full(n, deg_cap, dtype=np.int64)
    is_hub = np.zeros(n, dtype=bool)
    if hub_frac > 0 and n > 12:
        n_hub = max(1, int(round(hub_frac * n)))
        hubs = rng.choice(n // 2, size=min(n_hub, n // 2), replace=False)
        cap[hubs] = rng.randint(5, hub_max_deg + 1, size=len(hubs))
        is_hub[hubs] = True
    adj = [[] for _ in range(n)]
    deg = np.zeros(n, dtype=np.int64)
    for i in range(1, n):
        free = (cap[:i] - deg[:i]).clip(min=0)
        open_hubs = np.nonzero(is_hub[:i] & (free > 0))[0]
        if len(open_hubs) and rng.uniform() < 0.85:
            j = int(open_hubs[0])      # fill hubs in order so every target degree is reached
        elif free.sum() == 0:          # everyone saturated: atom i stays isolated (degree 0), still a valid input
            continue
        else:
            j = int(rng.choice(i, p=free / free.sum()))
        adj[i].append(j)
        adj[j].append(i)
        deg[i] += 1
        deg[j] += 1
    for _ in range(n // 8):
        a, b = (int(v) for v in rng.randint(0, n, size=2))
        if a == b or b in adj[a] or deg[a] >= cap[a] or deg[b] >= cap[b]:
            continue
        adj[a].append(b)
        adj[b].append(a)
        deg[a] += 1
        deg[b] += 1
    feats = np.zeros((n, N_ATOM_FEAT), dtype=np.int64)
    col = 0
    feats[np.arange(n), col + rng.randint(0, 44, size=n)] = 1
    col += 44
    feats[np.arange(n), col + np.minimum(deg, 10)] = 1
    col += 11
    feats[np.arange(n), col + rng.randint(0, 7, size=n)] = 1
    col += 7
    feats[:, col] = rng.randint(-1, 2, size=n)          # formal charge
    feats[:, col + 1] = rng.randint(0, 2, size=n)       # radical electrons
    col += 2
    feats[np.arange(n), col + rng.randint(0, 5, size=n)] = 1
    col += 5
    feats[:, col] = rng.randint(0, 2, size=n)
    col += 1
    feats[np.arange(n), col + rng.randint(0, 5, size=n)] = 1
    return feats, adj


def make_protein(rng, n_seg, vocab=VOCAB, window=WINDOW):
    prof = rng.randint(0, vocab, size=(int(n_seg), window)).astype(np.int64)
    n_valid_last = int(rng.randint(1, window + 1))
    prof[-1, n_valid_last:] = vocab
    return prof


def _clip_normal(rng, mean, std, lo, hi, size):
    return np.clip(np.rint(rng.normal(mean, std, size=size)), lo, hi).astype(np.int64)


def _clip_lognormal(rng, mu, sigma, lo, hi, size):
    return np.clip(np.rint(rng.lognormal(mu, sigma, size=size)), lo, hi).astype(np.int64)


SHAPES = {
    # name: (n_mols, n_prots, atoms sampler, deg_cap, hub_frac, segment sampler)
    "davis": dict(n_mols=72, n_prots=442,
                  atoms=lambda r, k: _clip_normal(r, 32, 6, 19, 46, k),
                  segs=lambda r, k: _clip_lognormal(r, 4.1, 0.5, 20, 232, k),
                  deg_cap=4, hub_frac=0.0),
    "kiba": dict(n_mols=2100, n_prots=230,
                 atoms=lambda r, k: _clip_normal(r, 28, 8, 10, 70, k),
                 segs=lambda r, k: _clip_lognormal(r, 4.1, 0.5, 20, 376, k),
                 deg_cap=4, hub_frac=0.0),
    "metz": dict(n_mols=240, n_prots=170,
                 atoms=lambda r, k: _clip_normal(r, 27, 6, 10, 56, k),
                 segs=lambda r, k: _clip_lognormal(r, 4.3, 0.6, 20, 376, k),
                 deg_cap=4, hub_frac=0.0),
    # config 5 stress variant: long protein sequences, P_max = 1024 segments (protein 0 is pinned to exactly 1024)
    "long1024": dict(n_mols=240, n_prots=48,
                     atoms=lambda r, k: _clip_normal(r, 27, 6, 10, 56, k),
                     segs=lambda r, k: np.concatenate([[1024], _clip_lognormal(r, 5.6, 0.8, 20, 1024, k - 1)]),
                     deg_cap=4, hub_frac=0.0),
    "graph4k": dict(n_mols=4096, n_prots=1,
                    atoms=lambda r, k: r.randint(20, 101, size=k).astype(np.int64),
                    segs=lambda r, k: np.full(k, 20, dtype=np.int64),
                    deg_cap=4, hub_frac=0.12),
}


class SyntheticPool(object):
    """A pool of synthetic drugs and targets of one dataset shape."""

    def __init__(self, shape="davis", seed=1, n_mols=None, n_prots=None, vocab=VOCAB):
        spec = SHAPES[shape]
        rng = np.random.RandomState(seed)
        self.shape = shape
        self.vocab = vocab
        n_mols = spec["n_mols"] if n_mols is None else n_mols
        n_prots = spec["n_prots"] if n_prots is None else n_prots
        self.mols = [make_molecule(rng, n, spec["deg_cap"], spec["hub_frac"])
                     for n in spec["atoms"](rng, n_mols)]
        self.prot_names = ["P%05d" % i for i in range(n_prots)]
        self.protein_profile = {name: make_protein(rng, s, vocab)
                                for name, s in zip(self.prot_names, spec["segs"](rng, n_prots))}
        self.ecfp = (rng.uniform(size=(n_mols, 1024)) < 0.06).astype(np.float32)
        self._psc = None
        self._psc_seed = seed

    @property
    def psc(self):
        """[n_prots, 8421] float32 protein-sequence-composition descriptors (the ``psc`` view; the real tables are
        z-scored composition counts, data/*/prot_desc.csv): seeded N(0,1), made on first use, from a stream of its own
        so that pools of one seed are identical whether or not the view is used."""
        if self._psc is None:
            rng = np.random.RandomState(self._psc_seed + 7919)
            self._psc = rng.normal(size=(len(self.prot_names), 8421)).astype(np.float32)
        return self._psc

    def epoch_pairs(self, n_pairs, seed=0):
        """A fixed training set of ``n_pairs`` (drug, target, label) triples over this pool (KIBA-shaped: ~118 k pairs over
        ~2 100 drugs x ~230 targets) -> (mol_ids, prot_ids, y [n,1], w [n,1]); sharded by
        jova_amd.data.sampler.ShardSampler."""
        rng = np.random.RandomState(seed)
        mi = rng.randint(0, len(self.mols), size=n_pairs)
        pi = rng.randint(0, len(self.prot_names), size=n_pairs)
        return mi, pi, rng.normal(size=(n_pairs, 1)), np.ones((n_pairs, 1))

    def sample_pairs(self, rng, batch_size):
        """-> (mol_ids [B], prot_ids [B], y [B,1] float64, w [B,1] float64)."""
        mi = rng.randint(0, len(self.mols), size=batch_size)
        pi = rng.randint(0, len(self.prot_names), size=batch_size)
        y = rng.normal(size=(batch_size, 1))
        w = np.ones((batch_size, 1))
        return mi, pi, y, w
