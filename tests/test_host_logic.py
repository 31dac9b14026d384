"""CPU tests of the product's host side: batch-graph builder vs golden vectors and vs the oracle, the C-ABI library
(loads, exports every symbol the header declares, signatures agree with the binding), module/state-dict surface."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

import helpers as H
from jova_amd import synth
from jova_amd.feat.mol_graphs import ConvMol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["toy4", "deg0_b16", "hubs_b8", "pair_b2", "davis_b128"])
def test_agglomerate_bit_exact_vs_reference_golden(name):
    z = H.load("index_" + name)
    mols = [ConvMol(f, a, smiles="m%d" % i) for i, (f, a) in enumerate(H.unpack_mols(z))]
    mm = ConvMol.agglomerate_mols(mols)
    assert mm.nodes.dtype == np.int64 and np.array_equal(mm.nodes, z["nodes"].astype(np.int64))
    assert mm.deg_slice.dtype == np.int64 and np.array_equal(mm.deg_slice, z["deg_slice"])
    assert np.array_equal(np.asarray(mm.membership), z["membership"])
    for d in range(11):
        got = mm.get_deg_adjacency_lists()[d]
        assert got.dtype == np.int32 and got.shape == z["adj_%d" % d].shape
        assert np.array_equal(got, z["adj_%d" % d])


@pytest.mark.parametrize("seed,shape,count", [(1, "davis", 64), (8, "graph4k", 48), (64, "kiba", 100)])
def test_agglomerate_matches_oracle_on_fresh_seeds(seed, shape, count):
    """Bigger seeded batches than the committed goldens: product builder vs the loop-faithful oracle."""
    from oracle import mol_graphs as og
    r = np.random.RandomState(seed)
    spec = synth.SHAPES[shape]
    raw = [synth.make_molecule(r, n, spec["deg_cap"], spec["hub_frac"]) for n in spec["atoms"](r, count)]
    if shape == "davis":
        raw += [synth.make_molecule(r, 1), synth.make_molecule(r, 2)]      # degree-0 atom, 2-atom molecule
    want = og.agglomerate([og.SortedMol(f, a) for f, a in raw])
    cms = [ConvMol(f, a, smiles=str(i)) for i, (f, a) in enumerate(raw)]
    mm = ConvMol.agglomerate_mols(cms)
    assert np.array_equal(mm.nodes, want["nodes"])
    assert np.array_equal(mm.deg_slice, want["deg_slice"])
    assert np.array_equal(np.asarray(mm.membership), want["membership"])
    for d in range(11):
        assert np.array_equal(mm.deg_adj_lists[d], want["deg_adj_lists"][d])
    # per-molecule surface used by the reference's collators
    for cm, (f, a) in zip(cms[:8], raw[:8]):
        sm = og.SortedMol(f, a)
        assert np.array_equal(cm.get_atom_features(), sm.atom_features)
        assert cm.get_adjacency_list() == sm.canon_adj
        assert np.array_equal(cm.deg_slice, sm.deg_slice)
        for d in range(11):
            assert np.array_equal(cm.deg_adj_lists[d], sm.deg_adj_lists[d])


def test_degree_overflow_raises():
    f = np.zeros((13, 75), dtype=np.int64)
    adj = [list(range(1, 13))] + [[0]] * 12          # centre atom with 12 neighbours > max_deg
    with pytest.raises(IndexError):
        ConvMol(f, adj)


def _header_decls():
    text = open(os.path.join(ROOT, "include", "jova_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(int|size_t|const char\*|void|int64_t)\s+(jova_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        args = m.group(3).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        decls[m.group(2)] = n
    return decls


def test_abi_library_exports_every_declared_symbol():
    from jova_amd import _lib
    decls = _header_decls()
    assert len(decls) >= 30
    L = _lib.lib()
    for name in decls:
        assert hasattr(L, name), "libjova_hip.so does not export %s" % name
    assert sorted(decls) == _lib.EXPORTS
    assert L.jova_abi_version() == 1
    for name, sig in _lib._SIGS.items():
        assert len(sig) == decls[name], "%s: binding has %d args, header %d" % (name, len(sig), decls[name])
    assert isinstance(L.jova_nonsat_workspace_bytes(ctypes.c_int64(1000)), int)


def test_abi_argument_errors_are_reported_without_a_gpu():
    from jova_amd import _lib
    with pytest.raises(RuntimeError, match="f_in"):
        _lib.call("jova_graphconv_fwd", None, 0, 500, 10, 500, 64, None, None, None, None, 1, None, None, None, None, None)
    with pytest.raises(RuntimeError, match="unit stride"):
        _lib.call("jova_gemm_f32", ctypes.c_void_p(16), 3, 3, ctypes.c_void_p(16), 1, 4, ctypes.c_void_p(16), 4, 4, 4,
                  4, 1.0, None, None, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, None)


def test_abi_accepts_an_empty_batch_without_a_gpu():
    """n_atoms == 0: the graph entry points return OK before any HIP call; per-atom pointers may be null."""
    from jova_amd import _lib
    tok = ctypes.c_void_p(16)                                    # non-null stand-in for the per-batch tables (never read)
    _lib.call("jova_graphconv_fwd", None, 0, 64, 0, 64, 64, tok, tok, tok, tok, 1, None, None, tok, None, None)
    _lib.call("jova_graphpool_fwd", None, 0, 64, tok, tok, None, None, None, None, None)
    _lib.call("jova_graphpool_bwd", None, None, 0, 64, tok, tok, None, None, 0, None, None, None, None)
    _lib.call("jova_row_table", 0, tok, tok, None, None, None, None)
    with pytest.raises(RuntimeError, match="null pointer"):     # ... but not when there are atoms
        _lib.call("jova_graphpool_fwd", None, 5, 64, tok, tok, None, None, None, None, None)


def test_block_order_is_a_permutation_with_one_eighth_of_every_bucket_per_xcd():
    """jova_block_order (host-only): a permutation of the 16-row blocks; workgroup b (XCD b % 8) gets blocks whose
    relative position inside their degree bucket lies in the (b % 8)-th eighth."""
    from jova_amd import _lib
    ds = np.array([[0, 0], [0, 1000], [1000, 3000], [4000, 2500], [6500, 700]] + [[7200, 0]] * 6, dtype=np.int64)
    for n in (7200, 7193, 16, 1):
        d = ds.copy()
        if n < 7200:
            d[4, 1] -= 7200 - n if n > 6500 else 0
        if n <= 16:
            d = np.array([[0, 0], [0, n]] + [[n, 0]] * 9, dtype=np.int64)
        nb = (n + 15) // 16
        o = np.full(nb, -1, dtype=np.int32)
        _lib.call("jova_block_order", ctypes.c_void_p(d.ctypes.data), n, ctypes.c_void_p(o.ctypes.data))
        assert sorted(o.tolist()) == list(range(nb))
        if n == 7200:
            for b in range(nb):
                row0 = 16 * int(o[b])
                k = [i for i in range(11) if d[i, 0] <= row0 < d[i, 0] + d[i, 1]][0]
                rel = (row0 - d[k, 0]) / float(d[k, 1])
                assert (b % 8) / 8.0 - 0.02 <= rel <= (b % 8 + 1) / 8.0 + 0.02, (b, row0, rel)


def test_host_symmetry_check_matches_a_brute_force_restatement():
    """collate.adjacency_is_symmetric (what the collator tells the Topology instead of a device->host read) against a
    plain-Python restatement of the device kernel's rule, on the golden batch graph and on broken variants of it."""
    from jova_amd.data.collate import adjacency_is_symmetric
    z = H.load("index_hubs_b8")                                  # reference-made batch graph with degrees up to 10
    ds = np.asarray(z["deg_slice"], dtype=np.int64)
    adj = [np.asarray(z["adj_%d" % d], dtype=np.int32).reshape(-1, d) for d in range(1, 11)]
    n = int(z["nodes"].shape[0])

    def brute(adj_lists):
        nbr = {}
        for d, a in enumerate(adj_lists, start=1):
            for r in range(a.shape[0]):
                nbr[int(ds[d, 0]) + r] = [int(v) for v in a[r]]
        for i, lst in nbr.items():
            if len(set(lst)) != len(lst) or i in lst or any(j < 0 or j >= n for j in lst):
                return False
            if any(i not in nbr.get(j, []) for j in lst):
                return False
        return True

    assert brute(adj) and adjacency_is_symmetric(ds, adj, n)
    k = max(d for d in range(1, 11) if adj[d - 1].size)            # a populated bucket
    for kind in ("self_loop", "duplicate", "one_way", "out_of_range"):
        bad = [a.copy() for a in adj]
        row0 = int(ds[k, 0])
        if kind == "self_loop":
            bad[k - 1][0, 0] = row0
        elif kind == "duplicate" and k >= 2:
            bad[k - 1][0, 1] = bad[k - 1][0, 0]
        elif kind == "one_way":
            others = [v for v in range(n) if v != row0 and v not in bad[k - 1][0]]
            bad[k - 1][0, 0] = others[0]
        elif kind == "out_of_range":
            bad[k - 1][0, 0] = n
        else:
            continue
        assert not brute(bad) and not adjacency_is_symmetric(ds, bad, n), kind


def test_state_dict_matches_reference_manifest():
    """Same keys and shapes as the reference's create_integrated_net at the tuned config => .mod files load as is."""
    from jova_amd.train import TUNED_DAVIS_HPARAMS, count_parameters, create_integrated_net
    man = json.load(open(os.path.join(H.GOLDEN, "manifest_tuned.json")))
    net = create_integrated_net(TUNED_DAVIS_HPARAMS, {})
    sd = net.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == man["state_dict"]
    assert count_parameters(net) == man["n_trainable"] == 4900207


def test_product_path_refuses_cpu_tensors():
    from jova_amd.nn.layers import GraphPool
    z = H.load("index_toy4")
    gl = H.oracle_glist(H.unpack_mols(z), torch.zeros(int(z["nodes"].shape[0]), 8))
    with pytest.raises(RuntimeError, match="GPU"):
        GraphPool()(gl)


# ---------------------------------------------------------------------------------------------- reference-shaped collators
class _Prot(object):
    def __init__(self, name):
        self.name = name

    def get_name(self):
        return self.name


class _EcfpMol(object):
    def __init__(self, bits):
        self.bits = bits

    def get_array(self):
        return self.bits


@pytest.mark.parametrize("name", ["toy4", "deg0_b16", "hubs_b8", "pair_b2", "davis_b128"])
def test_batch_collator_reference_api_is_bit_exact(name):
    """batch_collator / process_gconv_view_data / process_ecfp_view_data with the reference's signatures and return
    tuples (jova/data/data.py:166-306), fed duck-typed (mol, protein) pairs through DtiDataset items: the gconv list
    equals the reference-made index golden bit for bit (native merge, jova_collate_gconv)."""
    import torch
    from jova_amd.data.data import DtiDataset, batch_collator
    z = H.load("index_" + name)
    raw = H.unpack_mols(z)
    cms = [ConvMol(f, a, smiles=str(i)) for i, (f, a) in enumerate(raw)]
    rng = np.random.RandomState(5)
    prots = [_Prot("P%d" % (i % 3)) for i in range(len(cms))]
    desc = {"P%d" % k: rng.normal(size=(1, 17)) for k in range(3)}
    ecfps = [_EcfpMol((rng.uniform(size=32) < 0.2).astype(np.float64)) for _ in cms]
    y = [np.array([float(i)]) for i in range(len(cms))]
    w = [np.array([1.0]) for _ in cms]
    ds = DtiDataset(x_s=(list(zip(ecfps, prots)), list(zip(cms, prots))), y_s=(y, y), w_s=(w, w))
    batch = [ds[i] for i in range(len(ds))]
    bsz, data = batch_collator(batch, desc, {"ecfp8": True, "weave": False, "gconv": True}, cuda_prot=False)
    assert bsz == len(cms) and list(data) == ["ecfp8", "gconv"]
    (mol_data, prots_tensor, prot_names, x_data), yv, wv = data["gconv"]
    assert prot_names == [p.get_name() for p in prots] and x_data[0][0] is cms[0]
    assert prots_tensor.dtype == torch.float32 and tuple(prots_tensor.shape) == (len(cms), 17)
    assert np.array_equal(prots_tensor.numpy(), np.array([desc[n][0] for n in prot_names], dtype=np.float32))
    assert float(yv[len(cms) - 1][0]) == len(cms) - 1.0 and float(wv[0][0]) == 1.0
    atoms, deg_slice, memb, n_atoms = (t.cpu() if torch.is_tensor(t) else t for t in mol_data[:4])
    assert atoms.dtype == torch.int64 and np.array_equal(atoms.numpy(), z["nodes"].astype(np.int64))
    assert deg_slice.dtype == torch.int64 and np.array_equal(deg_slice.numpy(), z["deg_slice"])
    assert np.array_equal(memb.numpy(), z["membership"]) and list(n_atoms) == list(z["n_atoms"])
    assert len(mol_data) == 14
    for d in range(1, 11):
        a = mol_data[3 + d].cpu()
        assert a.dtype == torch.int32 and np.array_equal(a.numpy(), z["adj_%d" % d])
    (mols_tensor, prots2, names2, _), _, _ = data["ecfp8"]
    assert mols_tensor.dtype == torch.float32 and np.array_equal(mols_tensor.cpu().numpy(), np.array([e.bits for e in ecfps], np.float32))
    assert names2 == prot_names
    with pytest.raises(NotImplementedError):
        batch_collator(batch, desc, "no_such_view")


def test_native_collate_rejects_bad_molecules():
    from jova_amd.data.collate import gconv_input_list
    import torch
    m = ConvMol(np.zeros((3, 75), dtype=np.int64), [[1], [0, 2], [1]], smiles="a")
    m.nbr_flat = m.nbr_flat.copy()
    m.nbr_flat[0] = 7                                   # neighbour index outside the molecule
    with pytest.raises(RuntimeError, match="out of range"):
        gconv_input_list([m], torch.device("cpu"))
    empty = gconv_input_list([], torch.device("cpu"))   # an empty batch is legal
    assert empty[0].shape[0] == 0 and empty[3] == []


def test_protein_table_rejects_word_ids_outside_the_vocabulary():
    """nn.Embedding raises IndexError on an id > num_embeddings - 1 (models.py:704-715); the resident table is checked
    once when it is built (construction fails before any device work)."""
    from jova_amd.nn.models import ProteinTable
    prof = {"P0": np.array([[0, 5, 60]] * 2, dtype=np.int64), "P1": np.array([[3, 61, 2]], dtype=np.int64)}
    with pytest.raises(IndexError):
        ProteinTable(prof, torch.device("cpu"), vocab_size=60)
    with pytest.raises(IndexError):
        ProteinTable({"P0": np.array([[-1, 2, 3]])}, torch.device("cpu"), vocab_size=60)
    t = ProteinTable({"P0": prof["P0"]}, torch.device("cpu"), vocab_size=60)      # the padding id (= vocab) is legal
    assert t.table.shape == (2, 3) and t.row_of["P0"] == (0, 2)
