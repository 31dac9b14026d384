#!/usr/bin/env python3
"""Step time of the section-8(f3) view encoders inside the joint model (never timed before round 3): for each of
   gnn    ecfp8 + GraphNeuralNet2D + pcnn2d      (create_gnn_net, proj/dti/train_joint_attn.py:177-184)
   weave  ecfp8 + WeaveModel(2D) + pcnn2d        (create_weave_net, :119-152)
   rnn    ecfp8 + gconv + Prot2Vec/ProteinRNN    (create_prot_net 'rnn', :77-86)
one JSON line: fwd + bwd + SGD on Davis-shaped synthetic batches (B = 128) resident in HBM, the reference's default view
hparams (:1043-1066: weave dim 50, gnn dim 100 / 3 layers, rnn hidden 27), the rest the tuned Davis hparams.
    python tools/bench_views.py [view ...] [--steps K] [--warmup W]
Under `rocprofv3 --kernel-trace` the same command gives the kernels of each view (tools/kt_groups.py)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jova_amd import synth                                                  # noqa: E402
from jova_amd.data import data as D                                         # noqa: E402
from jova_amd.data.collate import DtiBatch, collate_pairs                   # noqa: E402
from jova_amd.feat.mol_graphs import ConvMol, WeaveMol                      # noqa: E402
from jova_amd.train import TUNED_DAVIS_HPARAMS, Trainer, count_parameters, create_integrated_net   # noqa: E402

N_FP = 2000                      # fingerprint vocabulary of the gnn view (r-radius subgraph ids; Davis: a few thousand)


class _GnnMol(object):
    def __init__(self, adjacency, fingerprints):
        self.adjacency, self.fingerprints = adjacency, fingerprints


class _Prot(object):
    def __init__(self, name):
        self._n = name

    def get_name(self):
        return self._n


class ViewBatch(DtiBatch):
    """DtiBatch whose compound view is not gconv: model_input() hands the view's own input structure."""

    def __init__(self, ecfp, view_in, prot_names, y, w, prot_input):
        super(ViewBatch, self).__init__(ecfp, [None, None, None, None], prot_names, y, w, prot_input)
        self.view_in = view_in

    def model_input(self):
        prot = self.prot_input if self.prot_input is not None else self.prot_names
        return [self.ecfp, self.view_in, prot]


def make_batches(view, pool, hp, device, model, n_batches, bsz):
    rng = np.random.RandomState(7)
    cms = [ConvMol(f, a, smiles=str(i)) for i, (f, a) in enumerate(pool.mols)]
    prot2vec = model[0].models[2][0]
    out = []
    for _ in range(n_batches):
        mi, pi, y, w = pool.sample_pairs(rng, bsz)
        base = collate_pairs(pool, cms, mi, pi, y, w, device, prot2vec)
        if view == "rnn":
            out.append(base)
            continue
        X = np.empty((bsz, 3, 1), dtype=object)
        for k, m in enumerate(mi):
            feats, adj = pool.mols[m]
            n = feats.shape[0]
            if view == "gnn":
                A = np.zeros((n, n), np.float32)
                for a, nb in enumerate(adj):
                    A[a, nb] = 1.0
                mol = _GnnMol(A, rng.randint(0, N_FP, size=n))
            else:
                pairs = rng.normal(size=(n, n, 14)).astype(np.float32)
                mol = WeaveMol(feats.astype(np.float32), pairs, smiles=str(m))
            X[k, 0, 0] = (mol, _Prot(pool.prot_names[pi[k]]))
        fn = D.process_gnn_view_data if view == "gnn" else D.process_weave_view_data
        mol_data, _, _, _ = fn(X, {pool.prot_names[i]: np.zeros((1, 4), np.float32) for i in pi}, 0, cuda_prot=False)
        out.append(ViewBatch(base.ecfp, mol_data, base.prot_names, base.y, base.w, base.prot_input))
    return out


def run(view, steps, warmup, device):
    hp = dict(TUNED_DAVIS_HPARAMS)
    hp["weave"] = {"dim": 50, "update_pairs": False}
    hp["gnn"] = {"fingerprint_size": N_FP, "num_layers": 3, "dim": 100}
    c_views = {"gnn": ("ecfp8", "gnn"), "weave": ("ecfp8", "weave"), "rnn": ("ecfp8", "gconv")}[view]
    hp["prot"] = dict(hp["prot"], model_types=["rnn" if view == "rnn" else "pcnn2d"], rnn_hidden_state_dim=27)
    pool = synth.SyntheticPool("davis", seed=1)
    torch.manual_seed(1)
    model = create_integrated_net(hp, pool.protein_profile, c_views=c_views).to(device).train()
    tr = Trainer(model, hp)
    bsz = hp["tr_batch_size"]
    batches = make_batches(view, pool, hp, device, model, 4, bsz)
    for i in range(warmup):
        tr.step(batches[i % len(batches)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss, _ = tr.step(batches[i % len(batches)])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tr.check_health()
    n_atoms = float(np.mean([b.view_in[0].shape[0] if isinstance(b, ViewBatch) else b.glist[0].shape[0] for b in batches]))
    return dict(view=view, metric="DTI training pairs/sec (fwd+bwd+SGD), one MI355X", value=round(steps * bsz / dt, 1),
                unit="pairs/s", ms_per_step=round(dt / steps * 1e3, 3), steps=steps, warmup=warmup, batch=bsz,
                atoms_per_batch=n_atoms, trainable_params=count_parameters(model), final_loss=round(float(loss.item()), 6),
                views="+".join(c_views) + "__" + hp["prot"]["model_types"][0], data="synthetic Davis-shaped, resident in HBM")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("views", nargs="*", default=["gnn", "weave", "rnn"])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for v in a.views:
        print(json.dumps(run(v, a.steps, a.warmup, dev)), flush=True)
