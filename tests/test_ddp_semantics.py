"""Data-parallel semantics: "the reference executed on each shard independently, gradients
averaged".  The reference has no distributed path, so the pin is the oracle per shard (BatchNorm statistics, A_max /
P_max padding and the nonsat stop rule stay per rank), averaged on the host.

CPU: ShardSampler shards are disjoint, equally sized and cover the epoch up to the dropped tail; the dropout key differs
per rank.  GPU (one GPU): G = 2 and G = 8 shards run one after another through the HIP path with the same weights, the flat
gradient buffers are summed as the all-reduce would and the optimizer applies 1/G -- compared with the oracle's per-shard
gradients averaged, and with its SGD step."""
import numpy as np
import pytest
import torch

import helpers as H

HP = {"attn_heads": 2, "attn_layers": 1, "lin_dims": [24, 20], "output_dim": 1, "latent_dim": 32, "dprob": 0.0,
      "optimizer": "sgd", "optimizer__global__lr": 4.70e-3, "optimizer__global__weight_decay": 1.91e-2,
      "tr_batch_size": 8,
      "prot": {"model_types": ["pcnn2d"], "vocab_size": 9048, "window": 11, "pcnn_num_layers": 3, "embedding_dim": 4,
               "psc_dim": 8421},
      "gconv": {"dim": 20}, "ecfp8": {"dim": 1024}}


def test_shard_sampler_is_a_disjoint_equal_cover():
    from jova_amd.data.sampler import ShardSampler
    n, bsz = 1003, 16
    for world in (1, 2, 8):
        shards = [ShardSampler(n, world, r, bsz, seed=5).indices() for r in range(world)]
        steps = n // (world * bsz)
        assert all(s.shape == (steps, bsz) for s in shards)
        flat = np.concatenate([s.reshape(-1) for s in shards])
        assert len(np.unique(flat)) == flat.size == steps * world * bsz           # disjoint, tail dropped
        assert flat.min() >= 0 and flat.max() < n
        again = ShardSampler(n, world, 0, bsz, seed=5)
        assert np.array_equal(again.indices(), shards[0])                          # deterministic
        again.set_epoch(1)
        assert not np.array_equal(again.indices(), shards[0])                      # reshuffled per epoch
    with pytest.raises(ValueError):
        ShardSampler(10, 2, 0, 16)


def test_dropout_key_differs_per_rank():
    from jova_amd.nn import functional as F_
    torch.manual_seed(1)
    F_.set_dropout_stream(0)
    k0 = F_._DropoutState.next()
    F_.set_dropout_stream(3)
    k3 = F_._DropoutState.next()
    F_.set_dropout_stream(0)
    assert k0[0] != k3[0] and k0[1] == k3[1] == 1 and F_._DropoutState.next() == k0


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 8])
def test_shards_run_sequentially_equal_oracle_per_shard_averaged(device, world):
    from jova_amd import synth
    from jova_amd.data.collate import collate_pairs
    from jova_amd.data.sampler import ShardSampler
    from jova_amd.feat.mol_graphs import ConvMol
    from jova_amd._lib import P, call, stream
    from jova_amd.train import Trainer, create_integrated_net
    from oracle import mol_graphs as og
    from oracle import nn_ref as R
    pool = synth.SyntheticPool("davis", seed=8, n_mols=40, n_prots=24)
    cms = [ConvMol(f, a, smiles=str(i)) for i, (f, a) in enumerate(pool.mols)]
    mi, pi, y, w = pool.epoch_pairs(200, seed=64)
    torch.manual_seed(1)
    net = create_integrated_net(HP, pool.protein_profile)
    sd0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.to(device).train()
    tr = Trainer(net, HP)
    bsz = HP["tr_batch_size"]
    shard_idx = [ShardSampler(len(mi), world, r, bsz, seed=3).indices()[0] for r in range(world)]

    def oracle(dtype):
        total, losses = None, []
        for idx in shard_idx:
            p = {k: (v.to(dtype).requires_grad_("running" not in k) if v.is_floating_point() else v) for k, v in sd0.items()}
            b = og.agglomerate([og.SortedMol(*pool.mols[i]) for i in mi[idx]])
            gl = og.gconv_input_list(b)
            glist = [torch.from_numpy(gl[0]), torch.from_numpy(gl[1]), torch.from_numpy(gl[2]), gl[3]] + \
                    [torch.from_numpy(a) for a in gl[4:]]
            ix = og.embedding_indices([pool.prot_names[i] for i in pi[idx]], pool.protein_profile, fill_val=pool.vocab)
            ob = dict(ecfp=torch.from_numpy(pool.ecfp[mi[idx]]).to(dtype), glist=glist, prot_idx=torch.from_numpy(ix))
            out, _ = R.dti_forward(p, ob, HP["attn_heads"], HP["prot"]["pcnn_num_layers"], len(HP["lin_dims"]))
            loss = R.mse_loss(out, torch.from_numpy(w[idx]).to(dtype), torch.from_numpy(y[idx]).to(dtype))
            loss.backward()
            losses.append(float(loss))
            g = {}
            for k, v in p.items():
                if torch.is_tensor(v) and v.requires_grad:
                    gk = v.grad.detach().double().clone() if v.grad is not None else torch.zeros_like(v).double()
                    if k.endswith("embedding.weight"):
                        gk[HP["prot"]["vocab_size"]] = 0
                    g[k] = gk
            total = g if total is None else {k: total[k] + g[k] for k in g}
        return {k: v / world for k, v in total.items()}, losses

    g64, losses64 = oracle(torch.float64)
    g32, _ = oracle(torch.float32)

    # HIP: every "rank" runs its shard with the same weights; the all-reduce is the sum of the flat gradient buffers
    summed = torch.zeros_like(tr.flat.grad)
    for r, idx in enumerate(shard_idx):
        batch = collate_pairs(pool, cms, mi[idx], pi[idx], y[idx], w[idx], device, net[0].models[2][0])
        tr.flat.zero_grad()
        loss, _ = tr.forward_loss(batch)
        loss.backward()
        assert abs(loss.item() - losses64[r]) <= 2e-5 * abs(losses64[r])
        summed += tr.flat.grad
    tr.flat.grad.copy_(summed)
    named = dict(net.named_parameters())
    gmax = max(float(v.abs().max()) for v in g64.values())
    e_all = (sum(float(((g32[k] - g64[k]) ** 2).sum()) for k in g64) / sum(float((g64[k] ** 2).sum()) for k in g64)) ** 0.5
    for k, want in g64.items():
        got = named[k].grad.detach().cpu().double() / world
        if float(want.abs().max()) < 1e-9 * gmax:      # analytically zero: rounding noise in every fp32 run
            assert float(got.abs().max()) <= max(1e-5 * gmax, 4 * float(g32[k].abs().max())), k
            continue
        e_ref = H.rel_err(g32[k].numpy(), want.numpy())
        e = H.rel_err(got.numpy(), want.numpy())
        assert e <= max(1e-5, 2 * max(e_ref, e_all)), "%s: averaged gradient err %.3e (fp32 oracle %.3e)" % (k, e, e_ref)
    # the optimizer applies the mean: grad_scale = 1 / world on the summed buffer (what Trainer.step does after all_reduce)
    lr, wd = HP["optimizer__global__lr"], HP["optimizer__global__weight_decay"]
    before = {k: v.detach().cpu().double().clone() for k, v in named.items()}
    call("jova_sgd_step", P(tr.flat.data), P(tr.flat.grad), None, tr.flat.numel, lr, wd, 0.0, 1.0 / world, 1, None, 0,
         stream())
    for k, want in g64.items():
        after = named[k].detach().cpu().double()
        step_want = -lr * (want + wd * before[k])
        bound = lr * max(1e-5, 2 * max(H.rel_err(g32[k].numpy(), want.numpy()), e_all)) * max(float(want.abs().max()), 1e-9 * gmax) \
            + 1e-6 * float(before[k].abs().max()) + lr * 1e-5 * gmax
        assert float((after - before[k] - step_want).abs().max()) <= bound, k
