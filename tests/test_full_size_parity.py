"""Full-size parity against fixtures made by the imported reference (tools/make_golden_full.py): the tuned Davis model
at B = 128, KIBA- and Metz+PSC-shaped batches, the default-hparams model at B = 256, a long-protein stress batch
(P_max = 1024, S = 1065 tokens) and the multi-task variant of config 5 (output_dim = 12 tasks, 70 % of the labels
missing: "full ToxCast"-shaped), each run by the reference in fp32 AND in double.

Bound per tensor: err(x, ref64) <= max(1e-5, 2 * err(ref32, ref64)).  What the fixtures show: at these
sizes the reference's own fp32 run sits 0.7 - 5e-5 from its double run on the outputs, and on the parameter gradients
anywhere from 4e-5 (median, KIBA-shaped batch) to 8e-3 (median, Davis batch; single tensors to 1e-1): when one of the
~10^5 pre-activations of the BatchNorm/ReLU stages lies within rounding of zero its mask flips, and one flipped sample
moves a weight-gradient row by ~1/sqrt(B).  So 1e-5 is reachable only where the reference reaches it: op level
(tests/test_gpu_ops.py goldens) and the outputs / loss here.

CPU part: the fp64 oracle against the double run (no mask flips between two double runs: 1e-9).
GPU part: the HIP path against the double run with the derived bounds; nonsat evaluation counts equal the reference's.
"""
import numpy as np
import pytest
import torch

import helpers as H


def _oracle_run(tag, dtype=torch.float64):
    from oracle import mol_graphs as og
    from oracle import nn_ref as R
    z, spec, hparams, pool, mi, pi, y, w = H.full_setup(tag)
    net = H.full_net(spec, hparams, pool)
    H.check_full_weights(z, net)
    p = {k: (v.detach().to(dtype).requires_grad_("running" not in k) if v.is_floating_point() else v)
         for k, v in net.state_dict().items()}
    b = og.agglomerate([og.SortedMol(*pool.mols[i]) for i in mi])
    gl = og.gconv_input_list(b)
    glist = [torch.from_numpy(gl[0]), torch.from_numpy(gl[1]), torch.from_numpy(gl[2]), gl[3]] + \
            [torch.from_numpy(a) for a in gl[4:]]
    idx = og.embedding_indices([pool.prot_names[i] for i in pi], pool.protein_profile, fill_val=pool.vocab)
    ob = dict(ecfp=torch.from_numpy(pool.ecfp[mi]).to(dtype), glist=glist, prot_idx=torch.from_numpy(idx),
              psc=torch.from_numpy(pool.psc[pi]).to(dtype) if "psc" in spec["views"] else None)
    hp = spec["hp"]
    out, n_eval = R.dti_forward(p, ob, hp["heads"], hp["pcnn_layers"], len(hp["lin_dims"]))
    loss = R.mse_loss(out, torch.from_numpy(w).to(dtype), torch.from_numpy(y).to(dtype))
    loss.backward()
    grads = {}
    for k, v in p.items():
        if torch.is_tensor(v) and v.requires_grad:
            g = v.grad.detach().double().numpy().copy() if v.grad is not None else np.zeros(tuple(v.shape))
            if k.endswith("embedding.weight"):
                g[hp["vocab"]] = 0                          # nn.Embedding(padding_idx): that row gets no gradient
            grads[k] = g
    return z, out.detach().double().numpy(), float(loss), n_eval, grads


@pytest.mark.parametrize("tag", ["davis_b128", "metz_psc_b128", "metz_psc_t12_b128"])
def test_oracle_fp64_equals_reference_double_run_at_full_size(tag):
    torch.set_num_threads(8)
    z, out, loss, n_eval, grads = _oracle_run(tag)
    H.check_oracle_projections(z, grads, tag)                        # every element of every gradient tensor
    assert n_eval == int(z["nonsat_evals"][1]) == int(z["nonsat_evals"][0])
    assert H.rel_err(out, z["out64"]) <= 1e-9
    assert abs(loss - float(z["loss64"])) <= 1e-10 * abs(float(z["loss64"]))
    gmax = max(float(z[k][1]) for k in z.files if k.startswith("gstat."))
    for k in z.files:
        if k.startswith("gs64."):
            name = k[len("gs64."):]
            g = grads[name].reshape(-1)[H.sample_index(grads[name].size)]
            mx = float(z["gstat." + name][1])
            # (analytically zero gradients are rounding noise of 1e-16 in both double runs: absolute term)
            assert float(np.abs(g - z[k]).max()) <= 1e-8 * mx + 1e-12 * gmax, name
            assert abs(float(np.sqrt((grads[name] ** 2).sum())) - float(z["gstat." + name][0])) <= \
                1e-8 * float(z["gstat." + name][0]) + 1e-12 * gmax, name


def test_fixture_bounds_are_what_the_docstring_says():
    """The numbers quoted above, asserted (so nobody reads the gradient bounds of the GPU test as hand-picked): outputs
    of the reference's fp32 run within 1e-4 of its double run; median gradient distance between 1e-5 and 2e-2."""
    for tag in H.FULL_TAGS:
        z = H.load("full_" + tag)
        assert 1e-6 <= H.rel_err(z["out32"], z["out64"]) <= 1e-4
        gmax = max(float(z[k][1]) for k in z.files if k.startswith("gstat."))
        rel = [float(z[k][2] / z[k][1]) for k in z.files if k.startswith("gstat.") and z[k][1] >= 1e-9 * gmax]
        assert 1e-5 <= float(np.median(rel)) <= 2e-2, (tag, np.median(rel))


def test_reference_fp32_scatter_of_the_joint_attention_stage():
    """What the REFERENCE's own fp32 arithmetic does to the joint-attention stage when only its summation order changes
    (stage_noise_davis_b128.npz, tools/make_golden_full.py:gen_stage_noise).  The imported reference's
    JointAttention runs in fp32 from the double run's stage inputs and upstream gradient and is compared with the double
    run, (a) with eight CPU thread counts and (b) six times with its latent dimension re-indexed (a permutation applied
    consistently to every parameter that touches the 512 latent features: the identical function, every contraction over
    the latent dimension summed in another order).
      * (a) changes little: torch's CPU GEMM sums in the same order whatever the thread count -- the gradient w.r.t. the
        graph view stays within 8e-7 of the double run;
      * (b) is what a different fp32 GEMM does: the stage output moves as in (a) (2e-5), but the gradient w.r.t. the graph
        view lands anywhere between 7e-7 and 9.1e-4 of its scale (three of six runs above 2.9e-4): one of the 6e7 FFN ReLU
        masks flips on an influential unit or it does not.
    So the 1.2e-4 / 1.3e-4 by which a forward on bf16 pieces / a zero-padded view projection moved that gradient in round 2
    are inside the reference's own fp32 scatter, and the stage-wise GPU test takes its bound for the view gradients from
    these runs (tests/test_gpu_ops.py::test_davis_full_size_stages_vs_oracle)."""
    z = H.load("stage_noise_davis_b128")
    assert len(z["threads"]) == 8 and len(z["perm.d_view1"]) == 6
    assert float(z["d_view1"].max()) <= 1e-6 and float(z["d_view0"].max()) <= 1e-6           # (a)
    assert 1e-5 <= float(z["ja"].max()) <= 5e-5 and 1e-5 <= float(z["perm.ja"].max()) <= 5e-5
    assert float(z["perm.d_view1"].min()) <= 1e-6 and float(z["perm.d_view1"].max()) >= 5e-4  # (b): all or nothing
    assert int((z["perm.d_view1"] > 1.3e-4).sum()) >= 3
    assert float(z["perm.d_view0"].max()) <= 1e-5


@pytest.mark.parametrize("tag", ["below", "above"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_oracle_nonsat_decision_next_to_the_threshold(tag, dtype):
    """Stop rule within 4e-7 of ep = 1e-4 (op_nonsat_margin.npz): the oracle takes the reference's evaluation count."""
    from oracle import nn_ref as R
    z = H.load("op_nonsat_margin")
    n_atoms = [int(v) for v in z["n_atoms"]]
    x = torch.from_numpy(z[tag + ".x"]).to(dtype).requires_grad_(True)
    out, n_eval = R.nonsat(R.gather2d(x, n_atoms))
    assert n_eval == int(z[tag + ".evals"])
    assert abs(float(z[tag + ".means"][int(z["k_star"])]) - 1e-4) < 1e-6
    assert H.rel_err(out.detach().numpy(), z[tag + ".out"]) <= 1e-5


# ---------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["below", "above"])
def test_hip_nonsat_decision_next_to_the_threshold(device, tag):
    from jova_amd.nn import functional as F_
    from jova_amd.nn.layers import GraphGather2D
    z = H.load("op_nonsat_margin")
    n_atoms = [int(v) for v in z["n_atoms"]]
    x = torch.from_numpy(z[tag + ".x"]).to(device).requires_grad_(True)
    out = GraphGather2D(activation="nonsat")([x, None, None, n_atoms], len(n_atoms))
    ctrl = F_.NonsatFunction.last_n_eval.cpu().tolist()
    assert ctrl[0] == int(z[tag + ".evals"]) and ctrl[2] == 0, "evaluations %s, reference %d" % (ctrl, int(z[tag + ".evals"]))
    H.assert_close(out, z[tag + ".out"], tol=1e-5, what="out")
    (out * torch.from_numpy(z[tag + ".gout"]).to(device)).sum().backward()
    H.assert_close(x.grad, z[tag + ".gx"], tol=1e-5, what="dx")


@pytest.mark.gpu
@pytest.mark.parametrize("tag", H.FULL_TAGS)
def test_hip_step_vs_reference_full_size(device, tag):
    from jova_amd.data.collate import collate_pairs
    from jova_amd.feat.mol_graphs import ConvMol
    from jova_amd.nn import functional as F_
    from jova_amd.train import Trainer
    z, spec, hparams, pool, mi, pi, y, w = H.full_setup(tag)
    net = H.full_net(spec, hparams, pool)
    H.check_full_weights(z, net)
    net = net.to(device).train()
    cms = [ConvMol(f, a, smiles=str(i)) for i, (f, a) in enumerate(pool.mols)]
    batch = collate_pairs(pool, cms, mi, pi, y, w, device, net[0].models[2][0], with_psc="psc" in spec["views"])
    tr = Trainer(net, hparams)
    tr.flat.zero_grad()
    loss, out = tr.forward_loss(batch)
    loss.backward()
    ctrl = F_.NonsatFunction.last_n_eval.cpu().tolist()
    assert ctrl[0] == int(z["nonsat_evals"][0]) == int(z["nonsat_evals"][1]) and ctrl[2] == 0
    e_ref = float(z["out_err32"])
    e_out = H.rel_err(out.detach().cpu().numpy(), z["out64"])
    assert e_out <= max(1e-5, 2 * e_ref), "outputs: err %.3e vs the reference's own fp32 %.3e" % (e_out, e_ref)
    l64, l32 = float(z["loss64"]), float(z["loss32"])
    assert abs(loss.item() - l64) <= max(1e-5, 2 * float(z["loss_err32"])) * abs(l64)
    grads = {k: p.grad.detach().cpu().double().numpy() for k, p in net.named_parameters()}
    emb = "0.models.2.0.embedding.weight"
    assert float(np.abs(grads[emb][spec["hp"]["vocab"]]).max()) == 0.0         # the padding row receives nothing
    H.check_full_grads(z, grads, tag)
    # whole tensors, not the 2 048-element sample: against the fp64 oracle run on this box's host cores, which the
    # fixture's whole-tensor projections tie to the reference's double run element by element
    torch.set_num_threads(min(16, torch.get_num_threads()) or 1)
    _, out64, loss64, _, g64 = _oracle_run(tag)
    assert H.rel_err(out64, z["out64"]) <= 1e-9
    H.check_oracle_projections(z, g64, tag)
    g32 = _oracle_run(tag, torch.float32)[4]
    H.check_full_grads_whole(z, grads, g64, tag, fp32_oracle=g32)
    if spec["hp"].get("out_dim", 1) > 1:
        # the evaluation metrics of a multi-task batch: per task on its labelled samples, observation-weighted nan-mean
        # (jova/metrics/evaluate.py:17 -> all_metrics.py:517-600), on the device, against the oracle on the same outputs
        from jova_amd.metrics import MetricsLog
        from oracle import metrics_ref as MR
        log = MetricsLog(device)
        score = log(out.detach() * batch.w, batch)
        want = MR.dti_metrics_multitask(y.astype(np.float32), (out.detach() * batch.w).cpu().numpy(), w)
        got = log.rows()[0, :4].cpu().numpy()
        assert np.allclose(got, want, rtol=1e-10, atol=1e-12), (got, want)
        assert abs(float(score) - want[3]) <= 1e-10
