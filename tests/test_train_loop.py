"""Row H of SURVEY.md section 8: the optimizer table and argument filter (proj/dti/train_joint_attn.py:276-296), StepLR
(:360, :529) and the epoch / phase loop with its fail-fast (:384-535).

The reference builds its optimizer and scheduler from torch.optim by name, so torch.optim / StepLR ARE the reference
here: the flat-buffer kernels are run next to them on the same gradients (GPU), and a 2-epoch fit() of the small
end-to-end model that crosses a scheduler boundary is compared with the fp64 oracle trained by torch.optim + StepLR."""
import numpy as np
import pytest
import torch

import helpers as H


def test_optimizer_kwargs_filter_and_table():
    from jova_amd.train import FlatOptimizer, optimizer_kwargs
    hp = {"optimizer": "adam", "optimizer__global__lr": 1e-3, "optimizer__adam__eps": 1e-6, "optimizer__sgd__momentum": 0.9,
          "optimizer__global__weight_decay": 0.01, "lr": 5}
    assert optimizer_kwargs(hp) == {"lr": 1e-3, "eps": 1e-6, "weight_decay": 0.01}

    class Flat:
        data = torch.zeros(8)
        grad = torch.zeros(8)
        numel = 8
    o = FlatOptimizer(Flat, "Adam", **optimizer_kwargs(hp))
    assert o.opts["eps"] == 1e-6 and o.lr == 1e-3 and o.opts["betas"] == (0.9, 0.999)
    with pytest.raises(TypeError, match="unexpected keyword"):
        FlatOptimizer(Flat, "sgd", betas=(0.9, 0.99))
    r = FlatOptimizer(Flat, "Rprop", lr=0.02)                      # the whole table of :276-285 is there
    assert r.opts["etas"] == (0.5, 1.2) and float(r.s2[0]) == pytest.approx(0.02)
    with pytest.raises(TypeError, match="unexpected keyword"):     # torch.optim.Rprop takes no weight_decay either
        FlatOptimizer(Flat, "rprop", weight_decay=0.01)
    assert FlatOptimizer(Flat, "ASGD").opts["t0"] == 1e6
    with pytest.raises(AssertionError):
        FlatOptimizer(Flat, "lbfgs")


def test_step_lr_schedule():
    from jova_amd.train import StepLR

    class Opt:
        lr = initial_lr = 0.5
    s = StepLR(Opt, step_size=3, gamma=0.1)
    p = torch.nn.Parameter(torch.zeros(1))
    ref_opt = torch.optim.SGD([p], lr=0.5)
    ref = torch.optim.lr_scheduler.StepLR(ref_opt, step_size=3, gamma=0.1)
    for _ in range(8):
        ref_opt.step()
        ref.step()
        s.step()
        assert abs(s.get_last_lr()[0] - ref.get_last_lr()[0]) <= 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw", [
    ("adam", dict(lr=2e-3, weight_decay=0.01)), ("adam", dict(lr=1e-3, betas=(0.8, 0.9), eps=1e-6)),
    ("rmsprop", dict(lr=1e-3, alpha=0.9, weight_decay=0.001)), ("rmsprop", dict(lr=1e-3, momentum=0.7)),
    ("adagrad", dict(lr=1e-2, lr_decay=0.1, initial_accumulator_value=0.2, weight_decay=0.01)),
    ("adadelta", dict(rho=0.8, weight_decay=0.01)), ("adamax", dict(lr=4e-4, weight_decay=7e-4)),
    ("sgd", dict(lr=4.7e-3, weight_decay=1.91e-2)), ("sgd", dict(lr=1e-2, momentum=0.9)),
    ("asgd", dict(lr=1e-2, weight_decay=0.01)), ("asgd", dict(lr=5e-2, lambd=1e-2, alpha=0.6, t0=3)),
    ("rprop", dict(lr=1e-2)), ("rprop", dict(lr=5e-3, etas=(0.4, 1.5), step_sizes=(1e-3, 2e-2)))])
def test_flat_optimizers_and_step_lr_equal_torch_optim(device, name, kw):
    """Eight steps on a 5 000-element flat buffer with a fresh random gradient each step; StepLR(step_size=3, gamma=0.1)
    stepped after every step, so the learning rate changes twice inside the run."""
    from jova_amd.train import FlatOptimizer, StepLR
    g = torch.Generator().manual_seed(11)
    n = 5000
    p0 = torch.randn(n, generator=g)
    cls = {"adam": torch.optim.Adam, "rmsprop": torch.optim.RMSprop, "adagrad": torch.optim.Adagrad,
           "adadelta": torch.optim.Adadelta, "adamax": torch.optim.Adamax, "sgd": torch.optim.SGD,
           "asgd": torch.optim.ASGD, "rprop": torch.optim.Rprop}[name]
    ref_p = torch.nn.Parameter(p0.double().clone())
    ref_opt = cls([ref_p], **kw)
    ref_sch = torch.optim.lr_scheduler.StepLR(ref_opt, step_size=3, gamma=0.1)

    class Flat:
        pass
    flat = Flat()
    flat.data, flat.grad, flat.numel = p0.clone().to(device), torch.zeros(n, device=device), n
    opt = FlatOptimizer(flat, name, **kw)
    sch = StepLR(opt, step_size=3, gamma=0.1)
    for k in range(8):
        grad = torch.randn(n, generator=g)
        if name == "rprop":
            # Rprop reads only signs: repeat some signs, flip others, and put exact zeros in (all three branches); the
            # gradient values are dyadic so that fp32 and the fp64 torch run see identical sign products
            grad = torch.round(grad * 8) / 8
        ref_p.grad = grad.double() * 0.5
        ref_opt.step()
        ref_sch.step()
        flat.grad.copy_(grad.to(device))
        opt.step(grad_scale=0.5)
        sch.step()
        assert abs(opt.lr - ref_opt.param_groups[0]["lr"]) <= 1e-12 * max(1.0, abs(opt.lr))
    H.assert_close(flat.data, ref_p.detach().float(), tol=2e-6, what=name)
    if name == "asgd":                                             # the averaged iterate torch keeps as state['ax']
        H.assert_close(opt.s1, ref_opt.state[ref_p]["ax"].float(), tol=2e-6, what="asgd ax")
    if name == "rprop":
        H.assert_close(opt.s2, ref_opt.state[ref_p]["step_size"].float(), tol=2e-6, what="rprop step sizes")


@pytest.mark.gpu
def test_nonfinite_flag_and_fail_fast(device):
    from jova_amd._lib import P, call, stream
    flag = torch.zeros(1, dtype=torch.int32, device=device)
    x = torch.randn(1000, device=device)
    call("jova_nonfinite_flag", P(x), 1000, P(flag), stream())
    assert int(flag.item()) == 0
    x[777] = float("inf")
    call("jova_nonfinite_flag", P(x), 1000, P(flag), stream())
    assert int(flag.item()) == 2                                   # bit 1: an infinity
    x[777] = 0.0
    call("jova_nonfinite_flag", P(x), 1000, P(flag), stream())
    assert int(flag.item()) == 2                                   # sticky until the host clears it
    x[3] = float("nan")
    call("jova_nonfinite_flag", P(x), 1000, P(flag), stream())
    assert int(flag.item()) == 3                                   # bit 0: a NaN (what the reference tests for)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["sgd", "adamax", "adam", "asgd", "rprop"])
def test_optimizer_kernels_freeze_on_the_flag_and_clear_consumed_gradients(device, name):
    """skip_flag set: parameters and optimizer state stay bit-identical (the reference never reaches optimizer.step on a
    NaN batch, train_joint_attn.py:468-486); zero_grads: the gradient buffer is cleared either way."""
    from jova_amd.train import FlatOptimizer

    class Flat:
        pass
    flat = Flat()
    n = 3000
    flat.data, flat.grad, flat.numel = torch.randn(n, device=device), torch.randn(n, device=device), n
    opt = FlatOptimizer(flat, name)
    opt.skip_flag = torch.zeros(1, dtype=torch.int32, device=device)
    opt.zero_grads = True
    opt.step()
    assert float(flat.grad.abs().max()) == 0.0
    after1 = flat.data.clone()
    st = [None if t is None else t.clone() for t in (opt.s1, opt.s2)]
    flat.grad.copy_(torch.full((n,), float("nan"), device=device))
    opt.skip_flag.fill_(1)
    opt.step()
    assert torch.equal(flat.data, after1) and float(flat.grad.abs().max()) == 0.0
    for a, b in zip(st, (opt.s1, opt.s2)):
        assert a is None or torch.equal(a, b)
    opt.skip_flag.zero_()
    flat.grad.copy_(torch.randn(n, device=device))
    opt.step()
    assert not torch.equal(flat.data, after1) and torch.isfinite(flat.data).all()


@pytest.mark.gpu
def test_nan_loss_in_epoch_zero_returns_the_last_finite_weights(device):
    """a NaN in epoch 0 (no best epoch yet).  The reference breaks out before backward / optimizer.step and
    hands back the model as it stands, i.e. the weights after the last finite batch (train_joint_attn.py:468-470, :351,
    :542).  Here: batch 0 is clean, batch 1 carries a NaN label, batch 2 is clean again -- the weights must equal the
    weights after batch 0 alone, bit for bit, and fit() must report the termination."""
    import __graft_entry__  # noqa: F401  (repo root on sys.path)
    from test_gpu_ops import _e2e_setup
    from jova_amd.data.collate import DtiBatch
    from jova_amd.train import Trainer
    z, hp, hparams, net, batch = _e2e_setup(device)
    hparams = dict(hparams, dprob=0.0)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    tr = Trainer(net, hparams)
    w0 = tr.flat.data.clone()
    tr.step(batch)
    torch.cuda.synchronize()
    want = tr.flat.data.clone()
    moved = float((want - w0).abs().max())
    assert moved > 0
    net.load_state_dict(sd0)
    tr2 = Trainer(net, hparams)
    bad_y = batch.y.clone()
    bad_y.view(-1)[0] = float("nan")
    bad = DtiBatch(batch.ecfp, batch.glist, batch.prot_names, bad_y, batch.w, batch.prot_input, batch.psc)
    res = tr2.fit(lambda ep: [batch, bad, batch], None, n_epochs=3)
    assert res["epoch"] == -1 and res["history"][-1]["terminated"] == "NaN loss" and len(res["history"]) == 1
    assert torch.isfinite(tr2.flat.data).all()
    # (split-K float atomics order the sums of a step differently from run to run: equal to a small fraction of one update)
    assert float((tr2.flat.data - want).abs().max()) <= 1e-3 * moved, "weights moved after the NaN batch"


@pytest.mark.gpu
def test_fit_two_epochs_across_a_scheduler_boundary_vs_oracle(device):
    """Trainer.fit on the small end-to-end model: 2 epochs x 2 batches, StepLR(step_size=1, gamma=0.01) so the second
    epoch runs at lr / 100, evaluation phase in eval mode after each epoch.  Oracle: the fp64 restatement trained with
    torch.optim.SGD + StepLR on the same batches; per-epoch mean losses, the learning rates and the evaluation-phase
    outputs (BatchNorm running statistics, dropout off) must agree."""
    from jova_amd.data.collate import DtiBatch, gconv_input_list
    from jova_amd.feat.mol_graphs import ConvMol
    from jova_amd.train import StepLR, Trainer, create_integrated_net
    from oracle import mol_graphs as og
    from oracle import nn_ref as R
    z = H.load("e2e_small")
    hp = H.hp(z)
    hparams = {"attn_heads": hp["heads"], "attn_layers": hp["attn_layers"], "lin_dims": hp["lin_dims"], "output_dim": 1,
               "latent_dim": hp["latent"], "dprob": 0.0, "optimizer": "sgd", "optimizer__global__lr": hp["lr"],
               "optimizer__sgd__weight_decay": hp["wd"], "optimizer__adam__eps": 1.0,
               "prot": {"model_types": ["pcnn2d"], "vocab_size": hp["vocab"], "window": hp["window"],
                        "pcnn_num_layers": hp["pcnn_layers"], "embedding_dim": hp["emb"]},
               "gconv": {"dim": hp["gconv_dim"]}, "ecfp8": {"dim": hp["ecfp_dim"]}}
    prof = H.profile(z)
    names = sorted(prof)
    raw = H.unpack_mols(z)
    net = create_integrated_net(hparams, prof)
    net.load_state_dict(H.sd(z, "sd0."))
    net = net.to(device)
    tr = Trainer(net, hparams)
    tr.scheduler = StepLR(tr.optimizer, step_size=1, gamma=0.01)
    cms = [ConvMol(f, a, smiles=str(i)) for i, (f, a) in enumerate(raw)]
    splits = [[0, 1, 2, 3, 4], [5, 6, 7, 3, 1]]                   # two batches of five pairs

    def batch(ids):
        gl = gconv_input_list([cms[i] for i in ids], device)
        return DtiBatch(torch.from_numpy(z["ecfp"][ids]).to(device), gl, [("davis", names[z["prot_ids"][i]]) for i in ids],
                        torch.from_numpy(z["y"][ids]).float().to(device), torch.from_numpy(z["w"][ids]).float().to(device))
    eval_out = []

    from jova_amd.metrics import MetricsLog
    mlog = MetricsLog(device)

    def evaluate(outputs, b):
        eval_out.append(outputs.detach().cpu().double())
        return mlog(outputs, b)                                    # device-side rms / ci / r2 / score, no host read
    res = tr.fit(lambda ep: [batch(s) for s in splits], lambda ep: [batch(splits[0])], n_epochs=2, evaluate=evaluate)

    # ---- oracle: fp64, torch.optim.SGD + StepLR, BatchNorm running statistics kept by hand for the eval phase
    p = {k: (v.double().requires_grad_("running" not in k and "num_batches" not in k) if v.is_floating_point() else v.clone())
         for k, v in H.sd(z, "sd0.").items()}
    params = [v for v in p.values() if torch.is_tensor(v) and v.requires_grad]
    opt = torch.optim.SGD(params, lr=hp["lr"], weight_decay=hp["wd"])
    sch = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.01)

    def obatch(ids):
        b = og.agglomerate([og.SortedMol(*raw[i]) for i in ids])
        gl = og.gconv_input_list(b)
        glist = [torch.from_numpy(gl[0]), torch.from_numpy(gl[1]), torch.from_numpy(gl[2]), gl[3]] + \
                [torch.from_numpy(a) for a in gl[4:]]
        idx = og.embedding_indices([names[z["prot_ids"][i]] for i in ids], prof, fill_val=hp["vocab"])
        return dict(ecfp=torch.from_numpy(z["ecfp"][ids]).double(), glist=glist, prot_idx=torch.from_numpy(idx))
    want_losses = []
    for epoch in range(2):
        tot = 0.0
        for ids in splits:
            opt.zero_grad()
            out, _ = R.dti_forward(p, obatch(ids), hp["heads"], hp["pcnn_layers"], len(hp["lin_dims"]))
            loss = R.mse_loss(out, torch.from_numpy(z["w"][ids]), torch.from_numpy(z["y"][ids]))
            loss.backward()
            opt.step()
            tot += float(loss)
        sch.step()
        want_losses.append(tot / 2)
    hist = res["history"]
    assert len(hist) == 2 and abs(hist[0]["lr"] - hp["lr"] * 0.01) < 1e-12 and abs(hist[1]["lr"] - hp["lr"] * 1e-4) < 1e-14
    for h, wl in zip(hist, want_losses):
        assert abs(h["loss"] - wl) <= 2e-5 * abs(wl), (h, wl)
    assert res["epoch"] in (0, 1) and np.isfinite(res["score"]) and len(eval_out) == 2
    # the evaluation score is Jova.evaluate's (nanmean([ci, r2]) - rms) of that epoch's eval-mode outputs
    from oracle import metrics_ref as MR
    want_scores = [MR.dti_metrics(z["y"][splits[0]].astype(np.float32), o.numpy(), z["w"][splits[0]])[3] for o in eval_out]
    assert [abs(h["score"] - wsc) < 1e-9 for h, wsc in zip(hist, want_scores)] == [True, True]
    assert abs(res["score"] - max(want_scores)) < 1e-9 and res["epoch"] == int(np.argmax(want_scores))
    assert not net.training                                        # the loop leaves the model in its last (eval) phase
    # eval-mode outputs are finite and differ between the epochs (the weights moved)
    assert torch.isfinite(eval_out[0]).all() and not torch.equal(eval_out[0], eval_out[1])
