"""Writes tests/golden/eval_small.npz and tests/golden/eval_small.mod -- the evaluation path (§8 f4), made by the
reference: its classes (imported from /root/reference through tools/make_golden.py) are trained for three steps with
dropout on, the state dict is written the way jova.utils.io.save_model does (torch.save(model.state_dict()) to
<name>.mod, io.py:314-325), then the reference runs the evaluation forward of Jova.evaluate_model
(train_joint_attn.py:595-627: model.eval(), no gradients) on two batches with an attention hook installed
(models.py:952-956).  Stored: the inputs, eval-mode predictions (fp32 model and an fp64 copy of it), per layer the hook's
head-averaged weights / x' / segment counts, and the per-batch metrics formed from the reference's own pieces (see
tools/make_golden_metrics.py).  The .mod file is data written by this script, loadable with weights_only=True.
"""
import copy
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as MG                                             # noqa: E402
from make_golden import M, nn                                        # noqa: E402
from make_golden_metrics import expected                            # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def main():
    hp = dict(gconv_dim=20, dprob=0.1, vocab=60, emb=4, window=11, pcnn_layers=3, ecfp_dim=1024, latent=32, heads=2,
              attn_layers=2, lin_dims=[24, 20], lr=4.70e-3, wd=1.91e-2, y_std=0.87, y_mean=5.45)
    sets = MG.mol_sets()
    mols = sets["deg0_b16"][3:12]
    names, profile = MG.make_profile(8, 6, 3, 10, hp["vocab"])
    r = np.random.RandomState(77)
    prot_ids = r.randint(0, len(names), size=len(mols))
    ecfp = (r.uniform(size=(len(mols), 1024)) < 0.06).astype(np.float32)
    y = r.normal(size=(len(mols), 1))
    w = np.ones((len(mols), 1))
    w[6] = 0.0                                                       # one masked pair in the second evaluation batch
    torch.manual_seed(3)
    net = MG.build_reference_net(hp, profile)
    captured = []
    joint = [m for m in net.modules() if isinstance(m, M.JointAttention)][0]
    opt = torch.optim.Adam(net.parameters(), lr=hp["lr"], weight_decay=hp["wd"])
    net.train()
    train_ids = [[0, 1, 2, 3, 4, 5], [3, 4, 5, 6, 7, 8], [0, 2, 4, 6, 8, 1]]

    def inputs(ids):
        mm, glist = MG.ref_batch([mols[i] for i in ids])
        return [torch.from_numpy(ecfp[ids]), (glist, len(ids)), [("davis", names[prot_ids[i]]) for i in ids]]
    for ids in train_ids:
        opt.zero_grad()
        out = net(inputs(ids))
        loss = nn.MSELoss()(out * torch.from_numpy(w[ids]).float(), torch.from_numpy(y[ids]).float())
        loss.backward()
        opt.step()
    torch.save(net.state_dict(), os.path.join(OUT, "eval_small.mod"))          # == jova.utils.io.save_model

    joint.attn_hook = lambda i, x, xp, wts, segs: captured.append((i, x.detach().clone(), xp.detach().clone(),
                                                                   wts.detach().clone(), list(segs)))
    net.eval()
    net64 = copy.deepcopy(net).double()
    [m for m in net64.modules() if isinstance(m, M.JointAttention)][0].attn_hook = None
    arrays = dict(MG.pack_mols(mols), prot_ids=prot_ids, ecfp=ecfp, y=y, w=w, hp=np.array(json.dumps(hp)))
    for nm in names:
        arrays["profile." + nm] = profile[nm]
    eval_ids = [[0, 1, 2, 3, 4], [5, 6, 7, 8]]
    arrays["eval_ids"] = np.array([",".join(map(str, e)) for e in eval_ids])
    float_ = torch.Tensor.float
    for b, ids in enumerate(eval_ids):
        del captured[:]
        with torch.set_grad_enabled(False):
            out = net(inputs(ids))
            torch.Tensor.float = lambda self: self.double()          # the reference casts inputs with .float() inside
            try:
                x64 = inputs(ids)
                x64[0] = x64[0].double()
                x64[1][0][0] = x64[1][0][0].double()
                out64 = net64(x64)
            finally:
                torch.Tensor.float = float_
        arrays["b%d.out" % b], arrays["b%d.out64" % b] = out.numpy(), out64.numpy()
        assert len(captured) == hp["attn_layers"]
        for (i, x, xp, wts, segs) in captured:
            arrays["b%d.hook%d.x" % (b, i)] = x.numpy()
            arrays["b%d.hook%d.xp" % (b, i)] = xp.numpy()
            arrays["b%d.hook%d.wts" % (b, i)] = wts.numpy()
            arrays["b%d.hook%d.segs" % (b, i)] = np.array(segs)
        # evaluate_model passes the raw predictions (no "* w"), train_joint_attn.py:619-627
        arrays["b%d.metrics" % b] = expected(y[ids].reshape(-1).astype(np.float32), out.numpy().reshape(-1),
                                             w[ids].reshape(-1), hp["y_std"], hp["y_mean"])
        print("batch", b, "out", out.numpy().ravel(), "err32", np.abs(out.numpy() - out64.numpy()).max(),
              "metrics", arrays["b%d.metrics" % b][:4])
    np.savez_compressed(os.path.join(OUT, "eval_small.npz"), **arrays)
    print("state dict keys:", len(net.state_dict()), "file bytes:", os.path.getsize(os.path.join(OUT, "eval_small.mod")))


if __name__ == "__main__":
    main()
