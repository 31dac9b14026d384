"""Where does the long-protein batch (P_max 1024) lose precision?  HIP vs fp64 oracle, stage by stage from the oracle's
stage inputs, next to the fp32 oracle (torch CPU) on the same inputs."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from oracle import mol_graphs as og, nn_ref as R
from jova_amd.data.collate import collate_pairs
from jova_amd.feat.mol_graphs import ConvMol

tag = sys.argv[1] if len(sys.argv) > 1 else "long1024_b128"
dev = torch.device("cuda:0")
z, spec, hparams, pool, mi, pi, y, w = H.full_setup(tag)
net = H.full_net(spec, hparams, pool)
sd0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
net = net.to(dev).train()
cms = [ConvMol(f, a, smiles=str(i)) for i, (f, a) in enumerate(pool.mols)]
batch = collate_pairs(pool, cms, mi, pi, y, w, dev, net[0].models[2][0], with_psc="psc" in spec["views"])
hp = spec["hp"]
torch.set_num_threads(16)

def stage(dtype):
    p = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd0.items()}
    b = og.agglomerate([og.SortedMol(*pool.mols[i]) for i in mi]); gl = og.gconv_input_list(b)
    glist = [torch.from_numpy(gl[0]), torch.from_numpy(gl[1]), torch.from_numpy(gl[2]), gl[3]] + [torch.from_numpy(a) for a in gl[4:]]
    idx = og.embedding_indices([pool.prot_names[i] for i in pi], pool.protein_profile, fill_val=pool.vocab)
    v_e = torch.from_numpy(pool.ecfp[mi]).to(dtype).unsqueeze(0)
    v_g, _ = R.gconv_net(glist, p, prefix="0.models.1.0.")
    n = hp["pcnn_layers"]
    v_p = R.pcnn2d(R.prot2vec(torch.from_numpy(idx), p["0.models.2.0.embedding.weight"]),
                   [p["0.models.2.1.lin_kernels.%d.weight" % i] for i in range(n)], [p["0.models.2.1.lin_kernels.%d.bias" % i] for i in range(n)])
    return p, [v_e, v_g, v_p]

with torch.no_grad():
    p64, views64 = stage(torch.float64)
    ja64 = R.joint_attention(views64, p64, hp["heads"], prefix="1.")
    p32, _ = stage(torch.float32)
    views32 = [v.float() for v in views64]
    ja32 = R.joint_attention(views32, p32, hp["heads"], prefix="1.")
    print("views: oracle fp32 JA from fp64 views, err vs fp64: %.3e" % H.rel_err(ja32.numpy(), ja64.numpy()))
    got_views = net[0](batch.model_input())
    for i, nm in enumerate(["ecfp", "gconv", "prot"]):
        print("view %s hip err %.3e" % (nm, H.rel_err(got_views[i].cpu().numpy(), views64[i].numpy())))
    ja_hip = net[1]([v.float().to(dev) for v in views64])
    print("JA hip from fp64 views err %.3e" % H.rel_err(ja_hip.cpu().numpy(), ja64.numpy()))
    # inside JA: pre-BN pooled sums
    def pooled(views, p, dtype):
        segs = [t.shape[0] for t in views]
        xs = [R.linear(t, p["1.lin_prjs.%d.weight" % i], p["1.lin_prjs.%d.bias" % i]) for i, t in enumerate(views)]
        x = torch.cat(xs, 0)
        xp, _ = R.mha_self(x, p["1.attention_models.0.in_proj_weight"], p["1.attention_models.0.in_proj_bias"],
                           p["1.attention_models.0.out_proj.weight"], p["1.attention_models.0.out_proj.bias"], hp["heads"])
        x1 = R.layer_norm(x + xp, p["1.attn_norm.weight"], p["1.attn_norm.bias"])
        h = torch.relu(R.linear(x1, p["1.res_blks.0.seg_lin.linear1.weight"], p["1.res_blks.0.seg_lin.linear1.bias"]))
        h = R.linear(h, p["1.res_blks.0.seg_lin.linear2.weight"], p["1.res_blks.0.seg_lin.linear2.bias"])
        x2 = R.layer_norm(x1 + h, p["1.res_blks.0.normalize.weight"], p["1.res_blks.0.normalize.bias"])
        return x, xp, x1, x2, torch.cat([t.sum(0) for t in torch.split(x2, segs, 0)], 1)
    s64 = pooled(views64, p64, torch.float64)
    s32 = pooled(views32, p32, torch.float32)
    for nm, a, b in zip(["proj", "attn out", "LN1", "LN2", "pooled"], s32, s64):
        print("oracle fp32 %-8s err %.3e" % (nm, H.rel_err(a.numpy(), b.numpy())))
    pb = s64[4]
    print("pooled: batch std / |mean| per column, median %.3e" % float((pb.std(0) / pb.mean(0).abs().clamp_min(1e-30)).median()))
    # HIP internals through the Function's saved tensors
    from jova_amd.nn import functional as F_
    ja = net[1]
    params = []
    for lin in ja.lin_prjs: params += [lin.weight, lin.bias]
    params += [ja.attn_norm.weight, ja.attn_norm.bias]
    for attn, res in zip(ja.attention_models, ja.res_blks):
        params += [attn.in_proj_weight, attn.in_proj_bias, attn.out_proj.weight, attn.out_proj.bias, res.seg_lin.linear1.weight,
                   res.seg_lin.linear1.bias, res.seg_lin.linear2.weight, res.seg_lin.linear2.bias, res.normalize.weight, res.normalize.bias]
    class Ctx: pass
    ctx = Ctx(); ctx.needs_input_grad = [False] * 40
    out = F_.JointAttentionFunction.forward(ctx, 3, hp["heads"], 1, 0.0, True, *([v.float().to(dev) for v in views64] + [q.detach() for q in params]))
    x, qkv, probs, o, xp, x1, mr1, hid, f, mr2, _ = ctx.saved[0]
    S, B, Ld = s64[0].shape
    print("hip proj err %.3e" % H.rel_err(x.cpu().numpy().reshape(S, B, Ld), s64[0].numpy()))
    print("hip attn out err %.3e" % H.rel_err(xp.cpu().numpy().reshape(S, B, Ld), s64[1].numpy()))
    print("hip LN1 err %.3e" % H.rel_err(x1.cpu().numpy().reshape(S, B, Ld), s64[2].numpy()))
    print("hip pooled err %.3e" % H.rel_err(out.cpu().numpy(), s64[4].numpy()))
