"""Model assembly and training step of the multi-view DTI model, MI355X side.

Counterpart of the reference's harness /root/reference/proj/dti/train_joint_attn.py for the view combination
``ecfp8-gconv__pcnn2d`` (plus ``psc``): create_ecfp_net :65-67, create_prot_net :70-116, create_gconv_net :155-174,
create_integrated_net :188-243 (same nn.Sequential structure => same state-dict keys, so reference ``.mod``
checkpoints load unchanged) and the loop body of Jova.train :428-486 (zero_grad -> forward -> MSELoss(out*w, y)
-> backward -> optimizer.step).

Differences that are deliberate (MI355X-first):
  * all parameters live in ONE flat fp32 buffer and all gradients in another; the weight-gradient kernels
    accumulate straight into it, data parallelism all-reduces that single buffer (RCCL), and the optimizer is one
    kernel over the flat buffers;
  * batches are collated to device-resident index tensors (jova_amd/data/collate.py) instead of per-step Python.
"""
import os

import torch
import torch.nn as nn

from . import _lib
from .nn import functional as F_
from .nn.layers import (BatchNorm1d, Dropout, ElementwiseBatchNorm, GraphConvLayer, GraphGather2D, GraphPool, Linear, ReLU,
                        Unsqueeze)
from .nn.models import (GraphConvSequential, GraphNeuralNet2D, JointAttention, NwayForward, Prot2Vec, ProteinCNN2D,
                        ProteinRNN)

TUNED_DAVIS_HPARAMS = {
    # first row of proj/dti/soek_res/bayopt_search_integrated_view_attn_no_gan_pcnn2d_ecfp8_gconv_dti_2019_11_27__21_57_05_gp.csv
    "attn_heads": 2, "attn_layers": 1, "lin_dims": [267, 637], "output_dim": 1, "latent_dim": 512, "dprob": 0.211,
    "tr_batch_size": 128, "optimizer": "sgd", "optimizer__global__lr": 4.70e-3,
    "optimizer__global__weight_decay": 1.91e-2, "explain_mode": False,
    "prot": {"model_types": ["pcnn2d"], "vocab_size": 9048, "window": 11, "pcnn_num_layers": 3, "embedding_dim": 22,
             "psc_dim": 8421},
    "gconv": {"dim": 194}, "ecfp8": {"dim": 1024},
}


def create_ecfp_net(hparams):
    return nn.Sequential(Unsqueeze(dim=0))


def create_gconv_net(hparams):
    dim = hparams["gconv"]["dim"]
    gconv_model = GraphConvSequential(GraphConvLayer(in_dim=75, out_dim=64), BatchNorm1d(64), ReLU(), GraphPool(),
                                      GraphConvLayer(in_dim=64, out_dim=64), BatchNorm1d(64), ReLU(), GraphPool(),
                                      Linear(in_features=64, out_features=dim), BatchNorm1d(dim), ReLU(),
                                      Dropout(hparams["dprob"]), GraphGather2D(activation='nonsat'))
    return nn.Sequential(gconv_model)


def create_prot_net(hparams, model_type, protein_profile):
    if model_type == "pcnn2d":
        return nn.Sequential(Prot2Vec(protein_profile=protein_profile, vocab_size=hparams["prot"]["vocab_size"],
                                      embedding_dim=hparams["prot"]["embedding_dim"], batch_first=False),
                             ProteinCNN2D(dim=hparams["prot"]["embedding_dim"], window=hparams["prot"]["window"],
                                          num_layers=hparams["prot"]["pcnn_num_layers"]))
    if model_type == "rnn":                                         # train_joint_attn.py:77-86
        return nn.Sequential(Prot2Vec(protein_profile=protein_profile, vocab_size=hparams["prot"]["vocab_size"],
                                      embedding_dim=hparams["prot"]["embedding_dim"], batch_first=False),
                             ProteinRNN(in_dim=hparams["prot"]["embedding_dim"] * hparams["prot"]["window"],
                                        hidden_dim=hparams["prot"]["rnn_hidden_state_dim"], dropout=hparams["dprob"],
                                        batch_first=False))
    if model_type == "p2v":
        return Prot2Vec(protein_profile=protein_profile, vocab_size=hparams["prot"]["vocab_size"],
                        embedding_dim=hparams["prot"]["embedding_dim"], batch_first=False)
    if model_type == "psc":
        return nn.Sequential(Unsqueeze(dim=0))
    raise NotImplementedError("protein view %r is outside the hot path (SURVEY.md section 8)" % model_type)


def create_weave_net(hparams):
    """proj/dti/train_joint_attn.py:119-152."""
    from .nn.models import WeaveModel
    from .utils.args import WeaveGatherArgs, WeaveLayerArgs
    weave_args = (
        WeaveLayerArgs(n_atom_input_feat=75, n_pair_input_feat=14, n_atom_output_feat=50, n_pair_output_feat=50,
                       n_hidden_AA=50, n_hidden_PA=50, n_hidden_AP=50, n_hidden_PP=50,
                       update_pair=hparams["weave"]["update_pairs"], activation='relu', batch_norm=True,
                       dropout=hparams["dprob"]),
        WeaveLayerArgs(n_atom_input_feat=50, n_pair_input_feat=14, n_atom_output_feat=hparams["weave"]["dim"],
                       n_pair_output_feat=50, n_hidden_AA=50, n_hidden_PA=50, n_hidden_AP=50, n_hidden_PP=50,
                       update_pair=hparams["weave"]["update_pairs"], batch_norm=True, dropout=hparams["dprob"],
                       activation='relu'),
    )
    wg_args = WeaveGatherArgs(conv_out_depth=hparams["weave"]["dim"], gaussian_expand=True, n_depth=hparams["weave"]["dim"])
    return nn.Sequential(WeaveModel(weave_args, weave_gath_arg=wg_args, weave_type='2D'))


def create_gnn_net(hparams):
    """proj/dti/train_joint_attn.py:177-184."""
    dim = hparams["gnn"]["dim"]
    gnn_model = GraphNeuralNet2D(num_fingerprints=hparams["gnn"]["fingerprint_size"], embedding_dim=dim,
                                 num_layers=hparams["gnn"]["num_layers"])
    return nn.Sequential(gnn_model, Linear(dim, dim), ElementwiseBatchNorm(dim), ReLU(), Dropout(hparams["dprob"]))


def create_integrated_net(hparams, protein_profile, c_views=("ecfp8", "gconv"), attn_hook=None):
    """Same module tree as the reference builder: 0 = NwayForward, 1 = JointAttention, then the FC head.
    ``attn_hook`` is the callback the reference installs in explain mode (train_joint_attn.py:224-227)."""
    creators = {"ecfp8": create_ecfp_net, "ecfp4": create_ecfp_net, "gconv": create_gconv_net, "gnn": create_gnn_net,
                "weave": create_weave_net}
    views, seg_dims = [], []
    for cview in c_views:
        views.append(creators[cview](hparams))
        seg_dims.append(hparams[cview]["dim"])
    for m_type in hparams["prot"]["model_types"]:
        views.append(create_prot_net(hparams, m_type, protein_profile))
        if m_type == "psc":
            seg_dims.append(hparams["prot"]["psc_dim"])
        elif m_type == "p2v":
            seg_dims.append(hparams["prot"]["embedding_dim"] * hparams["prot"]["window"])
        elif m_type == "rnn":
            seg_dims.append(hparams["prot"]["rnn_hidden_state_dim"])
        else:
            seg_dims.append(hparams["prot"]["embedding_dim"])
    layers = [NwayForward(models=views),
              JointAttention(d_dims=seg_dims, latent_dim=hparams["latent_dim"], num_heads=hparams["attn_heads"],
                             num_layers=hparams["attn_layers"], dprob=hparams["dprob"], attn_hook=attn_hook)]
    p = len(views) * hparams["latent_dim"]
    for dim in hparams["lin_dims"]:
        layers += [Linear(p, dim), BatchNorm1d(dim), ReLU(), Dropout(hparams["dprob"])]
        p = dim
    layers.append(Linear(in_features=p, out_features=hparams["output_dim"]))
    return nn.Sequential(*layers)


def evaluate_model(model, model_dir, model_name, batches, y_std=1.0, y_mean=0.0, device=None):
    """Jova.evaluate_model (train_joint_attn.py:545-644): load ``<model_dir>/<model_name>`` (a state dict as written by
    utils.io.save_model), switch to eval mode, run every batch without gradients and take the per-batch metrics on the
    raw predictions.  Predictions, labels and the metric rows stay on the device until the single copy at the end.
    Returns dict(metrics=per-metric nanmean, score=mean score, rows=[n_batches, 6], y_pred, y_true) with y_* un-z-scored
    (undo_transforms, :623-624)."""
    from .metrics import MetricsLog
    from .utils import io
    device = device if device is not None else next(model.parameters()).device
    model.load_state_dict(io.load_model(model_dir, model_name, dvc=device))
    model.eval()
    log = MetricsLog(device, y_std=y_std, y_mean=y_mean)
    preds, trues = [], []
    with torch.no_grad():
        for batch in batches:
            out = model(batch.model_input())
            log(out, batch)
            preds.append(out.reshape(-1).double() * y_std + y_mean)
            trues.append(batch.y.reshape(-1).double() * y_std + y_mean)
    summary = log.summary()
    score = summary.pop("score")
    return dict(metrics=summary, score=score, rows=log.rows().cpu().numpy(),
                y_pred=torch.cat(preds).cpu().numpy() if preds else None,
                y_true=torch.cat(trues).cpu().numpy() if trues else None)


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


class FlatParameters(object):
    """Re-homes every parameter of ``model`` into one flat fp32 buffer (and every gradient into another)."""

    def __init__(self, model):
        params = [p for p in model.parameters() if p.requires_grad]
        assert all(p.dtype == torch.float32 for p in params)
        n = sum(p.numel() for p in params)
        dev = params[0].device
        # 64-element alignment keeps every parameter 256-byte aligned for vector loads
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 63) // 64 * 64
        self.data = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, o in zip(params, offs):
            self.data[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.data[o:o + p.numel()].view(p.shape)
            p.grad = self.grad[o:o + p.numel()].view(p.shape)
        self.params, self.n_params, self.numel = params, n, total

    def zero_grad(self):
        """One fill kernel of the library over the flat buffer (Trainer.step does not need it: the optimizer kernels
        clear the gradient once they have consumed it)."""
        if self.grad.is_cuda:
            _lib.call("jova_zero", _lib.P(self.grad), self.grad.numel(), _lib.stream())
        else:                                   # host buffers: only the gloo rehearsal tests of the all-reduce get here
            self.grad.zero_()


class GradAllReduce(object):
    """Sum of a flat fp32 gradient tensor over the ranks of ``process_group``.

    RCCL ("nccl" backend) reduces the device buffer in place over xGMI: the production path, one process per GPU.
    With the gloo backend (CPU rehearsal; several ranks sharing ONE GPU on a development box, which RCCL refuses) the
    buffer is staged through pinned host memory and reduced there: gloo's own handling of device tensors did not
    complete on this image, and this keeps the N-rank flow testable with the real kernels."""

    def __init__(self, process_group=None):
        self.pg = process_group
        self.backend = torch.distributed.get_backend(process_group)
        self._host = {}                          # pinned staging buffers, one per dtype
        # gloo never sees a device tensor here: its device-tensor path hung inside the step on this image (DESIGN.md
        # section 5), so the only gloo form is the host-staged one
        self.host_staged = self.backend == "gloo"

    def supports_async(self, t):
        return not (self.host_staged and t.is_cuda)

    def __call__(self, t, async_op=False):
        if t.is_cuda and self.host_staged:
            if not t.is_contiguous():
                raise ValueError("GradAllReduce: the tensor must be contiguous (a copy would receive the result)")
            buf = self._host.get(t.dtype)
            if buf is None or buf.numel() < t.numel():
                buf = self._host[t.dtype] = torch.empty(t.numel(), dtype=t.dtype, pin_memory=True)
            h = buf[:t.numel()]
            flat = t.view(-1)
            h.copy_(flat, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            torch.distributed.all_reduce(h, group=self.pg)
            flat.copy_(h, non_blocking=True)
            return None
        return torch.distributed.all_reduce(t, group=self.pg, async_op=async_op)


class OverlappedGradSync(object):
    """Gradient all-reduce of the flat buffer in two pieces, the first one overlapped with the rest of the backward.

    Parameters are laid out in module order: [view encoders (model[0]) | JointAttention + FC head].  Backward runs the
    other way round, so the second slice (89 % of the bytes at the tuned Davis config) is final as soon as the
    gradient w.r.t. the encoders' outputs exists: a tensor hook on those outputs starts its all-reduce (async: RCCL
    runs it on its own stream, ordered behind the kernels enqueued so far) while the encoders' backward still runs;
    the small first slice follows after backward.  Same sums as one all-reduce of the whole buffer."""

    def __init__(self, flat, split, process_group=None, reducer=None):
        self.flat, self.split, self.pg = flat, int(split), process_group
        self.reduce = reducer if reducer is not None else GradAllReduce(process_group)
        self.work = None

    def watch(self, tensors):
        """Call on the encoders' outputs (forward); starts the early all-reduce when their gradient arrives."""
        self.work = None
        for t in tensors:
            if torch.is_tensor(t) and t.requires_grad:
                t.register_hook(self._fire)

    def _fire(self, grad):
        if self.work is None and self.split < self.flat.grad.numel():
            self.work = self.reduce(self.flat.grad[self.split:], async_op=True)
        return grad

    def finish(self):
        """After backward: reduce what is left and wait for the early piece."""
        if self.work is None:
            self.reduce(self.flat.grad)
            return
        if self.split > 0:
            self.reduce(self.flat.grad[:self.split])
        self.work.wait()
        self.work = None


def encoder_param_split(model, flat):
    """Offset in the flat buffers where the parameters of model[1:] start (model = nn.Sequential whose first child
    holds the view encoders); None when the model is not shaped like that."""
    kids = list(model.children()) if isinstance(model, nn.Sequential) else []
    if len(kids) < 2:
        return None
    first = {id(p) for p in kids[0].parameters()}
    off = 0
    seen_rest = False
    split = None
    for p in flat.params:
        if id(p) in first:
            if seen_rest:
                return None                  # not contiguous: fall back to one all-reduce
        elif not seen_rest:
            seen_rest, split = True, off
        off += (p.numel() + 63) // 64 * 64
    return split


# torch.optim constructor defaults of the optimizers in the reference's table (train_joint_attn.py:276-285)
_OPTIM_DEFAULTS = {
    "sgd": dict(lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False),
    "adamax": dict(lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0),
    "adam": dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False),
    "rmsprop": dict(lr=1e-2, alpha=0.99, eps=1e-8, weight_decay=0.0, momentum=0.0, centered=False),
    "adagrad": dict(lr=1e-2, lr_decay=0.0, weight_decay=0.0, initial_accumulator_value=0.0, eps=1e-10),
    "adadelta": dict(lr=1.0, rho=0.9, eps=1e-6, weight_decay=0.0),
    "asgd": dict(lr=1e-2, lambd=1e-4, alpha=0.75, t0=1e6, weight_decay=0.0),
    "rprop": dict(lr=1e-2, etas=(0.5, 1.2), step_sizes=(1e-6, 50.0)),
}
_OPTIM_KIND = {"adam": 2, "rmsprop": 3, "adagrad": 4, "adadelta": 5, "asgd": 6, "rprop": 7}


def optimizer_kwargs(hparams):
    """The reference's filter (train_joint_attn.py:288-296): keys ``optimizer__<name>__<arg>`` that name the chosen
    optimizer, and ``optimizer__global__<arg>``, become constructor arguments."""
    out = {}
    key = hparams["optimizer"]
    for k, v in hparams.items():
        if "optimizer__" in k:
            tup = k.split("__")
            if key == tup[1] or tup[1] == "global":
                out[tup[2]] = v
    return out


class FlatOptimizer(object):
    """One optimizer step = one kernel over the flat parameter / gradient buffers (torch.optim semantics)."""

    def __init__(self, flat, name, **kwargs):
        self.name = name.lower()
        if self.name not in _OPTIM_DEFAULTS:
            raise AssertionError("%s optimizer could not be found" % name)
        self.opts = dict(_OPTIM_DEFAULTS[self.name])
        for k, v in kwargs.items():
            if k not in self.opts:                      # what torch.optim's constructor would say
                raise TypeError("%s.__init__() got an unexpected keyword argument %r" % (self.name, k))
            self.opts[k] = v
        o = self.opts
        if o.get("amsgrad") or o.get("centered") or o.get("nesterov") or o.get("dampening"):
            raise NotImplementedError("amsgrad / centered / nesterov / dampening variants are not implemented")
        self.flat = flat
        self.lr = float(o["lr"])                        # current learning rate (a scheduler rewrites it)
        self.initial_lr = self.lr
        self.step_idx = 0
        # device flag (jova_nonfinite_flag) that freezes parameters and state while set; grads cleared after use
        self.skip_flag = None
        self.zero_grads = False
        z = lambda: torch.zeros_like(flat.data)        # noqa: E731
        self.s1 = self.s2 = None
        if self.name == "sgd":
            self.s1 = z() if o["momentum"] else None
        elif self.name in ("adamax", "adam", "adadelta"):
            self.s1, self.s2 = z(), z()
        elif self.name == "rmsprop":
            self.s1, self.s2 = z(), (z() if o["momentum"] else None)
        elif self.name == "adagrad":
            self.s1 = torch.full_like(flat.data, float(o["initial_accumulator_value"]))
        elif self.name == "asgd":
            self.s1 = z()                               # ax; eta and mu follow from the step count (kernel side)
        elif self.name == "rprop":
            self.s1, self.s2 = z(), torch.full_like(flat.data, self.lr)       # prev gradient, per-element step size

    def step(self, grad_scale=1.0):
        f, o = self.flat, self.opts
        self.step_idx += 1
        P_, st = _lib.P, _lib.stream()
        wd = float(o.get("weight_decay", 0.0))
        tail = (P_(self.skip_flag), int(self.zero_grads), st)
        if self.name == "sgd":
            _lib.call("jova_sgd_step", P_(f.data), P_(f.grad), P_(self.s1), f.numel, self.lr, wd, float(o["momentum"]),
                      grad_scale, int(self.step_idx == 1), *tail)
        elif self.name == "adamax":
            _lib.call("jova_adamax_step", P_(f.data), P_(f.grad), P_(self.s1), P_(self.s2), f.numel, self.lr,
                      float(o["betas"][0]), float(o["betas"][1]), float(o["eps"]), wd, grad_scale, self.step_idx, *tail)
        else:
            if self.name == "adam":
                h = (o["betas"][0], o["betas"][1], o["eps"], 0.0, 0.0)
            elif self.name == "rmsprop":
                h = (o["alpha"], o["eps"], o["momentum"], 0.0, 0.0)
            elif self.name == "adagrad":
                h = (o["lr_decay"], o["eps"], 0.0, 0.0, 0.0)
            elif self.name == "adadelta":
                h = (o["rho"], o["eps"], 0.0, 0.0, 0.0)
            elif self.name == "asgd":
                h = (o["lambd"], o["alpha"], o["t0"], 0.0, 0.0)
            else:                                       # rprop
                h = (o["etas"][0], o["etas"][1], 0.0, o["step_sizes"][0], o["step_sizes"][1])
            lr = self.lr
            if self.name == "asgd":
                # torch keeps eta as state, computed at the END of a step from the learning rate of that step: step k
                # runs with eta(lr of step k-1, k-1); the first step with eta = lr
                lr, self._asgd_prev_lr = (getattr(self, "_asgd_prev_lr", None) or self.lr), self.lr
            _lib.call("jova_optim_step", _OPTIM_KIND[self.name], P_(f.data), P_(f.grad), P_(self.s1), P_(self.s2), f.numel,
                      lr, wd, grad_scale, self.step_idx, *[float(v) for v in h], *tail)


class StepLR(object):
    """torch.optim.lr_scheduler.StepLR over a FlatOptimizer: lr = initial_lr * gamma ** (epoch // step_size).  The
    reference creates StepLR(optimizer, step_size=500, gamma=0.01) and steps it once per epoch
    (train_joint_attn.py:360, :529)."""

    def __init__(self, optimizer, step_size=500, gamma=0.01):
        self.optimizer, self.step_size, self.gamma = optimizer, int(step_size), float(gamma)
        self.last_epoch = 0

    def step(self):
        self.last_epoch += 1
        self.optimizer.lr = self.optimizer.initial_lr * self.gamma ** (self.last_epoch // self.step_size)

    def get_last_lr(self):
        return [self.optimizer.lr]


class ExpAverage(object):
    """jova/utils/math.py ExpAverage: value = beta * value + (1 - beta) * x (first update sets it)."""

    def __init__(self, beta, v=0.):
        self.beta, self.value = beta, v

    def update(self, x):
        self.value = self.beta * self.value + (1 - self.beta) * x


class Trainer(object):
    """One training step = the body of the reference's batch loop (train_joint_attn.py:428-486), no host sync."""

    def __init__(self, model, hparams, world_size=1, process_group=None, force_sync=False, rank=0):
        F_.set_dropout_stream(rank)             # ranks share torch.manual_seed (same weights) but not dropout masks
        self.rank = rank
        self.model = model
        self.hp = hparams
        self.flat = FlatParameters(model)
        self.world_size = world_size
        self.pg = process_group
        hp_opt = dict(hparams)
        hp_opt.setdefault("optimizer", "sgd")
        self.optimizer = FlatOptimizer(self.flat, hp_opt["optimizer"], **optimizer_kwargs(hp_opt))
        self.scheduler = StepLR(self.optimizer, step_size=500, gamma=0.01)       # train_joint_attn.py:360
        self.nan_flag = torch.zeros(1, dtype=torch.int32, device=self.flat.data.device)
        # a non-finite loss freezes the weights from that batch on (the reference leaves its loop before backward /
        # optimizer.step, train_joint_attn.py:468-470); the gradient buffer is cleared by the optimizer kernel itself
        self.optimizer.skip_flag = self.nan_flag
        self.optimizer.zero_grads = True
        self.loss_sum = torch.zeros(1, dtype=torch.float32, device=self.flat.data.device)
        self.sync = None
        # (force_sync: run the collectives of the N-rank step on a one-rank group -- rehearsal on a one-GPU box)
        self.collective = world_size > 1 or force_sync
        self.reduce = GradAllReduce(process_group) if self.collective else None
        # overlap knob: JOVA_NO_OVERLAP=1 -> one all-reduce after backward (also what a host-staged reduction does)
        if self.collective and not os.environ.get("JOVA_NO_OVERLAP") and self.reduce.supports_async(self.flat.grad):
            split = encoder_param_split(model, self.flat)
            if split is not None:
                self.sync = OverlappedGradSync(self.flat, split, process_group, self.reduce)

    @property
    def step_idx(self):
        return self.optimizer.step_idx

    @property
    def lr(self):
        return self.optimizer.lr

    def fit(self, train_batches, eval_batches=None, n_epochs=1, evaluate=None, epoch_ckpt=(2, 1.0), log=None):
        """The reference's epoch / phase loop (Jova.train, train_joint_attn.py:384-535) around ``step``.

        train_batches / eval_batches: callables ``epoch -> iterable of DtiBatch`` (e.g. a BatchPrefetcher over a
        ShardSampler epoch).  Per epoch: train phase (model.train(), every batch one ``step``), StepLR.step(), the
        exponential-average guard on the epoch loss (:522-527), then the evaluation phase (model.eval(), no gradients)
        whose mean score selects the best weights (:531-535; kept as a copy of the flat buffer, not a deep-copied state
        dict).  ``evaluate(outputs, batch) -> device tensor score`` (e.g. a metrics.MetricsLog); host reads happen
        once per phase: the non-finite flag, the epoch loss, the mean score.  Returns dict(score, epoch, history)."""
        best = dict(score=-10000.0, epoch=-1)
        best_wts = self.flat.data.clone()
        best_bufs = {k: v.clone() for k, v in self.model.named_buffers()}
        e_avg = ExpAverage(.01)
        history = []
        for epoch in range(n_epochs):
            self.model.train()
            self.loss_sum.zero_()
            n_steps = 0
            for batch in train_batches(epoch):
                self.step(batch)
                n_steps += 1
            self.check_health()
            bad = int(self.nan_flag.item())                        # one read per epoch (bit 0 NaN, bit 1 infinity)
            if bad:
                # The optimizer kernels stopped updating at the offending batch, so the weights are the last finite
                # ones -- what the reference returns when it breaks out at `str(loss.item()) == "nan"` (:468-470, :542).
                # An infinite loss ends the run here too (the reference would run one more batch into the NaN).
                history.append(dict(epoch=epoch, terminated="NaN loss" if bad & 1 else "infinite loss"))
                break
            ep_loss = float(self.loss_sum.item()) / max(n_steps, 1)
            e_avg.update(ep_loss)
            terminate = epoch_ckpt[0] > 1 and epoch % (epoch_ckpt[0] - 1) == 0 and epoch > 0 and e_avg.value > epoch_ckpt[1]
            self.scheduler.step()
            rec = dict(epoch=epoch, loss=ep_loss, lr=self.optimizer.lr)
            if eval_batches is not None and evaluate is not None:
                self.model.eval()
                scores = []
                with torch.no_grad():
                    for batch in eval_batches(epoch):
                        outputs = self.model(batch.model_input())
                        scores.append(evaluate(outputs * batch.w, batch).reshape(1))
                mean_score = float(torch.cat(scores).double().mean().item()) if scores else float("nan")
                rec["score"] = mean_score
                if best["score"] < mean_score:
                    best.update(score=mean_score, epoch=epoch)
                    best_wts.copy_(self.flat.data)
                    best_bufs = {k: v.clone() for k, v in self.model.named_buffers()}
            history.append(rec)
            if log is not None:
                log(rec)
            if terminate:
                break
        if best["epoch"] >= 0:
            self.flat.data.copy_(best_wts)
            for k, v in self.model.named_buffers():
                v.copy_(best_bufs[k])
        best["history"] = history
        return best

    def check_health(self):
        """Off the hot loop (the reference checks its loss every batch, train_joint_attn.py:468-470; here once per epoch
        or at the end of a run): raises if the nonsat kernel's barrier ever gave up (one device->host read)."""
        F_.NonsatFunction.raise_if_timed_out()

    def sync_buffers(self):
        """BatchNorm running statistics are per-rank (each rank normalises its own shard, as the reference would on that
        shard).  Before a checkpoint is written they are averaged over the ranks, so the saved state does not depend on
        which rank saves it (num_batches_tracked is identical on every rank)."""
        if not self.collective or self.world_size == 1:
            return
        for name, buf in self.model.named_buffers():
            if buf.is_floating_point():
                self.reduce(buf.data)
                buf.data.div_(self.world_size)

    def forward_loss(self, batch):
        self._grads_clean = False               # a backward of this loss will write into the flat gradient buffer
        if self.sync is None:
            outputs = self.model(batch.model_input())
        else:                                   # same module sequence, with a watch on the encoders' outputs
            kids = list(self.model.children())
            outputs = kids[0](batch.model_input())
            self.sync.watch(outputs)
            for m in kids[1:]:
                outputs = m(outputs)
        return F_.MSELossFunction.apply(outputs, batch.w, batch.y), outputs

    def step(self, batch):
        # optimizer.zero_grad() (train_joint_attn.py:428): the optimizer kernel of the previous step cleared the flat
        # buffer after consuming it; a fill is only needed when something else ran a backward in between
        if not getattr(self, "_grads_clean", False):
            self.flat.zero_grad()
        loss, outputs = self.forward_loss(batch)
        loss.backward()
        scale = 1.0
        if self.collective:
            if self.sync is not None:
                self.sync.finish()                                            # RCCL sums; mean via scale
            else:
                self.reduce(self.flat.grad)
            scale = 1.0 / self.world_size
        # fail-fast flag and epoch loss stay on the device (train_joint_attn.py:468-470, :489, :523)
        lv = loss.detach().reshape(1)
        _lib.call("jova_nonfinite_flag", _lib.P(lv), 1, _lib.P(self.nan_flag), _lib.stream())
        _lib.call("jova_axpy", _lib.P(lv), 1.0, 1, _lib.P(self.loss_sum), _lib.stream())
        self.optimizer.step(scale)
        self._grads_clean = bool(self.optimizer.zero_grads)
        return loss, outputs
