"""HIP-backed counterparts of the reference's hot-path models, same names / signatures / state-dict keys as
/root/reference/jova/nn/models.py: GraphConvSequential :331-349, NonsatActivation :448-478, NwayForward :601-611,
Prot2Vec :685-715 (+ _construct_embedding_indices :214-242), ProteinCNN / ProteinCNN2D :718-771,
JointAttention :909-973, SegmentWiseLinear :976-1000, SegmentWiseResBlock :1003-1018, GraphNeuralNet /
GraphNeuralNet2D :352-418.
"""
import numpy as np
import torch
import torch.nn as nn

from . import functional as F_
from .layers import (BatchNorm1d, Dropout, GraphConvLayer, GraphGather, Linear, ReLU, WeaveBatchNorm, WeaveDropout,
                     WeaveGather, WeaveGather2D, WeaveLayer, get_activation_func)


class NonsatActivation(nn.Module):
    def __init__(self, ep=1e-4, max_iter=100):
        super(NonsatActivation, self).__init__()
        self.ep = ep
        self.max_iter = max_iter

    def forward(self, x):
        return nonsat_activation(x, self.ep, self.max_iter)


def nonsat_activation(x, ep=1e-4, max_iter=100):
    return F_.NonsatFunction.apply(x, ep, max_iter)


class GraphConvSequential(nn.Sequential):
    """Threads ``input[0]`` (atom features) through the modules; the topology entries stay fixed (models.py:336-349).

    Peephole fusions that keep results identical: a GraphConvLayer directly followed by BatchNorm1d hands it the
    batch statistics computed in the convolution's own epilogue, and the BatchNorm backward applies the backward of
    the convolution's built-in ReLU (it reads that output anyway), so the convolution's backward receives dZ
    directly; BatchNorm1d directly followed by ReLU applies it in the same kernel.
    """

    def forward(self, input):
        batch_size = input[1]
        input = list(input[0])
        mods = list(self._modules.values())
        pending_stats = None
        pending_relu_out = None                     # output of a ReLU-activated conv that the next BatchNorm consumes
        k = 0
        while k < len(mods):
            module = mods[k]
            nxt = mods[k + 1] if k + 1 < len(mods) else None
            if isinstance(module, BatchNorm1d):
                fuse = isinstance(nxt, nn.ReLU)
                module.fused_relu = fuse
                fuse_in = pending_relu_out is not None and module.training and pending_relu_out is input[0] and \
                    getattr(input[0].grad_fn, "__class__", None).__name__ == "GraphConvFunctionBackward"
                try:
                    input[0] = module(input[0], pending_stats if module.training else None, input_relu=fuse_in)
                finally:
                    module.fused_relu = False
                if fuse_in:
                    pending_relu_out.grad_fn.mask_fused = True
                pending_stats = None
                pending_relu_out = None
                if fuse:
                    k += 1                          # the ReLU ran inside the BatchNorm kernel
            elif isinstance(module, (nn.BatchNorm1d, nn.Dropout, nn.Linear, nn.ReLU, NonsatActivation)):
                input[0] = module(input[0])
            elif isinstance(module, GraphGather):
                input[0] = module(input, batch_size)
            elif isinstance(module, GraphConvLayer) and isinstance(nxt, BatchNorm1d) and nxt.training and \
                    (module.activation is None or isinstance(module.activation, ReLU)):
                pending_stats = torch.empty((2, module.out_dim), dtype=torch.float64, device=input[0].device)
                input[0] = module(input, bn_stats=pending_stats)
                if isinstance(module.activation, ReLU):
                    pending_relu_out = input[0]
            else:
                input[0] = module(input)
            k += 1
        return input[0]


class GraphNeuralNet(nn.Module):
    """The GNN view encoder (jova/nn/models.py:352-395; Tsubaki et al.): fingerprint embedding, `num_layers` updates
    x <- x + A relu(W x + b) / (M - 1), then a per-molecule sum or mean.  Same constructor, parameters and state-dict
    keys as the reference; embedding, Linear + ReLU, the A-product and the residual run on the HIP kernels."""

    def __init__(self, num_fingerprints, embedding_dim, num_layers=3, update='mean', activation='relu'):
        super(GraphNeuralNet, self).__init__()
        if activation != 'relu':
            raise NotImplementedError("the HIP GraphNeuralNet implements the reference's default 'relu'")
        self.output = update
        self.activation = get_activation_func(activation)
        self.embed_fingerprint = nn.Embedding(num_fingerprints, int(embedding_dim))
        self.W_gnn = nn.ModuleList([nn.Linear(embedding_dim, embedding_dim) for _ in range(num_layers)])

    def _node_vectors(self, inputs):
        fingerprints, adjacency_matrices, M, axis = inputs
        F_._lib.require_gpu(fingerprints, M)
        if not isinstance(adjacency_matrices, F_.BlockAdjacency):
            F_._lib.require_gpu(adjacency_matrices)
        elif not adjacency_matrices.is_cuda:
            raise RuntimeError("jova_amd: tensors must live on the GPU (the HIP path has no CPU fallback)")
        x = F_.EmbeddingFunction.apply(self.embed_fingerprint.weight, fingerprints)
        for lin in self.W_gnn:
            hs = F_.LinearFunction.apply(x, lin.weight, lin.bias, True)
            x = F_.GnnUpdateFunction.apply(x, hs, adjacency_matrices, None if self.output == 'sum' else M)
        return x, [int(a) for a in axis]

    def forward(self, inputs):
        x, axis = self._node_vectors(inputs)
        # per-molecule reduction through the zero-padded [A_max, B, d] image (sum over the segment axis; 'mean'
        # divides by the molecule's size)
        padded = F_.Gather2DFunction.apply(x, _AxisTopology(axis, x.device))
        vec = padded.sum(0)
        if self.output != 'sum':
            vec = vec / torch.tensor(axis, dtype=vec.dtype, device=vec.device).unsqueeze(1)
        return vec


class GraphNeuralNet2D(GraphNeuralNet):
    """-> [number of segments (A_max), batch size, dimension] (jova/nn/models.py:398-418): the node vectors of every
    molecule, zero padded to the longest one, for the joint-attention block."""

    def forward(self, inputs):
        x, axis = self._node_vectors(inputs)
        return F_.Gather2DFunction.apply(x, _AxisTopology(axis, x.device))


class _AxisTopology(object):
    def __init__(self, axis, device):
        self.n_atoms_lst = list(axis)
        off = np.concatenate([[0], np.cumsum(axis)]).astype(np.int32)
        self.mol_off = torch.from_numpy(off).to(device, non_blocking=True)


def create_weave_layers(weave_args, update_pair):
    """jova/nn/models.py:164-179."""
    layers = []
    for weave_arg in weave_args:
        layers.append(WeaveLayer(*weave_arg.args))
        if weave_arg.use_batch_norm:
            layers.append(WeaveBatchNorm(atom_dim=weave_arg[2], pair_dim=weave_arg[3], update_pair=update_pair))
        if weave_arg.dropout > 0:
            layers.append(WeaveDropout(weave_arg.dropout, update_pair=update_pair))
    return layers


class WeaveSequential(nn.Sequential):
    """jova/nn/models.py:282-307: threads (A, P) through the weave layers, hands [A, P, atom_split, n_atoms_lst] to the
    gather layer."""

    def forward(self, input):
        input = list(input)
        A = P = None
        for module in self._modules.values():
            if A is not None:
                input[0] = A
            if P is not None:
                input[1] = P
            if isinstance(module, (WeaveBatchNorm, WeaveDropout)):
                A, P = module(A, P)
            elif isinstance(module, WeaveGather):
                return module([A, P, *input[4:]])
            else:
                A, P = module(input)
        return A, P


class WeaveModel(nn.Module):
    """jova/nn/models.py:245-279."""

    def __init__(self, weave_args, weave_gath_arg, update_pair=False, weave_type='1D', batch_first=False):
        super(WeaveModel, self).__init__()
        self.update_pair = update_pair
        layers = create_weave_layers(weave_args, update_pair)
        if weave_gath_arg:
            if weave_type.lower() == '1d':
                layers.append(WeaveGather(*weave_gath_arg.args))
            else:
                layers.append(WeaveGather2D(*weave_gath_arg, batch_first=batch_first))
        self.weave_seq = WeaveSequential(*layers)

    def forward(self, input, need_pair_feat=False):
        output = self.weave_seq(input)
        if need_pair_feat:
            return output
        return output[0]


class NwayForward(nn.Module):
    def __init__(self, models):
        super(NwayForward, self).__init__()
        self.models = nn.ModuleList(models)

    def forward(self, inputs):
        return [model(inputs[i]) for i, model in enumerate(self.models)]


class ProteinTable(object):
    """All protein profiles of a ``protein_profile`` dict resident in HBM as one int32 table [rows, window]."""

    def __init__(self, protein_profile, device, vocab_size=None):
        self.names = list(protein_profile.keys())
        self.row_of = {}
        rows, off = [], 0
        self.window = None
        for name in self.names:
            prof = np.asarray(protein_profile[name])
            if prof.ndim != 2:
                raise ValueError("protein profile %r must be [n_segments, window]" % (name,))
            self.window = prof.shape[1] if self.window is None else self.window
            self.row_of[name] = (off, prof.shape[0])
            rows.append(prof.astype(np.int32))
            off += prof.shape[0]
        table = np.concatenate(rows, axis=0) if rows else np.zeros((0, 1), np.int32)
        if vocab_size is not None and table.size and (table.min() < 0 or table.max() > vocab_size):
            # nn.Embedding raises on such an index at lookup time (models.py:704-715); the kernels index the table
            # without a test per element, so the table is checked once, here
            raise IndexError("protein profile holds word ids outside [0, %d] (min %d, max %d): index out of range"
                             % (vocab_size, int(table.min()), int(table.max())))
        self.table = torch.from_numpy(np.ascontiguousarray(table)).to(device)

    def batch_meta(self, prot_x):
        """[(source, name), ...] -> host int32 [B, 2] = (first table row, segment count) of each pair's protein."""
        return np.array([self.row_of[prot[1]] for prot in prot_x], dtype=np.int32).reshape(len(prot_x), 2)

    def batch(self, prot_x, device):
        """[(source, name), ...] -> (row offsets int32 [B], segment counts int32 [B], P_max)."""
        meta = self.batch_meta(prot_x)
        dev = torch.from_numpy(meta).to(device, non_blocking=True)
        return dev[:, 0].contiguous(), dev[:, 1].contiguous(), int(meta[:, 1].max())


def _construct_embedding_indices(prot_x, protein_profile, device, fill_val):
    """The reference's padded index cube [B, P_max, window] int64 (models.py:214-242), built on the GPU.
    Not used by Prot2Vec.forward (the kernel derives indices from the resident table); kept for callers/tests."""
    tab = ProteinTable({p[1]: protein_profile[p[1]] for p in prot_x}, device)
    off, ln, p_max = tab.batch(prot_x, device)
    dummy = torch.zeros((fill_val + 1, 1), device=device)
    out = torch.empty((p_max, len(prot_x), tab.window), device=device)
    idx = torch.empty((len(prot_x), p_max, tab.window), dtype=torch.int64, device=device)
    F_.call("jova_prot2vec_fwd", F_.P(tab.table), F_.P(off), F_.P(ln), F_.P(dummy), len(prot_x), p_max, tab.window, 1,
            fill_val, 0, F_.P(out), F_.P(idx), F_.stream())
    return idx


class Prot2Vec(nn.Module):
    def __init__(self, protein_profile, vocab_size, embedding_dim, activation='relu', batch_first=False):
        super(Prot2Vec, self).__init__()
        self._batch_first = batch_first
        self.protein_profile = protein_profile
        self.embedding = nn.Embedding(vocab_size + 1, int(embedding_dim), padding_idx=vocab_size)
        self.activation = get_activation_func(activation)
        self._table = None

    def _resident_table(self, device):
        if self._table is None or self._table.table.device != device:
            self._table = ProteinTable(self.protein_profile, device, vocab_size=self.embedding.num_embeddings - 1)
        return self._table

    def forward(self, input):
        w = self.embedding.weight
        tab = self._resident_table(w.device)
        off, ln, p_max = input if isinstance(input, tuple) and torch.is_tensor(input[0]) else tab.batch(input, w.device)
        fused_relu = isinstance(self.activation, ReLU)
        emb = F_.Prot2VecFunction.apply(w, tab.table, off, ln, p_max, tab.window, fused_relu)
        if not fused_relu:
            emb = self.activation(emb)
        if self._batch_first:
            emb = emb.permute(1, 0, 2)
        return emb


class ProteinRNN(nn.Module):
    """LSTM protein encoder (jova/nn/models.py:657-682): same constructor and state-dict keys (``model.weight_ih_l0`` ...);
    the HIP path covers what the DTI driver builds (train_joint_attn.py:77-86): one layer, one direction, hidden <= 64."""

    def __init__(self, in_dim, hidden_dim, dropout, num_layers=1, bidrectional=False, activation='relu',
                 batch_first=False):
        super(ProteinRNN, self).__init__()
        self.hidden_dim = int(hidden_dim)
        self.num_layers = int(num_layers)
        self.directions = max(1, int(bidrectional) + 1)
        if self.num_layers != 1 or self.directions != 1:
            raise NotImplementedError("the HIP ProteinRNN implements one layer, one direction (the DTI driver's use)")
        if activation != 'relu':
            raise NotImplementedError("the HIP ProteinRNN implements the reference's default 'relu'")
        self.activation = get_activation_func(activation)
        self.batch_first = batch_first
        if num_layers == 1:
            dropout = 0
        self.model = nn.LSTM(input_size=int(in_dim), hidden_size=int(self.hidden_dim), num_layers=int(self.num_layers),
                             batch_first=batch_first, dropout=dropout, bidirectional=bidrectional)

    def forward(self, x):
        if self.batch_first:
            x = x.permute(1, 0, 2)
        m = self.model
        out = F_.LstmFunction.apply(x, m.weight_ih_l0, m.weight_hh_l0, m.bias_ih_l0, m.bias_hh_l0, True)
        return out.permute(1, 0, 2) if self.batch_first else out


class ProteinCNN(nn.Module):
    def __init__(self, dim, window, activation='relu', num_layers=2, pooling_dim=1):
        super(ProteinCNN, self).__init__()
        self.activation = get_activation_func(activation)
        self.pooling_dim = pooling_dim
        self.lin_kernels = nn.ModuleList([Linear(dim * window, dim * window) for _ in range(num_layers - 1)])
        self.lin_kernels.append(Linear(dim * window, dim))

    def _convolve(self, prot_x):
        fused = isinstance(self.activation, ReLU)
        if fused and prot_x.is_cuda and F_.ProteinCnnFunction.usable(prot_x, [k.weight for k in self.lin_kernels]):
            params = []
            for kernel in self.lin_kernels:
                params += [kernel.weight, kernel.bias]
            return F_.ProteinCnnFunction.apply(prot_x, *params)
        for kernel in self.lin_kernels:
            if fused:
                prot_x = F_.LinearFunction.apply(prot_x, kernel.weight, kernel.bias, True)
            else:
                prot_x = self.activation(kernel(prot_x))
        return prot_x

    def forward(self, prot_x):
        return torch.mean(self._convolve(prot_x), dim=self.pooling_dim, keepdim=True)


class ProteinCNN2D(ProteinCNN):
    def forward(self, prot_x):
        return self._convolve(prot_x)


class SegmentWiseLinear(nn.Module):
    def __init__(self, d_model, activation='relu', inner_layer_dim=2048):
        super(SegmentWiseLinear, self).__init__()
        self.activation = get_activation_func(activation)
        self.linear1 = Linear(d_model, inner_layer_dim)
        self.linear2 = Linear(inner_layer_dim, d_model)

    def forward(self, x):
        if isinstance(self.activation, ReLU):
            return self.linear2(F_.LinearFunction.apply(x, self.linear1.weight, self.linear1.bias, True))
        return self.linear2(self.activation(self.linear1(x)))


class SegmentWiseResBlock(nn.Module):
    def __init__(self, d_model, activation='relu', inner_layer_dim=2048):
        super(SegmentWiseResBlock, self).__init__()
        self.normalize = nn.LayerNorm(d_model)
        self.seg_lin = SegmentWiseLinear(d_model, activation, inner_layer_dim)

    def forward(self, x):
        # stand-alone use only: inside JointAttention the whole block runs in JointAttentionFunction
        return self.normalize(x + self.seg_lin(x))


class JointAttention(nn.Module):
    """Joint-view attention fusion (models.py:909-973).  The per-view projections, the MultiheadAttention /
    LayerNorm / FFN layers and the view pooling run as ONE autograd node over the HIP kernels; the stock torch
    modules below are parameter containers only (same names, shapes and initialisation as the reference)."""

    def __init__(self, d_dims, latent_dim=256, num_heads=1, num_layers=1, dprob=0., activation='relu',
                 inner_layer_dim=2048, attn_hook=None):
        super(JointAttention, self).__init__()
        if attn_hook:
            assert callable(attn_hook), "Attention hook must be a function"
        if activation != 'relu':
            raise NotImplementedError("the fused joint-attention path implements the reference's default 'relu'")
        self.attn_hook = attn_hook
        self.latent_dim = latent_dim
        self.num_heads = num_heads
        self.dprob = dprob
        self.lin_prjs = nn.ModuleList([nn.Linear(in_dim, latent_dim) for in_dim in d_dims])
        self.attention_models = nn.ModuleList([nn.MultiheadAttention(embed_dim=latent_dim, num_heads=num_heads,
                                                                     dropout=dprob) for _ in range(num_layers)])
        self.res_blks = nn.ModuleList([SegmentWiseResBlock(d_model=latent_dim, activation=activation,
                                                           inner_layer_dim=inner_layer_dim) for _ in range(num_layers)])
        self.activation = get_activation_func(activation)
        self.dropout = Dropout(dprob)
        self.batch_norm = BatchNorm1d(latent_dim * len(d_dims))
        self.attn_norm = nn.LayerNorm(latent_dim)

    def forward(self, inputs):
        params = []
        for lin in self.lin_prjs:
            params += [lin.weight, lin.bias]
        params += [self.attn_norm.weight, self.attn_norm.bias]
        for attn, res in zip(self.attention_models, self.res_blks):
            params += [attn.in_proj_weight, attn.in_proj_bias, attn.out_proj.weight, attn.out_proj.bias,
                       res.seg_lin.linear1.weight, res.seg_lin.linear1.bias,
                       res.seg_lin.linear2.weight, res.seg_lin.linear2.bias,
                       res.normalize.weight, res.normalize.bias]
        x = F_.JointAttentionFunction.apply(len(inputs), self.num_heads, len(self.attention_models), self.dprob,
                                            self.training, self.attn_hook, *(list(inputs) + params))
        self.batch_norm.fused_relu = True
        try:
            x = self.batch_norm(x)
        finally:
            self.batch_norm.fused_relu = False
        return self.dropout(x)
