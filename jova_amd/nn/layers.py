"""HIP-backed counterparts of the reference's graph layers, same names / constructor and forward signatures /
state-dict keys as /root/reference/jova/nn/layers.py (GraphConvLayer :264-347, GraphPool :350-393,
GraphGather2D :419-437, Unsqueeze :550-556), plus HIP-backed drop-ins for the stock torch modules the reference's
model builders put between them (nn.Linear, nn.BatchNorm1d, nn.ReLU, nn.Dropout; proj/dti/train_joint_attn.py:155-171).
"""
import torch
import torch.nn as nn

from . import functional as F_


def get_activation_func(activation):
    """Same mapping as jova/utils/train_helpers.py:268-276, with HIP-backed modules where the hot path uses them."""
    from .models import NonsatActivation
    table = {'relu': ReLU(), 'leaky_relu': nn.LeakyReLU(.2), 'sigmoid': nn.Sigmoid(), 'tanh': nn.Tanh(),
             'softmax': nn.Softmax(), 'elu': nn.ELU(), 'nonsat': NonsatActivation()}
    return table.get(activation.lower(), ReLU())


class Layer(nn.Module):
    """Base class of the graph layers (layers.py:58-69)."""

    def __init__(self, activation=None):
        super(Layer, self).__init__()
        if activation and isinstance(activation, str):
            self.activation = get_activation_func(activation)
        else:
            self.activation = activation


class GraphConvLayer(Layer):
    """Degree-bucketed graph convolution; parameters ``linear_layers.{0..20}.{weight,bias}`` (layers.py:264-280)."""

    def __init__(self, in_dim, out_dim, min_deg=0, max_deg=10, activation='relu'):
        super(GraphConvLayer, self).__init__(activation)
        if min_deg != 0 or max_deg != 10:
            raise NotImplementedError("the HIP kernels are built for min_deg=0, max_deg=10 (the reference's defaults)")
        self.in_dim = in_dim
        self.out_dim = out_dim
        self.min_degree = min_deg
        self.max_degree = max_deg
        self.num_deg = 2 * max_deg + (1 - min_deg)
        self.linear_layers = nn.ModuleList([nn.Linear(in_dim, out_dim) for _ in range(self.num_deg)])

    def forward(self, input_data, bn_stats=None):
        """``bn_stats`` (optional fp64 [2, out_dim]) receives the column sum / sum of squares of the output from the
        same launch, for a BatchNorm1d that follows (GraphConvSequential passes it)."""
        atoms = input_data[0]
        topo = F_.topology_of(input_data)
        fused_relu = isinstance(self.activation, ReLU)
        stats = bn_stats if (fused_relu or self.activation is None) else None
        wb = [l.weight for l in self.linear_layers] + [l.bias for l in self.linear_layers]
        out = F_.GraphConvFunction.apply(atoms, topo, fused_relu, stats, *wb)
        if self.activation is not None and not fused_relu:
            out = self.activation(out)
        return out


class GraphPool(Layer):
    def __init__(self, min_degree=0, max_degree=10):
        super(GraphPool, self).__init__()
        self.min_degree = min_degree
        self.max_degree = max_degree

    def forward(self, input_data):
        return F_.GraphPoolFunction.apply(input_data[0], F_.topology_of(input_data))


class GraphGather(Layer):
    """1-D gather of the reference (layers.py:396-416) is broken upstream (NameError: scatter_add); kept as the
    isinstance anchor GraphConvSequential dispatches on."""

    def __init__(self, activation='tanh'):
        super(GraphGather, self).__init__(activation)

    def forward(self, input_data, batch_size):
        raise NotImplementedError("GraphGather (1-D) raises NameError in the reference too; use GraphGather2D")


class GraphGather2D(GraphGather):
    def __init__(self, activation='tanh', batch_first=False):
        super(GraphGather2D, self).__init__(activation)
        self._batch_first = batch_first

    def forward(self, input_data, batch_size):
        atoms, n_atoms_lst = input_data[0], input_data[3]
        topo = getattr(input_data[1], "_jova_topology", None) if torch.is_tensor(input_data[1]) else None
        if topo is None or topo.n_atoms_lst != [int(v) for v in n_atoms_lst]:
            topo = _GatherOnlyTopology(n_atoms_lst, atoms.device)
        outputs = F_.Gather2DFunction.apply(atoms, topo)
        mol_features = self.activation(outputs) if self.activation else outputs
        if self._batch_first:
            mol_features = mol_features.permute(1, 0, 2)
        return mol_features


class _GatherOnlyTopology(object):
    def __init__(self, n_atoms_lst, device):
        self.n_atoms_lst = [int(v) for v in n_atoms_lst]
        off = [0]
        for n in self.n_atoms_lst:
            off.append(off[-1] + n)
        self.mol_off = torch.tensor(off, dtype=torch.int32).to(device, non_blocking=True)


class Unsqueeze(nn.Module):
    def __init__(self, dim):
        super(Unsqueeze, self).__init__()
        self.dim = dim

    def forward(self, x):
        return x.unsqueeze(dim=self.dim)


# ------------------------------------------------------------------ HIP-backed stock modules
class Linear(nn.Linear):
    """nn.Linear whose forward/backward run the fp32 MFMA GEMM."""
    fused_relu = False

    def forward(self, x):
        return F_.LinearFunction.apply(x, self.weight, self.bias, self.fused_relu)


class BatchNorm1d(nn.BatchNorm1d):
    """nn.BatchNorm1d over [rows, C] (training: batch statistics + running-stat update; eval: running statistics)."""
    fused_relu = False

    def forward(self, x, stats=None, input_relu=False):
        """``input_relu``: x is a ReLU output that only this layer consumes; the training-mode backward then also
        applies that ReLU's backward (GraphConvSequential's conv -> BatchNorm peephole)."""
        if x.dim() != 2:
            raise ValueError("expected 2D input (got %dD input)" % x.dim())
        if self.training or not self.track_running_stats:
            mom = 0.1 if self.momentum is None else self.momentum
            rm = self.running_mean if self.track_running_stats else None
            rv = self.running_var if self.track_running_stats else None
            nb = self.num_batches_tracked if self.track_running_stats else None
            return F_.BatchNormFunction.apply(x, self.weight, self.bias, rm, rv, nb, self.eps, mom, self.fused_relu,
                                              stats, input_relu)
        return F_.batch_norm_eval(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps,
                                  self.fused_relu)


class ReLU(nn.ReLU):
    def forward(self, x):
        return F_.ReLUFunction.apply(x)


class Dropout(nn.Dropout):
    def forward(self, x):
        if not self.training or self.p == 0.:
            return x
        return F_.DropoutFunction.apply(x, self.p)


class ElementwiseBatchNorm(nn.Module):
    """BatchNorm over the last dimension of a tensor of any rank (jova/nn/layers.py:574-588): the tensor is viewed as
    [-1, dim]; the statistics and the normalisation run on the HIP BatchNorm1d."""

    def __init__(self, dim):
        super(ElementwiseBatchNorm, self).__init__()
        self.batch_norm = BatchNorm1d(dim)

    def forward(self, x):
        shape = x.shape
        x = self.batch_norm(x.contiguous().view(-1, shape[-1]))
        return x.view(*shape)


# ------------------------------------------------------------------ Weave view (jova/nn/layers.py:72-262), section 8(f3)
def _act(module, x):
    return module(x) if module else x


def _lin(layer, x, activation):
    """Linear followed by the layer's activation; Linear + ReLU is one fused GEMM epilogue."""
    if isinstance(activation, (ReLU, nn.ReLU)):
        return F_.LinearFunction.apply(x, layer.weight, layer.bias, True)
    return _act(activation, F_.LinearFunction.apply(x, layer.weight, layer.bias, False))


class WeaveLayer(Layer):
    """Weave convolution (layers.py:72-153): same constructor, parameters and forward contract.  The reference calls
    torch_scatter.scatter_add, whose import it has commented out (layers.py:17, :135): its published semantics
    (out[index[r]] += src[r]) are what jova_scatter_add_rows implements."""

    def __init__(self, n_atom_input_feat=75, n_pair_input_feat=14, n_atom_output_feat=50, n_pair_output_feat=50,
                 n_hidden_AA=50, n_hidden_PA=50, n_hidden_AP=50, n_hidden_PP=50, update_pair=True, activation='relu'):
        super(WeaveLayer, self).__init__(activation)
        self.update_pair = update_pair
        self.n_hidden_AA, self.n_hidden_PA = n_hidden_AA, n_hidden_PA
        self.n_hidden_AP, self.n_hidden_PP = n_hidden_AP, n_hidden_PP
        self.n_hidden_A = n_hidden_AA + n_hidden_PA
        self.n_hidden_P = n_hidden_AP + n_hidden_PP
        self.n_atom_input_feat, self.n_pair_input_feat = n_atom_input_feat, n_pair_input_feat
        self.n_atom_output_feat, self.n_pair_output_feat = n_atom_output_feat, n_pair_output_feat
        self.linear_AA = Linear(self.n_atom_input_feat, self.n_hidden_AA)
        self.linear_PA = Linear(self.n_pair_input_feat, self.n_hidden_PA)
        self.linear_A = Linear(self.n_hidden_A, self.n_atom_output_feat)
        if self.update_pair:
            self.linear_AP = Linear(self.n_atom_input_feat * 2, self.n_hidden_AP)
            self.linear_PP = Linear(self.n_pair_input_feat, self.n_hidden_PP)
            self.linear_P = Linear(self.n_hidden_P, self.n_pair_output_feat)

    def forward(self, input_data):
        atom_features, pair_features, pair_split, atom_to_pair, _, _ = input_data
        n_atoms = atom_features.shape[0]
        AA = _lin(self.linear_AA, atom_features, self.activation)
        PA = _lin(self.linear_PA, pair_features, self.activation)
        PA = F_.RowScatterAddFunction.apply(PA, pair_split, n_atoms)
        A = _lin(self.linear_A, torch.cat([AA, PA], dim=1), self.activation)
        if self.update_pair:
            f2 = 2 * self.n_atom_input_feat
            ij = F_.RowGatherFunction.apply(atom_features, atom_to_pair).view(-1, f2)
            ji = F_.RowGatherFunction.apply(atom_features, atom_to_pair.flip([1])).view(-1, f2)
            AP_ij = _lin(self.linear_AP, ij, self.activation)
            AP_ji = _lin(self.linear_AP, ji, self.activation)
            PP = _lin(self.linear_PP, pair_features, self.activation)
            P = _lin(self.linear_P, torch.cat([AP_ij + AP_ji, PP], dim=1), self.activation)
        else:
            P = pair_features
        return A, P


class WeaveGather(Layer):
    """Final layer of the weave convolution (layers.py:156-207): gaussian expansion, per-molecule sum, Linear + activation."""

    def __init__(self, conv_out_depth, n_depth=128, gaussian_expand=False, activation='tanh', epsilon=1e-7):
        super(WeaveGather, self).__init__(activation)
        self.n_depth = n_depth
        self.conv_out_depth = conv_out_depth
        self.gaussian_expand = gaussian_expand
        self.epsilon = epsilon
        self.gaussian_memberships = [(-1.645, 0.283), (-1.080, 0.170), (-0.739, 0.134), (-0.468, 0.118),
                                     (-0.228, 0.114), (0., 0.114), (0.228, 0.114), (0.468, 0.118), (0.739, 0.134),
                                     (1.080, 0.170), (1.645, 0.283)]
        self.linear = Linear(self.conv_out_depth * len(self.gaussian_memberships), self.n_depth)

    def gaussian_histogram(self, x):
        return F_.GaussHistFunction.apply(x, self.epsilon)

    def forward(self, input_data):
        outputs, pair_features, atom_split, _ = input_data
        if self.gaussian_expand:
            outputs = self.gaussian_histogram(outputs)
        n_mols = int(atom_split.max().item()) + 1 if atom_split.numel() else 0
        output_molecules = F_.RowScatterAddFunction.apply(outputs, atom_split, n_mols)
        if self.gaussian_expand:
            output_molecules = _lin(self.linear, output_molecules, self.activation)
        return output_molecules, pair_features


class WeaveGather2D(WeaveGather):
    """-> ([A_max, B, n_depth], pair features) (layers.py:210-232): per-atom gaussian expansion + Linear + activation, then the
    molecules zero padded to the longest and stacked (jova_gather2d)."""

    def __init__(self, *args, batch_first, **kwargs):
        super(WeaveGather2D, self).__init__(*args, **kwargs)
        self._batch_first = batch_first

    def forward(self, input_data):
        outputs, pair_features, atom_split, n_atoms_lst = input_data
        if self.gaussian_expand:
            outputs = self.gaussian_histogram(outputs)
            outputs = _lin(self.linear, outputs, self.activation)
        outputs = F_.Gather2DFunction.apply(outputs, _GatherOnlyTopology(n_atoms_lst, outputs.device))
        if self._batch_first:
            outputs = outputs.permute(1, 0, 2)
        return outputs, pair_features


class WeaveDropout(Layer):
    def __init__(self, prob, update_pair):
        super(WeaveDropout, self).__init__()
        self.update_pair = update_pair
        self.atom_dropout = Dropout(prob)
        if update_pair:
            self.pair_dropout = Dropout(prob)

    def forward(self, atom_input, pair_input):
        return self.atom_dropout(atom_input), self.pair_dropout(pair_input) if self.update_pair else pair_input


class WeaveBatchNorm(Layer):
    def __init__(self, atom_dim, update_pair, pair_dim, activation=None):
        super(WeaveBatchNorm, self).__init__(activation)
        self.update_pair = update_pair
        self.a_batchnorm = BatchNorm1d(atom_dim)
        if update_pair:
            self.p_batchnorm = BatchNorm1d(pair_dim)

    def forward(self, atom_input, pair_input):
        return self.a_batchnorm(atom_input), self.p_batchnorm(pair_input) if self.update_pair else pair_input
