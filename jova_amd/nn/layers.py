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
