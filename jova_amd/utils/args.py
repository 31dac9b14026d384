"""Argument wrappers of the reference's model builders (jova/utils/args.py:19-108), the ones the weave view uses."""
import torch.nn as nn


class Args(object):
    def __init__(self, activation, batch_norm, dropout):
        from ..nn.layers import ReLU
        from ..nn.models import NonsatActivation
        self._args = None
        if activation:
            activation = {'relu': ReLU(), 'leaky_relu': nn.LeakyReLU(.2), 'sigmoid': nn.Sigmoid(), 'tanh': nn.Tanh(),
                          'softmax': nn.Softmax(dim=1), 'elu': nn.ELU(),
                          'nonsat': NonsatActivation()}.get(activation.lower(), ReLU())
        self._activation = activation
        self._batch_norm = batch_norm
        self._dropout = dropout

    def __getitem__(self, i):
        return self._args[i]

    @property
    def args(self):
        return self._args

    @property
    def activation_function(self):
        return self._activation

    @property
    def use_batch_norm(self):
        return self._batch_norm

    @property
    def dropout(self):
        return self._dropout


class WeaveLayerArgs(Args):
    def __init__(self, n_atom_input_feat=75, n_pair_input_feat=14, n_atom_output_feat=50, n_pair_output_feat=50,
                 n_hidden_AA=50, n_hidden_PA=50, n_hidden_AP=50, n_hidden_PP=50, update_pair=True, activation=None,
                 batch_norm=False, dropout=-1):
        super(WeaveLayerArgs, self).__init__(activation, batch_norm, dropout)
        self._args = (n_atom_input_feat, n_pair_input_feat, n_atom_output_feat, n_pair_output_feat,
                      n_hidden_AA, n_hidden_PA, n_hidden_AP, n_hidden_PP, update_pair, self.activation_function)


class WeaveGatherArgs(Args):
    def __init__(self, conv_out_depth, n_depth=128, gaussian_expand=False, activation='tanh', epsilon=1e-7):
        super(WeaveGatherArgs, self).__init__(activation, False, -1)
        self._args = (conv_out_depth, n_depth, gaussian_expand, self.activation_function, epsilon)
