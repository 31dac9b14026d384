"""Parameter bags the weave builders take (contract: jova/utils/args.py, constructor keywords of ``WeaveLayerArgs`` :95-102
and ``WeaveGatherArgs`` :105-108).  ``create_weave_layers`` / ``WeaveModel`` read them through ``.args``, ``bag[i]``,
``*bag``, ``.use_batch_norm``, ``.dropout`` and ``.activation_function``; nothing else of the reference file is carried.

One immutable sequence type does all of it: the positional constructor arguments of the layer ARE the sequence (so
unpacking, indexing and ``len`` come from ``tuple``), the per-layer options ride as attributes."""


def _module_for(activation):
    """Activation name -> module instance (the table of nn.layers.get_activation_func); None / '' stay as they are."""
    if not activation:
        return activation
    from ..nn.layers import get_activation_func
    return get_activation_func(activation)


class LayerBag(tuple):
    """tuple of a layer's positional arguments + (activation_function, use_batch_norm, dropout)."""

    def __new__(cls, positional, activation_function=None, use_batch_norm=False, dropout=-1):
        bag = super(LayerBag, cls).__new__(cls, tuple(positional))
        bag.activation_function = activation_function
        bag.use_batch_norm = bool(use_batch_norm)
        bag.dropout = dropout
        return bag

    @property
    def args(self):
        return tuple(self)


def WeaveLayerArgs(n_atom_input_feat=75, n_pair_input_feat=14, n_atom_output_feat=50, n_pair_output_feat=50,
                   n_hidden_AA=50, n_hidden_PA=50, n_hidden_AP=50, n_hidden_PP=50, update_pair=True, activation=None,
                   batch_norm=False, dropout=-1):
    """Arguments of nn.layers.WeaveLayer in its positional order, then the activation module."""
    act = _module_for(activation)
    widths = (n_atom_input_feat, n_pair_input_feat, n_atom_output_feat, n_pair_output_feat, n_hidden_AA, n_hidden_PA,
              n_hidden_AP, n_hidden_PP)
    return LayerBag(widths + (update_pair, act), act, batch_norm, dropout)


def WeaveGatherArgs(conv_out_depth, n_depth=128, gaussian_expand=False, activation='tanh', epsilon=1e-7):
    """Arguments of nn.layers.WeaveGather / WeaveGather2D (no batch norm, no dropout on a gather layer)."""
    act = _module_for(activation)
    return LayerBag((conv_out_depth, n_depth, gaussian_expand, act, epsilon), act)
