"""Small conv building block.  `SingleConv` keeps the child names of the
reference's models/p2rnet/modules/sub_modules.py:88-113 (`conv`, `batchnorm`,
`groupnorm`, `ReLU`, `LeakyReLU`, `ELU`) so `state_dict` keys are identical."""
from torch import nn

_CONV = {1: nn.Conv1d, 2: nn.Conv2d, 3: nn.Conv3d}
_BN = {1: nn.BatchNorm1d, 2: nn.BatchNorm2d, 3: nn.BatchNorm3d}


def create_conv(in_channels, out_channels, kernel_size, order, num_groups, padding, ndim,
                negative_slope=1e-2):
    """List of (name, module) in the order spelled by `order`:
    c conv, b batchnorm, g groupnorm, r ReLU, l LeakyReLU, e ELU.  The conv has a
    bias only when no norm layer is present (sub_modules.py:61-63)."""
    assert 'c' in order, "Conv layer MUST be present"
    assert order[0] not in 'rle', 'Non-linearity cannot be the first operation in the layer'
    if ndim not in _CONV:
        raise NotImplementedError('Unknown ndim.')
    ci = order.index('c')
    out = []
    for i, ch in enumerate(order):
        width = in_channels if i < ci else out_channels
        if ch == 'c':
            out.append(('conv', _CONV[ndim](in_channels, out_channels, kernel_size, padding=padding,
                                            bias=not ('g' in order or 'b' in order))))
        elif ch == 'b':
            out.append(('batchnorm', _BN[ndim](width)))
        elif ch == 'g':
            groups = 1 if width < num_groups else num_groups
            assert width % groups == 0
            out.append(('groupnorm', nn.GroupNorm(num_groups=groups, num_channels=width)))
        elif ch == 'r':
            out.append(('ReLU', nn.ReLU(inplace=True)))
        elif ch == 'l':
            out.append(('LeakyReLU', nn.LeakyReLU(inplace=True, negative_slope=negative_slope)))
        elif ch == 'e':
            out.append(('ELU', nn.ELU(inplace=True)))
        else:
            raise ValueError(f"Unsupported layer type '{ch}'. MUST be one of ['b', 'g', 'r', 'l', 'e', 'c']")
    return out


class SingleConv(nn.Sequential):
    def __init__(self, in_channels, out_channels, kernel_size=3, order='gcr', num_groups=8, padding=1,
                 ndim=3, negative_slope=1e-2):
        super().__init__()
        for name, module in create_conv(in_channels, out_channels, kernel_size, order, num_groups,
                                        padding=padding, ndim=ndim, negative_slope=negative_slope):
            self.add_module(name, module)
