"""Hot-path subset of codes/models/archs/arch_util.py: initialize_weights (:8-25), make_layer
(:28-39), ResidualBlock_noBN (:121-139).  Parameter names / init identical to the reference."""
import torch.nn as nn
import torch.nn.init as init

from .. import functional as RF


def initialize_weights(net_l, scale=1):
    if not isinstance(net_l, list):
        net_l = [net_l]
    for net in net_l:
        for m in net.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                init.kaiming_normal_(m.weight, a=0, mode='fan_in')
                m.weight.data *= scale  # for residual block
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.BatchNorm2d):
                init.constant_(m.weight, 1)
                init.constant_(m.bias.data, 0.0)


def make_layer(basic_block, num_basic_block, **kwarg):
    return nn.Sequential(*[basic_block(**kwarg) for _ in range(num_basic_block)])


class ResidualBlock_noBN(nn.Module):
    """x + conv2(relu(conv1(x))): two fused kernels (ReLU in conv1's epilogue, the identity add
    in conv2's); in backward the identity gradient is added by conv1's data-gradient kernel."""

    def __init__(self, nf=64):
        super(ResidualBlock_noBN, self).__init__()
        self.conv1 = nn.Conv2d(nf, nf, 3, 1, 1, bias=True)
        self.conv2 = nn.Conv2d(nf, nf, 3, 1, 1, bias=True)
        initialize_weights([self.conv1, self.conv2], 0.1)

    def forward(self, x):
        return RF.res_block(x, self.conv1, self.conv2)
