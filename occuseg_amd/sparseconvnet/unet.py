"""scn.UNet -- the network the headline metric is defined on.

Same module tree (hence the same state_dict keys) as sparseconvnet/networkArchitectures.py:202-306:
per level  block* ; ConcatTable(Identity, BN -> Conv 2/2 -> U(next) -> BN -> Deconv 2/2) ; JoinTable ; block*,
residual block = ConcatTable(Identity | NiN, BN -> SC3 -> BN -> SC3) -> AddTable.
"""
from .containers import (AddTable, Identity, JoinConcatTable, JoinTable, NetworkInNetwork, ResidualConcatTable,
                         Sequential)
from .conv import Convolution, Deconvolution, SubmanifoldConvolution
from .norm import BatchNormLeakyReLU


def UNet(dimension, reps, nPlanes, residual_blocks=False, downsample=[2, 2], leakiness=0):
    def bn(n):
        return BatchNormLeakyReLU(n, leakiness=leakiness)

    def sc(a, b):
        return SubmanifoldConvolution(dimension, a, b, 3, False)

    def add_block(m, a, b):
        if residual_blocks:  # ResNet style
            shortcut = Identity() if a == b else NetworkInNetwork(a, b, False)
            body = Sequential().add(bn(a)).add(sc(a, b)).add(bn(b)).add(sc(b, b))
            # ResidualConcatTable: a ConcatTable whose last convolution adds the shortcut inside its kernel
            m.add(ResidualConcatTable().add(shortcut).add(body)).add(AddTable())
        else:  # VGG style
            m.add(Sequential().add(bn(a)).add(sc(a, b)))

    def level(planes):
        m = Sequential()
        for _ in range(reps):
            add_block(m, planes[0], planes[0])
        if len(planes) > 1:
            down_up = (Sequential()
                       .add(bn(planes[0]))
                       .add(Convolution(dimension, planes[0], planes[1], downsample[0], downsample[1], False))
                       .add(level(planes[1:]))
                       .add(bn(planes[1]))
                       .add(Deconvolution(dimension, planes[1], planes[0], downsample[0], downsample[1], False)))
            # (modules are created in the reference's order: same seed, same parameters)  The deconvolution's result is the
            # right half of the JoinTable below: it is written there directly
            list(down_up._modules.values())[-1].join_right = True
            # JoinConcatTable: a ConcatTable that lets the first batch norm of down_up add the left half of the joined gradient
            m.add(JoinConcatTable().add(Identity()).add(down_up))
            m.add(JoinTable())
            for i in range(reps):
                add_block(m, planes[0] * (2 if i == 0 else 1), planes[0])
        return m

    return level(list(nPlanes))
