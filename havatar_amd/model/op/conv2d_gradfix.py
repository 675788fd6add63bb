"""conv2d_gradfix API (reference model/op/conv2d_gradfix.py:12-92).

The reference's custom autograd path is only taken on torch 1.7-1.9 with cuDNN (:78-92); on every newer torch it
warns and calls torch.nn.functional directly.  On PyTorch-ROCm 2.x the behaviour is therefore exactly the
functional convs (MIOpen), which natively support the double backward R1/path-length regularisers need.  This
module keeps the public names (`conv2d`, `conv_transpose2d`, `no_weight_gradients`, `enabled`,
`weight_gradients_disabled`); `no_weight_gradients()` detaches the weight so no weight gradient is formed.
"""
import contextlib

from torch.nn import functional as F

enabled = True
weight_gradients_disabled = False


@contextlib.contextmanager
def no_weight_gradients():
    global weight_gradients_disabled
    old = weight_gradients_disabled
    weight_gradients_disabled = True
    try:
        yield
    finally:
        weight_gradients_disabled = old


def _w(weight):
    return weight.detach() if weight_gradients_disabled else weight


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    return F.conv2d(input=input, weight=_w(weight), bias=bias, stride=stride, padding=padding, dilation=dilation,
                    groups=groups)


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
    return F.conv_transpose2d(input=input, weight=_w(weight), bias=bias, stride=stride, padding=padding,
                              output_padding=output_padding, groups=groups, dilation=dilation)


def could_use_op(input):
    return False
