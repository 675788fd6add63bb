"""VolumeDecoder: learned constant -> 6x(trilinear x2, Conv3d 3^3, InstanceNorm3d, ReLU) -> Conv3d -> sigmoid -> [1,2,R,R,R]
(reference model/network/voxel_encoder.py:150-210).  Producer of the skinning volume; stays PyTorch (MIOpen Conv3d).
state_dict keys: init_lc, filters.{i}.up.1.{weight,bias}, final_conv.{weight,bias}."""
import math

import torch
import torch.nn as nn


class Upsample2xTrilinear(nn.Module):
    """nn.Upsample(mode='trilinear', scale_factor=2, align_corners=False) as three 1-D passes of slices and lerps:
    out[2i] = 0.25 x[i-1] + 0.75 x[i],  out[2i+1] = 0.75 x[i] + 0.25 x[i+1]   (indices clamped at the borders).
    Same operator (the 8 trilinear weights are exactly the products of these), parameter-free like the module it replaces, but
    its autograd is plain elementwise kernels: ATen's upsample_trilinear3d_backward (atomics) took 50 ms of a 155 ms training
    step on MI355X, this takes ~1 ms."""

    @staticmethod
    def _axis(x, dim):
        n = x.shape[dim]
        prev = torch.cat([x.narrow(dim, 0, 1), x.narrow(dim, 0, n - 1)], dim)
        nxt = torch.cat([x.narrow(dim, 1, n - 1), x.narrow(dim, n - 1, 1)], dim)
        even = 0.25 * prev + 0.75 * x
        odd = 0.75 * x + 0.25 * nxt
        shape = list(x.shape)
        shape[dim] = 2 * n
        return torch.stack([even, odd], dim + 1).reshape(shape)

    def forward(self, x):
        if x.is_cuda and x.dtype == torch.float32 and x.dim() == 5:
            from ...native.train_ops import upsample3d_2x            # one HIP launch each way (hav_upsample3d_2x_*)
            return upsample3d_2x(x)
        for dim in (2, 3, 4):
            x = self._axis(x, dim)
        return x


class UpConv3DBlock(nn.Module):
    def __init__(self, input_nc, output_nc, up_mode="upsample"):
        super().__init__()
        assert up_mode in ("upconv", "upsample")
        if up_mode == "upconv":
            self.up = nn.ConvTranspose3d(input_nc, output_nc, kernel_size=4, stride=2, padding=1, bias=True)
        else:
            self.up = nn.Sequential(Upsample2xTrilinear(),          # index 0 has no parameters: state_dict keys stay up.1.{weight,bias}
                                    nn.Conv3d(input_nc, output_nc, kernel_size=3, padding=1, stride=1))
        self.norm = nn.InstanceNorm3d(output_nc, affine=False)

    def forward(self, x):
        if isinstance(self.up, nn.Sequential) and x.is_cuda:
            from ...native.train_ops import conv3d_small, conv3d_small_eligible
            y = self.up[0](x)
            if conv3d_small_eligible(y, self.up[1]):
                # volumes of <= 8^3 voxels: the convolution as GEMMs over an explicit patch matrix (native/train_ops.py::Conv3dSmall)
                return self.norm(conv3d_small(y, self.up[1]))
            return self.norm(self.up[1](y))
        return self.norm(self.up(x))


class VolumeDecoder(nn.Module):
    def __init__(self, num_in=1024, num_out=1, final_res=32, up_mode="upsample"):
        super().__init__()
        self.num_in, self.num_out = num_in, num_out
        self.register_buffer("init_lc", torch.rand(1, num_in, 1, 1, 1))
        n_layers, l0 = int(math.log2(final_res)), int(math.log2(num_in))
        self.filters = nn.ModuleList([UpConv3DBlock(2 ** (l0 - i), 2 ** (l0 - i - 1), up_mode) for i in range(n_layers)])
        self.final_conv = nn.Conv3d(2 ** (l0 - n_layers), num_out, bias=True, kernel_size=3, padding=1, stride=1)
        for m in self.modules():                      # BaseNetwork.init_weights() defaults: xavier_normal_, gain 0.02 (:11-38)
            if isinstance(m, (nn.Conv3d, nn.ConvTranspose3d)):
                nn.init.xavier_normal_(m.weight, gain=0.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0.0)

    def forward(self):
        x = self.init_lc
        for f in self.filters:
            x = torch.relu(f(x))
        x = torch.sigmoid(self.final_conv(x))
        return torch.cat([x, 1 - x], dim=1)
