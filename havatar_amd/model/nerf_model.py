"""Tri-plane NeRF: two StyleGAN-style encoders -> 2 planes; bilinear feature gather; 176->128->128->{1, 64->3} MLP
(reference model/nerf_model.py:10-117).  On HIP tensors at inference only `set_conditional_embedding` (the encoders) runs
here; gather + PE + MLP are inside the fused ray-march kernel.  `sample_pts_triplane_feat` / `forward` keep the PyTorch
statement for CPU tensors and autograd.  Only enc_mode='split' (the Trainer's choice) is implemented."""
import logging
import os

import torch
import torch.nn as nn

from .network.embedder import get_embedder
from .styleUnet import StyleGAN_zxc
from ..utils.sh_util import eval_sh
from ..utils.util import create_UniformBoxWarp, sample_from_triplane_new

_MLP_MODE_LOGGED = False


class _SplitKLinear(torch.autograd.Function):
    """y = x W^T + b for x [n, in] with n in the hundreds of thousands (every sample of every ray).  Forward and dL/dx are
    ordinary GEMMs; the weight gradient dY^T X contracts over n, a shape ([out x n].[n x in], out,in <= 176) for which rocBLAS
    runs a single skinny tile (0.5-1 ms per layer per pass on MI355X).  It is evaluated as a batch of 64 partial GEMMs + a sum."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx = dy @ weight if ctx.needs_input_grad[0] else None
        dw = db = None
        if ctx.needs_input_grad[1]:
            n = x.shape[0]
            parts = 64
            while parts > 1 and n % parts:
                parts //= 2
            dw = torch.bmm(dy.reshape(parts, n // parts, -1).transpose(1, 2), x.reshape(parts, n // parts, -1)).sum(0)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(0)
        return dx, dw, db


def _linear(layer, x):
    if x.is_cuda and x.ndim == 2 and x.shape[0] >= 16384 and torch.is_grad_enabled() and layer.weight.requires_grad:
        return _SplitKLinear.apply(x, layer.weight, layer.bias)
    return layer(x)


class ConditionalTriplaneNeRFModel_multiRender_split_view(nn.Module):
    def __init__(self, XYZ_bounding, num_encoding_fn_xyz=8, latent_code_dim=32, triPlane_feat_dim=32, rgb_feat_dim=32,
                 triplane_res=256, use_emb=True, enc_mode="split", sh_deg=2, cond_latent=True, cond_c_dim=0):
        super().__init__()
        if enc_mode != "split":
            raise NotImplementedError("only enc_mode='split' is on the hot path (model/nerf_trainer.py:19-26)")
        self.name = "ConditionalTriplaneNeRFModel_multiRender_split_view"
        self.pos_embedder, self.dim_xyz = get_embedder(multires=num_encoding_fn_xyz, input_dims=3, include_input=False)
        self.sh_deg = sh_deg
        self.use_sh = sh_deg >= 1
        include_xyz = self.dim_xyz if use_emb else 0
        self.dim_latent_code = 0 if cond_latent else latent_code_dim
        self.triPlane_feat_dim = triPlane_feat_dim
        self.rgb_feat_dim = rgb_feat_dim * (sh_deg + 1) ** 2
        self.use_emb, self.cond_latent, self.cond_c_dim = use_emb, cond_latent, cond_c_dim
        self.shared_backbone = self.two_head = False
        enc = dict(out_ch=triPlane_feat_dim, out_size=triplane_res, style_dim=latent_code_dim, middle_size=16, zero_latent=False,
                   zero_noise=True, no_skip=True, n_mlp=4, inp_size=256)
        self.XY_gen = StyleGAN_zxc(inp_ch=7, **enc)
        self.YZ_gen = StyleGAN_zxc(inp_ch=13, **enc)
        self.gridwarper = create_UniformBoxWarp(XYZ_bounding)
        self.layers_xyz = nn.ModuleList([nn.Linear(2 * triPlane_feat_dim + self.dim_latent_code + include_xyz, 128), nn.Linear(128, 128)])
        self.fc_alpha = nn.Linear(128, 1)
        self.fc_rgbFeat = nn.Linear(128, 64)
        self.fc_rgb = nn.Linear(64, self.rgb_feat_dim)
        self.relu = torch.nn.functional.relu
        self.triPlane_embeddings = None
        if not cond_latent:
            self.register_buffer("zero_latent", torch.zeros(latent_code_dim, dtype=torch.float32).reshape(1, -1))

    def set_conditional_embedding(self, **cond):
        """front/left/right_render_cond [B,7,256,256], latents [B,L], cond_c [B,12] -> self.triPlane_embeddings [2,B,C,H,W] (:58-86).
        `left` is mirrored along W and loses its mask channel; YZ_gen sees cat[left(6), right(7)]."""
        styles = None
        if "latents" in cond:
            lat = cond["latents"]
            c = cond["cond_c"].reshape(lat.shape[0], -1)
            if self.cond_latent:
                styles = [torch.cat([lat, c], -1)] if self.cond_c_dim > 0 else [lat]
            else:
                styles = [self.zero_latent.expand(lat.shape[0], -1)]
        front, right = cond["front_render_cond"], cond["right_render_cond"]
        left = cond["left_render_cond"].flip(dims=[3])
        if left.shape[1] > 3:
            left = left[:, :-1]
        yz_in = torch.cat([left, right], dim=1)
        if front.is_cuda and not torch.is_grad_enabled() and os.environ.get("HAVATAR_ENC_STREAMS", "1") != "0":
            # The two generators are independent until the stack, and at 16^2..64^2 most of their ~90 launches each leave the
            # GPU nearly idle: run them on two HIP streams (fork/join by events; captured as two branches of the frame's hipGraph).
            cur = torch.cuda.current_stream(front.device)
            side = self.__dict__.get("_side_stream")
            if side is None or side.device != front.device:
                side = self.__dict__["_side_stream"] = torch.cuda.Stream(device=front.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                yz, _ = self.YZ_gen(styles, yz_in)
            xy, _ = self.XY_gen(styles, front)
            cur.wait_stream(side)
            yz.record_stream(cur)
        else:
            xy, _ = self.XY_gen(styles, front)
            yz, _ = self.YZ_gen(styles, yz_in)
        self.triPlane_embeddings = torch.stack([xy, yz], dim=0)

    def sample_pts_triplane_feat(self, batch_pts, bidx=None):
        """batch_pts [B,N,3] -> [B*N, 2C], feature index = 2*channel + plane (:88-99)."""
        q = self.gridwarper(batch_pts)
        planes = self.triPlane_embeddings if bidx is None else self.triPlane_embeddings[:, bidx]
        if q.is_cuda and q.dtype == torch.float32 and q.ndim == 3 and planes.ndim == 5 and planes.shape[0] == 2 \
                and os.environ.get("HAVATAR_HIP_GATHER", "1") != "0":
            # training on HIP tensors: one coalesced gather / scatter kernel pair instead of ATen's grid_sampler on NCHW planes
            from ..native.gather import triplane_gather
            return triplane_gather(q, planes)
        f = sample_from_triplane_new(q, planes, padding_mode="zeros")
        return f.reshape(-1, f.shape[-1] * f.shape[-2])

    def forward(self, inp, pts_feat):
        """inp [n,3(+3)], pts_feat [n,2C] -> [n, rgb(3) | feat(64) | alpha(1)] (:101-117)."""
        xyz, dirs = inp[..., :3], inp[..., 3:]
        return self.mlp(torch.cat([pts_feat, self.pos_embedder(xyz)], -1), dirs)

    def mlp(self, x, dirs=None):
        """x [n, 2C+48] = cat(features, encoding) -> [n, rgb(3) | feat(64) | alpha(1)]: the layers of forward() (:106-117).

        HIP tensors in the training path (HAVATAR_TRAIN_MLP=bf16, the default; BASELINE config 5): one autograd node on the bf16
        matrix cores, activations recomputed in the backward (native/mlp_train.py).  HAVATAR_TRAIN_MLP=torch keeps the fp32
        nn.Linear statement (rocBLAS), which is what CPU tensors always take."""
        if (x.is_cuda and x.dtype == torch.float32 and x.ndim == 2 and x.shape[1] == 176 and self.sh_deg == 0 and x.shape[0] >= 1024
                and torch.is_grad_enabled() and os.environ.get("HAVATAR_TRAIN_MLP", "bf16") == "bf16"):
            from ..native import mlp_train
            ws = self.mlp_tensors()
            if [tuple(w.shape) for w in ws] == mlp_train._SHAPES:          # other widths (rgb_feat_dim != 3, ...) keep the ATen statement
                global _MLP_MODE_LOGGED
                if not _MLP_MODE_LOGGED:
                    _MLP_MODE_LOGGED = True
                    logging.getLogger("havatar_amd").info(
                        "radiance MLP under autograd: bf16-operand MFMA node (HAVATAR_TRAIN_MLP=bf16; parameter gradients in the 2e-2 "
                        "class of fp32 nn.Linear) -- HAVATAR_TRAIN_MLP=torch keeps the reference's fp32 layers")
                return mlp_train.fused_mlp(x, ws)
        for layer in self.layers_xyz:
            x = self.relu(_linear(layer, x))
        alpha = _linear(self.fc_alpha, x)
        x = _linear(self.fc_rgbFeat, x)
        sh = _linear(self.fc_rgb, x)
        rgb = sh if self.sh_deg == 0 else eval_sh(self.sh_deg, sh.reshape(sh.shape[0], -1, (self.sh_deg + 1) ** 2), dirs)
        return torch.cat((rgb, x, alpha), dim=-1)

    def mlp_tensors(self):
        """nn.Linear-layout tensors in the order RayMarcher.set_mlp expects."""
        l0, l1 = self.layers_xyz
        return (l0.weight, l0.bias, l1.weight, l1.bias, self.fc_alpha.weight, self.fc_alpha.bias, self.fc_rgbFeat.weight,
                self.fc_rgbFeat.bias, self.fc_rgb.weight, self.fc_rgb.bias)
