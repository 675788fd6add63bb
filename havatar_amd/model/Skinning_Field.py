"""Two-bone linear-blend-skinning field sampled from a learned 64^3 weight volume
(reference model/Skinning_Field.py:43-132).  At inference on HIP tensors the lookup runs inside the fused ray-march kernel
(Trainer.predict_and_render_radiance); this module owns the volume (VolumeDecoder, frozen by fix_canonical_W) and keeps the
PyTorch forward for CPU tensors, autograd and the training utilities."""
import copy

import torch
import torch.nn as nn

from .network.voxel_encoder import VolumeDecoder
from ..utils.util import UniformBoxWarp, make_volume_pts, voxel_feature


class Deformation_Field_new(nn.Module):
    def __init__(self, gridwarper=None, options=None, need_nr=False, eval=False):
        super().__init__()
        if need_nr:
            raise NotImplementedError("the non-rigid branch is a broken stub in the reference (exit(0), :18-40) and never enabled")
        o = options or {}
        self.canonical_Wvolume = VolumeDecoder(num_in=o.get("init_length", 1024), num_out=1, final_res=o.get("vol_res", 64),
                                               up_mode=o.get("up_mode", "upsample"))
        self.gridwarper = (UniformBoxWarp(scales=(1 / 2.5, 1 / 2.5, 1 / 2.0), trans=(0.0, -0.0, -0.2)) if gridwarper is None
                           else copy.deepcopy(gridwarper))
        self.register_buffer("identity_trans", torch.eye(4, dtype=torch.float32)[:, :-1])
        self.nr_motion_field = None
        self.fix_canoW = False

    def fix_canonical_W(self):
        """Freeze the volume for inference: bone-1 weight forced to 1 on the y=0 slab and on z=0,y<R/8; bone 0 = 1 - bone 1 (:57-62)."""
        self.fix_canoW = True
        w = self.canonical_Wvolume().detach()
        w[:, 1:, :, 0, :] = 1.0
        w[:, 1:, :1, :w.shape[-1] // 8, :] = 1.0
        self.canonical_W = torch.cat([1 - w[:, 1:], w[:, 1:]], dim=1)

    def volume_once(self):
        """canonical_Wvolume() evaluated ONCE per set of decoder weights: the reference runs the same VolumeDecoder (no inputs, no
        RNG) for the coarse pass, for the fine pass and again for the smoothness term of the loss (Skinning_Field.py:79,
        train_avatar.py:124) -- three identical 64^3 evaluations and backward passes per step.  One node in the autograd graph
        gives the same value and the same (summed) gradient.  Keyed on the parameters' version counters, the hipGraph weights epoch
        and the grad mode, so an optimiser step, a load_state_dict or a no_grad validation render all re-evaluate."""
        from ..graph import weights_epoch
        key = (tuple(p._version for p in self.canonical_Wvolume.parameters()), tuple(p.data_ptr() for p in self.canonical_Wvolume.parameters()),
               weights_epoch(), torch.is_grad_enabled())
        c = self.__dict__.get("_vol_once")
        if c is None or c[0] != key:
            vol = self.canonical_Wvolume()
            c = (key, vol)
            self.__dict__["_vol_once"] = c
            if vol.requires_grad:         # a backward pass consumes the node (its buffers are freed): never hand it out again
                def _consumed(grad, d=self.__dict__, k=key):
                    if d.get("_vol_once", (None,))[0] == k:
                        d.pop("_vol_once", None)
                    return None
                vol.register_hook(_consumed)
        return c[1]

    def current_volume(self):
        """[1,2,R,R,R] volume the forward pass samples (frozen copy if fix_canonical_W() was called)."""
        return self.canonical_W if self.fix_canoW else self.volume_once()

    def sample_volume(self, pts, padding_mode="border"):
        vol = self.canonical_Wvolume()
        return voxel_feature(xyz=self.gridwarper(pts.unsqueeze(0)), volume_feat=vol[:, 0:1], padding_mode=padding_mode)[0]

    def forward(self, pts, pts_view, inv_Trans):
        """pts, pts_view [B,N,3]; inv_Trans [B,4,3] = [M ; tau].  p_i = (p + tau_i) M_i for T in {identity, inv_Trans};
        w_i = trilinear(W[i], boxwarp(p_i)) (border); p' = sum_i w_i p_i / (sum_i w_i + 1e-8); same blend for the view dirs (:70-98)."""
        B = inv_Trans.shape[0]
        Ts = [self.identity_trans.unsqueeze(0).expand(B, -1, -1), inv_Trans]
        w_c = self.current_volume().expand(B, -1, -1, -1, -1)
        p_i = [torch.matmul(pts + T[:, -1:], T[:, :3, :3]) for T in Ts]
        w = torch.cat([voxel_feature(xyz=self.gridwarper(p), volume_feat=w_c[:, i:i + 1]) for i, p in enumerate(p_i)], -1)
        w = w / (w.sum(dim=-1, keepdim=True) + 1e-8)
        out_pts = sum(w[:, :, i:i + 1] * p_i[i] for i in range(2))
        out_view = sum(w[:, :, i:i + 1] * torch.matmul(pts_view, Ts[i][:, :3, :3]) for i in range(2)) if pts_view is not None else 0
        return out_pts, out_view

    def pretrain_wc(self, num_iter=1, lr=1e-3, save_path=None, pose_space=False, vol_thr=None):
        """Warm-up: fit the volume to a box indicator with BCE on 20^3 jittered points (:101-125)."""
        if vol_thr is None:
            vol_thr = [[-0.5, 0.5], [-0.8, 0.5], [-0.3, 1.0]]
        opt = torch.optim.Adam([{"params": self.parameters()}], lr=lr)
        dev = self.identity_trans.device
        for _ in range(num_iter):
            pts = make_volume_pts(steps=20, perturb=True, gridwarper=self.gridwarper).to(dev)
            inside = torch.ones(pts.shape[0], dtype=torch.bool, device=dev)
            for a in range(3):
                inside &= (pts[:, a] > vol_thr[a][0]) & (pts[:, a] < vol_thr[a][1])
            vol = self.canonical_Wvolume()
            w = voxel_feature(xyz=self.gridwarper(pts.unsqueeze(0)), volume_feat=vol[:, 0:1] if pose_space else vol[:, 1:])
            gt = inside.to(w.dtype).unsqueeze(-1)          # (the reference's torch.zeros(...): the default dtype)
            loss = torch.nn.functional.binary_cross_entropy(torch.clamp(w, 0.0, 1.0)[0], gt)
            loss.backward()
            opt.step()
            opt.zero_grad()
        if save_path is not None:
            torch.save(self.canonical_Wvolume(), save_path)
        return float(loss.detach())

    def visualize_motion_weight_vol(self, path):
        """Dump the 20^3 lattice with bone-1 weights as vertex colours to a Wavefront .obj (:127-132)."""
        dev = self.identity_trans.device
        pts = make_volume_pts(steps=20, perturb=False, gridwarper=self.gridwarper).to(dev)
        w = voxel_feature(xyz=self.gridwarper(pts.unsqueeze(0)), volume_feat=self.canonical_Wvolume()[:, 1:])[0, :, 0]
        with open(path, "w") as f:
            for p, c in zip(pts.detach().cpu().tolist(), w.detach().cpu().tolist()):
                f.write("v %f %f %f %f %f %f\n" % (p[0], p[1], p[2], c, c, c))
