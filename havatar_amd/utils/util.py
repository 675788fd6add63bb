"""Functional helpers of the hot path (subset of the reference's utils/util.py that the path uses: :179-254, :359-418)."""
import numpy as np
import torch
import torch.nn.functional as F


def get_box_warp_param(X_bounding, Y_bounding, Z_bounding):
    """scale = 2/(max-min), trans = -scale*(max+min)/2 per axis (reference utils/util.py:179-186)."""
    scales, trans = [], []
    for lo, hi in (X_bounding, Y_bounding, Z_bounding):
        f = 2 / (hi - lo)
        scales.append(float(f))
        trans.append(float(-(f * (lo + hi) * 0.5)))
    return tuple(scales), tuple(trans)


def create_UniformBoxWarp(XYZ_bounding):
    b = [np.asarray(v) for v in XYZ_bounding]
    scales, trans = get_box_warp_param(b[0], b[1], b[2])
    return UniformBoxWarp_new(scales=scales, trans=trans)


class _BoxWarpBase(torch.nn.Module):
    def __init__(self, scales, trans):
        super().__init__()
        self.register_buffer("scale_factor", torch.tensor(scales, dtype=torch.float32).reshape(1, 3))
        self.register_buffer("trans_factor", torch.tensor(trans, dtype=torch.float32).reshape(1, 3))

    def _st(self, x):
        s, t = self.scale_factor.to(x.device), self.trans_factor.to(x.device)
        return (s.unsqueeze(0), t.unsqueeze(0)) if x.ndim == 3 else (s, t)


class UniformBoxWarp(_BoxWarpBase):
    """x -> 2*(x*scale + trans) (reference :195-211)."""

    def inv_trans(self, pts):
        return (pts * 0.5 - self.trans_factor.to(pts.device)) / self.scale_factor.to(pts.device)

    def forward(self, coordinates):
        s, t = self._st(coordinates)
        return 2.0 * ((coordinates * s) + t)


class UniformBoxWarp_new(_BoxWarpBase):
    """x -> x*scale + trans (reference :214-236)."""

    def inv_trans(self, coordinates):
        if isinstance(coordinates, np.ndarray):
            return (coordinates - self.trans_factor.cpu().numpy()) * (1.0 / self.scale_factor.cpu().numpy())
        s, t = self._st(coordinates)
        return (coordinates - t) * (1.0 / s)

    def forward(self, coordinates):
        s, t = self._st(coordinates)
        return (coordinates * s) + t


def make_volume_pts(steps=50, perturb=False, gridwarper=None, z_scale=1.0):
    """steps^3 lattice in [-1,1]^3 (optionally jittered / mapped back through the box warp) (reference :239-254)."""
    xs = torch.linspace(-1.0, 1.0, steps=steps, dtype=torch.float32)
    zs = torch.linspace(-1.0, 1.0, steps=int(steps * z_scale), dtype=torch.float32)
    xv, yv, zv = torch.meshgrid(xs, xs.clone(), zs, indexing="ij")
    pts = torch.stack([xv, yv, zv], dim=-1).reshape(-1, 3)
    if perturb:
        pts = pts + torch.rand_like(pts) * (2 / (steps - 1))
    if gridwarper is not None:
        pts = gridwarper.inv_trans(pts)
    return pts


def sample_from_2dgrid(coordinates, feat_grid, padding_mode="zeros"):
    """feat_grid [B,C,H,W], coordinates [B,N,2] -> [B,N,C] (bilinear, align_corners=True) (reference :395-406)."""
    out = F.grid_sample(feat_grid, coordinates.unsqueeze(-2), mode="bilinear", padding_mode=padding_mode, align_corners=True)
    return out[..., 0].permute(0, 2, 1)


def sample_from_triplane_new(coordinates, feat_grid, padding_mode="zeros"):
    """coordinates [B,N,3] (or [N,3]), feat_grid [P,B,C,H,W] (or [P,C,H,W]) -> [B,N,C,P]: plane 0 at (x,y), plane 1 at (z,y),
    plane 2 at (x,z) (reference :359-392)."""
    c = coordinates.unsqueeze(0) if coordinates.ndim == 2 else coordinates
    B = c.shape[0]
    g = feat_grid.unsqueeze(1).expand(-1, B, -1, -1, -1) if feat_grid.ndim == 4 else feat_grid
    P = feat_grid.shape[0]
    if P < 1 or P > 3:
        raise NotImplementedError
    axes = ([0, 1], [2, 1], [0, 2])
    out = torch.stack([sample_from_2dgrid(c[..., axes[p]], g[p], padding_mode) for p in range(P)], dim=-1)
    return out[0] if coordinates.ndim == 2 else out


def voxel_feature(xyz, volume_feat, padding_mode="border"):
    """xyz [B,N,3] in [-1,1]^3 (x->W, y->H, z->D), volume_feat [B,C,D,H,W] -> [B,N,C] (reference :409-418)."""
    B, N, _ = xyz.shape
    feat = F.grid_sample(volume_feat, xyz.reshape(B, N, 1, 1, 3), mode="bilinear", padding_mode=padding_mode, align_corners=True)
    return feat[:, :, :, 0, 0].permute(0, 2, 1)
