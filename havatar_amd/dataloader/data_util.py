"""Ray generation and sampling maps (reference: dataloader/data_util.py)."""
import numpy as np
import torch


def make_ray_importance_sampling_map(mask, p=0.9):
    """data_util.py:5-10: probability p spread over the foreground, 1-p over the rest, normalised to 1."""
    probs = np.full(mask.shape, 1 - p, dtype=np.float32)
    probs[mask > 0] = p
    return probs * (1 / probs.sum())


def meshgrid_xy(tensor1, tensor2):
    """np.meshgrid(..., indexing='xy') for tensors (data_util.py:13-25)."""
    ii, jj = torch.meshgrid(tensor1, tensor2, indexing="ij")
    return ii.transpose(-1, -2), jj.transpose(-1, -2)


def get_rays(H, W, intr, c2w, normalize=True):
    """data_util.py:28-56.  intr = (fx, fy, cx/W, cy/H); c2w [3,4]; pixel (x=i, y=j) -> K^-1 [i, j, 1], rotated by c2w[:3,:3].
    Returns rays_o [H,W,3] (expanded view of the camera centre) and rays_d [H,W,3].
    The device-side equivalent is hav_gen_rays (include/havatar.h)."""
    K = np.eye(3, dtype=np.float32)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = intr[0], intr[1], intr[2] * W, intr[3] * H
    K_inv = torch.from_numpy(np.linalg.inv(K)).to(c2w.device)
    xs = torch.linspace(0, W - 1, W, device=c2w.device)
    ys = torch.linspace(0, H - 1, H, device=c2w.device)
    j, i = torch.meshgrid(ys, xs, indexing="ij")                      # [H,W]: j = row (y), i = column (x)
    pix = torch.stack((i, j, torch.ones_like(i)), -1)                 # [H,W,3]
    dirs = (K_inv[None] @ pix[..., None])[..., 0]
    rays_d = (c2w[None, :3, :3] @ dirs[..., None])[..., 0]
    if normalize:
        rays_d = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d
