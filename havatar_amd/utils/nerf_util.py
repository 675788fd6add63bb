"""PyTorch statement of the compositing / resampling functions (reference utils/nerf_util.py:4-117).

Used for CPU tensors and for autograd (training) -- HIP tensors at inference go through the fused kernel instead."""
import torch


def cumprod_exclusive(tensor):
    """exclusive cumulative product along the last dim (reference :4-25)."""
    cp = torch.cumprod(tensor, -1)
    return torch.cat([torch.ones_like(cp[..., :1]), cp[..., :-1]], dim=-1)


def volume_render_radiance_field(radiance_field, depth_values, ray_directions, radiance_field_noise_std=0.0, act_feat=False,
                                 background_prior=None):
    """radiance_field [...,S,C+1] (last channel = density). Returns rgb_map, disp_map, acc_map, weights, depth_map
    (reference :28-73; like the reference this sigmoids radiance_field[..., :3] IN PLACE when act_feat is False)."""
    dists = depth_values[..., 1:] - depth_values[..., :-1]
    dists = torch.cat([dists, dists[..., -1:]], dim=-1) * ray_directions[..., None, :].norm(p=2, dim=-1)
    if act_feat is not None:
        if act_feat:
            radiance_field[..., :-1] = torch.sigmoid(radiance_field[..., :-1])
        else:
            radiance_field[..., :3] = torch.sigmoid(radiance_field[..., :3])
    noise = 0.0
    if radiance_field_noise_std > 0.0:
        noise = torch.randn(radiance_field[..., -1].shape, dtype=radiance_field.dtype, device=radiance_field.device) * radiance_field_noise_std
    sigma_a = torch.relu(radiance_field[..., -1] + noise)
    alpha = 1.0 - torch.exp(-sigma_a * dists)
    weights = alpha * cumprod_exclusive(1.0 - alpha + 1e-10)
    rgb_map = (weights[..., None] * radiance_field[..., :-1]).sum(dim=-2)
    depth_map = (weights * depth_values).sum(dim=-1)
    acc_map = weights.sum(dim=-1)
    disp_map = 1.0 / torch.max(1e-10 * torch.ones_like(depth_map), depth_map / acc_map)
    if background_prior is not None:
        rgb_map[..., :3] = rgb_map[..., :3] + (1.0 - acc_map[..., None]) * background_prior
    return rgb_map, disp_map, acc_map, weights, depth_map


def sample_pdf(bins, weights, num_samples, det=False):
    """inverse-CDF sampling (reference :76-117; stratified u drawn on the CPU exactly like the reference :93-96)."""
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, dim=-1, keepdim=True)
    cdf = torch.cumsum(pdf, dim=-1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], dim=-1)
    if det:
        u = torch.linspace(0.0, 1.0, steps=num_samples, dtype=weights.dtype, device=weights.device)
        u = u.expand(list(cdf.shape[:-1]) + [num_samples])
    else:
        s = 1 / num_samples
        u = (torch.arange(num_samples) * s).unsqueeze(0)
        if weights.is_cuda and torch.cuda.is_current_stream_capturing():
            # a captured training step (graph.GraphedTrainStep): a host draw would be frozen into the graph, so this one draw
            # comes from the device generator (graph-safe Philox offsets) instead of the CPU generator the reference uses
            u = (torch.arange(num_samples, device=weights.device) * s).unsqueeze(0) + torch.rand(list(cdf.shape[:-1]) + [num_samples], dtype=weights.dtype, device=weights.device) * (s - 1e-6)
        else:
            u = u + torch.rand(list(cdf.shape[:-1]) + [num_samples], dtype=weights.dtype) * (s - 1e-6)
            u = u.to(weights.device)
    u, cdf = u.contiguous(), cdf.contiguous()
    inds = torch.searchsorted(cdf.detach(), u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    inds_g = torch.stack((below, above), dim=-1)
    shape = (inds_g.shape[0], inds_g.shape[1], cdf.shape[-1])
    cdf_g = torch.gather(cdf.unsqueeze(1).expand(shape), 2, inds_g)
    bins_g = torch.gather(bins.unsqueeze(1).expand(shape), 2, inds_g)
    denom = cdf_g[..., 1] - cdf_g[..., 0]
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_g[..., 0]) / denom
    return bins_g[..., 0] + t * (bins_g[..., 1] - bins_g[..., 0])
