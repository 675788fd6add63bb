"""Sinusoidal positional encoding (reference model/network/embedder.py:4-61, 99-115).
Output order per frequency f = 2^k: [sin(x f) for x,y,z][sin(x f + pi/2) for x,y,z]; the cosine is a shifted sine."""
import math

import torch
import torch.nn as nn


class Embedder:
    def __init__(self, input_dims, num_freqs, max_freq_log2, include_input=True, log_sampling=True, device="cpu", **_):
        self.include_input = include_input
        self.N_freqs = num_freqs
        self.max_freq = max_freq_log2
        if log_sampling:
            fb = 2.0 ** torch.linspace(0.0, max_freq_log2, steps=num_freqs)
        else:
            fb = torch.linspace(2.0 ** 0.0, 2.0 ** max_freq_log2, steps=num_freqs)
        self.freq_bands = fb.reshape(num_freqs, 1).to(device)
        self.out_dim = (input_dims if include_input else 0) + input_dims * 2 * num_freqs

    def embed(self, x, alpha=None):
        if self.N_freqs == 0:
            return x
        if self.freq_bands.device != x.device:                                # once per device (a copy per call would also break
            self.freq_bands = self.freq_bands.to(x.device)                    # hipGraph capture of the training step)
        fb = self.freq_bands
        ang = x.unsqueeze(-2) * fb                                            # [..., F, C]
        feat = torch.sin(torch.stack((ang, ang + math.pi / 2), dim=-2))       # [..., F, 2, C]
        if alpha is not None:                                                 # coarse-to-fine window (unused on the path)
            a = torch.clip(alpha - fb, 0.0, 1.0)
            feat = (0.5 * (1 + torch.cos(math.pi * a + math.pi))).reshape(-1, 1, 1) * feat
        feat = feat.reshape(list(x.shape[:-1]) + [-1])
        return torch.cat([x, feat], dim=-1) if self.include_input else feat


def get_embedder(multires, i=0, input_dims=3, include_input=True, device="cpu"):
    """-> (embed_fn, out_dim).  The reference hard-codes device='cuda' (:99); the frequency table here follows the input."""
    if i == -1:
        return nn.Identity(), input_dims
    e = Embedder(input_dims=input_dims, num_freqs=multires, max_freq_log2=multires - 1, include_input=include_input,
                 log_sampling=True, device=device)
    return (lambda x, alpha=None, eo=e: eo.embed(x, alpha)), e.out_dim
