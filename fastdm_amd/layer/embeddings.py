"""Time / guidance / text conditioning and the RoPE table for FLUX (mirror of the parts of
fastdm/layer/embeddings.py the FLUX core uses: :18-69, :160-223, :377-410, :118-147, :527-590).

These are M = batch sized (a few KFLOP) or pure table construction: the tables are built with torch
ops exactly as the reference does (fp64 frequencies for RoPE) and cached per (ids) on the device.
"""
import math
from typing import List

import torch
import torch.nn.functional as F

from fastdm_amd.layer.qlinear import QLinear

BF16 = torch.bfloat16


def get_timestep_embedding(timesteps: torch.Tensor, embedding_dim: int, flip_sin_to_cos: bool = False,
                           downscale_freq_shift: float = 1, scale: float = 1, max_period: int = 10000):
    """embeddings.py:18-69."""
    assert len(timesteps.shape) == 1, "Timesteps should be a 1d-array"
    half_dim = embedding_dim // 2
    exponent = -math.log(max_period) * torch.arange(start=0, end=half_dim, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half_dim - downscale_freq_shift)
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    emb = scale * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half_dim:], emb[:, :half_dim]], dim=-1)
    if embedding_dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


class Timesteps:
    """embeddings.py:377-393."""

    def __init__(self, num_channels: int, flip_sin_to_cos: bool, downscale_freq_shift: float, scale: int = 1):
        self.num_channels, self.flip_sin_to_cos, self.downscale_freq_shift, self.scale = num_channels, flip_sin_to_cos, downscale_freq_shift, scale

    def forward(self, timesteps):
        return get_timestep_embedding(timesteps, self.num_channels, flip_sin_to_cos=self.flip_sin_to_cos,
                                      downscale_freq_shift=self.downscale_freq_shift, scale=self.scale)


class TimestepEmbedding:
    """embeddings.py:400-410: Linear -> SiLU -> Linear."""

    def __init__(self, in_features, out_features, data_type=BF16):
        self.linear1 = QLinear(in_features, out_features, data_type=data_type)
        self.linear2 = QLinear(out_features, out_features, data_type=data_type)

    def forward(self, sample):
        return self.linear2.forward(F.silu(self.linear1.forward(sample)))


class PixArtAlphaTextProjection:
    """embeddings.py:118-147 (act_fn 'silu' on the FLUX path)."""

    def __init__(self, in_features, hidden_size, out_features=None, act_fn="gelu_tanh", data_type=BF16):
        self.linear1 = QLinear(in_features, hidden_size, data_type=data_type)
        self.linear2 = QLinear(hidden_size, out_features or hidden_size, data_type=data_type)
        self.act_fn = act_fn

    def forward(self, caption):
        h = self.linear1.forward(caption)
        if self.act_fn == "gelu_tanh":
            h = F.gelu(h, approximate="tanh")
        elif self.act_fn == "silu":
            h = F.silu(h)
        return self.linear2.forward(h)


class CombinedTimestepGuidanceTextProjEmbeddings:
    """embeddings.py:569-590."""

    def __init__(self, embedding_dim, pooled_projection_dim, data_type=BF16):
        self.time_proj = Timesteps(num_channels=256, flip_sin_to_cos=True, downscale_freq_shift=0)
        self.timestep_embedder = TimestepEmbedding(256, embedding_dim, data_type=data_type)
        self.guidance_embedder = TimestepEmbedding(256, embedding_dim, data_type=data_type)
        self.text_embedder = PixArtAlphaTextProjection(pooled_projection_dim, embedding_dim, act_fn="silu", data_type=data_type)

    def forward(self, timestep, guidance, pooled_projection):
        timesteps_emb = self.timestep_embedder.forward(self.time_proj.forward(timestep).to(dtype=pooled_projection.dtype))
        guidance_emb = self.guidance_embedder.forward(self.time_proj.forward(guidance).to(dtype=pooled_projection.dtype))
        time_guidance_emb = timesteps_emb + guidance_emb
        return time_guidance_emb + self.text_embedder.forward(pooled_projection)


class CombinedTimestepTextProjEmbeddings:
    """embeddings.py:551-567 (guidance_embeds=False checkpoints, e.g. FLUX.1-schnell)."""

    def __init__(self, embedding_dim, pooled_projection_dim, data_type=BF16):
        self.time_proj = Timesteps(num_channels=256, flip_sin_to_cos=True, downscale_freq_shift=0)
        self.timestep_embedder = TimestepEmbedding(256, embedding_dim, data_type=data_type)
        self.text_embedder = PixArtAlphaTextProjection(pooled_projection_dim, embedding_dim, act_fn="silu", data_type=data_type)

    def forward(self, timestep, pooled_projection):
        timesteps_emb = self.timestep_embedder.forward(self.time_proj.forward(timestep).to(dtype=pooled_projection.dtype))
        return timesteps_emb + self.text_embedder.forward(pooled_projection)


def get_1d_rotary_pos_embed(dim: int, pos: torch.Tensor, theta: float = 10000.0, freqs_dtype=torch.float64):
    """embeddings.py:160-209 with use_real=True, repeat_interleave_real=True (the FLUX branch)."""
    assert dim % 2 == 0
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=freqs_dtype, device=pos.device)[: (dim // 2)] / dim))
    freqs = torch.outer(pos, freqs)
    return freqs.cos().repeat_interleave(2, dim=1).float(), freqs.sin().repeat_interleave(2, dim=1).float()


class FluxPosEmbed:
    """embeddings.py:527-549.  Built on the CPU in fp64 like the oracle (ids are tiny), then moved."""

    def __init__(self, theta: int, axes_dim: List[int]):
        self.theta, self.axes_dim = theta, axes_dim

    def forward(self, ids: torch.Tensor):
        dev = ids.device
        pos = ids.detach().float().cpu()
        cos_out, sin_out = [], []
        for i in range(pos.shape[-1]):
            cos, sin = get_1d_rotary_pos_embed(self.axes_dim[i], pos[:, i], theta=self.theta)
            cos_out.append(cos)
            sin_out.append(sin)
        return torch.cat(cos_out, dim=-1).to(dev), torch.cat(sin_out, dim=-1).to(dev)
