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


class QwenTimestepProjEmbeddings:
    """embeddings.py:746-760: Timesteps(256, flip_sin_to_cos, shift 0, scale=1000) -> TimestepEmbedding."""

    def __init__(self, embedding_dim, data_type=BF16):
        self.time_proj = Timesteps(num_channels=256, flip_sin_to_cos=True, downscale_freq_shift=0, scale=1000)
        self.timestep_embedder = TimestepEmbedding(in_features=256, out_features=embedding_dim, data_type=data_type)

    def forward(self, timestep, hidden_states):
        timesteps_proj = self.time_proj.forward(timestep)
        return self.timestep_embedder.forward(timesteps_proj.to(dtype=hidden_states.dtype))


class QwenEmbedRope:
    """Rotary table of Qwen-Image (the reference keeps diffusers' implementation, embeddings.py:762-857), restated from its
    formula rather than from its code:

        token (f, y, x) of image `idx` and axis a with width d_a:  angle_{a,k} = p_a * theta^(-2k / d_a),  k < d_a / 2
        p_frame = idx + f;   p_height = y - ceil(H / 2), p_width = x - ceil(W / 2) when `scale_rope` centres the grid, else y, x
        text token n: the same three axis tables at position n + max(H, W) (or max(H // 2, W // 2) when centred)
        entry = exp(i * angle) as complex64 (torch.polar of an fp32 angle = fp32(position) * fp32(inverse frequency))

    `forward` returns (video_freqs [sum F*H*W, hd/2], text_freqs [max_txt, hd/2]); tables are built on the CPU with the
    reference's dtypes (bit-identical: tests/golden/tiny_qwen.npz) and cached per image shape."""

    def __init__(self, theta: int, axes_dim: List[int], scale_rope=False):
        assert all(d % 2 == 0 for d in axes_dim)
        self.theta, self.axes_dim, self.scale_rope = theta, axes_dim, scale_rope
        self._inv_freq = [1.0 / torch.pow(theta, torch.arange(0, d, 2).to(torch.float32).div(d)) for d in axes_dim]
        self.rope_cache = {}

    def _axis(self, axis: int, positions: torch.Tensor) -> torch.Tensor:
        """complex64 [len(positions), d_axis / 2]: unit vectors at angle position * inverse frequency"""
        angle = positions.to(torch.int64).to(torch.float32)[:, None] * self._inv_freq[axis][None, :]
        return torch.polar(torch.ones_like(angle), angle)

    def _grid_positions(self, n: int) -> torch.Tensor:
        return torch.arange(n) - (n - n // 2) if self.scale_rope else torch.arange(n)

    def _video_table(self, idx: int, frames: int, height: int, width: int) -> torch.Tensor:
        tok = torch.arange(frames * height * width)
        per_axis = (self._axis(0, idx + torch.arange(frames))[tok // (height * width)],
                    self._axis(1, self._grid_positions(height))[(tok // width) % height],
                    self._axis(2, self._grid_positions(width))[tok % width])
        return torch.cat(per_axis, dim=-1).contiguous()

    def forward(self, video_fhw, txt_seq_lens, device):
        shapes = video_fhw[0] if isinstance(video_fhw, list) and isinstance(video_fhw[0], list) else video_fhw
        if not isinstance(shapes, list):
            shapes = [shapes]
        tables, first_text_pos = [], 0
        for idx, (frames, height, width) in enumerate(shapes):
            key = (idx, frames, height, width)
            if key not in self.rope_cache:
                self.rope_cache[key] = self._video_table(idx, frames, height, width)
            tables.append(self.rope_cache[key])
            extent = max(height // 2, width // 2) if self.scale_rope else max(height, width)
            first_text_pos = max(first_text_pos, extent)
        text_pos = first_text_pos + torch.arange(max(txt_seq_lens))
        text = torch.cat([self._axis(a, text_pos) for a in range(len(self.axes_dim))], dim=1)
        return torch.cat(tables, dim=0).to(device), text.to(device)


class WanRotaryPosEmbed:
    """embeddings.py:859-922: per-axis (t, h, w) cos/sin tables in fp64 -> fp32, gathered on the latent grid.
    `forward` returns (cos, sin) of shape [1, tokens, 1, head_dim] like the reference; `merged` gives the
    [tokens, head_dim] = [cos[:, 0::2] | sin[:, 1::2]] table the rotary kernel consumes (transformer.py:489-491)."""

    def __init__(self, attention_head_dim: int, patch_size, max_seq_len: int, theta: float = 10000.0):
        self.attention_head_dim, self.patch_size, self.max_seq_len = attention_head_dim, patch_size, max_seq_len
        h_dim = w_dim = 2 * (attention_head_dim // 6)
        t_dim = attention_head_dim - h_dim - w_dim
        cos_l, sin_l = [], []
        for dim in (t_dim, h_dim, w_dim):
            c, s_ = get_1d_rotary_pos_embed(dim, torch.arange(max_seq_len), theta)
            cos_l.append(c)
            sin_l.append(s_)
        self.freqs_cos, self.freqs_sin = torch.cat(cos_l, dim=1), torch.cat(sin_l, dim=1)
        self._merged = {}

    def _grid(self, table, ppf, pph, ppw):
        hd = self.attention_head_dim
        parts = table.split([hd - 2 * (hd // 3), hd // 3, hd // 3], dim=1)
        a = parts[0][:ppf].view(ppf, 1, 1, -1).expand(ppf, pph, ppw, -1)
        b = parts[1][:pph].view(1, pph, 1, -1).expand(ppf, pph, ppw, -1)
        c = parts[2][:ppw].view(1, 1, ppw, -1).expand(ppf, pph, ppw, -1)
        return torch.cat([a, b, c], dim=-1).reshape(1, ppf * pph * ppw, 1, -1)

    def forward(self, hidden_states: torch.Tensor):
        _, _, num_frames, height, width = hidden_states.shape
        p_t, p_h, p_w = self.patch_size
        ppf, pph, ppw = num_frames // p_t, height // p_h, width // p_w
        return (self._grid(self.freqs_cos, ppf, pph, ppw).to(hidden_states.device),
                self._grid(self.freqs_sin, ppf, pph, ppw).to(hidden_states.device))

    def merged(self, num_frames: int, height: int, width: int, device, dtype) -> torch.Tensor:
        key = (num_frames, height, width, str(device), dtype)
        if key not in self._merged:
            p_t, p_h, p_w = self.patch_size
            ppf, pph, ppw = num_frames // p_t, height // p_h, width // p_w
            cos = self._grid(self.freqs_cos, ppf, pph, ppw).reshape(ppf * pph * ppw, -1)
            sin = self._grid(self.freqs_sin, ppf, pph, ppw).reshape(ppf * pph * ppw, -1)
            self._merged = {key: torch.cat((cos[:, 0::2], sin[:, 1::2]), dim=-1).to(dtype).contiguous().to(device)}
        return self._merged[key]


class WanImageEmbedding:
    """embeddings.py:925-947 (Wan2.1-I2V): FP32LayerNorm -> FeedForward(mult 1, erf GELU) -> FP32LayerNorm on the CLIP image
    tokens; never quantised (wan.py:237-238).  The first/last-frame positional table (`pos_embed_seq_len`) is not built --
    upstream cannot load it either (`self.init(...)`, wan.py:243)."""

    def __init__(self, in_features: int, out_features: int, pos_embed_seq_len=None, data_type=BF16):
        from fastdm_amd.layer.normalization import FP32LayerNorm
        from fastdm_amd.layer.transformer import FeedForward
        if pos_embed_seq_len is not None:
            raise NotImplementedError("WanImageEmbedding.pos_embed (FLF2V) is not built")
        self.norm1 = FP32LayerNorm(in_features)
        self.ff = FeedForward(in_features, out_features, mult=1, activation_fn="gelu", data_type=data_type)
        self.norm2 = FP32LayerNorm(out_features)
        self.pos_embed = None

    def forward(self, encoder_hidden_states_image: torch.Tensor) -> torch.Tensor:
        b, n, c = encoder_hidden_states_image.shape
        h = self.norm1.forward_bf16(encoder_hidden_states_image.reshape(b * n, c).contiguous())[0]
        h = self.ff.forward(h)
        return self.norm2.forward_bf16(h.contiguous())[0].view(b, n, -1)


class WanTimeTextImageEmbedding:
    """embeddings.py:949-992.  The time embedder is an fp32 layer; the image embedder exists for Wan2.1-I2V checkpoints."""

    def __init__(self, dim: int, time_freq_dim: int, time_proj_dim: int, text_embed_dim: int, image_embed_dim=None,
                 pos_embed_seq_len=None, data_type=BF16):
        self.timesteps_proj = Timesteps(num_channels=time_freq_dim, flip_sin_to_cos=True, downscale_freq_shift=0)
        self.time_embedder = TimestepEmbedding(in_features=time_freq_dim, out_features=dim, data_type=torch.float32)
        self.time_proj = QLinear(dim, time_proj_dim, data_type=data_type)
        self.text_embedder = PixArtAlphaTextProjection(text_embed_dim, dim, act_fn="gelu_tanh", data_type=data_type)
        self.image_embedder = (WanImageEmbedding(image_embed_dim, dim, pos_embed_seq_len=pos_embed_seq_len, data_type=data_type)
                               if image_embed_dim is not None else None)

    def forward(self, timestep, encoder_hidden_states, encoder_hidden_states_image=None, timestep_seq_len=None):
        """timestep [B], or [B * S] with `timestep_seq_len` = S (wan2.2 ti2v: one timestep per token, embeddings.py:979-980):
        temb and timestep_proj then carry a token axis ([B, S, D] / [B, S, 6 D])."""
        ts = self.timesteps_proj.forward(timestep).to(torch.float32)
        temb = self.time_embedder.forward(ts).type_as(encoder_hidden_states)
        timestep_proj = self.time_proj.forward(F.silu(temb))
        if timestep_seq_len is not None:
            temb = temb.unflatten(0, (-1, timestep_seq_len))
            timestep_proj = timestep_proj.unflatten(0, (-1, timestep_seq_len))
        img = self.image_embedder.forward(encoder_hidden_states_image) if encoder_hidden_states_image is not None else None
        return temb, timestep_proj, self.text_embedder.forward(encoder_hidden_states), img


class PatchEmbed:
    """embeddings.py:270-360 as SD3 uses it: 2x2 patches (a convolution with kernel == stride, i.e. a GEMM over
    non-overlapping patches) plus the centre crop of a learned [max, max] positional table (`pos_embed_max_size`)."""

    def __init__(self, height=224, width=224, patch_size=16, in_channels=3, embed_dim=768, layer_norm=False, flatten=True, bias=True,
                 interpolation_scale=1, pos_embed_type="sincos", pos_embed_max_size=None, data_type=BF16):
        assert not layer_norm and flatten and pos_embed_max_size, "only the SD3 configuration is built"
        self.patch_size, self.pos_embed_max_size = patch_size, pos_embed_max_size
        self.proj = QLinear(in_channels * patch_size * patch_size, embed_dim, bias=bias, data_type=data_type)
        self.pos_embed = None        # [1, max*max, embed_dim], loaded from the checkpoint

    def cropped_pos_embed(self, height, width):
        height, width = height // self.patch_size, width // self.patch_size
        m = self.pos_embed_max_size
        if height > m:
            raise ValueError(f"Height ({height}) cannot be greater than `pos_embed_max_size`: {m}.")
        if width > m:
            raise ValueError(f"Width ({width}) cannot be greater than `pos_embed_max_size`: {m}.")
        top, left = (m - height) // 2, (m - width) // 2
        pe = self.pos_embed.reshape(1, m, m, -1)[:, top: top + height, left: left + width, :]
        return pe.reshape(1, -1, pe.shape[-1])

    def forward(self, latent):
        B, C, H, W = latent.shape
        p = self.patch_size
        h, w = H // p, W // p
        patches = latent.reshape(B, C, h, p, w, p).permute(0, 2, 4, 1, 3, 5).reshape(B * h * w, C * p * p).contiguous()
        x = self.proj.forward_q(patches).view(B, h * w, -1)
        return (x + self.cropped_pos_embed(H, W)).to(x.dtype)
