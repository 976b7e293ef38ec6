"""Radial block-sparse attention (mirror of fastdm/sparse/xsparse.py:7-250).

`SparseAttn.from_json/from_dict` -> `RadialAttn`; `post_init(video_token_num, num_frame)`; `apply(query, key, value)`
with [tokens, heads, head_dim] tensors.  The mask is the reference's "log mask": for every pair of frames (i, j) a band
|col - row| <= w(|i-j|) (frame 0 is an attention sink, far frame pairs are thinned out by a diagonal split), shrunk to
block granularity by column density, then converted to the attention kernel's 128 x 64 (query x key) blocks.  The
generation runs with torch ops on the query's device once per geometry and is cached; the attention itself is
`sparse_scaled_dot_product_attention` of the HIP library (bf16; the reference's CUDA kernels quantise inside).
"""
from typing import Any, Dict, Type

import torch

from fastdm_amd.kernel.operators_set import scaled_dot_product_attention, sparse_scaled_dot_product_attention
from fastdm_amd.sparse.config import SparseConfig


class SparseAttn:
    _registry: Dict[str, Type["SparseAttn"]] = {}

    def __init__(self, config: SparseConfig):
        self.config = config

    @classmethod
    def register(cls, name: str):
        def decorator(sub_cls):
            cls._registry[name.lower()] = sub_cls
            return sub_cls
        return decorator

    @classmethod
    def _make(cls, config: SparseConfig) -> "SparseAttn":
        sparse_cls = cls._registry.get(config.sparse_algorithm.lower())
        if sparse_cls is None:
            raise ValueError(f"Unknown sparse algorithm: {config.sparse_algorithm.lower()}")
        return sparse_cls(config)

    @classmethod
    def from_dict(cls, data: Dict[str, Any]) -> "SparseAttn":
        return cls._make(SparseConfig.from_dict(data))

    @classmethod
    def from_json(cls, file_path: str) -> "SparseAttn":
        return cls._make(SparseConfig.from_json(file_path))

    def apply(self, query, key, value, pre_defined_mask=None):
        raise NotImplementedError("Subclasses must implement apply method")


@SparseAttn.register("radial")
class RadialAttn(SparseAttn):
    def __init__(self, config):
        super().__init__(config)
        self.video_token_num = self.num_frame = None
        self._masks = {}      # (tokens, video_token_num, num_frame, device) -> int8 [tokens/128, tokens/64]

    def post_init(self, video_token_num=None, num_frame=None):
        self.video_token_num, self.num_frame = video_token_num, num_frame

    # ---- mask generation (xsparse.py:71-183)
    def get_window_width(self, i, j, token_per_frame):
        dist = abs(i - j)
        if self.config.model_type == "wan":
            if dist < 1:
                return token_per_frame
            if dist == 1:
                return token_per_frame // 2
        elif self.config.model_type == "hunyuan":
            if dist <= 1:
                return token_per_frame
        else:
            raise ValueError(f"Unknown model type: {self.config.model_type}")
        decay_length = 2 ** token_per_frame.bit_length() / 2 ** dist.bit_length() * self.config.decay_factor
        return decay_length if decay_length >= self.config.block_size else self.config.block_size

    def _pair_is_kept(self, i, j, token_per_frame) -> bool:
        """get_diagonal_split_mask: frame pairs whose decay length fell below a block are kept only every
        `split_factor`-th distance."""
        dist = abs(i - j)
        decay_length = 2 ** token_per_frame.bit_length() / 2 ** dist.bit_length()
        if decay_length >= self.config.block_size:
            return True
        return dist % int(self.config.block_size / decay_length) == 0

    def shrinkMaskStrict(self, mask):
        bs = self.config.block_size
        nb = mask.shape[0] // bs
        m = mask[: nb * bs, : nb * bs].view(nb, bs, nb, bs)
        col_density = m.sum(dim=1) / bs
        frac_dense = (col_density > 1 / 3).sum(dim=-1) / ((col_density > 0).sum(dim=-1) + 1e-9)
        return frac_dense > 0.6

    def gen_log_mask_shrinked(self, s, device):
        bs = self.config.block_size
        out = torch.zeros((s // bs, s // bs), device=device, dtype=torch.bool)
        tpf = self.video_token_num // self.num_frame
        border = self.video_token_num // bs
        out[border:] = True
        out[:, border:] = True
        delta = (torch.arange(tpf, device=device).view(1, -1) - torch.arange(tpf, device=device).view(-1, 1)).abs()
        padded_rows = ((tpf - 1) // bs + 1) * bs
        for i in range(self.num_frame):
            for j in range(self.num_frame):
                if j == 0 and self.config.model_type == "wan":
                    local = torch.ones((tpf, tpf), device=device, dtype=torch.bool)          # attention sink
                elif self._pair_is_kept(i, j, tpf):
                    local = delta <= self.get_window_width(i, j, tpf)
                else:
                    local = torch.zeros((tpf, tpf), device=device, dtype=torch.bool)
                r_off, c_off = (i * tpf) % bs, (j * tpf) % bs
                padded = torch.zeros((r_off + padded_rows, c_off + padded_rows), device=device, dtype=torch.bool)
                padded[r_off:r_off + tpf, c_off:c_off + tpf] = local
                blk = self.shrinkMaskStrict(padded)
                r0, c0 = (i * tpf) // bs, (j * tpf) // bs
                out[r0:r0 + blk.shape[0], c0:c0 + blk.shape[1]] |= blk
        return out

    def queryLogMask(self, query):
        return self.kernel_mask(query.shape[0], query.device)

    def kernel_mask(self, tokens: int, device) -> torch.Tensor:
        """int8 [tokens/128, tokens/64]: the log mask in the HIP attention kernel's block contract (the reference's
        sparge_mask_convert for a non-sm90 part, xsparse.py:227-250)."""
        key = (tokens, self.video_token_num, self.num_frame, str(device))
        if key not in self._masks:
            lm = self.gen_log_mask_shrinked(tokens, device)
            bs = self.config.block_size
            assert bs in (128, 64), "Radial Attention only supports block size of 128 or 64"
            if bs == 128:
                km = torch.repeat_interleave(lm, 2, dim=1)
            else:
                km = lm.view(lm.shape[0] // 2, 2, lm.shape[1]).max(dim=1).values
            self._masks[key] = km.to(torch.int8).contiguous()
        return self._masks[key]

    def apply(self, query, key, value, pre_defined_mask=None):
        """query/key/value: [tokens, heads, head_dim] (batch 1, as the reference's caller flattens it)."""
        from fastdm_amd.kernel.timing import span
        assert pre_defined_mask is None, "the text-masked variant (HunyuanVideo) is not built"
        t, h, d = query.shape
        km = self.kernel_mask(t, query.device)
        ck = ("expanded", t, h, str(query.device))
        if ck not in self._masks:
            self._masks[ck] = (km[None, None].expand(1, h, -1, -1).contiguous(), float(km.float().mean().item()))
        mask, density = self._masks[ck]
        v = self.video_token_num
        from fastdm_amd.kernel.utils import attention_pv_fp8
        mode = attention_pv_fp8() if d == 128 else 0      # the span names the kernel that runs (operators_set._attn routes on the same rule)
        with span(("attn_sparse_bf16", "attn_sparse_pvfp8", "attn_sparse_fp8")[mode], 4.0 * v * v * h * d * density):     # algorithmic work = the kept blocks only
            out = sparse_scaled_dot_product_attention(query[:v].reshape(1, v, h * d), key[:v].reshape(1, v, h * d),
                                                      value[:v].reshape(1, v, h * d), num_q_heads=h, num_kv_heads=h, head_dim=d,
                                                      is_causal=False, scale=1.0 / (d ** 0.5), sparse_mask=mask)
        return out.reshape(v, h, d)
