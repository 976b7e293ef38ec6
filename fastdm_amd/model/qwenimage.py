"""Qwen-Image transformer core on the HIP library (mirror of fastdm/model/qwenimage.py:17-343).

Same classes, constructor arguments, diffusers weight names and `forward` signature/returns as the reference.
A Qwen-Image block is arithmetically the FLUX double-stream block with the modulation projection inside the block
(qwenimage.py:73-124 vs flux.py:130-178): LayerNorm -> x*(1+scale)+shift -> joint [text ; image] attention with
q/k RMS-norm and GPT-J RoPE -> gate*y + residual -> LayerNorm/modulate -> tanh-GELU MLP -> gate*y + residual, so the
block runs on `fused_double_stream` (model/flux.py).  All 120 modulation projections of a forward depend only on
`temb`; they are evaluated as ONE bf16 GEMM per forward (`ModulationTable`, 13.6 GB of weights streamed once)
instead of 120 M=B GEMVs.  W8A8 int8 (asymmetric per-token activations, the reference's choice for Qwen,
qlinear.py:73) and fp8 share the same wiring.
"""
from typing import Any, Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from fastdm_amd.layer.embeddings import QwenEmbedRope, QwenTimestepProjEmbeddings
from fastdm_amd.layer.normalization import AdaLayerNormContinuous, ModulationTable, RMSNorm
from fastdm_amd.layer.qlinear import QLinear
from fastdm_amd.layer.transformer import Attention, FeedForward
from fastdm_amd.model.basemodel import BaseModelCore
from fastdm_amd.model.flux import fused_double_stream

BF16 = torch.bfloat16


class _ModProj:
    """Adapter that lets ModulationTable batch a block's `img_mod_proj` / `txt_mod_proj` QLinear."""

    def __init__(self, linear: QLinear):
        self.linear = linear
        self._cached_mod = None

    def modulation(self, temb: torch.Tensor) -> torch.Tensor:
        if self._cached_mod is not None:
            return self._cached_mod
        return self.linear.forward(F.silu(temb))


class QwenImageTransformerBlock:
    """fastdm/model/qwenimage.py:17-124."""

    def __init__(self, dim: int, num_attention_heads: int, attention_head_dim: int, qk_norm: str = "rms_norm", eps: float = 1e-6,
                 data_type=BF16):
        self.dim, self.num_attention_heads, self.attention_head_dim, self.eps = dim, num_attention_heads, attention_head_dim, eps
        self.img_mod_proj = QLinear(dim, 6 * dim, data_type=data_type)
        self.txt_mod_proj = QLinear(dim, 6 * dim, data_type=data_type)
        self.attn = Attention(query_dim=dim, added_kv_proj_dim=dim, dim_head=attention_head_dim, heads=num_attention_heads, out_dim=dim,
                              context_pre_only=False, bias=True, qk_norm=qk_norm, eps=eps, data_type=data_type)
        self.img_mlp = FeedForward(dim=dim, dim_out=dim, activation_fn="gelu-approximate", data_type=data_type)
        self.txt_mlp = FeedForward(dim=dim, dim_out=dim, activation_fn="gelu-approximate", data_type=data_type)
        self._img_mod, self._txt_mod = _ModProj(self.img_mod_proj), _ModProj(self.txt_mod_proj)

    def _modulate(self, x, mod_params):
        """qwenimage.py:52-55 (unfused helper kept for API parity; the block itself uses the fused kernel)."""
        shift, scale, gate = mod_params.chunk(3, dim=-1)
        return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1), gate.unsqueeze(1)

    def txt_modulated(self, encoder_hidden_states: torch.Tensor, temb: torch.Tensor) -> torch.Tensor:
        """The text stream after norm1 + modulation: TeaCache's probe for Qwen-Image (xcaching.py:171-180)."""
        from fastdm_amd.kernel import fused
        B, S, D = encoder_hidden_states.shape
        shift, scale = self._txt_mod.modulation(temb)[:, : 2 * D].chunk(2, dim=1)
        y, _, _ = fused.ln_modulate(encoder_hidden_states.reshape(B * S, D), scale, shift, S, quant=None, eps=self.eps)
        return y.view(B, S, D)

    def forward(self, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor, encoder_hidden_states_mask: torch.Tensor = None,
                temb: torch.Tensor = None, image_rotary_emb: Optional[torch.Tensor] = None,
                joint_attention_kwargs: Optional[Dict[str, Any]] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        return fused_double_stream(self.attn, self.img_mlp, self.txt_mlp, self._img_mod.modulation(temb), self._txt_mod.modulation(temb),
                                   hidden_states, encoder_hidden_states, image_rotary_emb)


class QwenImageTransformer2DModelCore(BaseModelCore):
    """fastdm/model/qwenimage.py:126-343."""

    def __init__(self, patch_size: int = 2, in_channels: int = 64, out_channels: Optional[int] = 16, num_layers: int = 60,
                 attention_head_dim: int = 128, num_attention_heads: int = 24, joint_attention_dim: int = 3584,
                 guidance_embeds: bool = False, axes_dims_rope: Tuple[int, int, int] = (16, 56, 56), data_type=BF16,
                 quant_dtype: Optional[torch.dtype] = torch.float8_e4m3fn, cache=None, oom_ressolve: bool = False):
        super().__init__(type="DiT")
        self.quant_dtype = quant_dtype
        self.need_resolve_oom = oom_ressolve      # accepted for signature parity; 288 GB HBM makes it moot
        self.out_channels = out_channels or in_channels
        self.inner_dim = num_attention_heads * attention_head_dim
        self.pos_embed = QwenEmbedRope(theta=10000, axes_dim=list(axes_dims_rope), scale_rope=True)
        self.time_text_embed = QwenTimestepProjEmbeddings(embedding_dim=self.inner_dim, data_type=data_type)
        self.txt_norm = RMSNorm(joint_attention_dim, eps=1e-6)
        self.img_in = QLinear(in_channels, self.inner_dim, data_type=data_type)
        self.txt_in = QLinear(joint_attention_dim, self.inner_dim, data_type=data_type)
        self.transformer_blocks = [QwenImageTransformerBlock(dim=self.inner_dim, num_attention_heads=num_attention_heads,
                                                             attention_head_dim=attention_head_dim, data_type=data_type)
                                   for _ in range(num_layers)]
        self.norm_out = AdaLayerNormContinuous(self.inner_dim, self.inner_dim, elementwise_affine=False, eps=1e-6, data_type=data_type)
        self.proj_out = QLinear(self.inner_dim, patch_size * patch_size * self.out_channels, data_type=data_type)
        self.cache = cache
        self.cache_config = cache.config if cache is not None else None
        self.enable_caching = self.cache_config.enable_caching if cache is not None else False
        self.mod_table = None
        self._rope_key, self._rope_val = None, None

    # ------------------------------------------------------------------ loading (qwenimage.py:201-243)
    def _pre_part_loading(self):
        self.init_weight(["time_text_embed.timestep_embedder.linear_1"], self.time_text_embed.timestep_embedder.linear1)
        self.init_weight(["time_text_embed.timestep_embedder.linear_2"], self.time_text_embed.timestep_embedder.linear2)
        self.txt_norm.weight = self.init_weight(["txt_norm.weight"])
        self.init_weight(["img_in"], self.img_in)
        self.init_weight(["txt_in"], self.txt_in)

    def _major_parts_loading(self):
        q = self.quant_dtype
        for i, blk in enumerate(self.transformer_blocks):
            p = f"transformer_blocks.{i}"
            self.init_weight([f"{p}.img_mod.1"], blk.img_mod_proj)      # never quantised on a large-memory card (qwenimage.py:165-166)
            self.init_weight([f"{p}.txt_mod.1"], blk.txt_mod_proj)
            blk.attn.norm_q_weight = self.init_weight([f"{p}.attn.norm_q.weight"])
            blk.attn.norm_k_weight = self.init_weight([f"{p}.attn.norm_k.weight"])
            self.init_weight([f"{p}.attn.to_q", f"{p}.attn.to_k", f"{p}.attn.to_v"], blk.attn.qkv, q)
            self.init_weight([f"{p}.attn.add_q_proj", f"{p}.attn.add_k_proj", f"{p}.attn.add_v_proj"], blk.attn.add_qkv_proj, q)
            self.init_weight([f"{p}.attn.to_out.0"], blk.attn.to_out, q)
            self.init_weight([f"{p}.attn.to_add_out"], blk.attn.to_add_out, q)
            blk.attn.norm_added_q_weight = self.init_weight([f"{p}.attn.norm_added_q.weight"])
            blk.attn.norm_added_k_weight = self.init_weight([f"{p}.attn.norm_added_k.weight"])
            self.init_weight([f"{p}.img_mlp.net.0.proj"], blk.img_mlp.act_fn.proj, q)
            self.init_weight([f"{p}.img_mlp.net.2"], blk.img_mlp.ff_out_proj, q)
            self.init_weight([f"{p}.txt_mlp.net.0.proj"], blk.txt_mlp.act_fn.proj, q)
            self.init_weight([f"{p}.txt_mlp.net.2"], blk.txt_mlp.ff_out_proj, q)

    def _post_part_loading(self):
        self.init_weight(["norm_out.linear"], self.norm_out.linear)
        self.init_weight(["proj_out"], self.proj_out)

    def _finalize_loading(self):
        layers = []
        for blk in self.transformer_blocks:
            layers += [blk._img_mod, blk._txt_mod]
        layers.append(self.norm_out)
        self.mod_table = ModulationTable(layers)

    # ------------------------------------------------------------------ forward (qwenimage.py:245-343)
    def _rope_cache(self, img_shapes, txt_seq_lens, device, dtype):
        shp = img_shapes[0] if isinstance(img_shapes, list) else img_shapes
        key = (str(shp), max(txt_seq_lens), str(device), dtype)
        if key != self._rope_key:
            img_freqs, txt_freqs = self.pos_embed.forward(img_shapes, txt_seq_lens, device="cpu")
            img_c = torch.cat([img_freqs.real, img_freqs.imag], dim=-1).to(dtype)
            txt_c = torch.cat([txt_freqs.real, txt_freqs.imag], dim=-1).to(dtype)
            self._rope_val = torch.cat([txt_c, img_c], dim=0).contiguous().to(device)   # [text ; image], [real | imag]
            self._rope_key = key
        return self._rope_val

    def run_blocks(self, hidden_states, encoder_hidden_states, temb, image_rotary_emb, attention_kwargs=None,
                   encoder_hidden_states_mask=None, start_block: int = 0, **_):
        for block in self.transformer_blocks[start_block:]:
            encoder_hidden_states, hidden_states = block.forward(hidden_states=hidden_states, encoder_hidden_states=encoder_hidden_states,
                                                                 encoder_hidden_states_mask=encoder_hidden_states_mask, temb=temb,
                                                                 image_rotary_emb=image_rotary_emb, joint_attention_kwargs=attention_kwargs)
        return hidden_states

    def forward(self, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor = None, encoder_hidden_states_mask: torch.Tensor = None,
                timestep: torch.Tensor = None, img_shapes: Optional[List[Tuple[int, int, int]]] = None,
                txt_seq_lens: Optional[List[int]] = None, guidance: torch.Tensor = None,
                attention_kwargs: Optional[Dict[str, Any]] = None, return_dict: bool = True):
        dev = self.img_in.weight.device
        hidden_states = self.img_in.forward(hidden_states.to(dev))
        timestep = timestep.to(dev).to(hidden_states.dtype)
        encoder_hidden_states = self.txt_in.forward(self.txt_norm.forward(encoder_hidden_states.to(dev)))
        assert guidance is None, "Qwen-Image is not guidance-distilled (QwenTimestepProjEmbeddings takes the timestep only)"
        temb = self.time_text_embed.forward(timestep, hidden_states)
        image_rotary_emb = self._rope_cache(img_shapes, txt_seq_lens, dev, hidden_states.dtype)

        if self.mod_table is not None:
            self.mod_table.begin_step(temb)
        try:
            if self.enable_caching:
                hidden_states = self.cache.apply_cache(model_type="qwenimage", hidden_states=hidden_states,
                                                       encoder_hidden_states=encoder_hidden_states,
                                                       encoder_hidden_states_mask=encoder_hidden_states_mask, temb=temb,
                                                       image_rotary_emb=image_rotary_emb, attention_kwargs=attention_kwargs,
                                                       transformer_blocks=self.transformer_blocks, core=self)
            else:
                hidden_states = self.run_blocks(hidden_states, encoder_hidden_states, temb, image_rotary_emb, attention_kwargs,
                                                encoder_hidden_states_mask)
            hidden_states = self.norm_out.forward(hidden_states.contiguous(), temb)
        finally:
            if self.mod_table is not None:
                self.mod_table.end_step()
        return (self.proj_out.forward(hidden_states),)
