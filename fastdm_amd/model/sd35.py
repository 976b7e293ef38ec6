"""Stable Diffusion 3 / 3.5 MMDiT core on the HIP library (mirror of fastdm/model/sd35.py:32-422).

Same classes, constructor arguments, diffusers weight names and `forward` signature/returns as the reference.  A
JointTransformerBlock is the double-stream block of FLUX / Qwen-Image without RoPE and with 64-wide heads, optionally
with a second, image-only attention fed by the same LayerNorm output (SD3.5's dual-attention layers) and, in the last
block, a text stream that ends in the attention (`context_pre_only`): it runs on `fused_double_stream` (model/flux.py).
All AdaLN projections of a forward are one bf16 GEMM (`ModulationTable`) -- except `norm_out.linear`, which the
reference quantises together with `proj_out` (sd35.py:326-329) and which therefore stays a QLinear of its own.
"""
from typing import Any, Dict, List, Optional, Tuple

import torch

from fastdm_amd.layer.embeddings import CombinedTimestepTextProjEmbeddings, PatchEmbed
from fastdm_amd.layer.normalization import (AdaLayerNormContinuous, AdaLayerNormZero, ModulationTable, SD35AdaLayerNormZeroX)
from fastdm_amd.layer.qlinear import QLinear
from fastdm_amd.layer.transformer import Attention, FeedForward
from fastdm_amd.model.basemodel import BaseModelCore
from fastdm_amd.model.flux import fused_double_stream

BF16 = torch.bfloat16


class JointTransformerBlock:
    """fastdm/model/sd35.py:32-200."""

    def __init__(self, dim: int, num_attention_heads: int, attention_head_dim: int, context_pre_only: bool = False,
                 qk_norm: Optional[str] = None, use_dual_attention: bool = False, data_type: torch.dtype = BF16):
        if qk_norm != "rms_norm":
            # upstream cannot run this either: its loader reads attn.norm_q / norm_k / norm_added_* from the checkpoint
            # unconditionally (fastdm/model/sd35.py:303-304,311-312) and its Attention always applies them
            # (layer/transformer.py:203,274-275), so an SD3.0 checkpoint (no q/k norm weights) fails in init_weight there:
            # there is no reference behaviour to match
            raise NotImplementedError("SD3.0 (qk_norm=None): the reference itself cannot load a checkpoint without q/k norm weights "
                                      "(fastdm/model/sd35.py:303-304); SD3.5 uses rms_norm")
        self.use_dual_attention, self.context_pre_only = use_dual_attention, context_pre_only
        self.norm1 = SD35AdaLayerNormZeroX(dim, data_type=data_type) if use_dual_attention else AdaLayerNormZero(dim, data_type=data_type)
        self.norm1_context = (AdaLayerNormContinuous(dim, dim, elementwise_affine=False, eps=1e-6, bias=True, data_type=data_type)
                              if context_pre_only else AdaLayerNormZero(dim, data_type=data_type))
        self.attn = Attention(query_dim=dim, added_kv_proj_dim=dim, dim_head=attention_head_dim, heads=num_attention_heads, out_dim=dim,
                              context_pre_only=context_pre_only, bias=True, qk_norm=qk_norm, eps=1e-6, data_type=data_type)
        self.attn2 = (Attention(query_dim=dim, dim_head=attention_head_dim, heads=num_attention_heads, out_dim=dim, bias=True,
                                qk_norm=qk_norm, eps=1e-6, data_type=data_type) if use_dual_attention else None)
        self.ff = FeedForward(dim=dim, dim_out=dim, activation_fn="gelu-approximate", data_type=data_type)
        self.ff_context = None if context_pre_only else FeedForward(dim=dim, dim_out=dim, activation_fn="gelu-approximate", data_type=data_type)

    def forward(self, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor, temb: torch.Tensor,
                joint_attention_kwargs: Optional[Dict[str, Any]] = None):
        return fused_double_stream(self.attn, self.ff, self.ff_context, self.norm1.modulation(temb),
                                   self.norm1_context.modulation(temb.to(encoder_hidden_states.dtype)),
                                   hidden_states, encoder_hidden_states, None, attn2=self.attn2, eps_img=self.norm1.norm_eps,
                                   context_pre_only=self.context_pre_only)


class _Config:
    pass


class SD3TransformerModelCore(BaseModelCore):
    """fastdm/model/sd35.py:202-422."""

    def __init__(self, sample_size: int = 128, patch_size: int = 2, in_channels: int = 16, num_layers: int = 24,
                 attention_head_dim: int = 64, num_attention_heads: int = 24, joint_attention_dim: int = 4096,
                 caption_projection_dim: int = 1536, pooled_projection_dim: int = 2048, out_channels: int = 16,
                 pos_embed_max_size: int = 384, dual_attention_layers: Tuple[int, ...] = tuple(range(13)),
                 qk_norm: Optional[str] = "rms_norm", data_type=BF16, quant_dtype: Optional[torch.dtype] = torch.float8_e4m3fn, cache=None):
        super().__init__(type="DiT")
        cfg = _Config()
        (cfg.in_channels, cfg.joint_attention_dim, cfg.sample_size, cfg.num_attention_heads, cfg.attention_head_dim, cfg.patch_size,
         cfg.pooled_projection_dim, cfg.caption_projection_dim, cfg.num_layers) = (
            in_channels, joint_attention_dim, sample_size, num_attention_heads, attention_head_dim, patch_size, pooled_projection_dim,
            caption_projection_dim, num_layers)
        self.config = cfg
        self.out_channels = out_channels if out_channels is not None else in_channels
        self.inner_dim = num_attention_heads * attention_head_dim
        assert caption_projection_dim == self.inner_dim, "the context embedder feeds the joint attention directly"
        self.quant_dtype = quant_dtype
        self.cache = cache
        self.cache_config = cache.config if cache is not None else None
        self.enable_caching = bool(self.cache_config.enable_caching) if cache is not None else False
        self.pos_embed = PatchEmbed(height=sample_size, width=sample_size, patch_size=patch_size, in_channels=in_channels,
                                    embed_dim=self.inner_dim, pos_embed_max_size=pos_embed_max_size, data_type=data_type)
        self.time_text_embed = CombinedTimestepTextProjEmbeddings(embedding_dim=self.inner_dim, pooled_projection_dim=pooled_projection_dim,
                                                                  data_type=data_type)
        self.context_embedder = QLinear(joint_attention_dim, caption_projection_dim, bias=True, data_type=data_type)
        self.transformer_blocks = [JointTransformerBlock(dim=self.inner_dim, num_attention_heads=num_attention_heads,
                                                         attention_head_dim=attention_head_dim, context_pre_only=(i == num_layers - 1),
                                                         qk_norm=qk_norm, use_dual_attention=(i in dual_attention_layers), data_type=data_type)
                                   for i in range(num_layers)]
        self.norm_out = AdaLayerNormContinuous(self.inner_dim, self.inner_dim, elementwise_affine=False, eps=1e-6, data_type=data_type)
        self.proj_out = QLinear(self.inner_dim, patch_size * patch_size * self.out_channels, bias=True, data_type=data_type)
        self.mod_table = None

    # ------------------------------------------------------------------ loading (sd35.py:286-333)
    def _pre_part_loading(self):
        w = self.init_weight(["pos_embed.proj.weight"])                     # [D, C, p, p]
        self.pos_embed.proj.device = self.device
        self.pos_embed.proj.bias = self.init_weight(["pos_embed.proj.bias"])
        self.pos_embed.proj.set_weight_nk(w.reshape(w.shape[0], -1).contiguous(), None)
        self.pos_embed.pos_embed = self.init_weight(["pos_embed.pos_embed"])
        t = self.time_text_embed
        self.init_weight(["time_text_embed.timestep_embedder.linear_1"], t.timestep_embedder.linear1)
        self.init_weight(["time_text_embed.timestep_embedder.linear_2"], t.timestep_embedder.linear2)
        self.init_weight(["time_text_embed.text_embedder.linear_1"], t.text_embedder.linear1)
        self.init_weight(["time_text_embed.text_embedder.linear_2"], t.text_embedder.linear2)
        self.init_weight(["context_embedder"], self.context_embedder)

    def _major_parts_loading(self):
        q = self.quant_dtype
        for i, blk in enumerate(self.transformer_blocks):
            p = f"transformer_blocks.{i}"
            self.init_weight([f"{p}.norm1.linear"], blk.norm1.linear)
            self.init_weight([f"{p}.norm1_context.linear"], blk.norm1_context.linear)
            blk.attn.norm_q_weight = self.init_weight([f"{p}.attn.norm_q.weight"])
            blk.attn.norm_k_weight = self.init_weight([f"{p}.attn.norm_k.weight"])
            self.init_weight([f"{p}.attn.to_q", f"{p}.attn.to_k", f"{p}.attn.to_v"], blk.attn.qkv, q)
            self.init_weight([f"{p}.attn.add_q_proj", f"{p}.attn.add_k_proj", f"{p}.attn.add_v_proj"], blk.attn.add_qkv_proj, q)
            self.init_weight([f"{p}.attn.to_out.0"], blk.attn.to_out, q)
            if not blk.context_pre_only:
                self.init_weight([f"{p}.attn.to_add_out"], blk.attn.to_add_out, q)
            blk.attn.norm_added_q_weight = self.init_weight([f"{p}.attn.norm_added_q.weight"])
            blk.attn.norm_added_k_weight = self.init_weight([f"{p}.attn.norm_added_k.weight"])
            if blk.use_dual_attention:
                blk.attn2.norm_q_weight = self.init_weight([f"{p}.attn2.norm_q.weight"])
                blk.attn2.norm_k_weight = self.init_weight([f"{p}.attn2.norm_k.weight"])
                self.init_weight([f"{p}.attn2.to_q", f"{p}.attn2.to_k", f"{p}.attn2.to_v"], blk.attn2.qkv, q)
                self.init_weight([f"{p}.attn2.to_out.0"], blk.attn2.to_out, q)
            self.init_weight([f"{p}.ff.net.0.proj"], blk.ff.act_fn.proj, q)
            self.init_weight([f"{p}.ff.net.2"], blk.ff.ff_out_proj, q)
            if not blk.context_pre_only:
                self.init_weight([f"{p}.ff_context.net.0.proj"], blk.ff_context.act_fn.proj, q)
                self.init_weight([f"{p}.ff_context.net.2"], blk.ff_context.ff_out_proj, q)

    def _post_part_loading(self):
        self.init_weight(["norm_out.linear"], self.norm_out.linear, self.quant_dtype)
        self.init_weight(["proj_out"], self.proj_out, self.quant_dtype)

    def _finalize_loading(self):
        layers = []
        for blk in self.transformer_blocks:
            layers += [blk.norm1, blk.norm1_context]
        self.mod_table = ModulationTable(layers)

    def run_blocks(self, hidden_states, encoder_hidden_states, temb, image_rotary_emb=None, joint_attention_kwargs=None, start_block: int = 0,
                   block_controlnet_hidden_states=None, **_):
        """Blocks [start_block:] (the step caches run the first block(s) themselves); ControlNet residuals are added after every
        block but the context_pre_only one (sd35.py:401-404)."""
        nb = len(self.transformer_blocks)
        for index_block in range(start_block, nb):
            block = self.transformer_blocks[index_block]
            encoder_hidden_states, hidden_states = block.forward(hidden_states=hidden_states, encoder_hidden_states=encoder_hidden_states,
                                                                 temb=temb, joint_attention_kwargs=joint_attention_kwargs)
            if block_controlnet_hidden_states is not None and block.context_pre_only is False:
                hidden_states = hidden_states + block_controlnet_hidden_states[index_block // (nb // len(block_controlnet_hidden_states))]
        return hidden_states

    # ------------------------------------------------------------------ forward (sd35.py:335-422)
    def forward(self, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor = None, pooled_projections: torch.Tensor = None,
                timestep: torch.Tensor = None, block_controlnet_hidden_states: List = None,
                joint_attention_kwargs: Optional[Dict[str, Any]] = None, return_dict: bool = True):
        dev = self.proj_out.weight.device
        hidden_states, encoder_hidden_states = hidden_states.to(dev), encoder_hidden_states.to(dev)
        pooled_projections, timestep = pooled_projections.to(dev), timestep.to(dev)
        height, width = hidden_states.shape[-2:]
        x = self.pos_embed.forward(hidden_states)
        temb = self.time_text_embed.forward(timestep, pooled_projections)
        enc = self.context_embedder.forward(encoder_hidden_states)
        self.mod_table.begin_step(temb)
        try:
            if self.enable_caching:       # sd35.py:384-392
                if block_controlnet_hidden_states is not None:
                    raise NotImplementedError("ControlNet residuals together with a step cache are not built")
                x = self.cache.apply_cache(model_type="sd35", hidden_states=x, encoder_hidden_states=enc, temb=temb,
                                           attention_kwargs=joint_attention_kwargs, transformer_blocks=self.transformer_blocks, core=self)
            else:
                x = self.run_blocks(x, enc, temb, None, joint_attention_kwargs, block_controlnet_hidden_states=block_controlnet_hidden_states)
        finally:
            self.mod_table.end_step()
        x = self.norm_out.forward(x.contiguous(), temb)
        x = self.proj_out.forward(x)
        p = self.config.patch_size
        h, w = height // p, width // p
        x = x.reshape(x.shape[0], h, w, p, p, self.out_channels)
        x = torch.einsum("nhwpqc->nchpwq", x)
        return (x.reshape(x.shape[0], self.out_channels, h * p, w * p),)
