"""Wan2.x video transformer core on the HIP library (mirror of fastdm/model/wan.py:20-379), T2V path.

Same classes, constructor arguments, diffusers weight names and `forward` signature/returns as the reference.
Block wiring on gfx950 (rounding order of the reference's fp32 modulation kept, wan.py:80-115):

  mods = scale_shift_table + temb.float()                                       (fp32 [B, 6, D], host-size)
  LN_f32 + (1+scale)*x+shift + one rounding + per-token quant   (fdm_ln_modulate_f32)
  -> qkv GEMM -> q/k RMS-norm across heads (strided rows of the fused projection) -> RoPE in place
  -> dense or radial block-sparse attention (v read in place from the fused projection)
  -> row quant -> to_out GEMM with the fp32-gate residual epilogue  x = bf16(x + y * gate)   (one rounding)
  -> affine LN_f32 (norm2) + quant -> cross-attention to the text (to_q / fused to_kv GEMMs, RMS-norms, dense
     attention Sq x 512) -> to_out GEMM with residual epilogue (gate = 1: a plain bf16 add)
  -> LN_f32 modulate + quant -> FFN (tanh-GELU epilogue) -> fp32-gate residual epilogue.

The 3-D patch embedding is a strided convolution with kernel == stride, i.e. a GEMM over non-overlapping patches:
the latent is unfolded once (torch view ops) and goes through the bf16 GEMM kernel.
"""
import math
from typing import Any, Dict, Optional, Tuple, Union

import torch

from fastdm_amd import _native as N
from fastdm_amd.kernel import fused
from fastdm_amd.layer.embeddings import WanRotaryPosEmbed, WanTimeTextImageEmbedding
from fastdm_amd.layer.normalization import FP32LayerNorm
from fastdm_amd.layer.qlinear import QLinear
from fastdm_amd.layer.transformer import FeedForward, WanAttention
from fastdm_amd.model.basemodel import BaseModelCore

BF16 = torch.bfloat16


class WanTransformerBlock:
    """fastdm/model/wan.py:20-115."""

    def __init__(self, dim: int, ffn_dim: int, num_heads: int, qk_norm: str = "rms_norm_across_heads", cross_attn_norm: bool = False,
                 eps: float = 1e-6, added_kv_proj_dim: Optional[int] = None, data_type=BF16):
        self.dim, self.eps, self.cross_attn_norm = dim, eps, cross_attn_norm
        self.norm1 = FP32LayerNorm(dim, eps, elementwise_affine=False)
        self.attn1 = WanAttention(dim=dim, heads=num_heads, dim_head=dim // num_heads, eps=eps, cross_attention_dim_head=None, data_type=data_type)
        self.attn2 = WanAttention(dim=dim, heads=num_heads, dim_head=dim // num_heads, eps=eps, added_kv_proj_dim=added_kv_proj_dim,
                                  cross_attention_dim_head=dim // num_heads, data_type=data_type)
        if cross_attn_norm:
            self.norm2 = FP32LayerNorm(dim, eps, elementwise_affine=True)
        self.ffn = FeedForward(dim, inner_dim=ffn_dim, activation_fn="gelu-approximate", data_type=data_type)
        self.norm3 = FP32LayerNorm(dim, eps, elementwise_affine=False)
        self.scale_shift_table = None      # [1, 6, dim], loaded
        self._ones = None

    def forward(self, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor, temb: torch.Tensor, rotary_emb,
                sparse_attn=None, layer_index: Optional[int] = None, encoder_q=None) -> torch.Tensor:
        """hidden_states [B, S, D] bf16; temb [B, 6, D], or [B, S, 6, D] for per-token modulation (wan2.2 ti2v, wan.py:77-88: the
        fp32 LN-modulate kernel and the fp32 gate epilogue then index their vectors per row instead of per batch element);
        rotary_emb: the merged [S, head_dim] table (or the reference's (cos, sin) pair); encoder_q: optional pre-quantised text
        rows shared by all blocks."""
        B, S, D = hidden_states.shape
        x = hidden_states.reshape(B * S, D)
        if temb.ndim == 4:
            shift_msa, scale_msa, gate_msa, c_shift, c_scale, c_gate = [t.reshape(B * S, D).contiguous() for t in
                                                                        (self.scale_shift_table.unsqueeze(0) + temb.float()).chunk(6, dim=2)]
            rpb = 1                      # rows per modulation vector
        else:
            shift_msa, scale_msa, gate_msa, c_shift, c_scale, c_gate = [t.squeeze(1).contiguous() for t in
                                                                        (self.scale_shift_table + temb.float()).chunk(6, dim=1)]
            rpb = S
        rope = rotary_emb
        if rope is not None and not torch.is_tensor(rope):
            rope = torch.cat((rope[0].squeeze()[:, 0::2], rope[1].squeeze()[:, 1::2]), dim=-1).to(hidden_states.dtype).contiguous()
        a1, a2 = self.attn1, self.attn2
        quant = a1.qkv.quant_type

        def pick(pair):
            return pair[1] if quant is not None else pair[0]

        # 1. self-attention
        n = pick(fused.ln_modulate_f32(x, scale_msa, shift_msa, rpb, quant=quant, eps=self.eps))
        o = a1.attend(n, B, S, rope, sparse_attn=sparse_attn, layer_index=layer_index or 0)
        oq = a1.to_out.quantize_input(o) if a1.to_out.quant_type is not None else o
        x1 = torch.empty_like(x)
        a1.to_out.forward_q(oq, epilogue=N.EPI_GATE_RESIDUAL_F32, gate=gate_msa, res=x, rows_per_batch=rpb, out=x1)
        # 2. cross-attention (plain bf16 residual add = gate 1)
        if self.cross_attn_norm:
            n = pick(self.norm2.forward_bf16(x1, quant=quant))
        else:
            n = a2.to_q.quantize_input(x1) if quant is not None else x1
        img_q, s_img = None, 0
        if a2.add_k_proj is not None:     # Wan2.1-I2V: the image-context rows sit in front of the 512 text rows
            s_img = encoder_hidden_states.shape[1] - a2.TEXT_LEN
            if isinstance(encoder_q, tuple):
                encoder_q, img_q = encoder_q
            else:
                encoder_q = None
            if img_q is None:
                ei = encoder_hidden_states[:, :s_img].reshape(B * s_img, -1).contiguous()
                img_q = a2.add_k_proj.quantize_input(ei) if a2.add_k_proj.quant_type is not None else ei
            encoder_hidden_states = encoder_hidden_states[:, s_img:]
        s_txt = encoder_hidden_states.shape[1]
        if encoder_q is None:
            e = encoder_hidden_states.reshape(B * s_txt, -1).contiguous()
            encoder_q = a2.to_kv.quantize_input(e) if a2.to_kv.quant_type is not None else e
        o = a2.attend(n, B, S, None, encoder_q, s_txt, img_q=img_q, img_len=s_img)
        oq = a2.to_out.quantize_input(o) if a2.to_out.quant_type is not None else o
        if self._ones is None:
            self._ones = torch.ones((1, D), device=x.device, dtype=BF16)
        x2 = torch.empty_like(x)
        a2.to_out.forward_q(oq, epilogue=N.EPI_GATE_RESIDUAL, gate=self._ones, res=x1, rows_per_batch=B * S, out=x2)
        # 3. feed-forward
        n = pick(fused.ln_modulate_f32(x2, c_scale, c_shift, rpb, quant=quant, eps=self.eps))
        x3 = self.ffn.forward_fused(n, c_gate, x2, rpb, gate_epilogue=N.EPI_GATE_RESIDUAL_F32)
        return x3.view(B, S, D)


class WanTransformer3DModelCore(BaseModelCore):
    """fastdm/model/wan.py:117-379."""

    def __init__(self, patch_size: Tuple[int] = (1, 2, 2), num_attention_heads: int = 40, attention_head_dim: int = 128,
                 in_channels: int = 16, out_channels: int = 16, text_dim: int = 4096, freq_dim: int = 256, ffn_dim: int = 13824,
                 num_layers: int = 40, cross_attn_norm: bool = True, qk_norm: Optional[str] = "rms_norm_across_heads", eps: float = 1e-6,
                 image_dim: Optional[int] = None, added_kv_proj_dim: Optional[int] = None, rope_max_seq_len: int = 1024,
                 pos_embed_seq_len: Optional[int] = None, data_type=BF16, quant_dtype: Optional[torch.dtype] = torch.float8_e4m3fn,
                 cache=None, sparse_attn=None) -> None:
        super().__init__("DiT")
        self.quant_dtype = quant_dtype
        inner_dim = num_attention_heads * attention_head_dim
        self.inner_dim, self.eps = inner_dim, eps
        self.out_channels = out_channels or in_channels
        self.rope = WanRotaryPosEmbed(attention_head_dim, patch_size, rope_max_seq_len)
        self.patch_embedding_kernel_size = self.patch_embedding_stride = tuple(patch_size)
        self.patch_embedding = QLinear(in_channels * math.prod(patch_size), inner_dim, data_type=data_type)   # conv3d(kernel == stride) as a GEMM
        self.condition_embedder = WanTimeTextImageEmbedding(dim=inner_dim, time_freq_dim=freq_dim, time_proj_dim=inner_dim * 6,
                                                            text_embed_dim=text_dim, image_embed_dim=image_dim,
                                                            pos_embed_seq_len=pos_embed_seq_len, data_type=data_type)
        self.blocks = [WanTransformerBlock(inner_dim, ffn_dim, num_attention_heads, qk_norm, cross_attn_norm, eps, added_kv_proj_dim,
                                           data_type=data_type) for _ in range(num_layers)]
        self.norm_out = FP32LayerNorm(inner_dim, eps, elementwise_affine=False)
        self.proj_out = QLinear(inner_dim, self.out_channels * math.prod(patch_size), data_type=data_type)
        self.scale_shift_table = None
        self.cache = cache
        self.cache_config = cache.config if cache is not None else None
        self.enable_caching = bool(self.cache_config.enable_caching) if cache is not None else False
        self._enc_q = None
        self.sparse_attn = sparse_attn

    # ------------------------------------------------------------------ loading (wan.py:225-281)
    def _pre_part_loading(self):
        w = self.init_weight(["patch_embedding.weight"])                  # [D, C, pt, ph, pw]
        b = self.init_weight(["patch_embedding.bias"])
        self.patch_embedding.device = self.device
        self.patch_embedding.bias = b
        self.patch_embedding.set_weight_nk(w.reshape(w.shape[0], -1).contiguous(), None)
        ce = self.condition_embedder
        self.init_weight(["condition_embedder.time_embedder.linear_1"], ce.time_embedder.linear1)
        self.init_weight(["condition_embedder.time_embedder.linear_2"], ce.time_embedder.linear2)
        self.init_weight(["condition_embedder.time_proj"], ce.time_proj)
        self.init_weight(["condition_embedder.text_embedder.linear_1"], ce.text_embedder.linear1)
        self.init_weight(["condition_embedder.text_embedder.linear_2"], ce.text_embedder.linear2)
        ie = ce.image_embedder
        if ie is not None:                                                   # wan.py:234-243
            ie.norm1.weight = self.init_weight(["condition_embedder.image_embedder.norm1.weight"]).to(torch.float32)
            ie.norm1.bias = self.init_weight(["condition_embedder.image_embedder.norm1.bias"]).to(torch.float32)
            self.init_weight(["condition_embedder.image_embedder.ff.net.0.proj"], ie.ff.act_fn.proj)
            self.init_weight(["condition_embedder.image_embedder.ff.net.2"], ie.ff.ff_out_proj)
            ie.norm2.weight = self.init_weight(["condition_embedder.image_embedder.norm2.weight"]).to(torch.float32)
            ie.norm2.bias = self.init_weight(["condition_embedder.image_embedder.norm2.bias"]).to(torch.float32)

    def _major_parts_loading(self):
        q = self.quant_dtype
        for i, blk in enumerate(self.blocks):
            p = f"blocks.{i}"
            blk.attn1.norm_q_weight = self.init_weight([f"{p}.attn1.norm_q.weight"])
            blk.attn1.norm_k_weight = self.init_weight([f"{p}.attn1.norm_k.weight"])
            self.init_weight([f"{p}.attn1.to_q", f"{p}.attn1.to_k", f"{p}.attn1.to_v"], blk.attn1.qkv, q)
            self.init_weight([f"{p}.attn1.to_out.0"], blk.attn1.to_out, q)
            blk.attn2.norm_q_weight = self.init_weight([f"{p}.attn2.norm_q.weight"])
            blk.attn2.norm_k_weight = self.init_weight([f"{p}.attn2.norm_k.weight"])
            self.init_weight([f"{p}.attn2.to_q"], blk.attn2.to_q, q)
            self.init_weight([f"{p}.attn2.to_k", f"{p}.attn2.to_v"], blk.attn2.to_kv, q)
            self.init_weight([f"{p}.attn2.to_out.0"], blk.attn2.to_out, q)
            if blk.attn2.added_kv_proj_dim is not None:                      # wan.py:262-265
                self.init_weight([f"{p}.attn2.add_k_proj"], blk.attn2.add_k_proj, q)
                self.init_weight([f"{p}.attn2.add_v_proj"], blk.attn2.add_v_proj, q)
                blk.attn2.norm_added_k_weight = self.init_weight([f"{p}.attn2.norm_added_k.weight"])
            if blk.cross_attn_norm:
                blk.norm2.weight = self.init_weight([f"{p}.norm2.weight"]).to(torch.float32)
                blk.norm2.bias = self.init_weight([f"{p}.norm2.bias"]).to(torch.float32)
            self.init_weight([f"{p}.ffn.net.0.proj"], blk.ffn.act_fn.proj, q)
            self.init_weight([f"{p}.ffn.net.2"], blk.ffn.ff_out_proj, q)
            blk.scale_shift_table = self.init_weight([f"{p}.scale_shift_table"])

    def _post_part_loading(self):
        self.init_weight(["proj_out"], self.proj_out)
        self.scale_shift_table = self.init_weight(["scale_shift_table"])

    # ------------------------------------------------------------------ forward (wan.py:283-379)
    def run_blocks(self, hidden_states, encoder_hidden_states, temb, image_rotary_emb, joint_attention_kwargs=None, start_block: int = 0, **_):
        """Blocks [start_block:] with their absolute layer index (the sparse-attention schedule depends on it; the reference's
        cache path numbers the remaining blocks from 1, caching/xcaching.py:91-96)."""
        for layer_index in range(start_block, len(self.blocks)):
            hidden_states = self.blocks[layer_index].forward(hidden_states, encoder_hidden_states, temb, image_rotary_emb, self.sparse_attn,
                                                             layer_index, encoder_q=self._enc_q)
        return hidden_states

    def forward(self, hidden_states: torch.Tensor, timestep: torch.Tensor, encoder_hidden_states: torch.Tensor,
                encoder_hidden_states_image: Optional[torch.Tensor] = None, return_dict: bool = True,
                attention_kwargs: Optional[Dict[str, Any]] = None) -> Union[torch.Tensor, Tuple[torch.Tensor]]:
        dev = self.proj_out.weight.device
        hidden_states, timestep, encoder_hidden_states = hidden_states.to(dev), timestep.to(dev), encoder_hidden_states.to(dev)
        B, C, Fr, H, W = hidden_states.shape
        p_t, p_h, p_w = self.patch_embedding_kernel_size
        ppf, pph, ppw = Fr // p_t, H // p_h, W // p_w
        S = ppf * pph * ppw
        if self.sparse_attn and self.sparse_attn.config.sparse_algorithm == "radial":
            self.sparse_attn.post_init(video_token_num=S, num_frame=ppf)
        rope = self.rope.merged(Fr, H, W, dev, hidden_states.dtype)
        # patchify: [B, C, F, H, W] -> [B*S, C*pt*ph*pw] rows in (f, h, w) token order, columns in the conv weight's (c, t, h, w) order
        patches = hidden_states.reshape(B, C, ppf, p_t, pph, p_h, ppw, p_w).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B * S, -1).contiguous()
        x = self.patch_embedding.forward_q(patches).view(B, S, self.inner_dim)
        if encoder_hidden_states_image is not None:
            encoder_hidden_states_image = encoder_hidden_states_image.to(dev)
        ts_len = None
        if timestep.ndim == 2:                        # wan2.2 ti2v: [B, S] timesteps, one per token (wan.py:319-322)
            ts_len, timestep = timestep.shape[1], timestep.flatten()
        temb, timestep_proj, enc, enc_img = self.condition_embedder.forward(timestep, encoder_hidden_states, encoder_hidden_states_image,
                                                                            timestep_seq_len=ts_len)
        timestep_proj = timestep_proj.unflatten(2, (6, -1)) if ts_len is not None else timestep_proj.unflatten(1, (6, -1))
        a2 = self.blocks[0].attn2
        e2 = enc.reshape(B * enc.shape[1], -1).contiguous()
        # every block quantises the same text rows for its to_kv projection: do it once
        enc_q = a2.to_kv.quantize_input(e2) if a2.to_kv.quant_type is not None else e2
        if enc_img is not None:                                              # wan.py:333-334: image context in front of the text
            ei = enc_img.reshape(B * enc_img.shape[1], -1).contiguous()
            enc_q = (enc_q, a2.add_k_proj.quantize_input(ei) if a2.add_k_proj.quant_type is not None else ei)
            enc = torch.concat([enc_img, enc], dim=1)
        self._enc_q = enc_q
        try:
            if self.enable_caching:       # wan.py:339-349
                x = self.cache.apply_cache(model_type="wan", hidden_states=x, encoder_hidden_states=enc, temb=timestep_proj, image_rotary_emb=rope,
                                           attention_kwargs=attention_kwargs, transformer_blocks=self.blocks, sparse_attn=self.sparse_attn, core=self)
            else:
                x = self.run_blocks(x, enc, timestep_proj, rope)
        finally:
            self._enc_q = None
        # output norm: (scale_shift_table + temb) in bf16, `1 + scale` rounded to bf16 by the reference's type promotion (wan.py:353-362)
        if temb.ndim == 3:                            # per token (wan.py:355-359)
            shift, scale = [t.reshape(B * S, -1) for t in (self.scale_shift_table.unsqueeze(0) + temb.unsqueeze(2)).chunk(2, dim=2)]
            out_rpb = 1
        else:
            shift, scale = [t.squeeze(1) for t in (self.scale_shift_table + temb.unsqueeze(1)).chunk(2, dim=1)]
            out_rpb = S
        scale_f = (1 + scale).float() - 1.0
        y, _ = fused.ln_modulate_f32(x.reshape(B * S, -1), scale_f.contiguous(), shift.float().contiguous(), out_rpb, quant=None, eps=self.eps)
        y = self.proj_out.forward_q(y)
        y = y.reshape(B, ppf, pph, ppw, p_t, p_h, p_w, -1).permute(0, 7, 1, 4, 2, 5, 3, 6)
        output = y.flatten(6, 7).flatten(4, 5).flatten(2, 3)
        return (output,)
