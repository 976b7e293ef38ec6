"""FLUX transformer core on the HIP library (mirror of fastdm/model/flux.py:17-494).

Same classes, constructor arguments, diffusers weight names and `forward` signature/returns as the
reference.  The block wiring is fused for gfx950:

  double block, per stream:  LN+modulate+quant  ->  qkv GEMM  ->  q/k RMS-norm + RoPE + scatter into
      the joint [text ; image] q/k/v  ->  flash attention  ->  row quant  ->  out-proj GEMM with
      gate*y + residual epilogue  ->  LN+modulate+quant  ->  FF1 GEMM with tanh-GELU epilogue  ->
      row quant  ->  FF2 GEMM with gate*y + residual epilogue                (19 launches / block)
  single block:  LN+modulate+quant (once, shared by proj_mlp and qkv; the reference quantises the
      same tensor twice, flux.py:61,63)  ->  qkv GEMM, proj_mlp GEMM with erf-GELU epilogue  ->
      norm/RoPE/scatter  ->  attention  ->  row quant of the *virtual* cat([attn, mlp])  ->
      proj_out GEMM with gate*y + residual epilogue                           (7 launches / block)

Rounding order follows the torch backend exactly (SURVEY §8c'); every fused kernel rounds to bf16
where the unfused torch ops would.
"""
from typing import Any, Dict, Optional, Tuple

import torch

from fastdm_amd import _native as N
from fastdm_amd.kernel import fused
from fastdm_amd.layer.embeddings import (CombinedTimestepGuidanceTextProjEmbeddings, CombinedTimestepTextProjEmbeddings,
                                         FluxPosEmbed)
from fastdm_amd.layer.normalization import (AdaLayerNormContinuous, AdaLayerNormZero, AdaLayerNormZeroSingle, ModulationTable)
from fastdm_amd.layer.qlinear import QLinear
from fastdm_amd.layer.transformer import _rows, Attention, FeedForward
from fastdm_amd.model.basemodel import BaseModelCore

BF16 = torch.bfloat16


class FluxSingleTransformerBlock:
    """fastdm/model/flux.py:17-76."""

    def __init__(self, dim, num_attention_heads, attention_head_dim, mlp_ratio=4.0, data_type=BF16):
        self.mlp_hidden_dim = int(dim * mlp_ratio)
        self.norm = AdaLayerNormZeroSingle(dim, data_type=data_type)
        self.proj_mlp = QLinear(dim, self.mlp_hidden_dim, bias=True, data_type=data_type)
        self.proj_out = QLinear(dim + self.mlp_hidden_dim, dim, bias=True, data_type=data_type)
        self.attn = Attention(query_dim=dim, dim_head=attention_head_dim, heads=num_attention_heads, out_dim=dim, bias=True,
                              qk_norm="rms_norm", eps=1e-6, pre_only=True, data_type=data_type)

    def forward(self, hidden_states: torch.Tensor, temb: torch.Tensor, image_rotary_emb=None, joint_attention_kwargs=None):
        B, S, D = hidden_states.shape
        x = hidden_states.reshape(B * S, D)
        shift, scale, gate = self.norm.modulation(temb).chunk(3, dim=1)
        quant = self.proj_mlp.quant_type
        nh, nq, _ = fused.ln_modulate(x, scale, shift, S, quant=quant, scale_has_one=self.norm.scale_has_one)
        xin = nq if quant is not None else nh
        # act_mlp (erf-GELU, flux.py:61): W8A8 -> deferred into the [attn | mlp] quantisation pass (layer/transformer.py FeedForward note),
        # bf16 -> the GEMM's epilogue
        deferred = self.proj_out.quant_type is not None
        if deferred:
            mlp = self.proj_mlp.forward_q(xin, epilogue=N.EPI_NONE)
            attn = self.attn.attend(xin, B, S, image_rotary_emb)
            cat = fused.quantize_rows(attn, mlp, self.proj_out.quant_type, act1=N.EPI_GELU_ERF)
        else:
            # bf16: the GEMM and the attention kernel write the two column ranges of proj_out's [attention | gelu(mlp)] operand directly
            # (row stride D + mlp_hidden_dim): the torch.cat of flux.py:69 never runs
            cat = torch.empty((B * S, D + self.mlp_hidden_dim), device=x.device, dtype=x.dtype)
            self.proj_mlp.forward_q(xin, epilogue=N.EPI_GELU_ERF, out=cat[:, D:])
            self.attn.attend(xin, B, S, image_rotary_emb, out=cat[:, :D])
        out = torch.empty_like(x)
        self.proj_out.forward_q(cat, epilogue=N.EPI_GATE_RESIDUAL, gate=gate, res=x, rows_per_batch=S, out=out)
        return out.view(B, S, D)


class FluxTransformerBlock:
    """fastdm/model/flux.py:78-178."""

    def __init__(self, dim, num_attention_heads, attention_head_dim, qk_norm="rms_norm", eps=1e-6, data_type=BF16):
        self.norm1 = AdaLayerNormZero(dim, data_type=data_type)
        self.norm1_context = AdaLayerNormZero(dim, data_type=data_type)
        self.attn = Attention(query_dim=dim, added_kv_proj_dim=dim, dim_head=attention_head_dim, heads=num_attention_heads,
                              out_dim=dim, context_pre_only=False, bias=True, qk_norm=qk_norm, eps=eps, data_type=data_type)
        self.ff = FeedForward(dim=dim, dim_out=dim, activation_fn="gelu-approximate", data_type=data_type)
        self.ff_context = FeedForward(dim=dim, dim_out=dim, activation_fn="gelu-approximate", data_type=data_type)

    def forward(self, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor, temb: torch.Tensor,
                image_rotary_emb=None, joint_attention_kwargs=None):
        return fused_double_stream(self.attn, self.ff, self.ff_context, self.norm1.modulation(temb), self.norm1_context.modulation(temb),
                                   hidden_states, encoder_hidden_states, image_rotary_emb,
                                   scale_has_one=(self.norm1.scale_has_one, self.norm1_context.scale_has_one))


def fused_double_stream(attn, ff_img, ff_txt, mod_img, mod_txt, hidden_states, encoder_hidden_states, image_rotary_emb,
                        attn2=None, eps_img: float = 1e-6, context_pre_only: bool = False, scale_has_one=(False, False)):
    """One double-stream (MMDiT) block on the fused kernels.  `mod_img` is the image stream's [B, 6*D] modulation projection
    in the order (shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp) -- [B, 9*D] with (shift_msa2, scale_msa2,
    gate_msa2) appended when `attn2` (SD3.5's second, image-only attention) is given; `mod_txt` likewise [B, 6*D], or
    [B, 2*D] = (scale, shift) for a `context_pre_only` last block whose text stream ends in the attention.
    Shared by the FLUX double block (flux.py:130-178), the Qwen-Image block (qwenimage.py:58-124) and SD3's
    JointTransformerBlock (sd35.py:133-200), whose arithmetic is identical.  `scale_has_one` = (image, text): the scale chunks of that
    stream's modulation already hold bf16(1 + scale) (ModulationTable(fold_one=True)).  Returns (encoder_hidden_states | None, hidden_states)."""
    B, s_img, D = hidden_states.shape
    s_txt = encoder_hidden_states.shape[1]
    quant = attn.qkv.quant_type
    x = hidden_states.reshape(B * s_img, D)
    e = encoder_hidden_states.reshape(B * s_txt, D)
    im = mod_img.chunk(9 if attn2 is not None else 6, dim=1)
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = im[:6]
    if context_pre_only:
        c_scale_msa, c_shift_msa = mod_txt.chunk(2, dim=1)          # AdaLayerNormContinuous: scale first
    else:
        c_shift_msa, c_scale_msa, c_gate_msa, c_shift_mlp, c_scale_mlp, c_gate_mlp = mod_txt.chunk(6, dim=1)

    def pick(t):
        return t[1] if quant is not None else t[0]

    one_img, one_txt = scale_has_one
    xin = pick(fused.ln_modulate(x, scale_msa, shift_msa, s_img, quant=quant, eps=eps_img, scale_has_one=one_img))
    ein = pick(fused.ln_modulate(e, c_scale_msa, c_shift_msa, s_txt, quant=quant, scale_has_one=one_txt))
    o = attn.attend(xin, B, s_img, image_rotary_emb, ein, s_txt).view(B, s_txt + s_img, D)

    # attention rows of a stream are a strided slice of the joint output [B, s_txt + s_img, D].  W8A8: every batch element's rows are quantised
    # into its slice of ONE compact operand per stream, then ONE launch projects image and text rows of the whole batch (round 6: a CFG batch of
    # two -- Qwen-Image, SD3.5 -- ran two launches of 1.3 rounds of tiles each: Qwen-Image to_out 2 x 98 us at 0.27 of peak).  bf16: per batch element.
    x_new = torch.empty_like(x)
    e_new = None if context_pre_only else torch.empty_like(e)
    to_o = attn.to_out
    to_ao = None if context_pre_only else attn.to_add_out
    if to_o.quant_type is not None and (to_ao is None or to_ao.quant_type == to_o.quant_type):
        aq_img = fused._alloc_q(B * s_img, D, to_o.quant_type, x.device)
        aq_txt = None if to_ao is None else fused._alloc_q(B * s_txt, D, to_ao.quant_type, x.device)
        for b in range(B):
            to_o.quantize_input(o[b, s_txt:], out=_rows(aq_img, b, s_img, B))
            if to_ao is not None:
                to_ao.quantize_input(o[b, :s_txt], out=_rows(aq_txt, b, s_txt, B))
        if to_ao is None:
            to_o.forward_q(aq_img, epilogue=N.EPI_GATE_RESIDUAL, gate=gate_msa, res=x, rows_per_batch=s_img, out=x_new)
        else:
            QLinear.forward_q_pair(to_o, aq_img, to_ao, aq_txt, epilogue=N.EPI_GATE_RESIDUAL, gate0=gate_msa, res0=x, rpb0=s_img, out0=x_new,
                                   gate1=c_gate_msa, res1=e, rpb1=s_txt, out1=e_new)
    else:
        for b in range(B):
            ai = o[b, s_txt:]
            aqi = to_o.quantize_input(ai) if to_o.quant_type is not None else ai
            if context_pre_only:
                to_o.forward_q(aqi, epilogue=N.EPI_GATE_RESIDUAL, gate=gate_msa[b:b + 1], res=x[b * s_img:(b + 1) * s_img], rows_per_batch=s_img,
                               out=x_new[b * s_img:(b + 1) * s_img])
                continue
            at = o[b, :s_txt]
            aqt = to_ao.quantize_input(at) if to_ao.quant_type is not None else at
            QLinear.forward_q_pair(to_o, aqi, to_ao, aqt, epilogue=N.EPI_GATE_RESIDUAL,
                                   gate0=gate_msa[b:b + 1], res0=x[b * s_img:(b + 1) * s_img], rpb0=s_img, out0=x_new[b * s_img:(b + 1) * s_img],
                                   gate1=c_gate_msa[b:b + 1], res1=e[b * s_txt:(b + 1) * s_txt], rpb1=s_txt, out1=e_new[b * s_txt:(b + 1) * s_txt])
    if attn2 is not None:
        # second, image-only attention on the same LayerNorm output with its own modulation (sd35.py:166-169)
        shift_msa2, scale_msa2, gate_msa2 = im[6:]
        xin2 = pick(fused.ln_modulate(x, scale_msa2, shift_msa2, s_img, quant=quant, eps=eps_img, scale_has_one=one_img))
        o2 = attn2.attend(xin2, B, s_img, None)
        oq2 = attn2.to_out.quantize_input(o2) if attn2.to_out.quant_type is not None else o2
        attn2.to_out.forward_q(oq2, epilogue=N.EPI_GATE_RESIDUAL, gate=gate_msa2, res=x_new, rows_per_batch=s_img, out=x_new)
    fin_x = pick(fused.ln_modulate(x_new, scale_mlp, shift_mlp, s_img, quant=quant, scale_has_one=one_img))
    if context_pre_only:
        x_out = ff_img.forward_fused(fin_x, gate_mlp, x_new, s_img)
        return None, x_out.view(B, s_img, D)
    fin_e = pick(fused.ln_modulate(e_new, c_scale_mlp, c_shift_mlp, s_txt, quant=quant, scale_has_one=one_txt))
    x, e = FeedForward.forward_fused_pair(ff_img, fin_x, gate_mlp, x_new, s_img, ff_txt, fin_e, c_gate_mlp, e_new, s_txt)
    return e.view(B, s_txt, D), x.view(B, s_img, D)


class FluxTransformer2DModelCore(BaseModelCore):
    """fastdm/model/flux.py:180-494."""

    def __init__(self, patch_size: int = 1, in_channels: int = 64, out_channels: int = 64, num_layers: int = 19,
                 num_single_layers: int = 38, attention_head_dim: int = 128, num_attention_heads: int = 24,
                 joint_attention_dim: int = 4096, pooled_projection_dim: int = 768, guidance_embeds: bool = True,
                 axes_dims_rope: Tuple[int] = (16, 56, 56), data_type=BF16, quant_dtype: Optional[torch.dtype] = torch.float8_e4m3fn,
                 cache=None, oom_ressolve: bool = False):
        super().__init__(type="DiT")
        self.need_resolve_oom = oom_ressolve      # accepted for signature parity; 288 GB HBM makes it moot
        self.out_channels = out_channels
        self.inner_dim = num_attention_heads * attention_head_dim
        self.guidance_embeds = guidance_embeds
        self.pos_embed = FluxPosEmbed(theta=10000, axes_dim=list(axes_dims_rope))
        cls = CombinedTimestepGuidanceTextProjEmbeddings if guidance_embeds else CombinedTimestepTextProjEmbeddings
        self.time_text_embed = cls(embedding_dim=self.inner_dim, pooled_projection_dim=pooled_projection_dim, data_type=data_type)
        self.context_embedder = QLinear(joint_attention_dim, self.inner_dim, data_type=data_type)
        self.x_embedder = QLinear(in_channels, self.inner_dim, data_type=data_type)
        self.quant_dtype = quant_dtype
        self.transformer_blocks = [FluxTransformerBlock(self.inner_dim, num_attention_heads, attention_head_dim, data_type=data_type)
                                   for _ in range(num_layers)]
        self.single_transformer_blocks = [FluxSingleTransformerBlock(self.inner_dim, num_attention_heads, attention_head_dim, data_type=data_type)
                                          for _ in range(num_single_layers)]
        self.norm_out = AdaLayerNormContinuous(self.inner_dim, self.inner_dim, elementwise_affine=False, eps=1e-6, data_type=data_type)
        self.proj_out = QLinear(self.inner_dim, patch_size * patch_size * self.out_channels, data_type=data_type)
        self.cache = cache
        self.cache_config = cache.config if cache is not None else None
        self.enable_caching = self.cache_config.enable_caching if cache is not None else False
        self.mod_table = None
        self._rope_key, self._rope_val = None, None

    # ------------------------------------------------------------------ loading (flux.py:274-332)
    def _pre_part_loading(self):
        t = self.time_text_embed
        self.init_weight(["time_text_embed.timestep_embedder.linear_1"], t.timestep_embedder.linear1)
        self.init_weight(["time_text_embed.timestep_embedder.linear_2"], t.timestep_embedder.linear2)
        if self.guidance_embeds:
            self.init_weight(["time_text_embed.guidance_embedder.linear_1"], t.guidance_embedder.linear1)
            self.init_weight(["time_text_embed.guidance_embedder.linear_2"], t.guidance_embedder.linear2)
        self.init_weight(["time_text_embed.text_embedder.linear_1"], t.text_embedder.linear1)
        self.init_weight(["time_text_embed.text_embedder.linear_2"], t.text_embedder.linear2)
        self.init_weight(["context_embedder"], self.context_embedder)
        self.init_weight(["x_embedder"], self.x_embedder)

    def _load_double_block(self, i, blk, src_i=None):
        p = f"transformer_blocks.{i if src_i is None else src_i}"
        q = self.quant_dtype
        self.init_weight([f"{p}.norm1.linear"], blk.norm1.linear)
        self.init_weight([f"{p}.norm1_context.linear"], blk.norm1_context.linear)
        blk.attn.norm_q_weight = self.init_weight([f"{p}.attn.norm_q.weight"])
        blk.attn.norm_k_weight = self.init_weight([f"{p}.attn.norm_k.weight"])
        self.init_weight([f"{p}.attn.to_q", f"{p}.attn.to_k", f"{p}.attn.to_v"], blk.attn.qkv, q)
        self.init_weight([f"{p}.attn.add_q_proj", f"{p}.attn.add_k_proj", f"{p}.attn.add_v_proj"], blk.attn.add_qkv_proj, q)
        self.init_weight([f"{p}.attn.to_out.0"], blk.attn.to_out, q)
        self.init_weight([f"{p}.attn.to_add_out"], blk.attn.to_add_out, q)
        blk.attn.norm_added_q_weight = self.init_weight([f"{p}.attn.norm_added_q.weight"])
        blk.attn.norm_added_k_weight = self.init_weight([f"{p}.attn.norm_added_k.weight"])
        self.init_weight([f"{p}.ff.net.0.proj"], blk.ff.act_fn.proj, q)
        self.init_weight([f"{p}.ff.net.2"], blk.ff.ff_out_proj, q)
        self.init_weight([f"{p}.ff_context.net.0.proj"], blk.ff_context.act_fn.proj, q)
        self.init_weight([f"{p}.ff_context.net.2"], blk.ff_context.ff_out_proj, q)

    def _load_single_block(self, i, blk, src_i=None):
        p = f"single_transformer_blocks.{i if src_i is None else src_i}"
        q = self.quant_dtype
        self.init_weight([f"{p}.norm.linear"], blk.norm.linear)
        self.init_weight([f"{p}.proj_mlp"], blk.proj_mlp, q)
        self.init_weight([f"{p}.proj_out"], blk.proj_out, q)
        blk.attn.norm_q_weight = self.init_weight([f"{p}.attn.norm_q.weight"])
        blk.attn.norm_k_weight = self.init_weight([f"{p}.attn.norm_k.weight"])
        self.init_weight([f"{p}.attn.to_q", f"{p}.attn.to_k", f"{p}.attn.to_v"], blk.attn.qkv, q)

    def _major_parts_loading(self):
        for i, blk in enumerate(self.transformer_blocks):
            self._load_double_block(i, blk)
        for i, blk in enumerate(self.single_transformer_blocks):
            self._load_single_block(i, blk)

    def _post_part_loading(self):
        self.init_weight(["norm_out.linear"], self.norm_out.linear)
        self.init_weight(["proj_out"], self.proj_out)

    def _finalize_loading(self):
        layers = []
        for blk in self.transformer_blocks:
            layers += [blk.norm1, blk.norm1_context]
        layers += [blk.norm for blk in self.single_transformer_blocks]
        layers.append(self.norm_out)
        self.mod_table = ModulationTable(layers, fold_one=True)     # every consumer above passes `scale_has_one` on

    # ------------------------------------------------------------------ forward (flux.py:334-494)
    def _rope_cache(self, txt_ids, img_ids, dtype):
        # tensors made under torch.inference_mode() -- how the reference's engine calls its pipelines, model_entry.py:787 -- carry no
        # version counter (reading `_version` raises): their table is rebuilt every forward (a few small launches) instead of cached
        # on an identity that cannot see in-place updates.  Found by tests/test_engine_checkpoint_gpu.py.
        if txt_ids.is_inference() or img_ids.is_inference():
            key = None
        else:
            key = (txt_ids.data_ptr(), img_ids.data_ptr(), tuple(txt_ids.shape), tuple(img_ids.shape), txt_ids._version, img_ids._version)
        if key is None or key != self._rope_key:
            ids = torch.cat((txt_ids, img_ids), dim=0)
            cos, sin = self.pos_embed.forward(ids)
            self._rope_val = torch.cat((cos[:, 0::2], sin[:, 1::2]), dim=-1).to(dtype).contiguous()
            self._rope_key = key
            self._rope_ids = (txt_ids, img_ids)   # keep the tensors alive so the pointers in the key stay unique
        return self._rope_val

    def run_blocks(self, hidden_states, encoder_hidden_states, temb, image_rotary_emb, joint_attention_kwargs=None,
                   controlnet_block_samples=None, controlnet_single_block_samples=None, start_block: int = 0,
                   controlnet_blocks_repeat: bool = False, **_):
        """Double blocks [start_block:], then all single blocks (start_block = 1: FBCache already ran block 0).  ControlNet
        residuals (tensors produced by a ControlNet the caller ran) are added after each block as flux.py:454-480 does."""
        import math
        nd, ns = len(self.transformer_blocks), len(self.single_transformer_blocks)
        for index_block in range(start_block, nd):
            encoder_hidden_states, hidden_states = self.transformer_blocks[index_block].forward(
                hidden_states=hidden_states, encoder_hidden_states=encoder_hidden_states, temb=temb, image_rotary_emb=image_rotary_emb,
                joint_attention_kwargs=joint_attention_kwargs)
            if controlnet_block_samples is not None:
                n = len(controlnet_block_samples)
                pick = index_block % n if controlnet_blocks_repeat else index_block // int(math.ceil(nd / n))
                hidden_states = hidden_states + controlnet_block_samples[pick]
        n_txt = encoder_hidden_states.shape[1]
        hidden_states = torch.cat([encoder_hidden_states, hidden_states], dim=1)
        for index_block, block in enumerate(self.single_transformer_blocks):
            hidden_states = block.forward(hidden_states=hidden_states, temb=temb, image_rotary_emb=image_rotary_emb,
                                          joint_attention_kwargs=joint_attention_kwargs)
            if controlnet_single_block_samples is not None:
                pick = index_block // int(math.ceil(ns / len(controlnet_single_block_samples)))
                hidden_states[:, n_txt:, ...] = hidden_states[:, n_txt:, ...] + controlnet_single_block_samples[pick]
        return hidden_states[:, n_txt:, ...]

    def forward(self, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor = None, pooled_projections: torch.Tensor = None,
                timestep: torch.Tensor = None, img_ids: torch.Tensor = None, txt_ids: torch.Tensor = None, guidance: torch.Tensor = None,
                joint_attention_kwargs: Optional[Dict[str, Any]] = None, controlnet_block_samples=None,
                controlnet_single_block_samples=None, return_dict: bool = True, controlnet_blocks_repeat: bool = False):
        dev = self.x_embedder.weight.device
        hidden_states = hidden_states.to(dev)
        encoder_hidden_states = encoder_hidden_states.to(dev)
        pooled_projections = pooled_projections.to(dev)
        hidden_states = self.x_embedder.forward(hidden_states)
        timestep = timestep.to(dev).to(hidden_states.dtype) * 1000          # bf16 multiply: 0.75 -> 752 (flux.py:399)
        if guidance is not None:
            guidance = guidance.to(dev).to(hidden_states.dtype) * 1000
        temb = (self.time_text_embed.forward(timestep, pooled_projections) if guidance is None
                else self.time_text_embed.forward(timestep, guidance, pooled_projections))
        encoder_hidden_states = self.context_embedder.forward(encoder_hidden_states)
        if txt_ids.ndim == 3:
            txt_ids = txt_ids[0]
        if img_ids.ndim == 3:
            img_ids = img_ids[0]
        image_rotary_emb = self._rope_cache(txt_ids, img_ids, hidden_states.dtype)

        if self.mod_table is not None:
            self.mod_table.begin_step(temb)
        try:
            if self.enable_caching:
                if controlnet_block_samples is not None or controlnet_single_block_samples is not None:
                    raise NotImplementedError("ControlNet residuals together with a step cache are not built")
                hidden_states = self.cache.apply_cache(model_type="flux", hidden_states=hidden_states,
                                                       encoder_hidden_states=encoder_hidden_states, temb=temb,
                                                       image_rotary_emb=image_rotary_emb, attention_kwargs=joint_attention_kwargs,
                                                       transformer_blocks=self.transformer_blocks,
                                                       single_transformer_blocks=self.single_transformer_blocks, core=self)
            else:
                hidden_states = self.run_blocks(hidden_states, encoder_hidden_states, temb, image_rotary_emb, joint_attention_kwargs,
                                                controlnet_block_samples, controlnet_single_block_samples,
                                                controlnet_blocks_repeat=controlnet_blocks_repeat)
            hidden_states = self.norm_out.forward(hidden_states.contiguous(), temb)
        finally:
            if self.mod_table is not None:
                self.mod_table.end_step()
        output = self.proj_out.forward(hidden_states)
        return (output,)
