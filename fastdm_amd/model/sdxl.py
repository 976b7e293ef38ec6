"""SDXL UNet core (reference fastdm/model/sdxl.py:17-470; the IP-adapter branch is not built).

Same constructor surface, diffusers key names and forward signature as the reference.  The reference hard-codes
SDXL-base's widths inside the class (sdxl.py:53-113); here they are constructor arguments that default to SDXL-base, so
the narrow configuration of the parity fixtures goes through the same code.  Transformer blocks run on the fused HIP
kernels (layer/unetblock.py).  Round 6: with `channels_last=True` (the DEFAULT now) every convolution is a HIP kernel too -- 3 x 3 as an
implicit GEMM with fp32 accumulation (kernel/fused.py:conv2d -> fdm_conv2d_3x3_nhwc_bf16; stride-2 downsampling, the nearest upsampling folded
into the addressing, the ResNet skip added in the epilogue), 1 x 1 shortcuts as GEMMs on the pixel matrix -- and the GroupNorm / SiLU chains use
the NHWC form of the fused kernel: 30.3 -> 27.3 ms per 1024 x 1024 graph step, and BASELINE configs[0] moves from 1.14-1.47 x the
two-implementation envelope ONTO it (1.02 1.01 1.03 1.01; MIOpen's bf16 convolutions were the whole excess: profiles/r06g / r06q_sdxl_isolate.txt).
`channels_last=False` keeps NCHW activations and the library calls the reference makes (MIOpen as cuDNN there).  `channels_last=True` keeps the convolution activations in NHWC (the token <-> image
permutes become views); measured on MI355X at 1024x1024 the NCHW MIOpen kernels win by 1.5 ms of a 42 ms step, so NCHW is
the default.  (MIOpen's NCHW solvers answer the first call of a shape differently from later calls -- rounding-level, amplified to
~0.06 by the UNet; channels-last was bit-stable run to run in the same experiment, tools/diag_sdxl.py.)
"""
from typing import Any, Dict, Tuple

import torch
import torch.nn.functional as F

from fastdm_amd.kernel import fused
from fastdm_amd.layer.embeddings import Timesteps, TimestepEmbedding
from fastdm_amd.layer.unetblock import (CrossAttnDownBlock2D, CrossAttnUpBlock2D, DownBlock2D, EncoderStates, UNetMidBlock2DCrossAttn,
                                        UpBlock2D)
from fastdm_amd.model.basemodel import BaseModelCore

BF16 = torch.bfloat16


class _Config:
    pass


class SDXLUNetModelCore(BaseModelCore):
    def __init__(self, sample_size=128, in_channels=4, addition_time_embed_dim=256, time_cond_proj_dim=None, is_ip_adapter=False,
                 ip_adapter_scale=0.6, diffu_ipadapter_encoder_hid_proj=None, data_type=BF16, quant_dtype: torch.dtype = torch.float8_e4m3fn,
                 block_out_channels: Tuple[int, int, int] = (320, 640, 1280), transformer_layers: Tuple[int, int] = (2, 10),
                 mid_layers: int = 10, time_proj_dim: int = 320, time_embed_dim: int = 1280, text_embed_dim: int = 1280,
                 channels_last: bool = True):
        super().__init__(type="Unet")
        if is_ip_adapter:
            raise NotImplementedError("the IP-adapter branch of the SDXL UNet is not built (DESIGN.md §8)")
        cfg = _Config()
        (cfg.in_channels, cfg.addition_time_embed_dim, cfg.sample_size, cfg.time_cond_proj_dim, cfg.is_ip_adapter, cfg.ip_adapter_scale) = (
            in_channels, addition_time_embed_dim, sample_size, time_cond_proj_dim, is_ip_adapter, ip_adapter_scale)
        self.config = cfg
        self.quant_dtype = quant_dtype
        self.channels_last = channels_last
        c0, c1, c2 = block_out_channels
        l1, l2 = transformer_layers
        self.block_out_channels = tuple(block_out_channels)
        self.conv_in_weight = self.conv_in_bias = None
        self.time_proj = Timesteps(time_proj_dim, True, 0)
        self.time_embedding = TimestepEmbedding(in_features=time_proj_dim, out_features=time_embed_dim, data_type=data_type)
        self.add_time_proj = Timesteps(addition_time_embed_dim, True, 0)
        self.add_embedding = TimestepEmbedding(in_features=text_embed_dim + 6 * addition_time_embed_dim, out_features=time_embed_dim,
                                               data_type=data_type)
        self.encoder_hid_proj = None
        self.down_blocks = [DownBlock2D(in_channels=c0, out_channels=c0, data_type=data_type),
                            CrossAttnDownBlock2D(in_channels=c0, out_channels=c1, n_layers=l1, data_type=data_type),
                            CrossAttnDownBlock2D(in_channels=c1, out_channels=c2, n_layers=l2, has_downsamplers=False, data_type=data_type)]
        self.up_blocks = [CrossAttnUpBlock2D(in_channels=c1, out_channels=c2, prev_output_channel=c2, n_layers=l2, data_type=data_type),
                          CrossAttnUpBlock2D(in_channels=c0, out_channels=c1, prev_output_channel=c2, n_layers=l1, data_type=data_type),
                          UpBlock2D(in_channels=c0, out_channels=c0, prev_output_channel=c1, data_type=data_type)]
        self.mid_block = UNetMidBlock2DCrossAttn(c2, data_type=data_type, n_layers=mid_layers)
        self.conv_norm_out_gemma = self.conv_norm_out_beta = None
        self.conv_out_weight = self.conv_out_bias = None

    # ------------------------------------------------------------------ loading (sdxl.py:178-373)
    def _conv(self, name):
        w = self.init_weight([name])
        return w.contiguous(memory_format=torch.channels_last) if (self.channels_last and w.dim() == 4) else w

    def _load_resnet(self, r, n):
        r.norm1_gamma, r.norm1_beta = self.init_weight([f"{n}.norm1.weight"]), self.init_weight([f"{n}.norm1.bias"])
        r.conv1_weight, r.conv1_bias = self._conv(f"{n}.conv1.weight"), self.init_weight([f"{n}.conv1.bias"])
        self.init_weight([f"{n}.time_emb_proj"], r.time_emb_proj)                     # never quantised (sdxl.py:196)
        r.norm2_gamma, r.norm2_beta = self.init_weight([f"{n}.norm2.weight"]), self.init_weight([f"{n}.norm2.bias"])
        r.conv2_weight, r.conv2_bias = self._conv(f"{n}.conv2.weight"), self.init_weight([f"{n}.conv2.bias"])
        if r.conv_shortcut:
            r.convshortcut_weight, r.convshortcut_bias = self._conv(f"{n}.conv_shortcut.weight"), self.init_weight([f"{n}.conv_shortcut.bias"])

    def _load_transformer(self, t, n):
        q = self.quant_dtype
        t.norm_gamma, t.norm_beta = self.init_weight([f"{n}.norm.weight"]), self.init_weight([f"{n}.norm.bias"])
        self.init_weight([f"{n}.proj_in"], t.proj_in, q)
        self.init_weight([f"{n}.proj_out"], t.proj_out, q)
        for j, b in enumerate(t.transformer_blocks):
            k = f"{n}.transformer_blocks.{j}"
            for i in (1, 2, 3):
                setattr(b, f"norm{i}_gamma", self.init_weight([f"{k}.norm{i}.weight"]))
                setattr(b, f"norm{i}_beta", self.init_weight([f"{k}.norm{i}.bias"]))
            self.init_weight([f"{k}.attn1.to_q", f"{k}.attn1.to_k", f"{k}.attn1.to_v"], b.attn1.qkv_proj, q)
            self.init_weight([f"{k}.attn1.to_out.0"], b.attn1.out_proj, q)
            self.init_weight([f"{k}.attn2.to_q"], b.attn2.q_proj, q)
            self.init_weight([f"{k}.attn2.to_k", f"{k}.attn2.to_v"], b.attn2.kv_proj, q)
            self.init_weight([f"{k}.attn2.to_out.0"], b.attn2.out_proj, q)
            self.init_weight([f"{k}.ff.net.0.proj"], b.ff.proj1, q)
            self.init_weight([f"{k}.ff.net.2"], b.ff.proj2, q)
            b.finalize()

    def _pre_part_loading(self):
        self.conv_in_weight, self.conv_in_bias = self._conv("conv_in.weight"), self.init_weight(["conv_in.bias"])
        for emb, n in ((self.time_embedding, "time_embedding"), (self.add_embedding, "add_embedding")):
            self.init_weight([f"{n}.linear_1"], emb.linear1, self.quant_dtype)         # quantised in the reference too (sdxl.py:181-184)
            self.init_weight([f"{n}.linear_2"], emb.linear2, self.quant_dtype)

    def _major_parts_loading(self):
        for m, blk in enumerate(self.down_blocks):
            for i, r in enumerate(blk.resnets):
                self._load_resnet(r, f"down_blocks.{m}.resnets.{i}")
            for i, t in enumerate(getattr(blk, "attentions", [])):
                self._load_transformer(t, f"down_blocks.{m}.attentions.{i}")
            if m < 2:
                blk.downsample_conv_weight = self._conv(f"down_blocks.{m}.downsamplers.0.conv.weight")
                blk.downsample_conv_bias = self.init_weight([f"down_blocks.{m}.downsamplers.0.conv.bias"])
        for i, r in enumerate(self.mid_block.resnets):
            self._load_resnet(r, f"mid_block.resnets.{i}")
        self._load_transformer(self.mid_block.attentions[0], "mid_block.attentions.0")
        for m, blk in enumerate(self.up_blocks):
            for i, r in enumerate(blk.resnets):
                self._load_resnet(r, f"up_blocks.{m}.resnets.{i}")
            for i, t in enumerate(getattr(blk, "attentions", [])):
                self._load_transformer(t, f"up_blocks.{m}.attentions.{i}")
            if m < 2:
                blk.upsample_conv_weight = self._conv(f"up_blocks.{m}.upsamplers.0.conv.weight")
                blk.upsample_conv_bias = self.init_weight([f"up_blocks.{m}.upsamplers.0.conv.bias"])

    def _post_part_loading(self):
        self.conv_norm_out_gemma, self.conv_norm_out_beta = self.init_weight(["conv_norm_out.weight"]), self.init_weight(["conv_norm_out.bias"])
        self.conv_out_weight, self.conv_out_bias = self._conv("conv_out.weight"), self.init_weight(["conv_out.bias"])

    # ------------------------------------------------------------------ forward (sdxl.py:375-470)
    def forward(self, sample, timesteps, encoder_hidden_states, added_cond_kwargs: Dict[str, Any], **kwargs):
        mid_res, down_res = kwargs.get("mid_block_additional_residual"), kwargs.get("down_block_additional_residuals")
        is_controlnet = mid_res is not None and down_res is not None            # sdxl.py:405
        timesteps = timesteps.expand(sample.shape[0])
        t_emb = self.time_proj.forward(timesteps).to(dtype=sample.dtype)
        emb = self.time_embedding.forward(t_emb)
        text_embeds, time_ids = added_cond_kwargs.get("text_embeds"), added_cond_kwargs.get("time_ids")
        time_embeds = self.add_time_proj.forward(time_ids.flatten()).reshape((text_embeds.shape[0], -1))
        add_embeds = torch.concat([text_embeds, time_embeds], dim=-1).to(emb.dtype)
        emb = emb + self.add_embedding.forward(add_embeds)

        if self.channels_last:
            sample = sample.contiguous(memory_format=torch.channels_last)
        enc = EncoderStates(encoder_hidden_states)        # quantised once for all 70 cross-attentions
        sample = fused.conv2d(sample, self.conv_in_weight, self.conv_in_bias, 1, 1)
        s0 = sample
        sample, (s1, s2, s3) = self.down_blocks[0].forward(sample, temb=emb)
        sample, (s4, s5, s6) = self.down_blocks[1].forward(sample, temb=emb, encoder_hidden_states=enc)
        sample, (s7, s8) = self.down_blocks[2].forward(sample, temb=emb, encoder_hidden_states=enc)
        if is_controlnet:               # residuals of a ControlNet the caller ran, added to the nine skip tensors (sdxl.py:427-432)
            s0, s1, s2, s3, s4, s5, s6, s7, s8 = (s + r for s, r in zip((s0, s1, s2, s3, s4, s5, s6, s7, s8), down_res))
            sample = s8     # upstream adds in place and `sample` IS s8 (the last down block has no downsampler): the mid block sees it
        sample = self.mid_block.forward(sample, emb, encoder_hidden_states=enc)
        if is_controlnet:
            sample = sample + mid_res
        sample = self.up_blocks[0].forward(hidden_states=sample, temb=emb, res_hidden_states_tuple=[s6, s7, s8], encoder_hidden_states=enc)
        sample = self.up_blocks[1].forward(hidden_states=sample, temb=emb, res_hidden_states_tuple=[s3, s4, s5], encoder_hidden_states=enc)
        sample = self.up_blocks[2].forward(hidden_states=sample, temb=emb, res_hidden_states_tuple=[s0, s1, s2])
        sample = fused.group_norm(sample, 32, self.conv_norm_out_gemma, self.conv_norm_out_beta, 1e-05, silu=True)
        sample = fused.conv2d(sample, self.conv_out_weight, self.conv_out_bias, 1, 1)
        return [sample.contiguous()]
