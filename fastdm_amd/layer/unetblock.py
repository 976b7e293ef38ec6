"""SDXL UNet blocks (reference fastdm/layer/unetblock.py:11-516, IP-adapter / ComfyUI branches not built).

Same classes and constructor arguments as the reference; the transformer blocks run on the fused kernels of the DiT path:
affine LayerNorm + per-token quantisation in one pass, fused-qkv / fused-kv GEMMs that write q|k|v planes, the 64-wide
head attention kernel (self: S x S, cross: S x 77), residual adds in the GEMM epilogue, GEGLU as one elementwise kernel.
The prompt embeddings are quantised once per UNet forward, not once per cross-attention (70 of them in SDXL-base).
"""
import torch
import torch.nn.functional as F

from fastdm_amd import _native as N
from fastdm_amd.kernel import fused
from fastdm_amd.kernel.fused import Quantized
from fastdm_amd.kernel.operators_set import gelu_and_mul
from fastdm_amd.layer.qlinear import QLinear
from fastdm_amd.layer.resnetblock import ResnetBlock2D

BF16 = torch.bfloat16


def _quant_in(lin: QLinear, x: torch.Tensor):
    """The GEMM input of `lin`: per-token quantised for W8A8 layers, the bf16 matrix itself otherwise."""
    return fused.quantize_rows(x, None, lin.quant_type) if lin.quant_type is not None else x


class EncoderStates:
    """The prompt embeddings [B, n_txt, 2048] as the cross-attention kv projections consume them: flattened to rows and
    quantised once per quantisation type (per-token scales do not depend on the layer)."""

    def __init__(self, encoder_hidden_states: torch.Tensor):
        self.batch, self.tokens, self.dim = encoder_hidden_states.shape
        self.rows = encoder_hidden_states.reshape(self.batch * self.tokens, self.dim).contiguous()
        self._q = {}

    def gemm_input(self, lin: QLinear):
        if lin.quant_type is None:
            return self.rows
        if lin.quant_type not in self._q:
            self._q[lin.quant_type] = fused.quantize_rows(self.rows, None, lin.quant_type)
        return self._q[lin.quant_type]


class Attention_SDXL:
    """unetblock.py:11-142: attn1 (cross_attention_dim None) has one fused qkv projection, attn2 a q projection of the
    image tokens and a fused kv projection of the prompt embeddings; no biases except out_proj; 64-wide heads."""

    def __init__(self, inner_dim, cross_attention_dim=None, num_heads=None, has_ipadapter=False, ipadp_scale=0.6, data_type=BF16):
        if has_ipadapter:
            raise NotImplementedError("IP-adapter cross-attention is not built (DESIGN.md §8)")
        self.inner_dim = inner_dim
        if num_heads is None:
            self.head_dim = 64
            self.num_heads = inner_dim // self.head_dim
        else:
            self.num_heads, self.head_dim = num_heads, inner_dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.is_cross = cross_attention_dim is not None
        if not self.is_cross:
            self.qkv_proj = QLinear(inner_dim, inner_dim * 3, bias=False, data_type=data_type)
        else:
            self.q_proj = QLinear(inner_dim, inner_dim, bias=False, data_type=data_type)
            self.kv_proj = QLinear(cross_attention_dim, inner_dim * 2, bias=False, data_type=data_type)
        self.out_proj = QLinear(inner_dim, inner_dim, data_type=data_type)

    def attend(self, xq, rows: int, batch: int, enc: "EncoderStates" = None) -> torch.Tensor:
        """xq: GEMM input (normalised tokens, quantised or bf16) [rows, C].  Returns the attention output [rows, C]."""
        C = self.inner_dim
        dev = (xq.q if isinstance(xq, Quantized) else xq).device
        if not self.is_cross:
            planes = torch.empty((3, rows, C), device=dev, dtype=BF16)
            QLinear.forward_q_planes([(self.qkv_proj, xq, 0)], planes)
            q, k, v = planes[0], planes[1], planes[2]
        else:
            q = self.q_proj.forward_q(xq)
            kv = torch.empty((2, enc.batch * enc.tokens, C), device=dev, dtype=BF16)
            QLinear.forward_q_planes([(self.kv_proj, enc.gemm_input(self.kv_proj), 0)], kv)
            k, v = kv[0], kv[1]
        return fused.attention(q, k, v, batch, self.num_heads, self.head_dim, self.scale)

    def forward(self, hidden_states, encoder_hidden_states=None, batch=2, extra_options={}):
        """Reference signature (unetblock.py:42): hidden_states [B*S, C] -> out_proj(attention) [B*S, C]."""
        enc = None
        if self.is_cross:
            enc = encoder_hidden_states if isinstance(encoder_hidden_states, EncoderStates) else EncoderStates(
                encoder_hidden_states if encoder_hidden_states is not None else hidden_states.view(batch, -1, self.inner_dim))
        x = hidden_states.contiguous()
        o = self.attend(_quant_in(self.q_proj if self.is_cross else self.qkv_proj, x), x.shape[0], batch, enc)
        return self.out_proj.forward_q(_quant_in(self.out_proj, o))


class FeedForward_SDXL:
    """unetblock.py:213-224: GEGLU feed-forward, proj1 -> x[:, :4C] * gelu_erf(x[:, 4C:]) -> proj2."""

    def __init__(self, in_features, out_features, data_type=BF16):
        self.proj1 = QLinear(in_features, out_features * 8, data_type=data_type)
        self.proj2 = QLinear(out_features * 4, out_features, data_type=data_type)

    def forward(self, x):
        h = self.proj1.forward_q(_quant_in(self.proj1, x.contiguous()))
        if self.proj2.quant_type is None or h.shape[1] // 2 > 16384:
            return self.proj2.forward_q(_quant_in(self.proj2, gelu_and_mul(h)))
        return self.proj2.forward_q(fused.gelu_and_mul_quant(h, self.proj2.quant_type))


class BasicTransformerBlock:
    """unetblock.py:260-298."""

    def __init__(self, hidden_size, has_ipadp=False, ipadp_scale=0.6, data_type=BF16, cross_attention_dim=2048):
        self.hidden_size = hidden_size
        self.norm1_gamma = self.norm1_beta = self.norm2_gamma = self.norm2_beta = self.norm3_gamma = self.norm3_beta = None
        self.attn1 = Attention_SDXL(hidden_size, data_type=data_type, has_ipadapter=False)
        self.attn2 = Attention_SDXL(hidden_size, cross_attention_dim, data_type=data_type, has_ipadapter=has_ipadp, ipadp_scale=ipadp_scale)
        self.ff = FeedForward_SDXL(hidden_size, hidden_size, data_type=data_type)
        self._affine = None

    def finalize(self):
        """Affine LayerNorm as the fp32 modulation kernel's (scale, shift) = (gamma - 1, beta): one fp32 pass, one rounding,
        like F.layer_norm."""
        self._affine = [((g.float() - 1.0)[None].contiguous(), b.float()[None].contiguous())
                        for g, b in ((self.norm1_gamma, self.norm1_beta), (self.norm2_gamma, self.norm2_beta), (self.norm3_gamma, self.norm3_beta))]
        self._ones = torch.ones((1, self.hidden_size), device=self.norm1_gamma.device, dtype=BF16)

    def _norm_in(self, x, idx: int, lin: QLinear):
        sc, sh = self._affine[idx]
        out, qz = fused.ln_modulate_f32(x, sc, sh, x.shape[0], quant=lin.quant_type, eps=1e-5)
        return qz if lin.quant_type is not None else out

    def _proj_add(self, lin: QLinear, h: torch.Tensor, res: torch.Tensor) -> torch.Tensor:
        """res + lin(h): the add rides in the GEMM epilogue with a unit gate (same roundings as `x = lin(h); x + res`)."""
        return lin.forward_q(_quant_in(lin, h), N.EPI_GATE_RESIDUAL, gate=self._ones, res=res, rows_per_batch=res.shape[0])

    def forward(self, x, encoder_hidden_states=None, batch_size=2, transformer_options={}):
        rows = x.shape[0]
        enc = encoder_hidden_states
        if enc is not None and not isinstance(enc, EncoderStates):
            enc = EncoderStates(enc)
        o = self.attn1.attend(self._norm_in(x, 0, self.attn1.qkv_proj), rows, batch_size)
        x = self._proj_add(self.attn1.out_proj, o, x)
        if enc is None:  # unetblock.py:287: attn2 falls back to self-attention through its own kv projection
            raise NotImplementedError("attn2 without encoder_hidden_states needs cross_attention_dim == hidden_size; not an SDXL case")
        o = self.attn2.attend(self._norm_in(x, 1, self.attn2.q_proj), rows, batch_size, enc)
        x = self._proj_add(self.attn2.out_proj, o, x)
        h = self.ff.proj1.forward_q(self._norm_in(x, 2, self.ff.proj1))
        lin = self.ff.proj2
        if lin.quant_type is None or h.shape[1] // 2 > 16384:
            return self._proj_add(lin, gelu_and_mul(h), x)
        # GEGLU and the per-token quantisation of proj2's input in one pass (same bits as gelu_and_mul -> quantize)
        return lin.forward_q(fused.gelu_and_mul_quant(h, lin.quant_type), N.EPI_GATE_RESIDUAL, gate=self._ones, res=x, rows_per_batch=x.shape[0])


class Transformer2DModel:
    """unetblock.py:332-367."""

    def __init__(self, in_channels, out_channels, n_layers, has_ipadp=False, ipadp_scale=0.6, data_type=BF16):
        self.norm_gamma = self.norm_beta = None
        self.proj_in = QLinear(in_channels, out_channels, data_type=data_type)
        self.transformer_blocks = [BasicTransformerBlock(out_channels, data_type=data_type, has_ipadp=has_ipadp, ipadp_scale=ipadp_scale)
                                   for _ in range(n_layers)]
        self.proj_out = QLinear(out_channels, out_channels, data_type=data_type)

    def forward(self, hidden_states, encoder_hidden_states=None, transformer_options={}):
        batch, inner_dim, height, width = hidden_states.shape
        res = hidden_states
        t = fused.group_norm(hidden_states, 32, self.norm_gamma, self.norm_beta, 1e-06)
        t = t.permute(0, 2, 3, 1).reshape(batch * height * width, inner_dim)      # a view when the tensor is channels-last
        t = self.proj_in.forward_q(_quant_in(self.proj_in, t.contiguous()))
        enc = encoder_hidden_states
        if enc is not None and not isinstance(enc, EncoderStates):
            enc = EncoderStates(enc)
        for block in self.transformer_blocks:
            t = block.forward(t, enc, batch, transformer_options)
        t = self.proj_out.forward_q(_quant_in(self.proj_out, t))
        # (written into a tensor laid out like `res`: the bare sum takes the permuted operand's strides, which sends the next ResNet's
        # GroupNorm down torch's strided path instead of the fused NCHW kernel)
        return torch.add(t.reshape(batch, height, width, inner_dim).permute(0, 3, 1, 2), res, out=torch.empty_like(res))


class DownBlock2D:
    """unetblock.py:383-399."""

    def __init__(self, in_channels, out_channels, data_type=BF16):
        self.resnets = [ResnetBlock2D(in_channels, out_channels, conv_shortcut=False, data_type=data_type),
                        ResnetBlock2D(out_channels, out_channels, conv_shortcut=False, data_type=data_type)]
        self.downsample_conv_weight = self.downsample_conv_bias = None

    def forward(self, hidden_states, temb):
        output_states = []
        for module in self.resnets:
            hidden_states = module.forward(hidden_states, temb)
            output_states.append(hidden_states)
        hidden_states = fused.conv2d(hidden_states, self.downsample_conv_weight, self.downsample_conv_bias, 2, 1)
        output_states.append(hidden_states)
        return hidden_states, output_states


class CrossAttnDownBlock2D:
    """unetblock.py:401-441."""

    def __init__(self, in_channels, out_channels, n_layers, has_downsamplers=True, has_ipadpt=False, ipadp_scale=0.6, data_type=BF16):
        self.has_cross_attention = True
        self.attentions = [Transformer2DModel(out_channels, out_channels, n_layers, has_ipadp=has_ipadpt, ipadp_scale=ipadp_scale, data_type=data_type)
                           for _ in range(2)]
        self.resnets = [ResnetBlock2D(in_channels, out_channels, data_type=data_type),
                        ResnetBlock2D(out_channels, out_channels, conv_shortcut=False, data_type=data_type)]
        self.has_downsamplers = has_downsamplers
        self.downsample_conv_weight = self.downsample_conv_bias = None

    def forward(self, hidden_states, temb, encoder_hidden_states, transformer_options={}):
        output_states = []
        for resnet, attn in zip(self.resnets, self.attentions):
            hidden_states = resnet.forward(hidden_states, temb)
            hidden_states = attn.forward(hidden_states, encoder_hidden_states=encoder_hidden_states, transformer_options=transformer_options)
            output_states.append(hidden_states)
        if self.has_downsamplers:
            hidden_states = fused.conv2d(hidden_states, self.downsample_conv_weight, self.downsample_conv_bias, 2, 1)
            output_states.append(hidden_states)
        return hidden_states, output_states


class CrossAttnUpBlock2D:
    """unetblock.py:443-476."""

    def __init__(self, in_channels, out_channels, prev_output_channel, n_layers, has_ipadpt=False, ipadp_scale=0.6, data_type=BF16):
        self.has_cross_attention = True
        self.attentions = [Transformer2DModel(out_channels, out_channels, n_layers, has_ipadp=has_ipadpt, ipadp_scale=ipadp_scale, data_type=data_type)
                           for _ in range(3)]
        self.resnets = [ResnetBlock2D(prev_output_channel + out_channels, out_channels, data_type=data_type),
                        ResnetBlock2D(2 * out_channels, out_channels, data_type=data_type),
                        ResnetBlock2D(out_channels + in_channels, out_channels, data_type=data_type)]
        self.upsample_conv_weight = self.upsample_conv_bias = None

    def forward(self, hidden_states, res_hidden_states_tuple, temb, encoder_hidden_states, transformer_options={}):
        for resnet, attn in zip(self.resnets, self.attentions):
            res_hidden_states = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            hidden_states = torch.cat([hidden_states, res_hidden_states], dim=1)
            hidden_states = resnet.forward(hidden_states, temb)
            hidden_states = attn.forward(hidden_states, encoder_hidden_states=encoder_hidden_states, transformer_options=transformer_options)
        # nearest 2x upsampling + 3 x 3 convolution: one HIP launch that reads the low-resolution tensor through the upsampling (channels-last)
        return fused.conv2d(hidden_states, self.upsample_conv_weight, self.upsample_conv_bias, 1, 1, upsample=True)


class UpBlock2D:
    """unetblock.py:478-494."""

    def __init__(self, in_channels, out_channels, prev_output_channel, data_type=BF16):
        self.resnets = [ResnetBlock2D(out_channels + prev_output_channel, out_channels, data_type=data_type),
                        ResnetBlock2D(out_channels * 2, out_channels, data_type=data_type),
                        ResnetBlock2D(out_channels + in_channels, out_channels, data_type=data_type)]

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None):
        for resnet in self.resnets:
            res_hidden_states = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            hidden_states = torch.cat([hidden_states, res_hidden_states], dim=1)
            hidden_states = resnet.forward(hidden_states, temb)
        return hidden_states


class UNetMidBlock2DCrossAttn:
    """unetblock.py:496-516."""

    def __init__(self, in_features, has_ipadpt=False, ipadp_scale=0.6, data_type=BF16, n_layers=10):
        self.has_cross_attention = True
        self.attentions = [Transformer2DModel(in_features, in_features, n_layers=n_layers, has_ipadp=has_ipadpt, ipadp_scale=ipadp_scale,
                                              data_type=data_type)]
        self.resnets = [ResnetBlock2D(in_features, in_features, conv_shortcut=False, data_type=data_type),
                        ResnetBlock2D(in_features, in_features, conv_shortcut=False, data_type=data_type)]

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, transformer_options={}):
        hidden_states = self.resnets[0].forward(hidden_states, temb)
        for attn, resnet in zip(self.attentions, self.resnets[1:]):
            hidden_states = attn.forward(hidden_states, encoder_hidden_states=encoder_hidden_states, transformer_options=transformer_options)
            hidden_states = resnet.forward(hidden_states, temb)
        return hidden_states
