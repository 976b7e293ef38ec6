"""Attention / FeedForward of the FLUX path (mirror of fastdm/layer/transformer.py:14-62,64-317).

`Attention.forward` / `FeedForward.forward` keep the reference's tensor-in/tensor-out signature
([B,S,D] bf16).  `Attention.joint_qkv` / `FeedForward.forward_fused` are the fused entry points the
block wiring uses: they take per-row quantised activations straight from the LN+modulate kernel,
run the GEMMs with fused epilogues, and write q/k/v directly into the joint [text ; image]
buffers (no torch.cat, no .contiguous(), no separate rms_norm / rope launches).
"""
from typing import Optional

import torch

from fastdm_amd import _native as N
from fastdm_amd.kernel import fused
from fastdm_amd.layer.qlinear import QLinear

BF16 = torch.bfloat16


class FeedForward:
    """transformer.py:14-62 with activation_fn in {'gelu', 'gelu-approximate'} (the DiT uses)."""

    def __init__(self, dim: int, dim_out: Optional[int] = None, mult: int = 4, activation_fn: str = "gelu-approximate",
                 inner_dim=None, bias: bool = True, data_type=BF16, **_):
        inner_dim = int(dim * mult) if inner_dim is None else inner_dim
        if activation_fn not in ("gelu", "gelu-approximate"):
            raise ValueError(f"activation_fn {activation_fn} is not on the DiT path of this build")
        self.epilogue = N.EPI_GELU_TANH if activation_fn == "gelu-approximate" else N.EPI_GELU_ERF
        self.act_fn = _ActProj(dim, inner_dim, bias, data_type)            # keeps the reference attribute path act_fn.proj
        self.ff_out_proj = QLinear(inner_dim, dim_out or dim, bias=bias, data_type=data_type)

    def forward_fused(self, xq, gate: torch.Tensor, res: torch.Tensor, rows_per_batch: int) -> torch.Tensor:
        """res + gate * W2(gelu(W1 x)); xq = quantised (or bf16) [M, D] input; returns the new residual stream."""
        h = self.act_fn.proj.forward_q(xq, epilogue=self.epilogue)
        hq = self.ff_out_proj.quantize_input(h) if self.ff_out_proj.quant_type is not None else h
        return self.ff_out_proj.forward_q(hq, epilogue=N.EPI_GATE_RESIDUAL, gate=gate, res=res, rows_per_batch=rows_per_batch, out=res)

    @staticmethod
    def forward_fused_pair(ff0, xq0, gate0, res0, rpb0, ff1, xq1, gate1, res1, rpb1):
        """The image- and text-stream FFNs of a double block side by side: each GEMM pair is one launch."""
        from fastdm_amd.layer.qlinear import QLinear as QL
        h0, h1 = QL.forward_q_pair(ff0.act_fn.proj, xq0, ff1.act_fn.proj, xq1, epilogue=ff0.epilogue)
        q0 = ff0.ff_out_proj.quantize_input(h0) if ff0.ff_out_proj.quant_type is not None else h0
        q1 = ff1.ff_out_proj.quantize_input(h1) if ff1.ff_out_proj.quant_type is not None else h1
        return QL.forward_q_pair(ff0.ff_out_proj, q0, ff1.ff_out_proj, q1, epilogue=N.EPI_GATE_RESIDUAL, gate0=gate0, res0=res0,
                                 rpb0=rpb0, out0=res0, gate1=gate1, res1=res1, rpb1=rpb1, out1=res1)

    def forward(self, hidden_states: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        shp = hidden_states.shape
        x = hidden_states.reshape(-1, shp[-1])
        p = self.act_fn.proj
        xq = p.quantize_input(x) if p.quant_type is not None else x.contiguous()
        h = p.forward_q(xq, epilogue=self.epilogue)
        hq = self.ff_out_proj.quantize_input(h) if self.ff_out_proj.quant_type is not None else h
        return self.ff_out_proj.forward_q(hq).view(*shp[:-1], -1)


class _ActProj:
    def __init__(self, dim_in, dim_out, bias, data_type):
        self.proj = QLinear(dim_in, dim_out, bias=bias, data_type=data_type)


class Attention:
    """transformer.py:64-317 restricted to what FLUX instantiates (model/flux.py:39-50,92-104):
    qk_norm='rms_norm', bias=True, either joint (added_kv_proj_dim, context_pre_only=False) or
    single-stream (pre_only=True)."""

    def __init__(self, query_dim: int, heads: int = 8, dim_head: int = 64, bias: bool = True, qk_norm: Optional[str] = "rms_norm",
                 added_kv_proj_dim: Optional[int] = None, out_dim: int = None, context_pre_only=None, pre_only=False,
                 eps: float = 1e-6, data_type=BF16, **_):
        assert qk_norm == "rms_norm"
        self.inner_dim = out_dim if out_dim is not None else dim_head * heads
        self.heads = self.inner_dim // dim_head
        self.head_dim = dim_head
        self.eps = eps
        self.scale = dim_head ** -0.5
        self.pre_only = pre_only
        self.context_pre_only = context_pre_only
        self.added_kv_proj_dim = added_kv_proj_dim
        self.norm_q_weight = self.norm_k_weight = None
        self.norm_added_q_weight = self.norm_added_k_weight = None
        self.qkv = QLinear(query_dim, 3 * self.inner_dim, bias=bias, data_type=data_type)
        if added_kv_proj_dim is not None:
            self.add_qkv_proj = QLinear(added_kv_proj_dim, 3 * self.inner_dim, bias=True, data_type=data_type)
        if not pre_only:
            self.to_out = QLinear(self.inner_dim, query_dim, bias=True, data_type=data_type)
        if context_pre_only is not None and not context_pre_only:
            self.to_add_out = QLinear(self.inner_dim, query_dim, bias=True, data_type=data_type)

    # ---- fused path: inputs are [B*S, D] matrices (quantised or bf16), output = attention result [B*S_tot, D]
    def attend(self, xq, batch: int, s_img: int, rope: Optional[torch.Tensor], enc_q=None, s_txt: int = 0) -> torch.Tensor:
        s_tot = s_img + s_txt
        dev = (xq.q if hasattr(xq, "q") else xq).device
        q = torch.empty((batch * s_tot, self.inner_dim), device=dev, dtype=BF16)
        k = torch.empty_like(q)
        v = torch.empty_like(q)
        if enc_q is not None:
            qkv, eqkv = QLinear.forward_q_pair(self.qkv, xq, self.add_qkv_proj, enc_q)
        else:
            qkv = self.qkv.forward_q(xq)
        fused.qk_norm_rope(qkv, self.norm_q_weight, self.norm_k_weight, rope, q, k, v, s_img, s_tot, s_txt, self.heads, self.head_dim, self.eps)
        if enc_q is not None:
            fused.qk_norm_rope(eqkv, self.norm_added_q_weight, self.norm_added_k_weight, rope, q, k, v, s_txt, s_tot, 0, self.heads,
                               self.head_dim, self.eps)
        return fused.attention(q, k, v, batch, self.heads, self.head_dim, self.scale)

    # ---- reference signature
    def forward(self, hidden_states: torch.Tensor, encoder_hidden_states: Optional[torch.Tensor] = None, attention_mask=None,
                **cross_attention_kwargs):
        rope = cross_attention_kwargs.get("image_rotary_emb")
        B, s_img, D = hidden_states.shape
        x = hidden_states.reshape(B * s_img, D)
        xq = self.qkv.quantize_input(x) if self.qkv.quant_type is not None else x.contiguous()
        if encoder_hidden_states is None:
            o = self.attend(xq, B, s_img, rope).view(B, s_img, self.inner_dim)
            return o if self.pre_only else self.to_out.forward(o)
        s_txt = encoder_hidden_states.shape[1]
        e = encoder_hidden_states.reshape(B * s_txt, D)
        eq = self.add_qkv_proj.quantize_input(e) if self.add_qkv_proj.quant_type is not None else e.contiguous()
        o = self.attend(xq, B, s_img, rope, eq, s_txt).view(B, s_txt + s_img, self.inner_dim)
        enc_o, img_o = o[:, :s_txt], o[:, s_txt:]
        if self.context_pre_only is not None and not self.context_pre_only:
            enc_o = self.to_add_out.forward(enc_o)
        if not self.pre_only:
            img_o = self.to_out.forward(img_o)
        return img_o, enc_o

    def forward_qwen(self, hidden_states: torch.Tensor, encoder_hidden_states: Optional[torch.Tensor] = None, attention_mask=None,
                     **cross_attention_kwargs):
        """transformer.py:319-391: the same joint [text ; image] attention as `forward` (the text mask is accepted and
        unused there too); returns (img_attn_output, txt_attn_output)."""
        if encoder_hidden_states is None:
            raise ValueError("QwenAttn forward requires encoder_hidden_states (text stream)")
        cross_attention_kwargs.pop("encoder_hidden_states_mask", None)
        return self.forward(hidden_states, encoder_hidden_states, attention_mask, **cross_attention_kwargs)
