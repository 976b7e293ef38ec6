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
from fastdm_amd.kernel.fused import Quantized
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

    # The GELU between the two linears.  W8A8 (fp8 / int8): the first GEMM runs with the PLAIN epilogue and the activation rides in the
    # per-token quantisation pass of the second linear's input (fused.quantize_rows(act0=...): same roundings in the same order, bit-identical
    # to GELU epilogue + quantiser).  The GEMMs are MFMA-issue-bound with one wave per SIMD -- a tile's GELU drain (9-20 k cycles of VALU)
    # cannot hide under one K step -- while the quantiser is HBM-bound with idle VALU slots (round 6: FLUX 323 -> 265 us per FF1 launch).
    # bf16 layers (no quantiser in between) keep the GELU in the GEMM epilogue.
    def _act_split(self):
        """(epilogue of the first GEMM, activation of the quantiser in front of the second)"""
        return (N.EPI_NONE, self.epilogue) if self.ff_out_proj.quant_type is not None else (self.epilogue, 0)

    def forward_fused(self, xq, gate: torch.Tensor, res: torch.Tensor, rows_per_batch: int, gate_epilogue: int = N.EPI_GATE_RESIDUAL) -> torch.Tensor:
        """res + gate * W2(gelu(W1 x)); xq = quantised (or bf16) [M, D] input; returns the new residual stream.
        gate_epilogue = N.EPI_GATE_RESIDUAL_F32 for Wan's fp32 gates."""
        epi, act = self._act_split()
        h = self.act_fn.proj.forward_q(xq, epilogue=epi)
        hq = self.ff_out_proj.quantize_input(h, act) if self.ff_out_proj.quant_type is not None else h
        return self.ff_out_proj.forward_q(hq, epilogue=gate_epilogue, gate=gate, res=res, rows_per_batch=rows_per_batch, out=res)

    @staticmethod
    def forward_fused_pair(ff0, xq0, gate0, res0, rpb0, ff1, xq1, gate1, res1, rpb1):
        """The image- and text-stream FFNs of a double block side by side: each GEMM pair is one launch."""
        from fastdm_amd.layer.qlinear import QLinear as QL
        (epi0, act0), (epi1, act1) = ff0._act_split(), ff1._act_split()
        if epi0 != epi1:                                         # (one stream quantised, the other not: never in the shipped models)
            epi0, act0, epi1, act1 = ff0.epilogue, 0, ff1.epilogue, 0
            assert epi0 == epi1
        h0, h1 = QL.forward_q_pair(ff0.act_fn.proj, xq0, ff1.act_fn.proj, xq1, epilogue=epi0)
        q0 = ff0.ff_out_proj.quantize_input(h0, act0) if ff0.ff_out_proj.quant_type is not None else h0
        q1 = ff1.ff_out_proj.quantize_input(h1, act1) if ff1.ff_out_proj.quant_type is not None else h1
        return QL.forward_q_pair(ff0.ff_out_proj, q0, ff1.ff_out_proj, q1, epilogue=N.EPI_GATE_RESIDUAL, gate0=gate0, res0=res0,
                                 rpb0=rpb0, out0=res0, gate1=gate1, res1=res1, rpb1=rpb1, out1=res1)

    def forward(self, hidden_states: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        shp = hidden_states.shape
        x = hidden_states.reshape(-1, shp[-1])
        p = self.act_fn.proj
        xq = p.quantize_input(x) if p.quant_type is not None else x.contiguous()
        epi, act = self._act_split()
        h = p.forward_q(xq, epilogue=epi)
        hq = self.ff_out_proj.quantize_input(h, act) if self.ff_out_proj.quant_type is not None else h
        return self.ff_out_proj.forward_q(hq).view(*shp[:-1], -1)


class _ActProj:
    def __init__(self, dim_in, dim_out, bias, data_type):
        self.proj = QLinear(dim_in, dim_out, bias=bias, data_type=data_type)


def _rows(x, b: int, rows_per_batch: int, batch: int):
    """Batch element b's rows of a [B*rows, K] GEMM input (a bf16 matrix or a `Quantized` with its per-row scales)."""
    if batch == 1:
        return x
    lo, hi = b * rows_per_batch, (b + 1) * rows_per_batch
    if isinstance(x, Quantized):
        return Quantized(x.q[lo:hi], x.scale[lo:hi], None if x.azp is None else x.azp[lo:hi])
    return x[lo:hi]


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
    def attend(self, xq, batch: int, s_img: int, rope: Optional[torch.Tensor], enc_q=None, s_txt: int = 0,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Both fused-qkv GEMMs write straight into three joint planes q|k|v [B*(S_txt+S_img), inner] (text rows first,
        as the reference concatenates them, transformer.py:288-296) through the GEMM's plane-split output; q/k are
        normalised + rotated in place and the attention kernel reads the planes -- no chunk, no cat, no copy of v."""
        s_tot = s_img + s_txt
        inner = self.inner_dim
        dev = (xq.q if hasattr(xq, "q") else xq).device
        planes = torch.empty((3, batch * s_tot, inner), device=dev, dtype=BF16)
        items = []
        for b in range(batch):
            items.append((self.qkv, _rows(xq, b, s_img, batch), b * s_tot + s_txt))
            if enc_q is not None:
                items.append((self.add_qkv_proj, _rows(enc_q, b, s_txt, batch), b * s_tot))
        QLinear.forward_q_planes(items, planes)
        q, k, v = planes[0], planes[1], planes[2]
        fused.qk_norm_rope_inplace(q[s_txt:], k[s_txt:], self.norm_q_weight, self.norm_k_weight, rope, batch * s_img, s_img, s_tot, s_txt,
                                   self.heads, self.head_dim, self.eps)
        if enc_q is not None:
            fused.qk_norm_rope_inplace(q, k, self.norm_added_q_weight, self.norm_added_k_weight, rope, batch * s_txt, s_txt, s_tot, 0,
                                       self.heads, self.head_dim, self.eps)
        return fused.attention(q, k, v, batch, self.heads, self.head_dim, self.scale, out=out)

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


class WanAttention:
    """transformer.py:393-534.  q/k RMS-norm runs ACROSS heads (over the whole inner dim), RoPE is GPT-J style on the merged
    [cos | sin] table, self-attention may be block-sparse (radial).  With `added_kv_proj_dim` (Wan2.1-I2V) the cross-attention
    also attends the image-context tokens -- the rows in front of the 512 text tokens -- through add_k_proj / add_v_proj and
    adds that result."""
    TEXT_LEN = 512          # "512 is the context length of the text encoder, hardcoded for now" (transformer.py:473-474)

    def __init__(self, dim: int, heads: int = 8, dim_head: int = 64, eps: float = 1e-5, dropout: float = 0.0,
                 added_kv_proj_dim: Optional[int] = None, cross_attention_dim_head: Optional[int] = None, processor=None,
                 is_cross_attention=None, is_causal: bool = False, data_type=BF16):
        self.inner_dim = dim_head * heads
        self.heads, self.head_dim, self.eps = heads, dim_head, eps
        self.added_kv_proj_dim = added_kv_proj_dim
        self.cross_attention_dim_head = cross_attention_dim_head
        self.kv_inner_dim = self.inner_dim if cross_attention_dim_head is None else cross_attention_dim_head * heads
        self.is_causal = is_causal
        self.scale = dim_head ** -0.5
        self.sdpa_kv_heads, self.sdpa_head_dim = heads, dim_head
        self.norm_q_weight = self.norm_k_weight = None
        if cross_attention_dim_head is not None:
            self.to_q = QLinear(dim, self.inner_dim, bias=True, data_type=data_type)
            self.to_kv = QLinear(dim, 2 * self.kv_inner_dim, bias=True, data_type=data_type)
        else:
            self.qkv = QLinear(dim, self.inner_dim + 2 * self.kv_inner_dim, bias=True, data_type=data_type)
        self.to_out = QLinear(self.inner_dim, dim, bias=True, data_type=data_type)
        self.add_k_proj = self.add_v_proj = None
        self.norm_added_k_weight = None
        if added_kv_proj_dim is not None:
            self.add_k_proj = QLinear(added_kv_proj_dim, self.inner_dim, bias=True, data_type=data_type)
            self.add_v_proj = QLinear(added_kv_proj_dim, self.inner_dim, bias=True, data_type=data_type)
        self.is_cross_attention = cross_attention_dim_head is not None

    # ---- fused path: x is a [B*S, D] matrix (Quantized or bf16); returns the attention result [B*S, inner] (before to_out)
    def attend(self, xq, batch: int, seq: int, rope=None, enc_q=None, enc_len: int = 0, sparse_attn=None, layer_index: int = 0,
               img_q=None, img_len: int = 0):
        """enc_q: GEMM input of the TEXT rows [B*enc_len, D]; img_q: of the image-context rows [B*img_len, added_kv_proj_dim]."""
        D = self.inner_dim
        if self.is_cross_attention:
            q = fused.rms_norm_rows(self.to_q.forward_q(xq), self.norm_q_weight, self.eps)
            kv = self.to_kv.forward_q(enc_q)
            k = fused.rms_norm_rows(kv[:, :D], self.norm_k_weight, self.eps)
            v, sk = kv[:, D:], enc_len
            if img_q is not None:       # transformer.py:502-507,531-532
                k_img = fused.rms_norm_rows(self.add_k_proj.forward_q(img_q), self.norm_added_k_weight, self.eps)
                v_img = self.add_v_proj.forward_q(img_q)
                return (fused.attention(q, k, v, batch, self.heads, self.head_dim, self.scale)
                        + fused.attention(q, k_img, v_img, batch, self.heads, self.head_dim, self.scale))
        else:
            qkv = self.qkv.forward_q(xq)
            # across-head norm and (when given) the rotary embedding in one pass over each row: the bits of rms_norm -> rotary_pos_embedding
            q = fused.rms_norm_rows(qkv[:, :D], self.norm_q_weight, self.eps, rope=rope, seq_len=seq, head_dim=self.head_dim)
            k = fused.rms_norm_rows(qkv[:, D:2 * D], self.norm_k_weight, self.eps, rope=rope, seq_len=seq, head_dim=self.head_dim)
            v, sk = qkv[:, 2 * D:], seq
        use_sparse = False
        if sparse_attn is not None and not self.is_cross_attention:
            cb = sparse_attn.config.current_steps_callback
            step = cb() if cb is not None else None
            step = 0 if step is None else step
            use_sparse = not (step < sparse_attn.config.dense_steps or layer_index < sparse_attn.config.dense_layers)
        if use_sparse:
            outs = []
            for b in range(batch):       # the reference flattens (b s) and its kernel call is per sample as well
                sl = slice(b * seq, (b + 1) * seq)
                outs.append(sparse_attn.apply(query=q[sl].view(seq, self.heads, self.head_dim), key=k[sl].view(seq, self.heads, self.head_dim),
                                              value=v[sl].reshape(seq, self.heads, self.head_dim)).reshape(seq, D))
            return outs[0] if batch == 1 else torch.cat(outs, dim=0)
        return fused.attention(q, k, v, batch, self.heads, self.head_dim, self.scale)

    # ---- reference signature
    def forward(self, hidden_states: torch.Tensor, encoder_hidden_states: Optional[torch.Tensor] = None, attention_mask=None,
                rotary_emb=None, sparse_attn=None, **kwargs) -> torch.Tensor:
        B, S, Dm = hidden_states.shape
        x = hidden_states.reshape(B * S, Dm)
        first = self.to_q if self.is_cross_attention else self.qkv
        xq = first.quantize_input(x.contiguous()) if first.quant_type is not None else x.contiguous()
        rope = None
        if rotary_emb is not None:
            rope = rotary_emb if torch.is_tensor(rotary_emb) else \
                torch.cat((rotary_emb[0].squeeze()[:, 0::2], rotary_emb[1].squeeze()[:, 1::2]), dim=-1).to(hidden_states.dtype).contiguous()
        enc_q, enc_len, img_q, img_len = None, 0, None, 0
        if self.is_cross_attention:
            if self.add_k_proj is not None:
                img_len = encoder_hidden_states.shape[1] - self.TEXT_LEN
                ei = encoder_hidden_states[:, :img_len].reshape(B * img_len, -1).contiguous()
                img_q = self.add_k_proj.quantize_input(ei) if self.add_k_proj.quant_type is not None else ei
                encoder_hidden_states = encoder_hidden_states[:, img_len:]
            enc_len = encoder_hidden_states.shape[1]
            e = encoder_hidden_states.reshape(B * enc_len, -1).contiguous()
            enc_q = self.to_kv.quantize_input(e) if self.to_kv.quant_type is not None else e
        o = self.attend(xq, B, S, rope, enc_q, enc_len, sparse_attn, kwargs.get("layer_index", 0), img_q, img_len)
        oq = self.to_out.quantize_input(o) if self.to_out.quant_type is not None else o
        return self.to_out.forward_q(oq).view(B, S, -1)
