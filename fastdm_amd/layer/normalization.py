"""AdaLN layers of the FLUX path (mirror of fastdm/layer/normalization.py:90-128,162-236).

`forward` keeps the reference's signature/returns.  `modulation()` exposes the [B, n*D] projection
so the block wiring can feed the fused LN+modulate(+quant) kernel; when the model core has batched
every AdaLN projection of the step into one GEMM (`ModulationTable`), it returns that slice.
"""
from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from fastdm_amd import _native as N
from fastdm_amd.kernel import fused
from fastdm_amd.kernel.timing import span
from fastdm_amd.layer.qlinear import QLinear

BF16 = torch.bfloat16


class RMSNorm:
    """normalization.py:16-43 with elementwise_affine=True: the rms_norm operator over the last dimension."""

    def __init__(self, dim, eps: float, elementwise_affine: bool = True):
        assert elementwise_affine
        self.dim, self.eps, self.weight = dim, eps, None

    def forward(self, hidden_states):
        from fastdm_amd.kernel.operators_set import rms_norm
        return rms_norm(hidden_states.contiguous(), self.weight, self.eps)


class FP32LayerNorm:
    """normalization.py:130-160: LayerNorm evaluated in fp32 whose result STAYS fp32 in the reference.  Here the
    consumers fuse what follows (modulation or a plain cast) into one kernel: `forward_bf16` = the affine form cast to
    the input dtype (`norm2(x).type_as(x)`, wan.py:105), `fused.ln_modulate_f32` for the modulated forms."""

    def __init__(self, normalized_shape, eps: float = 1e-5, elementwise_affine: bool = True, bias: bool = True):
        self.dim = normalized_shape if isinstance(normalized_shape, int) else normalized_shape[0]
        self.eps, self.elementwise_affine = eps, elementwise_affine
        self.weight = self.bias = None      # fp32 [dim], set by the loader when elementwise_affine
        self._sm1 = None

    def forward_bf16(self, x: torch.Tensor, quant=None):
        """bf16(LN_f32(x) * weight + bias) on a [rows, dim] matrix (optionally also quantised per row)."""
        if self._sm1 is None:
            w = self.weight if self.weight is not None else torch.ones(self.dim, device=x.device)
            b = self.bias if self.bias is not None else torch.zeros(self.dim, device=x.device)
            self._sm1 = ((w.float() - 1.0).reshape(1, -1).contiguous(), b.float().reshape(1, -1).contiguous())
        return fused.ln_modulate_f32(x, self._sm1[0], self._sm1[1], x.shape[0], quant=quant, want_bf16=quant is None, eps=self.eps)


class _AdaLNBase:
    chunks = 0
    scale_chunks = ()         # which of the `chunks` column blocks of the projection are `scale` (multiplied as 1 + scale)

    def __init__(self, embedding_dim: int, cond_dim: Optional[int] = None, bias=True, eps=1e-6, data_type=BF16):
        self.dim = embedding_dim
        self.linear = QLinear(cond_dim or embedding_dim, self.chunks * embedding_dim, bias=bias, data_type=data_type)
        self.norm_eps = eps
        self._cached_mod = None     # set by ModulationTable for the current step
        self.scale_has_one = False  # ... together with this: the cached scale chunks already hold bf16(1 + scale)

    def modulation(self, emb: torch.Tensor) -> torch.Tensor:
        """[B, chunks*D] = Linear(silu(emb)) (normalization.py:195,231; never quantised for FLUX)."""
        if self._cached_mod is not None:
            return self._cached_mod
        return self.linear.forward(F.silu(emb))

    def _plain_scale(self, folded: torch.Tensor, emb: torch.Tensor, idx: int) -> torch.Tensor:
        """The `scale` chunk a reference-signature forward() RETURNS (the caller applies x * (1 + scale) itself, flux.py / sd35.py).  Inside a
        ModulationTable(fold_one=True) step the cached chunk holds bf16(1 + scale): handing that out would add the 1 twice, and
        subtracting it again is not the same bits -- so the layer's own projection is evaluated for this one chunk (an M = 1 GEMV; the only
        in-step caller is the TeaCache probe of block 0)."""
        if not self.scale_has_one:
            return folded
        return self.linear.forward(F.silu(emb)).chunk(self.chunks, dim=1)[idx]

    def _apply(self, x, scale, shift):
        B, S, D = x.shape
        y, _, _ = fused.ln_modulate(x.reshape(B * S, D), scale, shift, S, quant=None, eps=self.norm_eps, scale_has_one=self.scale_has_one)
        return y.view(B, S, D)


class AdaLayerNormZero(_AdaLNBase):
    """normalization.py:162-199."""
    chunks = 6
    scale_chunks = (1, 4)

    def forward(self, x, emb=None):
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = self.modulation(emb).chunk(6, dim=1)
        return self._apply(x, scale_msa, shift_msa), gate_msa, shift_mlp, self._plain_scale(scale_mlp, emb, 4), gate_mlp


class SD35AdaLayerNormZeroX(_AdaLNBase):
    """normalization.py:45-88: nine chunks (the six of AdaLayerNormZero + shift/scale/gate of SD3.5's second attention);
    LayerNorm eps 1e-5 (the reference's default for this class)."""
    chunks = 9
    scale_chunks = (1, 4, 7)

    def __init__(self, embedding_dim: int, elementwise_affine=False, norm_type: str = "layer_norm", bias: bool = True, eps=1e-5,
                 data_type=BF16):
        assert not elementwise_affine and norm_type == "layer_norm"
        super().__init__(embedding_dim, bias=bias, eps=eps, data_type=data_type)

    def forward(self, hidden_states, emb=None):
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp, shift_msa2, scale_msa2, gate_msa2 = self.modulation(emb).chunk(9, dim=1)
        return (self._apply(hidden_states, scale_msa, shift_msa), gate_msa, shift_mlp, self._plain_scale(scale_mlp, emb, 4), gate_mlp,
                self._apply(hidden_states, scale_msa2, shift_msa2), gate_msa2)


class AdaLayerNormZeroSingle(_AdaLNBase):
    """normalization.py:201-236."""
    chunks = 3
    scale_chunks = (1,)

    def forward(self, x, emb=None):
        shift_msa, scale_msa, gate_msa = self.modulation(emb).chunk(3, dim=1)
        return self._apply(x, scale_msa, shift_msa), gate_msa


class AdaLayerNormContinuous(_AdaLNBase):
    """normalization.py:90-128: chunks are (scale, shift) -- scale first."""
    chunks = 2
    scale_chunks = (0,)

    def __init__(self, embedding_dim, conditioning_embedding_dim, elementwise_affine=False, eps=1e-5, bias=True, data_type=BF16):
        assert not elementwise_affine, "FLUX uses elementwise_affine=False (model/flux.py:259)"
        super().__init__(embedding_dim, conditioning_embedding_dim, bias=bias, eps=eps, data_type=data_type)

    def forward(self, x, conditioning_embedding):
        scale, shift = self.modulation(conditioning_embedding.to(x.dtype)).chunk(2, dim=1)
        return self._apply(x, scale, shift)


class ModulationTable:
    """All AdaLN projections of one denoise step as ONE [B,3072] x [3072, sum(n_i)] GEMM.

    Every AdaLN layer of FLUX consumes the same silu(temb) (model/flux.py:130-134,57; 6.46 GB of
    bf16 weights per step).  The weights are concatenated once at load time
    (each layer's QLinear keeps a view, so the layers still work on their own) and the step does a
    single weight-streaming launch instead of 77 M=1 GEMVs.
    """

    def __init__(self, layers, fold_one: bool = False):
        """fold_one: the table's GEMV also writes bf16(1 + scale) into the layers' `scale` chunks (fdm_linear_skinny_bf16's post_add)
        and marks the layers `scale_has_one` for the step, so fdm_ln_modulate multiplies by the stored value instead of adding and
        rounding once per element of every row.  Same bits: the reference forms `1 + scale` as a bf16 tensor (normalization.py:197,233).
        Only for consumers that pass `scale_has_one` on (the FLUX blocks); off, the table is the plain projection."""
        self.layers = list(layers)
        ws = [l.linear.weight.t() for l in self.layers]            # [n_i, K] each (K contiguous)
        K = ws[0].shape[1]
        total = sum(w.shape[0] for w in ws)
        dev = ws[0].device
        w_all = torch.empty((total, K), device=dev, dtype=ws[0].dtype)
        b_all = torch.empty((total,), device=dev, dtype=ws[0].dtype)
        self.post_add = torch.zeros((total,), device=dev, dtype=ws[0].dtype) if fold_one else None
        self.offsets = []
        off = 0
        for l, w in zip(self.layers, ws):
            n = w.shape[0]
            w_all[off:off + n].copy_(w)
            b_all[off:off + n].copy_(l.linear.bias)
            l.linear.weight = w_all[off:off + n].t()                # view into the table
            l.linear.bias = b_all[off:off + n]
            if fold_one:
                d = n // l.chunks
                for c in l.scale_chunks:
                    self.post_add[off + c * d:off + (c + 1) * d] = 1.0
            self.offsets.append((off, n))
            off += n
        self.table = QLinear(K, total, bias=True, data_type=ws[0].dtype)
        self.table.weight, self.table.bias = w_all.t(), b_all
        self.table.quant_type = None

    def begin_step(self, temb: torch.Tensor):
        a = F.silu(temb)
        folded = False
        if self.post_add is not None and a.dim() == 2 and a.shape[0] <= 8 and a.stride(1) == 1:
            wt = self.table.weight
            K, total = wt.shape
            mod = torch.empty((a.shape[0], total), device=a.device, dtype=a.dtype)
            with span("gemm_bf16", 2.0 * a.shape[0] * total * K):
                rc = N.lib().fdm_linear_skinny_bf16(N.ptr(mod), N.ptr(a), N.ptr(wt), N.ptr(self.table.bias), N.ptr(self.post_add),
                                                    a.shape[0], total, K, a.stride(0), wt.stride(1), mod.stride(0), N.stream())
            if rc == 0:
                folded = True
            elif rc != -2:                                          # FDM_ERR_UNSUPPORTED: not GEMV-shaped, take the tile kernel below
                N.check(rc, "modulation table")
        if not folded:
            mod = self.table.forward(a)                             # [B, total]
        for l, (off, n) in zip(self.layers, self.offsets):
            l._cached_mod = mod[:, off:off + n]
            l.scale_has_one = folded

    def end_step(self):
        for l in self.layers:
            l._cached_mod = None
            l.scale_has_one = False
