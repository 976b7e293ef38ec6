"""QLinear on the HIP library (mirror of fastdm/layer/qlinear.py:6-81).

Same constructor, `weight_loading_and_quant(src_weight, src_bias, quant_type)` contract (weights
arrive as [in, out] views, several are fused along N) and `forward`.  Differences by design:
weights are moved to the GPU first and quantised there by the same row-quant kernel that quantises
activations (identical arithmetic to fastdm/utils/quantization.py, verified bit-exact in the
tests), instead of on the CPU; and `forward_q` exposes the GEMM epilogues to the block wiring.
"""
from typing import Optional

import torch

from fastdm_amd import _native as N
from fastdm_amd.kernel import fused
from fastdm_amd.kernel.fused import Quantized
from fastdm_amd.kernel import timing
from fastdm_amd.kernel.timing import span

BF16 = torch.bfloat16
FP8 = torch.float8_e4m3fn


class QLinear:
    def __init__(self, in_features, out_features, bias=True, data_type=BF16, device_type="cuda"):
        self.in_features, self.out_features = in_features, out_features
        self.weight = None              # [K, N] view with stride(0) == 1 (the reference's column-major layout)
        self.bias = None
        self.device = device_type
        self.has_bias = bias
        self.dtype = data_type
        self.weight_quant_scale = None  # [N, 1] f32
        self.weight_asym_sumcol = None  # [1, N] i32 (int8 only)
        self.quant_type = None

    # ------------------------------------------------------------------ loading
    def weight_loading_and_quant(self, src_weight, src_bias, quant_type=None):
        """fastdm/layer/qlinear.py:18-54.  src_weight: list of [K, N_i] tensors (transposed views of
        the checkpoint's [N_i, K]); fused along N in list order."""
        if self.dtype == torch.float32:
            # fp32 layer (Wan's time embedder, embeddings.py:962: one row per batch element): torch.addmm like the
            # reference (qlinear.py:76); never quantised
            assert quant_type is None
            self.quant_type = None
            self.weight = torch.cat(list(src_weight), dim=1).to(device=self.device, dtype=torch.float32)
            self.bias = None if src_bias[0] is None else torch.cat(list(src_bias), dim=0).to(device=self.device, dtype=torch.float32)
            return
        assert self.dtype == BF16, "bf16 (or fp32 host-size) layers only in this build"
        w_nk = torch.cat([w.transpose(0, 1) for w in src_weight], dim=0).to(device=self.device, dtype=self.dtype).contiguous()
        if src_bias[0] is not None:
            assert self.has_bias, "Bias is not enabled but bias tensor is provided."
            self.bias = torch.cat(list(src_bias), dim=0).to(device=self.device, dtype=self.dtype).contiguous()
        self.set_weight_nk(w_nk, quant_type)

    def set_weight_nk(self, w_nk: torch.Tensor, quant_type=None):
        """w_nk: [N, K] bf16 on the GPU (K contiguous)."""
        self.quant_type = quant_type
        self.out_features, self.in_features = w_nk.shape
        if quant_type is None:
            self.weight = w_nk.t()
        elif quant_type in (FP8, torch.int8):
            n, k = w_nk.shape
            q = torch.empty((n, k), device=w_nk.device, dtype=quant_type)
            s = torch.empty((n, 1), device=w_nk.device, dtype=torch.float32)
            if quant_type == FP8:
                N.check(N.lib().fdm_quant_fp8_rowwise(N.ptr(q), N.ptr(s), N.ptr(w_nk), n, k, k, N.stream()), "weight quant fp8")
            else:  # symmetric per output channel (utils/quantization.py:29-33)
                N.check(N.lib().fdm_quant_int8_rowwise(N.ptr(q), N.ptr(s), None, N.ptr(w_nk), n, k, k, N.stream()), "weight quant int8")
                self.weight_asym_sumcol = q.to(torch.int32).sum(dim=1, dtype=torch.int32).reshape(1, n).contiguous()
            self.weight, self.weight_quant_scale = q.t(), s
        else:
            raise ValueError(f"Unsupported quantization type: {quant_type}")

    # ------------------------------------------------------------------ pre-quantised checkpoints (model/basemodel.py)
    def export_state(self):
        """The layer exactly as the kernels consume it: w [N, K] (fp8 / int8 / bf16 / fp32, K contiguous), per-output-channel
        scale, int8 column sums, bias."""
        d = {"w": self.weight.t().contiguous()}
        if self.quant_type is not None:
            d["s"] = self.weight_quant_scale
        if self.weight_asym_sumcol is not None:
            d["c"] = self.weight_asym_sumcol
        if self.bias is not None:
            d["b"] = self.bias
        return d

    def import_state(self, d, device=None):
        """Inverse of `export_state`: no quantisation pass, the stored bytes are the kernel operands."""
        dev = device or self.device
        w = d["w"].to(dev)
        if w.dtype in (FP8, torch.int8):
            self.quant_type = w.dtype
            self.weight_quant_scale = d["s"].to(dev)
            self.weight_asym_sumcol = d["c"].to(dev) if "c" in d else None
        else:
            assert w.dtype == self.dtype, f"stored {w.dtype} weight for a {self.dtype} layer"
            self.quant_type = None
        self.out_features, self.in_features = w.shape
        self.weight = w.t()
        self.bias = d["b"].to(dev) if "b" in d else None

    # ------------------------------------------------------------------ forward
    def quantize_input(self, x2d: torch.Tensor, act: int = 0, out: Optional[Quantized] = None) -> Quantized:
        """Per-token quantisation of this layer's input; act = N.EPI_GELU_TANH / N.EPI_GELU_ERF: x2d holds the pre-activations of the
        GELU in front of this layer and bf16(gelu(x)) is applied in the same pass (fused.quantize_rows)."""
        return fused.quantize_rows(x2d, None, self.quant_type, act0=act, out=out)

    def forward_q(self, x, epilogue: int = N.EPI_NONE, gate: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None,
                  rows_per_batch: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x: a `Quantized` (for W8A8 layers) or a bf16 [M,K] matrix (bf16 layers).  Epilogues:
        EPI_GELU_TANH / EPI_GELU_ERF / EPI_GATE_RESIDUAL (out = res + gate[b]*y; `out` may alias `res`)."""
        wt = self.weight  # [K, N], stride(0) == 1
        K, Nn = wt.shape
        a = x.q if isinstance(x, Quantized) else x
        M = a.shape[0]
        if out is None:
            out = torch.empty((M, Nn), device=a.device, dtype=self.dtype)
        g_ptr, g_stride = (N.ptr(gate), gate.stride(0)) if gate is not None else (None, 0)
        lib = N.lib()
        flops = 2.0 * M * Nn * K
        timing.log_gemm("fp8" if self.quant_type == FP8 else "int8" if self.quant_type == torch.int8 else "bf16", [(M, Nn, K, epilogue)])
        if self.quant_type == FP8:
          with span("gemm_w8a8_fp8_res" if epilogue >= N.EPI_GATE_RESIDUAL else "gemm_w8a8_fp8", flops):
            N.check(lib.fdm_gemm_w8a8_fp8(N.ptr(out), N.ptr(a), N.ptr(wt), N.ptr(x.scale), N.ptr(self.weight_quant_scale),
                                          N.ptr(self.bias), M, Nn, K, a.stride(0), wt.stride(1), out.stride(0), epilogue,
                                          g_ptr, g_stride, N.ptr(res), rows_per_batch, N.stream()), "fp8 linear")
        elif self.quant_type == torch.int8:
          import ctypes
          d, _, _ = self._desc(x, epilogue, gate, res, rows_per_batch, out)
          ws = N.gemm_int8_workspace(a.device)     # split-K tail: exact for int32 accumulators (bit-identical to the unsplit GEMM)
          with span("gemm_w8a8_int8_res" if epilogue >= N.EPI_GATE_RESIDUAL else "gemm_w8a8_int8", flops):
            N.check(lib.fdm_gemm_w8a8_int8_x2_ws(ctypes.byref(d), None, N.ptr(ws), ws.numel(), N.stream()), "int8 linear")
        else:
            assert a.dtype == self.dtype and a.stride(1) == 1
            with span("gemm_bf16_res" if epilogue >= N.EPI_GATE_RESIDUAL else "gemm_bf16", flops):
              N.check(lib.fdm_gemm_bf16(N.ptr(out), N.ptr(a), N.ptr(wt), N.ptr(self.bias), M, Nn, K, a.stride(0), wt.stride(1),
                                      out.stride(0), epilogue, g_ptr, g_stride, N.ptr(res), rows_per_batch, N.stream()),
                    "bf16 linear")
        return out

    def _desc(self, x, epilogue, gate, res, rows_per_batch, out):
        wt = self.weight
        K, Nn = wt.shape
        if self.quant_type is None:      # bf16 layer: x is the [M, K] activation matrix itself (fdm_gemm_bf16_desc has no scales)
            a = x
            assert a.dtype == self.dtype and a.stride(1) == 1
            M = a.shape[0]
            if out is None:
                out = torch.empty((M, Nn), device=a.device, dtype=self.dtype)
            d = N.GemmBf16Desc()
            d.out, d.a, d.b, d.bias = N.ptr(out), N.ptr(a), N.ptr(wt), N.ptr(self.bias)
            d.M, d.N, d.K, d.lda, d.ldb, d.ldc = M, Nn, K, a.stride(0), wt.stride(1), out.stride(0)
            d.epilogue = epilogue
            d.epi_gate, d.epi_gate_stride = (N.ptr(gate), gate.stride(0)) if gate is not None else (None, 0)
            d.epi_res, d.epi_rows_per_batch = N.ptr(res), rows_per_batch
            return d, out, 2.0 * M * Nn * K
        a = x.q
        M = a.shape[0]
        if out is None:
            out = torch.empty((M, Nn), device=a.device, dtype=self.dtype)
        if self.quant_type == torch.int8:
            d = N.GemmInt8Desc()
            d.azp_adj = N.ptr(self.weight_asym_sumcol) if x.azp is not None else None
            d.azp = N.ptr(x.azp)
        else:
            d = N.GemmFp8Desc()
        d.out, d.a, d.b = N.ptr(out), N.ptr(a), N.ptr(wt)
        d.scale_a, d.scale_b, d.bias = N.ptr(x.scale), N.ptr(self.weight_quant_scale), N.ptr(self.bias)
        d.M, d.N, d.K, d.lda, d.ldb, d.ldc = M, Nn, K, a.stride(0), wt.stride(1), out.stride(0)
        d.epilogue = epilogue
        d.epi_gate, d.epi_gate_stride = (N.ptr(gate), gate.stride(0)) if gate is not None else (None, 0)
        d.epi_res, d.epi_rows_per_batch = N.ptr(res), rows_per_batch
        return d, out, 2.0 * M * Nn * K

    @staticmethod
    def forward_q_pair(lin0, x0, lin1, x1, epilogue: int = N.EPI_NONE, gate0=None, res0=None, rpb0: int = 0, out0=None,
                       gate1=None, res1=None, rpb1: int = 0, out1=None):
        """Two independent linears (e.g. the image- and text-stream projection of a double block) in ONE launch
        when both are fp8; otherwise two launches.  Returns (out0, out1)."""
        if lin0.quant_type == FP8 and lin1.quant_type == FP8:
            import ctypes
            d0, o0, f0 = lin0._desc(x0, epilogue, gate0, res0, rpb0, out0)
            d1, o1, f1 = lin1._desc(x1, epilogue, gate1, res1, rpb1, out1)
            timing.log_gemm("fp8", [(d0.M, d0.N, d0.K, epilogue), (d1.M, d1.N, d1.K, epilogue)])
            with span("gemm_w8a8_fp8_res" if epilogue >= N.EPI_GATE_RESIDUAL else "gemm_w8a8_fp8", f0 + f1):
                N.check(N.lib().fdm_gemm_w8a8_fp8_x2(ctypes.byref(d0), ctypes.byref(d1), N.stream()), "fp8 linear pair")
            return o0, o1
        if lin0.quant_type == torch.int8 and lin1.quant_type == torch.int8:
            import ctypes
            d0, o0, f0 = lin0._desc(x0, epilogue, gate0, res0, rpb0, out0)
            d1, o1, f1 = lin1._desc(x1, epilogue, gate1, res1, rpb1, out1)
            timing.log_gemm("int8", [(d0.M, d0.N, d0.K, epilogue), (d1.M, d1.N, d1.K, epilogue)])
            with span("gemm_w8a8_int8_res" if epilogue >= N.EPI_GATE_RESIDUAL else "gemm_w8a8_int8", f0 + f1):
                ws = N.gemm_int8_workspace(x0.q.device)
                N.check(N.lib().fdm_gemm_w8a8_int8_x2_ws(ctypes.byref(d0), ctypes.byref(d1), N.ptr(ws), ws.numel(), N.stream()), "int8 linear pair")
            return o0, o1
        if lin0.quant_type is None and lin1.quant_type is None and lin0.dtype == BF16 and lin1.dtype == BF16 and min(x0.shape[0], x1.shape[0]) > 8:
            import ctypes                     # (M <= 8 keeps the GEMV kernel of fdm_gemm_bf16)
            d0, o0, f0 = lin0._desc(x0, epilogue, gate0, res0, rpb0, out0)
            d1, o1, f1 = lin1._desc(x1, epilogue, gate1, res1, rpb1, out1)
            timing.log_gemm("bf16", [(d0.M, d0.N, d0.K, epilogue), (d1.M, d1.N, d1.K, epilogue)])
            with span("gemm_bf16_res" if epilogue >= N.EPI_GATE_RESIDUAL else "gemm_bf16", f0 + f1):
                N.check(N.lib().fdm_gemm_bf16_x2(ctypes.byref(d0), ctypes.byref(d1), N.stream()), "bf16 linear pair")
            return o0, o1
        return (lin0.forward_q(x0, epilogue, gate0, res0, rpb0, out0), lin1.forward_q(x1, epilogue, gate1, res1, rpb1, out1))

    @staticmethod
    def forward_q_planes(items, planes: torch.Tensor):
        """Fused projections whose N = P*C output columns are split into P compact planes (q|k|v): `planes` is a
        contiguous [P, rows, C] buffer and `items` = [(linear, x, first_row)], each linear writing its M rows at
        first_row of every plane.  Two linears of one kind (fp8 / int8 / bf16) per launch through the GEMM's plane-split output."""
        import ctypes
        P_, rows, Cc = planes.shape
        assert planes.is_contiguous() and all(lin.weight.shape[1] == P_ * Cc for lin, _, _ in items)
        pending = []
        for lin, x, r0 in items:
            # (bf16 linears too since round 6: one plane-split launch for both streams instead of one launch per plane and stream)
            d, _, f = lin._desc(x, N.EPI_NONE, None, None, 0, planes[0, r0:])
            d.out_plane_cols, d.out_plane_stride = Cc, rows * Cc
            pending.append((lin.quant_type, d, f))
        while pending:
            qt, d0, f0 = pending.pop(0)
            mate = next((i for i, (q2, _, _) in enumerate(pending) if q2 == qt), None)
            if mate is not None:
                _, d1, f1 = pending.pop(mate)
            else:
                d1, f1 = type(d0)(), 0.0
                d1.M, d1.N, d1.K = 0, 1, 1          # an empty problem: skipped by the library
            timing.log_gemm("fp8" if qt == FP8 else "int8" if qt == torch.int8 else "bf16", [(d0.M, d0.N, d0.K, N.EPI_NONE), (d1.M, d1.N, d1.K, N.EPI_NONE)])
            with span("gemm_w8a8_fp8" if qt == FP8 else "gemm_w8a8_int8" if qt == torch.int8 else "gemm_bf16", f0 + f1):
                if qt is None:
                    N.check(N.lib().fdm_gemm_bf16_x2(ctypes.byref(d0), ctypes.byref(d1), N.stream()), "plane-split linear")
                elif qt == FP8:
                    N.check(N.lib().fdm_gemm_w8a8_fp8_x2(ctypes.byref(d0), ctypes.byref(d1), N.stream()), "plane-split linear")
                else:
                    ws = N.gemm_int8_workspace(planes.device)
                    N.check(N.lib().fdm_gemm_w8a8_int8_x2_ws(ctypes.byref(d0), ctypes.byref(d1), N.ptr(ws), ws.numel(), N.stream()),
                            "plane-split linear")

    def forward(self, input_tensor: torch.Tensor) -> torch.Tensor:
        """fastdm/layer/qlinear.py:56-81: quantise per token, scaled GEMM (+bias), restore the leading dims."""
        if self.dtype == torch.float32:
            y = input_tensor.float() @ self.weight
            return y if self.bias is None else y + self.bias
        ori_shape = input_tensor.shape
        x = input_tensor.reshape(-1, ori_shape[-1])
        if self.quant_type is not None:
            x = self.quantize_input(x if x.stride(1) == 1 else x.contiguous())
        elif not x.is_contiguous():
            x = x.contiguous()
        out = self.forward_q(x)
        return out.view(*ori_shape[:-1], self.weight.shape[-1])
