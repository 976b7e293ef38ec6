#!/usr/bin/env python3
"""Same-process A/B of the shipped fp8 GEMM kernels against hipBLASLt (torch._scaled_mm, row-wise scales) on the FLUX launch shapes,
operands CAPTURED from one forward of the synthetic model (real fp8 activation codes + that layer's quantised weights), warm and cold
(context numbers -- hipBLASLt is never on the product path).

    python3 tools/micro/gemm_lib_ab.py [--reps 20] > gpurun_out/<tag>_gemm_lib_ab.txt

warm: the same (A, B) pair launched back to back (weights stay in the 256 MiB Infinity Cache).
cold: every launch takes the NEXT layer's weights (19 / 38 layers x 28-47 MB: nothing survives a lap) and one of 3 activation buffers,
      with 290 MB of attention-sized traffic (a copy) between launches, as in the step.
Each pair of arms alternates launch by launch groups (ours, library, ours, ...) three times; the table shows the median.
"""
import argparse
import statistics
import sys

import torch

sys.path.insert(0, ".")
from fastdm_amd import _native as N  # noqa: E402
from fastdm_amd.kernel.fused import Quantized  # noqa: E402
from fastdm_amd.layer.qlinear import QLinear  # noqa: E402

FP8, BF16 = torch.float8_e4m3fn, torch.bfloat16
KNAME = {0: "gemm_kernel", 1: "gemm_w4_kernel", 2: "gemm_p4_kernel"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=2048)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    from fastdm_amd.model_entry import create_model
    from fastdm_amd.pipeline import SyntheticFluxPipeline
    from fastdm_amd.synthetic import SyntheticFluxStateDict
    model = create_model("flux", ckpt_path=SyntheticFluxStateDict(seed=0), dtype=BF16, quant_type=FP8, kernel_backend="hip").eval()
    pipe = SyntheticFluxPipeline(model, seed=0)
    inp = pipe.prepare(1, a.height, a.width, 512, seed=0)
    core = model.core_model
    # ---- capture: the Quantized input of the first launch of every (layer kind) during one forward
    cap = {}
    orig_desc = QLinear._desc

    def desc(self, x, epilogue, gate, res, rpb, out):
        key = (self.in_features, self.out_features, x.q.shape[0])
        if isinstance(x, Quantized) and key not in cap:
            cap[key] = Quantized(x.q.clone(), x.scale.clone(), None)
        return orig_desc(self, x, epilogue, gate, res, rpb, out)

    orig_fq = QLinear.forward_q

    def fq(self, x, *args, **kw):
        if isinstance(x, Quantized):
            key = (self.in_features, self.out_features, x.q.shape[0])
            if key not in cap:
                cap[key] = Quantized(x.q.clone(), x.scale.clone(), None)
        return orig_fq(self, x, *args, **kw)

    QLinear._desc, QLinear.forward_q = desc, fq
    with torch.no_grad():
        t = torch.full((1,), 1.0, device=dev, dtype=BF16)
        g = torch.full((1,), 3.5, device=dev, dtype=torch.float32)
        model(hidden_states=inp["hidden_states"], timestep=t, guidance=g, pooled_projections=inp["pooled_projections"],
              encoder_hidden_states=inp["encoder_hidden_states"], txt_ids=inp["txt_ids"], img_ids=inp["img_ids"], joint_attention_kwargs=None,
              return_dict=False)
    QLinear._desc, QLinear.forward_q = orig_desc, orig_fq
    torch.cuda.synchronize()
    dbl, sgl = core.transformer_blocks, core.single_transformer_blocks
    s_img = inp["hidden_states"].shape[1]
    s_all = s_img + 512
    kinds = [   # name, layers (weights to rotate through), M, epilogue of the shipped launch, gate + residual?
        ("double qkv      M=%d K=3072 N=9216" % s_img, [b.attn.qkv for b in dbl], s_img, N.EPI_NONE),
        ("double ff1      M=%d K=3072 N=12288" % s_img, [b.ff.act_fn.proj for b in dbl], s_img, N.EPI_NONE),
        ("double ff1+tanh M=%d K=3072 N=12288" % s_img, [b.ff.act_fn.proj for b in dbl], s_img, N.EPI_GELU_TANH),
        ("double to_out   M=%d K=3072 N=3072 +res" % s_img, [b.attn.to_out for b in dbl], s_img, N.EPI_GATE_RESIDUAL),
        ("double ff2      M=%d K=12288 N=3072 +res" % s_img, [b.ff.ff_out_proj for b in dbl], s_img, N.EPI_GATE_RESIDUAL),
        ("single qkv      M=%d K=3072 N=9216" % s_all, [b.attn.qkv for b in sgl], s_all, N.EPI_NONE),
        ("single mlp      M=%d K=3072 N=12288" % s_all, [b.proj_mlp for b in sgl], s_all, N.EPI_NONE),
        ("single mlp+erf  M=%d K=3072 N=12288" % s_all, [b.proj_mlp for b in sgl], s_all, N.EPI_GELU_ERF),
        ("single proj_out M=%d K=15360 N=3072 +res" % s_all, [b.proj_out for b in sgl], s_all, N.EPI_GATE_RESIDUAL),
    ]
    junk_src = torch.empty(145 << 20, dtype=torch.uint8, device=dev)
    junk_dst = torch.empty_like(junk_src)

    def evict():
        junk_dst.copy_(junk_src)     # 145 MB read + 145 MB written: the footprint of an attention call between two GEMMs of the step

    print(f"{'launch kind':48s} {'arm':22s} {'warm us':>9s} {'PF/s':>6s} {'frac':>6s} | {'cold us':>9s} {'PF/s':>6s} {'frac':>6s}")
    for name, layers, M, epi in kinds:
        lin0 = layers[0]
        key = (lin0.in_features, lin0.out_features, M)
        if key not in cap:
            print(f"{name}: no captured operand for {key}; have {sorted(cap)}")
            continue
        x = cap[key]
        xs = [x] + [Quantized(x.q.clone(), x.scale.clone(), None) for _ in range(2)]
        K_, Nn = lin0.in_features, lin0.out_features
        flops = 2.0 * M * Nn * K_
        out = torch.empty(M, Nn, dtype=BF16, device=dev)
        res = torch.randn(M, Nn, device=dev).to(BF16) if epi == N.EPI_GATE_RESIDUAL else None
        gate = torch.randn(1, Nn, device=dev).to(BF16) if res is not None else None

        def ours(i, cold):
            lin = layers[i % len(layers)] if cold else lin0
            lin.forward_q(xs[i % 3] if cold else x, epilogue=epi, gate=gate, res=res, rows_per_batch=M, out=out)

        def lib(i, cold):
            lin = layers[i % len(layers)] if cold else lin0
            xx = xs[i % 3] if cold else x
            y = torch._scaled_mm(xx.q, lin.weight, scale_a=xx.scale, scale_b=lin.weight_quant_scale.t(), bias=lin.bias, out_dtype=BF16)
            if epi == N.EPI_GELU_TANH:
                y = torch.nn.functional.gelu(y, approximate="tanh")
            elif epi == N.EPI_GELU_ERF:
                y = torch.nn.functional.gelu(y)
            elif epi == N.EPI_GATE_RESIDUAL:
                y = res + gate * y
            return y

        def timed(fn, cold):
            """per-launch GEMM time: HIP events around each launch alone (the eviction copy and, for the library arm, the unfused
            epilogue kernels are outside / inside as labelled)"""
            for i in range(3):
                fn(i, cold)
            ts = []
            for i in range(a.reps):
                if cold:
                    evict()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn(i + 3, cold)
                e1.record()
                ts.append((e0, e1))
            torch.cuda.synchronize()
            return statistics.median(x0.elapsed_time(x1) for x0, x1 in ts) * 1e3

        def lib_gemm_only(i, cold):
            lin = layers[i % len(layers)] if cold else lin0
            xx = xs[i % 3] if cold else x
            return torch._scaled_mm(xx.q, lin.weight, scale_a=xx.scale, scale_b=lin.weight_quant_scale.t(), bias=lin.bias, out_dtype=BF16)

        arms = [("ours", ours), ("hipBLASLt gemm only", lib_gemm_only)]
        if epi != N.EPI_NONE:
            arms.append(("hipBLASLt + torch epi", lib))
        res_tab = {nm: {"warm": [], "cold": []} for nm, _ in arms}
        try:
            for _ in range(3):
                for nm, fn in arms:
                    for cold in (False, True):
                        res_tab[nm]["cold" if cold else "warm"].append(timed(fn, cold))
        except Exception as e:  # noqa: BLE001
            print(f"{name}: failed: {str(e)[:200]}")
            continue
        ours(0, False)
        kern = KNAME.get(int(N.lib().fdm_gemm_last_kernel()), "?")
        for nm, _ in arms:
            w, c = statistics.median(res_tab[nm]["warm"]), statistics.median(res_tab[nm]["cold"])
            label = f"{nm} ({kern})" if nm == "ours" else nm
            print(f"{name:48s} {label:22s} {w:9.1f} {flops / w / 1e9:6.3f} {flops / w / 1e9 / 5.0:6.3f} | {c:9.1f} {flops / c / 1e9:6.3f} {flops / c / 1e9 / 5.0:6.3f}")
        sys.stdout.flush()
    # correctness of the comparison itself: same operands -> results within bf16 rounding of each other
    x = cap[(3072, 9216, s_img)]
    lin = dbl[0].attn.qkv
    o1 = lin.forward_q(x)
    o2 = torch._scaled_mm(x.q, lin.weight, scale_a=x.scale, scale_b=lin.weight_quant_scale.t(), bias=lin.bias, out_dtype=BF16)
    d = (o1.float() - o2.float()).abs()
    print(f"ours vs hipBLASLt on the captured qkv operands: max |diff| {d.max().item():.4g} on mean |y| {o2.float().abs().mean().item():.4g}; "
          f"{(d > 0).float().mean().item():.3%} of the elements differ")


if __name__ == "__main__":
    main()
