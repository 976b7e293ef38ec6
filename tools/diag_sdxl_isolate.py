#!/usr/bin/env python3
"""WHY does BASELINE configs[0] (SDXL-base 512 x 512, 4 steps) sit 1.15 -> 1.48 x the two-implementation envelope in bf16?
Runs the control comparison of tests/test_sdxl_gpu.py::test_baseline_config1_sdxl_against_the_two_implementation_control with the suspected
stages swapped, one at a time, for arithmetic that is the CPU reference's:
    default      the NCHW model (round 5's shipped path: fdm_group_norm, MIOpen bf16 convolutions)
    gn_torch     torch's GroupNorm / SiLU instead of fdm_group_norm (FDM_GN_TORCH=1)
    conv_fp32    every convolution as fp32 F.conv2d on the bf16 operands, rounded to bf16 once (what a CPU bf16 convolution computes)
    both         gn_torch + conv_fp32
    hip_nhwc     (round 6) the channels-last model: fdm_conv2d_3x3_nhwc_bf16 (fp32 accumulation) + fdm_group_norm_nhwc
and prints mean|HIP - ref| / mean|control - ref| per step and per variant (and the step-0 stage taps of the variant that moves it).
    python3 tools/diag_sdxl_isolate.py > gpurun_out/<tag>_sdxl_isolate.txt
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sdxl as OX  # noqa: E402   (checker side: the fixture's tap helper)

BF16 = torch.bfloat16
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    from oracle.make_golden import load as tload      # the suite's own fixture reader (bf16 stored as uint16)
    from fastdm_amd.model_entry import create_model
    from fastdm_amd.pipeline import SyntheticSDXLPipeline
    from fastdm_amd.synthetic import SyntheticSDXLStateDict, sdxl_step_inputs
    g, c = tload("sdxl_c1_512x512_4steps.npz"), tload("sdxl_c1_control.npz")
    seed, steps, hw = int(g["seed"]), int(g["steps"]), int(g["height"])
    real_conv = F.conv2d

    def conv_fp32(x, w, b=None, stride=1, padding=0, *a, **k):
        if x.dtype != BF16:
            return real_conv(x, w, b, stride, padding, *a, **k)
        return real_conv(x.float(), w.float(), None if b is None else b.float(), stride, padding, *a, **k).to(BF16)

    for tag, qt in (("bf16", None), ("fp8", torch.float8_e4m3fn)):
        model = create_model("sdxl", ckpt_path=SyntheticSDXLStateDict(seed=seed), dtype=BF16, quant_type=qt, kernel_backend="hip", channels_last=False).eval()
        pipe = SyntheticSDXLPipeline(model, seed=seed)
        ts, sig = pipe.schedule(steps)
        inp = sdxl_step_inputs(2, hw, hw, int(g["n_txt"]), seed=seed)
        sub = tuple(int(x) for x in c["tap_sub"])
        print(f"==== {tag}: mean|HIP - ref| / mean|control - ref| per step (limits of the test: measured default + 10 %)")
        model_cl = create_model("sdxl", ckpt_path=SyntheticSDXLStateDict(seed=seed), dtype=BF16, quant_type=qt, kernel_backend="hip", channels_last=True).eval()
        model_nchw = model
        for variant in ("default", "gn_torch", "conv_fp32", "both", "hip_nhwc", "nhwc_gn_torch", "nhwc_conv_torch", "nhwc_conv_fp32"):
            model = model_cl if "nhwc" in variant else model_nchw
            os.environ["FDM_GN_TORCH"] = "1" if variant in ("gn_torch", "both", "nhwc_gn_torch") else "0"
            os.environ["FDM_CONV_TORCH"] = "1" if variant in ("nhwc_conv_torch", "nhwc_conv_fp32") else "0"
            F.conv2d = conv_fp32 if variant in ("conv_fp32", "both", "nhwc_conv_fp32") else real_conv
            ratios, taps_line = [], []
            with torch.no_grad():
                for i in range(steps):
                    prev = (inp["sample"][0:1].float() * float(sig[0])).to(BF16) if i == 0 else g[f"lat_{tag}_{i - 1}"].cuda()
                    x = (prev.float() / ((float(sig[i]) ** 2 + 1.0) ** 0.5)).to(BF16)
                    taps = {}
                    restore = OX.sdxl_block_taps(model.core_model, taps, sub) if i == 0 else (lambda: None)
                    eps = model(x.expand(2, -1, -1, -1).contiguous(), torch.tensor(float(ts[i])).cuda(), inp["encoder_hidden_states"], inp["added_cond_kwargs"])[0]
                    restore()
                    ref = g[f"eps_{tag}_{i}"].float()
                    d_hip = (eps.float().cpu() - ref).abs().mean().item()
                    d_ctl = (c[f"ceps_{tag}_{i}"].float() - ref).abs().mean().item()
                    ratios.append(d_hip / d_ctl)
                    if i == 0:
                        for name in OX.SDXL_TAPS:
                            r = c[f"tap_{tag}_ref_{name}"].float()
                            th = (taps[name].float() - r).abs().mean().item()
                            tc = (c[f"tap_{tag}_ctl_{name}"].float() - r).abs().mean().item()
                            taps_line.append(f"{name} {th:.5f}/{tc:.5f}")
            F.conv2d = real_conv
            print(f"  {variant:10s} " + "  ".join(f"{r:.3f}" for r in ratios) + "    step-0 stage taps |HIP - ref| / |control - ref|: " + ", ".join(taps_line))
        os.environ["FDM_GN_TORCH"] = "0"
        os.environ["FDM_CONV_TORCH"] = "0"
        del model, model_cl, model_nchw, pipe
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
