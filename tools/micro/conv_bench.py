#!/usr/bin/env python3
"""The SDXL UNet's 3 x 3 convolutions at 1024 x 1024 (latent 128 x 128, CFG batch 2): fdm_conv2d_3x3_nhwc_bf16 against F.conv2d (MIOpen) on
channels-last and on NCHW tensors, us per launch (HIP events, 20 launches back to back)."""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from fastdm_amd.kernel import fused  # noqa: E402

BF16 = torch.bfloat16
SHAPES = [  # (C_in, C_out, H = W, stride, upsample, how many per forward)
    (320, 320, 128, 1, False, 4), (320, 320, 128, 2, False, 1), (320, 640, 64, 1, False, 1), (640, 640, 64, 1, False, 3), (640, 640, 64, 2, False, 1),
    (640, 1280, 32, 1, False, 1), (1280, 1280, 32, 1, False, 7), (2560, 1280, 32, 1, False, 2), (1920, 1280, 32, 1, False, 1), (1280, 1280, 32, 1, True, 1),
    (1920, 640, 64, 1, False, 1), (1280, 640, 64, 1, False, 1), (960, 640, 64, 1, False, 1), (640, 640, 64, 1, True, 1),
    (960, 320, 128, 1, False, 1), (640, 320, 128, 1, False, 2),
]


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = [0.0, 0.0, 0.0]
print(f"{'C_in':>5s} {'C_out':>5s} {'HW':>4s} {'s':>2s} {'up':>3s} {'x':>2s} {'GFLOP':>7s} | {'HIP us':>8s} {'frac':>6s} | {'MIOpen NHWC':>11s} | {'MIOpen NCHW':>11s}")
for ci, co, hw, st, up, cnt in SHAPES:
    x = torch.randn(2, ci, hw, hw, device="cuda").to(BF16)
    xl = x.contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, 3, 3, device="cuda") * (9 * ci) ** -0.5).to(BF16)
    wl = w.contiguous(memory_format=torch.channels_last)
    b = torch.randn(co, device="cuda").to(BF16)
    ho = 2 * hw if up else (hw - 1) // st + 1
    gf = 2.0 * 2 * ho * ho * co * 9 * ci / 1e9
    a = t(lambda: fused.conv2d(xl, wl, b, st, 1, upsample=up))
    m1 = t(lambda: F.conv2d(F.interpolate(xl, scale_factor=2.0, mode="nearest") if up else xl, wl, b, st, 1))
    m2 = t(lambda: F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x, w, b, st, 1))
    tot[0] += a * cnt
    tot[1] += m1 * cnt
    tot[2] += m2 * cnt
    print(f"{ci:5d} {co:5d} {hw:4d} {st:2d} {int(up):3d} {cnt:2d} {gf:7.1f} | {a:8.1f} {gf / a * 0.4:6.3f} | {m1:11.1f} | {m2:11.1f}")
print(f"per forward (weighted): HIP {tot[0] / 1e3:.2f} ms, MIOpen channels-last {tot[1] / 1e3:.2f} ms, MIOpen NCHW {tot[2] / 1e3:.2f} ms")
