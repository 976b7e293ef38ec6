"""Time the HBM-bound element-wise kernels at the FLUX C3 shapes (run on the GPU box): achieved GB/s = algorithmic bytes / time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fastdm_amd.kernel import fused
BF16, FP8 = torch.bfloat16, torch.float8_e4m3fn


def timeit(f, iters=50):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def report(name, sec, nbytes):
    print(f"{name:44s} {sec * 1e6:8.1f} us  {nbytes / sec / 1e12:6.2f} TB/s  ({nbytes / 1e6:.0f} MB)")


for rows in (8192, 8704, 512):
    x = torch.randn(rows, 3072, device="cuda").to(BF16)
    sc = (torch.randn(1, 3072, device="cuda") * 0.1).to(BF16)
    sh = (torch.randn(1, 3072, device="cuda") * 0.1).to(BF16)
    for q in (FP8, torch.int8):
        report(f"ln_modulate+{'fp8' if q == FP8 else 'int8'} quant [{rows},3072]", timeit(lambda: fused.ln_modulate(x, sc, sh, rows, quant=q)), rows * 3072 * 3)
a = torch.randn(8704, 3072, device="cuda").to(BF16)
m = torch.randn(8704, 12288, device="cuda").to(BF16)
for q in (FP8, torch.int8):
    t = 'fp8' if q == FP8 else 'int8'
    report(f"quant {t} cat2 [8704, 3072+12288]", timeit(lambda: fused.quantize_rows(a, m, q)), 8704 * 15360 * 3)
    report(f"quant {t} [8704, 12288]", timeit(lambda: fused.quantize_rows(m, None, q)), 8704 * 12288 * 3)
    report(f"quant {t} [8704, 3072]", timeit(lambda: fused.quantize_rows(a, None, q)), 8704 * 3072 * 3)
