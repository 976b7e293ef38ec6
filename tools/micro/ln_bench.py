"""Time fdm_ln_modulate (+fp8 quant) on the FLUX image stream: [8192, 3072] rows, with `scale` raw and with bf16(1 + scale)
supplied by the producer (scale_has_one); the two outputs must be bit-identical."""
import time, torch
from fastdm_amd.kernel import fused
BF16 = torch.bfloat16
x = torch.randn(8192, 3072, device="cuda").to(BF16)
sc = torch.randn(1, 3072, device="cuda").to(BF16) * 0.1
sh = torch.randn(1, 3072, device="cuda").to(BF16) * 0.1
sc1 = (1 + sc)
assert sc1.dtype == BF16
fa = lambda: fused.ln_modulate(x, sc, sh, 8192, quant=torch.float8_e4m3fn)
fb = lambda: fused.ln_modulate(x, sc1, sh, 8192, quant=torch.float8_e4m3fn, scale_has_one=True)
qa, qb = fa()[1], fb()[1]
print("bit-identical:", torch.equal(qa.q.view(torch.uint8), qb.q.view(torch.uint8)) and torch.equal(qa.scale, qb.scale))
for rep in range(3):
    for name, f in (("raw scale", fa), ("1 + scale supplied", fb)):
        for _ in range(5): f()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(100): f()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 100
        print(f"ln_modulate+fp8 quant 8192x3072 [{name}]: {dt * 1e6:.1f} us  {(8192 * 3072 * 3) / dt / 1e12:.2f} TB/s")
