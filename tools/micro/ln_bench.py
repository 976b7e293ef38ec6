"""Time fdm_ln_modulate (+fp8 quant) on the FLUX image stream: [8192, 3072] rows.  FDM_LN_THREADS=128|256 selects the experimental
row-width variants."""
import time, torch
from fastdm_amd.kernel import fused
BF16 = torch.bfloat16
x = torch.randn(8192, 3072, device="cuda").to(BF16)
sc = torch.randn(1, 3072, device="cuda").to(BF16) * 0.1
sh = torch.randn(1, 3072, device="cuda").to(BF16) * 0.1
f = lambda: fused.ln_modulate(x, sc, sh, 8192, quant=torch.float8_e4m3fn)
for _ in range(5): f()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(50): f()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 50
print(f"ln_modulate+fp8 quant 8192x3072: {dt * 1e6:.1f} us  {(8192 * 3072 * 3) / dt / 1e12:.2f} TB/s")
