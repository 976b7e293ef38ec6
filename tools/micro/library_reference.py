"""Context numbers, not product code: what the ROCm libraries shipped with this image reach on the FLUX shapes
(hipBLASLt through torch._scaled_mm / torch.mm, the flash kernel behind F.scaled_dot_product_attention)."""
import torch, time
dev = "cuda"
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
shapes = [("dbl qkv", 8704, 3072, 9216), ("dbl ff1", 8704, 3072, 12288), ("dbl ff2", 8704, 12288, 3072), ("sgl out", 9216, 15360, 3072), ("sgl qkv+mlp", 9216, 3072, 21504)]
for name, M, K, N in shapes:
    a = torch.randn(M, K, device=dev).to(torch.float8_e4m3fn)
    b = torch.randn(N, K, device=dev).to(torch.float8_e4m3fn).t()
    sa, sb = torch.rand(M, 1, device=dev) + 0.5, torch.rand(1, N, device=dev) + 0.5
    for tag, fn in (("fp8 rowwise", lambda: torch._scaled_mm(a, b, scale_a=sa, scale_b=sb, out_dtype=torch.bfloat16)),
                    ("fp8 tensorwise", lambda: torch._scaled_mm(a, b, scale_a=sa[:1, :1].reshape(()), scale_b=sb[:1, :1].reshape(()), out_dtype=torch.bfloat16))):
        try:
            t = bench(fn)
            print(f"{name:12s} {tag:15s} {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s  ({2*M*N*K/t/5e15:.3f} of 5 PF)")
        except Exception as e:
            print(f"{name:12s} {tag:15s} failed: {str(e)[:120]}")
    ab, bb = torch.randn(M, K, device=dev, dtype=torch.bfloat16), torch.randn(K, N, device=dev, dtype=torch.bfloat16)
    t = bench(lambda: ab @ bb)
    print(f"{name:12s} {'bf16 mm':15s} {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s  ({2*M*N*K/t/2.5e15:.3f} of 2.5 PF)")
import torch.nn.functional as F
for S, H, D in ((8704, 24, 128), (4608, 24, 128), (4429, 38, 64)):
    q, k, v = (torch.randn(1, H, S, D, device=dev, dtype=torch.bfloat16) for _ in range(3))
    try:
        t = bench(lambda: F.scaled_dot_product_attention(q, k, v), n=10)
        print(f"sdpa S={S} H={H} D={D}: {t*1e6:8.1f} us  {4*S*S*H*D/t/1e12:7.1f} TF/s  ({4*S*S*H*D/t/2.5e15:.3f} of 2.5 PF)")
    except Exception as e:
        print("sdpa failed", str(e)[:200])
