import torch
dev="cuda"
for name, M, K, N in (("dbl ff2", 8704, 12288, 3072), ("sgl out", 9216, 15360, 3072), ("dbl qkv", 8704, 3072, 9216)):
    a = torch.randn(M, K, device=dev).to(torch.float8_e4m3fn)
    b = torch.randn(N, K, device=dev).to(torch.float8_e4m3fn).t()
    sa, sb = torch.rand(M, 1, device=dev) + 0.5, torch.rand(1, N, device=dev) + 0.5
    for _ in range(3):
        torch._scaled_mm(a, b, scale_a=sa, scale_b=sb, out_dtype=torch.bfloat16)
torch.cuda.synchronize()
