"""Ad-hoc GPU diagnostics (not a test): prints error statistics for kernels under bring-up."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastdm_amd.kernel import operators_set as K
from fastdm_amd import _native as N
from oracle import ops as O

FP8, BF16 = torch.float8_e4m3fn, torch.bfloat16
torch.manual_seed(0)


def stats(name, got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    d = (got - ref).abs()
    rel = d / ref.abs().clamp(min=1e-6)
    nz = d > 0
    print(f"{name}: max|d|={d.max():.5g} mean|d|={d.mean():.5g} frac!=0={nz.float().mean():.4%} max rel={rel.max():.4g} "
          f"|ref|mean={ref.abs().mean():.4g}")
    if nz.any():
        idx = torch.nonzero(d == d.max())[0].tolist()
        print("   worst at", idx, "got", got[tuple(idx)].item(), "ref", ref[tuple(idx)].item())
    return d


def fp8_exact():
    # small integers: exact in fp8, products/sums exact in fp32 -> must match bit for bit
    for (M, Kd, Nn) in ((32, 64, 32), (256, 128, 256), (300, 384, 272), (2, 1280, 1280)):
        a = torch.randint(-3, 4, (M, Kd)).float()
        b = torch.randint(-3, 4, (Nn, Kd)).float()
        ref = a @ b.t()
        sa, sb = torch.ones(M, 1), torch.ones(Nn, 1)
        got = K.fp8_matmul(a.to(FP8).cuda(), b.to(FP8).cuda().t(), sa.cuda(), sb.cuda(), BF16, None)
        d = stats(f"fp8 int-valued {M}x{Kd}x{Nn}", got, ref.to(BF16))
        if d.max() > 0:
            bad = torch.nonzero(d > 0)
            print("   bad rows:", sorted(set(bad[:, 0].tolist()))[:20], " bad cols:", sorted(set(bad[:, 1].tolist()))[:20])
    # one-hot probing of the K mapping: A[i, ka]=1, B[j, kb]=1
    M = Nn = 32
    Kd = 64
    for ka in (0, 1, 15, 16, 31, 32, 47, 48, 63):
        a = torch.zeros(M, Kd); a[:, ka] = 1
        b = torch.zeros(Nn, Kd)
        for kb in range(Kd):
            b[kb % Nn, kb] += (kb + 1)
        got = K.fp8_matmul(a.to(FP8).cuda(), b.to(FP8).cuda().t(), torch.ones(M, 1).cuda(), torch.ones(Nn, 1).cuda(), BF16, None).cpu().float()
        print(f"   one-hot ka={ka}: row0 nonzeros ->", {int(j): got[0, j].item() for j in torch.nonzero(got[0]).flatten().tolist()})
    # denormal operands
    a = torch.full((64, 128), 2.0 ** -8)
    b = torch.full((64, 128), 1.0)
    got = K.fp8_matmul(a.to(FP8).cuda(), b.to(FP8).cuda().t(), torch.ones(64, 1).cuda(), torch.ones(64, 1).cuda(), BF16, None)
    print("   denormal a=2^-8 x 128 ones -> expect 0.5:", got[0, :4].tolist())
    # random
    for (M, Kd, Nn) in ((130, 384, 320), (512, 3072, 3072)):
        a = torch.randn(M, Kd).to(FP8); b = torch.randn(Nn, Kd).to(FP8)
        sa, sb = torch.rand(M, 1) + 0.5, torch.rand(Nn, 1) * 0.02 + 0.001
        got = K.fp8_matmul(a.cuda(), b.cuda().t(), sa.cuda(), sb.cuda(), BF16, None)
        ref = O.fp8_matmul(a, b.t(), sa, sb, BF16, None)
        stats(f"fp8 randn {M}x{Kd}x{Nn}", got, ref)
        ref64 = ((a.double() @ b.double().t()) * sa.double() * sb.double().t())
        stats("    got vs fp64", got, ref64.float())
        stats("    oracle vs fp64", ref, ref64.float())


def misc():
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(64, 2 * 160, generator=g) * 2).to(BF16)
    stats("gelu_and_mul", K.gelu_and_mul(x.cuda()), O.gelu_and_mul(x))
    x = (torch.randn(1, 500, 24, 128, generator=g) * 2).to(BF16); w = (1 + 0.1 * torch.randn(128, generator=g)).to(BF16)
    stats("rms 128", K.rms_norm(x.cuda(), w.cuda(), 1e-6), O.rms_norm(x, w, 1e-6))
    x = (torch.randn(2, 30, 5120, generator=g)).to(BF16); w = (1 + 0.1 * torch.randn(5120, generator=g)).to(BF16)
    stats("rms 5120", K.rms_norm(x.cuda(), w.cuda(), 1e-6), O.rms_norm(x, w, 1e-6))
    x = (torch.randn(3, 40960, generator=g)).to(BF16); w = torch.ones(40960, dtype=BF16)
    stats("rms 40960", K.rms_norm(x.cuda(), w.cuda(), 1e-5), O.rms_norm(x, w, 1e-5))
    M, Kd, Nn = 300, 72, 528
    a, w, bias = (torch.randn(M, Kd, generator=g)).to(BF16), (torch.randn(Nn, Kd, generator=g) * Kd ** -0.5).to(BF16), (torch.randn(Nn, generator=g) * 0.1).to(BF16)
    y = torch.addmm(bias.float(), a.float(), w.float().t()).to(BF16)
    out = torch.empty(M, Nn, dtype=BF16, device="cuda")
    ad, wd, bd = a.cuda(), w.cuda(), bias.cuda()
    N.check(N.lib().fdm_gemm_bf16(out.data_ptr(), ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), M, Nn, Kd, Kd, Kd, Nn, 0, None, 0, None, 0, N.stream()), "g")
    stats("bf16 gemm K=72", out, y)
    for epi, want in ((1, torch.nn.functional.gelu(y, approximate="tanh")), (2, torch.nn.functional.gelu(y))):
        N.check(N.lib().fdm_gemm_bf16(out.data_ptr(), ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), M, Nn, Kd, Kd, Kd, Nn, epi, None, 0, None, 0, N.stream()), "g")
        stats(f"bf16 gemm epi {epi}", out, want)


if __name__ == "__main__":
    fp8_exact()
    misc()
