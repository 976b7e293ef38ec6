"""Worker of tests/test_ops_gpu.py::test_gemm_p4_*: runs a fixed list of fp8 GEMM launches through the C-ABI under whatever
kernel policy the environment selects (FDM_GEMM_P4 / FDM_GEMM_KERNEL are read once per process) and prints one JSON line:
per case the kernel that ran (fdm_gemm_last_kernel) and a sha256 of the bf16 output bytes."""
import ctypes
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fastdm_amd import _native as N  # noqa: E402

BF16, FP8 = torch.bfloat16, torch.float8_e4m3fn


def problem(g, m, k, n, bias=True):
    a = (torch.randn(m, k, generator=g) * 1.5).to(FP8).cuda()
    b = (torch.randn(n, k, generator=g) * 1.5).to(FP8).cuda()          # [N, K]: the [K, N] weight with stride(0) == 1
    sa = (torch.rand(m, 1, generator=g) * 0.02 + 0.002).cuda()
    sb = (torch.rand(n, 1, generator=g) * 0.02 + 0.002).cuda()
    bs = (torch.randn(n, generator=g) * 0.5).to(BF16).cuda() if bias else None
    return dict(a=a, b=b, sa=sa, sb=sb, bias=bs, m=m, k=k, n=n)


def desc(pr, out, epi, ldc, plane_cols=0, plane_stride=0):
    d = N.GemmFp8Desc()
    d.out, d.a, d.b, d.scale_a, d.scale_b, d.bias = N.ptr(out), N.ptr(pr["a"]), N.ptr(pr["b"]), N.ptr(pr["sa"]), N.ptr(pr["sb"]), N.ptr(pr["bias"])
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc = pr["m"], pr["n"], pr["k"], pr["k"], pr["k"], ldc
    d.epilogue = epi
    d.epi_gate, d.epi_gate_stride, d.epi_res, d.epi_rows_per_batch = None, 0, None, 0
    d.out_plane_cols, d.out_plane_stride = plane_cols, plane_stride
    return d


def main():
    g = torch.Generator().manual_seed(1234)
    res = {}
    s = N.stream()
    # single problems, the three epilogues the persistent kernel implements, with and without a bias, 1.6 .. 4.8 rounds of 256 CUs
    for name, (m, k, n, epi, bias) in {
        "plain_bias": (4608, 1024, 4096, N.EPI_NONE, True),          # 18 x 16 = 288 tiles: the second round has 32 tiles
        "plain_nobias": (8192, 768, 3072, N.EPI_NONE, False),        # the shortest K the kernel takes (6 steps)
        "tanh": (4096, 1280, 6144, N.EPI_GELU_TANH, True),           # 384 tiles, an odd number of K-step pairs
        "erf": (8704, 1024, 3072, N.EPI_GELU_ERF, True),             # tiles_m = 34: the last raster group has 2 rows
        "erf_nobias": (2560, 2048, 8192, N.EPI_GELU_ERF, False),
    }.items():
        pr = problem(g, m, k, n, bias)
        out = torch.full((m, n), 3.0, dtype=BF16, device="cuda")
        rc = N.lib().fdm_gemm_w8a8_fp8(N.ptr(out), N.ptr(pr["a"]), N.ptr(pr["b"]), N.ptr(pr["sa"]), N.ptr(pr["sb"]), N.ptr(pr["bias"]), m, n, k, k, k, n,
                                       epi, None, 0, None, 0, s)
        assert rc == 0, (name, rc)
        kern = N.lib().fdm_gemm_last_kernel()
        torch.cuda.synchronize()
        res[name] = [kern, hashlib.sha256(out.cpu().view(torch.int16).numpy().tobytes()).hexdigest()]
    # image + text pair in one launch, fused q|k|v written as three planes (the FLUX double-block projection)
    hd3, s_txt, s_img, kk = 1024, 512, 7168, 1024      # 28 x 12 + 2 x 12 = 360 tiles
    p_img, p_txt = problem(g, s_img, kk, 3 * hd3), problem(g, s_txt, kk, 3 * hd3)
    planes = torch.full((3, s_txt + s_img, hd3), 5.0, dtype=BF16, device="cuda")
    d_txt = desc(p_txt, planes[0, :s_txt], N.EPI_NONE, hd3, hd3, (s_txt + s_img) * hd3)
    d_img = desc(p_img, planes[0, s_txt:], N.EPI_NONE, hd3, hd3, (s_txt + s_img) * hd3)
    rc = N.lib().fdm_gemm_w8a8_fp8_x2(ctypes.byref(d_img), ctypes.byref(d_txt), s)
    assert rc == 0, rc
    kern = N.lib().fdm_gemm_last_kernel()
    torch.cuda.synchronize()
    res["pair_planes"] = [kern, hashlib.sha256(planes.cpu().view(torch.int16).numpy().tobytes()).hexdigest()]
    # pair with a GELU epilogue, no planes
    p0, p1 = problem(g, 4096, 1024, 4096), problem(g, 256, 1024, 4096)      # 256 + 16 tiles
    o0, o1 = (torch.full((p["m"], 4096), 1.0, dtype=BF16, device="cuda") for p in (p0, p1))
    rc = N.lib().fdm_gemm_w8a8_fp8_x2(ctypes.byref(desc(p0, o0, N.EPI_GELU_TANH, 4096)), ctypes.byref(desc(p1, o1, N.EPI_GELU_TANH, 4096)), s)
    assert rc == 0, rc
    kern = N.lib().fdm_gemm_last_kernel()
    torch.cuda.synchronize()
    res["pair_tanh"] = [kern, hashlib.sha256(torch.cat([o0, o1]).cpu().view(torch.int16).numpy().tobytes()).hexdigest()]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
