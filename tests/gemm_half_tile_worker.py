"""Worker of tests/test_ops_gpu.py::test_gemm_half_tiles_*: a fixed list of GEMM launches on the 8-wave tile kernel (FDM_GEMM_KERNEL=w8) with
256 x 256 tiles or 256 x 128 half tiles (FDM_GEMM_HALF_TILES = 0 | 1, read once per process); prints one JSON line: per case a sha256 of the
bf16 output bytes.  A wave runs the same MFMAs in the same K order in both forms, so every bit must agree."""
import ctypes
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fastdm_amd import _native as N  # noqa: E402

BF16, FP8 = torch.bfloat16, torch.float8_e4m3fn


def sha(t):
    torch.cuda.synchronize()
    return hashlib.sha256(t.cpu().view(torch.int16).numpy().tobytes()).hexdigest()


W8 = 4 if os.environ.get("FDM_GEMM_HALF_TILES") == "1" else 0      # FDM_GEMM_KERNEL_W8_HALF / FDM_GEMM_KERNEL_W8: the kernel id proves which form ran


def main():
    g = torch.Generator().manual_seed(4321)
    res = {}
    s = N.stream()
    L = N.lib()
    # fp8: SD3.5 / SDXL shapes and ragged everything; epilogues none / tanh / erf / gate + residual
    for name, (m, k, n, epi) in {
        "fp8_sd35_out": (1178, 2432, 2432, N.EPI_GATE_RESIDUAL), "fp8_sdxl_l2": (2048, 1280, 1280, N.EPI_NONE), "fp8_sdxl_l1": (1000, 640, 640, N.EPI_GELU_ERF),
        "fp8_ragged": (301, 208, 392, N.EPI_GELU_TANH), "fp8_kv_154": (154, 2048, 2560, N.EPI_NONE), "fp8_one_col_tile": (513, 1024, 120, N.EPI_GATE_RESIDUAL),
    }.items():
        a = (torch.randn(m, k, generator=g) * 1.5).to(FP8).cuda()
        b = (torch.randn(n, k, generator=g) * 1.5).to(FP8).cuda()
        sa, sb = (torch.rand(m, 1, generator=g) * 0.02 + 0.002).cuda(), (torch.rand(n, 1, generator=g) * 0.02 + 0.002).cuda()
        bias = (torch.randn(n, generator=g) * 0.5).to(BF16).cuda()
        gate, r = (torch.randn(2, n, generator=g)).to(BF16).cuda(), torch.randn(m, n + 8, generator=g).to(BF16).cuda()      # (the residual shares the output's row stride)
        out = torch.full((m, n + 8), 3.0, dtype=BF16, device="cuda")
        rpb = (m + 1) // 2
        rc = L.fdm_gemm_w8a8_fp8(N.ptr(out), N.ptr(a), N.ptr(b), N.ptr(sa), N.ptr(sb), N.ptr(bias), m, n, k, k, k, n + 8, epi,
                                 N.ptr(gate) if epi == N.EPI_GATE_RESIDUAL else None, n, N.ptr(r) if epi == N.EPI_GATE_RESIDUAL else None, rpb, s)
        assert rc == 0 and L.fdm_gemm_last_kernel() == W8, (name, rc, L.fdm_gemm_last_kernel())
        assert (out[:, n:] == 3.0).all(), name
        res[name] = sha(out)
    # int8 with asymmetric zero points
    for name, (m, k, n, epi) in {"i8_sdxl": (2048, 1280, 1280, N.EPI_NONE), "i8_ragged": (333, 320, 648, N.EPI_GATE_RESIDUAL)}.items():
        a = torch.randint(-128, 128, (m, k), generator=g, dtype=torch.int8).cuda()
        b = torch.randint(-128, 128, (n, k), generator=g, dtype=torch.int8).cuda()
        sa, sb = (torch.rand(m, 1, generator=g) * 0.01).cuda(), (torch.rand(n, 1, generator=g) * 0.01).cuda()
        azp = torch.randint(-20, 20, (m, 1), generator=g, dtype=torch.int32).cuda()
        adj = b.to(torch.int32).sum(dim=1, dtype=torch.int32).reshape(1, n).contiguous()
        bias = (torch.randn(n, generator=g) * 0.5).to(BF16).cuda()
        gate, r = torch.randn(1, n, generator=g).to(BF16).cuda(), torch.randn(m, n, generator=g).to(BF16).cuda()
        out = torch.empty((m, n), dtype=BF16, device="cuda")
        rc = L.fdm_gemm_w8a8_int8(N.ptr(out), N.ptr(a), N.ptr(b), N.ptr(sa), N.ptr(sb), N.ptr(adj), N.ptr(azp), N.ptr(bias), m, n, k, k, k, n, epi,
                                  N.ptr(gate) if epi == N.EPI_GATE_RESIDUAL else None, n, N.ptr(r) if epi == N.EPI_GATE_RESIDUAL else None, m, s)
        assert rc == 0 and L.fdm_gemm_last_kernel() == W8, (name, rc)
        res[name] = sha(out)
    # bf16
    for name, (m, k, n, epi) in {"bf16_small": (300, 72, 528, N.EPI_GELU_TANH), "bf16_sdxl": (2048, 640, 1280, N.EPI_NONE)}.items():
        a, w = (torch.randn(m, k, generator=g)).to(BF16).cuda(), (torch.randn(n, k, generator=g) * k ** -0.5).to(BF16).cuda()
        bias = (torch.randn(n, generator=g) * 0.1).to(BF16).cuda()
        out = torch.empty((m, n), dtype=BF16, device="cuda")
        rc = L.fdm_gemm_bf16(N.ptr(out), N.ptr(a), N.ptr(w), N.ptr(bias), m, n, k, k, k, n, epi, None, 0, None, 0, s)
        assert rc == 0 and L.fdm_gemm_last_kernel() == W8, (name, rc)
        res[name] = sha(out)
    # two problems in one launch, outputs as three planes whose width (152) straddles the half tiles
    C, kk, m_img, m_txt = 152, 192, 300, 37
    planes = torch.full((3, m_img + m_txt, C), 5.0, dtype=BF16, device="cuda")
    ds = []
    keep = []
    for m, off in ((m_img, m_txt), (m_txt, 0)):
        a = (torch.randn(m, kk, generator=g) * 1.5).to(FP8).cuda()
        b = (torch.randn(3 * C, kk, generator=g) * 1.5).to(FP8).cuda()
        sa, sb = (torch.rand(m, 1, generator=g) * 0.02 + 0.002).cuda(), (torch.rand(3 * C, 1, generator=g) * 0.02 + 0.002).cuda()
        bias = (torch.randn(3 * C, generator=g) * 0.5).to(BF16).cuda()
        keep += [a, b, sa, sb, bias]
        d = N.GemmFp8Desc()
        d.out, d.a, d.b, d.scale_a, d.scale_b, d.bias = N.ptr(planes[0, off:]), N.ptr(a), N.ptr(b), N.ptr(sa), N.ptr(sb), N.ptr(bias)
        d.M, d.N, d.K, d.lda, d.ldb, d.ldc = m, 3 * C, kk, kk, kk, C
        d.epilogue = N.EPI_NONE
        d.epi_gate, d.epi_gate_stride, d.epi_res, d.epi_rows_per_batch = None, 0, None, 0
        d.out_plane_cols, d.out_plane_stride = C, (m_img + m_txt) * C
        ds.append(d)
    rc = L.fdm_gemm_w8a8_fp8_x2(ctypes.byref(ds[0]), ctypes.byref(ds[1]), s)
    assert rc == 0 and L.fdm_gemm_last_kernel() == W8, rc
    res["pair_planes"] = sha(planes)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
