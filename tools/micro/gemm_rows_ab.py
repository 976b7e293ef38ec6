#!/usr/bin/env python3
"""224-row against 256-row tiles of the 4-wave GEMM kernel (fdm_set_gemm_tile_rows), same process, warm operands, HIP events over 20 launches.

    python3 tools/micro/gemm_rows_ab.py > gpurun_out/<tag>_gemm_rows_ab.txt
"""
import statistics
import sys

import torch

sys.path.insert(0, ".")
from fastdm_amd import _native as N  # noqa: E402

FP8, BF16 = torch.float8_e4m3fn, torch.bfloat16
SHAPES = [   # name, dtype, M, N, K, epilogue
    ("flux single proj_out", "fp8", 8704, 3072, 15360, N.EPI_GATE_RESIDUAL), ("flux double ff2", "fp8", 8192, 3072, 12288, N.EPI_GATE_RESIDUAL),
    ("flux double to_out", "fp8", 8192, 3072, 3072, N.EPI_GATE_RESIDUAL), ("flux single qkv (p4 shape)", "fp8", 8704, 9216, 3072, N.EPI_NONE),
    ("flux 1024^2 qkv", "bf16", 4608, 9216, 3072, N.EPI_NONE), ("flux 1024^2 mlp", "bf16", 4608, 12288, 3072, N.EPI_NONE),
    ("flux 1024^2 proj_out", "bf16", 4608, 3072, 15360, N.EPI_GATE_RESIDUAL), ("qwen ff2", "int8", 6889, 3072, 12288, N.EPI_GATE_RESIDUAL),
    ("qwen to_out", "int8", 6889, 3072, 3072, N.EPI_GATE_RESIDUAL), ("sd3.5 ff2", "fp8", 4096, 2432 + 128, 9728 + 512, N.EPI_GATE_RESIDUAL),
]


def main():
    L, s = N.lib(), N.stream()
    g = torch.Generator().manual_seed(1)
    L.fdm_set_gemm_dispatch(1, -1, -1)
    print(f"{'shape':34s} {'dtype':5s} {'256 rows us':>12s} {'224 rows us':>12s} {'ratio':>6s} {'default':>10s}")
    for name, dt, m, n, k, epi in SHAPES:
        gr = epi == N.EPI_GATE_RESIDUAL
        bias = (torch.randn(n, generator=g) * 0.5).to(BF16).cuda()
        gate, r = torch.randn(1, n, generator=g).to(BF16).cuda(), torch.randn(m, n, generator=g).to(BF16).cuda()
        out = torch.empty((m, n), dtype=BF16, device="cuda")
        ga, ra = (N.ptr(gate), N.ptr(r)) if gr else (None, None)
        if dt == "fp8":
            a, b = (torch.randn(m, k, generator=g)).to(FP8).cuda(), (torch.randn(n, k, generator=g)).to(FP8).cuda()
            sa, sb = (torch.rand(m, 1, generator=g) * 0.02).cuda(), (torch.rand(n, 1, generator=g) * 0.02).cuda()
            fn = lambda: L.fdm_gemm_w8a8_fp8(N.ptr(out), N.ptr(a), N.ptr(b), N.ptr(sa), N.ptr(sb), N.ptr(bias), m, n, k, k, k, n, epi, ga, n, ra, m, s)  # noqa: E731
        elif dt == "int8":
            a = torch.randint(-128, 128, (m, k), generator=g, dtype=torch.int8).cuda()
            b = torch.randint(-128, 128, (n, k), generator=g, dtype=torch.int8).cuda()
            sa, sb = (torch.rand(m, 1, generator=g) * 0.01).cuda(), (torch.rand(n, 1, generator=g) * 0.01).cuda()
            fn = lambda: L.fdm_gemm_w8a8_int8(N.ptr(out), N.ptr(a), N.ptr(b), N.ptr(sa), N.ptr(sb), None, None, N.ptr(bias), m, n, k, k, k, n, epi, ga, n, ra, m, s)  # noqa: E731
        else:
            a, b = torch.randn(m, k, generator=g).to(BF16).cuda(), (torch.randn(n, k, generator=g) * k ** -0.5).to(BF16).cuda()
            fn = lambda: L.fdm_gemm_bf16(N.ptr(out), N.ptr(a), N.ptr(b), N.ptr(bias), m, n, k, 2 * k // 2, k, n, epi, ga, n, ra, m, s)  # noqa: E731
        t = {}
        ref = None
        for rows in (256, 224, 256, 224):
            L.fdm_set_gemm_tile_rows(rows)
            for _ in range(3):
                assert fn() == 0
            evs = []
            for _ in range(20):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            t.setdefault(rows, []).append(statistics.median(x.elapsed_time(y) for x, y in evs) * 1e3)
            if ref is None:
                ref = out.clone()
            else:
                assert torch.equal(ref, out), name
        L.fdm_set_gemm_tile_rows(-1)
        fn()
        auto = {1: "256", 5: "224"}.get(L.fdm_gemm_last_kernel(), str(L.fdm_gemm_last_kernel()))
        t256, t224 = min(t[256]), min(t[224])
        print(f"{name:34s} {dt:5s} {t256:12.1f} {t224:12.1f} {t224 / t256:6.3f} {auto:>10s}")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
