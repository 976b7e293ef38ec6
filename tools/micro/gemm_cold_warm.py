#!/usr/bin/env python3
"""Warm / cold time of single GEMM launches of the shipped library on arbitrary shapes (random codes): the cold-weight sensitivity of each tile
kernel (tools/micro/gemm_lib_ab.py does the same on operands captured from the FLUX model, against hipBLASLt).
    python3 tools/micro/gemm_cold_warm.py [dtype:M,N,K[,epi] ...]      dtype fp8 | int8 | bf16, epi 0 plain | 3 gate + residual
warm: one (A, B) pair back to back; cold: 12 rotating weight buffers, 3 activation buffers, a 145 MB device copy between launches."""
import statistics
import sys

import torch

sys.path.insert(0, ".")
from fastdm_amd import _native as N  # noqa: E402

BF16, FP8 = torch.bfloat16, torch.float8_e4m3fn
KNAME = {0: "gemm_kernel", 1: "gemm_w4_kernel", 2: "gemm_p4_kernel", 4: "gemm_kernel half tiles"}
DEFAULT = ["fp8:8858,2432,2432,3", "fp8:8858,7296,2432", "fp8:8858,9728,2432", "fp8:8858,2432,9728,3",            # SD3.5-large (CFG batch 2)
           "int8:14034,9216,3072", "int8:14034,12288,3072", "int8:14034,3072,12288,3", "int8:14034,3072,3072,3",     # Qwen-Image (batch-2 CFG)
           "fp8:8192,640,640,3", "fp8:8192,5120,640", "fp8:2048,1280,1280,3", "fp8:2048,10240,1280"]                 # SDXL 1024^2


def main():
    dev = torch.device("cuda", 0)
    junk_a, junk_b = torch.empty(145 << 20, dtype=torch.uint8, device=dev), torch.empty(145 << 20, dtype=torch.uint8, device=dev)
    lib = N.lib()
    print(f"{'shape':34s} {'kernel':24s} {'warm us':>9s} {'frac':>6s} | {'cold us':>9s} {'frac':>6s} | cold / warm")
    for spec in (sys.argv[1:] or DEFAULT):
        dt, dims = spec.split(":")
        v = [int(x) for x in dims.split(",")]
        M, Nn, K = v[:3]
        epi = v[3] if len(v) > 3 else 0
        g = torch.Generator(device="cpu").manual_seed(M + Nn + K)
        nb = max(3, min(12, (300 << 20) // (Nn * K * (2 if dt == "bf16" else 1)) + 1))

        def mk(rows):
            if dt == "fp8":
                return (torch.randn(rows, K, generator=g) * 1.5).to(FP8).to(dev)
            if dt == "int8":
                return torch.randint(-128, 128, (rows, K), generator=g, dtype=torch.int8).to(dev)
            return torch.randn(rows, K, generator=g).to(BF16).to(dev)
        As, Bs = [mk(M) for _ in range(3)], [mk(Nn) for _ in range(nb)]
        sa, sb = (torch.rand(M, 1, generator=g) * 0.02 + 0.01).to(dev), (torch.rand(Nn, 1, generator=g) * 0.02 + 0.01).to(dev)
        bias = torch.randn(Nn, generator=g).to(BF16).to(dev)
        res = torch.randn(M, Nn, generator=g).to(BF16).to(dev) if epi == 3 else None
        gate = torch.randn(1, Nn, generator=g).to(BF16).to(dev) if epi == 3 else None
        out = torch.empty(M, Nn, dtype=BF16, device=dev)
        adj = [b.to(torch.int32).sum(dim=1, dtype=torch.int32).reshape(1, Nn).contiguous() for b in Bs] if dt == "int8" else None
        azp = torch.randint(-20, 20, (M, 1), generator=g, dtype=torch.int32).to(dev) if dt == "int8" else None

        def launch(i, cold):
            a, j = (As[i % 3], i % nb) if cold else (As[0], 0)
            b = Bs[j]
            s = N.stream()
            if dt == "fp8":
                rc = lib.fdm_gemm_w8a8_fp8(N.ptr(out), N.ptr(a), N.ptr(b), N.ptr(sa), N.ptr(sb), N.ptr(bias), M, Nn, K, K, K, Nn, epi, N.ptr(gate), Nn, N.ptr(res), M, s)
            elif dt == "int8":
                rc = lib.fdm_gemm_w8a8_int8(N.ptr(out), N.ptr(a), N.ptr(b), N.ptr(sa), N.ptr(sb), N.ptr(adj[j]), N.ptr(azp), N.ptr(bias), M, Nn, K, K, K, Nn, epi,
                                            N.ptr(gate), Nn, N.ptr(res), M, s)
            else:
                rc = lib.fdm_gemm_bf16(N.ptr(out), N.ptr(a), N.ptr(b), N.ptr(bias), M, Nn, K, K, K, Nn, epi, N.ptr(gate), Nn, N.ptr(res), M, s)
            N.check(rc, spec)

        def timed(cold, reps=16):
            for i in range(3):
                launch(i, cold)
            ev = []
            for i in range(reps):
                if cold:
                    junk_b.copy_(junk_a)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                launch(i + 3, cold)
                e1.record()
                ev.append((e0, e1))
            torch.cuda.synchronize()
            return statistics.median(a.elapsed_time(b) for a, b in ev) * 1e3
        w = statistics.median(timed(False) for _ in range(3))
        c = statistics.median(timed(True) for _ in range(3))
        kern = KNAME.get(int(lib.fdm_gemm_last_kernel()), "?")
        fl = 2.0 * M * Nn * K
        peak = 2.5e15 if dt == "bf16" else 5e15
        print(f"{spec:34s} {kern:24s} {w:9.1f} {fl / w / 1e-6 / peak:6.3f} | {c:9.1f} {fl / c / 1e-6 / peak:6.3f} | {c / w:.3f}")
        sys.stdout.flush()
        del As, Bs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
