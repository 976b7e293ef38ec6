"""Per-shape micro-benchmarks of the hot kernels at the FLUX C3 shapes (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastdm_amd import _native as N
from fastdm_amd.kernel import fused

FP8, BF16 = torch.float8_e4m3fn, torch.bfloat16
dev = "cuda"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def gemm_fp8(M, K, Nn, epi):
    a = (torch.randn(M, K, device=dev) * 0.5).to(BF16)
    w = (torch.randn(Nn, K, device=dev) * K ** -0.5).to(BF16)
    aq = fused.quantize_rows(a, None, FP8)
    wq = fused.quantize_rows(w, None, FP8)
    bias = torch.randn(Nn, device=dev).to(BF16)
    out = torch.empty(M, Nn, device=dev, dtype=BF16)
    gate = torch.randn(1, Nn, device=dev).to(BF16)
    res = torch.randn(M, Nn, device=dev).to(BF16)
    lib = N.lib()

    def run():
        N.check(lib.fdm_gemm_w8a8_fp8(out.data_ptr(), aq.q.data_ptr(), wq.q.data_ptr(), aq.scale.data_ptr(), wq.scale.data_ptr(), bias.data_ptr(),
                                      M, Nn, K, K, K, Nn, epi, gate.data_ptr(), Nn, res.data_ptr(), M, N.stream()), "gemm")
    ms = timeit(run)
    return ms, 2.0 * M * K * Nn / (ms * 1e-3) / 1e12


def attn(B, S, H, hd):
    q, k, v = (torch.randn(B * S, H * hd, device=dev).to(BF16) for _ in range(3))
    ms = timeit(lambda: fused.attention(q, k, v, B, H, hd, hd ** -0.5), iters=10)
    return ms, 4.0 * B * S * S * H * hd / (ms * 1e-3) / 1e12


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "gemm"):
        shapes = [("dbl qkv img", 8192, 3072, 9216, 0, 19), ("dbl qkv txt", 512, 3072, 9216, 0, 19), ("dbl to_out img", 8192, 3072, 3072, 3, 19),
                  ("dbl to_out txt", 512, 3072, 3072, 3, 19), ("dbl ff1 img", 8192, 3072, 12288, 1, 19), ("dbl ff1 txt", 512, 3072, 12288, 1, 19),
                  ("dbl ff2 img", 8192, 12288, 3072, 3, 19), ("dbl ff2 txt", 512, 12288, 3072, 3, 19), ("sgl qkv", 8704, 3072, 9216, 0, 38),
                  ("sgl mlp", 8704, 3072, 12288, 2, 38), ("sgl out", 8704, 15360, 3072, 3, 38)]
        tot = 0.0
        for name, M, K, Nn, epi, cnt in shapes:
            ms, tf = gemm_fp8(M, K, Nn, epi)
            tot += ms * cnt
            tiles = -(-M // 256) * -(-Nn // 256)
            print(f"{name:16s} M={M:5d} K={K:5d} N={Nn:5d} epi={epi} tiles={tiles:5d} ({tiles/256:5.2f} rounds)  {ms*1e3:8.1f} us  {tf:7.1f} TF/s   x{cnt} = {ms*cnt:6.2f} ms")
        print(f"GEMM total per step: {tot:.2f} ms")
    if which in ("all", "attn"):
        for (B, S, H, hd) in ((1, 8704, 24, 128), (1, 4608, 24, 128), (2, 4096, 24, 64)):
            ms, tf = attn(B, S, H, hd)
            print(f"attn B={B} S={S} H={H} hd={hd}: {ms*1e3:8.1f} us  {tf:7.1f} TF/s")
