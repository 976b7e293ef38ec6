"""Time fdm_gemm_w8a8_int8_x2_ws with and without its split-K tail workspace (Qwen-Image shapes)."""
import ctypes, sys, torch
sys.path.insert(0, ".")
from fastdm_amd import _native as N

def run(M, K, Nn, M2=0, epi=N.EPI_NONE):
    dev = torch.device("cuda")
    def prob(m):
        a = torch.randint(-128, 128, (m, K), dtype=torch.int8, device=dev)
        b = torch.randint(-128, 128, (Nn, K), dtype=torch.int8, device=dev)
        sa, sb = torch.rand(m, 1, device=dev) * 0.01, torch.rand(Nn, 1, device=dev) * 0.01
        out = torch.empty((m, Nn), dtype=torch.bfloat16, device=dev)
        res = torch.randn((m, Nn), device=dev).to(torch.bfloat16)
        gate = torch.randn((1, Nn), device=dev).to(torch.bfloat16)
        d = N.GemmInt8Desc()
        d.out, d.a, d.b, d.scale_a, d.scale_b = N.ptr(out), N.ptr(a), N.ptr(b), N.ptr(sa), N.ptr(sb)
        d.M, d.N, d.K, d.lda, d.ldb, d.ldc = m, Nn, K, K, K, Nn
        d.epilogue = epi
        d.epi_gate, d.epi_gate_stride, d.epi_res, d.epi_rows_per_batch = N.ptr(gate), Nn, N.ptr(res), m
        return d, (a, b, sa, sb, out, res, gate)
    d0, keep0 = prob(M)
    d1, keep1 = prob(M2) if M2 else (None, None)
    ws = N.gemm_int8_workspace(dev)
    for tag, wp, wb in (("plain", None, 0), ("split", N.ptr(ws), ws.numel())):
        f = lambda: N.check(N.lib().fdm_gemm_w8a8_int8_x2_ws(ctypes.byref(d0), ctypes.byref(d1) if d1 else None, wp, wb, N.stream()), tag)
        for _ in range(5): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): f()
        e1.record(); torch.cuda.synchronize()
        tiles = -(-M // 256) * -(-Nn // 256) + (-(-M2 // 256) * -(-Nn // 256) if M2 else 0)
        print(f"M={M}+{M2} K={K} N={Nn} tiles={tiles} ({tiles/256:.2f} rounds) {tag}: {e0.elapsed_time(e1)/30*1e3:.1f} us")

run(6889, 12288, 3072, 128, N.EPI_GATE_RESIDUAL)
run(6889, 3072, 3072, 128, N.EPI_GATE_RESIDUAL)
run(6889, 3072, 12288, 128, N.EPI_GELU_TANH)
run(6889, 3072, 9216, 128)
run(8704, 3072, 9216, 512)
