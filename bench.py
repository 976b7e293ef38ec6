#!/usr/bin/env python3
"""Headline benchmark: FLUX.1-dev denoise step, fp8 W8A8, 1024x2048 (S = 8192 image + 512 text
tokens), 25-step flow-match schedule, synthetic weights and prompts (BASELINE.json configs[2]).

    python bench.py --gpus 1 --steps 25 --warmup 3
    python bench.py --gpus N --steps K --warmup W          # plain command: spawns one fresh child process per GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path: one transformer forward (19 double + 38 single blocks, full
width) plus the Euler latent update, with every input already resident in HBM.  With N GPUs each rank
denoises its own image (batch sharded, weights replicated, no per-step exchange) and the ranks meet
once in an RCCL all-gather of the final latents; `value` is the whole-job denoise rate
(N * K / max-over-ranks time).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_TFLOPS = {"fp8": 5000.0, "bf16": 2500.0, "int8": 5000.0,   # dense MFMA, /opt/skills/guides/MI355X_MICROARCH.md
               # --attn-pv-fp8: half of the attention flops (QK^T) at the bf16 rate, half (P.V) at the fp8 rate: 1 / (0.5 / 2500 + 0.5 / 5000)
               "bf16qk+fp8pv": 3333.3}      # (--attn-fp8: all of it at the fp8 rate)
IMAGE_STEPS = 25
# timing-span name -> HIP kernel it launches (the name rocprofv3 reports; profiles/*_summary.txt)
HIP_KERNEL = {"attn_fwd_bf16": "fdm::attn_fwd64_kernel<128, false>", "attn_sparse_bf16": "fdm::attn_fwd64_kernel<128, true>",
              "attn_fwd_bf16_hd64": "fdm::attn_fwd_kernel<64, 512, 64>",
              "attn_fwd_pvfp8": "fdm::attn_fwd64_pv8_kernel<128, false>", "attn_fwd_fp8": "fdm::attn_fwd64_qk8_kernel<128, false>",
              "attn_sparse_pvfp8": "fdm::attn_fwd64_pv8_kernel<128, true>", "attn_sparse_fp8": "fdm::attn_fwd64_qk8_kernel<128, true>",
              "conv3x3_bf16": "fdm::gemm_kernel<2, *, *, true> (3 x 3 convolution as an implicit GEMM)"}


def span_kernel(span: str, by_kernel=None) -> str:
    """rocprofv3 name of the kernel that carried most of a span's time.  GEMM spans are dispatched per launch (gemm.hip use_p4 / use_w4):
    the name comes from what fdm_gemm_last_kernel() reported for the span's launches (timing.Recorder `by_kernel`), with the template
    arguments the span implies (data type, gate + residual), never from a static table."""
    if span in HIP_KERNEL:
        return HIP_KERNEL[span]
    if not span.startswith("gemm") or not by_kernel:
        return span
    top = max(by_kernel, key=lambda k: by_kernel[k]["total_ms"])
    dt = 2 if "bf16" in span else (1 if "int8" in span else 0)
    res = "true" if span.endswith("_res") else "false"
    if top == "fdm::gemm_p4_kernel":
        return "fdm::gemm_p4_kernel<0>"            # plain epilogue carries most p4 launches (<1> = tanh-GELU, <2> = erf-GELU)
    if top == "fdm::gemm_kernel":
        return f"{top}<{dt}, {res}>"
    if top.startswith("fdm::gemm_w4_kernel"):      # <data type, gate + residual, 16-row blocks per wave: 8 = 256-row tiles, 7 = 224-row tiles>
        return f"fdm::gemm_w4_kernel<{dt}, {res}, {7 if '224' in top else 8}>"
    return top


def pmc_traffic(kernel):
    """(HBM bytes per launch of `kernel`, provenance) from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json,
    made by tools/gpu_evidence.sh: FETCH_SIZE and WRITE_SIZE in separate passes -- counters cannot be collected inside a
    timed run -- FETCH_SIZE doubled as the gfx950 guide prescribes).  The file records the digest of the kernel sources it
    was taken with: if the running library was built from other sources the number is stale and None is reported."""
    try:
        from fastdm_amd import _native
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            d = json.load(f)
        if d.get("srcdigest") != _native.current_source_digest():
            return None, f"profiles/pmc_traffic.json is stale (taken at source digest {str(d.get('srcdigest'))[:12]}, library is {_native.current_source_digest()[:12]})"
        return d["kernels"][kernel]["hbm_bytes_per_launch"], f"profiles/pmc_traffic.json ({d.get('note', '')[:60]}; source digest {d['srcdigest'][:12]} = this library)"
    except Exception as e:  # noqa: BLE001
        return None, f"no PMC record for this kernel ({type(e).__name__})"



class _reference_cpu_ops:
    """While the CPU baseline is timed the oracle's attention op is the reference torch backend's own call --
    ``F.scaled_dot_product_attention`` on bf16 [B, H, S, D] views (/root/reference/fastdm/kernel/torch/attention.py:34-43) --
    and its fp32 int8 accumulate instead of the oracle's per-head fp32 softmax loop and float64 int8 sums (oracle/ops.py, which exists to pin the arithmetic, not to be fast:
    it is >= 5x slower and made the round-3 baseline pessimistic).  With this substitution the oracle is bit-exact against
    the reference's fixtures (tests/test_oracle_golden.py::test_sdxl_tiny_unet_matches_reference)."""

    def __enter__(self):
        from oracle import ops as OPS
        import torch.nn.functional as F

        def torch_sdpa(query, key, value, hq, hk, hd, is_causal=False, scale=None, **_):
            b, s, c = query.shape
            q, k, v = (x.view(b, x.size(1), h, hd).transpose(1, 2) for x, h in ((query, hq), (key, hk), (value, hk)))
            return F.scaled_dot_product_attention(q, k, v, dropout_p=0.0, is_causal=is_causal, scale=scale).transpose(1, 2).contiguous().view(b, s, c)

        def int8_matmul_f32(a, b, scale_a, scale_b, out_dtype, azp_adj, azp, bias=None):
            # the reference's own accumulate (/root/reference/fastdm/kernel/torch/matrixmul.py:67-74): fp32 products, where the
            # oracle's checker form takes an exact float64 sum (about twice the CPU time)
            acc = a.float() @ b.float() - azp.float() @ azp_adj.float()
            out = (acc * (scale_a.float() * scale_b.reshape(1, -1).float())).to(out_dtype)
            return out + bias if bias is not None else out
        self._ops, self._saved = OPS, (OPS.scaled_dot_product_attention, OPS.int8_matmul)
        OPS.scaled_dot_product_attention = torch_sdpa
        OPS.int8_matmul = int8_matmul_f32
        return self

    def __exit__(self, *exc):
        self._ops.scaled_dot_product_attention, self._ops.int8_matmul = self._saved
        return False


def _best_of(fn, reps=2):
    """One untimed warm-up call (page-in, thread pool, oneDNN primitive caches), then the best of `reps` timed calls -- at the default
    thread count and, on hosts with more than 32 threads, once more at 32 (a 128-thread oneDNN matmul of these sizes ran SLOWER than
    8 threads of the build container: 7.4 s against 3.9 s for the same double block); the faster setting is kept and reported."""
    best, out, best_threads = None, None, torch.get_num_threads()
    default = torch.get_num_threads()
    for threads in ([default, 32] if default > 32 else [default]):
        torch.set_num_threads(threads)
        fn()
        for _ in range(reps if threads == default else 1):
            t0 = time.perf_counter()
            out = fn()
            dt = time.perf_counter() - t0
            if best is None or dt < best:
                best, best_threads = dt, threads
    torch.set_num_threads(best_threads)      # later legs start from the better setting
    return best, out


def _cpu_desc():
    return {"cores": torch.get_num_threads(), "nproc": os.cpu_count()}


def cpu_baseline_leg(height, width, n_txt, quant_tag):
    """The oracle (CPU restatement of the reference's torch backend) timed on this box's host cores
    on a bounded sample: ONE double-stream + ONE single-stream FLUX block at full width (S = 8704),
    extrapolated x19 / x38 to a denoise step (a full step is 165 TFLOP)."""
    from oracle import flux as OF
    cfg = OF.FluxConfig(num_layers=1, num_single_layers=1)
    quant = {"fp8": torch.float8_e4m3fn, "int8": torch.int8, "bf16": None}[quant_tag]
    sd = {}
    for name, shape, kind in OF.param_specs(cfg):
        if name.startswith(("transformer_blocks.0.", "single_transformer_blocks.0.")):
            sd[name] = OF.synth_tensor(name, shape, kind, 0)
    dbl = OF.FluxTransformerBlock(sd, 0, cfg, quant)
    sgl = OF.FluxSingleTransformerBlock(sd, 0, cfg, quant)
    n_img = (height // 16) * (width // 16)
    g = torch.Generator().manual_seed(0)
    hidden = torch.randn(1, n_img, cfg.inner_dim, generator=g).to(torch.bfloat16)
    enc = torch.randn(1, n_txt, cfg.inner_dim, generator=g).to(torch.bfloat16)
    temb = torch.randn(1, cfg.inner_dim, generator=g).to(torch.bfloat16)
    inp = OF.synth_inputs(cfg, 1, 2, 2, 4)
    from oracle import layers as OL
    ids = torch.zeros(n_img, 3)
    rope = OL.flux_rope_cache(torch.zeros(n_txt, 3), ids, cfg.axes_dims_rope)
    del inp
    with torch.no_grad(), _reference_cpu_ops():
        t_dbl, (enc2, hid2) = _best_of(lambda: dbl(hidden, enc, temb, rope))
        joint = torch.cat([enc2, hid2], dim=1)
        t_sgl, _ = _best_of(lambda: sgl(joint, temb, rope))
    step_s = 19 * t_dbl + 38 * t_sgl
    return {"value": 1.0 / step_s, "unit": "it/s", **_cpu_desc(), "kind": "port",
            "sample": f"oracle blocks with the reference torch backend's attention call (bf16 F.scaled_dot_product_attention), {quant_tag}: "
                      f"1 double block {t_dbl:.2f}s + 1 single block {t_sgl:.2f}s at full width S={n_img + n_txt} (one warm-up, best of 2, "
                      f"{torch.get_num_threads()} threads of {os.cpu_count()} logical CPUs), extrapolated x19/x38 to one denoise step ({step_s:.0f}s)"}


class DeviceState:
    """What THIS rank's GPU was doing around the timed steps, measured OUTSIDE the timed window (rocm-smi
    forked four times a second inside it, and its DPM level is not the clock an MFMA-dense kernel runs at):
      * `mfma_clock_mhz`: the shader clock the chip sustains under an fp8-MFMA + LDS-read loop on random data, from in-kernel
        delta(s_memtime) / delta(s_memrealtime) x 100 MHz stamps of the library's probe kernel (fdm_clock_probe), median over its
        256 workgroups, launched right after the timed steps while the package is still at its loaded operating point;
      * one `rocm-smi --showclocks --showpower` reading before the warm-up and one right after the probe (DPM level / package power:
        context only -- MI355X_MICROARCH.md: pp_dpm_sclk reads up to ~10 % above the in-kernel clock of an MFMA-dense loop)."""

    def __init__(self, device_index: int):
        self.dev, self.smi = device_index, {}
        self.probe = None

    @staticmethod
    def smi_usable() -> bool:
        """Not under a profiler (rocprofv3's tool library would ride into the child processes), and rocm-smi on PATH."""
        import shutil
        if any(k.startswith(("ROCP", "HSA_TOOLS")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "").lower():
            return False
        return shutil.which("rocm-smi") is not None

    @staticmethod
    def parse(out: str):
        """(shader clock MHz, package power W | None) from `rocm-smi --showclocks --showpower` text, None without a clock line."""
        import re
        sclk = re.search(r"sclk clock level:[^(]*\((\d+)Mhz\)", out)
        pw = re.search(r"Power \(W\):\s*([0-9.]+)", out)
        return (int(sclk.group(1)), float(pw.group(1)) if pw else None) if sclk else None

    def sample_smi(self, tag: str):
        import subprocess
        if not self.smi_usable():
            return
        try:
            out = subprocess.run(["rocm-smi", "-d", str(self.dev), "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            smp = self.parse(out)
            if smp:
                self.smi[tag] = {"sclk_dpm_mhz": smp[0], "package_power_w": smp[1]}
        except Exception:  # noqa: BLE001
            pass

    def run_probe(self, ms: float = 40.0):
        from fastdm_amd.kernel import fused
        try:
            self.probe = fused.clock_probe(torch.device("cuda", self.dev), ms)
        except Exception as e:  # noqa: BLE001
            self.probe = {"error": str(e)}

    def sample_in_step(self, one_step, seconds: float = 3.0):
        """package power / DPM clock WHILE the step runs -- an extra, UNTIMED repetition of the steps (about `seconds`
        of them, after the clock has stopped) with rocm-smi polled back to back from a thread.  Reported as `in_step`: what the package
        draws under exactly this kernel sequence, which the MFMA + LDS probe behind the steps does not show."""
        import subprocess
        import threading
        if not self.smi_usable():
            return
        samples, stop = [], threading.Event()

        def poll():
            while not stop.is_set():
                try:
                    out = subprocess.run(["rocm-smi", "-d", str(self.dev), "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
                    smp = self.parse(out)
                    if smp:
                        samples.append(smp)
                except Exception:  # noqa: BLE001
                    return

        torch.cuda.synchronize()
        th = threading.Thread(target=poll, daemon=True)
        t0 = time.perf_counter()
        th.start()
        n = 0
        while time.perf_counter() - t0 < seconds:
            one_step()
            torch.cuda.synchronize()
            n += 1
        dt = time.perf_counter() - t0
        stop.set()
        th.join(timeout=15)
        samples = samples[1:] if len(samples) > 2 else samples      # the first poll may have started before the GPU was busy
        if samples:
            pw = sorted(p for _, p in samples if p is not None)
            ck = sorted(c for c, _ in samples)
            self.smi["in_step"] = {"samples": len(samples), "steps": n, "ms_per_step_untimed": round(dt / max(n, 1) * 1e3, 2),
                                   "package_power_w": {"median": pw[len(pw) // 2], "min": pw[0], "max": pw[-1]} if pw else None,
                                   "sclk_dpm_mhz": {"median": ck[len(ck) // 2], "min": ck[0], "max": ck[-1]}}

    def summary(self):
        return {"mfma_clock_mhz": self.probe, "rocm_smi": self.smi or None,
                "source": "fdm_clock_probe (s_memtime / s_memrealtime stamps of an fp8 MFMA + LDS loop) right AFTER the timed steps; rocm-smi "
                          "once before the warm-up and once after the probe; `in_step`: rocm-smi polled during an extra UNTIMED repetition of the "
                          "steps (a synchronize per step, so its ms_per_step is not the benchmark's); nothing is sampled inside the timed window"}


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh child processes (one per GPU, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set) BEFORE anything in this process touches the GPU, relay rank 0's JSON line, and return
    non-zero if any rank failed.  The parent never initialises HIP and never re-execs itself; it polls ALL ranks, terminates
    the others when one dies (a dead rank otherwise leaves its siblings in the rendezvous until the RCCL timeout) and gives up
    after FDM_BENCH_TIMEOUT seconds (default 1500)."""
    from fastdm_amd.launch import spawn_self
    return spawn_self(__file__, sys.argv[1:], n, timeout=float(os.environ.get("FDM_BENCH_TIMEOUT", "1500")), same_device="--same-device" in sys.argv[1:])


def cpu_baseline_qwen(height, width, n_txt, quant_tag):
    """Qwen-Image: ONE double-stream block of the oracle at full width (S = image tokens + text tokens), x 60 blocks x 2 forwards
    (true CFG) per denoise step."""
    from oracle import qwen as OQ
    cfg = OQ.QwenConfig(num_layers=1)
    quant = {"fp8": torch.float8_e4m3fn, "int8": torch.int8, "bf16": None}[quant_tag]
    sd = {n: OQ.synth_tensor(n, sh, k, 0) for n, sh, k in OQ.param_specs(cfg) if n.startswith("transformer_blocks.0.")}
    blk = OQ.QwenImageTransformerBlock(sd, 0, cfg, quant)
    h, w = height // 16, width // 16
    D = cfg.num_attention_heads * cfg.attention_head_dim
    g = torch.Generator().manual_seed(0)
    hidden = torch.randn(1, h * w, D, generator=g).to(torch.bfloat16)
    enc = torch.randn(1, n_txt, D, generator=g).to(torch.bfloat16)
    temb = torch.randn(1, D, generator=g).to(torch.bfloat16)
    rope = OQ.qwen_rope_cache([(1, h, w)], n_txt, cfg.axes_dims_rope)
    with torch.no_grad(), _reference_cpu_ops():
        dt, _ = _best_of(lambda: blk(hidden, enc, temb, rope))
    step_s = dt * 60 * 2
    return {"value": 1.0 / step_s, "unit": "it/s", **_cpu_desc(), "kind": "port",
            "sample": f"oracle block with the reference torch backend's attention call (bf16 F.scaled_dot_product_attention), {quant_tag}: "
                      f"1 Qwen-Image block {dt:.2f}s at full width S={h * w + n_txt} (one warm-up, best of 2, {torch.get_num_threads()} threads of "
                      f"{os.cpu_count()} logical CPUs), extrapolated x60 blocks x2 CFG forwards to one denoise step ({step_s:.0f}s)"}


def cpu_baseline_sdxl(height, width, n_txt, quant_tag):
    """SDXL UNet (BASELINE.json configs[0] is its CPU path): ONE transformer layer of the oracle at each of the two attention
    levels (640 / 1280 channels, CFG batch 2), x the 10 / 60 layers of the UNet; the 52 convolutions and group norms (library calls
    on both sides) are not in the sample."""
    from oracle import sdxl as OS
    cfg = OS.SDXLConfig()
    quant = {"fp8": torch.float8_e4m3fn, "int8": torch.int8, "bf16": None}[quant_tag]
    g = torch.Generator().manual_seed(0)
    enc = torch.randn(2 * n_txt, cfg.cross_attention_dim, generator=g).to(torch.bfloat16).reshape(2, n_txt, -1)
    times = []
    for c, tokens, name in ((cfg.block_out_channels[1], (height // 16) * (width // 16), "down_blocks.1.attentions.0"),
                            (cfg.block_out_channels[2], (height // 32) * (width // 32), "down_blocks.2.attentions.0")):
        sd = {n: OS.synth_tensor(n, sh, k, 0) for n, sh, k in OS._transformer(name, c, 1, cfg.cross_attention_dim)}
        blk = OS.BasicTransformerBlock(sd, f"{name}.transformer_blocks.0", c, cfg.head_dim, quant)
        x = torch.randn(2 * tokens, c, generator=g).to(torch.bfloat16)
        with torch.no_grad(), _reference_cpu_ops():
            times.append(_best_of(lambda: blk(x, enc, 2))[0])
    n1 = cfg.transformer_layers[0] * 5                      # 2 down + 3 up attentions at level 1
    n2 = cfg.transformer_layers[1] * 5 + cfg.mid_layers     # 2 down + 3 up at level 2, + the mid block
    step_s = times[0] * n1 + times[1] * n2
    return {"value": 1.0 / step_s, "unit": "it/s", **_cpu_desc(), "kind": "port",
            "sample": f"oracle layers with the reference torch backend's attention call (bf16 F.scaled_dot_product_attention), {quant_tag}: "
                      f"1 transformer layer at 640 channels {times[0]:.2f}s + 1 at 1280 channels {times[1]:.2f}s, CFG batch 2, {height}x{width} "
                      f"(one warm-up, best of 2, {torch.get_num_threads()} threads of {os.cpu_count()} logical CPUs), extrapolated x{n1}/x{n2} layers "
                      f"to one UNet forward ({step_s:.1f}s; convolutions / group norms not included)"}


_RESULT_FD = None


def claim_stdout():
    """The contract is ONE JSON line on stdout.  Libraries write there too -- RCCL prints its version banner through C stdio,
    which a pipe delivers at process exit, i.e. BEHIND the JSON line -- so file descriptor 1 is pointed at stderr for the
    whole run and the result line goes to a private duplicate of the real stdout."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)


def emit_result(line: str):
    sys.stdout.flush()
    if _RESULT_FD is None:
        print(line, flush=True)
    else:
        os.write(_RESULT_FD, (line + "\n").encode())


ARCH_NAME = {"flux": "FLUX.1-dev", "qwen": "Qwen-Image", "wan": "Wan2.2-T2V-A14B", "sd35": "SD3.5-large", "sdxl": "SDXL-base-1.0"}


def main():
    global IMAGE_STEPS
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 25 (flux, qwen) / 3 (wan)")
    ap.add_argument("--warmup", type=int, default=None, help="default 3 (flux, qwen) / 1 (wan)")
    ap.add_argument("--model", choices=["flux", "qwen", "wan", "sd35", "sdxl"], default="flux",
                    help="flux = the headline workload; qwen = Qwen-Image (BASELINE.json configs[3]) and wan = Wan2.2-T2V-A14B "
                         "(configs[4]): a step is a cond + uncond forward pair (CFG)")
    ap.add_argument("--frames", type=int, default=81, help="wan: video frames")
    ap.add_argument("--sparse-config", default=None, help="wan: radial block-sparse attention JSON (examples/sparse/radial_attn_wan.json)")
    ap.add_argument("--start-step", type=int, default=0, help="schedule index of the first timed step (wan: >= dense_steps exercises the sparse steps)")
    ap.add_argument("--layers", type=int, default=None, help="wan / qwen: number of blocks (default 40 / 60; a reduced depth is reported in the workload string)")
    ap.add_argument("--height", type=int, default=None, help="default 1024 (flux) / 1328 (qwen)")
    ap.add_argument("--width", type=int, default=None, help="default 2048 (flux) / 1328 (qwen)")
    ap.add_argument("--max-seq-len", type=int, default=None, help="text tokens: default 512 (flux) / 128 (qwen)")
    ap.add_argument("--quant", choices=["fp8", "int8", "bf16"], default=None, help="default fp8 (flux) / int8 (qwen)")
    ap.add_argument("--cache-config", default=None, help="TeaCache JSON (changes the metric: reported as a different workload)")
    ap.add_argument("--cache-threshold", type=float, default=None,
                    help="override the cache config's threshold (random synthetic weights move the TeaCache probe ~0.3 per step, far more than a "
                         "trained model: the shipped 0.2 never skips on them; reported in the workload string)")
    ap.add_argument("--cache-negative", action="store_true", help="set the cache config's negtive_cache flag: separate cache entries for the "
                    "conditional / unconditional forward of a true-CFG step (the shipped teacache_qwenimage.json has it off)")
    ap.add_argument("--batch-cfg", action="store_true", help="qwen: conditional + unconditional prompt as one batch-2 forward per step (the default since round 6 "
                    "whenever it applies: no --cfg-pair, no step cache; bit-identical to two forwards)")
    ap.add_argument("--no-batch-cfg", action="store_true", help="qwen: two batch-1 forwards per step")
    ap.add_argument("--graph", action="store_true", help="sdxl: replay the UNet forward from a HIP graph (the step is launch-bound: ~2000 small kernels)")
    ap.add_argument("--nhwc", action="store_true", help="sdxl: channels-last activations + HIP convolutions (the default since round 6; kept for old command lines)")
    ap.add_argument("--nchw", action="store_true", help="sdxl: NCHW activations, MIOpen convolutions (the round-5 path)")
    ap.add_argument("--cfg-pair", action="store_true", help="qwen / wan / sd35 / sdxl with --gpus 2k: ranks (2i, 2i+1) share image i -- one runs the "
                    "conditional, the other the unconditional forward, one all-gather of the noise predictions per step (strong scaling of a step)")
    ap.add_argument("--images-per-rank", type=int, default=1, help="images each rank (or CFG pair) denoises, one after the other inside every step, as the "
                    "reference's per-process prompt loop does (examples/evaluation/clip_score.py:346-365); BASELINE.json configs[3] = 8 images on "
                    "4 CFG pairs: --gpus 8 --cfg-pair --images-per-rank 2.  A step still means one denoise step of EVERY image in flight")
    ap.add_argument("--attn-pv-fp8", action="store_true", help="OPT-IN variant, reported beside -- never instead of -- the headline: dense head_dim-128 "
                    "attention with e4m3 P and V on the fp8 matrix cores (fdm_attn_fwd_pvfp8: the reference's SageAttention class); the line "
                    "carries \"variant\": \"attn_pv_fp8\" and says so in dtype / workload")
    ap.add_argument("--attn-fp8", action="store_true", help="OPT-IN variant like --attn-pv-fp8 with QK^T on the fp8 matrix cores as well (fdm_attn_fwd_fp8): "
                    "\"variant\": \"attn_fp8\"")
    ap.add_argument("--same-device", action="store_true", help="PROTOCOL TEST ONLY: every rank on cuda:0 over gloo (a 1-GPU box exercising the N-rank "
                    "sharding / CFG-pair exchange / final gather with real HIP forwards); the line says so and is not a scaling number")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the per-kernel HIP-event spans")
    ap.add_argument("--kernel-timing-every", type=int, default=0, help="HIP-event spans on every N-th timed step (default: 5 when --steps >= 10, else 1)")
    ap.add_argument("--no-device-state", action="store_true", help="skip the MFMA clock probe / rocm-smi readings taken around (never inside) the timed steps")
    ap.add_argument("--allow-build-flags", action="store_true", help="A/B only: accept a library built with extra hipcc flags (FDM_LIB_DIR=tools/micro/_ab/<name>); "
                    "the line then carries lib_flags and \"experimental\": true and is not a headline number")
    args = ap.parse_args()
    dflt = {"flux": (1024, 2048, 512, "fp8"), "qwen": (1328, 1328, 128, "int8"), "wan": (768, 1280, 512, "fp8"), "sd35": (1024, 1024, 333, "fp8"), "sdxl": (1024, 1024, 77, "fp8")}[args.model]
    args.height, args.width = args.height or dflt[0], args.width or dflt[1]
    args.max_seq_len, args.quant = args.max_seq_len or dflt[2], args.quant or dflt[3]
    if args.model == "qwen":
        args.batch_cfg = (args.batch_cfg or not args.no_batch_cfg) and not args.cfg_pair and not args.cache_config
    if args.graph:
        args.no_kernel_timing = True       # HIP-event spans cannot be recorded inside a graph replay
    args.steps = args.steps if args.steps is not None else (3 if args.model == "wan" else 25)
    args.warmup = args.warmup if args.warmup is not None else (1 if args.model == "wan" else 3)

    for var in ("FDM_GEMM_DEBUG", "FDM_FP8_SW_CVT"):
        if os.environ.get(var):
            sys.exit(f"bench.py: refusing to run with {var} set (a bring-up switch of debug builds; it must not shape a benchmark number)")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))       # nothing above this line has touched the GPU
    if args.cfg_pair and (args.model == "flux" or args.gpus % 2):
        sys.exit("--cfg-pair needs a two-forward model (qwen / wan / sd35 / sdxl) and an even --gpus")
    if args.images_per_rank < 1 or (args.images_per_rank > 1 and (args.cache_config or args.graph)):
        sys.exit("--images-per-rank k > 1 runs the images one after the other through ONE model: a step cache / a captured graph holds the state "
                 "of one image only (use one process per image for those)")

    claim_stdout()
    if args.attn_pv_fp8 or args.attn_fp8:
        from fastdm_amd.kernel.utils import set_attention_pv_fp8
        set_attention_pv_fp8(2 if args.attn_fp8 else 1)
    elif os.environ.get("FDM_ATT_PV_FP8", "0") not in ("", "0"):
        sys.exit("bench.py: FDM_ATT_PV_FP8 is set: ask for a quantised-attention variant with --attn-pv-fp8 / --attn-fp8 (the default line is the bf16 headline)")
    from fastdm_amd import _native
    prov = _native.provenance()          # what the LOADED binary says it was built from (fdm_source_digest / fdm_build_flags)
    if prov["lib_flags"] and not args.allow_build_flags:
        sys.exit(f"bench.py: refusing to benchmark a library built with extra flags '{prov['lib_flags']}' ({prov['lib_path']}); "
                 "--allow-build-flags marks the line as an experiment")
    from fastdm_amd import distributed as D
    rank, world, local_rank = D.init_from_env(backend="gloo" if args.same_device else None)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run or let bench.py spawn the ranks"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from fastdm_amd.caching import AutoCache
    from fastdm_amd.kernel import timing
    from fastdm_amd.model_entry import create_model
    from fastdm_amd.pipeline import SyntheticFluxPipeline, SyntheticQwenPipeline, flow_match_sigmas
    from fastdm_amd.synthetic import SyntheticFluxStateDict, SyntheticQwenStateDict

    quant = {"fp8": torch.float8_e4m3fn, "int8": torch.int8, "bf16": None}[args.quant]
    cache = AutoCache.from_json(args.cache_config) if args.cache_config else None
    if cache is not None and args.cache_threshold is not None:
        cache.config.threshold = args.cache_threshold
    if cache is not None and args.cache_negative:
        cache.config.negtive_cache = True
    if args.model == "flux":
        model = create_model("flux", ckpt_path=SyntheticFluxStateDict(seed=0), dtype=torch.bfloat16, quant_type=quant,
                             kernel_backend="hip", cache=cache).eval()
        pipe = SyntheticFluxPipeline(model, seed=0)
    elif args.model == "qwen":
        mc = {"num_layers": args.layers} if args.layers else {}
        model = create_model("qwen", ckpt_path=SyntheticQwenStateDict(seed=0, **mc), dtype=torch.bfloat16, quant_type=quant,
                             kernel_backend="hip", cache=cache, out_channels=16, **mc).eval()
        pipe = SyntheticQwenPipeline(model, seed=0, batch_cfg=args.batch_cfg)
    elif args.model == "sd35":
        from fastdm_amd.synthetic import SyntheticSD3StateDict, sd3_step_inputs
        arch = dict(num_layers=38, num_attention_heads=38, attention_head_dim=64, caption_projection_dim=2432, pos_embed_max_size=192,
                    dual_attention_layers=())                                     # SD3.5-large
        model = create_model("sd35", ckpt_path=SyntheticSD3StateDict(seed=0, **arch), dtype=torch.bfloat16, quant_type=quant, kernel_backend="hip",
                             **arch).eval()
        pipe = None
    elif args.model == "sdxl":
        from fastdm_amd.synthetic import SyntheticSDXLStateDict, sdxl_step_inputs
        model = create_model("sdxl", ckpt_path=SyntheticSDXLStateDict(seed=0), dtype=torch.bfloat16, quant_type=quant, kernel_backend="hip",
                             channels_last=not args.nchw).eval()
        pipe = None
    else:
        from fastdm_amd.pipeline import SyntheticWanPipeline
        from fastdm_amd.sparse import SparseAttn
        from fastdm_amd.synthetic import SyntheticWanStateDict
        sparse = SparseAttn.from_json(args.sparse_config) if args.sparse_config else None
        mc = {"num_layers": args.layers} if args.layers else {}
        model = create_model("wan", ckpt_path=SyntheticWanStateDict(seed=0, **mc), dtype=torch.bfloat16, quant_type=quant, kernel_backend="hip",
                             model_config=mc, sparse_attn=sparse).eval()
        pipe = SyntheticWanPipeline(model, seed=0)
        if sparse:
            sparse.config.current_steps_callback = lambda: pipe.scheduler_step_index
    if cache:
        cache.config.current_steps_callback = lambda: pipe.scheduler_step_index
    # rank r denoises image r of the global batch (different latents/prompt per image); under --cfg-pair ranks 2i and 2i+1
    # share image i (fastdm_amd/distributed.py: CFGPair)
    pair = D.make_cfg_pairs(rank, world) if args.cfg_pair else None
    K_IMG = args.images_per_rank
    n_images = (world // 2 if pair is not None else world) * K_IMG
    my_images = D.images_of_rank(rank, world, K_IMG, cfg_pair=pair is not None)
    img_rank = my_images[0]
    if pair is not None:
        if args.model in ("qwen", "wan"):
            pipe.cfg_pair = pair
        elif args.model == "sd35":
            from fastdm_amd.pipeline import SyntheticSD3Pipeline
            pipe2 = SyntheticSD3Pipeline(model, seed=0, cfg_pair=pair)
        else:
            from fastdm_amd.pipeline import SyntheticSDXLPipeline
            pipe2 = SyntheticSDXLPipeline(model, seed=0, use_graph=args.graph, cfg_pair=pair)
    def make_inputs(img_id):
        global IMAGE_STEPS
        if args.model == "sd35":
            IMAGE_STEPS = 28
            inp = sd3_step_inputs(2, args.height // 8, args.width // 8, args.max_seq_len, seed=img_id)      # cond + uncond batched (CFG), as diffusers does
            n_img = (args.height // 16) * (args.width // 16)
            lin = np.linspace(1.0, 1.0 / IMAGE_STEPS, IMAGE_STEPS)
            sig = np.concatenate([3.0 * lin / (1.0 + 2.0 * lin), [0.0]]).astype(np.float32)      # flow-match schedule, shift 3
        elif args.model == "sdxl":
            IMAGE_STEPS = 30
            inp = sdxl_step_inputs(2, args.height // 8, args.width // 8, args.max_seq_len, seed=img_id)     # cond + uncond batched (CFG), as diffusers does
            inp["hidden_states"] = inp["sample"]
            n_img = (args.height // 16) * (args.width // 16)
            sig = np.concatenate([np.linspace(1.0, 1.0 / IMAGE_STEPS, IMAGE_STEPS), [0.0]]).astype(np.float32)
        elif args.model == "wan":
            IMAGE_STEPS = 40
            inp = pipe.prepare(1, args.height, args.width, args.frames, args.max_seq_len, seed=img_id)
            n_img = inp["hidden_states"].shape[2] * (inp["hidden_states"].shape[3] // 2) * (inp["hidden_states"].shape[4] // 2)
            sig = pipe.sigmas(IMAGE_STEPS)
        else:
            inp = pipe.prepare(1, args.height, args.width, args.max_seq_len, seed=img_id)
            n_img = inp["hidden_states"].shape[1]
            sig = flow_match_sigmas(IMAGE_STEPS, n_img) if args.model == "flux" else flow_match_sigmas(IMAGE_STEPS, n_img, max_seq_len=8192, max_shift=0.9)
        return inp, n_img, sig

    inps = []
    for img_id in my_images:
        inp, n_img, sig = make_inputs(img_id)
        inps.append(inp)
    inp = inps[0]
    guidance = torch.full((1,), 3.5, device=dev, dtype=torch.float32)

    states = [{"lat": x["hidden_states"].clone(), "i": 0} for x in inps]
    state = states[0]
    if args.model == "sdxl":
        from fastdm_amd.graph import GraphedForward
        unet = lambda lat, t, ip=inp: model(lat, t, ip["encoder_hidden_states"], ip["added_cond_kwargs"])[0]
        graphed = GraphedForward(unet, inp["hidden_states"], torch.zeros((), device=dev)) if args.graph else None     # (--graph: one image per rank)

        def sdxl_forward(lat, t, ip):
            if graphed is not None:
                return graphed(lat, torch.tensor(t, device=dev))
            return unet(lat, torch.tensor(t, device=dev), ip)

    def one_step():
        """one denoise step of every image this rank holds (the images take turns through the one model, as the reference's prompt loop)"""
        for st, ip in zip(states, inps):
            one_image_step(st, ip)

    def one_image_step(state, inp):
        i = (state["i"] + args.start_step) % IMAGE_STEPS
        if i == 0:
            state["lat"] = inp["hidden_states"].clone()
        if pipe is not None:
            pipe.scheduler_step_index = i
        if pair is not None and args.model == "sd35":
            state["lat"] = pipe2.step(state["lat"][0:1], inp, float(sig[i]), float(sig[i + 1]), 4.5)
            state["i"] += 1
            return
        if pair is not None and args.model == "sdxl":
            state["lat"] = pipe2.step(state["lat"][0:1], inp, float(sig[i]) * 999.0, float(sig[i]), float(sig[i + 1]), 5.0)
            state["i"] += 1
            return
        if args.model == "sd35":
            lat = state["lat"]
            t = torch.full((2,), float(sig[i]) * 1000.0, device=dev, dtype=torch.float32)
            v = model(hidden_states=lat, encoder_hidden_states=inp["encoder_hidden_states"], pooled_projections=inp["pooled_projections"],
                      timestep=t, return_dict=False)[0].float()
            noise = v[1:2] + 4.5 * (v[0:1] - v[1:2])
            state["lat"] = (lat.float() + float(sig[i + 1] - sig[i]) * noise).to(torch.bfloat16).expand(2, -1, -1, -1).contiguous()
            state["i"] += 1
            return
        if args.model == "sdxl":
            lat = state["lat"]
            eps = sdxl_forward(lat, float(sig[i]) * 999.0, inp).float()
            noise = eps[1:2] + 5.0 * (eps[0:1] - eps[1:2])
            state["lat"] = (lat.float() + float(sig[i + 1] - sig[i]) * noise).to(torch.bfloat16).expand(2, -1, -1, -1).contiguous()
            state["i"] += 1
            return
        if args.model == "wan":
            state["lat"] = pipe.step(state["lat"], inp, sig[i], sig[i + 1], 5.0)      # cond + uncond forward, CFG combine, Euler update
            state["i"] += 1
            return
        if args.model == "qwen":
            state["lat"] = pipe.step(state["lat"], inp, sig[i], sig[i + 1], 4.0)    # cond + uncond forward, CFG combine, Euler update
            state["i"] += 1
            return
        t = torch.full((1,), float(sig[i]), device=dev, dtype=torch.bfloat16)
        with model.cache_context("cond"):
            v = model(hidden_states=state["lat"], timestep=t, guidance=guidance, pooled_projections=inp["pooled_projections"],
                      encoder_hidden_states=inp["encoder_hidden_states"], txt_ids=inp["txt_ids"], img_ids=inp["img_ids"],
                      joint_attention_kwargs=None, return_dict=False)[0]
        state["lat"] = (state["lat"].float() + float(sig[i + 1] - sig[i]) * v.float()).to(torch.bfloat16)
        state["i"] += 1

    smi = DeviceState(local_rank) if (rank == 0 and not args.no_device_state and not args.same_device) else None
    if smi:
        smi.sample_smi("before")
    with torch.no_grad():
        for _ in range(args.warmup):
            one_step()
        def gather_final():
            own = torch.cat([st["lat"][0:1] for st in states], dim=0)        # sd35 / sdxl keep the CFG batch-2 copy of each latent this rank owns
            if pair is not None:
                return D.gather_pair_latents(own, pair, rank, world)
            return D.gather_latents(own, n_images, rank, world)
        gather_final()   # RCCL builds its all-gather channels on the first call: not part of a step
        for st in states:
            st["i"] = 0
        torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()
        # HIP-event spans on every 5th timed step (every step for short runs and step-cache runs, whose steps differ): see timing.Recorder
        every = args.kernel_timing_every or (5 if (args.steps >= 10 and cache is None) else 1)
        rec = timing.Recorder(every) if not args.no_kernel_timing else None
        t0 = time.perf_counter()
        if rec:
            rec.__enter__()
        for step_i in range(args.steps):
            if rec:
                rec.begin_step(step_i)
            one_step()
        if rec:
            rec.__exit__(None, None, None)
        torch.cuda.synchronize()
        t_steps = time.perf_counter() - t0          # this rank's own steps (a straggler shows here, not only in the max)
        final = gather_final()      # the job's single end-of-job collective (RCCL all-gather)
        torch.cuda.synchronize()
        t_gather = time.perf_counter() - t0 - t_steps
        D.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if smi:                                     # the clock has stopped: probe the loaded chip, then read rocm-smi once more
            smi.run_probe()
            smi.sample_smi("after")
            smi.sample_in_step(one_step)
    elapsed = D.max_over_ranks(elapsed, dev)
    per_rank = D.gather_floats([t_steps / args.steps * 1e3, t_gather * 1e3], dev)     # [world][2]
    assert torch.isfinite(final.float()).all(), "non-finite latents"

    if rank != 0:
        D.shutdown()
        return
    ms_per_step = elapsed / args.steps * 1e3
    IMG_NOTE = "" if K_IMG == 1 else f" x {K_IMG} images in turn per {'CFG pair' if pair is not None else 'GPU'}"
    skipped = 0
    if cache is not None:
        skipped = sum(1 for t in cache.trace[-(args.steps * (2 if args.model == "qwen" else 1)):] if not t[3])
    if args.model == "flux":
        workload = (f"FLUX.1-dev {args.quant} {args.height}x{args.width} (S={n_img}+{args.max_seq_len}), {IMAGE_STEPS}-step flow-match schedule, "
                    f"batch 1 per GPU{IMG_NOTE}, 19 double + 38 single blocks" + (f", TeaCache (threshold {cache.config.threshold}{' [OVERRIDDEN]' if args.cache_threshold is not None else ''}; {skipped}/{args.steps} steps skipped)" if cache else ""))
    elif args.model == "sd35":
        workload = (f"SD3.5-large {args.quant} {args.height}x{args.width} (S={n_img}+{args.max_seq_len}), 28-step schedule, CFG as one batch-2 forward "
                    f"per step{IMG_NOTE}, 38 joint blocks")
    elif args.model == "sdxl":
        workload = (f"SDXL-base-1.0 UNet {args.quant} {args.height}x{args.width} (latent {args.height // 8}x{args.width // 8}; {n_img} and {n_img // 4} "
                    f"tokens at the two attention levels, {args.max_seq_len} prompt tokens), 30-step schedule, CFG as one batch-2 forward per step{IMG_NOTE}, "
                    f"70 transformer layers + 52 convolutions ({'MIOpen, NCHW' if args.nchw else 'HIP implicit GEMM, channels-last'})")
    elif args.model == "wan":
        hs = inp["hidden_states"].shape
        nl = len(model.core_model.blocks)
        sp = model.core_model.sparse_attn
        mode = "dense attention" if sp is None else (f"radial block-sparse attention (block {sp.config.block_size}, decay {sp.config.decay_factor}, "
                                                     f"dense layers {sp.config.dense_layers}, dense steps {sp.config.dense_steps}; timed steps "
                                                     f"{args.start_step}..{args.start_step + args.steps - 1})")
        workload = (f"Wan2.2-T2V-A14B {args.quant} {args.height}x{args.width}x{args.frames}f (latents {tuple(hs[1:])}, S={n_img}+{args.max_seq_len}), "
                    f"40-step schedule, CFG (2 forwards per step), batch 1 per GPU{IMG_NOTE}, {nl} blocks"
                    + ("" if nl == 40 else " [REDUCED DEPTH]") + f", {mode}")
    else:
        nfw = 2 * args.steps
        workload = (f"Qwen-Image {args.quant} {args.height}x{args.width} (S={n_img}+{args.max_seq_len}), {IMAGE_STEPS}-step schedule, true CFG "
                    f"({'one batch-2 forward' if args.batch_cfg else '2 forwards'} per step), batch 1 per GPU{IMG_NOTE}, {args.layers or 60} double-stream blocks"
                    + (" [REDUCED DEPTH]" if args.layers and args.layers != 60 else "")
                    + (f", TeaCache (threshold {cache.config.threshold}{' [OVERRIDDEN]' if args.cache_threshold is not None else ''}{', negative cache' if args.cache_negative else ''}; {skipped}/{nfw} forwards skipped)" if cache else ""))
    out = {
        # value counts denoise steps of ALL images in flight (n_images = images-per-rank x n_gpus, or x n_gpus / 2 with --cfg-pair);
        # sec/image is per image SLOT (a rank or pair works through its images in turn) divided by the images it holds
        "metric": {"flux": f"denoise it/s (sec/image = 25 / (value / {n_images // K_IMG})), FLUX 1024x2048 fp8 @25 steps",
                   "qwen": f"denoise it/s (sec/image = 25 / (value / {n_images // K_IMG})), Qwen-Image 1328x1328 int8 @25 steps, true CFG",
                   "sd35": f"denoise it/s (sec/image = 28 / (value / {n_images // K_IMG})), SD3.5-large 1024x1024 fp8 @28 steps, CFG",
                   "sdxl": f"denoise it/s (sec/image = 30 / (value / {n_images // K_IMG})), SDXL-base UNet 1024x1024 fp8 @30 steps, CFG",
                   "wan": f"denoise it/s (sec/video = 40 / (value / {n_images // K_IMG})), Wan2.2-T2V-A14B 768x1280x81f fp8 @40 steps, CFG"}[args.model],
        "value": n_images * args.steps / elapsed,
        "unit": "it/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "sec_per_image": ms_per_step * IMAGE_STEPS / 1e3 / K_IMG,     # a step advances every image a rank (pair) holds
        "higher_is_better": True,
        "scaling": "strong" if (pair is not None and world == 2) else "weak",
        "scaling_note": (f"dp{world // 2} x cfg2 x {K_IMG} image(s) per pair: each image is strong-scaled over a GPU pair (cond / uncond forward), pairs are added weakly"
                         if pair is not None else f"dp{world}: {K_IMG} image(s) per GPU"),
        "per_rank_ms_per_step": [round(r[0], 3) for r in per_rank],
        "final_allgather_ms": round(max(r[1] for r in per_rank), 3),
        "vs_baseline": None,
        **prov, **({"experimental": True} if prov["lib_flags"] else {}),
        "device_state": smi.summary() if smi else None,
        "dtype": {"fp8": "fp8_e4m3 W8A8 GEMM (fp32 accumulate) + bf16 attention", "int8": "int8 W8A8 + bf16 attention", "bf16": "bf16"}[args.quant]
                 + (" -> OPT-IN: e4m3 QK^T + e4m3 P.V attention on head_dim-128 dense calls (fdm_attn_fwd_fp8)" if args.attn_fp8 else
                    " -> OPT-IN: bf16 QK^T + e4m3 P.V attention on head_dim-128 dense calls (fdm_attn_fwd_pvfp8)" if args.attn_pv_fp8 else ""),
        **({"variant": ("attn_fp8" if args.attn_fp8 else "attn_pv_fp8") + " (quantised attention, the reference's SageAttention class; NOT the headline arithmetic)"}
           if (args.attn_pv_fp8 or args.attn_fp8) else {}),
        "data": f"synthetic (counter-based random weights of the {ARCH_NAME[args.model]} architecture, synthetic latents/prompt embeddings)",
        "config": {"workload": workload, "global_batch": n_images, "height": args.height, "width": args.width,
                   "parallelism": (f"dp{world // 2} x cfg2 x {K_IMG} image(s) per pair (ranks 2i, 2i+1 run the conditional / unconditional forward of image i; one all-gather of the "
                                   f"two noise predictions per step inside the pair, one all-gather of final latents)" if pair is not None else
                                   f"dp{world} x {K_IMG} image(s) per GPU (image-sharded, one all-gather of final latents)"),
                   "ranks_in_process_group": D.group_size()},
    }
    if args.same_device:
        out["scaling_note"] = "PROTOCOL TEST: all ranks share cuda:0 over gloo (--same-device); `value` is not a scaling measurement. " + out["scaling_note"]
        out["config"]["transport"] = "gloo, every rank on cuda:0 (--same-device)"
    if rec:
        summ = rec.summary()
        n_timed = max(1, rec.sampled_steps)     # the steps whose launches carry HIP events
        kernels = {}
        for name, s in summ.items():
            dt = "bf16qk+fp8pv" if "pvfp8" in name else ("bf16" if "bf16" in name else ("int8" if "int8" in name else "fp8"))      # attn_fwd_fp8 -> "fp8"
            tf = s["work"] / (s["total_ms"] * 1e-3) / 1e12
            kernels[name] = {"launches_per_step": s["launches"] / n_timed, "avg_ms": s["avg_ms"], "ms_per_step": s["total_ms"] / n_timed,
                             "achieved_TFLOPs": tf, "peak_TFLOPs": PEAK_TFLOPS[dt], "frac": tf / PEAK_TFLOPS[dt]}
            if "by_kernel" in s:      # the tile kernel each launch of the span actually ran (fdm_gemm_last_kernel; one span can mix kernels)
                kernels[name]["hip_kernels"] = {k: {"launches_per_step": v["launches"] / n_timed, "ms_per_step": v["total_ms"] / n_timed,
                                                    "achieved_TFLOPs": v["work"] / (v["total_ms"] * 1e-3) / 1e12} for k, v in s["by_kernel"].items()}
        dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
        default_workload = (args.model, args.height, args.width, args.max_seq_len, args.quant, args.cache_config) == ("flux", 1024, 2048, 512, "fp8", None)
        dom_kernel = span_kernel(dom, summ[dom].get("by_kernel"))
        traffic, traffic_src = pmc_traffic(dom_kernel) if default_workload else (None, "not the default workload")
        out["roofline"] = {"kernel": dom_kernel, "span": dom, "bound": "mfma", "achieved": kernels[dom]["achieved_TFLOPs"],
                           "peak": kernels[dom]["peak_TFLOPs"], "unit": "TFLOP/s", "frac": kernels[dom]["frac"],
                           "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": kernels[dom]["avg_ms"],
                           "timed_launches": summ[dom]["launches"],
                           "event_sampling": f"HIP-event pairs around every launch of {n_timed} of the {args.steps} timed steps (every {rec.every}-th)"}
        out["kernels"] = kernels
        # the W8A8 GEMM family as ONE entry (its instantiations are separate spans above, so the family never shows up as the
        # "dominant kernel" although it is the largest share of the step)
        fam = [k for k in summ if k.startswith("gemm_w8a8")]
        if fam:
            fdt = "int8" if "int8" in fam[0] else "fp8"
            f_ms, f_work = sum(summ[k]["total_ms"] for k in fam), sum(summ[k]["work"] for k in fam)
            f_tf = f_work / (f_ms * 1e-3) / 1e12
            ran = sorted({k for f in fam for k in summ[f].get("by_kernel", {})})
            out["roofline_family"] = {"family": f"W8A8 {fdt} GEMM, all epilogues; kernels launched: {', '.join(ran) if ran else 'n/a'}", "spans": fam, "bound": "mfma",
                                      "achieved": f_tf, "peak": PEAK_TFLOPS[fdt], "unit": "TFLOP/s", "frac": f_tf / PEAK_TFLOPS[fdt],
                                      "ms_per_step": f_ms / n_timed, "share_of_step": f_ms / n_timed / ms_per_step}
        flops_step = sum(s["work"] for s in summ.values()) / n_timed
        out["step_TFLOP"] = flops_step / 1e12
        out["step_achieved_TFLOPs"] = flops_step / (ms_per_step * 1e-3) / 1e12
    if world == 1 and not args.no_cpu_baseline and args.model in ("flux", "qwen", "sdxl"):
        try:
            leg = {"flux": cpu_baseline_leg, "qwen": cpu_baseline_qwen, "sdxl": cpu_baseline_sdxl}[args.model]
            out["cpu_baseline"] = leg(args.height, args.width, args.max_seq_len, args.quant)
        except Exception as e:  # noqa: BLE001  -- the baseline leg must never take the GPU number down with it
            out["cpu_baseline"] = {"value": None, "unit": "it/s", "cores": torch.get_num_threads(), "kind": "port", "sample": f"failed: {e}"}
    emit_result(json.dumps(out))
    D.shutdown()


if __name__ == "__main__":
    main()
