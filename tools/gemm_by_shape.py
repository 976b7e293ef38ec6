#!/usr/bin/env python3
"""Per-dispatch, per-shape table of the GEMM launches INSIDE the denoise step.

    python3 tools/gemm_by_shape.py record seq.json [bench.py model flags]     # on the GPU box: one forward with the shape log on
    rocprofv3 --kernel-trace --output-format csv -d DIR -o run -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-device-state
    python3 tools/gemm_by_shape.py join DIR/.../run_kernel_trace.csv seq.json out.csv

`record` writes the launch sequence of ONE forward pass -- [{dtype, problems: [[M, N, K, epilogue], ...]}, ...] in launch order
(fastdm_amd/kernel/timing.py: log_gemm, called by every QLinear entry point).  `join` walks the trace's tile-GEMM dispatches
(gemm_p4_kernel / gemm_w4_kernel / gemm_kernel) in start order; every forward pass issues the same sequence, so dispatch i belongs to
launch i mod len(sequence).  Output: one row per launch KIND (dtype, shapes, epilogue, HIP kernel): launches per pass, mean / min / max us,
TFLOP per launch, PFLOP/s, fraction of the dense MFMA peak (5.0 PF fp8 / int8, 2.5 PF bf16).
"""
import csv
import json
import os
import statistics
import sys
from collections import OrderedDict, defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PEAK = {"fp8": 5000.0, "int8": 5000.0, "bf16": 2500.0}
EPI = {0: "plain", 1: "gelu_tanh", 2: "gelu_erf", 3: "gate+res", 4: "gate_f32+res"}
TILE_KERNELS = ("gemm_p4_kernel", "gemm_w4_kernel", "gemm_kernel<")


def record(path, argv):
    import torch
    sys.argv = ["bench.py"] + argv
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="flux")
    ap.add_argument("--quant", default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--max-seq-len", type=int, default=None)
    ap.add_argument("--batch-cfg", action="store_true")
    a = ap.parse_args(argv)
    dflt = {"flux": (1024, 2048, 512, "fp8"), "qwen": (1328, 1328, 128, "int8")}[a.model]
    h, w, sl, qn = a.height or dflt[0], a.width or dflt[1], a.max_seq_len or dflt[2], a.quant or dflt[3]
    quant = {"fp8": torch.float8_e4m3fn, "int8": torch.int8, "bf16": None}[qn]
    from fastdm_amd.kernel import timing
    from fastdm_amd.model_entry import create_model
    from fastdm_amd.pipeline import SyntheticFluxPipeline, SyntheticQwenPipeline, flow_match_sigmas
    from fastdm_amd.synthetic import SyntheticFluxStateDict, SyntheticQwenStateDict
    dev = torch.device("cuda", 0)
    with torch.no_grad():
        if a.model == "flux":
            model = create_model("flux", ckpt_path=SyntheticFluxStateDict(seed=0), dtype=torch.bfloat16, quant_type=quant, kernel_backend="hip").eval()
            pipe = SyntheticFluxPipeline(model, seed=0)
            inp = pipe.prepare(1, h, w, sl, seed=0)
            t = torch.full((1,), 1.0, device=dev, dtype=torch.bfloat16)
            g = torch.full((1,), 3.5, device=dev, dtype=torch.float32)

            def fwd():
                return model(hidden_states=inp["hidden_states"], timestep=t, guidance=g, pooled_projections=inp["pooled_projections"],
                             encoder_hidden_states=inp["encoder_hidden_states"], txt_ids=inp["txt_ids"], img_ids=inp["img_ids"],
                             joint_attention_kwargs=None, return_dict=False)[0]
        else:
            model = create_model("qwen", ckpt_path=SyntheticQwenStateDict(seed=0), dtype=torch.bfloat16, quant_type=quant, kernel_backend="hip",
                                 out_channels=16).eval()
            pipe = SyntheticQwenPipeline(model, seed=0, batch_cfg=a.batch_cfg)
            inp = pipe.prepare(1, h, w, sl, seed=0)
            sig = flow_match_sigmas(25, inp["hidden_states"].shape[1], max_seq_len=8192, max_shift=0.9)

            def fwd():
                return pipe.step(inp["hidden_states"], inp, sig[0], sig[1], 4.0)
        fwd()
        timing.shape_log = []
        fwd()
        seq, timing.shape_log = timing.shape_log, None
        torch.cuda.synchronize()
    json.dump({"argv": argv, "launches": seq}, open(path, "w"))
    print(f"{len(seq)} GEMM launches per pass -> {path}")


def join(trace_csv, seq_json, out_csv):
    seq_all = json.load(open(seq_json))["launches"]
    # launches the library sends to the GEMV kernel (fdm_gemm_bf16 / fdm_linear_skinny_bf16 with M <= 8) are not tile-kernel dispatches
    seq = [l for l in seq_all if not (l["dtype"] == "bf16" and all(p[0] <= 8 for p in l["problems"]))]
    rows = list(csv.DictReader(open(trace_csv)))
    name_key = "Kernel_Name" if "Kernel_Name" in rows[0] else "Name"
    disp = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[name_key]) for r in rows if any(k in r[name_key] for k in TILE_KERNELS)]
    disp.sort()
    n = len(seq)
    if len(disp) % n:
        print(f"WARNING: {len(disp)} tile-GEMM dispatches are not a multiple of the {n} launches of one pass (int8 split-K merge launches, a partial pass?)", file=sys.stderr)
    kinds = OrderedDict()
    for i, (t0, t1, name) in enumerate(disp):
        l = seq[i % n]
        kern = name.replace("void ", "").split("(")[0]
        key = (l["dtype"], tuple(tuple(p) for p in l["problems"]), kern)
        kinds.setdefault(key, []).append((t1 - t0) / 1e3)
    passes = len(disp) / n
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["dtype", "problems (M x N x K, epilogue)", "hip_kernel", "launches_per_pass", "mean_us", "min_us", "max_us", "TFLOP_per_launch",
                    "PFLOPs", "frac_of_peak", "ms_per_pass"])
        tot = defaultdict(float)
        for (dt, probs, kern), us in kinds.items():
            fl = sum(2.0 * m * nn * k for m, nn, k, _ in probs)
            mean = statistics.fmean(us)
            pf = fl / (mean * 1e-6) / 1e15
            desc = " + ".join(f"{m}x{nn}x{k} {EPI.get(e, e)}" for m, nn, k, e in probs)
            w.writerow([dt, desc, kern, round(len(us) / passes, 2), round(mean, 1), round(min(us), 1), round(max(us), 1), round(fl / 1e12, 4),
                        round(pf, 3), round(pf * 1e3 / PEAK[dt], 3), round(sum(us) / passes / 1e3, 3)])
            tot[dt] += sum(us) / passes / 1e3
        for dt, ms in tot.items():
            w.writerow([dt, "ALL", "", "", "", "", "", "", "", "", round(ms, 3)])
    print(open(out_csv).read())


if __name__ == "__main__":
    if sys.argv[1] == "record":
        record(sys.argv[2], sys.argv[3:])
    else:
        join(sys.argv[2], sys.argv[3], sys.argv[4])
