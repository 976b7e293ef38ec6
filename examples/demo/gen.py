#!/usr/bin/env python3
"""Text-to-image driver with the reference's flags (examples/demo/gen.py of the reference):

    python examples/demo/gen.py --architecture flux --height 1024 --width 2048 --steps 25 --use-fp8
    python examples/demo/gen.py --architecture sdxl --height 1024 --width 1024 --steps 30 --guidance-scale 5.0 --max-seq-len 77 --use-fp8
    python examples/demo/gen.py --architecture sd35 --height 1024 --width 1024 --steps 28 --guidance-scale 4.5 --max-seq-len 333 --use-fp8
    python examples/demo/gen.py --model-path /path/to/FLUX.1-dev --use-fp8 --output-path flux-fp8.png   # needs diffusers

Without --model-path (no checkpoints / diffusers in this image) the engine denoises with synthetic weights of the chosen
architecture (flux, qwen, wan, sd35, sdxl; wan-i2v needs a checkpoint) and synthetic prompt embeddings; the output is then the final packed latent tensor, saved with torch.save.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import torch  # noqa: E402

from fastdm_amd.common_args import get_text_gen_parser  # noqa: E402
from fastdm_amd.model_entry import FastDMEngine  # noqa: E402

if __name__ == "__main__":
    args = get_text_gen_parser().parse_args()
    print(f"kernel backend: {args.kernel_backend}; " + ("fp8 W8A8" if args.use_fp8 else "int8 W8A8" if args.use_int8 else "bf16"))
    t0 = time.time()
    engine = FastDMEngine(model_path=args.model_path, architecture=args.architecture, device=args.device, data_type=args.data_type,
                          use_fp8=args.use_fp8, use_int8=args.use_int8, kernel_backend=args.kernel_backend,
                          cache_config=args.cache_config, oom_resolve=args.oom_resolve, use_diffusers=args.use_diffusers,
                          task=args.task, sparse_attn_config=args.sparse_attn_config)
    torch.cuda.synchronize()
    print(f"Model loading latency: {time.time() - t0:.4f} seconds")

    def run():
        # the reference's call, argument for argument (/root/reference/examples/demo/gen.py:79-103)
        return engine.generate(prompt=args.prompts, negative_prompt=args.negative_prompts,
                               num_frames=args.num_frames if args.task in ("t2v", "i2v") else None, steps=args.steps,
                               guidance_scale=args.guidance_scale, gen_seed=args.seed, gen_width=args.width,
                               gen_height=args.height, max_seq_len=args.max_seq_len,
                               true_cfg_scale=args.true_cfg_scale if args.architecture == "qwen" else None,
                               src_image=args.image_path if args.task == "i2v" else None)

    for i in range(args.num_warmup_runs):
        print(f"warmup {i + 1}/{args.num_warmup_runs}")
        run()
    torch.cuda.synchronize()
    t0 = time.time()
    output = run()
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"Generation latency: {dt:.4f} seconds ({args.steps / dt:.2f} it/s)")
    if args.task in ("t2v", "i2v") and not torch.is_tensor(output):
        from diffusers.utils import export_to_video
        export_to_video(output, args.output_path, fps=args.fps)
    elif hasattr(output, "save"):
        output.save(args.output_path)
    else:
        path = os.path.splitext(args.output_path)[0] + ".latents.pt"
        torch.save(output.cpu(), path)
        print(f"synthetic pipeline: final latents {tuple(output.shape)} saved to {path}")
    print(f"max mem-usage: {torch.cuda.max_memory_allocated() / 2 ** 30:.2f} GB")
