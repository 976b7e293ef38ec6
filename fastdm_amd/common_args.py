"""Command-line flags of the reference's gen.py (fastdm/common_args.py:3-40), so its invocations carry over unchanged.
Declared as a table: (flag, kind, default, help); kind is a type, "flag" (store_true) or a tuple of choices."""
import argparse

_COMMON = [
    ("--use-diffusers", "flag", None, "use hf-diffusers"),
    ("--use-fp8", "flag", None, "fp8 (e4m3) W8A8 inference"),
    ("--use-int8", "flag", None, "int8 W8A8 inference"),
    ("--kernel-backend", str, "cuda", "kernel backend: hip (alias: cuda)"),
    ("--oom-resolve", "flag", None, "accepted for compatibility; unnecessary with 288 GB of HBM"),
    ("--model-path", str, "", "diffusion model directory ('' = synthetic weights)"),
    ("--data-type", str, "bfloat16", "activation data type: bfloat16"),
    ("--architecture", str, "flux", "model architecture: flux | qwen | wan | wan-i2v | sd35 | sdxl"),
    ("--cache-config", str, None, "step-cache JSON (examples/xcaching/configs/)"),
    ("--steps", int, 25, "denoise steps"),
    ("--guidance-scale", float, 3.5, "guidance scale"),
    ("--true-cfg-scale", float, 4.0, "true-CFG scale (Qwen-Image)"),
    ("--seed", int, 0, "generation seed"),
    ("--device", int, 0, "GPU index"),
    ("--num-warmup-runs", int, 0, "untimed generations before the timed one"),
    ("--prompts", str, "An astronaut riding a horse", "text prompt"),
    ("--negative-prompts", str, None, "negative text prompt"),
    ("--width", int, 2048, "output width"),
    ("--height", int, 1024, "output height"),
    ("--output-path", str, "output.png", "where the result is written"),
    ("--task", ("t2i", "t2v", "i2i", "i2v"), "t2i", "task type"),
]
_TEXT_GEN = [
    ("--max-seq-len", int, 512, "maximum prompt length in tokens"),
    ("--num-frames", int, 121, "video frames"),
    ("--fps", int, 24, "video frame rate"),
    ("--image-path", str, None, "input image (i2v)"),
    ("--sparse-attn-config", str, None, "sparse attention JSON (examples/sparse/)"),
]


def _add(parser, table):
    for flag, kind, default, text in table:
        if kind == "flag":
            parser.add_argument(flag, action="store_true", help=text)
        elif isinstance(kind, tuple):
            parser.add_argument(flag, type=str, default=default, choices=list(kind), help=text)
        else:
            parser.add_argument(flag, type=kind, default=default, help=text)


def add_common_args(parser):
    _add(parser, _COMMON)


def get_text_gen_parser():
    parser = argparse.ArgumentParser(description="Options for Text-to-Image Generation")
    _add(parser, _COMMON)
    _add(parser, _TEXT_GEN)
    return parser
