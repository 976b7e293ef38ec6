"""Worker for tests/test_multigpu_gpu.py (not a test module): runs the multi-process denoise paths on real devices and
checks them bit for bit against the single-process loop.

    python tests/multigpu_worker.py --mode flux_shard|qwen_pair|qwen_pair_teacache|sd3_pair --world 2 --backend nccl|gloo [--same-device]

Started without RANK in the environment it spawns `--world` fresh child processes (one per rank) BEFORE touching the GPU
and exits non-zero if any of them fails.  `--same-device` puts every rank on cuda:0 with the gloo backend: the transport
differs from RCCL but the protocol (who runs which forward, what is gathered, which cache entry a rank keeps) is the same,
so the bit-equality checks run on a 1-GPU box too."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def spawn(args):
    from fastdm_amd.launch import spawn_self
    rc = spawn_self(__file__, sys.argv[1:], args.world, timeout=float(os.environ.get("FDM_MULTIGPU_TIMEOUT", "900")), same_device=args.same_device)
    if rc == 0:
        print("MULTIGPU_OK")
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", required=True, choices=["flux_shard", "qwen_pair", "qwen_pair_teacache", "sd3_pair"])
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--same-device", action="store_true")
    ap.add_argument("--steps", type=int, default=4)
    args = ap.parse_args()
    if "RANK" not in os.environ:
        sys.exit(spawn(args))

    import torch
    from fastdm_amd import distributed as D
    from fastdm_amd.model_entry import create_model
    from fastdm_amd.pipeline import SyntheticFluxPipeline, SyntheticQwenPipeline, SyntheticSD3Pipeline, flow_match_sigmas
    rank, world, local_rank = D.init_from_env(backend=args.backend)
    torch.cuda.set_device(local_rank)
    BF16, FP8 = torch.bfloat16, torch.float8_e4m3fn
    dev = lambda d: {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in d.items()}

    with torch.no_grad():
        if args.mode == "flux_shard":
            from oracle import flux as OF
            cfg = OF.TINY
            model = create_model("flux", ckpt_path=OF.synth_state_dict(cfg, seed=3), dtype=BF16, quant_type=FP8, kernel_backend="hip",
                                 in_channels=cfg.in_channels, out_channels=cfg.out_channels, num_layers=cfg.num_layers,
                                 num_single_layers=cfg.num_single_layers, attention_head_dim=cfg.attention_head_dim,
                                 num_attention_heads=cfg.num_attention_heads, joint_attention_dim=cfg.joint_attention_dim,
                                 pooled_projection_dim=cfg.pooled_projection_dim, axes_dims_rope=cfg.axes_dims_rope).eval()
            pipe = SyntheticFluxPipeline(model)
            n_items = world + 1                                   # ragged on purpose: rank 0 owns two images

            def denoise(i):
                inp = dev(OF.synth_inputs(cfg, 1, 4, 6, 8, seed=100 + i))
                return pipe.denoise(inp["hidden_states"].clone(), inp, flow_match_sigmas(args.steps, 24), 3.5)
            mine = [denoise(i) for i in D.shard_indices(n_items, rank, world)]
            full = D.gather_latents(torch.cat(mine, dim=0), n_items, rank, world)
            if rank == 0:
                want = torch.cat([denoise(i) for i in range(n_items)], dim=0)
                assert torch.equal(full, want), "image-sharded latents differ from the single-process latents"
        elif args.mode in ("qwen_pair", "qwen_pair_teacache"):
            from oracle import qwen as OQ
            from fastdm_amd.caching import AutoCache
            cfg = OQ.TINY

            def build(pair):
                cache = None
                if args.mode == "qwen_pair_teacache":
                    cache = AutoCache.from_dict({"cache_algorithm": "teacache", "enable_caching": True, "threshold": 0.08, "negtive_cache": True,
                                                 "coefficients": [1.0, 0.0], "negtive_coefficients": [1.2, 0.0], "wamup_steps": 1})
                m = create_model("qwen", ckpt_path=OQ.synth_state_dict(cfg, seed=11), dtype=BF16, quant_type=torch.int8, kernel_backend="hip",
                                 cache=cache, in_channels=cfg.in_channels, out_channels=cfg.out_channels, num_layers=cfg.num_layers,
                                 attention_head_dim=cfg.attention_head_dim, num_attention_heads=cfg.num_attention_heads,
                                 joint_attention_dim=cfg.joint_attention_dim, axes_dims_rope=cfg.axes_dims_rope).eval()
                p = SyntheticQwenPipeline(m, cfg_pair=pair)
                if cache is not None:
                    cache.config.current_steps_callback = lambda: p.scheduler_step_index
                    cache.config.total_steps_callback = lambda: p.num_steps
                return p, cache

            def run(pipe, image):
                inp = dev(OQ.synth_inputs(cfg, 1, 4, 6, 8, seed=200 + image))
                neg = dev(OQ.synth_inputs(cfg, 1, 4, 6, 8, seed=900 + image))
                inp["negative_encoder_hidden_states"] = neg["encoder_hidden_states"]
                lat = inp["hidden_states"].clone()
                sig = flow_match_sigmas(args.steps, 24, max_seq_len=8192, max_shift=0.9)
                pipe.num_steps = args.steps
                for i in range(args.steps):
                    pipe.scheduler_step_index = i
                    lat = pipe.step(lat, inp, sig[i], sig[i + 1], 4.0)
                return lat
            pair = D.make_cfg_pairs(rank, world)
            pipe, cache = build(pair)
            lat = run(pipe, pair.image_index)
            full = D.gather_pair_latents(lat, pair, rank, world)
            decisions = [bool(t[3]) for t in cache.trace] if cache is not None else []
            if rank == 0:
                ref_pipe, ref_cache = build(None)
                want = run(ref_pipe, 0)          # image 0 = this pair's image
                assert torch.equal(full[0:1], want), "CFG-pair latents differ from the single-GPU two-forward latents"
                if ref_cache is not None:
                    # single-GPU trace alternates cond, uncond, cond, ...: this rank (role 0) must have taken the cond decisions
                    ref = [bool(t[3]) for t in ref_cache.trace]
                    assert decisions == ref[0::2], (decisions, ref)
                    assert not all(ref), "the TeaCache configuration never skipped: the test does not exercise the cache"
            elif rank == 1 and cache is not None:
                ref_pipe, ref_cache = build(None)
                run(ref_pipe, 0)
                ref = [bool(t[3]) for t in ref_cache.trace]
                assert decisions == ref[1::2], (decisions, ref)
        else:
            from oracle import sd35 as OS
            cfg = OS.TINY
            model = create_model("sd35", ckpt_path=OS.synth_state_dict(cfg, seed=5), dtype=BF16, quant_type=FP8, kernel_backend="hip",
                                 sample_size=cfg.sample_size, patch_size=cfg.patch_size, in_channels=cfg.in_channels, out_channels=cfg.out_channels,
                                 num_layers=cfg.num_layers, attention_head_dim=cfg.attention_head_dim, num_attention_heads=cfg.num_attention_heads,
                                 joint_attention_dim=cfg.joint_attention_dim, caption_projection_dim=cfg.caption_projection_dim,
                                 pooled_projection_dim=cfg.pooled_projection_dim, pos_embed_max_size=cfg.pos_embed_max_size,
                                 dual_attention_layers=cfg.dual_attention_layers).eval()
            inp = dev(OS.synth_inputs(cfg, 2, 8, 8, 7, seed=5))
            sig = SyntheticSD3Pipeline.sigmas(args.steps)

            def run(pipe):
                lat = inp["hidden_states"][0:1].clone()
                for i in range(args.steps):
                    lat = pipe.step(lat, inp, sig[i], sig[i + 1], 4.5)
                return lat
            pair = D.make_cfg_pairs(rank, world)
            lat = run(SyntheticSD3Pipeline(model, cfg_pair=pair))
            full = D.gather_pair_latents(lat, pair, rank, world)
            if rank == 0:
                assert torch.equal(full[0:1], run(SyntheticSD3Pipeline(model))), "sd35 CFG-pair latents differ from the batch-2 forward's"
    D.barrier()
    D.shutdown()


if __name__ == "__main__":
    main()
