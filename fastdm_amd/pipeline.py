"""A minimal FLUX denoise loop for machines without diffusers / checkpoints / text encoders / VAE.

It reproduces what diffusers' FluxPipeline does around `pipe.transformer`: packed
latents [B, (H/16)*(W/16), 64], img_ids grid, zero txt_ids, a flow-match Euler schedule with the
resolution-dependent time shift, `timestep = sigma` (i.e. t/1000) and `guidance` vectors, and
`latents += (sigma_next - sigma) * noise_pred` per step.  Prompt embeddings are synthetic (U[0,1) like
examples/profiling/flux_profiling.py:16-23).  The hot path inside the loop is exactly the
`transformer(hidden_states=..., timestep=..., guidance=..., pooled_projections=...,
encoder_hidden_states=..., txt_ids=..., img_ids=..., return_dict=False)[0]` call of the real pipeline.
"""
import math

import numpy as np
from typing import Optional

import torch

from fastdm_amd import synthetic


def flow_match_sigmas(num_steps: int, image_seq_len: int, base_seq_len=256, max_seq_len=4096, base_shift=0.5, max_shift=1.15):
    """FlowMatchEulerDiscreteScheduler with FLUX's dynamic shifting (diffusers FluxPipeline.calculate_shift +
    scheduler.set_timesteps(sigmas=linspace(1, 1/N, N), mu=mu)); returns N+1 sigmas ending in 0."""
    sigmas = np.linspace(1.0, 1.0 / num_steps, num_steps)
    m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
    mu = image_seq_len * m + (base_shift - m * base_seq_len)
    sigmas = math.exp(mu) / (math.exp(mu) + (1.0 / sigmas - 1.0))
    return np.concatenate([sigmas, [0.0]]).astype(np.float32)


class SyntheticFluxPipeline:
    def __init__(self, transformer, seed: int = 0):
        self.transformer = transformer
        self.seed = seed
        self.scheduler_step_index = None
        self.num_steps = 0
        self._inputs_key, self._inputs = None, None

    def prepare(self, batch, height, width, max_sequence_length, seed):
        key = (batch, height, width, max_sequence_length, seed)
        if key != self._inputs_key:
            cfg = self.transformer.config
            self._inputs = synthetic.flux_step_inputs(batch, height // 16, width // 16, max_sequence_length, cfg.in_channels, seed=seed,
                                                      joint_attention_dim=self.transformer.core_model.context_embedder.in_features,
                                                      pooled_projection_dim=self.transformer.core_model.time_text_embed.text_embedder.linear1.in_features)
            self._inputs_key = key
        return self._inputs

    @torch.no_grad()
    def denoise(self, latents, inp, sigmas, guidance_scale, step_callback=None):
        B = latents.shape[0]
        guidance = torch.full((B,), guidance_scale, device=latents.device, dtype=torch.float32)
        self.num_steps = len(sigmas) - 1
        for i in range(self.num_steps):
            self.scheduler_step_index = i
            t = torch.full((B,), float(sigmas[i]), device=latents.device, dtype=latents.dtype)
            with self.transformer.cache_context("cond"):
                noise_pred = self.transformer(hidden_states=latents, timestep=t, guidance=guidance,
                                              pooled_projections=inp["pooled_projections"],
                                              encoder_hidden_states=inp["encoder_hidden_states"], txt_ids=inp["txt_ids"],
                                              img_ids=inp["img_ids"], joint_attention_kwargs=None, return_dict=False)[0]
            # scheduler.step: x += (sigma_next - sigma) * v, computed in fp32 and cast back (diffusers upcasts the sample)
            latents = (latents.float() + float(sigmas[i + 1] - sigmas[i]) * noise_pred.float()).to(latents.dtype)
            if step_callback is not None:
                step_callback(i, latents)
        self.scheduler_step_index = None
        return latents

    def __call__(self, num_inference_steps=25, guidance_scale=3.5, width=2048, height=1024, max_sequence_length=512, batch=1,
                 seed=None, latents=None):
        seed = self.seed if seed is None else seed
        inp = self.prepare(batch, height, width, max_sequence_length, seed)
        lat = inp["hidden_states"].clone() if latents is None else latents
        sigmas = flow_match_sigmas(num_inference_steps, lat.shape[1])
        return self.denoise(lat, inp, sigmas, guidance_scale)


class SyntheticQwenPipeline:
    """What diffusers' QwenImagePipeline does around `pipe.transformer` with true CFG: per step a conditional and an
    unconditional forward (`cache_context("cond")` / `("uncond")`), `comb = uncond + scale * (cond - uncond)` rescaled
    to the conditional prediction's per-token norm, then the flow-match Euler update.  Latents are packed
    [B, (H/16)*(W/16), 64]; `timestep = sigma`; prompt / negative-prompt embeddings are synthetic."""

    def __init__(self, transformer, seed: int = 0, batch_cfg: Optional[bool] = None, cfg_pair=None):
        """cfg_pair: a `distributed.CFGPair` -- this rank runs only its branch of the CFG forward pair and the two noise
        predictions meet in one all-gather per step (the combine + Euler update then run identically on both ranks).
        batch_cfg: run the conditional and the unconditional prompt as ONE batch-2 forward (same bits as two forwards: all
        quantisation is per token) when both prompts have the same length and no step cache keys on the cond / uncond
        alternation; the N = 3072 GEMMs then fill 2.6 instead of 1.3 rounds of the chip.  None (default, round 6): on whenever it applies --
        no CFG pair, no step cache on the transformer (198 -> 179 ms per 1328 x 1328 step for the same bits)."""
        self.transformer = transformer
        self.seed = seed
        if batch_cfg is None:
            batch_cfg = cfg_pair is None and not getattr(getattr(transformer, "core_model", None), "enable_caching", False)
        self.batch_cfg = batch_cfg
        self.cfg_pair = cfg_pair
        if cfg_pair is not None and batch_cfg:
            raise ValueError("cfg_pair splits the two forwards over two GPUs; batch_cfg fuses them on one")
        self.scheduler_step_index = None
        self.num_steps = 0
        self._inputs_key, self._inputs = None, None

    def prepare(self, batch, height, width, max_sequence_length, seed):
        key = (batch, height, width, max_sequence_length, seed)
        if key != self._inputs_key:
            core = self.transformer.core_model
            pos = synthetic.qwen_step_inputs(batch, height // 16, width // 16, max_sequence_length, self.transformer.config.in_channels,
                                             core.txt_in.in_features, seed=seed)
            neg = synthetic.qwen_step_inputs(batch, height // 16, width // 16, max_sequence_length, self.transformer.config.in_channels,
                                             core.txt_in.in_features, seed=seed + 7919)
            pos["negative_encoder_hidden_states"] = neg["encoder_hidden_states"]
            self._inputs, self._inputs_key = pos, key
        return self._inputs

    def _forward(self, name, latents, t, enc, inp):
        with self.transformer.cache_context(name):
            return self.transformer(hidden_states=latents, timestep=t, guidance=None, encoder_hidden_states_mask=inp["encoder_hidden_states_mask"],
                                    encoder_hidden_states=enc, img_shapes=inp["img_shapes"], txt_seq_lens=inp["txt_seq_lens"],
                                    attention_kwargs=None, return_dict=False)[0]

    @torch.no_grad()
    def step(self, latents, inp, sigma, sigma_next, true_cfg_scale):
        t = torch.full((latents.shape[0],), float(sigma), device=latents.device, dtype=latents.dtype)
        use_cfg = true_cfg_scale is not None and true_cfg_scale > 1
        if use_cfg and self.batch_cfg:
            b = latents.shape[0]
            both = dict(inp, encoder_hidden_states_mask=torch.cat([inp["encoder_hidden_states_mask"]] * 2, dim=0),
                        img_shapes=list(inp["img_shapes"]) * 2, txt_seq_lens=list(inp["txt_seq_lens"]) * 2)
            out = self._forward("cond", torch.cat([latents, latents], dim=0), torch.cat([t, t], dim=0),
                                torch.cat([inp["encoder_hidden_states"], inp["negative_encoder_hidden_states"]], dim=0), both)
            cond, uncond = out[:b], out[b:]
        elif use_cfg and self.cfg_pair is not None:
            self.cfg_pair.pin_cache(self.transformer)
            if self.cfg_pair.role == 0:
                mine = self._forward("cond", latents, t, inp["encoder_hidden_states"], inp)
            else:
                mine = self._forward("uncond", latents, t, inp["negative_encoder_hidden_states"], inp)
            cond, uncond = self.cfg_pair.exchange(mine)
        else:
            cond = self._forward("cond", latents, t, inp["encoder_hidden_states"], inp)
        if use_cfg:
            if not self.batch_cfg and self.cfg_pair is None:
                uncond = self._forward("uncond", latents, t, inp["negative_encoder_hidden_states"], inp)
            comb = uncond.float() + true_cfg_scale * (cond.float() - uncond.float())
            cond_norm = torch.norm(cond.float(), dim=-1, keepdim=True)
            comb = comb * (cond_norm / torch.norm(comb, dim=-1, keepdim=True))
        else:
            comb = cond.float()
        return (latents.float() + float(sigma_next - sigma) * comb).to(latents.dtype)

    @torch.no_grad()
    def __call__(self, num_inference_steps=25, true_cfg_scale=4.0, width=1328, height=1328, max_sequence_length=128, batch=1, seed=None,
                 latents=None, **_):
        seed = self.seed if seed is None else seed
        inp = self.prepare(batch, height, width, max_sequence_length, seed)
        lat = inp["hidden_states"].clone() if latents is None else latents
        sigmas = flow_match_sigmas(num_inference_steps, lat.shape[1], max_seq_len=8192, base_shift=0.5, max_shift=0.9)
        self.num_steps = num_inference_steps
        for i in range(num_inference_steps):
            self.scheduler_step_index = i
            lat = self.step(lat, inp, sigmas[i], sigmas[i + 1], true_cfg_scale)
        self.scheduler_step_index = None
        return lat


class SyntheticWanPipeline:
    """What diffusers' WanPipeline (and, when the transformer takes more input than output channels, WanImageToVideoPipeline for
    Wan2.2-I2V) does around `pipe.transformer` with classifier-free guidance: latents
    [B, 16, (frames-1)/4+1, H/8, W/8], `timestep = sigma * 1000`, a conditional and an unconditional forward per step,
    `noise = uncond + guidance * (cond - uncond)`, flow-match Euler update (shift 5 for 720p-class outputs)."""

    def __init__(self, transformer, seed: int = 0, cfg_pair=None):
        self.transformer = transformer
        self.seed = seed
        self.cfg_pair = cfg_pair        # distributed.CFGPair: cond / uncond forward on two GPUs, one all-gather per step
        self.scheduler_step_index = None
        self.num_steps = 0
        self._inputs_key, self._inputs = None, None

    def prepare(self, batch, height, width, num_frames, max_sequence_length, seed):
        key = (batch, height, width, num_frames, max_sequence_length, seed)
        if key != self._inputs_key:
            cfg = self.transformer.model_config_dict
            lf = (num_frames - 1) // 4 + 1
            pos = synthetic.wan_step_inputs(batch, lf, height // 8, width // 8, max_sequence_length, cfg["in_channels"], cfg["text_dim"], seed=seed)
            neg = synthetic.wan_step_inputs(batch, lf, height // 8, width // 8, max_sequence_length, cfg["in_channels"], cfg["text_dim"],
                                            seed=seed + 7919)
            pos["negative_encoder_hidden_states"] = neg["encoder_hidden_states"]
            n_out = cfg.get("out_channels", cfg["in_channels"])
            if cfg["in_channels"] > n_out:
                # image-to-video (Wan2.2-I2V-A14B: 36 = 16 + 20 input channels, no image embedder): what WanImageToVideoPipeline feeds the
                # transformer is cat([latents, condition], dim=1), condition = 4 first-frame mask channels + 16 channels of the encoded
                # source image.  Synthetic: the latents are the first 16 channels of the generated tensor, the rest is the (fixed) condition.
                full = pos["hidden_states"]
                pos["hidden_states"], pos["condition"] = full[:, :n_out].contiguous(), full[:, n_out:].contiguous()
            self._inputs, self._inputs_key = pos, key
        return self._inputs

    def _denoiser(self, latents, inp, t, enc):
        x = latents if "condition" not in inp else torch.cat([latents, inp["condition"]], dim=1)
        return self.transformer(x, timestep=t, encoder_hidden_states=enc, return_dict=False)[0]

    @staticmethod
    def sigmas(num_steps: int, shift: float = 5.0):
        s = np.linspace(1.0, 1.0 / num_steps, num_steps)
        s = shift * s / (1.0 + (shift - 1.0) * s)
        return np.concatenate([s, [0.0]]).astype(np.float32)

    @torch.no_grad()
    def step(self, latents, inp, sigma, sigma_next, guidance_scale):
        t = torch.full((latents.shape[0],), float(sigma) * 1000.0, device=latents.device, dtype=torch.float32)
        if self.cfg_pair is not None and guidance_scale is not None and guidance_scale > 1:
            name, key = ("cond", "encoder_hidden_states") if self.cfg_pair.role == 0 else ("uncond", "negative_encoder_hidden_states")
            self.cfg_pair.pin_cache(self.transformer)
            with self.transformer.cache_context(name):
                mine = self._denoiser(latents, inp, t, inp[key])
            cond, uncond = self.cfg_pair.exchange(mine)
            noise = uncond.float() + guidance_scale * (cond.float() - uncond.float())
            return (latents.float() + float(sigma_next - sigma) * noise).to(latents.dtype)
        with self.transformer.cache_context("cond"):
            cond = self._denoiser(latents, inp, t, inp["encoder_hidden_states"])
        if guidance_scale is not None and guidance_scale > 1:
            with self.transformer.cache_context("uncond"):
                uncond = self._denoiser(latents, inp, t, inp["negative_encoder_hidden_states"])
            noise = uncond.float() + guidance_scale * (cond.float() - uncond.float())
        else:
            noise = cond.float()
        return (latents.float() + float(sigma_next - sigma) * noise).to(latents.dtype)

    @torch.no_grad()
    def __call__(self, num_inference_steps=40, guidance_scale=5.0, width=1280, height=768, num_frames=81, max_sequence_length=512, batch=1,
                 seed=None, latents=None, **_):
        seed = self.seed if seed is None else seed
        inp = self.prepare(batch, height, width, num_frames, max_sequence_length, seed)
        lat = inp["hidden_states"].clone() if latents is None else latents
        sig = self.sigmas(num_inference_steps)
        self.num_steps = num_inference_steps
        for i in range(num_inference_steps):
            self.scheduler_step_index = i
            lat = self.step(lat, inp, sig[i], sig[i + 1], guidance_scale)
        self.scheduler_step_index = None
        return lat


class SyntheticSD3Pipeline:
    """What diffusers' StableDiffusion3Pipeline does around `pipe.transformer`: latents [B, 16, H/8, W/8], classifier-free
    guidance as ONE batch-2 forward ([uncond ; cond] concatenated along the batch), `timestep = sigma * 1000`, flow-match Euler
    update with shift 3."""

    def __init__(self, transformer, seed: int = 0, cfg_pair=None):
        self.transformer = transformer
        self.seed = seed
        self.cfg_pair = cfg_pair        # distributed.CFGPair: role 0 runs the cond row, role 1 the uncond row, as batch-1 forwards
        self.scheduler_step_index = None
        self.num_steps = 0

    def prepare(self, height, width, max_sequence_length, seed):
        c = self.transformer.config
        return synthetic.sd3_step_inputs(2, height // 8, width // 8, max_sequence_length, c.in_channels, c.joint_attention_dim,
                                         c.pooled_projection_dim, seed=seed)

    @staticmethod
    def sigmas(num_steps: int, shift: float = 3.0):
        s = np.linspace(1.0, 1.0 / num_steps, num_steps)
        s = shift * s / (1.0 + (shift - 1.0) * s)
        return np.concatenate([s, [0.0]]).astype(np.float32)

    @torch.no_grad()
    def step(self, latents, inp, sigma, sigma_next, guidance_scale):
        """latents [1, C, h, w] -> next latents; rows 0 / 1 of the prompt tensors are the unconditional / conditional prompt."""
        if self.cfg_pair is not None:
            if getattr(self.transformer.core_model, "cache", None) is not None:
                # the single-GPU loop feeds the step cache ONE batch-2 tensor; a pair member would decide on its own row only
                raise ValueError("cfg_pair with a step cache is not defined for the batch-2 CFG families (sd35 / sdxl)")
            r = 1 - self.cfg_pair.role                                   # prompt row: 1 = conditional, 0 = unconditional
            t = torch.full((1,), float(sigma) * 1000.0, device=latents.device, dtype=torch.float32)
            mine = self.transformer(hidden_states=latents.contiguous(), encoder_hidden_states=inp["encoder_hidden_states"][r:r + 1],
                                    pooled_projections=inp["pooled_projections"][r:r + 1], timestep=t, return_dict=False)[0]
            cond, uncond = self.cfg_pair.exchange(mine)
            v = torch.cat([uncond, cond], dim=0).float()
        else:
            t = torch.full((2,), float(sigma) * 1000.0, device=latents.device, dtype=torch.float32)
            v = self.transformer(hidden_states=latents.expand(2, -1, -1, -1).contiguous(), encoder_hidden_states=inp["encoder_hidden_states"],
                                 pooled_projections=inp["pooled_projections"], timestep=t, return_dict=False)[0].float()
        noise = v[0:1] + guidance_scale * (v[1:2] - v[0:1])
        return (latents.float() + float(sigma_next - sigma) * noise).to(latents.dtype)

    @torch.no_grad()
    def __call__(self, num_inference_steps=28, guidance_scale=4.5, width=1024, height=1024, max_sequence_length=333, seed=None, latents=None, **_):
        seed = self.seed if seed is None else seed
        inp = self.prepare(height, width, max_sequence_length, seed)
        lat = inp["hidden_states"][0:1].clone() if latents is None else latents
        sig = self.sigmas(num_inference_steps)
        self.num_steps = num_inference_steps
        for i in range(num_inference_steps):
            self.scheduler_step_index = i
            lat = self.step(lat, inp, sig[i], sig[i + 1], guidance_scale)
        self.scheduler_step_index = None
        return lat


class SyntheticSDXLPipeline:
    """What diffusers' StableDiffusionXLPipeline does around `pipe.unet`: latents [B, 4, H/8, W/8], classifier-free guidance
    as one batch-2 UNet forward, epsilon prediction on a linearly spaced discrete-timestep grid with an Euler update in sigma
    space (sigma_t from the scaled-linear beta schedule, as EulerDiscreteScheduler does).  `use_graph=True` replays the UNet
    from a HIP graph (fastdm_amd/graph.py): the step is launch-bound."""

    def __init__(self, unet, seed: int = 0, use_graph: bool = False, cfg_pair=None):
        self.unet = unet
        self.seed = seed
        self.use_graph = use_graph
        self.cfg_pair = cfg_pair        # distributed.CFGPair: each rank runs its prompt row as a batch-1 UNet forward
        self.scheduler_step_index = None
        self.num_steps = 0
        self._graph_key, self._graphed = None, None

    @staticmethod
    def schedule(num_steps: int):
        betas = np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=np.float64) ** 2
        alphas = np.cumprod(1.0 - betas)
        sig_all = ((1.0 - alphas) / alphas) ** 0.5
        ts = np.linspace(999, 0, num_steps).round().astype(np.int64)
        return ts, np.concatenate([sig_all[ts], [0.0]]).astype(np.float32)

    def prepare(self, height, width, max_sequence_length, seed):
        return synthetic.sdxl_step_inputs(2, height // 8, width // 8, max_sequence_length, self.unet.config.in_channels, seed=seed)

    def _forward(self, sample, t, inp):
        fn = lambda s, tt: self.unet(s, tt, inp["encoder_hidden_states"], inp["added_cond_kwargs"])[0]
        if not self.use_graph:
            return fn(sample, t)
        key = (tuple(sample.shape), id(inp))
        if key != self._graph_key:
            from fastdm_amd.graph import GraphedForward
            self._graphed, self._graph_key = GraphedForward(fn, sample, t), key
        return self._graphed(sample, t)

    @torch.no_grad()
    def step(self, latents, inp, timestep, sigma, sigma_next, guidance_scale):
        x = (latents.float() / ((sigma ** 2 + 1.0) ** 0.5)).to(latents.dtype)              # scale_model_input
        if self.cfg_pair is not None:
            r = 1 - self.cfg_pair.role                                   # prompt row: 1 = conditional, 0 = unconditional
            if getattr(self, "_row_inp", None) is None or self._row_inp[0] is not inp:
                self._row_inp = (inp, {"encoder_hidden_states": inp["encoder_hidden_states"][r:r + 1].contiguous(),
                                       "added_cond_kwargs": {k: v[r:r + 1].contiguous() for k, v in inp["added_cond_kwargs"].items()}})
            mine = self._forward(x.contiguous(), torch.tensor(float(timestep), device=latents.device), self._row_inp[1])
            cond, uncond = self.cfg_pair.exchange(mine)
            eps = torch.cat([uncond, cond], dim=0).float()
            noise = eps[0:1] + guidance_scale * (eps[1:2] - eps[0:1])
            return (latents.float() + float(sigma_next - sigma) * noise).to(latents.dtype)
        eps = self._forward(x.expand(2, -1, -1, -1).contiguous(), torch.tensor(float(timestep), device=latents.device), inp).float()
        noise = eps[0:1] + guidance_scale * (eps[1:2] - eps[0:1])
        return (latents.float() + float(sigma_next - sigma) * noise).to(latents.dtype)

    @torch.no_grad()
    def __call__(self, num_inference_steps=30, guidance_scale=5.0, width=1024, height=1024, max_sequence_length=77, seed=None, latents=None, **_):
        seed = self.seed if seed is None else seed
        inp = self.prepare(height, width, max_sequence_length, seed)
        ts, sig = self.schedule(num_inference_steps)
        lat = (inp["sample"][0:1].float() * float(sig[0])).to(torch.bfloat16) if latents is None else latents    # init_noise_sigma
        self.num_steps = num_inference_steps
        for i in range(num_inference_steps):
            self.scheduler_step_index = i
            lat = self.step(lat, inp, ts[i], float(sig[i]), float(sig[i + 1]), guidance_scale)
        self.scheduler_step_index = None
        return lat
