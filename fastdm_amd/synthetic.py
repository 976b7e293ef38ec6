"""Synthetic FLUX weights and step inputs generated ON THE GPU by the library's counter-based
generator (fdm_synth_fill_bf16), bit-identical to the numpy twin the oracle uses (oracle/synth.py).

There are no checkpoints, text encoders or VAEs on the benchmark machines; the recipe follows
SURVEY.md §8d: Linear W ~ N(0, 1/fan_in), AdaLN/modulation linears and all biases N(0, 0.02^2),
RMS-norm weights 1 + 0.1 N(0,1), inputs U[0,1) as in examples/profiling/flux_profiling.py:16-23.
"""
import math

import torch

from fastdm_amd.kernel import fused

_IH_STD = math.sqrt(4.0 * (65536.0 ** 2 - 1.0) / 12.0)


def key_seed(name: str, base_seed: int = 0) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    h ^= (base_seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    return int((h ^ (h >> 33)) & 0x7FFFFFFF)


def _f32(x: float) -> float:
    return float(torch.tensor(x, dtype=torch.float64).to(torch.float32))


def normal(name, shape, std, offset=0.0, seed=0, device="cuda"):
    out = torch.empty(shape, device=device, dtype=torch.bfloat16)
    return fused.synth_fill(out, key_seed(name, seed), 0, _f32(std / _IH_STD), _f32(offset))


def uniform(name, shape, seed=0, device="cuda"):
    out = torch.empty(shape, device=device, dtype=torch.bfloat16)
    return fused.synth_fill(out, key_seed(name, seed), 1)


class SyntheticFluxStateDict(dict):
    """A lazily generated diffusers-named FLUX state_dict: tensors are produced on the GPU when the
    loader asks for them, so the full 24 GB bf16 checkpoint never exists at once."""

    def __init__(self, in_channels=64, out_channels=64, num_layers=19, num_single_layers=38, attention_head_dim=128,
                 num_attention_heads=24, joint_attention_dim=4096, pooled_projection_dim=768, mlp_ratio=4.0, seed=0, device="cuda"):
        super().__init__()
        self.seed, self.device = seed, device
        D, mlp, hd = attention_head_dim * num_attention_heads, int(attention_head_dim * num_attention_heads * mlp_ratio), attention_head_dim
        spec = {}

        def lin(name, k, n, kind="w"):
            spec[f"{name}.weight"] = ((n, k), kind)
            spec[f"{name}.bias"] = ((n,), "b")

        for p in ("timestep_embedder", "guidance_embedder"):
            lin(f"time_text_embed.{p}.linear_1", 256, D)
            lin(f"time_text_embed.{p}.linear_2", D, D)
        lin("time_text_embed.text_embedder.linear_1", pooled_projection_dim, D)
        lin("time_text_embed.text_embedder.linear_2", D, D)
        lin("context_embedder", joint_attention_dim, D)
        lin("x_embedder", in_channels, D)
        for i in range(num_layers):
            p = f"transformer_blocks.{i}"
            lin(f"{p}.norm1.linear", D, 6 * D, "w_mod")
            lin(f"{p}.norm1_context.linear", D, 6 * D, "w_mod")
            for n in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0", "to_add_out"):
                lin(f"{p}.attn.{n}", D, D)
            for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
                spec[f"{p}.attn.{n}.weight"] = ((hd,), "rms")
            for ff in ("ff", "ff_context"):
                lin(f"{p}.{ff}.net.0.proj", D, mlp)
                lin(f"{p}.{ff}.net.2", mlp, D)
        for i in range(num_single_layers):
            p = f"single_transformer_blocks.{i}"
            lin(f"{p}.norm.linear", D, 3 * D, "w_mod")
            lin(f"{p}.proj_mlp", D, mlp)
            lin(f"{p}.proj_out", D + mlp, D)
            for n in ("to_q", "to_k", "to_v"):
                lin(f"{p}.attn.{n}", D, D)
            for n in ("norm_q", "norm_k"):
                spec[f"{p}.attn.{n}.weight"] = ((hd,), "rms")
        lin("norm_out.linear", D, 2 * D, "w_mod")
        lin("proj_out", D, out_channels)
        self.spec = spec
        for k in spec:
            dict.__setitem__(self, k, None)

    def _make(self, name):
        shape, kind = self.spec[name]
        if kind == "w":
            return normal(name, shape, (1.0 / shape[1]) ** 0.5, 0.0, self.seed, self.device)
        if kind in ("w_mod", "b"):
            return normal(name, shape, 0.02, 0.0, self.seed, self.device)
        return normal(name, shape, 0.1, 1.0, self.seed, self.device)

    def __getitem__(self, name):
        return self._make(name)

    def get(self, name, default=None):
        return self._make(name) if name in self.spec else default


def flux_step_inputs(batch, img_h, img_w, n_txt=512, in_channels=64, joint_attention_dim=4096, pooled_projection_dim=768, seed=0,
                     device="cuda"):
    """Same tensors as oracle.flux.synth_inputs (bit-identical), created on the GPU."""
    n_img = img_h * img_w
    ids = torch.zeros(n_img, 3, device=device)
    ids[:, 1] = torch.arange(img_h, device=device).repeat_interleave(img_w).float()
    ids[:, 2] = torch.arange(img_w, device=device).repeat(img_h).float()
    return dict(
        hidden_states=uniform("input.latents", (batch, n_img, in_channels), seed, device),
        encoder_hidden_states=uniform("input.prompt_embeds", (batch, n_txt, joint_attention_dim), seed, device),
        pooled_projections=uniform("input.pooled", (batch, pooled_projection_dim), seed, device),
        img_ids=ids,
        txt_ids=torch.zeros(n_txt, 3, device=device),
    )


class SyntheticQwenStateDict(dict):
    """Lazily generated diffusers-named Qwen-Image state_dict (same recipe and key hashing as oracle.qwen.synth_state_dict,
    so the tensors are bit-identical to the oracle's)."""

    def __init__(self, patch_size=2, in_channels=64, out_channels=16, num_layers=60, attention_head_dim=128, num_attention_heads=24,
                 joint_attention_dim=3584, mlp_ratio=4.0, seed=0, device="cuda"):
        super().__init__()
        self.seed, self.device = seed, device
        D, mlp, hd = attention_head_dim * num_attention_heads, int(attention_head_dim * num_attention_heads * mlp_ratio), attention_head_dim
        spec = {}

        def lin(name, k, n, kind="w"):
            spec[f"{name}.weight"] = ((n, k), kind)
            spec[f"{name}.bias"] = ((n,), "b")

        lin("time_text_embed.timestep_embedder.linear_1", 256, D)
        lin("time_text_embed.timestep_embedder.linear_2", D, D)
        spec["txt_norm.weight"] = ((joint_attention_dim,), "rms")
        lin("img_in", in_channels, D)
        lin("txt_in", joint_attention_dim, D)
        for i in range(num_layers):
            p = f"transformer_blocks.{i}"
            lin(f"{p}.img_mod.1", D, 6 * D, "w_mod")
            lin(f"{p}.txt_mod.1", D, 6 * D, "w_mod")
            for n in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0", "to_add_out"):
                lin(f"{p}.attn.{n}", D, D)
            for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
                spec[f"{p}.attn.{n}.weight"] = ((hd,), "rms")
            for ff in ("img_mlp", "txt_mlp"):
                lin(f"{p}.{ff}.net.0.proj", D, mlp)
                lin(f"{p}.{ff}.net.2", mlp, D)
        lin("norm_out.linear", D, 2 * D, "w_mod")
        lin("proj_out", D, patch_size * patch_size * out_channels)
        self.spec = spec
        for k in spec:
            dict.__setitem__(self, k, None)

    def _make(self, name):
        shape, kind = self.spec[name]
        key = "qwen." + name
        if kind == "w":
            return normal(key, shape, (1.0 / shape[1]) ** 0.5, 0.0, self.seed, self.device)
        if kind in ("w_mod", "b"):
            return normal(key, shape, 0.02, 0.0, self.seed, self.device)
        return normal(key, shape, 0.1, 1.0, self.seed, self.device)

    def __getitem__(self, name):
        return self._make(name)

    def get(self, name, default=None):
        return self._make(name) if name in self.spec else default


def qwen_step_inputs(batch, img_h, img_w, n_txt=128, in_channels=64, joint_attention_dim=3584, seed=0, device="cuda"):
    """Same tensors as oracle.qwen.synth_inputs (bit-identical), created on the GPU (examples/profiling/qwenimg_profiling.py:31-37)."""
    return dict(
        hidden_states=uniform("qwen.input.latents", (batch, img_h * img_w, in_channels), seed, device),
        encoder_hidden_states=uniform("qwen.input.prompt_embeds", (batch, n_txt, joint_attention_dim), seed, device),
        encoder_hidden_states_mask=torch.ones(batch, n_txt, dtype=torch.int64, device=device),
        img_shapes=[[(1, img_h, img_w)]] * batch,
        txt_seq_lens=[n_txt] * batch,
    )


class SyntheticWanStateDict(dict):
    """Lazily generated diffusers-named Wan state_dict (T2V, or Wan2.1-I2V with image_dim / added_kv_proj_dim), bit-identical
    to oracle.wan.synth_state_dict."""

    def __init__(self, patch_size=(1, 2, 2), num_attention_heads=40, attention_head_dim=128, in_channels=16, out_channels=16, text_dim=4096,
                 freq_dim=256, ffn_dim=13824, num_layers=40, image_dim=None, added_kv_proj_dim=None, seed=0, device="cuda", **_):
        super().__init__()
        self.seed, self.device = seed, device
        D = num_attention_heads * attention_head_dim
        pt, ph, pw = patch_size
        spec = {}

        def lin(name, k, n, kind="w"):
            spec[f"{name}.weight"] = ((n, k), kind)
            spec[f"{name}.bias"] = ((n,), "b")

        spec["patch_embedding.weight"] = ((D, in_channels, pt, ph, pw), "conv")
        spec["patch_embedding.bias"] = ((D,), "b")
        lin("condition_embedder.time_embedder.linear_1", freq_dim, D)
        lin("condition_embedder.time_embedder.linear_2", D, D)
        lin("condition_embedder.time_proj", D, 6 * D, "w_mod")
        lin("condition_embedder.text_embedder.linear_1", text_dim, D)
        lin("condition_embedder.text_embedder.linear_2", D, D)
        if image_dim is not None:
            ie = "condition_embedder.image_embedder"
            spec[f"{ie}.norm1.weight"], spec[f"{ie}.norm1.bias"] = ((image_dim,), "rms"), ((image_dim,), "b")
            lin(f"{ie}.ff.net.0.proj", image_dim, image_dim)
            lin(f"{ie}.ff.net.2", image_dim, D)
            spec[f"{ie}.norm2.weight"], spec[f"{ie}.norm2.bias"] = ((D,), "rms"), ((D,), "b")
        for i in range(num_layers):
            p = f"blocks.{i}"
            for a in ("attn1", "attn2"):
                for n in ("to_q", "to_k", "to_v", "to_out.0"):
                    lin(f"{p}.{a}.{n}", D, D)
                spec[f"{p}.{a}.norm_q.weight"] = ((D,), "rms")
                spec[f"{p}.{a}.norm_k.weight"] = ((D,), "rms")
            if added_kv_proj_dim is not None:
                lin(f"{p}.attn2.add_k_proj", added_kv_proj_dim, D)
                lin(f"{p}.attn2.add_v_proj", added_kv_proj_dim, D)
                spec[f"{p}.attn2.norm_added_k.weight"] = ((D,), "rms")
            spec[f"{p}.norm2.weight"] = ((D,), "rms")
            spec[f"{p}.norm2.bias"] = ((D,), "b")
            lin(f"{p}.ffn.net.0.proj", D, ffn_dim)
            lin(f"{p}.ffn.net.2", ffn_dim, D)
            spec[f"{p}.scale_shift_table"] = ((1, 6, D), "table")
        lin("proj_out", D, out_channels * pt * ph * pw)
        spec["scale_shift_table"] = ((1, 2, D), "table")
        self.spec = spec
        for k in spec:
            dict.__setitem__(self, k, None)

    def _make(self, name):
        shape, kind = self.spec[name]
        key = "wan." + name
        if kind == "w":
            return normal(key, shape, (1.0 / shape[1]) ** 0.5, 0.0, self.seed, self.device)
        if kind == "conv":
            return normal(key, shape, (1.0 / math.prod(shape[1:])) ** 0.5, 0.0, self.seed, self.device)
        if kind == "table":
            return normal(key, shape, (1.0 / shape[-1]) ** 0.5, 0.0, self.seed, self.device)
        if kind in ("w_mod", "b"):
            return normal(key, shape, 0.02, 0.0, self.seed, self.device)
        return normal(key, shape, 0.1, 1.0, self.seed, self.device)

    def __getitem__(self, name):
        return self._make(name)

    def get(self, name, default=None):
        return self._make(name) if name in self.spec else default


def wan_step_inputs(batch, frames, height, width, n_txt=512, in_channels=16, text_dim=4096, seed=0, device="cuda"):
    """Same tensors as oracle.wan.synth_inputs (bit-identical): latents [B, C, F, H, W] ~ N(0,1), text [B, n_txt, text_dim]."""
    return dict(hidden_states=normal("wan.input.latents", (batch, in_channels, frames, height, width), 1.0, 0.0, seed, device),
                encoder_hidden_states=normal("wan.input.prompt_embeds", (batch, n_txt, text_dim), 1.0, 0.0, seed, device))


class SyntheticSD3StateDict(dict):
    """Lazily generated diffusers-named SD3.5 state_dict, bit-identical to oracle.sd35.synth_state_dict."""

    def __init__(self, patch_size=2, in_channels=16, num_layers=38, attention_head_dim=64, num_attention_heads=38, joint_attention_dim=4096,
                 caption_projection_dim=2432, pooled_projection_dim=2048, out_channels=16, pos_embed_max_size=192, dual_attention_layers=(),
                 mlp_ratio=4.0, seed=0, device="cuda", **_):
        super().__init__()
        self.seed, self.device = seed, device
        D, hd, p = attention_head_dim * num_attention_heads, attention_head_dim, patch_size
        mlp = int(D * mlp_ratio)
        spec = {}

        def lin(name, k, n, kind="w"):
            spec[f"{name}.weight"] = ((n, k), kind)
            spec[f"{name}.bias"] = ((n,), "b")

        spec["pos_embed.proj.weight"] = ((D, in_channels, p, p), "conv")
        spec["pos_embed.proj.bias"] = ((D,), "b")
        spec["pos_embed.pos_embed"] = ((1, pos_embed_max_size ** 2, D), "b")
        lin("time_text_embed.timestep_embedder.linear_1", 256, D)
        lin("time_text_embed.timestep_embedder.linear_2", D, D)
        lin("time_text_embed.text_embedder.linear_1", pooled_projection_dim, D)
        lin("time_text_embed.text_embedder.linear_2", D, D)
        lin("context_embedder", joint_attention_dim, caption_projection_dim)
        for i in range(num_layers):
            b = f"transformer_blocks.{i}"
            last, dual = i == num_layers - 1, i in dual_attention_layers
            lin(f"{b}.norm1.linear", D, (9 if dual else 6) * D, "w_mod")
            lin(f"{b}.norm1_context.linear", D, (2 if last else 6) * D, "w_mod")
            for n in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0"):
                lin(f"{b}.attn.{n}", D, D)
            if not last:
                lin(f"{b}.attn.to_add_out", D, D)
            for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
                spec[f"{b}.attn.{n}.weight"] = ((hd,), "rms")
            if dual:
                for n in ("to_q", "to_k", "to_v", "to_out.0"):
                    lin(f"{b}.attn2.{n}", D, D)
                spec[f"{b}.attn2.norm_q.weight"] = ((hd,), "rms")
                spec[f"{b}.attn2.norm_k.weight"] = ((hd,), "rms")
            lin(f"{b}.ff.net.0.proj", D, mlp)
            lin(f"{b}.ff.net.2", mlp, D)
            if not last:
                lin(f"{b}.ff_context.net.0.proj", D, mlp)
                lin(f"{b}.ff_context.net.2", mlp, D)
        lin("norm_out.linear", D, 2 * D, "w_mod")
        lin("proj_out", D, p * p * out_channels)
        self.spec = spec
        for k in spec:
            dict.__setitem__(self, k, None)

    def _make(self, name):
        shape, kind = self.spec[name]
        key = "sd3." + name
        if kind == "w":
            return normal(key, shape, (1.0 / shape[1]) ** 0.5, 0.0, self.seed, self.device)
        if kind == "conv":
            return normal(key, shape, (1.0 / math.prod(shape[1:])) ** 0.5, 0.0, self.seed, self.device)
        if kind in ("w_mod", "b"):
            return normal(key, shape, 0.02, 0.0, self.seed, self.device)
        return normal(key, shape, 0.1, 1.0, self.seed, self.device)

    def __getitem__(self, name):
        return self._make(name)

    def get(self, name, default=None):
        return self._make(name) if name in self.spec else default


def sd3_step_inputs(batch, height, width, n_txt=333, in_channels=16, joint_attention_dim=4096, pooled_projection_dim=2048, seed=0, device="cuda"):
    """Same tensors as oracle.sd35.synth_inputs (bit-identical); height/width are LATENT sizes (pixels / 8)."""
    return dict(hidden_states=normal("sd3.input.latents", (batch, in_channels, height, width), 1.0, 0.0, seed, device),
                encoder_hidden_states=uniform("sd3.input.prompt_embeds", (batch, n_txt, joint_attention_dim), seed, device),
                pooled_projections=uniform("sd3.input.pooled", (batch, pooled_projection_dim), seed, device))


class SyntheticSDXLStateDict(dict):
    """Lazily generated diffusers-named SDXL UNet state_dict, bit-identical to oracle.sdxl.synth_state_dict (the key
    inventory follows the loaders of fastdm/model/sdxl.py:178-373)."""

    def __init__(self, in_channels=4, block_out_channels=(320, 640, 1280), transformer_layers=(2, 10), mid_layers=10, cross_attention_dim=2048,
                 time_proj_dim=320, time_embed_dim=1280, addition_time_embed_dim=256, text_embed_dim=1280, seed=0, device="cuda", **_):
        super().__init__()
        self.seed, self.device = seed, device
        c0, c1, c2 = block_out_channels
        l1, l2 = transformer_layers
        te, x = time_embed_dim, cross_attention_dim
        spec = {}

        def lin(name, k, n, bias=True):
            spec[f"{name}.weight"] = ((n, k), "w")
            if bias:
                spec[f"{name}.bias"] = ((n,), "b")

        def norm(name, c):
            spec[f"{name}.weight"] = ((c,), "rms")
            spec[f"{name}.bias"] = ((c,), "b")

        def conv(name, cout, cin, k):
            spec[f"{name}.weight"] = ((cout, cin, k, k), "conv")
            spec[f"{name}.bias"] = ((cout,), "b")

        def resnet(name, cin, cout, shortcut):
            norm(f"{name}.norm1", cin)
            conv(f"{name}.conv1", cout, cin, 3)
            lin(f"{name}.time_emb_proj", te, cout)
            norm(f"{name}.norm2", cout)
            conv(f"{name}.conv2", cout, cout, 3)
            if shortcut:
                conv(f"{name}.conv_shortcut", cout, cin, 1)

        def transformer(name, c, layers):
            norm(f"{name}.norm", c)
            lin(f"{name}.proj_in", c, c)
            lin(f"{name}.proj_out", c, c)
            for j in range(layers):
                b = f"{name}.transformer_blocks.{j}"
                for n in ("norm1", "norm2", "norm3"):
                    norm(f"{b}.{n}", c)
                for n in ("to_q", "to_k", "to_v"):
                    lin(f"{b}.attn1.{n}", c, c, bias=False)
                lin(f"{b}.attn2.to_q", c, c, bias=False)
                lin(f"{b}.attn2.to_k", x, c, bias=False)
                lin(f"{b}.attn2.to_v", x, c, bias=False)
                lin(f"{b}.attn1.to_out.0", c, c)
                lin(f"{b}.attn2.to_out.0", c, c)
                lin(f"{b}.ff.net.0.proj", c, 8 * c)
                lin(f"{b}.ff.net.2", 4 * c, c)

        conv("conv_in", c0, in_channels, 3)
        for name, k in (("time_embedding", time_proj_dim), ("add_embedding", text_embed_dim + 6 * addition_time_embed_dim)):
            lin(f"{name}.linear_1", k, te)
            lin(f"{name}.linear_2", te, te)
        resnet("down_blocks.0.resnets.0", c0, c0, False)
        resnet("down_blocks.0.resnets.1", c0, c0, False)
        for m, (cin, cout, layers) in ((1, (c0, c1, l1)), (2, (c1, c2, l2))):
            resnet(f"down_blocks.{m}.resnets.0", cin, cout, True)
            resnet(f"down_blocks.{m}.resnets.1", cout, cout, False)
            for i in range(2):
                transformer(f"down_blocks.{m}.attentions.{i}", cout, layers)
        for m, c in ((0, c0), (1, c1)):
            conv(f"down_blocks.{m}.downsamplers.0.conv", c, c, 3)
        resnet("mid_block.resnets.0", c2, c2, False)
        resnet("mid_block.resnets.1", c2, c2, False)
        transformer("mid_block.attentions.0", c2, mid_layers)
        for m, (cin, cout, prev, layers) in ((0, (c1, c2, c2, l2)), (1, (c0, c1, c2, l1)), (2, (c0, c0, c1, 0))):
            for i, rin in enumerate((prev + cout, 2 * cout, cout + cin)):
                resnet(f"up_blocks.{m}.resnets.{i}", rin, cout, True)
            if m < 2:
                for i in range(3):
                    transformer(f"up_blocks.{m}.attentions.{i}", cout, layers)
                conv(f"up_blocks.{m}.upsamplers.0.conv", cout, cout, 3)
        norm("conv_norm_out", c0)
        conv("conv_out", in_channels, c0, 3)
        self.spec = spec
        for k in spec:
            dict.__setitem__(self, k, None)

    def _make(self, name):
        shape, kind = self.spec[name]
        key = "sdxl." + name
        if kind == "w":
            return normal(key, shape, (1.0 / shape[1]) ** 0.5, 0.0, self.seed, self.device)
        if kind == "conv":
            return normal(key, shape, (1.0 / math.prod(shape[1:])) ** 0.5, 0.0, self.seed, self.device)
        if kind == "b":
            return normal(key, shape, 0.02, 0.0, self.seed, self.device)
        return normal(key, shape, 0.1, 1.0, self.seed, self.device)

    def __getitem__(self, name):
        return self._make(name)

    def get(self, name, default=None):
        return self._make(name) if name in self.spec else default


def sdxl_step_inputs(batch, height, width, n_txt=77, in_channels=4, cross_attention_dim=2048, text_embed_dim=1280, seed=0, device="cuda"):
    """Same tensors as oracle.sdxl.synth_inputs (bit-identical); height/width are LATENT sizes (pixels / 8)."""
    ids = torch.tensor([height * 8, width * 8, 0, 0, height * 8, width * 8], dtype=torch.bfloat16, device=device).repeat(batch, 1)
    return dict(sample=normal("sdxl.input.latents", (batch, in_channels, height, width), 1.0, 0.0, seed, device),
                encoder_hidden_states=uniform("sdxl.input.prompt_embeds", (batch, n_txt, cross_attention_dim), seed, device),
                added_cond_kwargs={"text_embeds": uniform("sdxl.input.pooled", (batch, text_embed_dim), seed, device), "time_ids": ids})
