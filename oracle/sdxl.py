"""SDXL UNet oracle (torch-CPU restatement of fastdm/model/sdxl.py, fastdm/layer/unetblock.py, fastdm/layer/resnetblock.py
and Timesteps / TimestepEmbedding of fastdm/layer/embeddings.py:377-410).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Citations relative to /root/reference.

The reference core hard-codes SDXL-base's widths (320/640/1280, 2 and 10 transformer layers, sdxl.py:53-113); its blocks
take them as arguments.  This restatement is parametrised the way the blocks are, so a narrow UNet assembled from the
reference's own blocks (oracle/make_golden.py `sdxl`) pins it in seconds, and the defaults are SDXL-base.
"""
from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import layers as L
from . import ops
from . import synth
from .flux import bf16_from_bits, synth_dist


@dataclass
class SDXLConfig:
    in_channels: int = 4
    block_out_channels: Tuple[int, int, int] = (320, 640, 1280)
    transformer_layers: Tuple[int, int] = (2, 10)     # per attention of the 2nd / 3rd resolution level (sdxl.py:66,75,86,95)
    mid_layers: int = 10                              # unetblock.py:499
    cross_attention_dim: int = 2048                   # unetblock.py:270
    time_proj_dim: int = 320                          # sdxl.py:53
    time_embed_dim: int = 1280                        # sdxl.py:54, resnetblock.py:19
    addition_time_embed_dim: int = 256                # sdxl.py:55
    text_embed_dim: int = 1280                        # pooled text: 1280 + 6*256 = 2816 (sdxl.py:56)
    head_dim: int = 64                                # unetblock.py:19


TINY = SDXLConfig(block_out_channels=(32, 64, 128), transformer_layers=(1, 2), mid_layers=1)


# ------------------------------------------------------------------ parameter inventory (sdxl.py:178-373)
def _resnet(name, cin, cout, temb, shortcut):
    yield f"{name}.norm1.weight", (cin,), "rms"
    yield f"{name}.norm1.bias", (cin,), "b"
    yield f"{name}.conv1.weight", (cout, cin, 3, 3), "conv"
    yield f"{name}.conv1.bias", (cout,), "b"
    yield f"{name}.time_emb_proj.weight", (cout, temb), "w"
    yield f"{name}.time_emb_proj.bias", (cout,), "b"
    yield f"{name}.norm2.weight", (cout,), "rms"
    yield f"{name}.norm2.bias", (cout,), "b"
    yield f"{name}.conv2.weight", (cout, cout, 3, 3), "conv"
    yield f"{name}.conv2.bias", (cout,), "b"
    if shortcut:
        yield f"{name}.conv_shortcut.weight", (cout, cin, 1, 1), "conv"
        yield f"{name}.conv_shortcut.bias", (cout,), "b"


def _transformer(name, c, layers, cross):
    yield f"{name}.norm.weight", (c,), "rms"
    yield f"{name}.norm.bias", (c,), "b"
    for p in ("proj_in", "proj_out"):
        yield f"{name}.{p}.weight", (c, c), "w"
        yield f"{name}.{p}.bias", (c,), "b"
    for j in range(layers):
        b = f"{name}.transformer_blocks.{j}"
        for n in ("norm1", "norm2", "norm3"):
            yield f"{b}.{n}.weight", (c,), "rms"
            yield f"{b}.{n}.bias", (c,), "b"
        for n in ("to_q", "to_k", "to_v"):
            yield f"{b}.attn1.{n}.weight", (c, c), "w"                       # no bias (unetblock.py:33)
        yield f"{b}.attn2.to_q.weight", (c, c), "w"
        yield f"{b}.attn2.to_k.weight", (c, cross), "w"
        yield f"{b}.attn2.to_v.weight", (c, cross), "w"
        for a in ("attn1", "attn2"):
            yield f"{b}.{a}.to_out.0.weight", (c, c), "w"
            yield f"{b}.{a}.to_out.0.bias", (c,), "b"
        yield f"{b}.ff.net.0.proj.weight", (8 * c, c), "w"
        yield f"{b}.ff.net.0.proj.bias", (8 * c,), "b"
        yield f"{b}.ff.net.2.weight", (c, 4 * c), "w"
        yield f"{b}.ff.net.2.bias", (c,), "b"


def param_specs(cfg: SDXLConfig):
    """(diffusers key, shape, kind) for every tensor SDXLUNetModelCore loads."""
    c0, c1, c2 = cfg.block_out_channels
    l1, l2 = cfg.transformer_layers
    te, x = cfg.time_embed_dim, cfg.cross_attention_dim
    yield "conv_in.weight", (c0, cfg.in_channels, 3, 3), "conv"
    yield "conv_in.bias", (c0,), "b"
    for name, k in (("time_embedding", cfg.time_proj_dim), ("add_embedding", cfg.text_embed_dim + 6 * cfg.addition_time_embed_dim)):
        yield f"{name}.linear_1.weight", (te, k), "w"
        yield f"{name}.linear_1.bias", (te,), "b"
        yield f"{name}.linear_2.weight", (te, te), "w"
        yield f"{name}.linear_2.bias", (te,), "b"
    # down
    yield from _resnet("down_blocks.0.resnets.0", c0, c0, te, False)
    yield from _resnet("down_blocks.0.resnets.1", c0, c0, te, False)
    for m, (cin, cout, layers) in ((1, (c0, c1, l1)), (2, (c1, c2, l2))):
        yield from _resnet(f"down_blocks.{m}.resnets.0", cin, cout, te, True)
        yield from _resnet(f"down_blocks.{m}.resnets.1", cout, cout, te, False)
        for i in range(2):
            yield from _transformer(f"down_blocks.{m}.attentions.{i}", cout, layers, x)
    for m, c in ((0, c0), (1, c1)):
        yield f"down_blocks.{m}.downsamplers.0.conv.weight", (c, c, 3, 3), "conv"
        yield f"down_blocks.{m}.downsamplers.0.conv.bias", (c,), "b"
    # mid
    yield from _resnet("mid_block.resnets.0", c2, c2, te, False)
    yield from _resnet("mid_block.resnets.1", c2, c2, te, False)
    yield from _transformer("mid_block.attentions.0", c2, cfg.mid_layers, x)
    # up (unetblock.py:451-494: resnet inputs are cat([hidden, skip]))
    for m, (cin, cout, prev, layers) in ((0, (c1, c2, c2, l2)), (1, (c0, c1, c2, l1)), (2, (c0, c0, c1, 0))):
        for i, rin in enumerate((prev + cout, 2 * cout, cout + cin)):
            yield from _resnet(f"up_blocks.{m}.resnets.{i}", rin, cout, te, True)
        if m < 2:
            for i in range(3):
                yield from _transformer(f"up_blocks.{m}.attentions.{i}", cout, layers, x)
            yield f"up_blocks.{m}.upsamplers.0.conv.weight", (cout, cout, 3, 3), "conv"
            yield f"up_blocks.{m}.upsamplers.0.conv.bias", (cout,), "b"
    yield "conv_norm_out.weight", (c0,), "rms"
    yield "conv_norm_out.bias", (c0,), "b"
    yield "conv_out.weight", (cfg.in_channels, c0, 3, 3), "conv"
    yield "conv_out.bias", (cfg.in_channels,), "b"


def synth_tensor(name: str, shape, kind: str, seed: int = 0) -> torch.Tensor:
    if kind == "conv":
        std, off = (1.0 / int(np.prod(shape[1:]))) ** 0.5, 0.0
    else:
        std, off = synth_dist(kind, shape)
    return bf16_from_bits(synth.normal_bf16_bits(synth.key_seed("sdxl." + name, seed), int(np.prod(shape)), std, off), shape)


def synth_state_dict(cfg: SDXLConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    return {name: synth_tensor(name, shape, kind, seed) for name, shape, kind in param_specs(cfg)}


def synth_inputs(cfg: SDXLConfig, batch: int, height: int, width: int, n_txt: int = 77, seed: int = 0):
    """latents [B, 4, H, W] ~ N(0,1) (H, W = pixels / 8), prompt embeddings U[0,1) [B, n_txt, 2048], pooled U[0,1) [B, 1280],
    time_ids = (orig h, w, crop top, left, target h, w) as the SDXL pipeline builds them."""
    def make(tag, shape, normal=False):
        n = int(np.prod(shape))
        key = synth.key_seed(f"sdxl.input.{tag}", seed)
        bits = synth.normal_bf16_bits(key, n, 1.0, 0.0) if normal else synth.uniform_bf16_bits(key, n)
        return bf16_from_bits(bits, shape)

    ids = torch.tensor([height * 8, width * 8, 0, 0, height * 8, width * 8], dtype=torch.bfloat16).repeat(batch, 1)
    return dict(sample=make("latents", (batch, cfg.in_channels, height, width), True),
                encoder_hidden_states=make("prompt_embeds", (batch, n_txt, cfg.cross_attention_dim)),
                added_cond_kwargs={"text_embeds": make("pooled", (batch, cfg.text_embed_dim)), "time_ids": ids})


# ------------------------------------------------------------------ layers
class ResnetBlock2D:
    """fastdm/layer/resnetblock.py:9-50.  time_emb_proj is never quantised (sdxl.py:196)."""

    def __init__(self, sd, name, shortcut: bool):
        g = lambda k: sd[f"{name}.{k}"]
        self.n1w, self.n1b, self.c1w, self.c1b = g("norm1.weight"), g("norm1.bias"), g("conv1.weight"), g("conv1.bias")
        self.n2w, self.n2b, self.c2w, self.c2b = g("norm2.weight"), g("norm2.bias"), g("conv2.weight"), g("conv2.bias")
        self.temb = L.linear_from(sd, [f"{name}.time_emb_proj"])
        self.sw, self.sb = (g("conv_shortcut.weight"), g("conv_shortcut.bias")) if shortcut else (None, None)

    def __call__(self, x, temb):
        h = F.silu(F.group_norm(x, 32, self.n1w, self.n1b, 1e-5))
        h = F.conv2d(h, self.c1w, self.c1b, 1, 1)
        h = h + self.temb(F.silu(temb))[:, :, None, None]
        h = F.silu(F.group_norm(h, 32, self.n2w, self.n2b, 1e-5))
        h = F.conv2d(h, self.c2w, self.c2b, 1, 1)
        if self.sw is not None:
            x = F.conv2d(x, self.sw, self.sb, 1)
        return x + h


class AttentionSDXL:
    """fastdm/layer/unetblock.py:11-142 without the IP-adapter branches: attn1 = fused qkv self-attention, attn2 = q from the
    image tokens, fused kv from the prompt embeddings; 64-wide heads, no q/k norm, no RoPE."""

    def __init__(self, sd, name, cross: bool, head_dim: int, quant):
        self.cross = cross
        if cross:
            self.q = L.linear_from(sd, [f"{name}.to_q"], quant)
            self.kv = L.linear_from(sd, [f"{name}.to_k", f"{name}.to_v"], quant)
        else:
            self.qkv = L.linear_from(sd, [f"{name}.to_q", f"{name}.to_k", f"{name}.to_v"], quant)
        self.out = L.linear_from(sd, [f"{name}.to_out.0"], quant)
        self.head_dim = head_dim

    def __call__(self, x, enc, batch):
        c = x.shape[-1]
        heads = c // self.head_dim
        if self.cross:
            q = self.q(x).view(batch, -1, c)
            kv = self.kv(enc.reshape(-1, enc.shape[-1]))
            k, v = kv[:, :c].reshape(batch, -1, c), kv[:, c:].reshape(batch, -1, c)
        else:
            qkv = self.qkv(x)
            q, k, v = (qkv[:, i * c:(i + 1) * c].reshape(batch, -1, c) for i in range(3))
        o = ops.scaled_dot_product_attention(q.contiguous(), k.contiguous(), v.contiguous(), heads, heads, self.head_dim,
                                             scale=self.head_dim ** -0.5)
        return self.out(o.reshape(-1, c))


class BasicTransformerBlock:
    """fastdm/layer/unetblock.py:260-298: LN -> self-attn -> +res; LN -> cross-attn -> +res; LN -> GEGLU FF -> +res."""

    def __init__(self, sd, name, c, head_dim, quant):
        self.c = c
        self.norm = [(sd[f"{name}.norm{i}.weight"], sd[f"{name}.norm{i}.bias"]) for i in (1, 2, 3)]
        self.attn1 = AttentionSDXL(sd, f"{name}.attn1", False, head_dim, quant)
        self.attn2 = AttentionSDXL(sd, f"{name}.attn2", True, head_dim, quant)
        self.proj1 = L.linear_from(sd, [f"{name}.ff.net.0.proj"], quant)
        self.proj2 = L.linear_from(sd, [f"{name}.ff.net.2"], quant)

    def __call__(self, x, enc, batch):
        x = self.attn1(F.layer_norm(x, [self.c], *self.norm[0], eps=1e-5), None, batch) + x
        x = self.attn2(F.layer_norm(x, [self.c], *self.norm[1], eps=1e-5), enc, batch) + x
        return self.proj2(ops.gelu_and_mul(self.proj1(F.layer_norm(x, [self.c], *self.norm[2], eps=1e-5)))) + x


class Transformer2DModel:
    """fastdm/layer/unetblock.py:332-367: group-norm (eps 1e-6) -> tokens -> proj_in -> blocks -> proj_out -> NCHW + residual."""

    def __init__(self, sd, name, c, layers, head_dim, quant):
        self.nw, self.nb = sd[f"{name}.norm.weight"], sd[f"{name}.norm.bias"]
        self.proj_in = L.linear_from(sd, [f"{name}.proj_in"], quant)
        self.proj_out = L.linear_from(sd, [f"{name}.proj_out"], quant)
        self.blocks = [BasicTransformerBlock(sd, f"{name}.transformer_blocks.{j}", c, head_dim, quant) for j in range(layers)]

    def __call__(self, x, enc):
        b, c, h, w = x.shape
        t = F.group_norm(x, 32, self.nw, self.nb, eps=1e-6).permute(0, 2, 3, 1).reshape(b * h * w, c)
        t = self.proj_in(t)
        for blk in self.blocks:
            t = blk(t, enc, b)
        t = self.proj_out(t)
        return t.reshape(b, h, w, c).permute(0, 3, 1, 2).contiguous() + x


class SDXLCore:
    """fastdm/model/sdxl.py:17-470 (no IP-adapter, no ControlNet residuals).  time_embedding / add_embedding linears are
    quantised like the transformer linears (sdxl.py:181-184)."""

    def __init__(self, sd, cfg: SDXLConfig = SDXLConfig(), quant=torch.float8_e4m3fn):
        self.cfg = cfg
        hd = cfg.head_dim
        c0, c1, c2 = cfg.block_out_channels
        l1, l2 = cfg.transformer_layers
        self.sd = sd
        self.time_embedding = [L.linear_from(sd, [f"time_embedding.linear_{i}"], quant) for i in (1, 2)]
        self.add_embedding = [L.linear_from(sd, [f"add_embedding.linear_{i}"], quant) for i in (1, 2)]
        R = lambda n, sc: ResnetBlock2D(sd, n, sc)
        T = lambda n, c, l: Transformer2DModel(sd, n, c, l, hd, quant)
        self.down = [
            dict(res=[R("down_blocks.0.resnets.0", False), R("down_blocks.0.resnets.1", False)], attn=None, ds="down_blocks.0.downsamplers.0.conv"),
            dict(res=[R("down_blocks.1.resnets.0", True), R("down_blocks.1.resnets.1", False)],
                 attn=[T(f"down_blocks.1.attentions.{i}", c1, l1) for i in range(2)], ds="down_blocks.1.downsamplers.0.conv"),
            dict(res=[R("down_blocks.2.resnets.0", True), R("down_blocks.2.resnets.1", False)],
                 attn=[T(f"down_blocks.2.attentions.{i}", c2, l2) for i in range(2)], ds=None),
        ]
        self.mid_res = [R("mid_block.resnets.0", False), R("mid_block.resnets.1", False)]
        self.mid_attn = T("mid_block.attentions.0", c2, cfg.mid_layers)
        self.up = [
            dict(res=[R(f"up_blocks.0.resnets.{i}", True) for i in range(3)], attn=[T(f"up_blocks.0.attentions.{i}", c2, l2) for i in range(3)],
                 us="up_blocks.0.upsamplers.0.conv"),
            dict(res=[R(f"up_blocks.1.resnets.{i}", True) for i in range(3)], attn=[T(f"up_blocks.1.attentions.{i}", c1, l1) for i in range(3)],
                 us="up_blocks.1.upsamplers.0.conv"),
            dict(res=[R(f"up_blocks.2.resnets.{i}", True) for i in range(3)], attn=None, us=None),
        ]

    def forward(self, sample, timesteps, encoder_hidden_states, added_cond_kwargs, mid_block_additional_residual=None,
                down_block_additional_residuals=None):
        sd, cfg = self.sd, self.cfg
        timesteps = timesteps.expand(sample.shape[0])
        t_emb = L.timestep_embedding(timesteps, cfg.time_proj_dim).to(sample.dtype)                      # sdxl.py:382-383
        emb = self.time_embedding[1](F.silu(self.time_embedding[0](t_emb)))
        text_embeds, time_ids = added_cond_kwargs["text_embeds"], added_cond_kwargs["time_ids"]
        time_embeds = L.timestep_embedding(time_ids.flatten(), cfg.addition_time_embed_dim).reshape(text_embeds.shape[0], -1)
        add = torch.concat([text_embeds, time_embeds], dim=-1).to(emb.dtype)                            # sdxl.py:397-399
        emb = emb + self.add_embedding[1](F.silu(self.add_embedding[0](add)))
        x = F.conv2d(sample, sd["conv_in.weight"], sd["conv_in.bias"], 1, 1)
        skips = [x]
        for blk in self.down:                                                                             # unetblock.py:383-441
            for i, r in enumerate(blk["res"]):
                x = r(x, emb)
                if blk["attn"] is not None:
                    x = blk["attn"][i](x, encoder_hidden_states)
                skips.append(x)
            if blk["ds"] is not None:
                x = F.conv2d(x, sd[blk["ds"] + ".weight"], sd[blk["ds"] + ".bias"], 2, 1)
                skips.append(x)
        if mid_block_additional_residual is not None and down_block_additional_residuals is not None:    # sdxl.py:405,427-432
            skips = [sk + r for sk, r in zip(skips, down_block_additional_residuals)]
            x = skips[-1]           # upstream adds IN PLACE and the running sample is the last skip tensor (no downsampler on level 3)
        x = self.mid_res[0](x, emb)                                                                       # unetblock.py:504-516
        x = self.mid_attn(x, encoder_hidden_states)
        x = self.mid_res[1](x, emb)
        if mid_block_additional_residual is not None and down_block_additional_residuals is not None:    # sdxl.py:439-440
            x = x + mid_block_additional_residual
        for blk in self.up:                                                                               # unetblock.py:462-494
            for i, r in enumerate(blk["res"]):
                x = r(torch.cat([x, skips.pop()], dim=1), emb)
                if blk["attn"] is not None:
                    x = blk["attn"][i](x, encoder_hidden_states)
            if blk["us"] is not None:
                x = F.interpolate(x, scale_factor=2.0, mode="nearest")
                x = F.conv2d(x, sd[blk["us"] + ".weight"], sd[blk["us"] + ".bias"], 1, 1)
        x = F.silu(F.group_norm(x, 32, sd["conv_norm_out.weight"], sd["conv_norm_out.bias"], eps=1e-5))
        return [F.conv2d(x, sd["conv_out.weight"], sd["conv_out.bias"], 1, 1)]


SDXL_TAPS = ("down0", "down1", "down2", "mid", "up0", "up1", "up2")


def sdxl_block_taps(core, store, sub=(4, 2, 2)):
    """Record the (sub-sampled) output of each UNet stage of one forward: down_blocks[0..2], mid_block, up_blocks[0..2] -- the same
    attribute names in the reference's SDXLUNetModelCore and in fastdm_amd's (tests/test_sdxl_gpu.py uses this function on both).
    Returns a restore() callable."""
    saved = []

    def wrap(obj, name):
        orig = obj.forward

        def fwd(*a, **k):
            out = orig(*a, **k)
            t = out[0] if isinstance(out, (tuple, list)) else out
            store[name] = t.detach()[:, ::sub[0], ::sub[1], ::sub[2]].float().cpu().to(torch.bfloat16).contiguous()
            return out
        obj.forward = fwd
        saved.append((obj, orig))
    for i in range(3):
        wrap(core.down_blocks[i], f"down{i}")
    wrap(core.mid_block, "mid")
    for i in range(3):
        wrap(core.up_blocks[i], f"up{i}")

    def restore():
        for obj, orig in saved:
            obj.forward = orig
    return restore
