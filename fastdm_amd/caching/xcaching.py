"""TeaCache step skipping, host side (mirror of fastdm/caching/xcaching.py:9-276 for FLUX).

The decision stays on the host as the north star asks: one device-side reduction
(fdm_absdiff_sums) produces sum|mod - prev| and sum|prev|; two floats cross PCIe (the reference's
`.cpu().item()` sync, xcaching.py:214-215); the polynomial rescale and threshold test are Python.
The reference evaluates the ratio on bf16 tensors, so the means and the quotient are rounded to
bf16 here as well before they reach the polynomial.
"""
from typing import Any, Dict, Optional, Type

import numpy as np
import torch

from fastdm_amd.caching.config import CacheConfig
from fastdm_amd.kernel import fused


def _bf16(x: float) -> float:
    return float(torch.tensor(x, dtype=torch.float32).to(torch.bfloat16).float())


class AutoCache:
    _registry: Dict[str, Type["AutoCache"]] = {}

    def __init__(self, config: CacheConfig):
        self.config = config
        self.accumulated_rel_l1_distance_dict = {"positive": 0.0, "negative": 0.0}
        self.previous_modulated_input_dict = {"positive": None, "negative": None}
        self.previous_residual_dict = {"positive": None, "negative": None}
        self.cache_status = {"positive": True, "negative": False}
        self.trace = []   # (step, rel_distance | None, accumulated, should_calc): for tests / logging
        self.pinned_key = None

    def pin_key(self, key):
        """CFG-pair sharding (fastdm_amd/distributed.py): this process only ever runs ONE branch of the cond / uncond
        pair, so the positive / negative entry is fixed instead of alternating call by call -- the rank that runs the
        unconditional forwards pins 'negative' and then sees exactly the state sequence the single-GPU loop keeps under
        that key (its coefficients included)."""
        if key not in (None, "positive", "negative"):
            raise ValueError(f"unknown cache key {key!r}")
        self.pinned_key = key

    @classmethod
    def register(cls, name: str):
        def decorator(sub_cls):
            cls._registry[name.lower()] = sub_cls
            return sub_cls
        return decorator

    @classmethod
    def _make(cls, config: CacheConfig) -> "AutoCache":
        cache_cls = cls._registry.get(config.cache_algorithm.lower())
        if cache_cls is None:
            raise ValueError(f"Unknown cache algorithm: {config.cache_algorithm.lower()}")
        return cache_cls(config)

    @classmethod
    def from_dict(cls, data: Dict[str, Any]) -> "AutoCache":
        return cls._make(CacheConfig.from_dict(data))

    @classmethod
    def from_json(cls, file_path: str) -> "AutoCache":
        return cls._make(CacheConfig.from_json(file_path))

    def get_current_step(self):
        cb = self.config.current_steps_callback
        step = cb() if cb is not None else None
        return step if step is not None else 0

    def get_cache_key(self):
        """xcaching.py:66-76: alternate positive/negative on successive calls when negtive_cache is set."""
        if not self.config.negtive_cache:
            return "positive"
        if self.pinned_key is not None:
            return self.pinned_key
        cache_key = None
        for k in self.cache_status:
            if self.cache_status[k] and cache_key is None:
                cache_key = k
            self.cache_status[k] = not self.cache_status[k]
        return cache_key

    def rel_distance(self, modulated, previous) -> float:
        """mean|a - b| / mean|b| with the reference's bf16 roundings (xcaching.py:214-215,343-344); one D2H sync."""
        sums = fused.absdiff_sums(modulated.contiguous(), previous.contiguous()).cpu()
        n = float(modulated.numel())
        mean_diff, mean_prev = _bf16(float(sums[0]) / n), _bf16(float(sums[1]) / n)
        return _bf16(mean_diff / mean_prev)

    def apply_cache(self, *args, **kwargs):
        raise NotImplementedError("Subclasses must implement apply_cache method")


def _block_forward(model_type, block, hidden_states, encoder_hidden_states, temb, image_rotary_emb, attention_kwargs, encoder_mask=None):
    """One transformer block with the argument list its model family uses (xcaching.py:98-136,284-321): returns
    (encoder_hidden_states, hidden_states); Wan blocks have no text-stream output."""
    if model_type == "wan":
        return None, block.forward(hidden_states, encoder_hidden_states, temb, image_rotary_emb)
    kw = dict(hidden_states=hidden_states, encoder_hidden_states=encoder_hidden_states, temb=temb, joint_attention_kwargs=attention_kwargs)
    if model_type != "sd35":
        kw["image_rotary_emb"] = image_rotary_emb
    if model_type == "qwenimage":
        kw["encoder_hidden_states_mask"] = encoder_mask
    return block.forward(**kw)


@AutoCache.register("teacache")
class TeaCache(AutoCache):
    def __init__(self, config):
        super().__init__(config)
        self.coefficients = {"positive": config.coefficients, "negative": config.negtive_coefficients}

    def _get_modulated_input(self, model_type, hidden_states, temb, transformer_blocks, encoder_hidden_states=None):
        """xcaching.py:165-183: FLUX probes the first block's norm1 output of the IMAGE stream, Qwen-Image the
        modulated TEXT stream of block 0 (bf16 [B,S,D])."""
        if model_type == "qwenimage":
            return transformer_blocks[0].txt_modulated(encoder_hidden_states, temb)
        if model_type not in ("flux", "sd35"):
            raise ValueError(f"TeaCache unsupported model type: {model_type}. Supported types: qwenimage, flux, sd35")
        return transformer_blocks[0].norm1.forward(hidden_states, emb=temb)[0]

    def apply_cache(self, model_type: str, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor = None,
                    temb: torch.Tensor = None, image_rotary_emb: torch.Tensor = None, attention_kwargs: Optional[Dict[str, Any]] = None,
                    transformer_blocks=None, single_transformer_blocks=None, core=None, **_):
        """xcaching.py:187-276 (FLUX, Qwen-Image and SD3.5 branches; the blocks run through `core.run_blocks`)."""
        current_step = self.get_current_step()
        modulated_inp = self._get_modulated_input(model_type, hidden_states, temb, transformer_blocks, encoder_hidden_states)
        cache_key = self.get_cache_key()
        rel = None
        if current_step == 0:
            should_calc = True
            self.accumulated_rel_l1_distance_dict[cache_key] = 0
        else:
            rescale_func = np.poly1d(self.coefficients[cache_key])
            rel = self.rel_distance(modulated_inp, self.previous_modulated_input_dict[cache_key])
            self.accumulated_rel_l1_distance_dict[cache_key] += rescale_func(rel)
            if self.accumulated_rel_l1_distance_dict[cache_key] < self.config.threshold:
                should_calc = False
            else:
                should_calc = True
                self.accumulated_rel_l1_distance_dict[cache_key] = 0
        self.trace.append((current_step, rel, float(self.accumulated_rel_l1_distance_dict[cache_key]), should_calc, cache_key))
        self.previous_modulated_input_dict[cache_key] = modulated_inp
        if not should_calc:
            return hidden_states + self.previous_residual_dict[cache_key]
        ori_hidden_states = hidden_states.clone()     # the fused blocks update the residual stream in place
        out = core.run_blocks(hidden_states, encoder_hidden_states, temb, image_rotary_emb, attention_kwargs)
        self.previous_residual_dict[cache_key] = out - ori_hidden_states
        return out


@AutoCache.register("fbcache")
class FBCache(AutoCache):
    """First-block cache (xcaching.py:279-403): block 0 always runs; if its output moved little since the last computed
    step (accumulated relative L1 below the threshold, after `warmup_steps`), the remaining blocks are skipped and the
    cached residual `out - in` of the last full evaluation is added instead."""

    def apply_cache(self, model_type: str, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor = None,
                    encoder_hidden_states_mask=None, temb: torch.Tensor = None, image_rotary_emb: torch.Tensor = None,
                    attention_kwargs: Optional[Dict[str, Any]] = None, transformer_blocks=None, single_transformer_blocks=None,
                    core=None, **_):
        if model_type not in ("flux", "qwenimage", "sd35", "wan"):
            raise ValueError(f"FBCache unsupported model type: {model_type}. Supported types: qwenimage, flux, sd35, wan")
        current_step = self.get_current_step()
        first_encoder, first_hidden = _block_forward(model_type, transformer_blocks[0], hidden_states, encoder_hidden_states, temb,
                                                     image_rotary_emb, attention_kwargs, encoder_hidden_states_mask)
        cache_key = self.get_cache_key()
        prev = self.previous_modulated_input_dict[cache_key]
        rel = None
        if current_step <= self.config.warmup_steps or prev is None:
            should_calc = True
            self.accumulated_rel_l1_distance_dict[cache_key] = 0
        else:
            rel = self.rel_distance(first_hidden, prev)
            self.accumulated_rel_l1_distance_dict[cache_key] += rel
            if self.accumulated_rel_l1_distance_dict[cache_key] < self.config.threshold:
                should_calc = False
            else:
                should_calc = True
                self.accumulated_rel_l1_distance_dict[cache_key] = 0
        self.trace.append((current_step, rel, float(self.accumulated_rel_l1_distance_dict[cache_key]), should_calc, cache_key))
        self.previous_modulated_input_dict[cache_key] = first_hidden.clone()   # later blocks update the stream in place
        if not should_calc:
            return hidden_states + self.previous_residual_dict[cache_key]
        out = core.run_blocks(first_hidden, encoder_hidden_states if model_type == "wan" else first_encoder, temb, image_rotary_emb,
                              attention_kwargs, start_block=1, encoder_hidden_states_mask=encoder_hidden_states_mask)
        self.previous_residual_dict[cache_key] = out - hidden_states
        return out


@AutoCache.register("dicache")
class DiCache(AutoCache):
    """Probe-based cache (xcaching.py:418-556): the first `probe_depth` blocks always run; the relative change of their
    output (and optionally of their input) since the previous step accumulates -- in bf16, as the reference's 0-dim
    tensors do -- and while it stays under the threshold the remaining blocks are replaced by a cached residual,
    extrapolated from the last two computed residuals once two are available.  As in the reference the residual
    (`out - in` of the whole stack) is added to the PROBE output on skipped steps."""

    def __init__(self, config):
        super().__init__(config)
        keys = ("positive", "negative")
        self.previous_probe_stats_dict = {k: None for k in keys}
        self.previous_residual_window_dict = {k: [] for k in keys}

    def apply_cache(self, model_type: str, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor = None,
                    encoder_hidden_states_mask=None, temb: torch.Tensor = None, image_rotary_emb: torch.Tensor = None,
                    attention_kwargs: Optional[Dict[str, Any]] = None, transformer_blocks=None, single_transformer_blocks=None,
                    core=None, **_):
        if model_type == "wan":
            # upstream's wan branch calls _forward_wan_blocks without its sparse_attn argument (xcaching.py:468,514) and raises
            # a TypeError: there is no reference behaviour to reproduce
            raise NotImplementedError("DiCache for Wan: the reference's own call is broken (caching/xcaching.py:468); use FBCache")
        if model_type not in ("flux", "qwenimage", "sd35"):
            raise ValueError(f"DiCache unsupported model type: {model_type}. Supported types: qwenimage, flux, sd35")
        current_step = self.get_current_step()
        tcb = self.config.total_steps_callback
        total_steps = tcb() if tcb is not None and tcb() is not None else 25
        cache_key = self.get_cache_key()
        ori = hidden_states
        probe_h, probe_e = hidden_states, encoder_hidden_states
        for blk in transformer_blocks[: self.config.probe_depth]:          # blocks return fresh tensors: `ori` stays intact
            probe_e, probe_h = _block_forward(model_type, blk, probe_h, probe_e, temb, image_rotary_emb, attention_kwargs,
                                              encoder_hidden_states_mask)
        err = None
        if current_step <= int(self.config.ret_ratio * total_steps):
            should_calc = True
            self.accumulated_rel_l1_distance_dict[cache_key] = 0
        else:
            delta_y = self.rel_distance(probe_h, self.previous_probe_stats_dict[cache_key])
            if self.config.rel_l1_distance_algo == "delta_minus":
                delta_x = self.rel_distance(ori, self.previous_modulated_input_dict[cache_key])
                err = _bf16(abs(_bf16(delta_y - delta_x)))
            elif self.config.rel_l1_distance_algo == "delta_y":
                err = delta_y
            else:
                raise ValueError(f"unknown rel_l1_distance_algo: {self.config.rel_l1_distance_algo}")
            acc = _bf16(self.accumulated_rel_l1_distance_dict[cache_key] + err)
            self.accumulated_rel_l1_distance_dict[cache_key] = acc
            if acc < self.config.threshold:
                should_calc = False
            else:
                should_calc = True
                self.accumulated_rel_l1_distance_dict[cache_key] = 0
        self.trace.append((current_step, err, float(self.accumulated_rel_l1_distance_dict[cache_key]), should_calc, cache_key))
        self.previous_probe_stats_dict[cache_key] = probe_h.clone()
        self.previous_modulated_input_dict[cache_key] = ori.clone()
        if not should_calc:
            window = self.previous_residual_window_dict[cache_key]
            if len(window) >= 2:
                r2, r1 = window[-2], window[-1]
                gamma = (r2.abs().mean() / (r1 - r2).abs().mean()).clip(1, 1.5)      # bf16 0-dim tensor ops, as in the reference
                return probe_h + (r2 + gamma * (r1 - r2))
            return probe_h + self.previous_residual_dict[cache_key]
        out = core.run_blocks(probe_h, probe_e, temb, image_rotary_emb, attention_kwargs, start_block=self.config.probe_depth,
                              encoder_hidden_states_mask=encoder_hidden_states_mask)
        self.previous_residual_dict[cache_key] = out - ori
        self.previous_residual_window_dict[cache_key].append(self.previous_residual_dict[cache_key])
        return out
