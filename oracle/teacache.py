"""TeaCache oracle (restatement of fastdm/caching/{config,xcaching}.py for FLUX).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Citations relative to /root/reference.
"""
from dataclasses import dataclass, field
from typing import Callable, List, Optional

import numpy as np

# examples/xcaching/configs/teacache_flux.json
FLUX_COEFFICIENTS = [4.98651651e+02, -2.83781631e+02, 5.58554382e+01, -3.82021401e+00, 2.64230861e-01]


@dataclass
class TeaCacheConfig:
    """fastdm/caching/config.py:7-59 (fields that matter on the FLUX path)."""
    cache_algorithm: str = "teacache"
    enable_caching: bool = True
    threshold: float = 0.2
    coefficients: List[float] = field(default_factory=lambda: list(FLUX_COEFFICIENTS))
    current_steps_callback: Optional[Callable] = None
    negtive_cache: bool = False                                   # sic (fastdm/caching/config.py:16)
    negtive_coefficients: Optional[List[float]] = None


def poly(coeffs, x: float) -> float:
    """np.poly1d(coeffs)(x): highest power first (fastdm/caching/xcaching.py:212-216)."""
    return float(np.poly1d(coeffs)(x))


class TeaCacheDecision:
    """The host-side decision rule only (xcaching.py:205-225): feed the per-step relative
    L1 distance, get should_calc.  Used to pin the decision trace independently of tensors."""

    def __init__(self, coeffs, threshold):
        self.coeffs, self.threshold, self.acc = coeffs, threshold, 0.0

    def step(self, step_index: int, rel_distance: Optional[float]) -> bool:
        if step_index == 0:
            self.acc = 0
            return True
        self.acc += poly(self.coeffs, rel_distance)
        if self.acc < self.threshold:
            return False
        self.acc = 0
        return True


class TeaCache:
    """fastdm/caching/xcaching.py:156-276, FLUX branch, single ('positive') cache key."""

    def __init__(self, config: TeaCacheConfig):
        self.config = config
        self.decision = TeaCacheDecision(config.coefficients, config.threshold)
        self.prev_modulated = None
        self.prev_residual = None
        self.trace = []  # (step, rel_distance, acc_after, should_calc)

    def apply_cache(self, core, hidden, encoder, temb, rope):
        step = self.config.current_steps_callback()
        step = 0 if step is None else step
        # xcaching.py:165-183: modulated input = first block's norm1 on clones
        modulated = core.transformer_blocks[0].norm1(hidden.clone(), temb.clone())[0]
        rel = None
        if step != 0:
            rel = ((modulated - self.prev_modulated).abs().mean() / self.prev_modulated.abs().mean()).cpu().item()
        should_calc = self.decision.step(step, rel)
        self.trace.append((step, rel, float(self.decision.acc), should_calc))
        self.prev_modulated = modulated
        if not should_calc:
            return hidden + self.prev_residual
        out = core.run_blocks(hidden, encoder, temb, rope)
        self.prev_residual = out - hidden
        return out


class TeaCacheSD3(TeaCache):
    """The sd35 branch (xcaching.py:181-183,262-270): same probe as FLUX (block 0's norm1 output), blocks return
    (encoder, hidden) and the last one has no encoder output."""

    def apply_cache_sd35(self, core, hidden, encoder, temb):
        step = self.config.current_steps_callback()
        step = 0 if step is None else step
        modulated = core.transformer_blocks[0].norm1_modulated(hidden.clone(), temb.clone())
        rel = None
        if step != 0:
            rel = ((modulated - self.prev_modulated).abs().mean() / self.prev_modulated.abs().mean()).cpu().item()
        should_calc = self.decision.step(step, rel)
        self.trace.append((step, rel, float(self.decision.acc), should_calc))
        self.prev_modulated = modulated
        if not should_calc:
            return hidden + self.prev_residual
        x, enc = hidden, encoder
        for blk in core.transformer_blocks:
            enc, x = blk(x, enc, temb)
        self.prev_residual = x - hidden
        return x


class TeaCacheQwen:
    """fastdm/caching/xcaching.py:156-276, Qwen-Image branch: the modulated input is the TEXT stream of block 0
    (xcaching.py:171-180) and, with `negtive_cache`, cond / uncond forwards alternate between two cache keys with
    their own coefficients (xcaching.py:66-76,160-163)."""

    def __init__(self, config: TeaCacheConfig):
        self.config = config
        keys = ("positive", "negative")
        coeffs = {"positive": config.coefficients, "negative": config.negtive_coefficients}
        self.decision = {k: TeaCacheDecision(coeffs[k], config.threshold) for k in keys}
        self.prev_modulated = {k: None for k in keys}
        self.prev_residual = {k: None for k in keys}
        self.cache_status = {"positive": True, "negative": False}
        self.trace = []

    def get_cache_key(self):
        if not self.config.negtive_cache:
            return "positive"
        key = None
        for k in self.cache_status:
            if self.cache_status[k] and key is None:
                key = k
            self.cache_status[k] = not self.cache_status[k]
        return key

    def apply_cache_qwen(self, core, hidden, encoder, temb, rope):
        step = self.config.current_steps_callback()
        step = 0 if step is None else step
        modulated = core.transformer_blocks[0].modulated_inputs(hidden.clone(), encoder.clone(), temb.clone())[3]
        key = self.get_cache_key()
        rel = None
        if step != 0:
            prev = self.prev_modulated[key]
            rel = ((modulated - prev).abs().mean() / prev.abs().mean()).cpu().item()
        should_calc = self.decision[key].step(step, rel)
        self.trace.append((step, key, rel, float(self.decision[key].acc), should_calc))
        self.prev_modulated[key] = modulated
        if not should_calc:
            return hidden + self.prev_residual[key]
        out = core.run_blocks(hidden, encoder, temb, rope)
        self.prev_residual[key] = out - hidden
        return out


class FBCacheOracle:
    """fastdm/caching/xcaching.py:279-403 (FLUX and Qwen-Image branches, single cache key): block 0 always runs; after
    `warmup_steps` the remaining blocks are skipped while the accumulated relative L1 change of block 0's output stays
    below the threshold (no polynomial rescale)."""

    def __init__(self, threshold: float, warmup_steps: int, current_steps_callback):
        self.threshold, self.warmup_steps, self.cb = threshold, warmup_steps, current_steps_callback
        self.acc = 0.0
        self.prev = None
        self.prev_residual = None
        self.trace = []
        self.config = type("Cfg", (), {"enable_caching": True})()

    def _decide(self, first_hidden):
        step = self.cb()
        step = 0 if step is None else step
        rel = None
        if step <= self.warmup_steps or self.prev is None:
            calc, self.acc = True, 0
        else:
            rel = ((first_hidden - self.prev).abs().mean() / self.prev.abs().mean()).cpu().item()
            self.acc += rel
            if self.acc < self.threshold:
                calc = False
            else:
                calc, self.acc = True, 0
        self.trace.append((step, rel, float(self.acc), calc))
        self.prev = first_hidden.clone()
        return calc

    def apply_cache(self, core, hidden, encoder, temb, rope):          # FLUX core
        enc1, hid1 = core.transformer_blocks[0](hidden, encoder, temb, rope)
        if not self._decide(hid1):
            return hidden + self.prev_residual
        for blk in core.transformer_blocks[1:]:
            enc1, hid1 = blk(hid1, enc1, temb, rope)
        n_txt = enc1.shape[1]
        import torch
        x = torch.cat([enc1, hid1], dim=1)
        for blk in core.single_transformer_blocks:
            x = blk(x, temb, rope)
        out = x[:, n_txt:]
        self.prev_residual = out - hidden
        return out

    def apply_cache_wan(self, core, hidden, encoder, temb6, rope):     # Wan core: blocks return the hidden states only
        hid1 = core.blocks[0](hidden, encoder, temb6, rope)             # xcaching.py:288-290: no sparse mask on the probe block
        if not self._decide(hid1):
            return hidden + self.prev_residual
        frames = getattr(core, "_cache_frames", None)
        for li, blk in enumerate(core.blocks[1:], start=1):
            hid1 = blk(hid1, encoder, temb6, rope, core.block_mask(li, hid1.shape[1], frames, hid1.shape[0]) if frames else None)
        self.prev_residual = hid1 - hidden
        return hid1

    def apply_cache_qwen(self, core, hidden, encoder, temb, rope):     # Qwen-Image and SD3.5 cores
        enc1, hid1 = core.transformer_blocks[0](hidden, encoder, temb, rope)
        if not self._decide(hid1):
            return hidden + self.prev_residual
        for blk in core.transformer_blocks[1:]:
            enc1, hid1 = blk(hid1, enc1, temb, rope)
        self.prev_residual = hid1 - hidden
        return hid1


class DiCacheOracle:
    """fastdm/caching/xcaching.py:418-556 (FLUX and Qwen-Image branches, single cache key), tensor arithmetic exactly as the
    reference writes it (the accumulated distance becomes a 0-dim bf16 tensor after the first add)."""

    def __init__(self, threshold, probe_depth, ret_ratio, algo, current_steps_callback, total_steps):
        self.threshold, self.probe_depth, self.ret_ratio, self.algo = threshold, probe_depth, ret_ratio, algo
        self.cb, self.total_steps = current_steps_callback, total_steps
        self.acc = 0
        self.prev_in = self.prev_probe = self.prev_residual = None
        self.window = []
        self.trace = []
        self.config = type("Cfg", (), {"enable_caching": True})()

    def _run(self, core, hidden, encoder, temb, rope, flux: bool):
        step = self.cb()
        step = 0 if step is None else step
        ori = hidden.clone()
        enc = encoder
        for blk in core.transformer_blocks[: self.probe_depth]:
            enc, hidden = blk(hidden, enc, temb, rope)
        if step <= int(self.ret_ratio * self.total_steps):
            calc, self.acc = True, 0
        else:
            delta_x = (hidden - self.prev_in).abs().mean() / self.prev_in.abs().mean()
            delta_y = (hidden - self.prev_probe).abs().mean() / self.prev_probe.abs().mean()
            error = (delta_y - delta_x).abs() if self.algo == "delta_minus" else delta_y
            self.acc = self.acc + error
            if self.acc < self.threshold:
                calc = False
            else:
                calc, self.acc = True, 0
        self.trace.append((step, float(self.acc), calc))
        self.prev_probe = hidden.clone()
        self.prev_in = ori
        if not calc:
            if len(self.window) >= 2:
                import torch
                ind = torch.zeros_like(hidden)
                gamma = ((ind - self.window[-2]).abs().mean() / (self.window[-1] - self.window[-2]).abs().mean()).clip(1, 1.5)
                return hidden + (self.window[-2] + gamma * (self.window[-1] - self.window[-2]))
            return hidden + self.prev_residual
        for blk in core.transformer_blocks[self.probe_depth:]:
            enc, hidden = blk(hidden, enc, temb, rope)
        if flux:
            import torch
            n_txt = enc.shape[1]
            x = torch.cat([enc, hidden], dim=1)
            for blk in core.single_transformer_blocks:
                x = blk(x, temb, rope)
            hidden = x[:, n_txt:]
        self.prev_residual = hidden - ori
        self.window.append(self.prev_residual)
        return hidden

    def apply_cache(self, core, hidden, encoder, temb, rope):
        return self._run(core, hidden, encoder, temb, rope, flux=True)

    def apply_cache_qwen(self, core, hidden, encoder, temb, rope):
        return self._run(core, hidden, encoder, temb, rope, flux=False)
