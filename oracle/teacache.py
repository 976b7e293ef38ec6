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
