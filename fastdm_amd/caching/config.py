"""Step-cache configuration: field-for-field the reference's dataclasses (fastdm/caching/config.py:7-73), loaded by the
shared registry helper.  Keys the target dataclass does not declare are dropped, exactly like the reference (its shipped
fbcache JSONs spell `wamup_steps` and therefore run with the default of 6)."""
from dataclasses import dataclass, field
from typing import Callable, List, Optional

from fastdm_amd.utils.registry import RegisteredConfig


@dataclass
class CacheConfig(RegisteredConfig):
    _selector = "cache_algorithm"

    cache_algorithm: str
    enable_caching: bool = False
    threshold: float = 0.2
    current_steps_callback: Optional[Callable] = None     # () -> index of the denoise step being evaluated
    total_steps_callback: Optional[Callable] = None
    negtive_cache: Optional[bool] = False                 # (sic) separate cache entries for the negative-prompt forward


@CacheConfig.register("teacache")
@dataclass
class TeaCacheConfig(CacheConfig):
    coefficients: List[float] = field(default_factory=list)            # polynomial rescale, highest power first
    negtive_coefficients: List[float] = field(default_factory=list)


@CacheConfig.register("fbcache")
@dataclass
class FBCacheConfig(CacheConfig):
    warmup_steps: int = 6


@CacheConfig.register("dicache")
@dataclass
class DiCacheConfig(CacheConfig):
    probe_depth: int = 1                      # blocks that always run and serve as the probe
    ret_ratio: float = 0.2                    # the first ret_ratio * total_steps steps are always computed
    rel_l1_distance_algo: str = "delta_y"     # "delta_y" | "delta_minus"
