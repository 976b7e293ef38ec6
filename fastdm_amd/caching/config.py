"""Cache configuration (mirror of fastdm/caching/config.py:7-73): JSON -> dataclass registry; keys the
target dataclass does not declare are silently dropped, exactly like the reference (its shipped
JSONs rely on that)."""
import json
import os
from dataclasses import dataclass, field, fields
from typing import Any, Callable, Dict, List, Optional, Type


@dataclass
class CacheConfig:
    cache_algorithm: str
    enable_caching: bool = False
    threshold: float = 0.2
    current_steps_callback: Optional[Callable] = None
    total_steps_callback: Optional[Callable] = None
    negtive_cache: Optional[bool] = False   # (sic) spelling of the reference's JSON key

    _registry = None

    @classmethod
    def register(cls, name: str):
        def decorator(sub_cls):
            if CacheConfig._registry is None:
                CacheConfig._registry = {}
            CacheConfig._registry[name.lower()] = sub_cls
            return sub_cls
        return decorator

    @classmethod
    def from_dict(cls, data: Dict[str, Any]) -> "CacheConfig":
        algo = data.get("cache_algorithm")
        if algo is None:
            raise ValueError("cache_algorithm must be specified in the config data.")
        target: Type[CacheConfig] = (CacheConfig._registry or {}).get(algo.lower(), cls)
        names = {f.name for f in fields(target)}
        return target(**{k: v for k, v in data.items() if k in names})

    @classmethod
    def from_json(cls, file_path: str) -> "CacheConfig":
        if not os.path.exists(file_path):
            raise FileNotFoundError(f"file is not exist: {file_path}")
        with open(file_path, "r", encoding="utf-8") as f:
            return cls.from_dict(json.load(f))


@CacheConfig.register("teacache")
@dataclass
class TeaCacheConfig(CacheConfig):
    coefficients: List[float] = field(default_factory=list)
    negtive_coefficients: List[float] = field(default_factory=list)


@CacheConfig.register("fbcache")
@dataclass
class FBCacheConfig(CacheConfig):
    """fastdm/caching/config.py:69-72.  (The shipped JSONs spell the key 'wamup_steps', which the loader drops: they run
    with the default of 6.)"""
    warmup_steps: int = 6
