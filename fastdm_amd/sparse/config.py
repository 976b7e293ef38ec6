"""Sparse-attention configuration (mirror of fastdm/sparse/config.py:1-66): same JSON keys, same registry."""
import json
import os
from dataclasses import dataclass, fields
from typing import Any, Callable, Dict, Optional, Type


@dataclass
class SparseConfig:
    sparse_algorithm: str
    enable_sparse: bool = False
    block_size: int = 128

    _registry = {}

    @classmethod
    def register(cls, name: str):
        def decorator(sub_cls):
            SparseConfig._registry[name.lower()] = sub_cls
            return sub_cls
        return decorator

    @classmethod
    def from_dict(cls, data: Dict[str, Any]) -> "SparseConfig":
        algo = data.get("sparse_algorithm")
        if algo is None:
            raise ValueError("sparse_algorithm must be specified in the config data.")
        target: Type[SparseConfig] = SparseConfig._registry.get(algo.lower(), cls)
        names = {f.name for f in fields(target)}
        return target(**{k: v for k, v in data.items() if k in names})

    @classmethod
    def from_json(cls, file_path: str) -> "SparseConfig":
        if not os.path.exists(file_path):
            raise FileNotFoundError(f"file is not exist: {file_path}")
        with open(file_path, "r", encoding="utf-8") as f:
            return cls.from_dict(json.load(f))


@SparseConfig.register("radial")
@dataclass
class RadialAttnConfig(SparseConfig):
    """fastdm/sparse/config.py:52-62."""
    backend: str = "sparse_sageattn"
    decay_factor: float = 0.5
    dense_layers: int = 1
    dense_steps: int = 5
    model_type: str = "wan"
    video_token_num: int = 25440
    num_frame: int = 16
    current_steps_callback: Optional[Callable] = None
