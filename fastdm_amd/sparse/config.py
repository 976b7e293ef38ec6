"""Sparse-attention configuration: the reference's fields (fastdm/sparse/config.py:7-62) on the shared registry helper."""
from dataclasses import dataclass
from typing import Callable, Optional

from fastdm_amd.utils.registry import RegisteredConfig


@dataclass
class SparseConfig(RegisteredConfig):
    _selector = "sparse_algorithm"

    sparse_algorithm: str
    enable_sparse: bool = False
    block_size: int = 128                                  # tokens per mask block (64 or 128)


@SparseConfig.register("radial")
@dataclass
class RadialAttnConfig(SparseConfig):
    backend: str = "sparse_sageattn"                       # accepted for JSON compatibility; there is one backend here
    decay_factor: float = 0.5
    dense_layers: int = 1                                  # layers [0, dense_layers) always attend densely
    dense_steps: int = 5                                   # so do denoise steps [0, dense_steps)
    model_type: str = "wan"
    video_token_num: int = 25440
    num_frame: int = 16
    current_steps_callback: Optional[Callable] = None
