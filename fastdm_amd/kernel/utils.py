"""Host-side helpers with the reference's names (fastdm/kernel/utils.py:11-45)."""
import os
import time
from typing import Callable

import torch

_BACKENDS = ("hip", "cuda")  # "cuda" is accepted so unchanged gen.py invocations work; both mean the HIP library


def set_global_backend(backend: str):
    """fastdm/kernel/utils.py:11-15.  There is exactly one backend here (hand-written gfx950 HIP)."""
    if backend not in _BACKENDS:
        raise ValueError(f"Unsupported backend: {backend}. Available backends: hip (alias: cuda).")
    os.environ["KERNEL_BACKEND"] = "hip"


def get_available_backends(op_name: str) -> list:
    return ["hip"]


def benchmark_kernel(op_name: str, impl_func: Callable, iterations: int = 50, *args, **kwargs):
    """fastdm/kernel/utils.py:17-34: 3 warm-ups, `iterations` timed calls between device syncs."""
    for _ in range(3):
        impl_func(*args, **kwargs)
    torch.cuda.synchronize()
    start = time.perf_counter()
    for _ in range(iterations):
        result = impl_func(*args, **kwargs)
    torch.cuda.synchronize()
    return (time.perf_counter() - start) / iterations, result


def kernel_output_assert_close(actual: torch.Tensor, expected: torch.Tensor, rtol: float = None, atol: float = None):
    """fastdm/kernel/utils.py:36-45."""
    torch.testing.assert_close(actual, expected.to(actual.dtype), rtol=rtol, atol=atol)
