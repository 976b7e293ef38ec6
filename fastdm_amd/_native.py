"""ctypes binding of libfastdm_amd.so (the C ABI declared in include/fastdm_amd.h).

The signatures below mirror the header one-to-one; tests/test_abi.py checks that every `fdm_*`
prototype in the header is present here and exported by the library.  There is no fallback: if the
library cannot be found (or built in-tree with hipcc) importing a kernel raises.
"""
import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_lib", "libfastdm_amd.so")
BUILD_SCRIPT = os.path.join(_HERE, "csrc", "build.sh")

P, I64, I32, U32, F32 = C.c_void_p, C.c_int64, C.c_int, C.c_uint32, C.c_float

# name -> (restype, argtypes); order = include/fastdm_amd.h
SIGNATURES = {
    "fdm_version": (I32, []),
    "fdm_error_string": (C.c_char_p, [I32]),
    "fdm_last_hip_error_string": (C.c_char_p, []),
    "fdm_rms_norm": (I32, [P, P, P, F32, I64, I32, P]),
    "fdm_rms_norm_strided": (I32, [P, P, P, F32, I64, I32, I64, I64, P]),
    "fdm_rope_inplace": (I32, [P, P, P, I64, I64, I32, I32, I32, I32, I64, I64, I32, P]),
    "fdm_quant_fp8_rowwise": (I32, [P, P, P, I64, I32, I64, P]),
    "fdm_quant_int8_rowwise": (I32, [P, P, P, P, I64, I32, I64, P]),
    "fdm_gelu_and_mul": (I32, [P, P, I64, I32, P]),
    "fdm_gemm_w8a8_fp8": (I32, [P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I32, P, I64, P, I64, P]),
    "fdm_gemm_w8a8_fp8_x2": (I32, [P, P, P]),
    "fdm_gemm_w8a8_int8_x2": (I32, [P, P, P]),
    "fdm_gemm_w8a8_int8_x2_ws": (I32, [P, P, P, I64, P]),
    "fdm_gemm_int8_workspace_bytes": (I64, []),
    "fdm_gemm_last_kernel": (I32, []),
    "fdm_gemm_w8a8_int8": (I32, [P, P, P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I32, P, I64, P, I64, P]),
    "fdm_gemm_bf16": (I32, [P, P, P, P, I64, I64, I64, I64, I64, I64, I32, P, I64, P, I64, P]),
    "fdm_linear_skinny_bf16": (I32, [P, P, P, P, P, I64, I64, I64, I64, I64, I64, P]),
    "fdm_attn_fwd": (I32, [P, P, P, P, I64, I64, I64, I32, I32, I32, I64, I64, I64, I64, F32, P, I32, I32, P, I64, P]),
    "fdm_attn_fwd_workspace_bytes": (I64, [I32]),
    "fdm_ln_modulate": (I32, [P, P, P, P, I32, P, P, P, I64, P, P, P, I64, I32, I64, F32, I32, P]),
    "fdm_ln_modulate_f32": (I32, [P, P, P, P, I32, P, P, P, I64, I64, I32, I64, F32, P]),
    "fdm_gate_residual": (I32, [P, P, P, P, I64, I64, I32, I64, P]),
    "fdm_gelu": (I32, [P, P, I64, I32, P]),
    "fdm_quant_rowwise_cat2": (I32, [P, P, P, I32, P, I32, I64, P, I32, I64, I64, P]),
    "fdm_qk_norm_rope": (I32, [P, P, P, P, P, P, P, I64, I64, I64, I64, I32, I32, F32, P]),
    "fdm_qk_norm_rope_inplace": (I32, [P, P, P, P, I64, I64, I64, I64, I64, I64, I32, I32, F32, P]),
    "fdm_absdiff_sums": (I32, [P, P, P, I64, P]),
    "fdm_abi_version": (I32, []),
    "fdm_synth_fill_bf16": (I32, [P, I64, U32, I32, F32, F32, P]),
    "fdm_debug_cvt_fp8": (I32, [P, P, I64, I32, P]),
}

ABI_VERSION = 11         # FDM_ABI_VERSION of include/fastdm_amd.h these SIGNATURES were written against
EPI_NONE, EPI_GELU_TANH, EPI_GELU_ERF, EPI_GATE_RESIDUAL, EPI_GATE_RESIDUAL_F32 = 0, 1, 2, 3, 4


class GemmFp8Desc(C.Structure):
    """struct fdm_gemm_fp8_desc (include/fastdm_amd.h)."""
    _fields_ = [("out", P), ("a", P), ("b", P), ("scale_a", P), ("scale_b", P), ("bias", P),
                ("M", I64), ("N", I64), ("K", I64), ("lda", I64), ("ldb", I64), ("ldc", I64),
                ("epilogue", I64), ("epi_gate", P), ("epi_gate_stride", I64), ("epi_res", P), ("epi_rows_per_batch", I64),
                ("out_plane_cols", I64), ("out_plane_stride", I64)]



class GemmInt8Desc(C.Structure):
    """struct fdm_gemm_int8_desc (include/fastdm_amd.h)."""
    _fields_ = [("out", P), ("a", P), ("b", P), ("scale_a", P), ("scale_b", P), ("azp_adj", P), ("azp", P), ("bias", P),
                ("M", I64), ("N", I64), ("K", I64), ("lda", I64), ("ldb", I64), ("ldc", I64),
                ("epilogue", I64), ("epi_gate", P), ("epi_gate_stride", I64), ("epi_res", P), ("epi_rows_per_batch", I64),
                ("out_plane_cols", I64), ("out_plane_stride", I64)]


_lib = None
_lock = threading.Lock()


class NativeLibraryError(ImportError):
    pass


def _source_digest(srcs) -> str:
    import hashlib
    h = hashlib.sha256()
    for path in sorted(srcs):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def current_source_digest() -> str:
    """Content digest of everything libfastdm_amd.so is built from (kernel sources, build script, the C-ABI header)."""
    srcs = [os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc"))
            if f.endswith((".hip", ".h", ".sh", ".inc", ".py"))]      # .inc: the generated assembly bodies of gemm_p4_kernel, .py: their generator
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "fastdm_amd.h"))
    return _source_digest(srcs)


def build(force: bool = False) -> str:
    """Compile the HIP sources in-tree for gfx950 (hipcc cross-compiles without a GPU).  The library is rebuilt only
    when the sources' content digest changed (file times do not survive a copy to another machine)."""
    digest = current_source_digest()
    stamp = LIB_PATH + ".srcdigest"

    def current():
        return os.path.exists(LIB_PATH) and os.path.exists(stamp) and open(stamp).read().strip() == digest

    if not force and current():
        return LIB_PATH
    # one builder at a time: the ranks of a multi-GPU launch all come through here
    import fcntl
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    with open(LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or not current():          # another process may have finished the build while this one waited
            res = subprocess.run(["bash", BUILD_SCRIPT], capture_output=True, text=True)
            if res.returncode != 0 or not os.path.exists(LIB_PATH):
                raise NativeLibraryError(f"building libfastdm_amd.so failed:\n{res.stdout}\n{res.stderr}")
            tmp = f"{stamp}.{os.getpid()}"
            with open(tmp, "w") as f:
                f.write(digest)
            os.replace(tmp, stamp)           # atomic, like the library itself (build.sh links to a temporary name and renames)
    return LIB_PATH


def lib():
    """Load (once) and return the ctypes handle; raises NativeLibraryError if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        # Build artefacts are git-ignored, so a checkout can hold a library older than the sources: whenever a compiler is
        # at hand, build() compares the sources' digest with the one the library was built from and rebuilds on mismatch.
        # (On the GPU box the prebuilt library travels with its digest file and matches.)
        have_hipcc = os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"))
        if not os.path.exists(LIB_PATH) or (have_hipcc and os.environ.get("FDM_NO_REBUILD") != "1"):
            try:
                build()
            except Exception as e:  # noqa: BLE001
                raise NativeLibraryError(
                    f"fastdm_amd: HIP library {LIB_PATH} is missing and could not be built ({e}). "
                    "This package has no CPU / eager fallback.") from e
        # torch bundles its own libamdhip64; it must be the HIP runtime of the process.  Loading our library first
        # would pull in /opt/rocm's copy, and kernels launched through that second runtime fail with
        # "no ROCm-capable device is detected" once torch has claimed the GPU.
        import torch  # noqa: F401
        try:
            handle = C.CDLL(LIB_PATH)
        except OSError as e:
            raise NativeLibraryError(f"fastdm_amd: cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError -> missing export: let it propagate loudly
            fn.restype, fn.argtypes = res, args
        if handle.fdm_abi_version() != ABI_VERSION:
            raise NativeLibraryError(f"fastdm_amd: {LIB_PATH} was built from another revision of include/fastdm_amd.h (ABI "
                                     f"{handle.fdm_abi_version()}, these bindings expect {ABI_VERSION}); rebuild it with __graft_entry__.build()")
        _lib = handle
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().fdm_error_string(rc).decode()
        if rc == -3:
            msg += ": " + lib().fdm_last_hip_error_string().decode()
        raise RuntimeError(f"{what} failed: {msg} (code {rc})")


def ptr(t):
    return None if t is None else t.data_ptr()


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


_attn_ws = {}


def attn_workspace(device, head_dim):
    """Per-device scratch for fdm_attn_fwd's split-KV tail balancing (allocated once; fdm_attn_fwd_workspace_bytes: 136 MB at D=128, 69 MB at D=64)."""
    import torch
    key = (device.index if device.index is not None else torch.cuda.current_device(), head_dim)
    ws = _attn_ws.get(key)
    if ws is None:
        nbytes = int(lib().fdm_attn_fwd_workspace_bytes(head_dim))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _attn_ws[key] = ws
    return ws


import collections

_gemm_ws = collections.OrderedDict()      # (device, stream handle) -> buffer, least recently used first
_GEMM_WS_MAX = 8                           # 8 x 64 MiB: more streams than that in flight at once is not a denoise loop


def gemm_int8_workspace(device):
    """Scratch of fdm_gemm_w8a8_int8_x2_ws's split-K tail (64 MiB), one buffer per (device, STREAM): the library's merge launch has
    consumed it before the next GEMM's parts write only when both are on one stream, so launches on another stream get their own
    buffer instead of silently sharing partial sums.  Allocate it before any HIP-graph capture (`warm_gemm_int8_workspace`): an
    allocation inside a capture would land in the graph's private pool."""
    import torch
    dev = device.index if device.index is not None else torch.cuda.current_device()
    key = (dev, int(torch.cuda.current_stream(dev).cuda_stream))
    ws = _gemm_ws.get(key)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("fastdm_amd: the int8 split-K workspace of this stream must be allocated before graph capture "
                               "(call fastdm_amd._native.warm_gemm_int8_workspace(device) on the capture stream first)")
        ws = torch.empty(int(lib().fdm_gemm_int8_workspace_bytes()), dtype=torch.uint8, device=device)
        _gemm_ws[key] = ws
        # bounded: a stream handle is not an identity (a destroyed stream's handle can come back), and every stream that ever ran an
        # int8 GEMM used to keep its 64 MiB for good.  The least recently used entry goes; torch's caching allocator keeps the block
        # alive until the work queued on it has run.
        while len(_gemm_ws) > _GEMM_WS_MAX:
            _gemm_ws.popitem(last=False)
    else:
        _gemm_ws.move_to_end(key)
    return ws


def warm_gemm_int8_workspace(device):
    """Allocate the int8 split-K workspace of the current stream eagerly (model load / warm-up path)."""
    return gemm_int8_workspace(device)
