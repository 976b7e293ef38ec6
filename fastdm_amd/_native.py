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
_PRODUCT_LIB_DIR = os.path.join(_HERE, "_lib")
# FDM_LIB_DIR: run the whole Python stack on an A/B variant built by `FDM_OUT=tools/micro/_ab/<name> csrc/build.sh -D...`
# (same sources, extra flags; bench.py refuses to print a headline line from one unless told to)
LIB_DIR = os.path.abspath(os.environ.get("FDM_LIB_DIR") or _PRODUCT_LIB_DIR)
LIB_PATH = os.path.join(LIB_DIR, "libfastdm_amd.so")
BUILD_SCRIPT = os.path.join(_HERE, "csrc", "build.sh")

P, I64, I32, U32, F32 = C.c_void_p, C.c_int64, C.c_int, C.c_uint32, C.c_float

# name -> (restype, argtypes); order = include/fastdm_amd.h
SIGNATURES = {
    "fdm_version": (I32, []),
    "fdm_error_string": (C.c_char_p, [I32]),
    "fdm_last_hip_error_string": (C.c_char_p, []),
    "fdm_rms_norm": (I32, [P, P, P, F32, I64, I32, P]),
    "fdm_rms_norm_strided": (I32, [P, P, P, F32, I64, I32, I64, I64, P]),
    "fdm_rms_norm_rope_strided": (I32, [P, P, P, F32, I64, I32, I64, I64, P, I64, I32, P]),
    "fdm_rope_inplace": (I32, [P, P, P, I64, I64, I32, I32, I32, I32, I64, I64, I32, P]),
    "fdm_quant_fp8_rowwise": (I32, [P, P, P, I64, I32, I64, P]),
    "fdm_quant_int8_rowwise": (I32, [P, P, P, P, I64, I32, I64, P]),
    "fdm_gelu_and_mul": (I32, [P, P, I64, I32, P]),
    "fdm_gemm_w8a8_fp8": (I32, [P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I32, P, I64, P, I64, P]),
    "fdm_gemm_w8a8_fp8_x2": (I32, [P, P, P]),
    "fdm_gemm_bf16_x2": (I32, [P, P, P]),
    "fdm_gemm_w8a8_int8_x2": (I32, [P, P, P]),
    "fdm_gemm_w8a8_int8_x2_ws": (I32, [P, P, P, I64, P]),
    "fdm_gemm_int8_workspace_bytes": (I64, []),
    "fdm_gemm_last_kernel": (I32, []),
    "fdm_set_gemm_dispatch": (I32, [I32, I32, I32]),
    "fdm_set_gemm_tile_rows": (I32, [I32]),
    "fdm_gemm_w8a8_int8": (I32, [P, P, P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I32, P, I64, P, I64, P]),
    "fdm_gemm_bf16": (I32, [P, P, P, P, I64, I64, I64, I64, I64, I64, I32, P, I64, P, I64, P]),
    "fdm_conv2d_3x3_nhwc_bf16": (I32, [P, P, P, P, P, P, I64, I64, I64, I64, I64, I32, I32, P, I64, P]),
    "fdm_linear_skinny_bf16": (I32, [P, P, P, P, P, I64, I64, I64, I64, I64, I64, P]),
    "fdm_attn_fwd": (I32, [P, P, P, P, I64, I64, I64, I32, I32, I32, I64, I64, I64, I64, F32, P, I32, I32, P, I64, P]),
    "fdm_attn_fwd_causal": (I32, [P, P, P, P, I64, I64, I64, I32, I32, I32, I64, I64, I64, I64, F32, P]),
    "fdm_attn_fwd_workspace_bytes": (I64, [I32]),
    "fdm_attn_pvfp8_workspace_bytes": (I64, [I64, I64, I32, I32]),
    "fdm_attn_fwd_pvfp8": (I32, [P, P, P, P, I64, I64, I64, I32, I32, I32, I64, I64, I64, I64, F32, P, I64, P, I64, P]),
    "fdm_attn_fp8_workspace_bytes": (I64, [I64, I64, I32, I32]),
    "fdm_attn_fwd_fp8": (I32, [P, P, P, P, I64, I64, I64, I32, I32, I32, I64, I64, I64, I64, F32, P, I64, P, I64, P]),
    "fdm_attn_fwd_quant_masked": (I32, [P, P, P, P, I64, I64, I64, I32, I32, I32, I64, I64, I64, I64, F32, P, I32, I32, I32, P, I64, P]),
    "fdm_ln_modulate": (I32, [P, P, P, P, I32, P, P, P, I64, P, P, P, I64, I32, I64, F32, I32, P]),
    "fdm_ln_modulate_f32": (I32, [P, P, P, P, I32, P, P, P, I64, I64, I32, I64, F32, P]),
    "fdm_gate_residual": (I32, [P, P, P, P, I64, I64, I32, I64, P]),
    "fdm_group_norm_workspace_bytes": (I64, [I64, I64, I64, I32]),
    "fdm_group_norm": (I32, [P, P, P, P, P, I64, I64, I64, I32, F32, I32, P, I64, P]),
    "fdm_group_norm_nhwc_workspace_bytes": (I64, [I64, I64, I64, I32]),
    "fdm_group_norm_nhwc": (I32, [P, P, P, P, P, I64, I64, I64, I32, F32, I32, P, I64, P]),
    "fdm_gelu_and_mul_quant": (I32, [P, P, P, I32, P, I64, I32, P]),
    "fdm_gelu": (I32, [P, P, I64, I32, P]),
    "fdm_quant_rowwise_cat2": (I32, [P, P, P, I32, P, I32, I64, P, I32, I64, I64, P]),
    "fdm_quant_rowwise_cat2_act": (I32, [P, P, P, I32, P, I32, I64, I32, P, I32, I64, I32, I64, P]),
    "fdm_qk_norm_rope": (I32, [P, P, P, P, P, P, P, I64, I64, I64, I64, I32, I32, F32, P]),
    "fdm_qk_norm_rope_inplace": (I32, [P, P, P, P, I64, I64, I64, I64, I64, I64, I32, I32, F32, P]),
    "fdm_absdiff_sums": (I32, [P, P, P, I64, P]),
    "fdm_abi_version": (I32, []),
    "fdm_source_digest": (C.c_char_p, []),
    "fdm_build_flags": (C.c_char_p, []),
    "fdm_clock_probe": (I32, [P, P, I32, I32, P]),
    "fdm_synth_fill_bf16": (I32, [P, I64, U32, I32, F32, F32, P]),
    "fdm_debug_cvt_fp8": (I32, [P, P, I64, I32, P]),
}

ABI_VERSION = 20         # FDM_ABI_VERSION of include/fastdm_amd.h these SIGNATURES were written against
EPI_NONE, EPI_GELU_TANH, EPI_GELU_ERF, EPI_GATE_RESIDUAL, EPI_GATE_RESIDUAL_F32 = 0, 1, 2, 3, 4


class GemmFp8Desc(C.Structure):
    """struct fdm_gemm_fp8_desc (include/fastdm_amd.h)."""
    _fields_ = [("out", P), ("a", P), ("b", P), ("scale_a", P), ("scale_b", P), ("bias", P),
                ("M", I64), ("N", I64), ("K", I64), ("lda", I64), ("ldb", I64), ("ldc", I64),
                ("epilogue", I64), ("epi_gate", P), ("epi_gate_stride", I64), ("epi_res", P), ("epi_rows_per_batch", I64),
                ("out_plane_cols", I64), ("out_plane_stride", I64)]



class GemmBf16Desc(C.Structure):
    """struct fdm_gemm_bf16_desc (include/fastdm_amd.h)."""
    _fields_ = [("out", P), ("a", P), ("b", P), ("bias", P),
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


def current_source_digest() -> str:
    """Content digest of everything libfastdm_amd.so is built from (kernel sources, the generated assembly bodies and their
    generator, the build script, the C-ABI header): sha256 of the manifest "<sha256 of file>  <basename>\\n", basenames in C
    order -- csrc/build.sh computes the same value and compiles it INTO the library (fdm_source_digest())."""
    import hashlib
    csrc = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h", ".sh", ".inc", ".py"))]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "fastdm_amd.h"))
    manifest = ""
    for path in sorted(srcs, key=lambda p: os.path.basename(p).encode()):
        with open(path, "rb") as f:
            manifest += f"{hashlib.sha256(f.read()).hexdigest()}  {os.path.basename(path)}\n"
    return hashlib.sha256(manifest.encode()).hexdigest()


def embedded_provenance(path=None):
    """(source digest, extra build flags) compiled into a library file, read from its what(1)-style markers WITHOUT loading it
    (a stale library has to be replaced before dlopen); (None, None) for a file without them."""
    import re
    try:
        with open(path or LIB_PATH, "rb") as f:
            blob = f.read()
    except OSError:
        return None, None
    d = re.search(rb"@\(#\)fdm_source_digest=([0-9a-f]{64})\0", blob)
    fl = re.search(rb"@\(#\)fdm_build_flags=([^\0]*)\0", blob)
    return (d.group(1).decode() if d else None), (fl.group(1).decode() if fl else None)


def is_current(path=None) -> bool:
    return embedded_provenance(path)[0] == current_source_digest()


def _run_build():
    env = dict(os.environ)
    if LIB_DIR != _PRODUCT_LIB_DIR:
        env["FDM_OUT"] = LIB_DIR
    flags = os.environ.get("FDM_LIB_FLAGS", "").split() if LIB_DIR != _PRODUCT_LIB_DIR else []
    return subprocess.run(["bash", BUILD_SCRIPT] + flags, capture_output=True, text=True, env=env)


def build(force: bool = False) -> str:
    """Compile the HIP sources in-tree for gfx950 (hipcc cross-compiles without a GPU).  The library is rebuilt only when the
    digest compiled into it differs from the sources' (file times do not survive a copy to another machine, and a stamp file
    beside the binary can lie about it: round 4 shipped a rejected experiment under HEAD's stamp)."""
    if not force and is_current():
        return LIB_PATH
    # one builder at a time: the ranks of a multi-GPU launch all come through here
    import fcntl
    os.makedirs(LIB_DIR, exist_ok=True)
    with open(LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or not is_current():          # another process may have finished the build while this one waited
            res = _run_build()
            if res.returncode != 0 or not os.path.exists(LIB_PATH):
                raise NativeLibraryError(f"building libfastdm_amd.so failed:\n{res.stdout}\n{res.stderr}")
            if not is_current():
                raise NativeLibraryError(f"{BUILD_SCRIPT} produced a library whose embedded digest {embedded_provenance()[0]} is not the "
                                         f"sources' {current_source_digest()} (sources edited during the build?)")
    return LIB_PATH


_lib_error = None


def lib():
    """Load (once) and return the ctypes handle; raises NativeLibraryError if unavailable.  A failed attempt is remembered: the next call
    raises the same error at once instead of compiling for minutes again (a test suite on a box with a stale library would otherwise
    rebuild once per test)."""
    global _lib, _lib_error
    if _lib is not None:
        return _lib
    if _lib_error is not None:
        raise _lib_error
    try:
        return _load()
    except NativeLibraryError as e:
        _lib_error = e
        raise


def _load():
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        # Build artefacts are git-ignored, so a checkout can hold a library older than the sources.  The digest compiled into
        # the binary is compared with the sources' BEFORE dlopen: on mismatch the library is rebuilt when a compiler is at hand
        # and refused otherwise -- never loaded.  (On the GPU box the prebuilt library travels with the sources and matches.)
        have_hipcc = os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")) and os.environ.get("FDM_NO_REBUILD") != "1"
        if not is_current():
            if not have_hipcc:
                raise NativeLibraryError(
                    f"fastdm_amd: {LIB_PATH} is missing or was built from other sources (embedded digest {embedded_provenance()[0]}, "
                    f"sources {current_source_digest()}) and there is no hipcc to rebuild it (or FDM_NO_REBUILD=1). "
                    "This package has no CPU / eager fallback.")
            try:
                build()
            except Exception as e:  # noqa: BLE001
                raise NativeLibraryError(
                    f"fastdm_amd: HIP library {LIB_PATH} is missing or stale and could not be built ({e}). "
                    "This package has no CPU / eager fallback.") from e
        # torch bundles its own libamdhip64; it must be the HIP runtime of the process.  Loading our library first
        # would pull in /opt/rocm's copy, and kernels launched through that second runtime fail with
        # "no ROCm-capable device is detected" once torch has claimed the GPU.
        import torch  # noqa: F401
        try:
            handle = C.CDLL(LIB_PATH)
        except OSError as e:
            raise NativeLibraryError(f"fastdm_amd: cannot load {LIB_PATH}: {e}") from e
        for name in ("fdm_abi_version", "fdm_source_digest", "fdm_build_flags"):      # identity first, before anything else is bound
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = SIGNATURES[name]
        if handle.fdm_abi_version() != ABI_VERSION:
            raise NativeLibraryError(f"fastdm_amd: {LIB_PATH} was built from another revision of include/fastdm_amd.h (ABI "
                                     f"{handle.fdm_abi_version()}, these bindings expect {ABI_VERSION}); rebuild it with __graft_entry__.build()")
        if handle.fdm_source_digest().decode() != current_source_digest():
            raise NativeLibraryError(f"fastdm_amd: {LIB_PATH} reports source digest {handle.fdm_source_digest().decode()}, the sources are "
                                     f"{current_source_digest()}")
        flags = handle.fdm_build_flags().decode()
        if flags and LIB_DIR == _PRODUCT_LIB_DIR:
            raise NativeLibraryError(f"fastdm_amd: the product library {LIB_PATH} was built with extra flags '{flags}'; A/B variants "
                                     "live under tools/micro/_ab/<name> (FDM_LIB_DIR), never in fastdm_amd/_lib")
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError -> missing export: let it propagate loudly
            fn.restype, fn.argtypes = res, args
        _apply_env_dispatch(handle)
        _lib = handle
    return _lib


def gemm_dispatch_from_env(env=None):
    """(kernel, p4, half_tiles) for fdm_set_gemm_dispatch from FDM_GEMM_KERNEL = auto|p4|w4|w8, FDM_GEMM_P4 = 0|1|2 and
    FDM_GEMM_HALF_TILES = 0|1 (A/B runs and the parity tests' worker processes).  The LIBRARY reads no environment variable: the host
    side translates these once, when it loads the library."""
    env = os.environ if env is None else env
    k = {"": -1, "auto": -1, "w8": 0, "w4": 1, "p4": 2}.get(env.get("FDM_GEMM_KERNEL", ""), None)
    if k is None:
        raise ValueError("FDM_GEMM_KERNEL: auto | p4 | w4 | w8")
    p4 = int(env["FDM_GEMM_P4"]) if env.get("FDM_GEMM_P4", "") != "" else -1
    half = int(env["FDM_GEMM_HALF_TILES"]) if env.get("FDM_GEMM_HALF_TILES", "") != "" else -1
    return k, p4, half


def _apply_env_dispatch(handle):
    k, p4, half = gemm_dispatch_from_env()
    if (k, p4, half) != (-1, -1, -1):
        rc = handle.fdm_set_gemm_dispatch(k, p4, half)
        if rc != 0:
            raise ValueError(f"FDM_GEMM_KERNEL / FDM_GEMM_P4 / FDM_GEMM_HALF_TILES = {k}, {p4}, {half}: rejected by fdm_set_gemm_dispatch ({rc})")
    rows = os.environ.get("FDM_GEMM_TILE_ROWS", "")       # 256 | 224: tile height of the 4-wave kernel (A/B); unset = the policy of gemm.hip tile_rows_224
    if rows != "":
        rc = handle.fdm_set_gemm_tile_rows(int(rows))
        if rc != 0:
            raise ValueError(f"FDM_GEMM_TILE_ROWS = {rows}: 256 | 224")


def provenance() -> dict:
    """What the LOADED library says about itself (bench.py prints this on its JSON line)."""
    h = lib()
    return {"lib_digest": h.fdm_source_digest().decode(), "lib_flags": h.fdm_build_flags().decode(),
            "lib_path": os.path.relpath(LIB_PATH, os.path.dirname(_HERE))}


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().fdm_error_string(rc).decode()
        if rc == -3:
            msg += ": " + lib().fdm_last_hip_error_string().decode()
        raise RuntimeError(f"{what} failed: {msg} (code {rc})")


def ptr(t):
    return None if t is None else t.data_ptr()


_raw_stream = None


def stream(device_index=None):
    """HIP stream handle of torch's current stream (of `device_index`, default: the current device).  Every launch asks for it, so it takes the
    two C calls behind torch.cuda.current_stream() directly: building the torch.cuda.Stream object costs 5-8 us per call (it probes the device
    count), which was a quarter of the host time of an eager SDXL step (cProfile, round 5)."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        raw, cur = getattr(torch._C, "_cuda_getCurrentRawStream", None), getattr(torch._C, "_cuda_getDevice", None)
        if raw is not None and cur is not None:
            _raw_stream = lambda idx: raw(cur() if idx is None else idx)
        else:
            _raw_stream = lambda idx: torch.cuda.current_stream(idx).cuda_stream
    return _raw_stream(device_index)


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


class StreamScratch:
    """A growing scratch buffer per (device, stream), rewritten by every call that uses it.  The rule of `gemm_int8_workspace`:
    a buffer handed out while its stream is CAPTURING has its address baked into a HIP graph, so the key is pinned from then on -- when a
    later, larger request on a stream with the same handle (torch reuses its 32 pooled handles) replaces a pinned buffer, the old one is
    RETIRED, not freed: a replay of the old graph keeps writing into memory this cache still owns.  Growth during a capture raises (the
    allocation would land in the graph's private pool): warm the stream up with the largest shape first."""

    def __init__(self, what, min_bytes=0):
        self.what, self.min_bytes = what, min_bytes
        self.buf, self.pinned, self.retired = {}, set(), []

    def get(self, device, need):
        import torch
        dev = device.index if device.index is not None else torch.cuda.current_device()
        key = (dev, int(stream(dev)))
        capturing = torch.cuda.is_current_stream_capturing()
        ws = self.buf.get(key)
        if ws is None or ws.numel() < need:
            if capturing:
                raise RuntimeError(f"fastdm_amd: the {self.what} of this stream must be allocated (at its largest size) before graph capture")
            if ws is not None and key in self.pinned:
                self.retired.append(ws)
            ws = torch.empty(max(int(need), self.min_bytes), dtype=torch.uint8, device=device)
            self.buf[key] = ws
        if capturing:
            self.pinned.add(key)
        return ws


_attn_vq_ws = StreamScratch("fp8 attention image workspace")


def attn_pvfp8_workspace(device, batch, sk, kv_heads, head_dim, mode=1):
    """Per-(device, stream) buffer for the e4m3 images of fdm_attn_fwd_pvfp8 (mode 1: V) / fdm_attn_fwd_fp8 (mode 2: V and K); rewritten by
    every call; grows to the largest shape seen (StreamScratch: pinned once a capture has seen it)."""
    fn = lib().fdm_attn_fp8_workspace_bytes if mode == 2 else lib().fdm_attn_pvfp8_workspace_bytes
    return _attn_vq_ws.get(device, int(fn(batch, sk, kv_heads, head_dim)))


import collections

_gemm_ws = collections.OrderedDict()      # (device, stream handle) -> buffer, least recently used first
_gemm_ws_pinned = set()                    # keys whose buffer address may be baked into a captured HIP graph: never evicted
_GEMM_WS_MAX = 8                           # unpinned buffers PER DEVICE (8 x 64 MiB): more streams in flight is not a denoise loop


def gemm_int8_workspace(device, pin=False):
    """Scratch of fdm_gemm_w8a8_int8_x2_ws's split-K tail (64 MiB), one buffer per (device, STREAM): the library's merge launch has
    consumed it before the next GEMM's parts write only when both are on one stream, so launches on another stream get their own
    buffer instead of silently sharing partial sums.  Allocate it before any HIP-graph capture (`warm_gemm_int8_workspace`): an
    allocation inside a capture would land in the graph's private pool.  A buffer that was warmed explicitly, or is handed out
    while a capture is in flight, is PINNED: a graph replay writes split-K partials to that address for as long as the graph
    lives, so it must never go back to the allocator."""
    import torch
    dev = device.index if device.index is not None else torch.cuda.current_device()
    key = (dev, int(stream(dev)))
    capturing = torch.cuda.is_current_stream_capturing()
    ws = _gemm_ws.get(key)
    if ws is None:
        if capturing:
            raise RuntimeError("fastdm_amd: the int8 split-K workspace of this stream must be allocated before graph capture "
                               "(call fastdm_amd._native.warm_gemm_int8_workspace(device) on the capture stream first)")
        ws = torch.empty(int(lib().fdm_gemm_int8_workspace_bytes()), dtype=torch.uint8, device=device)
        _gemm_ws[key] = ws
        # bounded per device: a stream handle is not an identity (a destroyed stream's handle can come back), and every stream that
        # ever ran an int8 GEMM used to keep its 64 MiB for good.  The least recently used UNPINNED entry of this device goes; for
        # eager use torch's caching allocator keeps the block alive until the work queued on it has run.
        mine = [k for k in _gemm_ws if k[0] == dev and k not in _gemm_ws_pinned]
        for k in mine[:max(0, len(mine) - _GEMM_WS_MAX)]:
            del _gemm_ws[k]
    else:
        _gemm_ws.move_to_end(key)
    if pin or capturing:
        _gemm_ws_pinned.add(key)
    return ws


def warm_gemm_int8_workspace(device):
    """Allocate (and pin) the int8 split-K workspace of the current stream eagerly (model load / warm-up path before a capture)."""
    return gemm_int8_workspace(device, pin=True)
