"""CPU-only: the C-ABI library loads and exports every symbol include/fastdm_amd.h declares,
and the ctypes table mirrors the header.  No compute calls (no GPU here)."""
import pytest
import os
import re

from fastdm_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_prototypes():
    src = open(os.path.join(ROOT, "include", "fastdm_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|int64_t|const char\*)\s+(fdm_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = [a.strip() for a in m.group(3).split(",") if a.strip() and a.strip() != "void"]
        protos[m.group(2)] = args
    return protos


def test_library_builds_and_loads():
    path = N.build()
    assert os.path.exists(path)
    lib = N.lib()
    assert lib.fdm_version() >= 100
    assert lib.fdm_error_string(0) == b"ok"
    assert b"unsupported" in lib.fdm_error_string(-2)


def test_abi_version_stamp_agrees_everywhere():
    """include/fastdm_amd.h, the built library and the ctypes table carry the same FDM_ABI_VERSION: a library built from another
    revision of the header is refused at load time instead of receiving arguments at the wrong positions."""
    src = open(os.path.join(ROOT, "include", "fastdm_amd.h")).read()
    hdr = int(re.search(r"#define\s+FDM_ABI_VERSION\s+(\d+)", src).group(1))
    assert hdr == N.ABI_VERSION == N.lib().fdm_abi_version()


def test_every_header_symbol_is_exported_and_bound():
    protos = header_prototypes()
    assert len(protos) >= 19
    lib = N.lib()
    assert set(protos) == set(N.SIGNATURES), set(protos) ^ set(N.SIGNATURES)
    for name, args in protos.items():
        assert hasattr(lib, name), f"{name} not exported"
        assert len(args) == len(N.SIGNATURES[name][1]), f"{name}: header has {len(args)} args, binding {len(N.SIGNATURES[name][1])}"


def test_argument_kinds_match_header():
    """pointer <-> c_void_p, int64_t <-> c_int64, int <-> c_int, float <-> c_float, uint32_t <-> c_uint32."""
    import ctypes as C
    kind = {C.c_void_p: "ptr", C.c_int64: "i64", C.c_int: "i32", C.c_float: "f32", C.c_uint32: "u32"}
    for name, args in header_prototypes().items():
        for decl, ct in zip(args, N.SIGNATURES[name][1]):
            if "*" in decl or decl.startswith("fdm_stream_t"):
                want = "ptr"
            elif decl.startswith("int64_t"):
                want = "i64"
            elif decl.startswith("uint32_t"):
                want = "u32"
            elif decl.startswith("float"):
                want = "f32"
            else:
                want = "i32"
            assert kind[ct] == want, f"{name}: '{decl}' bound as {kind[ct]}"


def test_ops_reject_cpu_tensors_loudly():
    import pytest
    import torch
    from fastdm_amd.kernel import operators_set as ops
    x = torch.zeros(4, 128, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):
        ops.rms_norm(x, torch.ones(128, dtype=torch.bfloat16), 1e-6)
    with pytest.raises(RuntimeError):
        ops.quantize_to_fp8(x)


@pytest.mark.parametrize("script", ["audit_gemm_asm.sh", "audit_attn_asm.sh"])
def test_hand_scheduled_kernels_pass_their_asm_audit(script):
    """gemm_w4_kernel and attn_fwd64_kernel keep accumulators in registers only their inline asm touches.  The audits compile the
    kernels to assembly (hipcc cross-compiles without a GPU) and fail on any compiler-generated AGPR reference, scratch access
    or real call in them -- a spill into a[..] corrupts accumulators silently, and did once reach a commit in a rarely taken edge
    path before this test existed."""
    import shutil
    import subprocess
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc here")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "tools", "micro", script)
    N.lib()                                           # (re)builds if the sources changed: build.sh leaves this build's device assembly
    asm = os.path.join(root, "fastdm_amd", "_lib", ("gemm" if "gemm" in script else "attention") + ".gfx950.s")
    env = dict(os.environ, FDM_AUDIT_ASM=asm) if os.path.exists(asm) else dict(os.environ)   # else the audit compiles (minutes)
    r = subprocess.run(["bash", path], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
