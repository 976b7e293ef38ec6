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


# ------------------------------------------------------------------ provenance: the binary says what it was built from
def _sha(path):
    import hashlib
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def test_library_carries_the_digest_of_the_sources_it_was_built_from():
    """No stamp file: the digest compiled into libfastdm_amd.so (read from the file before dlopen AND returned by
    fdm_source_digest()) equals the digest of the sources in the tree, and the product library has no extra build flags."""
    lib = N.lib()
    want = N.current_source_digest()
    emb, flags = N.embedded_provenance()
    assert emb == want == lib.fdm_source_digest().decode()
    assert flags == "" == lib.fdm_build_flags().decode()
    assert N.provenance() == {"lib_digest": want, "lib_flags": "", "lib_path": "fastdm_amd/_lib/libfastdm_amd.so"}
    assert not os.path.exists(N.LIB_PATH + ".srcdigest")


def test_build_script_and_python_agree_on_the_digest_rule():
    """build.sh computes the digest with sha256sum, _native with hashlib: the same manifest rule (else every import rebuilds)."""
    import subprocess
    csrc = os.path.join(ROOT, "fastdm_amd", "csrc")
    sh = r'''
for f in "$1"/*.hip "$1"/*.h "$1"/*.sh "$1"/*.inc "$1"/*.py "$1/../../include/fastdm_amd.h"; do [ -f "$f" ] && printf '%s %s\n' "$(basename "$f")" "$f"; done \
 | LC_ALL=C sort | while read -r base path; do printf '%s  %s\n' "$(sha256sum < "$path" | cut -d' ' -f1)" "$base"; done | sha256sum | cut -d' ' -f1'''
    out = subprocess.run(["bash", "-c", sh, "x", csrc], capture_output=True, text=True, check=True).stdout.strip()
    assert out == N.current_source_digest()
    assert "digest_manifest | sha256sum" in open(os.path.join(csrc, "build.sh")).read()


def test_build_script_refuses_extra_flags_in_the_product_directory():
    import subprocess
    before = _sha(N.LIB_PATH)
    r = subprocess.run(["bash", N.BUILD_SCRIPT, "-DFDM_GEMM_STAMPS"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "never go into" in r.stderr, r.stderr[-500:]
    r = subprocess.run(["bash", N.BUILD_SCRIPT], capture_output=True, text=True, timeout=120, env=dict(os.environ, FDM_SKIP_AUDIT="1"))
    assert r.returncode == 2, "a product build without its audits must be refused too"
    assert _sha(N.LIB_PATH) == before


def test_ab_variant_builds_beside_the_product_library_and_says_what_it_is(tmp_path):
    """An A/B build (extra -D flags) goes to its own directory under tools/micro/_ab/ and records its flags; the product library
    is byte for byte untouched.  (Two small units only -- the full kernels take minutes; the routing is what is tested.)"""
    import shutil
    import subprocess
    if not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc here")
    N.lib()
    before = _sha(N.LIB_PATH)
    out = os.path.join(ROOT, "tools", "micro", "_ab", "test_routing")
    shutil.rmtree(out, ignore_errors=True)
    try:
        r = subprocess.run(["bash", N.BUILD_SCRIPT, "-DFDM_GEMM_STAMPS", "-DFDM_AB_TEST=1"], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, FDM_OUT=out, FDM_BUILD_UNITS="capi"))
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        emb, flags = N.embedded_provenance(os.path.join(out, "libfastdm_amd.so"))
        assert emb == N.current_source_digest()
        assert "-DFDM_GEMM_STAMPS" in flags and "-DFDM_AB_TEST=1" in flags and "units: capi" in flags
        assert _sha(N.LIB_PATH) == before and N.embedded_provenance()[1] == ""
        # and a flagged binary is refused as THE product library (here: the variant's directory declared to be the product's)
        code = """
from fastdm_amd import _native as N
N._PRODUCT_LIB_DIR = N.LIB_DIR
try:
    N.lib()
except N.NativeLibraryError as e:
    assert "extra flags" in str(e) and "FDM_GEMM_STAMPS" in str(e), str(e)
    print("REFUSED")
"""
        r = _run_native_snippet(code, out)
        assert r.returncode == 0 and "REFUSED" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]
    finally:
        shutil.rmtree(out, ignore_errors=True)


def _run_native_snippet(code, lib_dir, **env):
    import subprocess
    import sys
    e = dict(os.environ, FDM_LIB_DIR=str(lib_dir), **env)
    return subprocess.run([sys.executable, "-c", f"import sys; sys.path.insert(0, {ROOT!r})\n" + code], capture_output=True, text=True, timeout=600, env=e)


def _stale_copy(tmp_path):
    """A copy of the product library whose embedded digest has been overwritten (= a binary built from other sources)."""
    import shutil
    N.lib()
    dst = tmp_path / "libfastdm_amd.so"
    shutil.copy(N.LIB_PATH, dst)
    blob = dst.read_bytes()
    marker = b"@(#)fdm_source_digest="
    at = blob.index(marker) + len(marker)
    dst.write_bytes(blob[:at] + b"f" * 64 + blob[at + 64:])
    assert N.embedded_provenance(str(dst))[0] == "f" * 64
    return dst


def test_stale_library_is_rebuilt_before_it_is_loaded(tmp_path):
    """The loader compares the digest INSIDE the binary with the sources' before dlopen and rebuilds on mismatch.  (The build step is
    replaced by a recorder that installs the good library: the logic is under test, not the 3-minute compile.)"""
    _stale_copy(tmp_path)
    code = f"""
import shutil, types
from fastdm_amd import _native as N
calls = []
def fake_build():
    calls.append(1)
    assert N.embedded_provenance()[0] == "f" * 64          # still the stale file, and it has NOT been loaded
    shutil.copy({N.LIB_PATH!r}, N.LIB_PATH)
    return types.SimpleNamespace(returncode=0, stdout="", stderr="")
N._run_build = fake_build
h = N.lib()
assert calls == [1], calls
assert h.fdm_source_digest().decode() == N.current_source_digest()
print("REBUILT", N.LIB_PATH)
"""
    r = _run_native_snippet(code, tmp_path)
    assert r.returncode == 0 and "REBUILT" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]


def test_stale_library_without_a_compiler_is_refused_not_loaded(tmp_path):
    _stale_copy(tmp_path)
    code = """
from fastdm_amd import _native as N
try:
    N.lib()
except N.NativeLibraryError as e:
    assert "built from other sources" in str(e) and "ffffffff" in str(e), str(e)
    print("REFUSED")
"""
    for env in ({"HIPCC": "/nonexistent/hipcc"}, {"FDM_NO_REBUILD": "1"}):
        r = _run_native_snippet(code, tmp_path, **env)
        assert r.returncode == 0 and "REFUSED" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]


def test_bench_line_carries_the_library_identity():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'if prov["lib_flags"] and not args.allow_build_flags' in src and '"experimental": True' in src
    assert "**prov" in src            # lib_digest / lib_flags / lib_path on the JSON line


def test_stream_scratch_never_frees_a_buffer_a_capture_has_seen(monkeypatch):
    """the per-(device, stream) scratch caches (GroupNorm statistics, fp8 attention images) follow the int8 split-K
    workspace's rule -- a buffer handed out during a capture is pinned; when a later, larger request on the same stream handle replaces
    it the old tensor is retired (kept), and growth during a capture raises."""
    import torch
    state = {"capturing": False}
    monkeypatch.setattr(N, "stream", lambda idx=None: 1234)
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: state["capturing"])
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    sc = N.StreamScratch("test scratch", min_bytes=64)
    dev = torch.device("cpu")
    a = sc.get(dev, 100)
    assert a.numel() == 100 and sc.get(dev, 50) is a and not sc.retired            # eager: reuse, grow freely
    b = sc.get(dev, 200)
    assert b.numel() == 200 and not sc.retired                                      # the unpinned predecessor simply goes
    state["capturing"] = True
    assert sc.get(dev, 200) is b and (0, 1234) in sc.pinned                          # handed out during a capture: pinned
    with pytest.raises(RuntimeError):
        sc.get(dev, 300)                                                             # growth inside a capture
    state["capturing"] = False
    c = sc.get(dev, 300)
    assert c.numel() == 300 and sc.retired == [b] and sc.get(dev, 10) is c           # the captured address stays owned by the cache


def test_gemm_dispatch_env_is_translated_on_the_host_side():
    """The library reads no environment variable: FDM_GEMM_KERNEL / FDM_GEMM_P4 / FDM_GEMM_HALF_TILES become one
    fdm_set_gemm_dispatch call when _native loads the library; csrc has no getenv outside FDM_DEBUG_BUILD / A/B blocks."""
    assert N.gemm_dispatch_from_env({}) == (-1, -1, -1)
    assert N.gemm_dispatch_from_env({"FDM_GEMM_KERNEL": "w8", "FDM_GEMM_HALF_TILES": "1"}) == (0, -1, 1)
    assert N.gemm_dispatch_from_env({"FDM_GEMM_KERNEL": "w4", "FDM_GEMM_P4": "0"}) == (1, 0, -1)
    assert N.gemm_dispatch_from_env({"FDM_GEMM_P4": "2"}) == (-1, 2, -1)
    with pytest.raises(ValueError):
        N.gemm_dispatch_from_env({"FDM_GEMM_KERNEL": "fast"})
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in ("gemm.hip", "attention.hip", "elementwise.hip", "capi.hip"):
        src = open(os.path.join(root, "fastdm_amd", "csrc", f)).read()
        depth_ok = True
        lines = src.split("\n")
        guard = []          # stack of preprocessor conditions
        for ln in lines:
            t = ln.strip()
            if t.startswith(("#ifdef", "#ifndef", "#if ")):
                guard.append(t)
            elif t.startswith("#endif") and guard:
                guard.pop()
            if "getenv(" in t and not t.startswith("//"):
                assert any(("FDM_DEBUG_BUILD" in g or "ENV_KNOBS" in g or "FDM_GEMM_AB" in g) for g in guard), f"{f}: getenv outside a debug / A-B block: {t}"
