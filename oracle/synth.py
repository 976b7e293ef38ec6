"""Counter-based synthetic data generator (numpy side).

The reference ships no weights or fixtures; its profiling script draws
``torch.rand`` inputs (``examples/profiling/flux_profiling.py:16-23``).  torch's
RNG streams are not stable across builds/devices, so both this oracle and the
product (``fastdm_amd/csrc/synth.hip``) use the same *integer-only* generator:

    z      = splitmix64(seed * GOLDEN + index)            (uint64, wrap-around)
    normal = (sum of the four 16-bit fields of z) - 131070   -> Irwin-Hall(4)
    value  = float32(normal) * float32(scale)  [+ float32(offset)]  -> RNE to bf16
    uniform= float32(z >> 40) * 2^-24                      in [0, 1)

Only IEEE add/mul and integer ops are used, so CPU and GPU agree bit-for-bit.
"""
import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_IH_STD = float(np.sqrt(4.0 * (65536.0 ** 2 - 1.0) / 12.0))  # std of the sum of 4 U{0..65535}


def _splitmix64_inplace(x, tmp):
    """x (uint64 array) <- splitmix64 finaliser of x, using tmp as scratch (no allocations)."""
    np.add(x, GOLDEN, out=x)
    np.right_shift(x, np.uint64(30), out=tmp); np.bitwise_xor(x, tmp, out=x)
    np.multiply(x, np.uint64(0xBF58476D1CE4E5B9), out=x)
    np.right_shift(x, np.uint64(27), out=tmp); np.bitwise_xor(x, tmp, out=x)
    np.multiply(x, np.uint64(0x94D049BB133111EB), out=x)
    np.right_shift(x, np.uint64(31), out=tmp); np.bitwise_xor(x, tmp, out=x)
    return x


class _Scratch:
    """Reusable chunk buffers: fresh allocations are page-fault bound in sandboxes."""
    def __init__(self, chunk):
        self.base = np.arange(chunk, dtype=np.uint64)
        self.x = np.empty(chunk, dtype=np.uint64)
        self.t = np.empty(chunk, dtype=np.uint64)
        self.a = np.empty(chunk, dtype=np.uint64)


_CHUNK = 1 << 18
_scratch = None


def _bits_chunks(seed, n, start=0):
    """Yield (offset, uint64 view of m hashed words) for [start, start+n) in chunks; views are reused."""
    global _scratch
    if _scratch is None:
        _scratch = _Scratch(_CHUNK)
    sc = _scratch
    with np.errstate(over="ignore"):
        s0 = np.uint64(seed) * GOLDEN
        for s in range(0, n, _CHUNK):
            m = min(_CHUNK, n - s)
            x = sc.x[:m]
            np.add(sc.base[:m], s0 + np.uint64(start + s), out=x)
            _splitmix64_inplace(x, sc.t[:m])
            yield s, x


def _bits(seed, n, start=0):
    out = np.empty(n, dtype=np.uint64)
    for s, x in _bits_chunks(seed, n, start):
        out[s:s + x.shape[0]] = x
    return out


def f32_to_bf16_bits(x):
    """float32 ndarray -> uint16 bf16 bit patterns, round-to-nearest-even."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)
    return r.astype(np.uint16)


def bf16_bits_to_f32(b):
    return (np.ascontiguousarray(b, dtype=np.uint16).astype(np.uint32) << np.uint32(16)).view(np.float32)


def normal_scale(std):
    """float32 multiplier that turns the centred Irwin-Hall integer into N(0, std^2)."""
    return np.float32(std / _IH_STD)


def normal_bf16_bits(seed, n, std, offset=0.0):
    """n bf16 values (uint16 bits) ~ offset + N(0, std^2)."""
    out = np.empty(n, dtype=np.uint16)
    sc = normal_scale(std)
    off = np.float32(offset)
    m16 = np.uint64(0xFFFF)
    for s, z in _bits_chunks(seed, n):
        m = z.shape[0]
        a = _scratch.a[:m]; t = _scratch.t[:m]
        np.bitwise_and(z, m16, out=a)
        np.right_shift(z, np.uint64(16), out=t); np.bitwise_and(t, m16, out=t); np.add(a, t, out=a)
        np.right_shift(z, np.uint64(32), out=t); np.bitwise_and(t, m16, out=t); np.add(a, t, out=a)
        np.right_shift(z, np.uint64(48), out=t); np.add(a, t, out=a)
        v = (a.astype(np.int32) - np.int32(131070)).astype(np.float32) * sc
        if offset != 0.0:
            v = v + off
        out[s:s + m] = f32_to_bf16_bits(v)
    return out


def uniform_bf16_bits(seed, n):
    """n bf16 values (uint16 bits) ~ U[0,1) (24-bit uniform rounded to bf16; may round to 1.0)."""
    out = np.empty(n, dtype=np.uint16)
    for s, z in _bits_chunks(seed, n):
        m = z.shape[0]
        t = _scratch.t[:m]
        np.right_shift(z, np.uint64(40), out=t)
        v = t.astype(np.float32) * np.float32(2.0 ** -24)
        out[s:s + m] = f32_to_bf16_bits(v)
    return out


def key_seed(name, base_seed=0):
    """Stable 31-bit stream id for a tensor name (FNV-1a over the utf-8 name, mixed with base_seed)."""
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    h ^= (base_seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    return int((h ^ (h >> 33)) & 0x7FFFFFFF)
