"""Portable, counter-based seeded generator (splitmix64 + Box-Muller).

Used on BOTH sides of every parity check (reference shim in the build container, the oracle and
the HIP path on the GPU box) so that weights/inputs are bit-identical without shipping them and
without relying on ``torch.manual_seed`` streams ("portable seeded generator").
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _stream(seed: int, tag: int, n: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([seed * 1000003 + tag], dtype=np.uint64))[0]
        idx = np.arange(n, dtype=np.uint64)
        return _splitmix64(idx * np.uint64(0xD1342543DE82EF95) + base)


def uniform(seed: int, tag: int, n: int) -> np.ndarray:
    """float64 uniforms in (0, 1)."""
    bits = _stream(seed, tag, n) >> np.uint64(11)
    return (bits.astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal(seed: int, tag: int, shape, std: float = 1.0) -> np.ndarray:
    """float32 N(0, std^2) of the given shape."""
    n = int(np.prod(shape)) if len(shape) else 1
    m = (n + 1) // 2
    u1 = uniform(seed, 2 * tag, m)
    u2 = uniform(seed, 2 * tag + 1, m)
    r = np.sqrt(-2.0 * np.log(u1))
    out = np.empty(2 * m, dtype=np.float64)
    out[0::2] = r * np.cos(2.0 * np.pi * u2)
    out[1::2] = r * np.sin(2.0 * np.pi * u2)
    return (out[:n] * std).astype(np.float32).reshape(shape)


def integers(seed: int, tag: int, shape, lo: int, hi: int) -> np.ndarray:
    """int64 uniform integers in [lo, hi)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform(seed, tag, n)
    return (lo + np.floor(u * (hi - lo)).astype(np.int64)).reshape(shape)


def name_tag(name: str) -> int:
    """Stable 31-bit tag for a parameter name (FNV-1a)."""
    h = 0x811C9DC5
    for c in name.encode():
        h = ((h ^ c) * 0x01000193) & 0xFFFFFFFF
    return h & 0x7FFFFFFF
