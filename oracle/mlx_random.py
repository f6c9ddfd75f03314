"""CPU restatement (numpy) of the random stream the reference draws its noise from.  TEST INFRASTRUCTURE
ONLY: imported by tests/, __graft_entry__.smoke() and nothing else; the product path is
`flux2mi_mlx_random_*` (csrc/elementwise.hip, csrc/capi.cpp).

The reference seeds and draws with `MLXRandom.seed(seed)` + `MLXRandom.normal(shape)`
(Flux2Core/Pipeline/LatentUtils.swift:32-41, Flux2Pipeline.swift:1203-1204, 2313, 2403).  MLXRandom is a
third-party dependency that is NOT in /root/reference: mlx-swift, pinned `exact: "0.31.6"` in
Package.swift:23, which wraps MLX's C++ `mlx/random.cpp` + the `RandomBits` primitive.  What follows is
that published algorithm restated from memory of the MLX sources:

  key(seed)            = uint32[2] {seed >> 32, seed & 0xffffffff}
  global state         : seed(s) sets state = key(s); every keyless draw does
                         (state, sub) = split(state) and draws with `sub`
  split(key)           = bits(shape [2,2], width 4, key) -> rows are the two new keys
  bits(n words, key)   : half = n/2 (out_skip/2); word i and word half+i are the two outputs of
                         threefry2x32(key, counter (i, half+i)); an odd n gets its middle..last word from
                         counter (0-based half, 0)  [see `bits` below for the exact tail]
  uniform(lo, hi)      = min(float32(bits) / float32(0xffffffff), nextafter(1,0)) * (hi - lo) + lo
  normal               = sqrt(2) * erfinv(uniform(nextafter(-1, 0), 1)), erfinv = the single-precision
                         polynomial pair of MLX's erf.cpp (log-based, two branches at |t| > 6.125)

PARITY STATUS: threefry2x32 is pinned against the Random123 known-answer vectors (tests/
test_mlx_random_oracle.py).  The MLX sequencing around it (counter layout, split, the uniform/normal
transforms) is **parity unpinned**: no MLX runtime exists in this image, and the reference holds no golden
noise vectors for it.  The first MLX-side dump of `MLXRandom.seed(42); MLXRandom.normal([8])` settles it.
"""
import numpy as np

_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))
_M32 = np.uint64(0xFFFFFFFF)


def threefry2x32(key, c0, c1):
    """Threefry-2x32, 20 rounds (Salmon et al., Random123).  key: two ints; c0, c1: uint32 arrays."""
    k0, k1 = np.uint32(key[0]), np.uint32(key[1])
    ks = (k0, k1, np.uint32(0x1BD11BDA) ^ k0 ^ k1)
    x0 = (np.asarray(c0, dtype=np.uint32) + ks[0]).astype(np.uint32)
    x1 = (np.asarray(c1, dtype=np.uint32) + ks[1]).astype(np.uint32)
    with np.errstate(over="ignore"):
        for r in range(5):
            for rot in _ROT[r % 2]:
                x0 = (x0 + x1).astype(np.uint32)
                x1 = ((x1 << np.uint32(rot)) | (x1 >> np.uint32(32 - rot))).astype(np.uint32)
                x1 = x1 ^ x0
            x0 = (x0 + ks[(r + 1) % 3]).astype(np.uint32)
            x1 = (x1 + ks[(r + 2) % 3] + np.uint32(r + 1)).astype(np.uint32)
    return x0, x1


def key(seed):
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return (seed >> 32, seed & 0xFFFFFFFF)


def bits(k, n):
    """n uint32 words drawn with one key (RandomBits, width 4)."""
    out = np.zeros(n, dtype=np.uint32)
    half = n // 2
    odd = n % 2
    if half:
        i = np.arange(half, dtype=np.uint32)
        a, b = threefry2x32(k, i, i + np.uint32(half + odd))
        out[:half] = a
        out[half + odd:] = b
    if odd:
        a, _ = threefry2x32(k, np.array([half], dtype=np.uint32), np.array([0], dtype=np.uint32))
        out[half] = a[0]
    return out


def split(k):
    w = bits(k, 4)
    return (int(w[0]), int(w[1])), (int(w[2]), int(w[3]))


def uniform(k, n, lo=0.0, hi=1.0):
    u = bits(k, n).astype(np.float32) / np.float32(4294967295.0)
    u = np.minimum(u, np.nextafter(np.float32(1.0), np.float32(0.0)))
    lo, hi = np.float32(lo), np.float32(hi)
    return (np.float32(hi - lo) * u + lo).astype(np.float32)


def erfinv_f32(a):
    """Single-precision erfinv in the form MLX uses (t = log(1 - a^2), two polynomial branches)."""
    a = np.asarray(a, dtype=np.float32)
    a64 = a.astype(np.float64)
    fma = lambda x, y, z: (x.astype(np.float64) * y.astype(np.float64) + np.float64(z)).astype(np.float32)  # one rounding
    with np.errstate(divide="ignore"):
        t = np.log((1.0 - a64 * a64).astype(np.float32)).astype(np.float32)  # log(fma(a, -a, 1))
    big = np.float32([3.03697567e-10, 2.93243101e-8, 1.22150334e-6, 2.84108955e-5, 3.93552968e-4, 3.02698812e-3,
                      4.83185798e-3, -2.64646143e-1, 8.40016484e-1])
    small = np.float32([5.43877832e-9, 1.43285448e-7, 1.22774793e-6, 1.12963626e-7, -5.61530760e-5, -1.47697632e-4,
                        2.31468678e-3, 1.15392581e-2, -2.32015476e-1, 8.86226892e-1])
    pb = np.full_like(t, big[0])
    for c in big[1:]:
        pb = fma(pb, t, c)
    ps = np.full_like(t, small[0])
    for c in small[1:]:
        ps = fma(ps, t, c)
    return (a * np.where(np.abs(t) > np.float32(6.125), pb, ps)).astype(np.float32)


def normal(k, n):
    lo = np.nextafter(np.float32(-1.0), np.float32(0.0))
    return (np.float32(np.sqrt(2.0)) * erfinv_f32(uniform(k, n, lo, 1.0))).astype(np.float32)


class KeySequence:
    """The global generator state: `MLXRandom.seed` + keyless draws."""

    def __init__(self, seed=0):
        self.state = key(seed)

    def seed(self, seed):
        self.state = key(seed)

    def next_key(self):
        self.state, sub = split(self.state)
        return sub

    def normal(self, shape):
        return normal(self.next_key(), int(np.prod(shape))).reshape(shape)

    def uniform(self, shape, lo=0.0, hi=1.0):
        return uniform(self.next_key(), int(np.prod(shape)), lo, hi).reshape(shape)
