"""TEST INFRASTRUCTURE ONLY — Philox4x32-10 in plain Python, mirroring the device reset RNG.

The reference draws its resets from NumPy's process-global MT19937
(``multiagent/environment.py:192-196``); the HIP reset kernel uses a counter-based
Philox stream instead (SURVEY.md §7 hard part 4).  To compare the device reset with
the oracle bit for bit, the oracle's reset can draw from this same stream:
key = (seed_lo, seed_hi), counter = (draw_index, env, episode, TAG); every draw call
consumes one 128-bit block = two doubles built like NumPy's ``random_double``
(53 high bits of a 64-bit word times 2**-53).
"""
M0, M1 = 0xD2511F53, 0xCD9E8D57
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF
TAG = 0x464D4152  # 'FMAR'


def philox4x32_10(ctr, key):
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for r in range(10):
        if r:
            k0 = (k0 + W0) & MASK
            k1 = (k1 + W1) & MASK
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> 32, p0 & MASK
        hi1, lo1 = p1 >> 32, p1 & MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & MASK, lo1, (hi0 ^ c3 ^ k1) & MASK, lo0
    return c0, c1, c2, c3


def block_to_doubles(blk):
    a = ((blk[1] << 32) | blk[0]) >> 11
    b = ((blk[3] << 32) | blk[2]) >> 11
    return a * (1.0 / 9007199254740992.0), b * (1.0 / 9007199254740992.0)


class PhiloxStream:
    """Sequential draw stream of one (seed, env, episode)."""

    def __init__(self, seed, env, episode):
        seed, env, episode = int(seed), int(env), int(episode)  # never NumPy ints: products must not wrap
        self.key = (seed & MASK, (seed >> 32) & MASK)
        self.env = env & MASK
        self.episode = episode & MASK
        self.idx = 0

    def _next(self):
        blk = philox4x32_10((self.idx & MASK, self.env, self.episode, TAG), self.key)
        self.idx += 1
        return block_to_doubles(blk)

    # interface used by oracle resets -------------------------------------------------
    def uniform_pair(self, lo, hi):
        import numpy as np
        u0, u1 = self._next()
        return np.array([lo + (hi - lo) * u0, lo + (hi - lo) * u1])

    def uniform(self, lo, hi):
        u0, _ = self._next()
        return lo + (hi - lo) * u0

    def choice_hv(self):
        u0, _ = self._next()
        return 'H' if u0 < 0.5 else 'V'
