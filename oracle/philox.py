"""TEST INFRASTRUCTURE ONLY -- NumPy restatement of the counter-based RNG the HIP engine uses.

The reference (rueberger/MJHMC) draws from NumPy's process-global MT19937 stream in particle
order (mjhmc/misc/utils.py:31-49, mjhmc/samplers/hmc_state.py:24-26,121-129).  A serial global
stream cannot be reproduced by 100k independent device lanes, so the engine's production mode
uses Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11) keyed by
the *global particle id*.  This file restates the same bit-level recipe in NumPy so the oracle
can consume exactly the integers the kernels consume.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it.

Counter / key layout (must match mjhmc_amd/csrc/philox.hpp):
    key = (seed_lo, seed_hi)
    ctr = (particle_id, tick_lo, tick_hi, slot)
    tick : 64-bit attempt counter. tick 0 is the initial-momentum draw; every launch of the
           jump kernel (retries included) uses a fresh tick, so a retried iteration sees fresh
           noise exactly like the reference re-drawing from its stream (markov_jump_hmc.py:376-389)
    slot : j                -> normal pair j  (dims 2j, 2j+1) of the momentum-refresh noise
           0x80000000       -> unit exponentials for the L and F waiting times (words 01, 23)
           0x80000001       -> unit exponential for the R waiting time (words 01); words 23 = the
                               accept uniform of the discrete-time control samplers
           0x80000002       -> flip uniform (words 01) of the control samplers
"""
import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)
_SH32 = np.uint64(32)

SLOT_EXP_LF = 0x80000000
SLOT_EXP_R = 0x80000001
SLOT_FLIP = 0x80000002


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32 with 10 rounds. All inputs broadcastable uint32-valued arrays/ints."""
    c0, c1, c2, c3 = np.broadcast_arrays(*[np.asarray(c, dtype=np.uint64) for c in (c0, c1, c2, c3)])
    c0, c1, c2, c3 = c0.copy(), c1.copy(), c2.copy(), c3.copy()
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> _SH32, p0 & _MASK
        hi1, lo1 = p1 >> _SH32, p1 & _MASK
        n0 = hi1 ^ c1 ^ np.uint64(k0)
        n2 = hi0 ^ c3 ^ np.uint64(k1)
        c0, c1, c2, c3 = n0, lo1, n2, lo0
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return (c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32))


def u53(a, b):
    """Two 32-bit words -> double in (0, 1]: ((a>>5)*2^26 + (b>>6) + 1) * 2^-53 (every step exact)."""
    x = (a.astype(np.uint64) >> np.uint64(5)) * np.uint64(67108864) + (b.astype(np.uint64) >> np.uint64(6))
    return (x.astype(np.float64) + 1.0) * (1.0 / 9007199254740992.0)


class PhiloxStream(object):
    """Per-particle streams keyed by global particle id; mirrors the device draws."""

    def __init__(self, seed, particle_ids):
        seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.k0 = seed & 0xFFFFFFFF
        self.k1 = seed >> 32
        self.pid = np.asarray(particle_ids, dtype=np.uint64)

    def _call(self, tick, slot):
        tick = int(tick)
        return philox4x32_10(self.pid, tick & 0xFFFFFFFF, (tick >> 32) & 0xFFFFFFFF, slot, self.k0, self.k1)

    def normals(self, ndims, tick):
        """(ndims, n) standard normals: Box-Muller on pair j -> dims (2j, 2j+1)."""
        n = self.pid.shape[0]
        npairs = (ndims + 1) // 2
        out = np.empty((2 * npairs, n), dtype=np.float64)
        tick = int(tick)
        slots = np.arange(npairs, dtype=np.uint64)[:, None]
        w0, w1, w2, w3 = philox4x32_10(self.pid[None, :], tick & 0xFFFFFFFF, (tick >> 32) & 0xFFFFFFFF,
                                       slots, self.k0, self.k1)
        u1 = u53(w0, w1)
        u2 = u53(w2, w3)
        r = np.sqrt(-2.0 * np.log(u1))
        ang = 2.0 * np.pi * u2
        out[0::2] = r * np.cos(ang)
        out[1::2] = r * np.sin(ang)
        return out[:ndims]

    def unit_exponentials(self, tick):
        """(3, n): unit-rate exponentials for the L, F, R clocks, e = -log(u)."""
        w0, w1, w2, w3 = self._call(tick, SLOT_EXP_LF)
        v0, v1, _, _ = self._call(tick, SLOT_EXP_R)
        return np.stack([-np.log(u53(w0, w1)), -np.log(u53(w2, w3)), -np.log(u53(v0, v1))])

    def accept_uniforms(self, tick):
        _, _, v2, v3 = self._call(tick, SLOT_EXP_R)
        return u53(v2, v3)

    def flip_uniforms(self, tick):
        w0, w1, _, _ = self._call(tick, SLOT_FLIP)
        return u53(w0, w1)
