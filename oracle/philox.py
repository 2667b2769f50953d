"""Philox4x32-10 counter-based streams (numpy) -- TEST INFRASTRUCTURE ONLY.

The reference draws from numpy's global MT19937 (``np.random.multivariate_normal``,
``metropolisengine/metropolis_engine.py:268,300``) and CPython's ``random.uniform``
(``:335``); neither can be reproduced on a GPU, and the reference consumes the
uniform only on uphill moves (``:329-335``) so its stream position depends on
history.  The build therefore defines its own stream, addressed by
``(seed, global chain id, step index, word)``, identical on every GPU count:

    key     = (seed & 0xffffffff, seed >> 32)
    counter = (chain & 0xffffffff, chain >> 32, step & 0xffffffff,
               ((step >> 32) & 0xffff) << 16 | block)

One step of a chain with ``ND = nr + 2*nc`` Gaussian degrees of freedom uses
words ``0 .. W-1`` (``W = 2*ceil(ND/2)``) as Box-Muller pairs and word ``W`` for
the accept uniform; word ``w`` is output ``w % 4`` of Philox block ``w // 4``.

Philox4x32-10 is the generator behind hiprand's ``HIPRAND_RNG_PSEUDO_PHILOX4_32_10``
(Salmon et al., SC'11); this file is pinned by the Random123 known-answer
vectors in ``tests/test_oracle_philox.py``.
"""
import numpy as np

PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = 0x9E3779B9
PHILOX_W1 = 0xBB67AE85
_MASK32 = np.uint64(0xFFFFFFFF)
_SH32 = np.uint64(32)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Ten Philox rounds on arrays of 32-bit words (held as uint64)."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & _MASK32 for c in (c0, c1, c2, c3))
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for rnd in range(10):
        p0 = PHILOX_M0 * c0
        p1 = PHILOX_M1 * c2
        hi0, lo0 = p0 >> _SH32, p0 & _MASK32
        hi1, lo1 = p1 >> _SH32, p1 & _MASK32
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0 = (k0 + PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + PHILOX_W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def step_block(seed, chain_ids, step, block):
    """The four 32-bit outputs of Philox block ``block`` of step ``step``."""
    chain_ids = np.asarray(chain_ids, dtype=np.uint64)
    seed = int(seed)
    step = int(step)
    c0 = chain_ids & _MASK32
    c1 = chain_ids >> _SH32
    c2 = np.full_like(chain_ids, step & 0xFFFFFFFF)
    c3 = np.full_like(chain_ids, (((step >> 32) & 0xFFFF) << 16) | (int(block) & 0xFFFF))
    return philox4x32_10(c0, c1, c2, c3, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)


def step_words(seed, chain_ids, step, n_words):
    """Words ``0 .. n_words-1`` of one step, shape ``[n_chains, n_words]`` (uint64 holding u32)."""
    chain_ids = np.asarray(chain_ids, dtype=np.uint64)
    out = np.empty((chain_ids.shape[0], n_words), dtype=np.uint64)
    for block in range((n_words + 3) // 4):
        words = step_block(seed, chain_ids, step, block)
        for j in range(4):
            w = 4 * block + j
            if w < n_words:
                out[:, w] = words[j]
    return out


def unit_open(words):
    """Map a u32 word to the open unit interval: ``(w + 0.5) * 2**-32``."""
    return (words.astype(np.float64) + 0.5) * (1.0 / 4294967296.0)


def n_normal_words(n_normals):
    return 2 * ((n_normals + 1) // 2)


def step_draws(seed, chain_ids, step, n_normals):
    """Standard normals ``[n_chains, n_normals]`` and the accept uniform ``[n_chains]`` of one step.

    Box-Muller on word pairs ``(2p, 2p+1)``: ``g[2p] = r cos(t)``, ``g[2p+1] = r sin(t)``,
    ``r = sqrt(-2 ln u1)``, ``t = 2 pi u2``.
    """
    w_norm = n_normal_words(n_normals)
    words = step_words(seed, chain_ids, step, w_norm + 1)
    u = unit_open(words)
    n_chains = u.shape[0]
    g = np.empty((n_chains, w_norm), dtype=np.float64)
    r = np.sqrt(-2.0 * np.log(u[:, 0:w_norm:2]))
    theta = (2.0 * np.pi) * u[:, 1:w_norm:2]
    g[:, 0::2] = r * np.cos(theta)
    g[:, 1::2] = r * np.sin(theta)
    return g[:, :n_normals], u[:, w_norm]
