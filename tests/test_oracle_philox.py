"""Pins oracle/philox.py: Random123 known-answer vectors for philox4x32-10 and stream-shape properties."""
import numpy as np
import pytest

from oracle import philox

# Random123 kat_vectors: philox4x32 10 <ctr x4> <key x2> <expected x4>
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_philox_known_answers():
    for ctr, key, want in KAT:
        got = philox.philox4x32_10(*[np.array([c], dtype=np.uint64) for c in ctr], key[0], key[1])
        assert tuple(int(g[0]) for g in got) == want


def test_counter_layout():
    chains = np.array([0, 1, (1 << 32) + 5], dtype=np.uint64)
    seed = (0xa4093822) | (0x299f31d0 << 32)
    step = (3 << 32) | 0x13198a2e
    got = philox.step_block(seed, chains, step, block=2)
    for i, chain in enumerate(chains):
        want = philox.philox4x32_10(np.array([int(chain) & 0xffffffff], dtype=np.uint64),
                                    np.array([int(chain) >> 32], dtype=np.uint64),
                                    np.array([0x13198a2e], dtype=np.uint64),
                                    np.array([(3 << 16) | 2], dtype=np.uint64), 0xa4093822, 0x299f31d0)
        assert [int(g[i]) for g in got] == [int(w[0]) for w in want]


def test_draw_shapes_and_moments():
    chains = np.arange(20000, dtype=np.uint64)
    g, u = philox.step_draws(seed=99, chain_ids=chains, step=7, n_normals=5)
    assert g.shape == (20000, 5) and u.shape == (20000,)
    assert np.all((u > 0) & (u < 1))
    assert abs(g.mean()) < 0.02 and abs(g.var() - 1) < 0.03
    assert abs(u.mean() - 0.5) < 0.01
    # the accept uniform is word W = 2*ceil(5/2) = 6
    words = philox.step_words(99, chains, 7, 7)
    assert np.array_equal(philox.unit_open(words[:, 6]), u)
    # streams of different steps / chains are distinct
    g2, _ = philox.step_draws(seed=99, chain_ids=chains, step=8, n_normals=5)
    assert not np.allclose(g, g2)


def test_stream_contract_equals_rocrand_philox_engine():
    """The vendor generator named by BASELINE.json ("hiprand Philox"): rocRAND's philox4x32_10_engine, evaluated on the
    host through oracle/c/librocrand_check.so, yields exactly the build's block for
    (seed, subsequence = step_lo | word3 << 32, offset = 4 * chain) -- see oracle/c/rocrand_check.cpp."""
    import ctypes
    import os
    import subprocess
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "c")
    lib_path = os.path.join(here, "librocrand_check.so")
    if not os.path.exists(lib_path):
        done = subprocess.run(["make", "-C", here, "-s", "librocrand_check.so"], capture_output=True, text=True)
        if done.returncode != 0:
            pytest.skip("rocRAND headers / hipcc not available: " + done.stderr[-200:])
    lib = ctypes.CDLL(lib_path)
    lib.me_rocrand_philox_block.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32,
                                            ctypes.POINTER(ctypes.c_uint32)]
    out = (ctypes.c_uint32 * 4)()
    rng = np.random.default_rng(11)
    cases = [(2026, 0, 0, 0), (2026, (1 << 20) - 1, 999, 4), (0xDEADBEEFCAFEF00D, (1 << 40) + 12345, (1 << 33) + 9, 16),
             (1, 2 ** 62 - 1, 2 ** 47 + 5, 255)]
    cases += [(int(rng.integers(0, 2 ** 63)), int(rng.integers(0, 2 ** 62)), int(rng.integers(0, 2 ** 48)),
               int(rng.integers(0, 2 ** 16))) for _ in range(200)]
    for seed, chain, step, block in cases:
        lib.me_rocrand_philox_block(seed, chain, step, block, out)
        want = philox.step_block(seed, [chain], step, block)
        assert list(out) == [int(w[0]) for w in want], (seed, chain, step, block)
