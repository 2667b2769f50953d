"""Pins oracle/philox.py: Random123 known-answer vectors for philox4x32-10 and stream-shape properties."""
import numpy as np

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
